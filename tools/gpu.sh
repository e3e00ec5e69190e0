#!/bin/bash
# tools/gpu.sh - everything that is run on the GPU box, one script:   gpurun -- bash tools/gpu.sh <tag> <action> [args...]
# Outputs go to gpurun_out/<tag>/. Development aid: nothing here is part of the product or of the test suite.
#
#   driver                       what the driver runs at round end: pytest -m gpu -x, smoke(), the default bench line
#   tests [pytest args]          a pytest selection (default: the whole -m gpu suite, no -x)
#   bench [bench.py args]        one bench line -> bench.jsonl
#   steps <config> <n>           per-step phase times over n steps (periodic slow steps: two streams on one hardware queue)
#   knob <ENV> <config> v1 v2 .. one bench line per value of an environment variable (GPU_MAX_HW_QUEUES, ...)
#   profiles <config>            bench lines + rocprofv3 kernel stats + the two PMC passes of one configuration (-> profiles/)
#   stats <config>               rocprofv3 --kernel-trace --stats of one configuration (kernel_stats.csv)
#   profile-build <config> [ENV=v ...]   a -DQVZ_PROFILE build in /tmp (cycle counters of the coder / model / Lloyd-Max kernels)
#   streams                      stream-count sweep of the default workload (bits per symbol against ms per step)
#   cli                          the full-size file-to-file tests through the qvz binary
#   e2e [pause]                  the qvz binary file to file at 100M x 150, several runs, `pause` seconds between processes
#   pmf                          PMF parity tests, then the pass's roofline fraction with one and with three clusters
#   codebook                     codebook parity tests, then the phase with the device's entropies and with the host's libm
#   decoder                      decoder parity tests, then the full-size round trip
#   decoder-profile <config>     a -DQVZ_PROFILE build in /tmp: cycles per symbol of the decoder's chain (table wait / the rest)
#   n2 [reads]                   the driver's N = 2 launch line of bench.py with both ranks on this box's one GPU (gloo), then N = 1 beside it
#   host                         cores, memory, tmpfs and I/O rates of the box (tools/io_bench.c)
tag=${1:?tag}; action=${2:?action}; shift 2
root=$(pwd)
out=$root/gpurun_out/$tag
mkdir -p $out
export TMPDIR=/tmp
line() {   # one bench line summarised: ms per step, per-step times, phases
  python3 -c "
import json,sys
lines=[l for l in sys.stdin.read().splitlines() if l.startswith('{')]
d=json.loads(lines[-1]); print(d['ms_per_step'], d['step_ms'], 'bits/sym', d['bits_per_symbol'])
for k,v in d['step_phase_ms'].items(): print('   ', k, v)"
}
case $action in
driver)
  timeout -k 10 1100 python3 -m pytest tests -x -q -m gpu --durations=15 > $out/pytest.log 2>&1; rc=$?
  echo "pytest rc=$rc" | tee -a $out/pytest.log; grep -E "passed|failed|FAILED|ERROR" $out/pytest.log | tail -15
  [ $rc -eq 0 ] || exit $rc
  timeout -k 10 300 python3 -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" > $out/smoke.log 2>&1 || { tail -20 $out/smoke.log; exit 1; }
  tail -2 $out/smoke.log
  timeout -k 10 600 python3 bench.py > $out/bench.jsonl 2> $out/bench.err; echo "bench rc=$?"; tail -c 600 $out/bench.err ;;
tests)
  timeout -k 10 1100 python3 -m pytest ${@:-tests -m gpu} -q -s --durations=10 > $out/pytest.log 2>&1; echo "pytest rc=$?"; tail -25 $out/pytest.log ;;
bench)
  timeout -k 10 900 python3 bench.py "$@" > $out/bench.jsonl 2> $out/bench.err; echo "bench rc=$?"; tail -c 400 $out/bench.err; line < $out/bench.jsonl ;;
steps)
  timeout -k 10 600 python3 bench.py --config ${1:-2} --steps ${2:-13} --warmup 2 --no-cpu-baseline --no-extras 2> $out/err.txt | tee $out/steps.jsonl | line ;;
knob)
  env_name=$1; cfg=$2; shift 2
  for v in "$@"; do
    echo "== config $cfg, $env_name=$v"
    env $env_name=$v QVZ_CB_DEBUG=1 timeout -k 10 400 python3 bench.py --config $cfg --steps 4 --warmup 2 --no-cpu-baseline --no-extras 2> $out/err_$v.txt | line || { tail -5 $out/err_$v.txt; exit 1; }
  done 2>&1 | tee $out/knob_$env_name.txt ;;
profiles)
  bash tools/collect_profiles.sh $tag ${1:-2} ;;
stats)
  cd /tmp
  timeout -k 10 400 rocprofv3 --kernel-trace --stats --output-format csv -d $out/stats -- python3 $root/bench.py --config ${1:-2} --steps 3 --warmup 1 --no-cpu-baseline --no-extras > $out/stats.log 2>&1
  find $out -name "*kernel_stats.csv" -exec cp {} $out/kernel_stats.csv \; ; find $out -name "*kernel_trace.csv" -delete; rm -rf $out/stats
  head -30 $out/kernel_stats.csv | cut -c1-70,160-330 ;;
profile-build)
  cfg=${1:-2}; shift
  cp -r . /tmp/prof_build && cd /tmp/prof_build && touch qvz_amd/csrc/*.hip &&
  make -s HIPFLAGS="--offload-arch=gfx950 -O3 -ffp-contract=off -fPIC -std=c++17 -Iinclude -Iqvz_amd/csrc -DQVZ_PROFILE" > /dev/null 2>&1 || exit 1
  env "$@" QVZ_CODER_DEBUG=1 QVZ_CB_DEBUG=1 timeout -k 10 400 python3 bench.py --config $cfg --steps 2 --warmup 1 --no-cpu-baseline --no-extras 2> $out/profile_build.txt | line
  grep -E "cycles|lm_design|segment|streams made|host side|before the first chunk" $out/profile_build.txt | tail -30 ;;
streams)
  : > $out/sweep.jsonl
  for s in 1024 2048 4096 8192 16384; do
    timeout -k 10 400 python3 bench.py --shards $s --steps 2 --warmup 1 --no-cpu-baseline --no-extras >> $out/sweep.jsonl 2>> $out/sweep.err || exit 1
  done
  python3 -c "
import json
for l in open('$out/sweep.jsonl'):
    d=json.loads(l); print(d['config']['coder'].split()[0], d['ms_per_step'], d['bits_per_symbol'], d['phase_seconds']['coder'])" ;;
cli)
  timeout -k 10 900 python3 -m pytest tests/test_gpu_cli_fullsize.py -x -q -s > $out/pytest_cli.log 2>&1; echo "pytest rc=$?"
  grep -E "encode file|decode file|input text|passed|failed|Error|error" $out/pytest_cli.log | tail -20 ;;
e2e)
  pause=${1:-6}; d=/dev/shm/qvz_e2e_$$; mkdir -p $d
  python3 -c "
import sys
from qvz_amd import host, kernels as K
host.egress_rows(K.qgen1(3, 100_000_000, 150), 100_000_000, '$d/in.txt')"
  export QVZ_WELL_SEED=debug QVZ_SHARDS=8192 QVZ_TIMING=1 QVZ_CODER_DEBUG=1
  for flags in "-c 1 -d L -f 0.5" "-c 1 -d L -f 0.5" "-c 3 -d M -r 1.0" "-u $d/out.u -c 1 -d L -f 0.5"; do
    sleep $pause; echo "== qvz -s $flags (after ${pause} s)"
    qvz_amd/bin/qvz -s $flags $d/in.txt $d/out.qset 2>&1 | grep -E "qvz_timing|host side" | cut -c1-260
  done | tee $out/runs.txt
  sleep $pause; ( time qvz_amd/bin/qvz -x $d/out.qset $d/out.dec ) 2>&1 | grep -E "real|qvz_timing" | tee -a $out/runs.txt
  cmp $d/out.dec $d/out.u && echo "decoded == -u" | tee -a $out/runs.txt; rm -rf $d ;;
pmf)
  timeout -k 10 600 python3 -m pytest tests/test_gpu_kernels.py -x -q -k pmf > $out/pytest.log 2>&1; echo "pytest rc=$?"; tail -3 $out/pytest.log
  for cfg in 2 3; do
    timeout -k 10 300 python3 bench.py --config $cfg --steps 4 --warmup 1 --no-cpu-baseline --no-extras 2> $out/err$cfg.txt | python3 -c "
import json,sys; d=json.loads(sys.stdin.readline()); print('config $cfg', d['roofline_pmf_hist'])"
  done ;;
codebook)
  timeout -k 10 900 python3 -m pytest tests/test_gpu_codebook.py tests/test_gpu_encode.py tests/test_gpu_dropin.py -x -q > $out/pytest.log 2>&1; echo "pytest rc=$?"; tail -4 $out/pytest.log
  for cfg in 2 1 3; do
    for how in default host-libm; do
      echo "== config $cfg, $how"; unset QVZ_CB_LOG2
      [ $how = host-libm ] && export QVZ_CB_LOG2=host
      QVZ_CB_DEBUG=1 timeout -k 10 300 python3 bench.py --config $cfg --steps 4 --warmup 2 --no-cpu-baseline --no-extras 2> $out/err_${cfg}_$how.txt | line
      grep -m2 "qvz_hip_generate_codebook" $out/err_${cfg}_$how.txt | cut -c1-220
    done
  done 2>&1 | tee $out/cb.txt ;;
decoder)
  timeout -k 10 600 python3 -m pytest tests/test_gpu_decoder.py -x -q > $out/pytest.log 2>&1; echo "pytest rc=$?"; tail -3 $out/pytest.log
  for cfg in 2 3; do
    timeout -k 10 400 python3 bench.py --config $cfg --steps 1 --warmup 1 --no-cpu-baseline --no-extras --roundtrip 2> $out/err$cfg.txt | python3 -c "
import json,sys; print('config $cfg', json.loads(sys.stdin.readline())['roundtrip'])"
  done ;;
decoder-profile)
  cfg=${1:-2}
  cp -r . /tmp/prof_build && cd /tmp/prof_build && touch qvz_amd/csrc/qvz_decoder_gpu.hip &&
  make -s HIPFLAGS="--offload-arch=gfx950 -O3 -ffp-contract=off -fPIC -std=c++17 -Iinclude -Iqvz_amd/csrc -DQVZ_PROFILE" > /dev/null 2>&1 || exit 1
  timeout -k 10 400 python3 bench.py --config $cfg --steps 1 --warmup 1 --no-cpu-baseline --no-extras --roundtrip 2> $out/decoder_profile.txt | python3 -c "
import json,sys; print('config $cfg', json.loads(sys.stdin.readline())['roundtrip'])"
  grep "k_decode_shards" $out/decoder_profile.txt
  echo "-- every stream on one set of tables (the tables come from the cache; the output is wrong and refused)"
  QVZ_DEC_SHARED_ARENA=1 timeout -k 10 400 python3 bench.py --config $cfg --steps 1 --warmup 1 --no-cpu-baseline --no-extras --roundtrip > /dev/null 2> $out/decoder_profile_shared.txt
  grep "k_decode_shards" $out/decoder_profile_shared.txt ;;
n2)
  reads=${1:-20000000}
  QVZ_BENCH_ONE_DEVICE=1 QVZ_BENCH_BACKEND=gloo timeout -k 10 500 python3 -m torch.distributed.run --nnodes=1 --nproc-per-node 2 --master-addr 127.0.0.1 \
    --master-port 29617 bench.py --gpus 2 --reads $reads --steps 2 --warmup 1 > $out/n2.jsonl 2> $out/n2.err; echo "rc=$?"
  line < $out/n2.jsonl
  timeout -k 10 300 python3 bench.py --reads $reads --steps 2 --warmup 1 --no-cpu-baseline --no-extras 2> $out/n1.err | tee $out/n1.jsonl | line ;;
host)
  { nproc; free -g; df -h /dev/shm /tmp .; grep -m1 "model name" /proc/cpuinfo; } 2>&1 | tee $out/host.txt
  gcc -O2 -o /tmp/io_bench tools/io_bench.c -lpthread && /tmp/io_bench /dev/shm/iob 16 4 8 16 2>&1 | tee $out/io_bench.txt ;;
*) echo "unknown action $action"; exit 2 ;;
esac

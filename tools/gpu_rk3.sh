#!/bin/bash
# where the model kernel's time goes with three clusters: knobs on the shipped build, then the per-section cycle counters
# of a development build (-DQVZ_PROFILE, built into /tmp, not shipped)
export TMPDIR=/tmp
run() {  # label, config, env...
  local label=$1 cfg=$2; shift 2
  echo "== $label"
  env "$@" QVZ_CODER_DEBUG=1 timeout -k 10 300 python3 bench.py --config $cfg --steps 2 --warmup 1 --no-cpu-baseline --no-extras --no-u 2> /tmp/err.txt | python3 -c "import json,sys; d=json.loads(sys.stdin.read()); print(d['ms_per_step'], {k:round(v*1e3,1) for k,v in d['phase_seconds'].items() if 'coder' in k})" ; local rc=$?
  grep -E "k_model_ranks cycles|model waves|segment" /tmp/err.txt | tail -9
  [ $rc -ne 0 ] && tail -3 /tmp/err.txt
  return $rc
}
run "cfg3 default" 3 X=1 || exit 1
run "cfg3 W=2" 3 QVZ_MODEL_WAVES=2 || exit 1
run "cfg3 no overlap" 3 QVZ_CODER_OVERLAP=0 || exit 1
run "cfg2 no overlap" 2 QVZ_CODER_OVERLAP=0 || exit 1
set -e
cp -r . /tmp/prof_build && cd /tmp/prof_build
touch qvz_amd/csrc/qvz_coder_gpu.hip
make -s HIPFLAGS="--offload-arch=gfx950 -O3 -ffp-contract=off -fPIC -std=c++17 -Iinclude -Iqvz_amd/csrc -DQVZ_PROFILE" > /dev/null 2>&1
set +e
run "PROFILE cfg3" 3 X=1 || exit 1
run "PROFILE cfg2" 2 X=1 || exit 1
run "PROFILE cfg3 no overlap" 3 QVZ_CODER_OVERLAP=0 || exit 1
run "PROFILE cfg3 no overlap, no record store" 3 QVZ_CODER_OVERLAP=0 QVZ_RK_EXP=1 || true
run "PROFILE cfg3 no overlap, no cumulative loop" 3 QVZ_CODER_OVERLAP=0 QVZ_RK_EXP=4 || true
run "PROFILE cfg3 no overlap, no table update" 3 QVZ_CODER_OVERLAP=0 QVZ_RK_EXP=8 || true

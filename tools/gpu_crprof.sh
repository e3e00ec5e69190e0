#!/bin/bash
# per-role busy / waiting cycles of the coder kernel (development build with -DQVZ_PROFILE, built into /tmp)
export TMPDIR=/tmp
set -e
cp -r . /tmp/prof_build && cd /tmp/prof_build
touch qvz_amd/csrc/qvz_coder_gpu.hip
make -s HIPFLAGS="--offload-arch=gfx950 -O3 -ffp-contract=off -fPIC -std=c++17 -Iinclude -Iqvz_amd/csrc -DQVZ_PROFILE" > /dev/null 2>&1
set +e
run() {
  local label=$1 cfg=$2; shift 2
  echo "== $label"
  env "$@" QVZ_CODER_DEBUG=1 timeout -k 10 300 python3 bench.py --config $cfg --steps 2 --warmup 1 --no-cpu-baseline --no-extras --no-u 2> /tmp/err.txt | python3 -c "import json,sys; d=json.loads(sys.stdin.read()); print(d['ms_per_step'], {k:round(v*1e3,1) for k,v in d['phase_seconds'].items() if 'coder' in k})"; local rc=$?
  grep -E "k_code_records cycles|k_model_ranks cycles|segment|streams made" /tmp/err.txt | tail -12
  [ $rc -ne 0 ] && tail -3 /tmp/err.txt
  return $rc
}
run "cfg2 overlapped" 2 X=1 || exit 1
run "cfg2 alone" 2 QVZ_CODER_OVERLAP=0 || exit 1

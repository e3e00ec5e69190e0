#!/bin/bash
# model kernel time against workgroups per CU (QVZ_MODEL_LDS_PAD)
export TMPDIR=/tmp
run() {
  local label=$1 cfg=$2; shift 2
  echo "== $label"
  env "$@" timeout -k 10 300 python3 bench.py --config $cfg --steps 2 --warmup 1 --no-cpu-baseline --no-extras --no-u 2> /tmp/err.txt | python3 -c "import json,sys; d=json.loads(sys.stdin.read()); print(d['ms_per_step'], {k:round(v*1e3,1) for k,v in d['phase_seconds'].items() if 'coder' in k})"; local rc=$?
  [ $rc -ne 0 ] && tail -3 /tmp/err.txt
  return $rc
}
run "cfg2 alone, 3 WGs/CU (42.9 KB)" 2 QVZ_CODER_OVERLAP=0 || exit 1
run "cfg2 alone, 2 WGs/CU (+12 KB)" 2 QVZ_CODER_OVERLAP=0 QVZ_MODEL_LDS_PAD=12288 || exit 1
run "cfg2 alone, 1 WG/CU (+40 KB)" 2 QVZ_CODER_OVERLAP=0 QVZ_MODEL_LDS_PAD=40960 || exit 1
run "cfg2 overlapped, +12 KB" 2 QVZ_MODEL_LDS_PAD=12288 || exit 1

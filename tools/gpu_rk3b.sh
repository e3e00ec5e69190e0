#!/bin/bash
export TMPDIR=/tmp
timeout -k 10 400 python3 -m pytest tests/test_gpu_pipeline.py tests/test_gpu_encode.py -m gpu -q -x 2>&1 | tail -3
run() {
  local label=$1 cfg=$2; shift 2
  echo "== $label"
  env "$@" QVZ_CODER_DEBUG=1 timeout -k 10 300 python3 bench.py --config $cfg --steps 3 --warmup 1 --no-cpu-baseline --no-extras --no-u 2> /tmp/err.txt | python3 -c "import json,sys; d=json.loads(sys.stdin.read()); print(d['ms_per_step'], {k:round(v*1e3,1) for k,v in d['phase_seconds'].items() if 'coder' in k})"; local rc=$?
  grep -E "model waves" /tmp/err.txt | tail -1
  [ $rc -ne 0 ] && tail -3 /tmp/err.txt
  return $rc
}
run "cfg3 default" 3 X=1 || exit 1
run "cfg3 no overlap" 3 QVZ_CODER_OVERLAP=0 || exit 1
run "cfg2 default" 2 X=1 || exit 1

#!/bin/bash
export TMPDIR=/tmp
run() {
  local label=$1 cfg=$2; shift 2
  echo "== $label"
  env "$@" timeout -k 10 300 python3 bench.py --config $cfg --steps 6 --warmup 1 --no-cpu-baseline --no-extras --no-u 2> /tmp/err.txt | python3 -c "import json,sys; d=json.loads(sys.stdin.read()); print(d['ms_per_step'], d['step_ms'], {k:round(v*1e3,1) for k,v in d['phase_seconds'].items() if 'coder' in k})"; local rc=$?
  [ $rc -ne 0 ] && tail -3 /tmp/err.txt
  return $rc
}
run "cfg3 first process" 3 X=1 || exit 1
run "cfg3 second process" 3 X=1 || exit 1
run "cfg2 third process" 2 X=1 || exit 1

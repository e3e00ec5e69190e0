#!/bin/bash
export TMPDIR=/tmp
run() {
  local label=$1 cfg=$2; shift 2
  echo "== $label"
  env "$@" QVZ_CODER_DEBUG=1 timeout -k 10 300 python3 bench.py --config $cfg --steps 4 --warmup 1 --no-cpu-baseline --no-extras --no-u 2> /tmp/err.txt | python3 -c "import json,sys; d=json.loads(sys.stdin.read()); print(d['ms_per_step'], {k:round(v*1e3,1) for k,v in d['phase_seconds'].items() if 'coder' in k})"; local rc=$?
  grep -E "segment" /tmp/err.txt | awk '{print $3, $4, $8-$6, $12-$10}' | tr '\n' ';' ; echo
  [ $rc -ne 0 ] && tail -3 /tmp/err.txt
  return $rc
}
run "cfg3 prio" 3 X=1 || exit 1
run "cfg3 noprio" 3 QVZ_CODER_PRIORITY=0 || exit 1
run "cfg3 prio" 3 X=1 || exit 1
run "cfg3 noprio" 3 QVZ_CODER_PRIORITY=0 || exit 1
run "cfg2 prio" 2 X=1 || exit 1
run "cfg2 noprio" 2 QVZ_CODER_PRIORITY=0 || exit 1

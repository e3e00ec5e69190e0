#!/bin/bash
export TMPDIR=/tmp
run() {
  local label=$1 cfg=$2; shift 2
  echo "== $label"
  env "$@" QVZ_CODER_DEBUG=1 timeout -k 10 300 python3 bench.py --config $cfg --steps 4 --warmup 1 --no-cpu-baseline --no-extras 2> /tmp/err.txt | python3 -c "import json,sys; d=json.loads(sys.stdin.read()); print(d['ms_per_step'], d['step_ms'], {k:round(v*1e3,1) for k,v in d['phase_seconds'].items() if 'coder' in k})"; local rc=$?
  grep -E "segment" /tmp/err.txt | tail -10 | awk '{printf "%s %s m %s-%s c %s-%s; ", $2, $3, $6, $8, $11, $13} END {print ""}'
  [ $rc -ne 0 ] && tail -3 /tmp/err.txt
  return $rc
}
for cfg in 2 3; do
for t in 0 1 2 3; do run "cfg$cfg taper $t" $cfg QVZ_CODER_TAPER=$t || exit 1; done
done

#!/bin/bash
# coder waves at raised (default) against normal issue priority (QVZ_CODER_PRIO)
export TMPDIR=/tmp
run() {
  echo "== config $1: $2"
  env $2 timeout -k 10 300 python3 bench.py --config $1 --steps 8 --warmup 2 --no-cpu-baseline --no-extras 2> /tmp/err.txt | python3 -c "
import json,sys; lines=[l for l in sys.stdin.read().splitlines() if l.startswith('{')]; d=json.loads(lines[-1]); print(d['ms_per_step'], d['step_ms'])
print('  ', {k:round(v*1e3,1) for k,v in d['phase_seconds'].items() if 'coder' in k})" || { tail -20 /tmp/err.txt; exit 1; }
}
for cfg in 3 2 1; do for p in 1 0; do run $cfg QVZ_CODER_PRIO=$p || exit 1; done; done

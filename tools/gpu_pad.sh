#!/bin/bash
# LDS padding of the coder workgroups (keeps a third model workgroup off their CUs) now that the coder kernel gets a head start
export TMPDIR=/tmp
run() {
  echo "== config $1: $2"
  env $2 QVZ_CODER_DEBUG=1 timeout -k 10 300 python3 bench.py --config $1 --steps 8 --warmup 2 --no-cpu-baseline --no-extras 2> /tmp/err.txt | python3 -c "
import json,sys; lines=[l for l in sys.stdin.read().splitlines() if l.startswith('{')]; d=json.loads(lines[-1]); print(d['ms_per_step'], d['step_ms'])
print('  ', {k:round(v*1e3,1) for k,v in d['phase_seconds'].items() if 'coder' in k})" || { tail -20 /tmp/err.txt; exit 1; }
}
run 2 X=1 || exit 1
run 2 QVZ_CODER_LDS_PAD=0 || exit 1
run 2 QVZ_CODER_LDS_PAD=4096 || exit 1
run 2 QVZ_CODER_LDS_PAD=16384 || exit 1

#!/bin/bash
# head start of the coder kernel over the next model kernel (QVZ_CODER_HEADSTART_US): 0 / 40 / 100 us
export TMPDIR=/tmp
run() {
  echo "== config $1: $2"
  env $2 timeout -k 10 300 python3 bench.py --config $1 --steps 12 --warmup 2 --no-cpu-baseline --no-extras 2> /tmp/err.txt | python3 -c "
import json,sys; lines=[l for l in sys.stdin.read().splitlines() if l.startswith('{')]; d=json.loads(lines[-1]); print(d['ms_per_step'], d['step_ms'])
for k,v in d['step_phase_ms'].items():
    if k in ('coder','coder_interval_kernels'): print(' ', k, v)" || { tail -20 /tmp/err.txt; exit 1; }
}
for us in 0 40 100; do run 3 QVZ_CODER_HEADSTART_US=$us || exit 1; done
for us in 0 40; do run 2 QVZ_CODER_HEADSTART_US=$us || exit 1; done
run 1 QVZ_CODER_HEADSTART_US=0; run 1 QVZ_CODER_HEADSTART_US=40

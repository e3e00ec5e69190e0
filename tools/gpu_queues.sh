#!/bin/bash
# GPU_MAX_HW_QUEUES = 4 / 8 / 16 for the three configurations
export TMPDIR=/tmp
run() {
  echo "== config $1: $2"
  env $2 timeout -k 10 300 python3 bench.py --config $1 --steps 9 --warmup 2 --no-cpu-baseline --no-extras 2> /tmp/err.txt | python3 -c "
import json,sys; lines=[l for l in sys.stdin.read().splitlines() if l.startswith('{')]; d=json.loads(lines[-1]); print(d['ms_per_step'], d['step_ms'])
for k,v in d['step_phase_ms'].items():
    if k in ('codebook','coder','coder_interval_kernels','coder_model_kernels'): print(' ', k, v)" || { cat /tmp/err.txt | tail -40; exit 1; }
}
for cfg in 3 2 1; do
for q in 4 8 16; do run $cfg GPU_MAX_HW_QUEUES=$q || exit 1; done
done

#!/bin/bash
# per-step phase times over many steps: periodic slow steps mean two streams share a hardware queue
export TMPDIR=/tmp
for cfg in 2 3 1; do
echo "== config $cfg"
timeout -k 10 300 python3 bench.py --config $cfg --steps 13 --warmup 2 --no-cpu-baseline --no-extras 2> /tmp/err.txt | python3 -c "
import json,sys; d=json.loads(sys.stdin.read()); print(d['ms_per_step'], d['step_ms'])
for k,v in d['step_phase_ms'].items(): print(' ', k, v)" || { tail -3 /tmp/err.txt; exit 1; }
done

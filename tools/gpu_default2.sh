#!/bin/bash
export TMPDIR=/tmp
for q in 8 4; do
echo "== GPU_MAX_HW_QUEUES=$q"
GPU_MAX_HW_QUEUES=$q timeout -k 10 600 python3 bench.py --no-cpu-baseline 2> /tmp/err.txt | python3 -c "
import json,sys; lines=[l for l in sys.stdin.read().splitlines() if l.startswith('{')]; d=json.loads(lines[-1]); print(d['ms_per_step'], d['step_ms'])
for o in d.get('other_configs',[]): print(o['config'], o['ms_per_step'], o['step_ms'], {k:round(v*1e3,1) for k,v in o['phase_seconds'].items()})" || { tail -20 /tmp/err.txt; exit 1; }
done

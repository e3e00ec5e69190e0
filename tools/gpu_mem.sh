#!/bin/bash
# HBM head-room after every step (QVZ_BENCH_MEM=1)
export TMPDIR=/tmp
for cfg in 3 2; do
echo "== config $cfg"
QVZ_BENCH_MEM=1 timeout -k 10 300 python3 bench.py --config $cfg --steps 6 --warmup 2 --no-cpu-baseline --no-extras 2> /tmp/err.txt | python3 -c "
import json,sys; lines=[l for l in sys.stdin.read().splitlines() if l.startswith('{')]; d=json.loads(lines[-1]); print(d['ms_per_step'], d['step_ms'])"
grep "^step" /tmp/err.txt
done

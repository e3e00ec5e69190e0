#!/bin/bash
# the N > 1 code path (RCCL all-reduces, its streams) rehearsed with a one-rank group, and the plain one: per-step regularity
export TMPDIR=/tmp
run() {
  echo "== config $1: $2 $3"
  env $2 $3 QVZ_CODER_DEBUG=1 timeout -k 10 300 python3 bench.py --config $1 --steps 9 --warmup 2 --no-cpu-baseline --no-extras 2> /tmp/err.txt | python3 -c "
import json,sys; lines=[l for l in sys.stdin.read().splitlines() if l.startswith('{')]; d=json.loads(lines[-1]); print(d['ms_per_step'], d['step_ms'])
for k,v in d['step_phase_ms'].items():
    if k in ('codebook','coder'): print(' ', k, v)" || { cat /tmp/err.txt | tail -40; exit 1; }
  grep "streams made" /tmp/err.txt | tail -1
}
for cfg in 2 3; do
run $cfg X=1 Y=1 || exit 1
run $cfg QVZ_BENCH_FORCE_DIST=1 Y=1 || exit 1
run $cfg QVZ_BENCH_FORCE_DIST=1 GPU_MAX_HW_QUEUES=8 || exit 1
done

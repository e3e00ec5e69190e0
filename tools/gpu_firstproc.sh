#!/bin/bash
# is the first GPU process of a fresh box slower (lost model / coder overlap)? per-step times + the coder time line of every call
export TMPDIR=/tmp
QVZ_CODER_DEBUG=1 timeout -k 10 300 python3 bench.py --config ${1:-2} --steps 6 --warmup 1 --no-cpu-baseline --no-extras 2> /tmp/err.txt | python3 -c "
import json,sys; lines=[l for l in sys.stdin.read().splitlines() if l.startswith('{')]; d=json.loads(lines[-1]); print(d['ms_per_step'], d['step_ms'])
for k,v in d['step_phase_ms'].items():
    if k in ('codebook','coder','coder_model_kernels','coder_interval_kernels'): print(' ', k, v)"
grep -E "streams made|segment 6|segment 0" /tmp/err.txt | head -20

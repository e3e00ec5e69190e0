#!/bin/bash
export TMPDIR=/tmp
timeout -k 10 600 python3 -m pytest tests/test_gpu_pipeline.py tests/test_gpu_encode.py tests/test_gpu_decoder.py -m gpu -q -x 2>&1 | tail -3 || exit 1
for cfg in 2 3; do
echo "== config $cfg"
timeout -k 10 300 python3 bench.py --config $cfg --steps 7 --warmup 2 --no-cpu-baseline --no-extras 2> /tmp/err.txt | python3 -c "
import json,sys; lines=[l for l in sys.stdin.read().splitlines() if l.startswith('{')]; d=json.loads(lines[-1]); print(d['ms_per_step'], d['step_ms'])
for k,v in d['step_phase_ms'].items():
    if 'coder' in k: print(' ', k, v)" || { tail -20 /tmp/err.txt; exit 1; }
done

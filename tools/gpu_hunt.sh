#!/bin/bash
# many short encodes (configs[1]) to catch a call that takes much longer than its kernels; such a call prints its time line
export TMPDIR=/tmp
for r in 1 2 3 4; do
timeout -k 10 300 python3 bench.py --config 1 --steps 400 --warmup 2 --no-cpu-baseline --no-extras 2> /tmp/err_$r.txt | python3 -c "
import json,sys; lines=[l for l in sys.stdin.read().splitlines() if l.startswith('{')]; d=json.loads(lines[-1]); s=d['step_ms']; import statistics as st
print(d['ms_per_step'], 'median', st.median(s), 'max', max(s), [(i,x) for i,x in enumerate(s) if x > 2*st.median(s)])
for i,x in enumerate(s):
    if x > 2*st.median(s): print('   step', i, {k:v[i] for k,v in d['step_phase_ms'].items()})"
grep -A12 "this call took\|host side" /tmp/err_$r.txt | head -40
done

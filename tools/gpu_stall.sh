#!/bin/bash
# hunting the occasional long coder call: many steps, blocking wait against polling wait
export TMPDIR=/tmp
for mode in poll block; do
echo "== wait: $mode"
QVZ_CODER_WAIT=$mode QVZ_CODER_DEBUG=1 timeout -k 10 300 python3 bench.py --config ${1:-2} --steps 60 --warmup 1 --no-cpu-baseline --no-extras 2> /tmp/err.txt | python3 -c "
import json,sys; lines=[l for l in sys.stdin.read().splitlines() if l.startswith('{')]; d=json.loads(lines[-1]); s=d['step_ms']; import statistics as st
print(d['ms_per_step'], 'median', st.median(s), 'max', max(s), 'steps over median+5:', [(i,x) for i,x in enumerate(s) if x > st.median(s)+5])
for i,x in enumerate(s):
    if x > st.median(s)+5: print('   step', i, {k:v[i] for k,v in d['step_phase_ms'].items()})"
grep "host side" /tmp/err.txt | awk '{print $12, $15}' | sort -n | tail -3 | tr '\n' ';'; echo
done

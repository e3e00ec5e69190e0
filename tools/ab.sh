#!/bin/bash
# tools/ab.sh <tag> <config> <rounds>: the tree's library against ab_base/ (a copy of HEAD built by hand), alternating; development aid
tag=${1:?tag}; cfg=${2:-2}; rounds=${3:-2}
root=$(pwd); out=$root/gpurun_out/$tag; mkdir -p $out
one() { # dir label env...
  dir=$1; label=$2; shift 2
  ( cd $dir && env "$@" timeout -k 10 300 python3 bench.py --config $cfg --steps 5 --warmup 2 --no-cpu-baseline --no-extras 2>> $out/err.txt ) | python3 -c "
import json,sys
d=json.loads([l for l in sys.stdin.read().splitlines() if l.startswith('{')][-1]); p=d['step_phase_ms']
print('$label', d['ms_per_step'], 'codebook', p['codebook'][:3], 'coder', p['coder'][:3], 'model', p['coder_model_kernels'][:2], 'interval', p['coder_interval_kernels'][:2])"
}
for r in $(seq $rounds); do
  one $root/ab_base base A=1
  one $root tree A=1
done | tee $out/ab.txt

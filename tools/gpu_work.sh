#!/bin/bash
# record workspace size (QVZ_WORK_GIB) for configs[2] and configs[3]
export TMPDIR=/tmp
for cfg in 2 3; do
for w in 96 124; do
  echo "== cfg$cfg QVZ_WORK_GIB=$w"
  QVZ_WORK_GIB=$w timeout -k 10 300 python3 bench.py --config $cfg --steps 4 --warmup 1 --no-cpu-baseline --no-extras | python3 -c "import json,sys; d=json.loads(sys.stdin.read()); print(d['ms_per_step'], d['step_ms'], {k:round(v*1e3,1) for k,v in d['phase_seconds'].items() if 'coder' in k})" || exit 1
done
done

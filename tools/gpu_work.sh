#!/bin/bash
export TMPDIR=/tmp
for w in 48 96 150; do
  echo "== QVZ_WORK_GIB=$w"
  QVZ_WORK_GIB=$w timeout -k 10 300 python3 bench.py --config 3 --steps 2 --warmup 1 --no-cpu-baseline --no-extras --no-u | python3 -c "import json,sys; d=json.loads(sys.stdin.read()); print(d['ms_per_step'], {k:round(v*1e3,1) for k,v in d['phase_seconds'].items() if 'coder' in k})" || exit 1
done

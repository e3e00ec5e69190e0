#!/bin/bash
# WELL generator pass size (QVZ_WELL_PASS_LOG2)
export TMPDIR=/tmp
for p in 22 24 26; do
  echo "== pass 2^$p"
  QVZ_WELL_PASS_LOG2=$p timeout -k 10 200 python3 bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-extras | python3 -c "import json,sys; d=json.loads(sys.stdin.read()); print(d['ms_per_step'], d['phase_seconds'], d['gpu_kernel_ms_per_step']['well_draws'])" || exit 1
done

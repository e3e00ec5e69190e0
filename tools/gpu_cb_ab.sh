#!/bin/bash
# codebook hand-over variants (QVZ_CB_SYNC / QVZ_CB_STAGE) side by side
out=gpurun_out/$1
mkdir -p $out
export TMPDIR=/tmp
for v in "" "QVZ_CB_FLAGKERNEL=1" "" "QVZ_CB_FLAGKERNEL=1"; do
  for cfg in 3; do
    echo "== cfg $cfg [$v]"
    env $v QVZ_CB_DEBUG=1 timeout -k 10 200 python3 bench.py --config $cfg --steps 3 --warmup 1 --no-cpu-baseline --no-extras 2> $out/e.txt | python3 -c "import json,sys; d=json.loads(sys.stdin.read()); print(d['ms_per_step'], d['phase_seconds']['codebook'])" || exit 1
    grep generate_codebook $out/e.txt | tail -3 | cut -c1-150
  done
done

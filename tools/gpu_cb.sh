#!/bin/bash
# codebook phase split (QVZ_CB_DEBUG) of the three BASELINE configurations
out=gpurun_out/$1
mkdir -p $out
export TMPDIR=/tmp
for cfg in 2 3 1; do
  QVZ_CB_DEBUG=1 timeout -k 10 200 python3 bench.py --config $cfg --steps 3 --warmup 1 --no-cpu-baseline --no-extras > $out/cb_cfg$cfg.jsonl 2> $out/cb_cfg$cfg.err || exit 1
  grep generate_codebook $out/cb_cfg$cfg.err | tail -4
done

#!/bin/bash
# quantizer variants with three clusters (QVZ_QUANTIZE_T, QVZ_QUANTIZE_RW)
export TMPDIR=/tmp
timeout -k 10 300 python3 -m pytest tests/test_gpu_kernels.py tests/test_gpu_pipeline.py -m gpu -q -k "quant or shard or stream" 2>&1 | tail -2
for v in "QVZ_QUANTIZE_T=512" "" "QVZ_QUANTIZE_RW=2"; do
  echo "== [$v]"
  env $v timeout -k 10 300 python3 bench.py --config 3 --steps 3 --warmup 1 --no-cpu-baseline --no-extras | python3 -c "import json,sys; d=json.loads(sys.stdin.read()); print(d['ms_per_step'], d['phase_seconds']['quantize'], d['gpu_kernel_ms_per_step']['quantize'])" || exit 1
done

#!/bin/bash
export TMPDIR=/tmp
timeout -k 10 400 python3 -m pytest tests/test_gpu_kernels.py -m gpu -q -x -k "kmeans" 2>&1 | tail -15 || exit 1
run() {
  local label=$1 cfg=$2; shift 2
  echo "== $label"
  env "$@" timeout -k 10 300 python3 bench.py --config $cfg --steps 3 --warmup 1 --no-cpu-baseline --no-extras --no-u 2> /tmp/err.txt | python3 -c "import json,sys; d=json.loads(sys.stdin.read()); print(d['ms_per_step'], {k:round(v*1e3,1) for k,v in d['phase_seconds'].items()}, d['kmeans_iterations'], {k:round(v,2) for k,v in d['gpu_kernel_ms_per_step'].items() if 'kmeans' in k})"; local rc=$?
  [ $rc -ne 0 ] && tail -3 /tmp/err.txt
  return $rc
}
run "cfg3 fused" 3 X=1 || exit 1
run "cfg3 two-pass" 3 QVZ_KMEANS_FUSED=0 || exit 1

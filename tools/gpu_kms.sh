#!/bin/bash
# k-means parity tests, then the one-pass kernel against the two-pass kernels at 100M x 150 for k = 3, 5, 8
export TMPDIR=/tmp
timeout -k 10 400 python3 -m pytest tests/test_gpu_kernels.py -m gpu -q -x -k "kmeans" 2>&1 | tail -6 || exit 1
timeout -k 10 200 python3 - <<'PY'
import time, torch, numpy as np
from qvz_amd import kernels as K
n, c = 100_000_000, 150
sym = K.qgen1(3, n, c)
for k in (3, 5, 8):
    idx = torch.randint(0, n, (k,), device="cuda")
    means = sym[:, idx].t().contiguous()
    labels = torch.zeros(sym.shape[1], dtype=torch.uint8, device="cuda")
    counts = torch.zeros(k, dtype=torch.int32, device="cuda"); acc = torch.zeros((k, c), dtype=torch.int64, device="cuda")
    for fused in ("1", "0"):
        import os; os.environ["QVZ_KMEANS_FUSED"] = fused
        K.kmeans_step(sym, n, means, labels, counts, acc); torch.cuda.synchronize()
        t = time.perf_counter()
        for _ in range(3): K.kmeans_step(sym, n, means, labels, counts, acc)
        torch.cuda.synchronize()
        print(f"k={k} fused={fused}: {(time.perf_counter()-t)/3*1e3:.2f} ms per iteration")
PY

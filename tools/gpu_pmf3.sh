#!/bin/bash
# PMF pass with three clusters: tile / replica / group sweep (tools/microbench-like, through bench.py --config 3)
export TMPDIR=/tmp
python3 - <<'PY'
import os, sys, time
sys.path.insert(0, ".")
import torch
from qvz_amd import kernels as K
n, c, k = 100_000_000, 150, 3
sym = K.qgen1(3, n, c)
labels = (torch.arange(K.ld_for(n), device="cuda", dtype=torch.int64) * 2654435761 % 7 % 3).to(torch.uint8)
ref = None
for env in [{}, {"QVZ_PMF_Q": "4"}, {"QVZ_PMF_Q": "6"}, {"QVZ_PMF_G": "2", "QVZ_PMF_Q": "4"}, {"QVZ_PMF_G": "2", "QVZ_PMF_Q": "6"}, {"QVZ_PMF_G": "1", "QVZ_PMF_Q": "6"},
            {"QVZ_PMF_G": "2", "QVZ_PMF_R": "2"}]:
    for kk in ("QVZ_PMF_Q", "QVZ_PMF_G", "QVZ_PMF_R"):
        os.environ.pop(kk, None)
    os.environ.update(env)
    cnt = torch.zeros((k, c, 72, 72), dtype=torch.int32, device="cuda")
    K.pmf_hist(sym, n, labels=labels, k=k, cnt=cnt)
    ms = []
    for _ in range(3):
        cnt.zero_()
        a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        a.record(); K.pmf_hist(sym, n, labels=labels, k=k, cnt=cnt); b.record(); torch.cuda.synchronize()
        ms.append(a.elapsed_time(b))
    if ref is None: ref = cnt.clone()
    print(env, "ms", round(min(ms), 3), "frac", round((n * c + n) / (min(ms) * 1e-3) / 8e12, 4), "same", bool(torch.equal(cnt, ref)))
PY

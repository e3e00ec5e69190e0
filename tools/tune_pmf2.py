#!/usr/bin/env python3
"""Development aid: sweep the column-walking PMF kernel (R replicas, G column tiles, Q groups per thread)."""
import itertools, json, os, sys
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from qvz_amd import kernels as K
n, c = int(sys.argv[1]) if len(sys.argv) > 1 else 100_000_000, 150
k = int(sys.argv[2]) if len(sys.argv) > 2 else 1
sym = K.qgen1(3, n, c)
labels = (torch.arange(sym.shape[1], device="cuda") % k).to(torch.uint8) if k > 1 else None
os.environ["QVZ_PMF_ALGO"] = "0"
ref = torch.zeros((k, c, 72, 72), dtype=torch.int32, device="cuda"); K.pmf_hist(sym, n, labels=labels, k=k, cnt=ref)
os.environ["QVZ_PMF_ALGO"] = "1"
combos = [(0, 0, 0)] + list(itertools.product([int(x) for x in os.environ.get('SWEEP_R','1,2,3').split(',')], [int(x) for x in os.environ.get('SWEEP_G','1,2,4,8').split(',')], [int(x) for x in os.environ.get('SWEEP_Q','1,2,4').split(',')]))
for R, G, Q in combos:
    for key, v in (("QVZ_PMF_R", R), ("QVZ_PMF_G", G), ("QVZ_PMF_Q", Q)):
        if v: os.environ[key] = str(v)
        else: os.environ.pop(key, None)
    cnt = torch.zeros_like(ref)
    try:
        K.pmf_hist(sym, n, labels=labels, k=k, cnt=cnt); torch.cuda.synchronize()
    except Exception as e:
        print(R, G, Q, "failed", str(e)[:80]); continue
    if not os.environ.get('QVZ_PMF_ABLATE'): assert torch.equal(ref, cnt), (R, G, Q)
    ts = []
    for _ in range(4):
        a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        cnt.zero_(); a.record(); K.pmf_hist(sym, n, labels=labels, k=k, cnt=cnt); b.record(); b.synchronize()
        ts.append(a.elapsed_time(b))
    ms = float(np.median(ts))
    print(json.dumps(dict(R=R, G=G, Q=Q, ms=round(ms, 3), GBps=round(n * c / ms / 1e6, 1), frac=round(n * c / ms / 1e6 / 8000, 3))), flush=True)

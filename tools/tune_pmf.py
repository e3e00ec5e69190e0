#!/usr/bin/env python3
"""Development aid: sweep the PMF kernel's tuning knobs (replicas, block size, slabs) on QGEN-1 data."""
import itertools, json, os, sys
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from qvz_amd import kernels as K

n, c = int(sys.argv[1]) if len(sys.argv) > 1 else 20_000_000, 150
k = int(sys.argv[2]) if len(sys.argv) > 2 else 1
sym = K.qgen1(3, n, c)
labels = (torch.arange(sym.shape[1], device="cuda") % k).to(torch.uint8) if k > 1 else None
ref = None
for R, T, SL in itertools.product([int(x) for x in os.environ.get('SWEEP_R', '1,2,3,4,6,8,11,16').split(',')], [512, 1024], [0]):
    os.environ["QVZ_PMF_R"], os.environ["QVZ_PMF_T"] = str(R), str(T)
    os.environ.pop("QVZ_PMF_SLABS", None)
    cnt = torch.zeros((k, c, 72, 72), dtype=torch.int32, device="cuda")
    try:
        K.pmf_hist(sym, n, labels=labels, k=k, cnt=cnt)
    except Exception as e:
        print(R, T, "failed", e); continue
    torch.cuda.synchronize()
    if ref is None: ref = cnt.clone()
    if not os.environ.get('QVZ_PMF_ABLATE'): assert torch.equal(ref, cnt)
    ts = []
    for _ in range(5):
        a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        cnt.zero_(); a.record(); K.pmf_hist(sym, n, labels=labels, k=k, cnt=cnt); b.record(); b.synchronize()
        ts.append(a.elapsed_time(b))
    ms = float(np.median(ts))
    print(json.dumps(dict(R=R, T=T, ms=round(ms, 3), GBps=round(n * c / ms / 1e6, 1), frac=round(n * c / ms / 1e6 / 8000, 3))), flush=True)

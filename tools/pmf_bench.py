#!/usr/bin/env python3
"""The PMF pass alone at full size: one line per QVZ_PMF setting (test hooks of qvz_hip_pmf_hist: algo, q = 16-byte groups per
thread, aa = tile side), with one and with three clusters (uniform labels, or one k-means assignment's uneven ones), the
counters of every setting compared with the first one's. Ten seconds on the GPU box. Development aid.

    python tools/pmf_bench.py --reads 100000000 --cols 150 --labels kmeans --settings "q=6;q=4"
"""
import argparse
import json
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from qvz_amd import kernels as K  # noqa: E402


def timeit(fn, reps):
    fn()
    torch.cuda.synchronize()
    ts = []
    for _ in range(reps):
        a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        a.record()
        fn()
        b.record()
        b.synchronize()
        ts.append(a.elapsed_time(b))
    return float(np.median(ts)), float(np.min(ts))


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--reads", type=int, default=100_000_000)
    ap.add_argument("--cols", type=int, default=150)
    ap.add_argument("--reps", type=int, default=7)
    ap.add_argument("--clusters", default="1,3")
    ap.add_argument("--labels", default="uniform", choices=["uniform", "kmeans"])
    ap.add_argument("--settings", default="q=6;q=4;q=2")
    args = ap.parse_args()
    n, c = args.reads, args.cols
    sym = K.qgen1(2, n, c)
    ld = sym.shape[1]
    for k in [int(x) for x in args.clusters.split(",")]:
        labels = None
        if k > 1:
            if args.labels == "kmeans":     # one assignment pass to k evenly spaced reads: uneven clusters that follow the data
                lines = [int(x) - int(x) % 4 for x in np.linspace(0, n - 4, k)]
                means = torch.stack([K.qgen1(2, 4, c, first=ln)[:, 0] for ln in lines]).contiguous()
                labels = torch.zeros(ld, dtype=torch.uint8, device="cuda")
                counts = torch.zeros(k, dtype=torch.int32, device="cuda")
                K.kmeans_assign(sym, n, means, labels=labels, counts=counts)
                print("cluster sizes", counts.cpu().tolist(), flush=True)
            else:
                g = torch.Generator(device="cuda")
                g.manual_seed(5)
                labels = torch.randint(0, k, (ld,), dtype=torch.uint8, device="cuda", generator=g)
        first = None
        for setting in args.settings.split(";"):
            os.environ["QVZ_PMF"] = setting
            cnt = torch.zeros((k, c, 72, 72), dtype=torch.int32, device="cuda")
            K.pmf_hist(sym, n, labels=labels, k=k, cnt=cnt)
            same = None
            if first is None:
                first = cnt.clone()
                assert int(first.to(torch.int64).sum().item()) == n * c
            else:
                same = bool(torch.equal(first, cnt))
            ms, mn = timeit(lambda: K.pmf_hist(sym, n, labels=labels, k=k, cnt=cnt), args.reps)
            alg = n * c + (n if k > 1 else 0)
            print(json.dumps(dict(k=k, setting=setting, ms=round(ms, 3), ms_min=round(mn, 3), frac_of_8TBps=round(alg / mn / 1e6 / 8000, 4),
                                  same_as_first=same)), flush=True)


if __name__ == "__main__":
    main()

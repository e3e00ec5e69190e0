#!/usr/bin/env python3
"""Per-kernel timing of the N-dependent phases on synthetic QGEN-1 data (HIP events on the launch stream).
Development aid: prints one JSON line per kernel with algorithmic GB/s against the 8 TB/s HBM peak.

    python tools/microbench.py --reads 10000000 --cols 150 [--clusters 3] [--reps 5]
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

PEAK = 8000.0  # GB/s, /opt/skills/guides/MI355X_MICROARCH.md


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
    ap.add_argument("--reads", type=int, default=10_000_000)
    ap.add_argument("--cols", type=int, default=150)
    ap.add_argument("--clusters", type=int, default=1)
    ap.add_argument("--reps", type=int, default=5)
    args = ap.parse_args()
    n, c, k = args.reads, args.cols, args.clusters
    sym = K.empty_matrix(n, c)
    ld = sym.shape[1]
    res = []

    def report(name, ms, mn, bytes_alg):
        res.append(dict(kernel=name, ms=round(ms, 4), ms_min=round(mn, 4), alg_GBps=round(bytes_alg / ms / 1e6, 1),
                        frac_of_8TBps=round(bytes_alg / ms / 1e6 / PEAK, 4), Gsym_per_s=round(n * c / ms / 1e6, 2)))
        print(json.dumps(res[-1]), flush=True)

    ms, mn = timeit(lambda: K.qgen1(2, n, c, out=sym), args.reps)
    report("qgen1", ms, mn, n * c)

    labels = torch.zeros(ld, dtype=torch.uint8, device="cuda")
    # centroids = k evenly spaced reads, regenerated with QGEN-1 (no torch indexing of the >2^31-element matrix)
    lines = [int(x) - int(x) % 4 for x in np.linspace(0, n - 4, k)]
    means = torch.stack([K.qgen1(2, 4, c, first=ln)[:, 0] for ln in lines]).contiguous()
    counts = torch.zeros(k, dtype=torch.int32, device="cuda")
    ms, mn = timeit(lambda: K.kmeans_assign(sym, n, means, labels=labels, counts=counts), args.reps)
    report(f"kmeans_assign(k={k})", ms, mn, n * c + n)
    acc = torch.zeros((k, c), dtype=torch.int64, device="cuda")
    ms, mn = timeit(lambda: K.kmeans_accumulate(sym, n, labels, k, acc=acc), args.reps)
    report(f"kmeans_accumulate(k={k})", ms, mn, n * c + (n * c if k > 1 else 0))

    cnt = torch.zeros((k, c, 72, 72), dtype=torch.int32, device="cuda")
    ms, mn = timeit(lambda: K.pmf_hist(sym, n, labels=labels if k > 1 else None, k=k, cnt=cnt), args.reps)
    report(f"pmf_hist(k={k})", ms, mn, n * c + (n if k > 1 else 0))

    nwords = (n * c + 3) // 4
    words = torch.empty(nwords, dtype=torch.int32, device="cuda")
    state = np.full(32, 0x55555555, dtype=np.uint32)
    ms, mn = timeit(lambda: K.well_words(state, 0, nwords, out=words), args.reps)
    report("well_words", ms, mn, 4 * nwords)
    del words
    draws = K.empty_matrix(n, c)
    ms, mn = timeit(lambda: K.well_draws(state, 0, n, c, out=draws), args.reps)
    report("well_draws(words+transpose)", ms, mn, 3 * n * c)

    # identity-ish tables: one context per previous value, 2-state quantizers (shape of the work, not of a real codebook)
    ct = torch.zeros((k, c, 72), dtype=torch.int16, device="cuda")
    ctx = (torch.arange(72, device="cuda") // 8).to(torch.int16)
    ct[:] = ctx | (64 << 8)
    qs = torch.zeros((k, c, 144, 72), dtype=torch.int16, device="cuda")
    x = torch.arange(72, device="cuda")
    qv = ((x // 8) * 8).to(torch.int16)
    qs[:] = (qv | ((x // 8).to(torch.int16) << 8))[None, None, None, :]
    dist = torch.from_numpy(np.array([[(a - b) ** 2 for a in range(72)] for b in range(72)], dtype=np.float64).reshape(-1)).cuda()
    ms, mn = timeit(lambda: K.quantize(sym, n, ct, qs, draws, dist, labels=labels if k > 1 else None, k=k), args.reps)
    report("quantize(qv+qi+st+err)", ms, mn, n * c * 5)

    text = torch.empty(n * (c + 1), dtype=torch.uint8, device="cuda")
    ms, mn = timeit(lambda: K.untranspose_text(sym, n), args.reps)
    report("untranspose_text", ms, mn, 2 * n * c)
    t = K.untranspose_text(sym, n)
    ms, mn = timeit(lambda: K.transpose_text(t, n, c), args.reps)
    report("transpose_text", ms, mn, 2 * n * c)


if __name__ == "__main__":
    main()

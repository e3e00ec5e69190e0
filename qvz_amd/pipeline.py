"""One pass of the encode hot path over a resident symbol matrix, optionally sharded over ranks.

Plumbing only: every phase is a call into the C-ABI (HIP kernels / host C). torch supplies device tensors, the
stream and - when a process group is given - the RCCL all-reduces that make read-sharded ranks see the statistics
of the whole file:

    k-means iteration : all_reduce(SUM) of acc[k][C] (i64) and count[k] (i32)          ~3.6 KB
    PMF counters      : all_reduce(SUM) of cnt[k][C][72][72] (i32), once                 3.1 MB per cluster
    distortion        : all_reduce(SUM) of one double

Codebook generation is N-independent and deterministic, so every rank generates the same codebooks from the
reduced counters; quantisation and coding are embarrassingly parallel per shard.
"""
import ctypes
import time

import numpy as np
import torch

from . import host
from . import kernels as K

LINES_PER_BLOCK = 1_000_000   # reference include/lines.h:12


def reference_centroid_lines(n_total, k, reseed=True):
    """Read indices the reference picks as initial centroids (cluster.c:198-201): libc rand(), fresh-process state."""
    libc = ctypes.CDLL(None)
    if reseed:
        libc.srand(1)
    blocks = (n_total + LINES_PER_BLOCK - 1) // LINES_PER_BLOCK
    lines = []
    for _ in range(k):
        b = libc.rand() % blocks
        count = min(LINES_PER_BLOCK, n_total - b * LINES_PER_BLOCK)
        lines.append(b * LINES_PER_BLOCK + libc.rand() % count)
    return lines


def rank_read_range(n_total, world, rank):
    """Contiguous reads [first, last) of `rank` when a file of n_total reads is sharded over `world` ranks: equal
    shares rounded up to a multiple of 4 reads, so every shard's first draw (first*C) sits on a WELL word boundary
    (4 reads x C draws = C whole 4-draw words) for any C."""
    per = -(-n_total // world)
    per4 = (per + 3) & ~3
    first = min(rank * per4, n_total)
    last = n_total if rank == world - 1 else min((rank + 1) * per4, n_total)
    return first, last


class Timer:
    """Per-phase device timing with events recorded on the stream the kernels are launched on."""

    def __init__(self, enabled=True):
        self.enabled = enabled
        self.spans = []

    def span(self, name):
        return _Span(self, name)

    def totals_ms(self):
        torch.cuda.synchronize()
        out = {}
        for name, a, b in self.spans:
            out.setdefault(name, []).append(a.elapsed_time(b))
        return out


class _Span:
    def __init__(self, timer, name):
        self.t, self.name = timer, name

    def __enter__(self):
        if self.t.enabled:
            self.a = torch.cuda.Event(enable_timing=True)
            self.a.record()
        return self

    def __exit__(self, *exc):
        if self.t.enabled:
            b = torch.cuda.Event(enable_timing=True)
            b.record()
            self.t.spans.append((self.name, self.a, b))


def encode_pass(sym, n, first_read, n_total, means0, *, clusters=1, mode=host.MODE_RATIO, ratio=0.5, dist="M",
                threshold=4.0, well_state=None, shards=1, threads=1, group=None, timer=None, keep=False, device_index=0,
                coder="host", work_bytes=None, keep_device=False, reuse=None, want_reuse=False):
    """Runs k-means -> PMF -> codebooks -> WELL -> quantize -> feed -> sharded arithmetic coding for the reads
    [first_read, first_read+n) held in `sym` (this rank's shard of a file with n_total reads).
    coder: "host" = `threads` host threads over the packed feed (qvz_encode_shards_device); "gpu" = the column-parallel
    model + lane-per-stream coder fed from the quantizer's planes (qvz_hip_encode_planes); "gpu-feed" = the older
    lane-per-stream kernel over the packed feed (qvz_hip_encode_shards). All produce byte-identical streams.

    means0: (clusters, c) uint8 device tensor of the initial centroids (symbols). first_read must be a multiple of 4.
    Returns a dict with stream sizes, distortion, codebooks, phase wall times; with keep=True also the device outputs
    and the streams as host bytes; with keep_device=True (GPU coders) the coded streams stay on the device
    (`coded`, stream g at g * `cap_per_shard`) next to `qv` and `labels`.
    reuse / want_reuse: the cluster labels, the PMF counters and the WELL draws do not depend on -f / -r / -d, so a
    rate-distortion sweep (the reference's generate_rd.sh: 20 encodes of one file) computes them once: pass
    want_reuse=True on the first operating point and hand its result["reuse"] to the others (rd_sweep below).
    """
    import torch.distributed as dist_mod
    c, ld = sym.shape
    k = clusters
    assert first_read % 4 == 0, "shards must start on a WELL word boundary"
    timer = timer or Timer(False)
    dev = sym.device
    wall = {}
    t0 = time.perf_counter()

    # ---- k-means (cluster.c:212-244) ----
    means = means0.clone()
    labels = torch.zeros(ld, dtype=torch.uint8, device=dev)
    counts = torch.zeros(k, dtype=torch.int32, device=dev)
    acc = torch.zeros((k, c), dtype=torch.int64, device=dev)
    iters = 0
    if reuse is not None:
        labels, means, iters = reuse["labels"], reuse["means"], reuse["kmeans_iterations"]
    # One cluster: every read is in it whatever the centroid is (cluster.c:75-98 finds the nearest of one), so there is
    # no assignment pass, and the one accumulation the reference's loop amounts to (column sums of the bytes) is read off
    # the PMF counters below instead of costing a pass over the symbols of its own.
    while k > 1 and reuse is None:
        counts.zero_()
        acc.zero_()
        with timer.span("kmeans_assign"):
            K.kmeans_assign(sym, n, means, labels=labels, counts=counts)
        with timer.span("kmeans_accumulate"):
            K.kmeans_accumulate(sym, n, labels, k, acc=acc)
        if group is not None:
            dist_mod.all_reduce(acc, group=group)
            dist_mod.all_reduce(counts, group=group)
        moved, status = K.kmeans_update(acc, counts, means)
        iters += 1
        mv = moved[0].item()
        if int(status.item()):
            raise RuntimeError("a cluster went empty during k-means (reference: division by zero, cluster.c:113)")
        if not (mv > threshold) or iters >= 1000:
            break
    torch.cuda.synchronize()
    wall["kmeans"] = time.perf_counter() - t0

    # ---- conditional-PMF counters (codebook.c:193-205) ----
    t1 = time.perf_counter()
    if reuse is not None:
        cnt = reuse["cnt"]
    else:
        cnt = torch.zeros((k, c, 72, 72), dtype=torch.int32, device=dev)
        with timer.span("pmf_hist"):
            K.pmf_hist(sym, n, labels=labels if k > 1 else None, k=k, cnt=cnt)
        if group is not None:
            dist_mod.all_reduce(cnt, group=group)
    if k == 1 and reuse is None:
        # sum over the reads of byte_s = sum over x of (x + 33) * #(X_s = x), exact in int64; a second iteration would
        # recompute the same sums, so the iteration count is 2 if this update moved the centroid by more than the
        # threshold (cluster.c:230-233), else 1
        col = cnt[0].sum(dim=1, dtype=torch.int64)                                   # (c, 72)
        acc[0] = (col * (torch.arange(72, device=dev, dtype=torch.int64) + 33)).sum(dim=1)
        counts[0] = col[0].sum().to(torch.int32)
        moved, status = K.kmeans_update(acc, counts, means)
        iters = 2 if moved[0].item() > threshold else 1
    torch.cuda.synchronize()
    wall["pmf"] = time.perf_counter() - t1

    # ---- WELL draws (well.c:8-46): they depend on the seed only, so they are generated on a side stream while the
    # codebooks (small latency-bound kernels + host work) are being made ----
    if well_state is None:
        well_state = np.full(32, 0x55555555, dtype=np.uint32)
    main_stream = torch.cuda.current_stream()
    side = torch.cuda.Stream(device=dev)
    if reuse is not None and reuse.get("draws") is not None:
        draws = reuse["draws"]
    else:
        draws = K.empty_matrix(n, c)         # allocated on the main stream (the caching allocator pools per stream)
        side.wait_stream(main_stream)
        with torch.cuda.stream(side):
            with timer.span("well_draws"):
                K.well_draws(well_state, first_read, n, c, out=draws)

    # ---- codebooks (codebook.c:208-468), identical on every rank ----
    t2 = time.perf_counter()
    dmat = K.distortion_matrix(dist)
    cbs, deferred = [], 0
    if k == 1:
        cb, deferred = K.generate_codebook(cnt[0], dmat, mode, ratio)
        cbs.append(cb)
    else:
        # the clusters' codebooks are independent and each generation is latency-bound (small kernels, two host round
        # trips per column): one host thread and one HIP stream per cluster (ctypes releases the GIL in the call)
        from concurrent.futures import ThreadPoolExecutor
        torch.cuda.synchronize()
        streams = [torch.cuda.Stream(device=dev) for _ in range(k)]

        def one(j):
            torch.cuda.set_device(dev)
            with torch.cuda.stream(streams[j]):
                return K.generate_codebook(cnt[j], dmat, mode, ratio)

        with ThreadPoolExecutor(max_workers=k) as pool:
            for cb, d in pool.map(one, range(k)):
                cbs.append(cb)
                deferred += d
    ct, qs = K.upload_tables(cbs, dev)
    torch.cuda.synchronize()
    wall["codebook"] = time.perf_counter() - t2

    # ---- quantize (qv_compressor.c:76-135) ----
    t3 = time.perf_counter()
    main_stream.wait_stream(side)
    ddist = torch.from_numpy(dmat).to(dev)
    # the quantized values themselves (the -u product) are written only for a caller that keeps them: the coder needs
    # the quantizer index and the state, the distortion the per-read error
    want = ("qv", "qi", "st", "err") if (keep or keep_device) else ("qi", "st", "err")
    with timer.span("quantize"):
        out = K.quantize(sym, n, ct, qs, draws, ddist, labels=labels if k > 1 else None, k=k, want=want)
    if not want_reuse and reuse is None:
        del draws
    feed = None
    if coder != "gpu" or keep:
        with timer.span("pack_feed"):
            feed = K.pack_feed(out["qi"], out["st"], n)
    if int(out["status"].item()):
        raise RuntimeError("quantizer met a context the codebook does not define")
    torch.cuda.synchronize()
    wall["quantize"] = time.perf_counter() - t3

    # ---- arithmetic coding: `shards` independent reference-format streams on `threads` host threads ----
    t4 = time.perf_counter()
    if coder == "gpu":
        with timer.span("arith_shards"):
            sizes, streams, coded = host.encode_planes_gpu(out["qi"], out["st"], labels if k > 1 else None, n, c,
                                                           [cb.handle for cb in cbs], shards, want_streams=keep,
                                                           work_bytes=work_bytes)
        wall["coder_model_kernels"], wall["coder_interval_kernels"] = (x * 1e-3 for x in host.encode_planes_gpu.last_ms)
        stream_sizes = sizes
    elif coder == "gpu-feed":
        with timer.span("arith_shards"):
            sizes, streams, coded = host.encode_shards_gpu(feed, labels if k > 1 else None, n, c, [cb.handle for cb in cbs],
                                                           shards, want_streams=keep)
        stream_sizes = sizes
    else:
        streams = host.encode_shards_device(feed, labels if k > 1 else None, n, c, [cb.handle for cb in cbs], shards, threads,
                                            device=device_index)
        stream_sizes = [len(s) for s in streams]
    wall["coder"] = time.perf_counter() - t4

    err_sum = (out["err"] / float(c)).sum()
    if group is not None:
        dist_mod.all_reduce(err_sum, group=group)
    distortion = float(err_sum.item()) / float(n_total)
    wall["total"] = time.perf_counter() - t0
    res = dict(stream_bytes=stream_sizes, total_stream_bytes=sum(stream_sizes), distortion=distortion,
               kmeans_iterations=iters, deferred=deferred, wall=wall, codebooks=cbs)
    if want_reuse:
        res["reuse"] = dict(labels=labels, means=means, kmeans_iterations=iters, cnt=cnt, draws=draws)
    if keep_device and coder != "host":
        res.update(coded=coded, cap_per_shard=coded.numel() // shards, qv=out["qv"], labels=labels)
    if keep:
        res.update(streams=streams, qv=out["qv"], labels=labels, cnt=cnt, means=means, err=out["err"], feed=feed, qi=out["qi"], st=out["st"])
    return res


def rd_sweep(sym, n, first_read, n_total, means0, points, **kw):
    """Rate-distortion sweep (reference generate_rd.sh:1-17): `points` is a list of
    (mode, ratio, dist) operating points of the same input; cluster labels, PMF counters and WELL draws are computed for
    the first point only. Every point's outputs equal those of a separate encode_pass. Returns the list of results."""
    out, reuse = [], None
    for mode, ratio, dist in points:
        r = encode_pass(sym, n, first_read, n_total, means0, mode=mode, ratio=ratio, dist=dist, reuse=reuse,
                        want_reuse=reuse is None, **kw)
        if reuse is None:
            reuse = r.pop("reuse")
        out.append(r)
    return out

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
import os
import threading
import time

import numpy as np
import torch

from . import host
from . import kernels as K

LINES_PER_BLOCK = 1_000_000   # reference include/lines.h:12
_RAND_LOCK = threading.Lock()


def reference_centroid_lines(n_total, k, reseed=True):
    """Read indices the reference picks as initial centroids (cluster.c:198-201): libc rand(), fresh-process state."""
    libc = ctypes.CDLL(None)
    blocks = (n_total + LINES_PER_BLOCK - 1) // LINES_PER_BLOCK
    lines = []
    with _RAND_LOCK:                  # libc's generator is the process's: ranks that are threads (LocalWorld) draw one after the other
        if reseed:
            libc.srand(1)
        for _ in range(k):
            b = libc.rand() % blocks
            count = min(LINES_PER_BLOCK, n_total - b * LINES_PER_BLOCK)
            lines.append(b * LINES_PER_BLOCK + libc.rand() % count)
    return lines


def rank_read_range(n_total, world, rank):
    """Contiguous reads [first, last) of `rank` when a file of n_total reads is sharded over `world` ranks: the same
    split as the coder's shards (qvz_shard_bounds: groups of 4 reads dealt evenly), so every rank's first draw (first*C)
    sits on a WELL word boundary (4 reads x C draws = C whole 4-draw words) for any C."""
    groups = n_total // 4
    first = 4 * (groups * rank // world)
    last = n_total if rank == world - 1 else 4 * (groups * (rank + 1) // world)
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


_KEPT_STREAMS = {}


def kept_stream(device, name, beside=()):
    """A side stream that is made once per (device, role) and kept for the life of the process. The runtime spreads streams
    over a handful of hardware queues (four by default) and a queue runs its kernels in order; torch hands out its pooled
    streams round robin, so a fresh `torch.cuda.Stream()` per encode lands on the queue of the stream it is meant to run
    BESIDE every fourth time (measured: every fourth step's codebook phase waited 25 ms for the WELL kernels) - and which
    queue a stream gets depends on what the process created before it (with an RCCL group up, the first stream taken
    shared the default stream's queue: every step's codebook phase waited). So a kept stream is tested, when it is made,
    for running side by side with the streams in `beside` (qvz_hip_streams_overlap, ~1 ms each); a candidate that does
    not is dropped for the next one of torch's pool."""
    key = (str(device), name)
    st = _KEPT_STREAMS.get(key)
    if st is None:
        import ctypes
        from ._lib import check, lib
        for _ in range(12):
            st = torch.cuda.Stream(device=device)
            good = True
            for other in beside:
                flag = ctypes.c_int(0)
                check(lib().qvz_hip_streams_overlap(st.cuda_stream, other.cuda_stream, ctypes.byref(flag)))
                if not flag.value:
                    good = False
                    break
            if good:
                break
        _KEPT_STREAMS[key] = st
    return st


class HipBackend:
    """The compute side of encode_pass: every method is one call into the C-ABI (HIP kernels / host C) on this process's
    GPU. encode_pass itself holds only what ranks have to agree on - which reads a rank owns, what is summed across ranks
    and when, which WELL offset a shard starts from - so that the same schedule can be driven by another backend (the
    tests run it over two gloo ranks with the CPU oracle as backend, tests/cpu_backend.py)."""

    def __init__(self, device_index=0, timer=None):
        self.device = torch.device("cuda", device_index)
        self.device_index = device_index
        self.timer = timer or Timer(False)
        self._side = None
        from ._lib import lib
        lib().qvz_hip_codebook_warmup()     # once per process: glibc's log2 as restated for the device against this host's (include/qvz_codebook.h)

    def zeros(self, shape, dtype):
        return torch.zeros(shape, dtype=dtype, device=self.device)

    def sync(self):
        torch.cuda.synchronize()

    def read_rows(self, sym, idx):
        """symbols of the reads idx (local indices) as a (len(idx), c) uint8 tensor"""
        return sym[:, torch.as_tensor(idx, dtype=torch.long, device=sym.device)].t().contiguous()

    def kmeans_step(self, sym, n, means, labels, counts, acc):
        with self.timer.span("kmeans_step"):
            K.kmeans_step(sym, n, means, labels=labels, counts=counts, acc=acc)

    def kmeans_update(self, acc, counts, means):
        return K.kmeans_update(acc, counts, means)

    def pmf_hist(self, sym, n, labels, k, cnt):
        with self.timer.span("pmf_hist"):
            K.pmf_hist(sym, n, labels=labels, k=k, cnt=cnt)

    def start_draws(self, well_state, first_read, n, c):
        """WELL draws (well.c:8-46) of this shard: they depend on the seed only, so they are generated on a side stream
        while k-means, the counters and the codebooks (small latency-bound kernels + host work) are being made."""
        main_stream = torch.cuda.current_stream()
        self._side = kept_stream(self.device, "well", beside=(main_stream,))
        draws = K.empty_matrix(n, c)         # allocated on the main stream (the caching allocator pools per stream)
        self._side.wait_stream(main_stream)
        with torch.cuda.stream(self._side):
            with self.timer.span("well_draws"):
                K.well_draws(well_state, first_read, n, c, out=draws)
        return draws

    def codebooks(self, cnt, dmat, mode, ratio):
        k = cnt.shape[0]
        cbs, deferred = [], 0
        if k == 1:
            cb, deferred = K.generate_codebook(cnt[0], dmat, mode, ratio)
            return [cb], deferred
        # the clusters' codebooks are independent and each generation is latency-bound (small kernels, host round trips):
        # one host thread and one HIP stream per cluster (ctypes releases the GIL in the call)
        from concurrent.futures import ThreadPoolExecutor
        main = torch.cuda.current_stream()
        # cluster 0 on the caller's stream (idle otherwise), the others on kept streams of their own
        streams = [main]
        for j in range(1, k):
            beside = tuple(streams) + ((self._side,) if self._side is not None else ())
            streams.append(kept_stream(self.device, f"codebook{j}", beside=beside))
        for st in streams[1:]:
            st.wait_stream(main)      # the counters are ready when the main stream gets here; NOT a device-wide synchronise:
                                      # that would wait for the WELL draws on the side stream, which are meant to run beside this

        def one(j):
            torch.cuda.set_device(self.device)
            with torch.cuda.stream(streams[j]):
                return K.generate_codebook(cnt[j], dmat, mode, ratio)

        with ThreadPoolExecutor(max_workers=k) as pool:
            for cb, d in pool.map(one, range(k)):
                cbs.append(cb)
                deferred += d
        return cbs, deferred

    def quantize(self, sym, n, cbs, draws, dmat, labels, k, want):
        ct, qs = K.upload_tables(cbs, self.device)
        if self._side is not None:
            torch.cuda.current_stream().wait_stream(self._side)
        ddist = torch.from_numpy(dmat).to(self.device)
        with self.timer.span("quantize"):
            return K.quantize(sym, n, ct, qs, draws, ddist, labels=labels, k=k, want=want)

    def code(self, out, labels, n, c, cbs, shards, threads, coder, keep, work_bytes):
        """-> (stream sizes, streams as bytes | None, coded device tensor | None, feed | None, (model ms, coder ms) | None)"""
        handles = [cb.handle for cb in cbs]
        feed = None
        if coder != "gpu" or keep:
            with self.timer.span("pack_feed"):
                feed = K.pack_feed(out["qi"], out["st"], n)
        if coder == "gpu":
            with self.timer.span("arith_shards"):
                sizes, streams, coded = host.encode_planes_gpu(out["qi"], out["st"], labels, n, c, handles, shards,
                                                               want_streams=keep, work_bytes=work_bytes)
            return sizes, streams, coded, feed, host.encode_planes_gpu.last_ms
        streams = host.encode_shards_device(feed, labels, n, c, handles, shards, threads, device=self.device_index)
        return [len(s) for s in streams], streams, None, feed, None


def all_reduce_sum(t, group):
    """SUM of `t` over the ranks of `group`, in place: a torch.distributed process group (RCCL when the ranks are the GPUs of a
    node - backend "nccl" -, gloo between CPU ranks), or a group object with an all_reduce method of its own (LocalWorld)."""
    if hasattr(group, "all_reduce"):
        group.all_reduce(t)
    else:
        import torch.distributed as dist_mod
        dist_mod.all_reduce(t, group=group)


class LocalWorld:
    """`world` ranks as THREADS of one process that share one device: the read-sharded schedule of encode_pass with more parts
    than a box has GPUs (a test box has one, and allows few processes on it), the counterpart of the C driver's host-staged sum
    for an ordinal listed more than once (host/qvz_multi.c). group(rank) is what a rank passes to encode_pass as `group`; the
    sum is taken when all ranks have arrived. serial(backend) wraps a compute backend so that one rank at a time is inside the
    C library (its coder keeps per-device buffers and streams; the library is not re-entrant by contract)."""

    def __init__(self, world):
        import threading
        self.world = world
        self._barrier = threading.Barrier(world)
        self._slots = [None] * world
        self._lock = threading.Lock()

    def group(self, rank):
        return _LocalGroup(self, rank)

    def serial(self, backend):
        return _SerialBackend(backend, self._lock)


class _LocalGroup:
    def __init__(self, world, rank):
        self._w, self.rank = world, rank

    def all_reduce(self, t):
        w = self._w
        w._slots[self.rank] = t
        w._barrier.wait()                                  # every rank's tensor is there (same stream: in issue order)
        total = w._slots[0].clone()
        for other in w._slots[1:]:
            total += other
        w._barrier.wait()                                  # ... and has been read by everybody before anyone overwrites its own
        t.copy_(total)


class _SerialBackend:
    def __init__(self, inner, lock):
        self._inner, self._lock = inner, lock

    def __getattr__(self, name):
        attr = getattr(self._inner, name)
        if not callable(attr):
            return attr

        def locked(*a, **kw):
            with self._lock:
                return attr(*a, **kw)
        return locked


def fetch_centroids(backend, sym, n, first_read, n_total, k, group=None, lines=None):
    """The k reads the reference copies as initial centroids (cluster.c:198-201): their global numbers come from libc
    rand() in a fresh process's state, identical on every rank; each is fetched from the rank that owns it and made
    global by a SUM over ranks (k * c bytes; SURVEY.md section 8e, collective 1)."""
    if lines is None:
        lines = reference_centroid_lines(n_total, k)
    c = sym.shape[0]
    means = backend.zeros((k, c), torch.int32)
    mine = [(j, ln - first_read) for j, ln in enumerate(lines) if first_read <= ln < first_read + n]
    if mine:
        rows = backend.read_rows(sym, [i for _, i in mine])
        means[torch.as_tensor([j for j, _ in mine], dtype=torch.long, device=means.device)] = rows.to(torch.int32)
    if group is not None:
        all_reduce_sum(means, group)
    return means.to(torch.uint8).contiguous()


def encode_pass(sym, n, first_read, n_total, means0, *, clusters=1, mode=host.MODE_RATIO, ratio=0.5, dist="M",
                threshold=4.0, well_state=None, shards=1, threads=1, group=None, timer=None, keep=False, device_index=0,
                coder="host", work_bytes=None, keep_device=False, reuse=None, want_reuse=False, backend=None):
    """Runs k-means -> PMF -> codebooks -> WELL -> quantize -> feed -> sharded arithmetic coding for the reads
    [first_read, first_read+n) held in `sym` (this rank's shard of a file with n_total reads).
    coder: "host" = `threads` host threads over the packed feed (qvz_encode_shards_device); "gpu" = the column-parallel
    model + lane-per-stream coder fed from the quantizer's planes (qvz_hip_encode_planes). Byte-identical streams.

    means0: (clusters, c) uint8 device tensor of the initial centroids (symbols), or None: fetch_centroids gets them from
    the ranks that own them. first_read must be a multiple of 4.
    Returns a dict with stream sizes, distortion, codebooks, phase wall times; with keep=True also the device outputs
    and the streams as host bytes; with keep_device=True (GPU coders) the coded streams stay on the device
    (`coded`, stream g at g * `cap_per_shard`) next to `qv` and `labels`.
    reuse / want_reuse: the cluster labels, the PMF counters and the WELL draws do not depend on -f / -r / -d, so a
    rate-distortion sweep (the reference's generate_rd.sh: 20 encodes of one file) computes them once: pass
    want_reuse=True on the first operating point and hand its result["reuse"] to the others (rd_sweep below).
    backend: who computes (default: HipBackend, the HIP kernels on this process's GPU).
    """
    c, ld = sym.shape
    k = clusters
    assert first_read % 4 == 0, "shards must start on a WELL word boundary"
    be = backend or HipBackend(device_index, timer)
    wall = {}
    t0 = time.perf_counter()

    # ---- k-means (cluster.c:212-244) ----
    if means0 is None:
        means0 = fetch_centroids(be, sym, n, first_read, n_total, k, group)
    means = means0.clone()
    labels = be.zeros(ld, torch.uint8)
    counts = be.zeros(k, torch.int32)
    acc = be.zeros((k, c), torch.int64)
    iters = 0
    means_assigned = means0          # the means the LAST assignment pass measured against: what the final labels belong to (cluster.c:221-233)
    if reuse is not None:
        labels, means, iters = reuse["labels"], reuse["means"], reuse["kmeans_iterations"]
        means_assigned = reuse.get("means_assigned", means)
    # One cluster: every read is in it whatever the centroid is (cluster.c:75-98 finds the nearest of one), so there is
    # no assignment pass, and the one accumulation the reference's loop amounts to (column sums of the bytes) is read off
    # the PMF counters below instead of costing a pass over the symbols of its own.
    while k > 1 and reuse is None:
        counts.zero_()
        acc.zero_()
        be.kmeans_step(sym, n, means, labels, counts, acc)
        if group is not None:
            all_reduce_sum(acc, group)
            all_reduce_sum(counts, group)
        means_assigned = means.clone()
        moved, status = be.kmeans_update(acc, counts, means)
        iters += 1
        mv = moved[0].item()
        if int(status.item()):
            raise RuntimeError("a cluster went empty during k-means (reference: division by zero, cluster.c:113)")
        if not (mv > threshold) or iters >= 1000:
            break
    be.sync()
    wall["kmeans"] = time.perf_counter() - t0

    # ---- conditional-PMF counters (codebook.c:193-205) ----
    t1 = time.perf_counter()
    if reuse is not None:
        cnt = reuse["cnt"]
    else:
        cnt = be.zeros((k, c, 72, 72), torch.int32)
        be.pmf_hist(sym, n, labels if k > 1 else None, k, cnt)
        if group is not None:
            all_reduce_sum(cnt, group)
    if k == 1 and reuse is None:
        # sum over the reads of symbol_s = sum over x of x * #(X_s = x), exact in int64 (means are kept as symbols, byte - 33:
        # floor((sum + 33 n) / n) = floor(sum / n) + 33); a second iteration would recompute the same sums, so the iteration
        # count is 2 if this update moved the centroid by more than the threshold (cluster.c:230-233), else 1
        col = cnt[0].sum(dim=1, dtype=torch.int64)                                   # (c, 72)
        acc[0] = (col * torch.arange(72, device=col.device, dtype=torch.int64)).sum(dim=1)
        counts[0] = col[0].sum().to(torch.int32)
        moved, status = be.kmeans_update(acc, counts, means)
        iters = 2 if moved[0].item() > threshold else 1
    be.sync()
    wall["pmf"] = time.perf_counter() - t1

    # ---- WELL draws of this shard: generator offset first_read * c (jump-ahead). They depend on the seed only and run on a
    # side stream beside the codebook generation (after the bandwidth-bound passes above, which they would slow down) ----
    if well_state is None:
        well_state = np.full(32, 0x55555555, dtype=np.uint32)
    if reuse is not None and reuse.get("draws") is not None:
        draws = reuse["draws"]
    else:
        draws = be.start_draws(well_state, first_read, n, c)

    # ---- codebooks (codebook.c:208-468), identical on every rank ----
    t2 = time.perf_counter()
    dmat = K.distortion_matrix(dist)
    cbs, deferred = be.codebooks(cnt, dmat, mode, ratio)
    be.sync()
    wall["codebook"] = time.perf_counter() - t2

    # ---- quantize (qv_compressor.c:76-135) ----
    t3 = time.perf_counter()
    # the quantized values themselves (the -u product) are written only for a caller that keeps them: the coder needs
    # the quantizer index and the state, the distortion the per-read error
    want = ("qv", "qi", "st", "err") if (keep or keep_device) else ("qi", "st", "err")
    out = be.quantize(sym, n, cbs, draws, dmat, labels if k > 1 else None, k, want)
    if not want_reuse and reuse is None:
        del draws
    if int(out["status"].item()):
        raise RuntimeError("quantizer met a context the codebook does not define")
    be.sync()
    wall["quantize"] = time.perf_counter() - t3

    # ---- arithmetic coding: `shards` independent reference-format streams ----
    t4 = time.perf_counter()
    stream_sizes, streams, coded, feed, kernel_ms = be.code(out, labels if k > 1 else None, n, c, cbs, shards, threads, coder,
                                                            keep, work_bytes)
    if kernel_ms is not None:
        wall["coder_model_kernels"], wall["coder_interval_kernels"] = (x * 1e-3 for x in kernel_ms)
    wall["coder"] = time.perf_counter() - t4

    err_sum = (out["err"] / float(c)).sum()
    if group is not None:
        all_reduce_sum(err_sum, group)
    distortion = float(err_sum.item()) / float(n_total)
    wall["total"] = time.perf_counter() - t0
    res = dict(stream_bytes=stream_sizes, total_stream_bytes=sum(stream_sizes), distortion=distortion,
               kmeans_iterations=iters, deferred=deferred, wall=wall, codebooks=cbs)
    if want_reuse:
        res["reuse"] = dict(labels=labels, means=means, means_assigned=means_assigned, kmeans_iterations=iters, cnt=cnt, draws=draws)
    if keep_device and coder != "host":
        res.update(coded=coded, cap_per_shard=coded.numel() // shards, qv=out["qv"], labels=labels)
    if keep:
        res.update(streams=streams, qv=out["qv"], labels=labels, cnt=cnt, means=means, means_assigned=means_assigned, err=out["err"], feed=feed, qi=out["qi"], st=out["st"])
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

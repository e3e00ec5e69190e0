"""Bindings of the host C layer (include/qvz_coder.h, include/qvz_encode.h). No torch needed here: the arithmetic
coder and the file-level driver are plain C inside libqvzhip.so."""
import ctypes as C

import numpy as np

from ._lib import QvzHipError, lib

MODE_RATIO, MODE_FIXED = 0, 1
DIST = {"A": 1, "M": 2, "L": 3}
SEED_TIME, SEED_DEBUG, SEED_EXPLICIT = 0, 1, 2


class Options(C.Structure):
    """qvz_options_t (include/qvz_encode.h)."""
    _fields_ = [("verbose", C.c_int), ("stats", C.c_int), ("mode", C.c_int), ("clusters", C.c_int),
                ("distortion", C.c_int), ("ratio", C.c_double), ("cluster_threshold", C.c_double),
                ("uncompressed_name", C.c_char_p), ("device", C.c_int), ("seed_mode", C.c_int),
                ("well_state", C.c_uint32 * 32), ("chunk_reads", C.c_uint64), ("dist_file", C.c_char_p), ("input_format", C.c_int)]


class Result(C.Structure):
    """qvz_result_t (include/qvz_encode.h)."""
    _fields_ = [("lines", C.c_uint64), ("columns", C.c_uint32), ("stream_bytes", C.c_uint64),
                ("file_bytes", C.c_uint64), ("distortion", C.c_double), ("kmeans_iterations", C.c_uint32),
                ("deferred_decisions", C.c_uint64), ("t_upload", C.c_double), ("t_cluster", C.c_double),
                ("t_stats", C.c_double), ("t_codebook", C.c_double), ("t_encode", C.c_double), ("t_total", C.c_double),
                ("t_quantize", C.c_double), ("t_coder", C.c_double), ("t_egress", C.c_double)]


def default_options():
    o = Options()
    lib().qvz_options_default(C.byref(o))
    return o


def encode_file(in_path, out_path, clusters=1, mode=MODE_RATIO, ratio=0.5, dist="M", threshold=4.0, u_path=None,
                well_state=None, seed_mode=None, chunk_reads=0, device=0):
    """Whole-file encode on the GPU (the reference's encode(), src/main.c:18-127). Returns a Result."""
    o = default_options()
    o.clusters, o.mode, o.ratio, o.distortion, o.cluster_threshold = clusters, mode, ratio, DIST.get(dist, dist), threshold
    o.device, o.chunk_reads = device, chunk_reads
    if u_path:
        o.uncompressed_name = u_path.encode()
    if well_state is not None:
        o.seed_mode = SEED_EXPLICIT
        for i, v in enumerate(np.asarray(well_state, dtype=np.uint32)):
            o.well_state[i] = int(v)
    elif seed_mode is not None:
        o.seed_mode = seed_mode
    r = Result()
    rc = lib().qvz_encode_file(in_path.encode(), out_path.encode(), C.byref(o), C.byref(r))
    if rc:
        raise QvzHipError(f"qvz_encode_file failed ({rc}): {lib().qvz_encode_last_error().decode(errors='replace')}")
    return r


def encode_file_sharded(in_path, out_prefix, shards, clusters=1, mode=MODE_RATIO, ratio=0.5, dist="M", threshold=4.0,
                        u_path=None, seed_mode=None, device=0):
    """Whole-file encode with sharded output (qvz_encode_file_sharded): writes out_prefix.0000 ... as reference-format
    files coded by the GPU coder. Returns a Result."""
    o = default_options()
    o.clusters, o.mode, o.ratio, o.distortion, o.cluster_threshold = clusters, mode, ratio, DIST.get(dist, dist), threshold
    o.device = device
    if u_path:
        o.uncompressed_name = u_path.encode()
    if seed_mode is not None:
        o.seed_mode = seed_mode
    r = Result()
    rc = lib().qvz_encode_file_sharded(in_path.encode(), out_prefix.encode(), C.byref(o), shards, C.byref(r))
    if rc:
        raise QvzHipError(f"qvz_encode_file_sharded failed ({rc}): {lib().qvz_encode_last_error().decode(errors='replace')}")
    return r


class ShardsetInfo(C.Structure):
    """qvz_shardset_info_t (include/qvz_shardset.h)."""
    _fields_ = [("clusters", C.c_uint32), ("columns", C.c_uint32), ("shards", C.c_uint32), ("lines", C.c_uint64),
                ("header_bytes", C.c_uint64), ("codebook_bytes", C.c_uint64), ("stream_bytes", C.c_uint64),
                ("file_bytes", C.c_uint64)]


def encode_file_shardset(in_path, out_path, shards, clusters=1, mode=MODE_RATIO, ratio=0.5, dist="M", threshold=4.0,
                         u_path=None, seed_mode=None, device=0):
    """Whole-file sharded encode into ONE shard-set file (qvz_encode_file_shardset): codebooks stored once."""
    o = default_options()
    o.clusters, o.mode, o.ratio, o.distortion, o.cluster_threshold = clusters, mode, ratio, DIST.get(dist, dist), threshold
    o.device = device
    if u_path:
        o.uncompressed_name = u_path.encode()
    if seed_mode is not None:
        o.seed_mode = seed_mode
    r = Result()
    rc = lib().qvz_encode_file_shardset(in_path.encode(), out_path.encode(), C.byref(o), shards, C.byref(r))
    if rc:
        raise QvzHipError(f"qvz_encode_file_shardset failed ({rc}): {lib().qvz_encode_last_error().decode(errors='replace')}")
    return r


def encode_file_devices(in_path, out_path, devices, shards, clusters=1, mode=MODE_RATIO, ratio=0.5, dist="M", threshold=4.0,
                        u_path=None, seed_mode=None):
    """One file on several devices of this process (qvz_encode_file_devices): the shard set a single device writes for
    the same `shards`."""
    o = default_options()
    o.clusters, o.mode, o.ratio, o.distortion, o.cluster_threshold = clusters, mode, ratio, DIST.get(dist, dist), threshold
    if u_path:
        o.uncompressed_name = u_path.encode()
    if seed_mode is not None:
        o.seed_mode = seed_mode
    r = Result()
    dv = (C.c_int * len(devices))(*devices)
    rc = lib().qvz_encode_file_devices(in_path.encode(), out_path.encode(), C.byref(o), dv, len(devices), shards, C.byref(r))
    if rc:
        raise QvzHipError(f"qvz_encode_file_devices failed ({rc}): {lib().qvz_encode_last_error().decode(errors='replace')}")
    return r


class ResidentProducts(C.Structure):
    """qvz_resident_products_t (include/qvz_encode.h)."""
    _fields_ = [("d_coded", C.c_void_p), ("cap_per_stream", C.c_uint64), ("d_streams", C.c_void_p), ("h_sizes", C.POINTER(C.c_uint64)),
                ("shards", C.c_uint32), ("clusters", C.c_uint32),
                ("stream_bytes", C.c_uint64), ("d_qv", C.c_void_p), ("d_labels", C.c_void_p), ("codebooks", C.POINTER(C.c_void_p)),
                ("well_state", C.c_uint32 * 32), ("model_kernels_ms", C.c_double), ("coder_kernels_ms", C.c_double),
                ("kmeans_kernels_ms", C.c_double), ("pmf_kernel_ms", C.c_double), ("quantize_kernel_ms", C.c_double)]


ALLREDUCE_FN = C.CFUNCTYPE(C.c_int, C.c_void_p, C.c_void_p, C.c_uint64, C.c_int, C.c_void_p)


class _DevicePtr:
    """A device buffer the C side owns, seen as a tensor (the __cuda_array_interface__ protocol; no copy)."""

    def __init__(self, ptr, count, typestr):
        self.__cuda_array_interface__ = {"shape": (int(count),), "typestr": typestr, "data": (int(ptr), False), "version": 2}


def device_view(ptr, count, dtype="|u1"):
    import torch
    return torch.as_tensor(_DevicePtr(ptr, count, dtype), device="cuda")


def group_allreduce(group):
    """qvz_allreduce_fn over a process group (torch.distributed: RCCL between the GPUs of a node, gloo in rehearsals) or a
    pipeline.LocalWorld group: the sum the parts of one file owe each other."""
    from .pipeline import all_reduce_sum

    def cb(_ctx, d_buf, count, width, _stream):
        try:
            t = device_view(d_buf, count, {4: "<i4", 8: "<i8", -8: "<f8"}[width])     # two's complement: the unsigned sum bit for bit
            all_reduce_sum(t, group)
            return 0
        except Exception as exc:  # noqa: BLE001 - reported through the C side's error code
            cb.error = repr(exc)
            return 1
    cb.error = None
    fn = ALLREDUCE_FN(cb)
    fn._py = cb
    return fn


class Resident:
    """qvz_resident_* (include/qvz_encode.h): the whole hot path in C over a symbol matrix that is in HBM - what bench.py times.
    The handle keeps its device buffers between encodes."""

    def __init__(self, device=0):
        self.h = C.c_void_p()
        rc = lib().qvz_resident_open(device, C.byref(self.h))
        if rc:
            raise QvzHipError(f"qvz_resident_open failed ({rc}): {lib().qvz_encode_last_error().decode(errors='replace')}")

    def encode(self, sym, n, shards, clusters=1, mode=MODE_RATIO, ratio=0.5, dist="M", threshold=4.0, want_qv=False, first_read=0,
               n_total=None, group=None, centroid_lines=None, seed_mode=SEED_DEBUG, well_state=None):
        """-> (Result, ResidentProducts). sym: (c, ld) uint8 device tensor of reads [first_read, first_read + n) of n_total."""
        o = default_options()
        o.clusters, o.mode, o.ratio, o.distortion, o.cluster_threshold = clusters, mode, ratio, DIST.get(dist, dist), threshold
        o.seed_mode = seed_mode
        if well_state is not None:
            o.seed_mode = SEED_EXPLICIT
            for i, v in enumerate(np.asarray(well_state, dtype=np.uint32)):
                o.well_state[i] = int(v)
        c, ld = sym.shape
        fn = group_allreduce(group) if group is not None else None
        lines = (C.c_uint64 * clusters)(*[int(x) for x in centroid_lines]) if centroid_lines is not None else None
        r = Result()
        rc = lib().qvz_resident_encode(self.h, C.c_void_p(sym.data_ptr()), n, c, ld, first_read, n if n_total is None else n_total,
                                       C.byref(o), shards, int(want_qv), lines, C.cast(fn, C.c_void_p) if fn else None, None, C.byref(r))
        if rc:
            extra = f" [all-reduce: {fn._py.error}]" if fn is not None and fn._py.error else ""
            raise QvzHipError(f"qvz_resident_encode failed ({rc}): {lib().qvz_encode_last_error().decode(errors='replace')}{extra}")
        p = ResidentProducts()
        rc = lib().qvz_resident_products(self.h, C.byref(p))
        if rc:
            raise QvzHipError(f"qvz_resident_products failed ({rc}): {lib().qvz_encode_last_error().decode(errors='replace')}")
        return r, p

    @staticmethod
    def codebook_bytes(p):
        """the codebooks of the last encode in the file format (write_codebook, codebook.c:500-555), one bytes object per cluster"""
        out = []
        for j in range(p.clusters):
            h = C.c_void_p(p.codebooks[j])
            buf = np.zeros(lib().qvz_codebook_write(None, h), dtype=np.uint8)
            lib().qvz_codebook_write(buf.ctypes.data_as(C.c_void_p), h)
            out.append(buf.tobytes())
        return out

    def pack(self):
        """qvz_resident_pack: the last encode's streams back to back on the device -> uint8 device tensor (a view)"""
        d = C.c_void_p()
        rc = lib().qvz_resident_pack(self.h, C.byref(d))
        if rc:
            raise QvzHipError(f"qvz_resident_pack failed ({rc}): {lib().qvz_encode_last_error().decode(errors='replace')}")
        p = ResidentProducts()
        lib().qvz_resident_products(self.h, C.byref(p))
        return device_view(d.value, max(int(p.stream_bytes), 1))[: int(p.stream_bytes)]

    def streams(self, p):
        """the last encode's streams as a list of bytes (packed on the device, then copied)"""
        sizes = [int(p.h_sizes[g]) for g in range(p.shards)]
        packed = self.pack().cpu().numpy().tobytes()
        out, at = [], 0
        for sz in sizes:
            out.append(packed[at:at + sz])
            at += sz
        return out

    def close(self):
        if self.h:
            lib().qvz_resident_close(self.h)
            self.h = C.c_void_p()

    def __del__(self):
        try:
            self.close()
        except Exception:  # noqa: BLE001 - interpreter shutdown
            pass


class SweepPoint(C.Structure):
    """qvz_sweep_point_t (include/qvz_encode.h)."""
    _fields_ = [("mode", C.c_int), ("ratio", C.c_double), ("distortion", C.c_int), ("dist_file", C.c_char_p)]


def sweep_points(text, dist="M", dist_file=None):
    """qvz_sweep_parse: "rd" (the reference script's twenty points) or "<f|r>:<v>,<v>,..." -> list of (mode, ratio)."""
    o = default_options()
    o.distortion = DIST.get(dist, dist)
    if dist_file:
        o.dist_file = dist_file.encode()
    pts = (SweepPoint * 100)()
    n = lib().qvz_sweep_parse(text.encode(), C.byref(o), pts, 100)
    return [(pts[i].mode, pts[i].ratio) for i in range(n)]


def encode_file_sweep(in_path, out_path, points, shards=0, keep_all=True, clusters=1, dist="M", threshold=4.0, u_path=None,
                      seed_mode=None, device=0, stats=False, dist_file=None):
    """Rate-distortion sweep in one job (qvz_encode_file_sweep; reference generate_rd.sh:1-17): `points` is a list of
    (mode, ratio) or (mode, ratio, dist) operating points; labels and counters are made once. Outputs out_path.NN (-u: u_path.NN)
    when keep_all. Returns the list of Results."""
    o = default_options()
    o.clusters, o.cluster_threshold, o.device, o.stats = clusters, threshold, device, int(stats)
    if u_path:
        o.uncompressed_name = u_path.encode()
    if seed_mode is not None:
        o.seed_mode = seed_mode
    pts = (SweepPoint * len(points))()
    keep = []
    for i, pt in enumerate(points):
        d = pt[2] if len(pt) > 2 else dist
        pts[i].mode, pts[i].ratio = pt[0], pt[1]
        if d == "D" or (isinstance(d, str) and d not in DIST):
            keep.append((dist_file if d == "D" else d).encode())
            pts[i].distortion, pts[i].dist_file = 4, keep[-1]
        else:
            pts[i].distortion = DIST.get(d, d)
    res = (Result * len(points))()
    rc = lib().qvz_encode_file_sweep(in_path.encode(), out_path.encode(), C.byref(o), pts, len(points), shards, int(keep_all), res)
    if rc:
        raise QvzHipError(f"qvz_encode_file_sweep failed ({rc}): {lib().qvz_encode_last_error().decode(errors='replace')}")
    return list(res)


def shardset_stat(path):
    si = ShardsetInfo()
    rc = lib().qvz_shardset_stat(path.encode(), C.byref(si))
    if rc:
        raise QvzHipError(f"qvz_shardset_stat failed ({rc}): {lib().qvz_encode_last_error().decode(errors='replace')}")
    return si


def shardset_split(path, prefix, first=0, count=0, device=0):
    rc = lib().qvz_shardset_split(path.encode(), prefix.encode(), first, count, device)
    if rc:
        raise QvzHipError(f"qvz_shardset_split failed ({rc}): {lib().qvz_encode_last_error().decode(errors='replace')}")


def gather_streams(coded, cap_per_shard, sizes):
    """Packs the streams of a GPU coder call (stream g at g * cap_per_shard) back to back on the device."""
    import torch
    shards = len(sizes)
    total = int(sum(sizes))
    packed = torch.empty(max(total, 1), dtype=torch.uint8, device=coded.device)
    sz = (C.c_uint64 * shards)(*[int(x) for x in sizes])
    rc = lib().qvz_hip_gather_streams(C.c_void_p(coded.data_ptr()), cap_per_shard, sz, shards, C.c_void_p(packed.data_ptr()),
                                      C.c_void_p(torch.cuda.current_stream().cuda_stream))
    if rc:
        raise QvzHipError(f"qvz_hip_gather_streams failed ({rc}): {lib().qvz_hip_last_error().decode(errors='replace')}")
    return packed[:total]


def shardset_write(path, columns, lines, well_state, codebook_handles, packed_host, sizes):
    """Writes a shard set (include/qvz_shardset.h) from streams packed back to back in host memory (numpy uint8 / pinned
    tensor). Returns its ShardsetInfo."""
    k, shards = len(codebook_handles), len(sizes)
    arr = (C.c_void_p * k)(*codebook_handles)
    sz = (C.c_uint64 * shards)(*[int(x) for x in sizes])
    ws = np.ascontiguousarray(well_state, dtype=np.uint32)
    libc = C.CDLL(None)
    libc.fopen.restype = C.c_void_p
    libc.fopen.argtypes = [C.c_char_p, C.c_char_p]
    libc.fclose.argtypes = [C.c_void_p]
    f = libc.fopen(path.encode(), b"wb")
    if not f:
        raise OSError(f"cannot open {path}")
    si = ShardsetInfo()
    ptr = packed_host.data_ptr() if hasattr(packed_host, "data_ptr") else packed_host.ctypes.data
    rc = lib().qvz_shardset_write(f, k, columns, lines, shards, ws.ctypes.data_as(C.POINTER(C.c_uint32)), arr, C.c_void_p(ptr), 0, sz,
                                  C.byref(si))
    libc.fclose(f)
    if rc:
        raise QvzHipError(f"qvz_shardset_write failed ({rc}): {lib().qvz_encode_last_error().decode(errors='replace')}")
    return si


def ingest_rows(text_host, n, columns, sym):
    """qvz_ingest_rows: host text rows (numpy uint8 / tensor, n * (columns+1) bytes) -> the device symbol matrix `sym`
    through the C driver's pinned staging path. Returns the number of bad bytes."""
    bad = C.c_uint32(0)
    ptr = text_host.data_ptr() if hasattr(text_host, "data_ptr") else text_host.ctypes.data
    rc = lib().qvz_ingest_rows(C.c_void_p(ptr), n, columns, C.c_void_p(sym.data_ptr()), sym.shape[1], C.byref(bad))
    if rc:
        raise QvzHipError(f"qvz_ingest_rows failed ({rc}): {lib().qvz_encode_last_error().decode(errors='replace')}")
    return int(bad.value)


def egress_rows(sym, n, path):
    """A (columns, ld) uint8 device matrix written to `path` as text rows (qvz_egress_rows: pinned slabs, threaded writes)."""
    import torch
    torch.cuda.synchronize()
    c, ld = sym.shape
    rc = lib().qvz_egress_rows(C.c_void_p(sym.data_ptr()), n, c, ld, path.encode())
    if rc:
        raise QvzHipError(f"qvz_egress_rows failed ({rc}): {lib().qvz_encode_last_error().decode(errors='replace')}")


def shard_bounds(n, shards):
    """first[g] for g = 0..shards; every shard starts at a multiple of 4 reads (a WELL word boundary)."""
    first = (C.c_uint64 * (shards + 1))()
    lib().qvz_shard_bounds(n, shards, first)
    return [int(x) for x in first]


def encode_shards_device(feed, labels, n, columns, codebook_handles, shards, threads, device=0):
    """feed: device tensor (n, columns) int16; labels: device uint8 tensor or None. Returns list of bytes streams.
    The C side runs `threads` host threads, each pulling its shards' feed from HBM and coding them."""
    k = len(codebook_handles)
    arr = (C.c_void_p * k)(*codebook_handles)
    streams = (C.c_void_p * shards)()
    sizes = (C.c_uint64 * shards)()
    rc = lib().qvz_encode_shards_device(C.c_void_p(feed.data_ptr()), C.c_void_p(labels.data_ptr()) if labels is not None else None,
                                        n, columns, k, arr, shards, threads, device, streams, sizes)
    if rc:
        raise QvzHipError(f"qvz_encode_shards_device failed ({rc}): {lib().qvz_hip_last_error().decode(errors='replace')}")
    libc = C.CDLL(None)
    libc.free.argtypes = [C.c_void_p]
    out = []
    for g in range(shards):
        out.append(C.string_at(streams[g], sizes[g]) if sizes[g] else b"")
        libc.free(streams[g])
    return out


_KEPT_WORK = {}


def kept_workspace(device, nbytes):
    """The record workspace of the GPU coder (~100 GB at the BASELINE shapes), allocated once per device and kept. Taken
    afresh from torch's caching allocator for every encode it was usually the cached block again - but the allocator splits
    large free blocks for smaller requests (the 15 GB draw matrix of the next encode), and an encode that then found no
    block of 100 GB cost a 100 ms allocation, sometimes after the cache had been emptied to make room (seen as one step in
    twenty taking 315 instead of 216 ms). release_workspace() gives it back."""
    import torch
    key = str(device)
    w = _KEPT_WORK.get(key)
    if w is None or w.numel() < nbytes:
        _KEPT_WORK.pop(key, None)
        w = None
        w = _KEPT_WORK[key] = torch.empty(nbytes, dtype=torch.uint8, device=device)
    return w


def release_workspace(device=None):
    """Gives back the kept record workspace (and the coder's kept internal buffers of the current device)."""
    for key in [k for k in _KEPT_WORK if device is None or k == str(device)]:
        del _KEPT_WORK[key]
    lib().qvz_hip_encode_planes_release()


def planes_bytes_per_read(shards, columns, k):
    """Record-workspace bytes that one read of every stream takes in qvz_hip_encode_planes: 8 bytes per record, streams
    padded to a multiple of 64, records per read (columns, behind 8 reserved positions - the cluster id and seven zero
    records - when k > 1) padded to a multiple of 8."""
    return 8 * ((shards + 63) // 64 * 64) * ((columns + (8 if k > 1 else 0) + 7) // 8 * 8)


def encode_planes_gpu(qi, st, labels, n, columns, codebook_handles, shards, cap_per_shard=None, want_streams=True,
                      work_bytes=None, bounds=None):
    """GPU sharded coder fed from the quantizer's planes (qvz_hip_encode_planes): the model runs with a lane per
    (stream, column), the interval arithmetic with a lane per stream. qi/st: (columns, ld) uint8 device tensors.
    work_bytes: size of the record workspace (default: everything at once, capped at 96 GiB).
    bounds: the cut into shards (shards + 1 read numbers, every start a multiple of 4; default qvz_shard_bounds).
    Returns (sizes, streams|None, out_tensor)."""
    import torch
    k = len(codebook_handles)
    arr = (C.c_void_p * k)(*codebook_handles)
    ld = qi.shape[1]
    cap_given = cap_per_shard
    if cap_per_shard is None:
        per = (n + shards - 1) // shards + 4
        cap_per_shard = (per * columns * 3) // 4 + 512          # 6 bits/symbol + the 256-byte margin the emitter wants
    cap_per_shard = (cap_per_shard + 3) & ~3
    per_read = planes_bytes_per_read(shards, columns, k)
    longest = ((n + shards - 1) // shards + 7) & ~3
    if bounds is not None:
        per = max(b - a for a, b in zip(bounds, bounds[1:]))
        longest = (per + 7) & ~3
        if cap_given is None:
            cap_per_shard = (((per + 4) * columns * 3) // 4 + 512 + 3) & ~3
    if work_bytes is None:
        # default: everything at once, capped at 96 GiB of the 288; fewer, longer segments mean fewer save/restore round trips of the
        # model tables between segments
        work_bytes = min(per_read * longest, 96 << 30)
    work_bytes = max(work_bytes, per_read * 4)
    work = kept_workspace(qi.device, work_bytes)
    out = torch.empty(shards * cap_per_shard, dtype=torch.uint8, device=qi.device)
    sizes = (C.c_uint64 * shards)()
    stream = C.c_void_p(torch.cuda.current_stream().cuda_stream)
    lab = C.c_void_p(labels.data_ptr()) if labels is not None else None
    if bounds is None:
        rc = lib().qvz_hip_encode_planes(C.c_void_p(qi.data_ptr()), C.c_void_p(st.data_ptr()), ld, lab, n, columns, k, arr, shards,
                                         C.c_void_p(out.data_ptr()), cap_per_shard, sizes, C.c_void_p(work.data_ptr()), work_bytes,
                                         stream)
    else:
        hb = (C.c_uint64 * (shards + 1))(*[int(b) for b in bounds])
        rc = lib().qvz_hip_encode_planes_bounds(C.c_void_p(qi.data_ptr()), C.c_void_p(st.data_ptr()), ld, lab, n, columns, k, arr,
                                                shards, hb, C.c_void_p(out.data_ptr()), cap_per_shard, sizes,
                                                C.c_void_p(work.data_ptr()), work_bytes, stream)
    if rc:
        raise QvzHipError(f"qvz_hip_encode_planes failed ({rc}): {lib().qvz_hip_last_error().decode(errors='replace')}")
    a, b = C.c_double(), C.c_double()
    lib().qvz_hip_encode_planes_last_ms(C.byref(a), C.byref(b))
    encode_planes_gpu.last_ms = (a.value, b.value)      # (model kernels, coder kernels) of this call
    sizes = np.ctypeslib.as_array(sizes).tolist()   # (8192 Python ints by a list comprehension over the ctypes array took 0.6 ms)
    streams = None
    if want_streams:
        host_out = out.view(shards, cap_per_shard).cpu().numpy()
        streams = [host_out[g, : sizes[g]].tobytes() for g in range(shards)]
    return sizes, streams, out


def decode_shards_gpu(coded, cap_per_shard, sizes, n, columns, codebook_handles, shards, draws, want_labels=False, bounds=None):
    """GPU decoder of sharded streams (qvz_hip_decode_shards_bounds). coded: uint8 device tensor with stream g at
    g * cap_per_shard; draws: (columns, ld) draw matrix of the same reads; bounds: the cut (default qvz_shard_bounds).
    Returns (qv plane (columns, ld), labels|None)."""
    import torch
    k = len(codebook_handles)
    arr = (C.c_void_p * k)(*codebook_handles)
    ld = draws.shape[1]
    qv = torch.zeros((columns, ld), dtype=torch.uint8, device=draws.device)
    labels = torch.zeros(ld, dtype=torch.uint8, device=draws.device) if (k > 1 or want_labels) else None
    sz = (C.c_uint64 * shards)(*[int(x) for x in sizes])
    stream = C.c_void_p(torch.cuda.current_stream().cuda_stream)
    hb = (C.c_uint64 * (shards + 1))(*[int(b) for b in bounds]) if bounds is not None else None
    rc = lib().qvz_hip_decode_shards_bounds(C.c_void_p(coded.data_ptr()), cap_per_shard, sz, n, columns, k, arr, shards, hb,
                                            C.c_void_p(draws.data_ptr()), ld, C.c_void_p(qv.data_ptr()),
                                            C.c_void_p(labels.data_ptr()) if labels is not None else None, stream)
    if rc:
        raise QvzHipError(f"qvz_hip_decode_shards failed ({rc}): {lib().qvz_hip_last_error().decode(errors='replace')}")
    return qv, labels


def shard_file_image(columns, shard_reads, codebook_handles, well_state_at_shard, stream):
    """Reference-format .q image of one shard (decodable by the unmodified reference decoder)."""
    k = len(codebook_handles)
    arr = (C.c_void_p * k)(*codebook_handles)
    ws = np.ascontiguousarray(well_state_at_shard, dtype=np.uint32)
    wp = ws.ctypes.data_as(C.POINTER(C.c_uint32))
    n = lib().qvz_shard_file_image(None, k, columns, shard_reads, arr, wp, None, len(stream))
    buf = np.zeros(n, dtype=np.uint8)
    sbuf = np.frombuffer(stream, dtype=np.uint8) if len(stream) else np.zeros(1, dtype=np.uint8)
    lib().qvz_shard_file_image(buf.ctypes.data_as(C.c_void_p), k, columns, shard_reads, arr, wp,
                               sbuf.ctypes.data_as(C.c_void_p), len(stream))
    return buf.tobytes()


class Coder:
    """One arithmetic-coded stream (include/qvz_coder.h). handles: qvz_codebook_t* of each cluster."""

    def __init__(self, columns, codebook_handles):
        k = len(codebook_handles)
        arr = (C.c_void_p * k)(*codebook_handles)
        self.h = lib().qvz_coder_create(k, columns, arr)
        if not self.h:
            raise MemoryError("qvz_coder_create")
        self.columns = columns

    def encode(self, feed, labels=None):
        feed = np.ascontiguousarray(feed, dtype=np.uint16)
        n = feed.shape[0]
        lp = None
        if labels is not None:
            labels = np.ascontiguousarray(labels, dtype=np.uint8)
            lp = labels.ctypes.data_as(C.c_void_p)
        if lib().qvz_coder_encode(self.h, lp, feed.ctypes.data_as(C.c_void_p), n):
            raise QvzHipError("coder feed addresses an undefined quantizer or state")

    def encode_records(self, rec, reads, pitch):
        """rec: uint64 array (.., 2) of records in the single-stream layout of qvz_hip_model_session_records."""
        rec = np.ascontiguousarray(rec, dtype=np.uint64)
        if lib().qvz_coder_encode_records(self.h, rec.ctypes.data_as(C.c_void_p), reads, pitch):
            raise QvzHipError("qvz_coder_encode_records failed")

    def finish(self):
        lib().qvz_coder_finish(self.h)

    def take(self):
        sz = C.c_uint64(0)
        p = lib().qvz_coder_data(self.h, C.byref(sz))
        data = C.string_at(p, sz.value) if sz.value else b""
        lib().qvz_coder_drop(self.h, sz.value)
        return data

    def total_bytes(self):
        return int(lib().qvz_coder_bytes(self.h))

    def __del__(self):
        try:
            lib().qvz_coder_destroy(self.h)
        except Exception:
            pass

"""ctypes binding of oracle/libqvzoracle.so (the CPU restatement). TEST INFRASTRUCTURE ONLY.

Nothing under qvz_amd/ imports this module; tests, smoke() and bench.py's cpu_baseline leg do.
"""
import ctypes as C
import os
import subprocess

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
ORACLE_DIR = os.path.join(ROOT, "oracle")
LIB_PATH = os.path.join(ORACLE_DIR, "libqvzoracle.so")
A = 72

MODE_RATIO, MODE_FIXED = 0, 1
DIST_MANHATTAN, DIST_MSE, DIST_LORENTZ = 1, 2, 3
DIST_BY_LETTER = {"M": DIST_MSE, "L": DIST_LORENTZ, "A": DIST_MANHATTAN}


class _Codebook(C.Structure):
    _fields_ = [
        ("C", C.c_uint32),
        ("nctx", C.POINTER(C.c_uint8)),
        ("ctx_sym", C.POINTER(C.c_uint8)),
        ("ctx_idx", C.POINTER(C.c_uint8)),
        ("qratio", C.POINTER(C.c_uint8)),
        ("ratio", C.POINTER(C.c_double)),
        ("q", C.POINTER(C.c_uint8)),
        ("nstates", C.POINTER(C.c_uint8)),
        ("recon", C.POINTER(C.c_uint8)),
        ("out_idx", C.POINTER(C.c_uint8)),
        ("mse", C.POINTER(C.c_double)),
    ]


_lib = None


def build():
    subprocess.run(["make", "-s", "-C", ORACLE_DIR, "libqvzoracle.so"], check=True)


def lib():
    global _lib
    if _lib is None:
        if not os.path.exists(LIB_PATH):
            build()
        L = C.CDLL(LIB_PATH)
        u8p, u32p, u64p, f64p = (C.POINTER(C.c_uint8), C.POINTER(C.c_uint32), C.POINTER(C.c_uint64),
                                 C.POINTER(C.c_double))
        L.orc_codebook_alloc.restype = C.POINTER(_Codebook)
        L.orc_codebook_alloc.argtypes = [C.c_uint32]
        L.orc_codebook_free.argtypes = [C.POINTER(_Codebook)]
        L.orc_distortion.argtypes = [C.c_int, f64p]
        L.orc_pick_centroids.argtypes = [C.c_uint64, C.c_uint32, C.c_int, u64p]
        L.orc_kmeans.restype = C.c_int
        L.orc_kmeans.argtypes = [u8p, C.c_uint64, C.c_uint32, C.c_uint32, C.c_double, u64p, u8p, u8p, u32p]
        L.orc_stats.argtypes = [u8p, u8p, C.c_uint64, C.c_uint32, C.c_uint32, u32p]
        L.orc_assign.argtypes = [u8p, C.c_uint64, C.c_uint32, C.c_uint32, u8p, u8p, u32p]
        L.orc_assign.restype = None
        L.orc_marginals.argtypes = [u32p, C.c_uint32, f64p]
        L.orc_lloyd_max.restype = C.c_double
        L.orc_lloyd_max.argtypes = [f64p, f64p, C.c_uint32, u8p, u8p, u8p, f64p]
        L.orc_generate_codebook.argtypes = [u32p, C.c_uint32, f64p, C.c_int, C.c_double, C.POINTER(_Codebook)]
        L.orc_write_header.restype = C.c_size_t
        L.orc_write_header.argtypes = [u8p, C.c_uint32, C.c_uint32, C.c_uint64]
        L.orc_write_codebook.restype = C.c_size_t
        L.orc_write_codebook.argtypes = [u8p, C.POINTER(_Codebook)]
        L.orc_encode_stream.restype = C.c_uint64
        L.orc_encode_stream.argtypes = [u8p, u8p, C.c_uint64, C.c_uint32, C.c_uint32,
                                        C.POINTER(C.POINTER(_Codebook)), f64p, u32p, u8p, u8p, C.c_uint64,
                                        f64p, u8p, u8p]
        L.orc_encode.restype = C.c_uint64
        L.orc_encode.argtypes = [u8p, C.c_uint64, C.c_uint32, C.c_uint32, C.c_int, C.c_double, C.c_int,
                                 C.c_double, u32p, u8p, C.c_uint64, u8p, f64p]
        L.orc_decode.restype = C.c_uint64
        L.orc_decode.argtypes = [u8p, C.c_uint64, u8p, C.c_uint64]
        L.orc_qgen1_rows.argtypes = [C.c_uint64, C.c_uint64, C.c_uint64, C.c_uint32, u8p]
        L.orc_well_states_at.argtypes = [u32p, u64p, C.c_uint64, u32p]
        L.orc_well_states_at.restype = None
        _lib = L
    return _lib


def _p(a, ty):
    return a.ctypes.data_as(C.POINTER(ty))


WELL_DEBUG_STATE = np.full(32, 0x55555555, dtype=np.uint32)  # reference -DDEBUG seed, qv_stream.c:79-83


def well_states_at(words, state0=WELL_DEBUG_STATE):
    """WELL-1024a states (n = 0 form, as a file stores them) after words[i] generator calls from state0, by STEPPING the
    oracle's generator (well.c:8-24 restated): the reference-derived WELL state of a shard in the middle of a file.
    Any order of words; ~1.5 ns per word of the largest."""
    words = np.asarray(words, dtype=np.uint64)
    order = np.argsort(words, kind="stable")
    srt = np.ascontiguousarray(words[order])
    out = np.zeros((len(srt), 32), dtype=np.uint32)
    s0 = np.ascontiguousarray(state0, dtype=np.uint32)
    lib().orc_well_states_at(_p(s0, C.c_uint32), _p(srt, C.c_uint64), len(srt), _p(out, C.c_uint32))
    res = np.zeros_like(out)
    res[order] = out
    return res


def qgen1_rows(seed, n, c, first=0):
    """QGEN-1 reads [first, first+n) as an (n, c+1) uint8 text-layout array."""
    rows = np.zeros((n, c + 1), dtype=np.uint8)
    lib().orc_qgen1_rows(seed, first, n, c, _p(rows, C.c_uint8))
    return rows


def distortion(kind):
    d = np.zeros(A * A, dtype=np.float64)
    assert lib().orc_distortion(DIST_BY_LETTER.get(kind, kind), _p(d, C.c_double)) == 0
    return d


def rows_from_text(buf, c):
    """Text file bytes -> (n, c+1) uint8 view (reference layout, src/lines.c:44-71)."""
    a = np.frombuffer(buf, dtype=np.uint8)
    n = a.size // (c + 1)
    return np.ascontiguousarray(a[: n * (c + 1)].reshape(n, c + 1))


def columns_of(path_or_bytes):
    data = path_or_bytes if isinstance(path_or_bytes, (bytes, bytearray)) else open(path_or_bytes, "rb").read()
    return data.index(b"\n")


def pick_centroids(n, k, reseed=True):
    out = np.zeros(k, dtype=np.uint64)
    lib().orc_pick_centroids(n, k, 1 if reseed else 0, _p(out, C.c_uint64))
    return out


def kmeans(rows, k, threshold=4.0, centroids=None):
    n, c1 = rows.shape
    c = c1 - 1
    if centroids is None:
        centroids = pick_centroids(n, k)
    labels = np.zeros(n, dtype=np.uint8)
    means = np.zeros((k, c), dtype=np.uint8)
    counts = np.zeros(k, dtype=np.uint32)
    iters = lib().orc_kmeans(_p(rows, C.c_uint8), n, c, k, float(threshold), _p(centroids, C.c_uint64),
                             _p(labels, C.c_uint8), _p(means, C.c_uint8), _p(counts, C.c_uint32))
    return labels, means, counts, iters


def assign(rows, means_raw):
    """One assignment pass (cluster.c:136-187) of the reads against k means given as raw bytes -> (labels, counts)."""
    n, c1 = rows.shape
    means_raw = np.ascontiguousarray(means_raw, dtype=np.uint8)
    k = means_raw.shape[0]
    assert means_raw.shape == (k, c1 - 1)
    labels = np.zeros(n, dtype=np.uint8)
    counts = np.zeros(k, dtype=np.uint32)
    lib().orc_assign(_p(rows, C.c_uint8), n, c1 - 1, k, _p(means_raw, C.c_uint8), _p(labels, C.c_uint8), _p(counts, C.c_uint32))
    return labels, counts


def stats(rows, labels, k, out=None):
    """cnt[k][c][72][72] of the reads; out: counters of earlier pieces of the same file to add to."""
    n, c1 = rows.shape
    c = c1 - 1
    cnt = np.zeros((k, c, A, A), dtype=np.uint32) if out is None else out
    assert cnt.shape == (k, c, A, A) and cnt.dtype == np.uint32 and cnt.flags.c_contiguous
    lib().orc_stats(_p(rows, C.c_uint8), _p(labels, C.c_uint8), n, c, k, _p(cnt, C.c_uint32))
    return cnt


def marginals(cnt_cluster):
    c = cnt_cluster.shape[0]
    m = np.zeros((c, A), dtype=np.float64)
    cc = np.ascontiguousarray(cnt_cluster)
    lib().orc_marginals(_p(cc, C.c_uint32), c, _p(m, C.c_double))
    return m


def lloyd_max(p, d, S):
    p = np.ascontiguousarray(p, dtype=np.float64)
    q = np.zeros(A, np.uint8)
    recon = np.zeros(A, np.uint8)
    oidx = np.zeros(A, np.uint8)
    mse = C.c_double(0)
    H = lib().orc_lloyd_max(_p(p, C.c_double), _p(d, C.c_double), S, _p(q, C.c_uint8), _p(recon, C.c_uint8),
                            _p(oidx, C.c_uint8), C.byref(mse))
    return q, recon[:S].copy(), oidx, mse.value, H


class Codebook:
    """Owns one orc_codebook_t; exposes its tables as numpy views."""

    def __init__(self, c):
        self.c = c
        self.ptr = lib().orc_codebook_alloc(c)
        s = self.ptr.contents
        as_np = np.ctypeslib.as_array
        self.nctx = as_np(s.nctx, (c,))
        self.ctx_sym = as_np(s.ctx_sym, (c, A))
        self.ctx_idx = as_np(s.ctx_idx, (c, A))
        self.qratio = as_np(s.qratio, (c, A))
        self.ratio = as_np(s.ratio, (c, A))
        self.q = as_np(s.q, (c, 2 * A, A))
        self.nstates = as_np(s.nstates, (c, 2 * A))
        self.recon = as_np(s.recon, (c, 2 * A, A))
        self.out_idx = as_np(s.out_idx, (c, 2 * A, A))
        self.mse = as_np(s.mse, (c, 2 * A))

    def __del__(self):
        try:
            lib().orc_codebook_free(self.ptr)
        except Exception:
            pass

    def serialize(self):
        n = lib().orc_write_codebook(None, self.ptr)
        buf = np.zeros(n, dtype=np.uint8)
        lib().orc_write_codebook(_p(buf, C.c_uint8), self.ptr)
        return buf.tobytes()


def generate_codebook(cnt_cluster, d, mode, ratio):
    c = cnt_cluster.shape[0]
    cb = Codebook(c)
    cc = np.ascontiguousarray(cnt_cluster)
    lib().orc_generate_codebook(_p(cc, C.c_uint32), c, _p(d, C.c_double), mode, float(ratio), cb.ptr)
    return cb


def write_header(k, c, n):
    buf = np.zeros(9, dtype=np.uint8)
    lib().orc_write_header(_p(buf, C.c_uint8), k, c, n)
    return buf.tobytes()


def encode_stream(rows, labels, cbs, d, well_state=WELL_DEBUG_STATE, want_u=True, want_feed=False):
    n, c1 = rows.shape
    c = c1 - 1
    k = len(cbs)
    arr = (C.POINTER(_Codebook) * k)(*[cb.ptr for cb in cbs])
    u = np.zeros((n, c1), dtype=np.uint8) if want_u else None
    cap = n * c * 2 + 1024
    stream = np.zeros(cap, dtype=np.uint8)
    dist = C.c_double(0)
    qi = np.zeros((n, c), np.uint8) if want_feed else None
    st = np.zeros((n, c), np.uint8) if want_feed else None
    ws = np.ascontiguousarray(well_state, dtype=np.uint32)
    nb = lib().orc_encode_stream(_p(rows, C.c_uint8), _p(labels, C.c_uint8), n, c, k, arr, _p(d, C.c_double),
                                 _p(ws, C.c_uint32), _p(u, C.c_uint8) if want_u else None,
                                 _p(stream, C.c_uint8), cap, C.byref(dist),
                                 _p(qi, C.c_uint8) if want_feed else None,
                                 _p(st, C.c_uint8) if want_feed else None)
    assert nb != 2 ** 64 - 1
    return dict(stream=stream[:nb].tobytes(), u=u, distortion=dist.value, qi=qi, st=st)


def encode(rows, k=1, mode=MODE_RATIO, ratio=0.5, dist="M", threshold=4.0, well_state=WELL_DEBUG_STATE,
           want_u=True):
    """Whole encode(): returns dict(q=bytes, u=bytes|None, rate, distortion, size, iters)."""
    n, c1 = rows.shape
    c = c1 - 1
    cap = 9 + k * (c * (A + 2 * A * A + 3) + 16) + 128 + n * c * 2 + 1024
    q = np.zeros(cap, dtype=np.uint8)
    u = np.zeros(n * c1, dtype=np.uint8) if want_u else None
    st = np.zeros(4, dtype=np.float64)
    ws = np.ascontiguousarray(well_state, dtype=np.uint32)
    nb = lib().orc_encode(_p(rows, C.c_uint8), n, c, k, mode, float(ratio), DIST_BY_LETTER.get(dist, dist),
                          float(threshold), _p(ws, C.c_uint32), _p(q, C.c_uint8), cap,
                          _p(u, C.c_uint8) if want_u else None, _p(st, C.c_double))
    assert nb != 2 ** 64 - 1, "oracle encode failed"
    return dict(q=q[:nb].tobytes(), u=u.tobytes() if want_u else None, rate=st[0], distortion=st[1],
                size=int(st[2]), iters=int(st[3]))


def decode(qbytes):
    f = np.frombuffer(qbytes, dtype=np.uint8)
    c = int.from_bytes(qbytes[1:5], "big")
    n = int.from_bytes(qbytes[5:9], "big")
    out = np.zeros(n * (c + 1), dtype=np.uint8)
    nb = lib().orc_decode(_p(f, C.c_uint8), f.size, _p(out, C.c_uint8), out.size)
    assert nb != 2 ** 64 - 1, "oracle decode failed"
    return out[:nb].tobytes()

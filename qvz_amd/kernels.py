"""Thin torch-tensor wrappers over the C-ABI (include/qvz_hip.h). torch is used for device memory and stream
handles only; all arithmetic happens in the HIP kernels of qvz_amd/csrc.

Symbol matrices are torch.uint8 tensors of shape (c, ld), column-major in the reference's terms: row s of the
tensor is column s of the reads, ld = qvz_hip_ld(n) (n padded to 256).
"""
import ctypes as C

import numpy as np
import torch

from ._lib import check, lib

A = 72


def _stream():
    return C.c_void_p(torch.cuda.current_stream().cuda_stream)


def _ptr(t):
    return C.c_void_p(t.data_ptr()) if t is not None else C.c_void_p(0)


def ld_for(n):
    return int(lib().qvz_hip_ld(n))


def empty_matrix(n, c, device="cuda"):
    return torch.empty((c, ld_for(n)), dtype=torch.uint8, device=device)


def _dims(sym, n):
    c, ld = sym.shape
    assert sym.dtype == torch.uint8 and sym.is_contiguous() and ld == ld_for(n), (sym.shape, n)
    return c, ld


def qgen1(seed, n, c, first=0, out=None):
    """QGEN-1 reads [first, first+n) generated on device in symbol-matrix layout."""
    sym = out if out is not None else empty_matrix(n, c)
    c_, ld = _dims(sym, n)
    check(lib().qvz_hip_qgen1(seed, first, n, c_, _ptr(sym), ld, _stream()))
    return sym


def transpose_text(text, n, c):
    """text: uint8 device tensor with n*(c+1) bytes in the reference's file layout -> (sym, bad_count)."""
    sym = empty_matrix(n, c, text.device)
    bad = torch.zeros(1, dtype=torch.int32, device=text.device)
    check(lib().qvz_hip_transpose_text(_ptr(text), n, c, _ptr(sym), sym.shape[1], _ptr(bad), _stream()))
    return sym, bad


def untranspose_text(qv, n):
    c, ld = _dims(qv, n)
    text = torch.empty(n * (c + 1), dtype=torch.uint8, device=qv.device)
    check(lib().qvz_hip_untranspose_text(_ptr(qv), n, c, ld, _ptr(text), _stream()))
    return text


def kmeans_assign(sym, n, means, labels=None, counts=None):
    c, ld = _dims(sym, n)
    k = means.shape[0]
    assert means.dtype == torch.uint8 and means.shape == (k, c) and means.is_contiguous()
    if labels is None:
        labels = torch.zeros(ld, dtype=torch.uint8, device=sym.device)
    if counts is None:
        counts = torch.zeros(k, dtype=torch.int32, device=sym.device)
    check(lib().qvz_hip_kmeans_assign(_ptr(sym), n, c, ld, k, _ptr(means), _ptr(labels), _ptr(counts), _stream()))
    return labels, counts


def kmeans_accumulate(sym, n, labels, k, acc=None):
    c, ld = _dims(sym, n)
    if acc is None:
        acc = torch.zeros((k, c), dtype=torch.int64, device=sym.device)
    check(lib().qvz_hip_kmeans_accumulate(_ptr(sym), _ptr(labels), n, c, ld, k, _ptr(acc), _stream()))
    return acc


def kmeans_step(sym, n, means, labels=None, counts=None, acc=None):
    """Assignment and per-cluster column sums of one k-means iteration in one pass (qvz_hip_kmeans_step)."""
    c, ld = _dims(sym, n)
    k = means.shape[0]
    assert means.dtype == torch.uint8 and means.shape == (k, c) and means.is_contiguous()
    if labels is None:
        labels = torch.zeros(ld, dtype=torch.uint8, device=sym.device)
    if counts is None:
        counts = torch.zeros(k, dtype=torch.int32, device=sym.device)
    if acc is None:
        acc = torch.zeros((k, c), dtype=torch.int64, device=sym.device)
    check(lib().qvz_hip_kmeans_step(_ptr(sym), n, c, ld, k, _ptr(means), _ptr(labels), _ptr(counts), _ptr(acc), _stream()))
    return labels, counts, acc


def kmeans_update(acc, counts, means):
    k, c = means.shape
    moved = torch.zeros(k + 1, dtype=torch.float64, device=means.device)   # [0] = max, [1+j] per cluster
    status = torch.zeros(1, dtype=torch.int32, device=means.device)
    check(lib().qvz_hip_kmeans_update(_ptr(acc), _ptr(counts), k, c, _ptr(means), _ptr(moved), _ptr(status),
                                      _stream()))
    return moved, status


def pmf_hist(sym, n, labels=None, k=1, cnt=None):
    c, ld = _dims(sym, n)
    if cnt is None:
        cnt = torch.zeros((k, c, A, A), dtype=torch.int32, device=sym.device)
    check(lib().qvz_hip_pmf_hist(_ptr(sym), _ptr(labels), n, c, ld, k, _ptr(cnt), _stream()))
    return cnt


def _state_arg(state):
    st = np.ascontiguousarray(state, dtype=np.uint32)
    assert st.shape == (32,)
    return st, st.ctypes.data_as(C.POINTER(C.c_uint32))


def well_words(state, first_word, nwords, device="cuda", out=None):
    st, sp = _state_arg(state)
    words = out if out is not None else torch.empty(nwords, dtype=torch.int32, device=device)
    check(lib().qvz_hip_well_words(sp, first_word, nwords, _ptr(words), _stream()))
    return words


def well_draws(state, first_read, n, c, out=None):
    """Draw matrix (c, ld) uint8 of reads [first_read, first_read+n): the layout of the symbol matrix."""
    st, sp = _state_arg(state)
    draws = out if out is not None else empty_matrix(n, c)
    c_, ld = _dims(draws, n)
    check(lib().qvz_hip_well_draws(sp, first_read, n, c_, _ptr(draws), ld, _stream()))
    return draws


def well_jump(state, words):
    st, sp = _state_arg(state)
    out = np.zeros(32, dtype=np.uint32)
    check(lib().qvz_hip_well_jump(sp, words, out.ctypes.data_as(C.POINTER(C.c_uint32))))
    return out


def quantize(sym, n, ct, qs, draws, dist, labels=None, k=1, want=("qv", "qi", "st", "err")):
    """ct: (k, c, 72) int16/uint16-as-int16, qs: (k, c, 144, 72); draws: (c, ld) uint8 draw matrix of the same reads
    (well_draws)."""
    c, ld = _dims(sym, n)
    assert draws.shape == sym.shape and draws.dtype == torch.uint8
    out = quantize_outputs(sym, n, want)
    status = out["status"]
    check(lib().qvz_hip_quantize(_ptr(sym), _ptr(labels), n, c, ld, k, _ptr(ct), _ptr(qs), _ptr(draws), ld,
                                 _ptr(dist), _ptr(out["qv"]), _ptr(out["qi"]), _ptr(out["st"]), ld, _ptr(out["err"]),
                                 _ptr(status), _stream()))
    return out


def quantize_outputs(sym, n, want):
    """The quantizer's output planes (None where not wanted), per-read errors and status word, allocated."""
    c, ld = _dims(sym, n)
    dev = sym.device
    out = {}
    for name in ("qv", "qi", "st"):
        out[name] = torch.empty((c, ld), dtype=torch.uint8, device=dev) if name in want else None
    out["err"] = torch.empty(n, dtype=torch.float64, device=dev) if "err" in want else None
    out["status"] = torch.zeros(1, dtype=torch.int32, device=dev)
    return out


def pack_feed(qi, st, n):
    """Column-major (qi, st) -> read-major int16 feed of shape (n, c): qi | st << 8."""
    c, ld = _dims(qi, n)
    feed = torch.empty((n, c), dtype=torch.int16, device=qi.device)
    check(lib().qvz_hip_pack_feed(_ptr(qi), _ptr(st), n, c, ld, _ptr(feed), _stream()))
    return feed


# ------------------------------------------------------------------------------------------------
# codebooks (include/qvz_codebook.h)
# ------------------------------------------------------------------------------------------------
class _CodebookStruct(C.Structure):
    _fields_ = [("C", C.c_uint32), ("nctx", C.POINTER(C.c_uint8)), ("ctx_sym", C.POINTER(C.c_uint8)),
                ("ctx_idx", C.POINTER(C.c_uint8)), ("qratio", C.POINTER(C.c_uint8)), ("ratio", C.POINTER(C.c_double)),
                ("q", C.POINTER(C.c_uint8)), ("nstates", C.POINTER(C.c_uint8)), ("recon", C.POINTER(C.c_uint8)),
                ("out_idx", C.POINTER(C.c_uint8))]


class Codebook:
    """Owns one qvz_codebook_t (host, flat) and exposes its arrays as numpy views."""

    def __init__(self, c):
        self.c = c
        self.handle = lib().qvz_codebook_alloc(c)
        if not self.handle:
            raise MemoryError("qvz_codebook_alloc")
        s = C.cast(self.handle, C.POINTER(_CodebookStruct)).contents
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

    def __del__(self):
        try:
            lib().qvz_codebook_free(self.handle)
        except Exception:
            pass

    def serialize(self):
        n = lib().qvz_codebook_write(None, self.handle)
        buf = np.zeros(n, dtype=np.uint8)
        lib().qvz_codebook_write(buf.ctypes.data_as(C.c_void_p), self.handle)
        return buf.tobytes()

    def tables(self):
        """(ct, qs) host arrays in the quantizer's device-table layout (include/qvz_hip.h)."""
        ct = np.zeros((self.c, A), dtype=np.uint16)
        qs = np.zeros((self.c, 2 * A, A), dtype=np.uint16)
        lib().qvz_codebook_tables(self.handle, ct.ctypes.data_as(C.c_void_p), qs.ctypes.data_as(C.c_void_p))
        return ct, qs


def distortion_matrix(kind):
    code = {"A": 1, "M": 2, "L": 3}.get(kind, kind)
    d = np.zeros(A * A, dtype=np.float64)
    check(lib().qvz_distortion_matrix(code, d.ctypes.data_as(C.POINTER(C.c_double))))
    return d


def write_header(clusters, columns, lines):
    buf = np.zeros(9, dtype=np.uint8)
    lib().qvz_codebook_write_header(buf.ctypes.data_as(C.c_void_p), clusters, columns, lines)
    return buf.tobytes()


def generate_codebook(cnt_cluster, dist, mode, ratio, guard=0.0):
    """cnt_cluster: (c, 72, 72) int32 device tensor of one cluster. Returns (Codebook, deferred_decisions)."""
    c = cnt_cluster.shape[0]
    assert cnt_cluster.is_contiguous() and cnt_cluster.dtype == torch.int32
    cb = Codebook(c)
    d = np.ascontiguousarray(dist, dtype=np.float64)
    check(lib().qvz_hip_generate_codebook(_ptr(cnt_cluster), c, d.ctypes.data_as(C.POINTER(C.c_double)), mode,
                                          float(ratio), float(guard), cb.handle, _stream()))
    return cb, int(lib().qvz_hip_codebook_deferred_decisions())


def upload_tables(cbs, device="cuda"):
    """Stack the quantizer tables of k codebooks on the device: ct (k, c, 72), qs (k, c, 144, 72) as int16."""
    cts, qss = zip(*[cb.tables() for cb in cbs])
    ct = torch.from_numpy(np.stack(cts).view(np.int16)).to(device)
    qs = torch.from_numpy(np.stack(qss).view(np.int16)).to(device)
    return ct, qs

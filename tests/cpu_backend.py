"""A compute backend for qvz_amd.pipeline.encode_pass made of the CPU oracle (TEST INFRASTRUCTURE). It lets the product's
own read-sharding schedule - who owns which reads, what is all-reduced and when, which WELL offset a shard starts from -
run on CPU-only `gloo` ranks; the arithmetic behind every method is oracle/qvz_oracle.c, never product code."""
import ctypes as C

import numpy as np
import torch

import oracle_lib as O


def advance_well(state, words):
    """Steps the reference recurrence `words` times and returns the state rotated to n = 0 (what qvz_hip_well_jump gives)."""

    class W(C.Structure):
        _fields_ = [("state", C.c_uint32 * 32), ("n", C.c_uint32), ("bit_output", C.c_uint32), ("bits_left", C.c_uint32)]

    lib = O.lib()
    lib.orc_well_word.restype = C.c_uint32
    w = W()
    for i in range(32):
        w.state[i] = int(state[i])
    for _ in range(words):
        lib.orc_well_word(C.byref(w))
    return np.array([w.state[(w.n + k) & 31] for k in range(32)], dtype=np.uint32)


def sym_from_rows(rows):
    """(n, c+1) text rows -> the product's symbol matrix layout (c, ld) uint8, ld = n rounded up to 256"""
    n, c = rows.shape[0], rows.shape[1] - 1
    ld = (n + 255) & ~255
    sym = torch.zeros((c, ld), dtype=torch.uint8)
    sym[:, :n] = torch.from_numpy(np.ascontiguousarray((rows[:, :c] - 33).T))
    return sym


class OracleBackend:
    device = torch.device("cpu")

    def __init__(self):
        self.well = None

    def zeros(self, shape, dtype):
        return torch.zeros(shape, dtype=dtype)

    def sync(self):
        pass

    def _rows(self, sym, n):
        c = sym.shape[0]
        rows = np.full((n, c + 1), 10, dtype=np.uint8)
        rows[:, :c] = sym[:, :n].numpy().T + 33
        return rows

    def read_rows(self, sym, idx):
        return sym[:, torch.as_tensor(idx, dtype=torch.long)].t().contiguous()

    def kmeans_step(self, sym, n, means, labels, counts, acc):
        self.kmeans_assign(sym, n, means, labels, counts)
        self.kmeans_accumulate(sym, n, labels, means.shape[0], acc)

    def kmeans_assign(self, sym, n, means, labels, counts):
        x = sym[:, :n].numpy().T.astype(np.int64)
        d = ((x[:, None, :] - means.numpy()[None].astype(np.int64)) ** 2).sum(axis=2)
        lab = d.argmin(axis=1).astype(np.uint8)                                  # lowest id wins ties, cluster.c:155-163
        labels[:n] = torch.from_numpy(lab)
        counts += torch.from_numpy(np.bincount(lab, minlength=means.shape[0]).astype(np.int32))

    def kmeans_accumulate(self, sym, n, labels, k, acc):
        x = sym[:, :n].numpy().T.astype(np.int64)
        lab = labels[:n].numpy()
        for j in range(k):
            acc[j] += torch.from_numpy(x[lab == j].sum(axis=0))

    def kmeans_update(self, acc, counts, means):                                  # recalculate_means, cluster.c:104-128
        k = means.shape[0]
        moved = torch.zeros(k + 1, dtype=torch.float64)
        status = torch.zeros(1, dtype=torch.int32)
        for j in range(k):
            if int(counts[j]) == 0:
                status[0] = 1
                continue
            new = (acc[j] // int(counts[j])).to(torch.uint8)
            moved[1 + j] = float(((new.to(torch.int64) - means[j].to(torch.int64)) ** 2).sum())
            means[j] = new
        moved[0] = moved[1:].max()
        return moved, status

    def pmf_hist(self, sym, n, labels, k, cnt):
        lab = labels[:n].numpy() if labels is not None else np.zeros(n, np.uint8)
        cnt += torch.from_numpy(O.stats(self._rows(sym, n), np.ascontiguousarray(lab), k).astype(np.int32))

    def start_draws(self, well_state, first_read, n, c):
        self.well = advance_well(well_state, first_read * c // 4)                 # this shard's generator, n = 0
        return None

    def codebooks(self, cnt, dmat, mode, ratio):
        c = cnt.numpy().astype(np.uint32)
        return [O.generate_codebook(c[j], dmat, mode, ratio) for j in range(c.shape[0])], 0

    def quantize(self, sym, n, cbs, draws, dmat, labels, k, want):
        lab = labels[:n].numpy() if labels is not None else np.zeros(n, np.uint8)
        self._enc = O.encode_stream(self._rows(sym, n), np.ascontiguousarray(lab), cbs, dmat, well_state=self.well, want_feed=True)
        c = sym.shape[0]
        u = self._enc["u"][:, :c] - 33
        err = dmat[self._rows(sym, n)[:, :c].astype(np.int64) - 33 + 72 * u.astype(np.int64)].sum(axis=1)
        plane = lambda a: torch.from_numpy(np.ascontiguousarray(a.T))
        return dict(qv=plane(u), qi=plane(self._enc["qi"]), st=plane(self._enc["st"]), err=torch.from_numpy(err),
                    status=torch.zeros(1, dtype=torch.int32))

    def code(self, out, labels, n, c, cbs, shards, threads, coder, keep, work_bytes):
        assert shards == 1, "the oracle backend codes one stream per rank"
        s = self._enc["stream"]
        return [len(s)], [s], None, None, None

"""Host-side workers for full-size checks by the CPU oracle (TEST INFRASTRUCTURE): a file too long for one core - 100M reads -
is regenerated, re-labelled and re-counted in contiguous pieces by child processes that never touch the GPU (this file run as a
script: it imports numpy and the oracle only). Reference: src/cluster.c:136-187 (assignment), src/codebook.c:185-205 (counters)."""
import os
import subprocess
import sys
import tempfile

import numpy as np

CHUNK = 1 << 17


def _piece(seed, first, n, c, k, means_path, labels_path, out_path):
    import oracle_lib as O
    cnt = np.zeros((k, c, 72, 72), dtype=np.uint32)
    means_raw = np.load(means_path) if k > 1 else None
    gpu_labels = np.memmap(labels_path, dtype=np.uint8, mode="r") if k > 1 else None
    mismatch, counts = -1, np.zeros(k, dtype=np.uint64)
    for a in range(first, first + n, CHUNK):
        m = min(CHUNK, first + n - a)
        rows = O.qgen1_rows(seed, m, c, first=a)
        if k > 1:
            lab, cn = O.assign(rows, means_raw)
            counts += cn
            if mismatch < 0 and not np.array_equal(lab, gpu_labels[a:a + m]):
                mismatch = a + int(np.flatnonzero(lab != gpu_labels[a:a + m])[0])
        else:
            lab = np.zeros(m, dtype=np.uint8)
            counts[0] += m
        O.stats(rows, lab, k, out=cnt)
    np.savez(out_path, cnt=cnt, mismatch=np.int64(mismatch), counts=counts)


def recount(seed, n, c, k, means_raw=None, gpu_labels=None, workers=None, tmpdir="/dev/shm"):
    """QGEN-1 reads [0, n): the oracle assigns every read to the nearest of `means_raw` (k x c raw bytes; k > 1) and counts its
    symbol pairs under ITS label. -> (cnt[k][c][72][72] u32, number of the first read whose label differs from gpu_labels or -1,
    reads per cluster). One child process per piece (whole multiples of CHUNK reads), counters summed over the pieces (u32:
    n < 2^32)."""
    workers = workers or max(1, min(16, (os.cpu_count() or 2) - 1))
    with tempfile.TemporaryDirectory(prefix="qvz_recount_", dir=tmpdir if os.path.isdir(tmpdir) else None) as td:
        means_path, labels_path = os.path.join(td, "means.npy"), os.path.join(td, "labels.u8")
        if k > 1:
            np.save(means_path, np.ascontiguousarray(means_raw, dtype=np.uint8))
            np.ascontiguousarray(gpu_labels[:n], dtype=np.uint8).tofile(labels_path)
        per = -(-(-(-n // workers)) // CHUNK) * CHUNK
        procs = []
        for i, a in enumerate(range(0, n, per)):
            out = os.path.join(td, f"piece{i}.npz")
            args = [sys.executable, os.path.abspath(__file__)] + [str(x) for x in (seed, a, min(per, n - a), c, k)] + [means_path, labels_path, out]
            procs.append((subprocess.Popen(args), out))
        cnt = np.zeros((k, c, 72, 72), dtype=np.uint32)
        counts = np.zeros(k, dtype=np.uint64)
        mismatch = -1
        for p, out in procs:
            assert p.wait(timeout=900) == 0, "an oracle worker failed"
            z = np.load(out)
            cnt += z["cnt"]
            counts += z["counts"]
            mm = int(z["mismatch"])
            if mm >= 0 and (mismatch < 0 or mm < mismatch):
                mismatch = mm
    return cnt, mismatch, counts


if __name__ == "__main__":
    sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
    _piece(*[int(x) for x in sys.argv[1:6]], *sys.argv[6:9])

"""world_size-2 `gloo` tests (CPU) of the read-sharding protocol used by qvz_amd/pipeline.py and bench.py --gpus N:
what is reduced (k-means accumulators, PMF counters, distortion), where shards may start (multiples of 4 reads = WELL
word boundaries) and which WELL offset each shard uses. Compute is done by the CPU oracle here (no GPU in this
environment); the same protocol runs over RCCL with the HIP kernels on the GPU box."""
import os
import socket

import numpy as np
import pytest

import oracle_lib as O

torch = pytest.importorskip("torch")
import torch.distributed as dist  # noqa: E402
import torch.multiprocessing as mp  # noqa: E402


def free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def advance_well(state, words):
    """CPU stand-in for qvz_hip_well_jump: step the reference recurrence and store the state rotated to n = 0."""
    import ctypes as C

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


N, COLS, K, SEED = 3000, 24, 2, 13


def _rank_main(rank, world, port, out):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from qvz_amd import pipeline as P
    first, last = P.rank_read_range(N, world, rank)
    rows = O.qgen1_rows(SEED, last - first, COLS, first=first)             # this rank's reads only
    n = last - first

    # initial centroids: every rank derives the same global read indices; the owner's bytes are made global by a SUM
    lines = P.reference_centroid_lines(N, K)
    means = torch.zeros((K, COLS), dtype=torch.int64)
    for j, ln in enumerate(lines):
        if first <= ln < last:
            means[j] = torch.from_numpy(rows[ln - first, :COLS].astype(np.int64))
    dist.all_reduce(means)
    means = means.numpy().astype(np.uint8)

    # k-means: local assignment + accumulation, all_reduce(SUM) of acc / counts, identical integer update everywhere
    iters = 0
    while True:
        d = ((rows[:, None, :COLS].astype(np.int64) - means[None].astype(np.int64)) ** 2).sum(axis=2)
        labels = d.argmin(axis=1).astype(np.uint8)                         # lowest id wins ties, cluster.c:155-163
        acc = torch.zeros((K, COLS), dtype=torch.int64)
        cnt = torch.zeros(K, dtype=torch.int64)
        for j in range(K):
            acc[j] = torch.from_numpy(rows[labels == j, :COLS].astype(np.int64).sum(axis=0))
            cnt[j] = int((labels == j).sum())
        dist.all_reduce(acc)
        dist.all_reduce(cnt)
        new = (acc.numpy() // cnt.numpy()[:, None]).astype(np.uint8)
        moved = ((new.astype(np.int64) - means.astype(np.int64)) ** 2).sum(axis=1).max()
        means = new
        iters += 1
        if not moved > 4:
            break

    # PMF counters: local counts, all_reduce(SUM) (u32 sums are exact for N < 2^32)
    counts = torch.from_numpy(O.stats(rows, labels, K).astype(np.int64))
    dist.all_reduce(counts)
    counts = counts.numpy().astype(np.uint32)

    # codebooks from the reduced counters: identical on every rank
    dmat = O.distortion("M")
    cbs = [O.generate_codebook(counts[j], dmat, O.MODE_RATIO, 0.5) for j in range(K)]
    blob = b"".join(cb.serialize() for cb in cbs)

    # quantize this shard with the generator advanced to its first draw (shards start on word boundaries)
    assert (first * COLS) % 4 == 0
    state = advance_well(O.WELL_DEBUG_STATE, first * COLS // 4)
    enc = O.encode_stream(rows, labels, cbs, dmat, well_state=state)
    err = torch.tensor([enc["distortion"] * n], dtype=torch.float64)
    dist.all_reduce(err)
    out.put((rank, first, last, iters, means.tobytes(), labels.tobytes(), counts.tobytes(), blob, enc["u"].tobytes(),
             float(err.item()) / N, enc["stream"], state.tobytes()))
    dist.destroy_process_group()


def test_two_rank_gloo_protocol_equals_single_process():
    world = 2
    ctx = mp.get_context("spawn")
    out = ctx.Queue()
    port = free_port()
    procs = [ctx.Process(target=_rank_main, args=(r, world, port, out)) for r in range(world)]
    for p in procs:
        p.start()
    got = sorted(out.get(timeout=120) for _ in range(world))
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0

    rows = O.qgen1_rows(SEED, N, COLS)
    labels, means, counts_k, iters = O.kmeans(rows, K)
    cnt = O.stats(rows, labels, K)
    dmat = O.distortion("M")
    cbs = [O.generate_codebook(cnt[j], dmat, O.MODE_RATIO, 0.5) for j in range(K)]
    single = O.encode_stream(rows, labels, cbs, dmat)
    blob = b"".join(cb.serialize() for cb in cbs)

    assert got[0][1] == 0 and got[-1][2] == N and got[0][2] == got[1][1]
    assert got[0][2] % 4 == 0, "shards must start on WELL word boundaries"
    u = b""
    for rank, first, last, it, m, lab, c, b, ushard, distortion, stream, state in got:
        assert it == iters
        assert m == means.tobytes()
        assert lab == labels[first:last].tobytes()
        assert c == cnt.tobytes()
        assert b == blob
        assert abs(distortion - single["distortion"]) < 1e-9 * max(1.0, single["distortion"])
        u += ushard
        # each shard is a reference-format file of its own: header + codebooks + rotated WELL state + stream
        image = O.write_header(K, COLS, last - first) + blob + state + stream
        assert O.decode(image) == ushard
    assert u == single["u"].tobytes(), "concatenated shard outputs must equal the single-stream -u output"


def test_rank_read_ranges_are_word_aligned_and_cover_everything():
    from qvz_amd import pipeline as P
    for n_total in (1, 5, 4096, 10 ** 6 + 3, 10 ** 9):
        for world in (1, 2, 3, 4, 8):
            prev = 0
            for r in range(world):
                a, b = P.rank_read_range(n_total, world, r)
                assert a == prev and b >= a
                assert a % 4 == 0 or a == b      # only an empty trailing shard may start off a word boundary
                prev = b
            assert prev == n_total

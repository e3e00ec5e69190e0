"""qvz_resident_* (include/qvz_encode.h): the whole hot path in C over a symbol matrix that is already in HBM - the entry bench.py
times. Against the reference's fixtures and the CPU oracle: labels, codebook bytes, quantized values, distortion, and every
stream as a reference-format file of its reads; a handle re-used for other shapes and flags; eight parts of one file as threads
with the all-reduce handed in as a C callback."""
import ctypes as C

import numpy as np
import pytest

import golden_util as G
import oracle_lib as O

pytestmark = pytest.mark.gpu
torch = pytest.importorskip("torch")


def upload(rows):
    from qvz_amd import kernels as K
    n, c = rows.shape[0], rows.shape[1] - 1
    sym = K.empty_matrix(n, c)
    sym.zero_()
    sym[:, :n] = torch.from_numpy(np.ascontiguousarray((rows[:, :c] - 33).T)).cuda()
    return sym


def check_against_oracle(h, rows, opts, r, p, shards, first=0, n_total=None, whole=None):
    """one part's products against the single-process oracle of the WHOLE file (whole = its rows when `rows` is a part)"""
    from qvz_amd import host
    whole = rows if whole is None else whole
    n, c = rows.shape[0], rows.shape[1] - 1
    k = opts["k"]
    labels, means, _, iters = O.kmeans(whole, k)
    cnt = O.stats(whole, labels, k)
    dmat = O.distortion(opts["dist"])
    cbs = [O.generate_codebook(cnt[j], dmat, opts["mode"], opts["ratio"]) for j in range(k)]
    single = O.encode_stream(whole, labels, cbs, dmat)
    assert r.kmeans_iterations == iters and (r.lines, r.columns) == (n, c)
    assert host.Resident.codebook_bytes(p) == [cb.serialize() for cb in cbs]
    ld = (n + 255) & ~255
    assert np.array_equal(host.device_view(p.d_labels, ld).cpu().numpy()[:n], labels[first:first + n])
    qv = host.device_view(p.d_qv, c * ld).cpu().numpy().reshape(c, ld)[:, :n].T + 33
    assert np.array_equal(qv, single["u"][first:first + n, :c])
    assert abs(r.distortion - single["distortion"]) < 1e-9 * max(1.0, single["distortion"])
    bounds = host.shard_bounds(n, shards)
    assert not p.d_streams and p.d_coded and p.cap_per_stream >= max(int(p.h_sizes[g]) for g in range(shards))   # packed on request only
    streams = h.streams(p)
    assert r.stream_bytes == sum(len(x) for x in streams) == p.stream_bytes
    for g in range(shards):
        a, b = first + bounds[g], first + bounds[g + 1]
        state = O.well_states_at([a * c // 4])[0]
        want = O.encode_stream(whole[a:b], labels[a:b], cbs, dmat, well_state=state, want_u=False)["stream"]
        assert streams[g] == want, f"stream {g} differs from the oracle's"
    return single


@pytest.mark.parametrize("name,shards", [("t2k_c3_f05_M", 5), ("cfg1_f05_L", 16), ("t2k_r10_M", 1), ("twolevel_c2_f05", 2), ("uniform72_r20_A", 3)])
def test_resident_encode_equals_the_oracle_and_the_reference(name, shards):
    from qvz_amd import host
    e = G.CASES[name]
    rows = G.case_rows(name)
    opts = G.case_opts(name)
    res = host.Resident(0)
    hmode = host.MODE_RATIO if opts["mode"] == O.MODE_RATIO else host.MODE_FIXED
    r, p = res.encode(upload(rows), rows.shape[0], shards, clusters=opts["k"], mode=hmode, ratio=opts["ratio"], dist=opts["dist"], want_qv=True)
    single = check_against_oracle(res, rows, opts, r, p, shards)
    assert G.md5(single["u"].tobytes()) == e["u_md5"] and f"{r.distortion:.4f}" == f"{e['distortion']:.4f}"
    if shards == 1:                                               # one stream: the reference's own stream, byte for byte
        assert res.streams(p)[0] == G.gz(name + ".q.gz")[-e["size"]:]
    assert p.pmf_kernel_ms > 0 and p.quantize_kernel_ms > 0 and p.model_kernels_ms > 0 and p.coder_kernels_ms > 0
    res.close()


def test_one_handle_many_encodes():
    """The handle keeps buffers, never results: encodes of different shapes and flags on one handle, in both orders, equal the oracle."""
    from qvz_amd import host
    res = host.Resident(0)
    for name, shards in [("cfg1_f05_L", 8), ("t2k_c3_f05_M", 3), ("cfg1_f05_L", 8), ("wide_c151", 2)]:
        rows, opts = G.case_rows(name), G.case_opts(name)
        hmode = host.MODE_RATIO if opts["mode"] == O.MODE_RATIO else host.MODE_FIXED
        r, p = res.encode(upload(rows), rows.shape[0], shards, clusters=opts["k"], mode=hmode, ratio=opts["ratio"], dist=opts["dist"], want_qv=True)
        check_against_oracle(res, rows, opts, r, p, shards)
    # without the quantized values: nothing written, the streams unchanged
    r2, p2 = res.encode(upload(rows), rows.shape[0], shards, clusters=opts["k"], mode=hmode, ratio=opts["ratio"], dist=opts["dist"])
    assert not p2.d_qv and r2.stream_bytes == r.stream_bytes
    res.close()


def test_bad_arguments_are_refused():
    from qvz_amd import host
    from qvz_amd._lib import QvzHipError
    rows = G.case_rows("t2k_r10_M")
    sym = upload(rows)
    res = host.Resident(0)
    with pytest.raises(QvzHipError, match="all-reduce"):
        res.encode(sym, rows.shape[0], 2, first_read=4, n_total=rows.shape[0] + 4)   # a part without its collective
    with pytest.raises(QvzHipError):
        res.encode(sym, rows.shape[0], 0)
    with pytest.raises(QvzHipError):
        res.encode(sym, rows.shape[0], rows.shape[0])                        # more streams than groups of four reads
    r, p = res.encode(sym, rows.shape[0], 2)                                 # the handle is still good
    assert r.stream_bytes > 0
    res.close()


def test_eight_parts_through_the_c_entry_with_a_callback_all_reduce():
    """Eight parts of one file (threads of this process, pipeline.LocalWorld), each a qvz_resident_encode of its reads with the
    all-reduce handed in as a C callback - the entry bench.py --gpus 8 drives, one rank per GPU over RCCL there. -c 3: the centroid
    reads lie on three different parts."""
    import threading
    from qvz_amd import host, kernels as K, pipeline as P
    n_total, cols, k, seed, world, shards = 1_000_000, 50, 3, 23, 8, 3
    rows = O.qgen1_rows(seed, n_total, cols)
    lines = [int(x) for x in O.pick_centroids(n_total, k)]
    assert len({next(r for r in range(world) if P.rank_read_range(n_total, world, r)[0] <= ln < P.rank_read_range(n_total, world, r)[1]) for ln in lines}) == 3
    lw = P.LocalWorld(world)
    handles = [host.Resident(0) for _ in range(world)]
    out, errors = [None] * world, []
    lock = threading.Lock()

    class Serial:                                                 # one part at a time inside the library between collectives
        def __init__(self, g):
            self.g = g

        def all_reduce(self, t):
            lock.release()
            try:
                self.g.all_reduce(t)
            finally:
                lock.acquire()

    def part(rank):
        try:
            first, last = P.rank_read_range(n_total, world, rank)
            sym = K.qgen1(seed, last - first, cols, first=first)
            with lock:
                out[rank] = (first, last) + handles[rank].encode(sym, last - first, shards, clusters=k, mode=host.MODE_FIXED, ratio=1.0, dist="M",
                                                                 want_qv=True, first_read=first, n_total=n_total, group=Serial(lw.group(rank)))
        except BaseException as exc:  # noqa: BLE001
            errors.append((rank, repr(exc)))
            lw._barrier.abort()

    threads = [threading.Thread(target=part, args=(r,)) for r in range(world)]
    for t in threads:
        t.start()
    for t in threads:
        t.join(timeout=600)
    assert not errors, errors
    opts = dict(k=k, mode=O.MODE_FIXED, ratio=1.0, dist="M")
    for rank, (first, last, r, p) in enumerate(out):
        check_against_oracle(handles[rank], rows[first:last], opts, r, p, shards, first=first, n_total=n_total, whole=rows)
    for h in handles:
        h.close()

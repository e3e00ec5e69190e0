"""GPU tests of the whole device pipeline (qvz_amd/pipeline.py) incl. the sharded coder: every shard must be a file
the reference decoder accepts, and the decodes must concatenate to the single-stream -u output."""
import numpy as np
import pytest

import golden_util as G
import oracle_lib as O

pytestmark = pytest.mark.gpu
torch = pytest.importorskip("torch")


def setup_case(name):
    from qvz_amd import kernels as K, pipeline as P
    from test_gpu_kernels import upload_sym
    rows = G.case_rows(name)
    opts = G.case_opts(name)
    n, c = rows.shape[0], rows.shape[1] - 1
    sym = upload_sym(K, rows)
    lines = P.reference_centroid_lines(n, opts["k"])
    means0 = torch.from_numpy(np.stack([rows[i, :c] - 33 for i in lines]).astype(np.uint8)).cuda()
    return rows, opts, n, c, sym, means0


@pytest.mark.parametrize("name", ["t2k_r10_M", "t2k_c3_f05_M", "twolevel_c2_f05", "cfg1_f05_L", "rescale_140k_c4",
                                  "rescale_c2_100k_c3"])
def test_single_shard_equals_reference_stream(name):
    from qvz_amd import kernels as K, pipeline as P
    rows, opts, n, c, sym, means0 = setup_case(name)
    r = P.encode_pass(sym, n, 0, n, means0, clusters=opts["k"], mode=opts["mode"], ratio=opts["ratio"], dist=opts["dist"],
                      shards=1, threads=1, keep=True)
    e = G.CASES[name]
    assert r["total_stream_bytes"] == e["size"]
    text = K.untranspose_text(r["qv"], n).cpu().numpy().tobytes()
    assert G.md5(text) == e["u_md5"]
    blob = K.write_header(opts["k"], c, n) + b"".join(cb.serialize() for cb in r["codebooks"]) + O.WELL_DEBUG_STATE.tobytes()
    assert G.md5(blob + r["streams"][0]) == e["q_md5"]
    assert abs(r["distortion"] - e["distortion"]) < 1e-4


@pytest.mark.parametrize("k,dist,mode,ratio", [(5, "M", "f", 0.6), (4, "L", "r", 1.5)])
def test_more_clusters_than_the_lds_quantizer_holds_equals_the_oracle(k, dist, mode, ratio):
    """Four and five clusters (the general quantizer kernel, model tables of several clusters per column, a cluster-id
    symbol per read): whole file image and -u text equal the oracle's, single stream and three shards through both coders."""
    from qvz_amd import host, kernels as K, pipeline as P
    from test_gpu_kernels import upload_sym
    n, c = 6000, 21
    rows = O.qgen1_rows(31 + k, n, c)
    omode = O.MODE_RATIO if mode == "f" else O.MODE_FIXED
    ref = O.encode(rows, k=k, mode=omode, ratio=ratio, dist=dist)
    sym = upload_sym(K, rows)
    lines = P.reference_centroid_lines(n, k)
    means0 = torch.from_numpy(np.stack([rows[i, :c] - 33 for i in lines]).astype(np.uint8)).cuda()
    kw = dict(clusters=k, mode=host.MODE_RATIO if mode == "f" else host.MODE_FIXED, ratio=ratio, dist=dist, keep=True)
    r = P.encode_pass(sym, n, 0, n, means0, shards=1, coder="gpu", **kw)
    assert K.untranspose_text(r["qv"], n).cpu().numpy().tobytes() == ref["u"]
    blob = K.write_header(k, c, n) + b"".join(cb.serialize() for cb in r["codebooks"]) + O.WELL_DEBUG_STATE.tobytes()
    assert blob + r["streams"][0] == ref["q"]
    a = P.encode_pass(sym, n, 0, n, means0, shards=3, threads=2, coder="host", **kw)
    b = P.encode_pass(sym, n, 0, n, means0, shards=3, coder="gpu", **kw)
    assert a["streams"] == b["streams"]


@pytest.mark.parametrize("name,shards,threads", [("t2k_r10_M", 2, 2), ("t2k_c3_f05_M", 5, 3), ("cfg1_f05_L", 16, 8),
                                                  ("tiny5_r10", 3, 2)])
def test_every_shard_is_reference_decodable(name, shards, threads):
    from qvz_amd import host, kernels as K, pipeline as P
    rows, opts, n, c, sym, means0 = setup_case(name)
    r = P.encode_pass(sym, n, 0, n, means0, clusters=opts["k"], mode=opts["mode"], ratio=opts["ratio"], dist=opts["dist"],
                      shards=shards, threads=threads, keep=True)
    u = K.untranspose_text(r["qv"], n).cpu().numpy().tobytes()
    assert G.md5(u) == G.CASES[name]["u_md5"], "quantized symbols must not depend on the sharding of the coder"
    first = host.shard_bounds(n, shards)
    handles = [cb.handle for cb in r["codebooks"]]
    decoded = b""
    for g in range(shards):
        reads = first[g + 1] - first[g]
        if reads == 0:
            continue
        state = K.well_jump(O.WELL_DEBUG_STATE, first[g] * c // 4)      # generator at the shard's first draw, n = 0
        image = host.shard_file_image(c, reads, handles, state, r["streams"][g])
        decoded += O.decode(image)                                      # stands in for the reference's `qvz -x`
    assert decoded == u
    single = G.CASES[name]["size"]
    assert r["total_stream_bytes"] >= single * 0.98   # re-learning the model per shard costs bits, it never saves many


@pytest.mark.parametrize("name,shards", [("t2k_r10_M", 1), ("t2k_r10_M", 7), ("t2k_c3_f05_M", 4), ("cfg1_f05_L", 64),
                                         ("cfg1_r10_M", 300), ("const_f05", 2), ("uniform72_r20_A", 5), ("tiny5_r10", 2),
                                         ("rescale_140k_c4", 1), ("rescale_c2_100k_c3", 1), ("rescale_c2_100k_c3", 3),
                                         ("r35_high_rate", 3), ("wide_c151", 2), ("tiny5_r10", 4), ("c2_f08", 70),
                                         ("t2k_c3_f05_M", 130)])
def test_gpu_coder_streams_equal_host_coder_streams(name, shards, monkeypatch):
    """The GPU coder (model kernel + lane-per-stream interval chain over packed 8-byte records) and the host C coder must emit the
    same bytes for every shard; with one shard that is the reference's own stream."""
    from qvz_amd import host, pipeline as P
    rows, opts, n, c, sym, means0 = setup_case(name)
    kw = dict(clusters=opts["k"], mode=opts["mode"], ratio=opts["ratio"], dist=opts["dist"], shards=shards, keep=True)
    a = P.encode_pass(sym, n, 0, n, means0, threads=4, coder="host", **kw)
    b = P.encode_pass(sym, n, 0, n, means0, coder="gpu", **kw)
    assert a["stream_bytes"] == b["stream_bytes"]
    for g in range(shards):
        assert a["streams"][g] == b["streams"][g], f"shard {g} differs between host and GPU coder"
    # the record workspace squeezed to 8 reads per stream: many segments, coder state carried between launches
    small = host.planes_bytes_per_read(shards, c, opts["k"]) * 8
    d = P.encode_pass(sym, n, 0, n, means0, coder="gpu", work_bytes=small, **kw)
    assert d["streams"] == a["streams"], "segmented record workspace changed a stream"
    if shards == 1:
        assert b["total_stream_bytes"] == G.CASES[name]["size"]
        if G.CASES[name]["stored"]:
            assert b["streams"][0] == G.gz(name + ".q.gz")[-G.CASES[name]["size"]:]


def test_one_newton_step_is_enough_on_this_device(monkeypatch):
    """The coder's loaders refine v_rcp_f64 with ONE Newton step when that is within 2^-47 of 1 / (n 2^10) for every n a packed
    record can hold - the proof at `Record` needs 2^-46 - and the library measures it over all n = 1 .. 2^19 + 1024 before it
    relies on it. On gfx950 it must hold (2^-48.8 measured), or the fast kernel would silently be off; and the two-step
    kernel (test hook QVZ_CODER=newton=2) codes the same streams."""
    import ctypes as C
    from qvz_amd import pipeline as P
    from qvz_amd._lib import lib
    worst = C.c_double()
    assert lib().qvz_coder_rcp_one_step_ok_ex(C.byref(worst)) == 1, worst.value
    assert 0.0 < worst.value <= 2.0 ** -47
    rows, opts, n, c, sym, means0 = setup_case("cfg1_f05_L")
    kw = dict(clusters=opts["k"], mode=opts["mode"], ratio=opts["ratio"], dist=opts["dist"], shards=16, keep=True)
    a = P.encode_pass(sym, n, 0, n, means0, coder="gpu", **kw)
    monkeypatch.setenv("QVZ_CODER", "newton=2")
    assert lib().qvz_coder_rcp_one_step_ok() == 0
    b = P.encode_pass(sym, n, 0, n, means0, coder="gpu", **kw)
    assert a["streams"] == b["streams"]


@pytest.mark.parametrize("name,shards", [("t2k_c3_f05_M", 4), ("cfg1_f05_L", 16), ("rescale_c2_100k_c3", 1), ("r35_high_rate", 2)])
def test_gpu_model_kernels_agree(name, shards, monkeypatch):
    """The prefix-count model (a wave per (stream, column), tables in LDS) and the lane-per-column model (tables in
    HBM, the fall-back for codebooks whose column does not fit LDS) must leave the same records, i.e. the same streams."""
    from qvz_amd import host, pipeline as P
    rows, opts, n, c, sym, means0 = setup_case(name)
    kw = dict(clusters=opts["k"], mode=opts["mode"], ratio=opts["ratio"], dist=opts["dist"], shards=shards, keep=True)
    a = P.encode_pass(sym, n, 0, n, means0, coder="gpu", **kw)
    handles = [cb.handle for cb in a["codebooks"]]
    labels = a["labels"] if opts["k"] > 1 else None
    monkeypatch.setenv("QVZ_CODER", "columns")                          # test hook: the fall-back kernel whatever the tables' size
    _, streams, _ = host.encode_planes_gpu(a["qi"], a["st"], labels, n, c, handles, shards)
    assert streams == a["streams"]
    small = host.planes_bytes_per_read(shards, c, opts["k"]) * 128      # 128 reads per segment
    _, streams, _ = host.encode_planes_gpu(a["qi"], a["st"], labels, n, c, handles, shards, work_bytes=small)
    assert streams == a["streams"]
    monkeypatch.delenv("QVZ_CODER")
    _, streams, _ = host.encode_planes_gpu(a["qi"], a["st"], labels, n, c, handles, shards, work_bytes=small)
    assert streams == a["streams"]


def test_gpu_coder_reports_overflow_and_bad_feed():
    from qvz_amd import host, kernels as K, pipeline as P
    rows, opts, n, c, sym, means0 = setup_case("t2k_r10_M")
    r = P.encode_pass(sym, n, 0, n, means0, mode=opts["mode"], ratio=opts["ratio"], dist=opts["dist"], shards=2, coder="gpu",
                      keep=True)
    handles = [cb.handle for cb in r["codebooks"]]
    with pytest.raises(Exception, match="cap_per_shard"):
        host.encode_planes_gpu(r["qi"], r["st"], None, n, c, handles, 2, cap_per_shard=64)
    qi = r["qi"].clone()
    qi[3, 10] = 143
    with pytest.raises(Exception, match="model table"):
        host.encode_planes_gpu(qi, r["st"], None, n, c, handles, 2)
    with pytest.raises(Exception, match="fewer than 4 reads"):
        lib_small = torch.empty(64, dtype=torch.uint8, device="cuda")
        from qvz_amd._lib import lib
        import ctypes as C
        arr = (C.c_void_p * 1)(*handles)
        sizes = (C.c_uint64 * 2)()
        out = torch.empty(2 * 4096, dtype=torch.uint8, device="cuda")
        rc = lib().qvz_hip_encode_planes(C.c_void_p(r["qi"].data_ptr()), C.c_void_p(r["st"].data_ptr()), r["qi"].shape[1], None, n, c,
                                         1, arr, 2, C.c_void_p(out.data_ptr()), 4096, sizes, C.c_void_p(lib_small.data_ptr()), 64, None)
        if rc:
            raise RuntimeError(lib().qvz_hip_last_error().decode())


def test_read_sharded_ranks_equal_single_pass():
    """Two 'ranks' in one process: each quantizes its half with the global counters (summed by hand here, by RCCL in
    bench.py) and its WELL offset; the halves must reproduce the single-pass outputs bit for bit."""
    from qvz_amd import kernels as K
    rows = O.qgen1_rows(6, 9000, 40)
    n, c = 9000, 40
    from test_gpu_kernels import upload_sym
    full_sym = upload_sym(K, rows)
    cnt_full = K.pmf_hist(full_sym, n)
    halves = [(0, 4500), (4500, 9000)]      # 4500 is a multiple of 4
    cnt = None
    for a, b in halves:
        cnt = K.pmf_hist(upload_sym(K, rows[a:b]), b - a, cnt=cnt)
    assert torch.equal(cnt, cnt_full)
    d = K.distortion_matrix("M")
    cb, _ = K.generate_codebook(cnt[0], d, 0, 0.5)
    ct, qs = K.upload_tables([cb])
    dd = torch.from_numpy(d).cuda()
    full = K.quantize(full_sym, n, ct, qs, K.well_draws(O.WELL_DEBUG_STATE, 0, n, c), dd)
    for a, b in halves:
        part = K.quantize(upload_sym(K, rows[a:b]), b - a, ct, qs, K.well_draws(O.WELL_DEBUG_STATE, a, b - a, c), dd)
        assert torch.equal(part["qv"][:, : b - a], full["qv"][:, a:b])
        assert torch.equal(part["err"], full["err"][a:b])


@pytest.mark.parametrize("name", ["t2k_r10_M", "t2k_c3_f05_M"])
def test_rd_sweep_equals_separate_encodes(name):
    """Rate-distortion sweep (generate_rd.sh): labels, counters and draws computed once, every operating point must come
    out exactly as a separate encode of the same input."""
    from qvz_amd import host, pipeline as P
    rows, opts, n, c, sym, means0 = setup_case(name)
    points = [(host.MODE_RATIO, 0.3, "M"), (host.MODE_FIXED, 1.5, "L"), (host.MODE_RATIO, 0.8, "A"), (opts["mode"], opts["ratio"], opts["dist"])]
    kw = dict(clusters=opts["k"], shards=3, coder="gpu", keep=True)
    swept = P.rd_sweep(sym, n, 0, n, means0, points, **kw)
    for (mode, ratio, dist), got in zip(points, swept):
        alone = P.encode_pass(sym, n, 0, n, means0, mode=mode, ratio=ratio, dist=dist, **kw)
        assert got["streams"] == alone["streams"]
        assert got["distortion"] == alone["distortion"]
        assert torch.equal(got["qv"][:, :n], alone["qv"][:, :n])
    assert swept[-1]["total_stream_bytes"] >= G.CASES[name]["size"] * 0.98
    rates = [8.0 * r["total_stream_bytes"] / (n * c) for r in swept[:3]]
    assert rates[0] < rates[2]                       # -f 0.3 codes fewer bits than -f 0.8


def test_rccl_group_of_one_rank_changes_nothing():
    """The multi-GPU code path (RCCL all_reduce of k-means sums, counts, PMF counters and the distortion) on a
    one-rank process group: same streams as without a group. (More ranks are covered on CPU with gloo, test_dist_cpu.)"""
    import os
    import socket
    import torch.distributed as dist
    from qvz_amd import pipeline as P
    if dist.is_initialized():
        pytest.skip("a process group already exists in this process")
    with socket.socket() as sk:
        sk.bind(("127.0.0.1", 0))
        port = sk.getsockname()[1]
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    dist.init_process_group("nccl", rank=0, world_size=1, device_id=torch.device("cuda", 0))
    try:
        for name in ("t2k_c3_f05_M", "cfg1_f05_L"):
            rows, opts, n, c, sym, means0 = setup_case(name)
            kw = dict(clusters=opts["k"], mode=opts["mode"], ratio=opts["ratio"], dist=opts["dist"], shards=4, coder="gpu", keep=True)
            a = P.encode_pass(sym, n, 0, n, means0, **kw)
            b = P.encode_pass(sym, n, 0, n, means0, group=dist.group.WORLD, **kw)
            assert a["streams"] == b["streams"] and a["distortion"] == b["distortion"]
            assert a["kmeans_iterations"] == b["kmeans_iterations"]
    finally:
        dist.destroy_process_group()


def test_kept_streams_run_beside_the_streams_they_are_meant_to():
    """qvz_hip_streams_overlap: a stream never overlaps itself; the kept WELL / codebook streams of the pipeline were tested
    against the default stream and each other when they were made (a process's streams share a handful of hardware
    queues, and a queue runs its kernels in order), and the probe says so again; two encodes give the same stream object."""
    import ctypes
    from qvz_amd import pipeline as P
    from qvz_amd._lib import check, lib

    def overlap(a, b):
        flag = ctypes.c_int(-1)
        check(lib().qvz_hip_streams_overlap(a, b, ctypes.byref(flag)))
        return flag.value

    main = torch.cuda.current_stream()
    dev = torch.device("cuda", torch.cuda.current_device())
    assert overlap(main.cuda_stream, main.cuda_stream) == 0
    well = P.kept_stream(dev, "well", beside=(main,))
    assert well is P.kept_stream(dev, "well", beside=(main,))
    cb1 = P.kept_stream(dev, "codebook1", beside=(main, well))
    cb2 = P.kept_stream(dev, "codebook2", beside=(main, well, cb1))
    assert overlap(well.cuda_stream, well.cuda_stream) == 0
    pairs = [(main, well), (main, cb1), (well, cb1), (main, cb2), (well, cb2), (cb1, cb2)]
    assert all(overlap(a.cuda_stream, b.cuda_stream) == 1 for a, b in pairs)

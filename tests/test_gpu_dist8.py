"""BASELINE configs[4]'s partition - the reads over EIGHT parts, -c 3 -d M -r 1.0, 150 columns - rehearsed on the one GPU of a
test box against the CPU ORACLE's single-process result (the 1B-read run itself needs the 8-GPU node; reference
src/cluster.c:136-171, src/codebook.c:185-205 are what the parts must agree on):

  * the PRODUCT's schedule (qvz_amd/pipeline.py, what bench.py --gpus 8 runs) with eight ranks as threads of this process
    (pipeline.LocalWorld; a box allows at most six processes on its card, so eight gloo processes on cuda:0 are not to be had)
    and the HIP kernels as compute backend;
  * the C driver (host/qvz_multi.c: qvz_encode_file_devices) with the ordinal 0 listed eight times.

2M reads x 150 columns: libc rand() puts the three centroid reads on ranks 7, 6 and 4; every part's labels, the summed counters,
the codebooks, the quantized rows, the distortion and every stream - byte for byte, the oracle coding the part's reads from the
WELL state it steps to itself - must be the oracle's."""
import numpy as np
import pytest

import oracle_lib as O
from test_dist_cpu import run_local_world

pytestmark = pytest.mark.gpu
torch = pytest.importorskip("torch")

N, COLS, K, SEED, WORLD = 2_000_000, 150, 3, 21, 8
_oracle = {}


def oracle_side():
    """the single-process oracle result of the whole file (once per session: ~40 s of one host core)"""
    if not _oracle:
        rows = O.qgen1_rows(SEED, N, COLS)
        labels, means, counts, iters = O.kmeans(rows, K)
        assert iters > 1 and counts.min() > 0
        cnt = O.stats(rows, labels, K)
        dmat = O.distortion("M")
        cbs = [O.generate_codebook(cnt[j], dmat, O.MODE_FIXED, 1.0) for j in range(K)]
        single = O.encode_stream(rows, labels, cbs, dmat)
        _oracle.update(rows=rows, labels=labels, means=means, iters=iters, cnt=cnt, dmat=dmat, cbs=cbs, u=single["u"],
                       distortion=single["distortion"], blob=b"".join(cb.serialize() for cb in cbs))
    return _oracle


def oracle_stream(o, a, b):
    """the oracle's stream of the reads [a, b) as a part of the file: WELL state stepped to the part's first draw"""
    state = O.well_states_at([a * COLS // 4])[0]
    return state, O.encode_stream(o["rows"][a:b], o["labels"][a:b], o["cbs"], o["dmat"], well_state=state, want_u=False)["stream"]


def test_centroid_reads_lie_on_three_ranks():
    from qvz_amd import pipeline as P
    lines = P.reference_centroid_lines(N, K)
    assert lines == [int(x) for x in O.pick_centroids(N, K)]
    owners = sorted(next(r for r in range(WORLD) if P.rank_read_range(N, WORLD, r)[0] <= ln < P.rank_read_range(N, WORLD, r)[1]) for ln in lines)
    assert owners == [4, 6, 7]


def test_eight_ranks_with_the_hip_backend_equal_the_oracle():
    from qvz_amd import host, kernels as Kn, pipeline as P
    shards = 3
    got = run_local_world(WORLD, N, COLS, SEED, lambda first, n: Kn.qgen1(SEED, n, COLS, first=first), lambda: P.HipBackend(0),
                          clusters=K, mode=host.MODE_FIXED, ratio=1.0, dist="M", shards=shards, coder="gpu")
    o = oracle_side()
    assert got[0][0] == 0 and got[-1][1] == N and all(a[1] == b[0] and a[1] % 4 == 0 for a, b in zip(got, got[1:]))
    for rank, (first, last, r, _) in enumerate(got):
        n = last - first
        assert r["kmeans_iterations"] == o["iters"], rank
        assert r["means"].cpu().numpy().tobytes() == (o["means"] - 33).tobytes(), rank
        assert np.array_equal(r["labels"][:n].cpu().numpy(), o["labels"][first:last]), f"rank {rank}: labels"
        assert np.array_equal(r["cnt"].cpu().numpy().astype(np.uint32), o["cnt"]), f"rank {rank}: counters after the sum"
        assert b"".join(cb.serialize() for cb in r["codebooks"]) == o["blob"], f"rank {rank}: codebooks"
        assert np.array_equal(r["qv"][:, :n].cpu().numpy().T + 33, o["u"][first:last, :COLS]), f"rank {rank}: quantized rows"
        assert abs(r["distortion"] - o["distortion"]) < 1e-9 * max(1.0, o["distortion"])
        bounds = host.shard_bounds(n, shards)
        for g in range(shards):
            _, want = oracle_stream(o, first + bounds[g], first + bounds[g + 1])
            assert r["streams"][g] == want, f"rank {rank} shard {g}: stream differs from the oracle's"


def test_c_driver_with_eight_parts_writes_the_oracles_files(tmp_path):
    """qvz_encode_file_devices with devices = [0] * 8 (host-staged sums): the shard set, split into reference-format files, is
    header + the oracle's codebooks + the oracle-stepped WELL state + the oracle's stream for every shard; the -u text and the
    distortion are the single-process oracle's."""
    from qvz_amd import host
    from test_gpu_encode import fresh_process_rand
    o = oracle_side()
    shards = 24
    inp, out, u = str(tmp_path / "in.txt"), str(tmp_path / "many.qset"), str(tmp_path / "many.u")
    open(inp, "wb").write(o["rows"].tobytes())
    fresh_process_rand()
    r = host.encode_file_devices(inp, out, [0] * WORLD, shards, clusters=K, mode=host.MODE_FIXED, ratio=1.0, dist="M", u_path=u,
                                 seed_mode=host.SEED_DEBUG)
    assert r.kmeans_iterations == o["iters"] and (r.lines, r.columns) == (N, COLS)
    assert open(u, "rb").read() == o["u"].tobytes()
    assert abs(r.distortion - o["distortion"]) < 1e-9 * max(1.0, o["distortion"])
    prefix = str(tmp_path / "part")
    host.shardset_split(out, prefix)
    bounds = host.shard_bounds(N, shards)
    total = 0
    for g in range(shards):
        a, b = bounds[g], bounds[g + 1]
        state, stream = oracle_stream(o, a, b)
        want = O.write_header(K, COLS, b - a) + o["blob"] + state.tobytes() + stream
        assert open(f"{prefix}.{g:04d}", "rb").read() == want, f"shard {g}"
        total += len(stream)
    assert r.stream_bytes == total

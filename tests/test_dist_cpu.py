"""world_size-2 `gloo` tests (CPU) of the read-sharding schedule of qvz_amd/pipeline.py (what bench.py --gpus N runs):
the PRODUCT's encode_pass, fetch_centroids and rank_read_range are driven here - which reads a rank owns, the centroid
reads fetched from their owners, the all-reduces of the k-means accumulators, the PMF counters and the distortion, the
WELL offset of every shard - with the CPU oracle injected as the compute backend (tests/cpu_backend.py; no GPU in this
environment). On the GPU box the same code runs over RCCL with the HIP backend."""
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


from cpu_backend import OracleBackend, advance_well, sym_from_rows  # noqa: E402

N, COLS, K, SEED = 3000, 24, 2, 13


def _rank_main(rank, world, port, out, k, mode, ratio, dist_kind):
    """One rank of the PRODUCT's schedule (qvz_amd.pipeline.encode_pass with its `group` branch, fetch_centroids,
    rank_read_range) over gloo, with the CPU oracle as compute backend."""
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from qvz_amd import pipeline as P
    first, last = P.rank_read_range(N, world, rank)
    rows = O.qgen1_rows(SEED, last - first, COLS, first=first)             # this rank's reads only
    n = last - first
    sym = sym_from_rows(rows)
    be = OracleBackend()
    r = P.encode_pass(sym, n, first, N, None, clusters=k, mode=mode, ratio=ratio, dist=dist_kind, shards=1, group=dist.group.WORLD,
                      keep=True, backend=be)
    blob = b"".join(cb.serialize() for cb in r["codebooks"])
    out.put((rank, first, last, r["kmeans_iterations"], r["means"].numpy().tobytes(), r["labels"][:n].numpy().tobytes(),
             r["cnt"].numpy().astype(np.uint32).tobytes(), blob, (r["qv"][:, :n].numpy().T + 33).tobytes(), r["distortion"],
             r["streams"][0], be.well.tobytes()))
    dist.destroy_process_group()


@pytest.mark.parametrize("k,mode,ratio,dist_kind", [(K, O.MODE_RATIO, 0.5, "M"), (1, O.MODE_FIXED, 1.0, "L")])
def test_two_rank_gloo_protocol_equals_single_process(k, mode, ratio, dist_kind):
    world = 2
    ctx = mp.get_context("spawn")
    out = ctx.Queue()
    port = free_port()
    procs = [ctx.Process(target=_rank_main, args=(r, world, port, out, k, mode, ratio, dist_kind)) for r in range(world)]
    for p in procs:
        p.start()
    got = sorted(out.get(timeout=120) for _ in range(world))
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0

    rows = O.qgen1_rows(SEED, N, COLS)
    labels, means, counts_k, iters = O.kmeans(rows, k)
    cnt = O.stats(rows, labels, k)
    dmat = O.distortion(dist_kind)
    cbs = [O.generate_codebook(cnt[j], dmat, mode, ratio) for j in range(k)]
    single = O.encode_stream(rows, labels, cbs, dmat)
    blob = b"".join(cb.serialize() for cb in cbs)

    assert got[0][1] == 0 and got[-1][2] == N and got[0][2] == got[1][1]
    assert got[0][2] % 4 == 0, "shards must start on WELL word boundaries"
    u = b""
    for rank, first, last, it, m, lab, c, b, ushard, distortion, stream, state in got:
        assert it == iters
        assert m == (means - 33).tobytes()                                 # the product keeps means as symbols (byte - 33)
        assert lab == labels[first:last].tobytes()
        assert c == cnt.tobytes()
        assert b == blob
        assert abs(distortion - single["distortion"]) < 1e-9 * max(1.0, single["distortion"])
        ushard = np.frombuffer(ushard, dtype=np.uint8).reshape(last - first, COLS)
        ushard = np.concatenate([ushard, np.full((last - first, 1), 10, np.uint8)], axis=1).tobytes()
        u += ushard
        # each shard is a reference-format file of its own: header + codebooks + rotated WELL state + stream
        image = O.write_header(k, COLS, last - first) + blob + state + stream
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


def run_local_world(world, n_total, cols, seed, make_sym, make_backend, **kw):
    """The product's schedule with `world` ranks as threads of this process (pipeline.LocalWorld): every rank owns
    rank_read_range's reads, gets make_sym(first, n) as its matrix and make_backend() (serialised) as its compute backend.
    -> [(first, last, result of encode_pass)] by rank."""
    import threading
    from qvz_amd import pipeline as P
    lw = P.LocalWorld(world)
    backends = [lw.serial(make_backend()) for _ in range(world)]              # made on this thread: one-time set-up is not re-entrant
    out, errors = [None] * world, []

    def rank_main(rank):
        try:
            first, last = P.rank_read_range(n_total, world, rank)
            sym = make_sym(first, last - first)
            out[rank] = (first, last, P.encode_pass(sym, last - first, first, n_total, None, group=lw.group(rank), keep=True,
                                                    backend=backends[rank], **kw), backends[rank])
        except BaseException as exc:  # noqa: BLE001 - a rank that dies must not leave the others in the barrier
            errors.append((rank, repr(exc)))
            lw._barrier.abort()

    threads = [threading.Thread(target=rank_main, args=(r,)) for r in range(world)]
    for t in threads:
        t.start()
    for t in threads:
        t.join(timeout=900)
    assert not errors, errors
    return out


def test_eight_rank_partition_equals_single_process():
    """BASELINE configs[4]'s partition - eight parts, -c 3 -d M -r 1.0 - at a size the oracle finishes in seconds, ranks as
    threads (pipeline.LocalWorld): the centroid reads lie on three different ranks, every rank's labels, all-reduced counters,
    codebooks, quantized rows and stream equal the single-process oracle's; the streams byte for byte, against the oracle coding
    the rank's reads from the WELL state it steps to itself."""
    from qvz_amd import pipeline as P
    n_total, cols, k, seed, world = 10000, 24, 3, 1, 8
    lines = P.reference_centroid_lines(n_total, k)
    owners = {next(r for r in range(world) if P.rank_read_range(n_total, world, r)[0] <= ln < P.rank_read_range(n_total, world, r)[1]) for ln in lines}
    assert len(owners) == 3, (lines, owners)
    got = run_local_world(world, n_total, cols, seed, lambda first, n: sym_from_rows(O.qgen1_rows(seed, n, cols, first=first)), OracleBackend,
                          clusters=k, mode=O.MODE_FIXED, ratio=1.0, dist="M", shards=1)
    rows = O.qgen1_rows(seed, n_total, cols)
    labels, means, _, iters = O.kmeans(rows, k)
    cnt = O.stats(rows, labels, k)
    dmat = O.distortion("M")
    cbs = [O.generate_codebook(cnt[j], dmat, O.MODE_FIXED, 1.0) for j in range(k)]
    single = O.encode_stream(rows, labels, cbs, dmat)
    blob = b"".join(cb.serialize() for cb in cbs)
    assert got[0][0] == 0 and got[-1][1] == n_total and all(a[1] == b[0] for a, b in zip(got, got[1:]))
    for first, last, r, be in got:
        assert r["kmeans_iterations"] == iters and r["means"].numpy().tobytes() == (means - 33).tobytes()
        assert r["labels"][: last - first].numpy().tobytes() == labels[first:last].tobytes()
        assert r["cnt"].numpy().astype(np.uint32).tobytes() == cnt.tobytes()
        assert b"".join(cb.serialize() for cb in r["codebooks"]) == blob
        assert abs(r["distortion"] - single["distortion"]) < 1e-9 * max(1.0, single["distortion"])
        assert (r["qv"][:, : last - first].numpy().T + 33).tobytes() == single["u"][first:last, :cols].tobytes()
        state = O.well_states_at([first * cols // 4])[0]
        assert r["streams"][0] == O.encode_stream(rows[first:last], labels[first:last], cbs, dmat, well_state=state)["stream"]

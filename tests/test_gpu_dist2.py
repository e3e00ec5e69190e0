"""Two ranks of the PRODUCT's read-sharded schedule (qvz_amd/pipeline.py: what bench.py --gpus N runs) with the REAL compute
backend - the HIP kernels - on the one GPU a test box has: both ranks use cuda:0, the collectives go over `gloo` (RCCL refuses
two ranks on one device; on a multi-GPU node the same code runs over RCCL, one rank per GPU). tests/test_dist_cpu.py drives the
same schedule with the CPU oracle as backend; this one proves that nothing in the HIP path depends on a rank seeing the whole
file: k-means labels and means, PMF counters, codebooks and the quantized values of either rank equal the oracle's
single-process results for the whole file, and every stream a rank codes is the reference-format file of its reads (WELL state
advanced to the rank's and the shard's first draw)."""
import os
import socket

import numpy as np
import pytest

import oracle_lib as O

pytestmark = pytest.mark.gpu
torch = pytest.importorskip("torch")
import torch.distributed as dist  # noqa: E402
import torch.multiprocessing as mp  # noqa: E402

N, COLS, SEED, SHARDS = 20000, 30, 17, 3


def _rank_main(rank, world, port, out, k, mode, ratio, dist_kind):
    try:
        os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
        torch.cuda.set_device(0)
        dist.init_process_group("gloo", rank=rank, world_size=world)
        from qvz_amd import host, kernels as K, pipeline as P
        first, last = P.rank_read_range(N, world, rank)
        n = last - first
        sym = K.qgen1(SEED, n, COLS, first=first)                            # this rank's reads only
        r = P.encode_pass(sym, n, first, N, None, clusters=k, mode=mode, ratio=ratio, dist=dist_kind, shards=SHARDS, coder="gpu",
                          group=dist.group.WORLD, keep=True)
        bounds = host.shard_bounds(n, SHARDS)
        states = [K.well_jump(O.WELL_DEBUG_STATE, (first + b) * COLS // 4).tobytes() for b in bounds[:-1]]
        out.put((rank, first, last, r["kmeans_iterations"], r["means"].cpu().numpy().tobytes(), r["labels"][:n].cpu().numpy().tobytes(),
                 r["cnt"].cpu().numpy().astype(np.uint32).tobytes(), b"".join(cb.serialize() for cb in r["codebooks"]),
                 (r["qv"][:, :n].cpu().numpy().T + 33).tobytes(), r["distortion"], r["streams"], bounds, states))
        dist.destroy_process_group()
    except Exception as exc:  # noqa: BLE001 - handed to the parent, which fails or skips with it
        out.put((rank, "error", repr(exc)))


@pytest.mark.parametrize("k,mode,ratio,dist_kind", [(2, "f", 0.5, "M"), (1, "r", 1.0, "L")])
def test_two_ranks_on_one_gpu_with_the_hip_backend(k, mode, ratio, dist_kind):
    from qvz_amd import host
    world = 2
    ctx = mp.get_context("spawn")
    out = ctx.Queue()
    with socket.socket() as sk:
        sk.bind(("127.0.0.1", 0))
        port = sk.getsockname()[1]
    hmode, omode = (host.MODE_RATIO, O.MODE_RATIO) if mode == "f" else (host.MODE_FIXED, O.MODE_FIXED)
    procs = [ctx.Process(target=_rank_main, args=(r, world, port, out, k, hmode, ratio, dist_kind)) for r in range(world)]
    for p in procs:
        p.start()
    got = sorted(out.get(timeout=300) for _ in range(world))
    for p in procs:
        p.join(timeout=60)
    errors = [g for g in got if g[1] == "error"]
    if errors and any("gloo" in e[2].lower() or "cuda" in e[2].lower() and "allreduce" in e[2].lower() for e in errors):
        pytest.skip(f"this torch build's gloo does not reduce device tensors: {errors[0][2][:200]}")
    assert not errors, errors

    rows = O.qgen1_rows(SEED, N, COLS)
    labels, means, counts_k, iters = O.kmeans(rows, k)
    cnt = O.stats(rows, labels, k)
    dmat = O.distortion(dist_kind)
    cbs = [O.generate_codebook(cnt[j], dmat, omode, ratio) for j in range(k)]
    single = O.encode_stream(rows, labels, cbs, dmat)
    blob = b"".join(cb.serialize() for cb in cbs)
    assert got[0][1] == 0 and got[-1][2] == N and got[0][2] == got[1][1] and got[0][2] % 4 == 0
    u = b""
    for rank, first, last, it, m, lab, c, b, ushard, distortion, streams, bounds, states in got:
        assert it == iters
        assert m == (means - 33).tobytes()
        assert lab == labels[first:last].tobytes()
        assert c == cnt.tobytes(), "PMF counters after the all-reduce are the whole file's"
        assert b == blob, "both ranks generate the whole file's codebooks"
        assert abs(distortion - single["distortion"]) < 1e-9 * max(1.0, single["distortion"])
        ushard = np.frombuffer(ushard, dtype=np.uint8).reshape(last - first, COLS)
        ushard = np.concatenate([ushard, np.full((last - first, 1), 10, np.uint8)], axis=1)
        u += ushard.tobytes()
        for g in range(SHARDS):                                             # every stream is a reference-format file of its reads
            a, e = bounds[g], bounds[g + 1]
            image = O.write_header(k, COLS, e - a) + blob + states[g] + streams[g]
            assert O.decode(image) == ushard[a:e].tobytes(), f"rank {rank} shard {g}"
    assert u == single["u"].tobytes(), "the ranks' quantized values, concatenated, are the single-process -u output"


def test_bench_launch_line_with_two_ranks(tmp_path):
    """The driver's N > 1 command for bench.py, word for word (python -m torch.distributed.run --nnodes=1 --nproc-per-node 2
    --master-addr 127.0.0.1 --master-port P bench.py --gpus 2 ...), on the one GPU of a test box: QVZ_BENCH_ONE_DEVICE puts both
    ranks on cuda:0 and QVZ_BENCH_BACKEND=gloo carries the collectives. Rank 0 alone prints, one JSON line, with the whole-job
    value (both ranks' symbols over the max-over-ranks time) and the keys the contract names."""
    import json
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    s = socket.socket(); s.bind(("127.0.0.1", 0)); port = s.getsockname()[1]; s.close()
    reads, steps = 2_000_000, 2
    env = dict(os.environ, QVZ_BENCH_ONE_DEVICE="1", QVZ_BENCH_BACKEND="gloo")
    p = subprocess.run([sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2", "--master-addr", "127.0.0.1",
                        "--master-port", str(port), os.path.join(root, "bench.py"), "--gpus", "2", "--reads", str(reads), "--steps", str(steps),
                        "--warmup", "1"], cwd=root, env=env, capture_output=True, text=True, timeout=400)
    assert p.returncode == 0, p.stderr[-2000:]
    lines = [l for l in p.stdout.splitlines() if l.startswith("{")]
    assert len(lines) == 1, p.stdout[-2000:]
    d = json.loads(lines[0])
    assert d["n_gpus"] == 2 and d["steps"] == steps and d["warmup"] == 1 and d["scaling"] == "weak" and d["higher_is_better"] is True
    assert d["metric"] == "quality-symbols/s (encode)" and d["unit"] == "quality-symbols/s" and d["dtype"] == "u8" and d["vs_baseline"] is None
    assert d["config"]["reads_per_gpu"] == reads and d["config"]["parallelism"] == "read-sharded x2"
    assert d["value"] == pytest.approx(2 * reads * 150 / (d["ms_per_step"] * 1e-3), rel=1e-3)
    assert {"bound", "achieved", "peak", "unit", "frac", "traffic"} <= set(d["roofline"])
    assert "cpu_baseline" not in d or d["cpu_baseline"] is None            # rank 0 at N = 1 only
    assert 0.5 < d["bits_per_symbol"] < 3.0

"""Full-size parity (BASELINE.json configs[1], configs[2] and configs[3]: 10M / 100M reads x 150 columns on one MI355X), inside -m gpu.

What can be checked at that size without a 100M-read CPU run:
  (a) every one of the 8192 GPU-coded streams decodes (qvz_hip_decode_shards) to exactly the quantizer's output, cluster
      ids included - the encode -> decode round trip over all 1.5e10 symbols;
  (b) the PMF counters of the whole matrix, copied from the device, go through the CPU ORACLE's codebook generation
      (orc_generate_codebook, N-independent work): its bytes must equal the GPU-generated codebooks';
  (c) 68 of the 8192 shards - the first two, the last two and 64 drawn pseudo-randomly - are re-encoded by the oracle's
      start_qv_compression restatement (orc_encode_stream) from CPU-generated reads: quantized rows and the arithmetic
      stream must equal the GPU's byte for byte. The WELL state at each shard's first draw is found by STEPPING the
      oracle's generator through the file's whole draw stream (3.75e9 words, a few seconds of one host core) - no
      product code is involved in the oracle's leg, the jump-ahead included;
  (d) counters: every (cluster, column) tile sums to the number of reads labelled with that cluster;
  (e) the labels and the counters of the WHOLE matrix against the oracle: host processes (tests/oracle_workers.py, no GPU)
      regenerate all reads from QGEN-1's definition, assign each to the nearest of the means the last assignment pass measured
      against (orc_assign, cluster.c:136-187) and count its symbol pairs under the oracle's own label (orc_stats,
      codebook.c:185-205): every one of the 1e8 labels and every one of the k x 150 x 72 x 72 counters must equal the GPU's -
      nothing of (b) and (c) rests on labels or counters only the GPU has seen.
"""
import numpy as np
import pytest

import oracle_lib as O

pytestmark = pytest.mark.gpu
torch = pytest.importorskip("torch")

C, SHARDS = 150, 8192
CONFIGS = {
    "configs1_c1_M_r10": dict(reads=10_000_000, clusters=1, mode="r", ratio=1.0, dist="M", seed=2),
    "configs2_c1_L_f05": dict(reads=100_000_000, clusters=1, mode="f", ratio=0.5, dist="L", seed=3),
    "configs3_c3_M_r10": dict(reads=100_000_000, clusters=3, mode="r", ratio=1.0, dist="M", seed=3),
}
_well_cache = {}


def oracle_shards(n):
    """The shards the oracle re-encodes and the WELL state (n = 0 form) at each one's first draw, by stepping."""
    if n not in _well_cache:
        from qvz_amd import host
        first = host.shard_bounds(n, SHARDS)
        rng = np.random.default_rng(20261004)
        picks = sorted({0, 1, SHARDS - 2, SHARDS - 1} | set(int(g) for g in rng.choice(np.arange(2, SHARDS - 2), size=64, replace=False)))
        states = O.well_states_at([first[g] * C // 4 for g in picks])
        _well_cache[n] = (first, picks, states)
    return _well_cache[n]


@pytest.mark.parametrize("name", sorted(CONFIGS))
def test_full_size_encode_round_trip_and_oracle_parity(name):
    from qvz_amd import host, kernels as K, pipeline as P
    cfg = CONFIGS[name]
    import gc
    host.release_workspace()                 # the coder's kept record workspace of an earlier test
    gc.collect()
    torch.cuda.empty_cache()                 # blocks an earlier test left in torch's caching allocator count as used otherwise
    free, _ = torch.cuda.mem_get_info()
    k, seed = cfg["clusters"], cfg["seed"]
    mode = host.MODE_RATIO if cfg["mode"] == "f" else host.MODE_FIXED
    omode = O.MODE_RATIO if cfg["mode"] == "f" else O.MODE_FIXED
    n = cfg["reads"]
    need = 250 if n > 50_000_000 else 40
    if free < need * 2 ** 30:
        pytest.skip(f"needs ~{need} GB of HBM, {free >> 30} GiB free")
    sym = K.qgen1(seed, n, C)
    r = P.encode_pass(sym, n, 0, n, None, clusters=k, mode=mode, ratio=cfg["ratio"], dist=cfg["dist"], shards=SHARDS,
                      coder="gpu", keep_device=True, want_reuse=True)
    sizes = r["stream_bytes"]
    assert len(sizes) == SHARDS and min(sizes) > 0
    labels = r["labels"]

    # (d) counters: one count per read and column, in the tile of the read's cluster
    cnt = r["reuse"]["cnt"]
    per_cluster = torch.bincount(labels[:n].to(torch.int64), minlength=k)
    tile_sums = cnt.sum(dim=(2, 3), dtype=torch.int64)                           # (k, C)
    assert torch.equal(tile_sums, per_cluster[:, None].expand(k, C))

    # (e) every label and every counter against the oracle's own, all reads regenerated on host cores
    import oracle_workers as W
    means_raw = (r["reuse"]["means_assigned"].cpu().numpy() + 33).astype(np.uint8) if k > 1 else None
    labels_h = labels[:n].cpu().numpy() if k > 1 else None
    ocnt, mismatch, ocounts = W.recount(seed, n, C, k, means_raw=means_raw, gpu_labels=labels_h)
    assert mismatch < 0, f"read {mismatch}: the GPU's k-means label differs from the oracle's"
    assert np.array_equal(ocounts.astype(np.int64), per_cluster.cpu().numpy())
    assert np.array_equal(ocnt, cnt.cpu().numpy().astype(np.uint32)), "PMF counters of the whole matrix differ from the oracle's"
    del ocnt, labels_h

    # (a) decode all streams on the GPU: must reproduce the quantizer's plane (and the labels)
    qv, lab = host.decode_shards_gpu(r["coded"], r["cap_per_shard"], sizes, n, C, [cb.handle for cb in r["codebooks"]], SHARDS,
                                     r["reuse"]["draws"])
    assert torch.equal(qv[:, :n], r["qv"][:, :n]), "decoded symbols differ from the quantizer's output"
    if k > 1:
        assert torch.equal(lab[:n], labels[:n]), "decoded cluster ids differ from the k-means labels"
    del qv, lab

    # (b) the oracle generates the codebooks from the full-size counters
    dmat = O.distortion(cfg["dist"])
    cnt_h = cnt.cpu().numpy().astype(np.uint32)
    ocbs = [O.generate_codebook(cnt_h[j], dmat, omode, cfg["ratio"]) for j in range(k)]
    for j in range(k):
        assert ocbs[j].serialize() == r["codebooks"][j].serialize(), f"codebook of cluster {j} differs from the oracle's"

    # (c) 68 shards re-encoded by the oracle from CPU-generated reads, WELL states by stepping the oracle's generator
    first, picks, states = oracle_shards(n)
    assert len(picks) == 68
    coded = r["coded"]
    cap = r["cap_per_shard"]
    for g, state in zip(picks, states):
        a, b = first[g], first[g + 1]
        rows = O.qgen1_rows(seed, b - a, C, first=a)
        assert np.array_equal(sym[:, a:b].cpu().numpy(), (rows[:, :C].T - 33).astype(np.uint8))
        lab_g = np.ascontiguousarray(labels[a:b].cpu().numpy())
        enc = O.encode_stream(rows, lab_g, ocbs, dmat, well_state=state)
        got_u = (r["qv"][:, a:b].cpu().numpy().T + 33).astype(np.uint8)
        assert np.array_equal(got_u, enc["u"][:, :C]), f"shard {g}: quantized rows differ from the oracle's"
        stream = coded[g * cap: g * cap + sizes[g]].cpu().numpy().tobytes()
        assert stream == enc["stream"], f"shard {g}: GPU-coded stream differs from the oracle's"
    # the product's jump-ahead lands on the same states (it is what a shard file's header carries)
    for g, state in list(zip(picks, states))[::17]:
        assert np.array_equal(K.well_jump(O.WELL_DEBUG_STATE, first[g] * C // 4), state)
    del r, sym, coded, cnt, labels
    host.release_workspace()
    gc.collect()
    torch.cuda.empty_cache()


def test_configs4_rank_7_of_8_on_one_gpu():
    """BASELINE configs[4] (1B reads x 150 columns, -c 3 -d M -r 1.0 over 8 GPUs) cannot run here; what one GPU can prove of it:
    rank 7's share - reads [875M, 1B), 125M x 150 - goes through the product's N > 1 code path (pipeline.encode_pass with a
    one-rank RCCL group: all-reduces of the k-means sums, the PMF counters and the distortion), fits the 288 GB, and draws from
    the WELL stream 3.28e10 words in (first_read * C / 4 > 2^32) exactly: the first and the last of its shards are re-encoded
    by the oracle with WELL states that come from the REFERENCE's generator stepped that far (tests/golden/well_far.json)
    and on from there by the oracle's. Centroids: the three reads libc rand() picks among the 1B are generated by the oracle
    (QGEN-1 is counter based), as the ranks that own them would contribute them."""
    import gc
    import os
    import socket
    import torch.distributed as dist
    import golden_util as G
    from qvz_amd import host, kernels as K, pipeline as P
    host.release_workspace()
    gc.collect()
    torch.cuda.empty_cache()
    free, _ = torch.cuda.mem_get_info()
    if free < 262 * 2 ** 30:
        pytest.skip(f"needs ~262 GB of HBM, {free >> 30} GiB free")
    if dist.is_initialized():
        pytest.skip("a process group already exists in this process")
    n_total, world, rank, k, seed = 1_000_000_000, 8, 7, 3, 4
    first_read, last = P.rank_read_range(n_total, world, rank)
    n = last - first_read
    assert (first_read, n) == (875_000_000, 125_000_000)
    far = {w: st for w, st, _ in G.well_far()}
    word0 = first_read * C // 4
    assert word0 in far and word0 > 2 ** 32
    lines = P.reference_centroid_lines(n_total, k)
    means0 = torch.from_numpy(np.stack([O.qgen1_rows(seed, 1, C, first=ln)[0, :C] - 33 for ln in lines]).astype(np.uint8)).cuda()
    with socket.socket() as sk:
        sk.bind(("127.0.0.1", 0))
        port = sk.getsockname()[1]
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    dist.init_process_group("nccl", rank=0, world_size=1, device_id=torch.device("cuda", 0))
    try:
        sym = K.qgen1(seed, n, C, first=first_read)
        r = P.encode_pass(sym, n, first_read, n_total, means0, clusters=k, mode=host.MODE_FIXED, ratio=1.0, dist="M", shards=SHARDS,
                          coder="gpu", keep_device=True, want_reuse=True, group=dist.group.WORLD)
    finally:
        dist.destroy_process_group()
    sizes = r["stream_bytes"]
    assert len(sizes) == SHARDS and min(sizes) > 0
    labels, cnt = r["labels"], r["reuse"]["cnt"]
    # every stream decodes to the quantizer's plane (the draws are the ones of reads 875M.. of the 1B-read file)
    qv, lab = host.decode_shards_gpu(r["coded"], r["cap_per_shard"], sizes, n, C, [cb.handle for cb in r["codebooks"]], SHARDS,
                                     r["reuse"]["draws"])
    assert torch.equal(qv[:, :n], r["qv"][:, :n]) and torch.equal(lab[:n], labels[:n])
    del qv, lab
    # the oracle's leg: codebooks from this rank's counters, two shards from reads and WELL states it makes itself
    dmat = O.distortion("M")
    cnt_h = cnt.cpu().numpy().astype(np.uint32)
    ocbs = [O.generate_codebook(cnt_h[j], dmat, O.MODE_FIXED, 1.0) for j in range(k)]
    for j in range(k):
        assert ocbs[j].serialize() == r["codebooks"][j].serialize()
    first = host.shard_bounds(n, SHARDS)
    picks = [0, SHARDS - 1]
    states = O.well_states_at([first[g] * C // 4 for g in picks], state0=far[word0])
    coded, cap = r["coded"], r["cap_per_shard"]
    for g, state in zip(picks, states):
        a, b = first[g], first[g + 1]
        rows = O.qgen1_rows(seed, b - a, C, first=first_read + a)
        assert np.array_equal(sym[:, a:b].cpu().numpy(), (rows[:, :C].T - 33).astype(np.uint8))
        enc = O.encode_stream(rows, np.ascontiguousarray(labels[a:b].cpu().numpy()), ocbs, dmat, well_state=state)
        assert np.array_equal((r["qv"][:, a:b].cpu().numpy().T + 33).astype(np.uint8), enc["u"][:, :C]), f"shard {g}: quantized rows differ"
        assert coded[g * cap: g * cap + sizes[g]].cpu().numpy().tobytes() == enc["stream"], f"shard {g}: stream differs from the oracle's"
        np.testing.assert_array_equal(K.well_jump(O.WELL_DEBUG_STATE, (first_read + a) * C // 4), state)
    del r, sym, coded, cnt, labels
    host.release_workspace()
    gc.collect()
    torch.cuda.empty_cache()

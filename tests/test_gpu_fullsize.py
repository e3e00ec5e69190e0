"""Full-size parity (BASELINE.json configs[2] and configs[3]: 100M reads x 150 columns on one MI355X), inside -m gpu.

What can be checked at that size without a 100M-read CPU run:
  (a) every one of the 8192 GPU-coded streams decodes (qvz_hip_decode_shards) to exactly the quantizer's output, cluster
      ids included - the encode -> decode round trip over all 1.5e10 symbols;
  (b) the PMF counters of the whole matrix, copied from the device, go through the CPU ORACLE's codebook generation
      (orc_generate_codebook, N-independent work): its bytes must equal the GPU-generated codebooks';
  (c) single shards (the first, and one in the middle with its jumped WELL state) are re-encoded by the oracle's
      start_qv_compression restatement (orc_encode_stream) from CPU-generated reads: quantized rows and the arithmetic
      stream must equal the GPU's byte for byte;
  (d) counters: every (cluster, column) tile sums to the number of reads labelled with that cluster.
"""
import numpy as np
import pytest

import oracle_lib as O

pytestmark = pytest.mark.gpu
torch = pytest.importorskip("torch")

N, C, SHARDS = 100_000_000, 150, 8192
CONFIGS = {
    "configs2_c1_L_f05": dict(clusters=1, mode="f", ratio=0.5, dist="L", seed=3),
    "configs3_c3_M_r10": dict(clusters=3, mode="r", ratio=1.0, dist="M", seed=3),
}


@pytest.mark.parametrize("name", sorted(CONFIGS))
def test_full_size_encode_round_trip_and_oracle_parity(name):
    from qvz_amd import host, kernels as K, pipeline as P
    cfg = CONFIGS[name]
    import gc
    host.release_workspace()                 # the coder's kept record workspace of an earlier test
    gc.collect()
    torch.cuda.empty_cache()                 # blocks an earlier test left in torch's caching allocator count as used otherwise
    free, _ = torch.cuda.mem_get_info()
    if free < 250 * 2 ** 30:
        pytest.skip(f"needs ~250 GB of HBM, {free >> 30} GiB free")
    k, seed = cfg["clusters"], cfg["seed"]
    mode = host.MODE_RATIO if cfg["mode"] == "f" else host.MODE_FIXED
    omode = O.MODE_RATIO if cfg["mode"] == "f" else O.MODE_FIXED
    n = N
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

    # (c) shard 0 and a middle shard, re-encoded by the oracle from CPU-generated reads
    first = host.shard_bounds(n, SHARDS)
    coded = r["coded"]
    cap = r["cap_per_shard"]
    for g in (0, 4097):
        a, b = first[g], first[g + 1]
        rows = O.qgen1_rows(seed, b - a, C, first=a)
        assert np.array_equal(sym[:, a:b].cpu().numpy(), (rows[:, :C].T - 33).astype(np.uint8))
        lab_g = np.ascontiguousarray(labels[a:b].cpu().numpy())
        state = O.WELL_DEBUG_STATE if g == 0 else K.well_jump(O.WELL_DEBUG_STATE, a * C // 4)
        enc = O.encode_stream(rows, lab_g, ocbs, dmat, well_state=state)
        got_u = (r["qv"][:, a:b].cpu().numpy().T + 33).astype(np.uint8)
        assert np.array_equal(got_u, enc["u"][:, :C]), f"shard {g}: quantized rows differ from the oracle's"
        stream = coded[g * cap: g * cap + sizes[g]].cpu().numpy().tobytes()
        assert stream == enc["stream"], f"shard {g}: GPU-coded stream differs from the oracle's"
    del r, sym, coded, cnt, labels
    host.release_workspace()
    gc.collect()
    torch.cuda.empty_cache()

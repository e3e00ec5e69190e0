"""GPU parity tests (run with -m gpu on an MI355X): every HIP kernel behind include/qvz_hip.h against the CPU
oracle on the same seeded inputs, plus size-independent properties at BASELINE.json's full shapes."""
import numpy as np
import pytest

import golden_util as G
import oracle_lib as O

pytestmark = pytest.mark.gpu

torch = pytest.importorskip("torch")
A = 72


@pytest.fixture(scope="module")
def K():
    from qvz_amd import kernels
    return kernels


def rows_to_sym(rows):
    """(n, c+1) text rows -> (c, n) symbols."""
    return np.ascontiguousarray(rows[:, :-1].T - 33).astype(np.uint8)


def upload_sym(K, rows):
    n, c = rows.shape[0], rows.shape[1] - 1
    sym = K.empty_matrix(n, c)
    sym.zero_()
    sym[:, :n] = torch.from_numpy(rows_to_sym(rows)).cuda()
    return sym


def tables_from_codebooks(cbs):
    """ct/qs device tables in the layout of include/qvz_hip.h from oracle codebooks (test helper)."""
    k, c = len(cbs), cbs[0].c
    ct = np.zeros((k, c, A), dtype=np.uint16)
    qs = np.zeros((k, c, 2 * A, A), dtype=np.uint16)
    for j, cb in enumerate(cbs):
        idx = cb.ctx_idx.astype(np.uint16)
        qr = np.take_along_axis(cb.qratio.astype(np.uint16), np.minimum(idx, A - 1).astype(np.int64), axis=1)
        ct[j] = np.where(idx == 0xFF, 0xFF, idx | (qr << 8))
        q = cb.q.astype(np.int64)
        st = np.take_along_axis(cb.out_idx.astype(np.uint16), q, axis=2)
        qs[j] = cb.q.astype(np.uint16) | (st << 8)
    return torch.from_numpy(ct.view(np.int16)).cuda(), torch.from_numpy(qs.view(np.int16)).cuda()


# ------------------------------------------------------------------------------------------------
@pytest.mark.parametrize("n,c,first", [(5000, 37, 0), (1021, 150, 123457), (4, 2, 7), (70000, 100, 3)])
def test_qgen1_matches_cpu(K, n, c, first):
    sym = K.qgen1(9, n, c, first=first)
    ref = rows_to_sym(O.qgen1_rows(9, n, c, first=first))
    np.testing.assert_array_equal(sym[:, :n].cpu().numpy(), ref)


@pytest.mark.parametrize("n,c", [(1000, 30), (257, 151), (3, 5), (66000, 64)])
def test_transpose_roundtrip(K, n, c):
    rows = O.qgen1_rows(3, n, c)
    text = torch.from_numpy(rows.reshape(-1)).cuda()
    sym, bad = K.transpose_text(text, n, c)
    assert int(bad.item()) == 0
    np.testing.assert_array_equal(sym[:, :n].cpu().numpy(), rows_to_sym(rows))
    back = K.untranspose_text(sym, n)
    np.testing.assert_array_equal(back.cpu().numpy(), rows.reshape(-1))


def test_transpose_flags_bad_bytes(K):
    rows = O.qgen1_rows(3, 500, 20)
    rows[17, 3] = 20          # below '!' : index -13 in the reference (UB)
    rows[400, 19] = 33 + 72   # above the 72-symbol alphabet
    rows[123, 20] = ord("x")  # missing newline
    _, bad = K.transpose_text(torch.from_numpy(rows.reshape(-1)).cuda(), 500, 20)
    assert int(bad.item()) == 3


# ------------------------------------------------------------------------------------------------
@pytest.mark.parametrize("name", ["t2k_c3_f05_M", "twolevel_c2_f05", "cfg1_c3_r10_M"])
def test_kmeans_full_loop_matches_reference_phase(K, name):
    """Host loop of do_kmeans_clustering (cluster.c:212-244) over the three k-means kernels."""
    rows = G.case_rows(name)
    k = G.case_opts(name)["k"]
    ph = G.phase(name)
    n, c = rows.shape[0], rows.shape[1] - 1
    sym = upload_sym(K, rows)
    cent = O.pick_centroids(n, k)
    means = torch.from_numpy(np.stack([rows[int(i), :c] - 33 for i in cent]).astype(np.uint8)).cuda()
    iters = 0
    while True:
        labels, counts, acc = K.kmeans_step(sym, n, means)
        moved, status = K.kmeans_update(acc, counts, means)
        iters += 1
        assert int(status.item()) == 0
        if not (moved[0].item() > 4.0) or iters >= 1000:
            break
    np.testing.assert_array_equal(labels[:n].cpu().numpy(), ph["labels"])
    np.testing.assert_array_equal(counts.cpu().numpy().astype(np.uint32), ph["counts"])
    np.testing.assert_array_equal(means.cpu().numpy() + 33, ph["means"])


@pytest.mark.parametrize("k", [1, 2, 5, 9, 17])
def test_kmeans_single_iteration_random_means(K, k):
    n, c = 12345, 61
    rows = O.qgen1_rows(21, n, c)
    rng = np.random.default_rng(k)
    cent = np.sort(rng.choice(n, size=k, replace=False)).astype(np.uint64)
    sym = upload_sym(K, rows)
    means_np = np.stack([rows[int(i), :c] - 33 for i in cent]).astype(np.uint8)
    means = torch.from_numpy(means_np.copy()).cuda()
    labels, counts = K.kmeans_assign(sym, n, means)
    # oracle: one assignment pass == labels after iteration 1 when threshold is huge
    olab, omeans, ocounts, it = O.kmeans(rows, k, threshold=1e30, centroids=cent)
    if it < 0:
        pytest.skip("empty cluster in this draw")
    np.testing.assert_array_equal(labels[:n].cpu().numpy(), olab)
    np.testing.assert_array_equal(counts.cpu().numpy().astype(np.uint32), ocounts)
    acc = K.kmeans_accumulate(sym, n, labels, k)
    moved, status = K.kmeans_update(acc, counts, means)
    assert int(status.item()) == 0
    np.testing.assert_array_equal(means.cpu().numpy() + 33, omeans)
    ref_acc = np.zeros((k, c), dtype=np.int64)
    np.add.at(ref_acc, olab, (rows[:, :c] - 33).astype(np.int64))
    np.testing.assert_array_equal(acc.cpu().numpy(), ref_acc)


@pytest.mark.parametrize("k,n,c,seed", [(3, 12345, 150, 5), (2, 70001, 100, 6), (4, 4099, 160, 7), (3, 255, 37, 8), (3, 256, 4, 9),
                                         (2, 17, 1, 10), (3, 1_000_003, 36, 11), (4, 300_000, 151, 12), (3, 5000, 161, 13),
                                         (5, 20_000, 150, 14), (7, 9_999, 33, 15), (8, 250_000, 160, 16), (9, 5000, 40, 17)])
def test_kmeans_one_pass_step_equals_the_two_pass_kernels(K, k, n, c, seed):
    """qvz_hip_kmeans_step (tile in LDS: distances by byte dot products, sums by an i8 MFMA against the one-hot labels)
    == qvz_hip_kmeans_assign + qvz_hip_kmeans_accumulate == numpy, bit for bit: ragged tails (n not a multiple of 16 or 256),
    fewer reads than a tile, more tiles than the persistent grid holds at once (1e6 reads), 151 and 160 columns (partial and
    full last column tile), 161 columns and 9 clusters (the two-pass fallback), 5 .. 8 clusters (two 16-byte centroid reads per
    column quad, sums in the second row group of the MFMA result), column counts that are not multiples of 4."""
    rows = O.qgen1_rows(seed, n, c)
    rng = np.random.default_rng(seed)
    cent = np.sort(rng.choice(n, size=k, replace=False))
    sym = upload_sym(K, rows)
    x = (rows[:, :c] - 33).astype(np.int64)
    means_np = x[cent].astype(np.uint8)
    means = torch.from_numpy(means_np.copy()).cuda()
    labels, counts, acc = K.kmeans_step(sym, n, means)
    labels2, counts2 = K.kmeans_assign(sym, n, means)
    acc2 = K.kmeans_accumulate(sym, n, labels2, k)
    d = np.stack([((x - means_np[j].astype(np.int64)) ** 2).sum(axis=1) for j in range(k)], axis=1)
    lab = d.argmin(axis=1)
    ref_acc = np.stack([x[lab == j].sum(axis=0) for j in range(k)])
    np.testing.assert_array_equal(labels[:n].cpu().numpy(), lab)
    np.testing.assert_array_equal(counts.cpu().numpy(), np.bincount(lab, minlength=k))
    np.testing.assert_array_equal(acc.cpu().numpy(), ref_acc)
    assert torch.equal(labels[:n], labels2[:n]) and torch.equal(counts, counts2) and torch.equal(acc, acc2)
    # += semantics of counts and sums (the drivers zero them per iteration)
    K.kmeans_step(sym, n, means, labels=labels, counts=counts, acc=acc)
    assert torch.equal(counts, 2 * counts2) and torch.equal(acc, 2 * acc2)


def test_kmeans_one_pass_step_with_ties_and_extreme_symbols(K):
    """Equal distances go to the lowest cluster id (cluster.c:155-163) in the dot-product form too, and symbols up to 127
    (far beyond the 72-symbol alphabet) stay exact in the signed i8 operand."""
    n, c, k = 3000, 24, 4
    rng = np.random.default_rng(3)
    x = rng.integers(0, 128, size=(n, c), dtype=np.int64)
    x[::7] = 127
    means_np = np.stack([x[1], x[1], x[2], x[2]]).astype(np.uint8)        # duplicate centroids: every read ties between two
    rows = np.full((n, c + 1), 10, dtype=np.uint8)
    rows[:, :c] = x + 33
    sym = upload_sym(K, rows)
    labels, counts, acc = K.kmeans_step(sym, n, torch.from_numpy(means_np.copy()).cuda())
    d = ((x[:, None, :] - means_np[None].astype(np.int64)) ** 2).sum(axis=2)
    lab = d.argmin(axis=1)
    assert set(np.unique(lab)) <= {0, 2}
    ref_acc = np.zeros((k, c), dtype=np.int64)
    np.add.at(ref_acc, lab, x)
    np.testing.assert_array_equal(labels[:n].cpu().numpy(), lab)
    np.testing.assert_array_equal(acc.cpu().numpy(), ref_acc)


def test_kmeans_empty_cluster_is_reported(K):
    rows = O.qgen1_rows(1, 300, 10)
    sym = upload_sym(K, rows)
    means = torch.from_numpy(np.stack([rows[0, :10] - 33, np.full(10, 71)]).astype(np.uint8)).cuda()
    labels, counts = K.kmeans_assign(sym, 300, means)
    acc = K.kmeans_accumulate(sym, 300, labels, 2)
    _, status = K.kmeans_update(acc, counts, means)
    assert int(counts[1].item()) == 0 and int(status.item()) == 1


# ------------------------------------------------------------------------------------------------
@pytest.mark.parametrize("n,c,k", [(2000, 30, 1), (10000, 100, 1), (4097, 7, 3), (30000, 40, 9), (5, 7, 1)])
def test_pmf_counts_match_oracle(K, n, c, k):
    rows = O.qgen1_rows(5, n, c)
    rng = np.random.default_rng(n)
    labels_np = rng.integers(0, k, size=n).astype(np.uint8)
    sym = upload_sym(K, rows)
    lab = torch.zeros(sym.shape[1], dtype=torch.uint8, device="cuda")
    lab[:n] = torch.from_numpy(labels_np).cuda()
    cnt = K.pmf_hist(sym, n, labels=lab if k > 1 else None, k=k)
    ref = O.stats(rows, labels_np if k > 1 else np.zeros(n, np.uint8), k)
    np.testing.assert_array_equal(cnt.cpu().numpy().astype(np.uint32), ref)


def test_pmf_full_alphabet_and_fixture(K):
    rows = G.case_rows("uniform72_r20_A")
    n = rows.shape[0]
    cnt = K.pmf_hist(upload_sym(K, rows), n)
    np.testing.assert_array_equal(cnt.cpu().numpy().astype(np.uint32), O.stats(rows, np.zeros(n, np.uint8), 1))
    ph = G.phase("t2k_r10_M")
    rows = G.case_rows("t2k_r10_M")
    cnt = K.pmf_hist(upload_sym(K, rows), rows.shape[0])
    np.testing.assert_array_equal(cnt.cpu().numpy().astype(np.uint32), ph["cnt"])


@pytest.mark.parametrize("aa", [0, 20, 50, 72])
def test_pmf_kernel_variants_agree(K, aa, monkeypatch):
    """The two PMF kernels (two reads per symbol; column walk) count the same pairs, with a tile hint smaller than the
    alphabet (symbols beyond it take the global-atomic path) and a ragged last group."""
    n, c = 70001, 12
    rows = O.qgen1_rows(11, n, c)
    ref = O.stats(rows, np.zeros(n, np.uint8), 1)
    sym = upload_sym(K, rows)
    for algo, q in (("0", "0"), ("1", "0"), ("1", "4"), ("1", "6")):
        # test hook: the kernel and its 16-byte groups per thread (0 = by size) that full-size inputs select, and the tile's side
        monkeypatch.setenv("QVZ_PMF", f"aa={aa},algo={algo},q={q}")
        cnt = K.pmf_hist(sym, n)
        np.testing.assert_array_equal(cnt.cpu().numpy().astype(np.uint32), ref, err_msg=f"algo={algo} Q={q}")


@pytest.mark.parametrize("k,aa,q", [(2, 0, 0), (3, 0, 4), (3, 30, 6), (4, 0, 6), (4, 72, 2), (5, 0, 4), (9, 40, 2)])
def test_pmf_with_labels_counts_like_the_oracle(K, k, aa, q, monkeypatch):
    """Up to four clusters the walking kernel packs a thread's labels to 2 bits each and keeps them in registers over all columns;
    more clusters: the label bytes, asked for again in every column (nine: two passes of the tiles). Same counters as the oracle's
    calculate_statistics restatement (codebook.c:193-205), with a ragged last group, a tile hint below the alphabet, and a
    caller's label vector that holds values outside the clusters (those reads count nowhere)."""
    n, c = 70001, 14
    rows = O.qgen1_rows(21, n, c)
    rng = np.random.default_rng(k * 1000 + aa)
    labels_np = rng.integers(0, k, size=n).astype(np.uint8)
    sym = upload_sym(K, rows)
    lab = torch.zeros(sym.shape[1], dtype=torch.uint8, device="cuda")
    lab[:n] = torch.from_numpy(labels_np).cuda()
    ref = O.stats(rows, labels_np, k)
    monkeypatch.setenv("QVZ_PMF", f"algo=1,aa={aa},q={q}")
    packed = K.pmf_hist(sym, n, labels=lab, k=k)
    np.testing.assert_array_equal(packed.cpu().numpy().astype(np.uint32), ref)
    # a label vector with strangers in it: cluster ids k .. 255 in 1 % of the reads (also 3 -> the 2-bit field's own value when k = 3)
    odd = labels_np.copy()
    where = rng.choice(n, size=n // 100, replace=False)
    odd[where] = rng.integers(k, 256, size=where.size).astype(np.uint8)
    odd[where[:50]] = k
    lab[:n] = torch.from_numpy(odd).cuda()
    packed = K.pmf_hist(sym, n, labels=lab, k=k)
    keep = odd < k
    ref_odd = O.stats(np.ascontiguousarray(rows[keep]), np.ascontiguousarray(odd[keep]), k)
    np.testing.assert_array_equal(packed.cpu().numpy().astype(np.uint32), ref_odd)


def test_pmf_accumulates_across_calls(K):
    """Counters are += : two shards of reads give the single-pass counts (the multi-GPU reduction relies on it)."""
    rows = O.qgen1_rows(8, 9000, 25)
    full = K.pmf_hist(upload_sym(K, rows), 9000)
    part = K.pmf_hist(upload_sym(K, rows[:5000]), 5000)
    part = K.pmf_hist(upload_sym(K, rows[5000:]), 4000, cnt=part)
    assert torch.equal(full, part)


# ------------------------------------------------------------------------------------------------
def cpu_well_words(state, first, count):
    import ctypes as C

    class W(C.Structure):
        _fields_ = [("state", C.c_uint32 * 32), ("n", C.c_uint32), ("bit_output", C.c_uint32), ("bits_left", C.c_uint32)]

    lib = O.lib()
    lib.orc_well_word.restype = C.c_uint32
    w = W()
    for i in range(32):
        w.state[i] = int(state[i])
    out = np.zeros(count, dtype=np.uint32)
    for i in range(first + count):
        v = lib.orc_well_word(C.byref(w))
        if i >= first:
            out[i - first] = v
    return out


@pytest.mark.parametrize("first,count", [(0, 4), (0, 40000), (5, 100), (8191, 3), (8192, 8193), (123457, 20001)])
def test_well_words_match_reference_recurrence(K, first, count):
    state = O.WELL_DEBUG_STATE
    words = K.well_words(state, first, count).cpu().numpy().view(np.uint32)
    np.testing.assert_array_equal(words, cpu_well_words(state, first, count))
    if first == 0 and count == 4:
        assert [hex(int(x)) for x in words] == G.MANIFEST["well_kat"]["first_words"]


def test_well_random_state_and_jump(K):
    rng = np.random.default_rng(77)
    state = rng.integers(0, 2 ** 31, size=32, dtype=np.uint32)   # rand() values, qv_stream.c:81
    words = K.well_words(state, 1000, 3000).cpu().numpy().view(np.uint32)
    np.testing.assert_array_equal(words, cpu_well_words(state, 1000, 3000))
    # a jumped state, stored with n = 0, continues the same stream (what a shard's .q header holds)
    jumped = K.well_jump(state, 2500)
    np.testing.assert_array_equal(cpu_well_words(jumped, 0, 64), cpu_well_words(state, 2500, 64))
    # far jump: consistency of two routes
    a = K.well_jump(K.well_jump(state, 10 ** 9), 3 * 10 ** 9 + 17)
    b = K.well_jump(state, 4 * 10 ** 9 + 17)
    np.testing.assert_array_equal(a, b)


def test_well_jump_ahead_matches_the_reference_generator_beyond_2_to_the_32(K):
    """BASELINE configs[4] draws 3.75e10 words; rank 7 of 8 starts 32 812 500 000 words in. The GF(2) jump-ahead
    (qvz_hip_well_jump, qvz_hip_well_jump_many) against states of the reference's own generator stepped that far
    (tests/golden/well_far.json): bits 32-35 of the word offset are pinned by a reference-derived vector, not by a second
    route through the same code."""
    import ctypes as C
    from qvz_amd._lib import lib
    far = G.well_far()
    for words, state, nxt in far:
        got = K.well_jump(O.WELL_DEBUG_STATE, words)
        np.testing.assert_array_equal(got, state, err_msg=f"jump by {words} words")
        assert K.well_words(got, 0, 4).cpu().numpy().view(np.uint32).tolist() == nxt
    # all of them in one batched call, and a relative jump from one golden state to the next
    w = (C.c_uint64 * len(far))(*[f[0] for f in far])
    out = np.zeros((len(far), 32), dtype=np.uint32)
    st = np.ascontiguousarray(O.WELL_DEBUG_STATE)
    assert lib().qvz_hip_well_jump_many(st.ctypes.data_as(C.POINTER(C.c_uint32)), w, len(far), out.ctypes.data_as(C.POINTER(C.c_uint32))) == 0
    np.testing.assert_array_equal(out, np.stack([f[1] for f in far]))
    np.testing.assert_array_equal(K.well_jump(far[3][1], far[4][0] - far[3][0]), far[4][1])
    # draws of reads far into a 1B-read file: the word stream behind first_read * C draws
    first_read, c, n = 875_000_000, 150, 64
    draws = K.well_draws(O.WELL_DEBUG_STATE, first_read, n, c)
    np.testing.assert_array_equal(draws[:, :n].cpu().numpy(), cpu_draws(far[4][1], 0, n, c))


def cpu_draws(state, first_read, n, c):
    """(c, n) matrix of 7-bit draws as the reference consumes them: draw g = read*c + column (well.c:33-46)."""
    g0 = first_read * c
    words = cpu_well_words(state, g0 // 4, (g0 + n * c + 3) // 4 - g0 // 4)
    g = g0 + np.arange(n * c, dtype=np.int64)
    d = (words[g // 4 - g0 // 4] >> (7 * (g % 4)).astype(np.uint32)) & 127
    return d.reshape(n, c).T.astype(np.uint8)


@pytest.mark.parametrize("first,n,c", [(0, 1000, 30), (0, 5, 7), (2501, 3498, 33), (123456, 70001, 150), (7, 300, 1022),
                                       (4_194_301, 10, 2)])
def test_well_draw_matrix_matches_reference_draw_order(K, first, n, c):
    draws = K.well_draws(O.WELL_DEBUG_STATE, first, n, c)
    np.testing.assert_array_equal(draws[:, :n].cpu().numpy(), cpu_draws(O.WELL_DEBUG_STATE, first, n, c))
    if first == 0 and n >= 1000:
        assert draws[:12, 0].cpu().tolist() == G.MANIFEST["well_kat"]["first_draws"]      # read 0, columns 0..11


# ------------------------------------------------------------------------------------------------
@pytest.mark.parametrize("name", ["t2k_r10_M", "t2k_f05_L", "t2k_c3_f05_M", "uniform72_r20_A", "c2_f08",
                                  "tiny5_r10", "const_f05", "r35_high_rate", "cfg1_r10_M"])
def test_quantize_matches_reference_u_file(K, name):
    """qv output == the reference's -u dump; coder feed and per-read error == oracle."""
    rows = G.case_rows(name)
    opts = G.case_opts(name)
    n, c, k = rows.shape[0], rows.shape[1] - 1, opts["k"]
    labels_np = O.kmeans(rows, k)[0]
    cnt = O.stats(rows, labels_np, k)
    d = O.distortion(opts["dist"])
    cbs = [O.generate_codebook(cnt[j], d, opts["mode"], opts["ratio"]) for j in range(k)]
    ref = O.encode_stream(rows, labels_np, cbs, d, want_feed=True)
    assert G.md5(ref["u"].tobytes()) == G.CASES[name]["u_md5"]

    sym = upload_sym(K, rows)
    lab = torch.zeros(sym.shape[1], dtype=torch.uint8, device="cuda")
    lab[:n] = torch.from_numpy(labels_np).cuda()
    ct, qs = tables_from_codebooks(cbs)
    draws = K.well_draws(O.WELL_DEBUG_STATE, 0, n, c)
    out = K.quantize(sym, n, ct, qs, draws, torch.from_numpy(d).cuda(), labels=lab, k=k)
    assert int(out["status"].item()) == 0
    text = K.untranspose_text(out["qv"], n).cpu().numpy().tobytes()
    assert G.md5(text) == G.CASES[name]["u_md5"], "-u differs from the reference"
    np.testing.assert_array_equal(out["qi"][:, :n].cpu().numpy().T, ref["qi"])
    np.testing.assert_array_equal(out["st"][:, :n].cpu().numpy().T, ref["st"])
    # per-read error is accumulated in the reference's order: bit-identical, so is the sequential total
    err = out["err"].cpu().numpy()
    total = 0.0
    for e in err:
        total += e / float(c)
    assert total / float(n) == ref["distortion"]


def test_quantize_shard_equals_slice_of_full(K):
    """Reads [a,b) quantized alone with the draws of reads [a,b) reproduce the same rows (multi-GPU sharding)."""
    rows = O.qgen1_rows(4, 6000, 33)
    n, c = 6000, 33
    cnt = O.stats(rows, np.zeros(n, np.uint8), 1)
    d = O.distortion("M")
    cb = O.generate_codebook(cnt[0], d, O.MODE_RATIO, 0.6)
    ct, qs = tables_from_codebooks([cb])
    dd = torch.from_numpy(d).cuda()
    state = O.WELL_DEBUG_STATE
    full = K.quantize(upload_sym(K, rows), n, ct, qs, K.well_draws(state, 0, n, c), dd)
    a, b = 2501, 5999      # deliberately not multiples of 4
    part = K.quantize(upload_sym(K, rows[a:b]), b - a, ct, qs, K.well_draws(state, a, b - a, c), dd)
    for key in ("qv", "qi", "st"):
        assert torch.equal(part[key][:, : b - a], full[key][:, a:b]), key
    assert torch.equal(part["err"], full["err"][a:b])


def test_quantize_kernels_agree_and_general_distortion_matrix(K, monkeypatch):
    """The LDS-table kernel (distortion = f(|x - y|)) and the L1/L2-gather kernel give the same planes and the same
    per-read error bits; a matrix that is not a function of |x - y| goes through the general kernel and its per-read
    error is the column-ordered sum of d[x + 72 * qv]."""
    n, c = 9001, 29
    rows = O.qgen1_rows(5, n, c)
    cnt = O.stats(rows, np.zeros(n, np.uint8), 1)
    d = O.distortion("L")
    cb = O.generate_codebook(cnt[0], d, O.MODE_RATIO, 0.8)
    ct, qs = tables_from_codebooks([cb])
    sym = upload_sym(K, rows)
    draws = K.well_draws(O.WELL_DEBUG_STATE, 0, n, c)
    a = K.quantize(sym, n, ct, qs, draws, torch.from_numpy(d).cuda())
    monkeypatch.setenv("QVZ_QUANTIZE_KERNEL", "global")
    b = K.quantize(sym, n, ct, qs, draws, torch.from_numpy(d).cuda())
    monkeypatch.delenv("QVZ_QUANTIZE_KERNEL")
    for key in ("qv", "qi", "st"):
        assert torch.equal(a[key][:, :n], b[key][:, :n]), key
    assert torch.equal(a["err"], b["err"])
    rng = np.random.default_rng(7)
    g = np.ascontiguousarray(rng.random((72, 72)))               # no symmetry, no |x - y| structure
    e = K.quantize(sym, n, ct, qs, draws, torch.from_numpy(g.reshape(-1)).cuda())
    assert torch.equal(e["qv"][:, :n], a["qv"][:, :n])
    qv = e["qv"][:, :n].cpu().numpy().astype(np.int64)           # (c, n)
    x = (rows[:, :c].T.astype(np.int64) - 33)
    want = np.zeros(n)
    for s_ in range(c):
        want = g.reshape(-1)[x[s_] + 72 * qv[s_]] if s_ == 0 else want + g.reshape(-1)[x[s_] + 72 * qv[s_]]
    np.testing.assert_array_equal(e["err"].cpu().numpy(), want)


def test_quantize_reports_undefined_context(K):
    rows = O.qgen1_rows(4, 400, 12)
    cnt = O.stats(rows, np.zeros(400, np.uint8), 1)
    d = O.distortion("M")
    cb = O.generate_codebook(cnt[0], d, O.MODE_RATIO, 0.5)
    ct, qs = tables_from_codebooks([cb])
    ct[0, 5, :] = 0xFF   # column 5 defines no context at all
    out = K.quantize(upload_sym(K, rows), 400, ct, qs, K.well_draws(O.WELL_DEBUG_STATE, 0, 400, 12),
                     torch.from_numpy(d).cuda())
    assert int(out["status"].item()) == 1


# ------------------------------------------------------------------------------------------------
# BASELINE.json shapes: properties that do not need the oracle to finish at this size
# ------------------------------------------------------------------------------------------------
@pytest.mark.parametrize("n,c,algo", [(10_000_000, 150, ""), (10_000_000, 150, "1")])
def test_full_size_pmf_properties(K, n, c, algo, monkeypatch):
    if algo:
        monkeypatch.setenv("QVZ_PMF", f"algo={algo},q=6")   # the walking kernel with six groups per thread, as bench.py's 100M reads select it
    sym = K.qgen1(2, n, c)
    cnt = K.pmf_hist(sym, n).to(torch.int64)[0]          # (c, 72, 72)
    per_col = cnt.sum(dim=(1, 2))
    assert torch.all(per_col == n), "every column must count every read exactly once"
    # flow conservation: reads with X_{s-1}=p counted as 'prev' in column s == counted as 'cur' in column s-1
    cur_marg = cnt.sum(dim=1)                            # (c, 72): occurrences of each symbol in column s
    prev_marg = cnt.sum(dim=2)                           # (c, 72): occurrences as previous symbol
    assert torch.equal(prev_marg[1:], cur_marg[:-1])
    # and against an independent device histogram of the raw column
    col = 77
    hist = torch.bincount(sym[col, :n].to(torch.int64), minlength=A)
    assert torch.equal(hist, cur_marg[col])
    # shard additivity at scale (what the RCCL all-reduce sums)
    half = n // 2
    a = K.pmf_hist(K.qgen1(2, half, c), half)
    b = K.pmf_hist(K.qgen1(2, n - half, c, first=half), n - half, cnt=a)
    assert torch.equal(b.to(torch.int64)[0], cnt)


def test_full_size_kmeans_properties(K):
    n, c, k = 10_000_000, 150, 3
    sym = K.qgen1(2, n, c)
    means = sym[:, [11, 5000001, 9999999]].T.contiguous()
    labels, counts = K.kmeans_assign(sym, n, means)
    assert int(counts.sum().item()) == n
    assert torch.equal(torch.bincount(labels[:n].to(torch.int64), minlength=k), counts.to(torch.int64))
    acc = K.kmeans_accumulate(sym, n, labels, k)
    col = 33
    for j in range(k):
        assert int(acc[j, col].item()) == int(sym[col, :n][labels[:n] == j].to(torch.int64).sum().item())
    # idempotence: re-assigning with the same means gives the same labels
    labels2, _ = K.kmeans_assign(sym, n, means)
    assert torch.equal(labels, labels2)

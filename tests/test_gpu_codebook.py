"""GPU parity of codebook generation (Lloyd-Max, rate search, context recursion) against the CPU oracle and the
reference's own codebook bytes. Bit-exact: the codebook decides the bitstream."""
import numpy as np
import pytest

import golden_util as G
import oracle_lib as O

pytestmark = pytest.mark.gpu
torch = pytest.importorskip("torch")


@pytest.fixture(scope="module")
def K():
    from qvz_amd import kernels
    return kernels


def oracle_codebooks(name):
    rows = G.case_rows(name)
    opts = G.case_opts(name)
    labels = O.kmeans(rows, opts["k"])[0]
    cnt = O.stats(rows, labels, opts["k"])
    d = O.distortion(opts["dist"])
    return rows, opts, cnt, d, [O.generate_codebook(cnt[j], d, opts["mode"], opts["ratio"]) for j in range(opts["k"])]


def assert_same_codebook(cb, ref):
    assert cb.serialize() == ref.serialize(), "serialised codebook differs"
    np.testing.assert_array_equal(cb.nctx, ref.nctx)
    for s in range(cb.c):
        n = int(ref.nctx[s])
        assert cb.ratio[s, :n].tobytes() == ref.ratio[s, :n].tobytes(), f"ratio bits differ in column {s}"
        np.testing.assert_array_equal(cb.ctx_sym[s, :n], ref.ctx_sym[s, :n])
        np.testing.assert_array_equal(cb.ctx_idx[s], ref.ctx_idx[s])
        np.testing.assert_array_equal(cb.nstates[s, : 2 * n], ref.nstates[s, : 2 * n])
        np.testing.assert_array_equal(cb.recon[s, : 2 * n], ref.recon[s, : 2 * n])
        np.testing.assert_array_equal(cb.out_idx[s, : 2 * n], ref.out_idx[s, : 2 * n])


def test_distortion_matrices_bit_identical(K):
    for kind in "MLA":
        assert K.distortion_matrix(kind).tobytes() == O.distortion(kind).tobytes()


@pytest.mark.parametrize("name", sorted(G.CASES))
def test_codebook_matches_oracle_and_reference_bytes(K, name):
    rows, opts, cnt, d, refs = oracle_codebooks(name)
    n, c = rows.shape[0], rows.shape[1] - 1
    blob = K.write_header(opts["k"], c, n)
    for j in range(opts["k"]):
        cb, deferred = K.generate_codebook(torch.from_numpy(cnt[j].view(np.int32)).cuda(), K.distortion_matrix(opts["dist"]),
                                           opts["mode"], opts["ratio"])
        assert_same_codebook(cb, refs[j])
        blob += cb.serialize()
    if G.CASES[name]["stored"]:
        q = G.gz(name + ".q.gz")
        assert q[: len(blob)] == blob, "codebook section differs from the reference's .q file"


@pytest.mark.parametrize("name", ["t2k_r10_M", "t2k_f05_L", "uniform72_f09_M", "r35_high_rate"])
def test_host_decides_everything_with_infinite_guard(K, name, monkeypatch):
    """Rare-branch test: with a huge guard band every search decision is deferred to host libm and the search is
    resumed state by state; the codebook must not change. (QVZ_CB_LOG2=host: the hand-over in which the device's entropies
    only steer - what runs when the device's log2 has not been verified against this host's.)"""
    monkeypatch.setenv("QVZ_CB_LOG2", "host")
    rows, opts, cnt, d, refs = oracle_codebooks(name)
    cb, deferred = K.generate_codebook(torch.from_numpy(cnt[0].view(np.int32)).cuda(), d, opts["mode"], opts["ratio"],
                                       guard=1e30)
    assert deferred > 0
    assert_same_codebook(cb, refs[0])
    cb2, deferred2 = K.generate_codebook(torch.from_numpy(cnt[0].view(np.int32)).cuda(), d, opts["mode"], opts["ratio"])
    assert deferred2 <= deferred
    assert_same_codebook(cb2, refs[0])


@pytest.mark.parametrize("name", sorted(G.CASES))
def test_device_entropies_decide_like_the_hosts_libm(K, name, monkeypatch):
    """The device evaluates the reference's entropies itself - glibc's log2 restated (qvz_amd/csrc/qvz_log2_glibc.h), checked
    against this host's log2 on 2^20 arguments when the process first generates a codebook - and takes every
    `entropy < target` decision and every ratio from them; with QVZ_CB_LOG2=host the host's libm does, as before. Same
    codebooks either way, equal to the oracle's, and the check must have PASSED on this box (otherwise the fast path would
    silently be off)."""
    from qvz_amd._lib import lib
    assert lib().qvz_hip_codebook_exact_entropies() == 1, "the device's log2 does not reproduce this host's libm"
    rows, opts, cnt, d, refs = oracle_codebooks(name)
    for j in range(opts["k"]):
        dev_cnt = torch.from_numpy(cnt[j].view(np.int32)).cuda()
        # (1) the default: the host's column-by-column loop over the device's exact entropies
        a, deferred = K.generate_codebook(dev_cnt, d, opts["mode"], opts["ratio"])
        assert deferred == 0                                      # nothing is deferred to the host when the device's values are exact
        assert_same_codebook(a, refs[j])
        # (2) the band inside which the host's libm re-decides (1e-12 by default: about one decision in 1e12) made huge: every
        # decision goes through the re-check, the libm's targets and entropies must equal the device's bit for bit, same book
        import ctypes
        r0, m0 = ctypes.c_ulonglong(), ctypes.c_ulonglong()
        lib().qvz_hip_codebook_exact_recheck_counts(ctypes.byref(r0), ctypes.byref(m0))
        monkeypatch.setenv("QVZ_CB_LOG2", "band=1e30")
        e, deferred = K.generate_codebook(dev_cnt, d, opts["mode"], opts["ratio"])
        monkeypatch.delenv("QVZ_CB_LOG2")
        r1, m1 = ctypes.c_ulonglong(), ctypes.c_ulonglong()
        lib().qvz_hip_codebook_exact_recheck_counts(ctypes.byref(r1), ctypes.byref(m1))
        assert_same_codebook(e, a)
        assert m1.value == m0.value, "the host's libm disagrees with the device's entropies by a bit"
        assert r1.value >= r0.value                               # (a book whose every context is dead takes no decision at all)
        # (3) ... and with the host's libm taking every decision (what runs when the check fails)
        monkeypatch.setenv("QVZ_CB_LOG2", "host")
        c_, _ = K.generate_codebook(dev_cnt, d, opts["mode"], opts["ratio"])
        monkeypatch.delenv("QVZ_CB_LOG2")
        assert_same_codebook(c_, a)


@pytest.mark.parametrize("name,how", [("r35_high_rate", ""), ("r35_high_rate", "host"), ("cfg1_c3_r10_M", ""), ("cfg1_c3_r10_M", "host")])
def test_walks_known_in_advance_give_the_oracles_codebook(K, name, how, monkeypatch, capfd):
    """A row whose own entropy lies below its target (a -r rate no quantizer of that context can reach): every step of the
    reference's walk (codebook.c:244-258) says "below" and it ends at the alphabet's size; the GPU path makes the designs A - 1
    and A only. The oracle walks all 72 state counts as the reference does: same bytes, with the device's entropies and with
    the host's libm, and the short cut must actually have been taken on these inputs."""
    import re
    monkeypatch.setenv("QVZ_CB_DEBUG", "1")
    if how:
        monkeypatch.setenv("QVZ_CB_LOG2", how)
    rows, opts, cnt, d, refs = oracle_codebooks(name)
    taken = 0
    for j in range(opts["k"]):
        capfd.readouterr()
        cb, _ = K.generate_codebook(torch.from_numpy(cnt[j].view(np.int32)).cuda(), d, opts["mode"], opts["ratio"])
        err = capfd.readouterr().err
        assert_same_codebook(cb, refs[j])
        m = re.search(r"walks known in advance \(row entropy below the target\): (\d+) contexts", err)
        assert m, err[-400:]
        taken += int(m.group(1))
    assert taken > 0


@pytest.mark.parametrize("mode,target", [("ratio", 1.25), ("fixed", 5.5), ("fixed", 0.999999999)])
def test_targets_no_row_can_reach(K, mode, target, monkeypatch, capfd):
    """-f above 1 (every row's target lies above its own entropy), a -r rate beyond log2 of the alphabet in use, and a rate a
    hair below 1 bit: the oracle walks state count by state count as the reference does (codebook.c:244-258), the GPU path
    skips the walks it can prove; same codebook bytes."""
    import re
    monkeypatch.setenv("QVZ_CB_DEBUG", "1")
    full = G.case_rows("t2k_r10_M")
    rows = np.ascontiguousarray(np.concatenate([full[:, :12], full[:, -1:]], axis=1))   # twelve columns of the fixture's reads (+ the newline)
    cnt = O.stats(rows, np.zeros(rows.shape[0], np.uint8), 1)
    d = O.distortion("M")
    m = O.MODE_RATIO if mode == "ratio" else O.MODE_FIXED
    ref = O.generate_codebook(cnt[0], d, m, target)
    capfd.readouterr()
    cb, _ = K.generate_codebook(torch.from_numpy(cnt[0].view(np.int32)).cuda(), d, m, target)
    err = capfd.readouterr().err
    assert_same_codebook(cb, ref)
    taken = int(re.search(r"walks known in advance \(row entropy below the target\): (\d+) contexts", err).group(1))
    assert taken > 0 or target < 1.0


def test_device_tables_match_python_builder(K):
    from test_gpu_kernels import tables_from_codebooks
    rows, opts, cnt, d, refs = oracle_codebooks("t2k_c3_f05_M")
    cbs = [K.generate_codebook(torch.from_numpy(cnt[j].view(np.int32)).cuda(), d, opts["mode"], opts["ratio"])[0]
           for j in range(opts["k"])]
    ct, qs = K.upload_tables(cbs)
    rct, rqs = tables_from_codebooks(refs)
    # only defined entries are compared: contexts beyond nctx are never addressed
    assert torch.equal(ct, rct)
    for j, ref in enumerate(refs):
        for s in range(ref.c):
            n = 2 * int(ref.nctx[s])
            assert torch.equal(qs[j, s, :n], rqs[j, s, :n])


def test_end_to_end_device_pipeline_reproduces_u_file(K):
    """PMF counters -> codebook -> tables -> WELL -> quantize, all on the device, against the reference -u dump."""
    name = "cfg1_r10_M"
    rows = G.case_rows(name)
    opts = G.case_opts(name)
    n, c = rows.shape[0], rows.shape[1] - 1
    sym = K.qgen1(1, n, c)
    cnt = K.pmf_hist(sym, n)
    dist = K.distortion_matrix(opts["dist"])
    cb, _ = K.generate_codebook(cnt[0], dist, opts["mode"], opts["ratio"])
    ct, qs = K.upload_tables([cb])
    draws = K.well_draws(O.WELL_DEBUG_STATE, 0, n, c)
    out = K.quantize(sym, n, ct, qs, draws, torch.from_numpy(dist).cuda())
    assert int(out["status"].item()) == 0
    text = K.untranspose_text(out["qv"], n).cpu().numpy().tobytes()
    assert G.md5(text) == G.CASES[name]["u_md5"]
    q = G.gz(name + ".q.gz")
    blob = K.write_header(1, c, n) + cb.serialize()
    assert q[: len(blob)] == blob


def test_wide_counters_for_files_beyond_the_u32_limit():
    """More than 2^32 - 1 reads: the reference's u32 counts and totals wrap (include/pmf.h:35-36,
    src/codebook.c:482), so a file that large is counted in parts (< 2^32 reads each) whose counters are widened to 64 bits
    and summed (host/qvz_multi.c). The 64-bit route must give the u32 route's codebook wherever both apply, and - counts
    scaled by 2^22, totals beyond 2^32 - the very same codebook again (every probability is the same double)."""
    import ctypes as C
    from qvz_amd import kernels as K
    from qvz_amd._lib import check, lib
    from test_gpu_kernels import upload_sym
    rows = G.case_rows("t2k_f05_L")
    n, c = rows.shape[0], rows.shape[1] - 1
    cnt = K.pmf_hist(upload_sym(K, rows), n)[0]
    d = O.distortion("L")
    dp = d.ctypes.data_as(C.POINTER(C.c_double))
    narrow, _ = K.generate_codebook(cnt, d, O.MODE_RATIO, 0.5)
    wide = torch.zeros(cnt.numel(), dtype=torch.int64, device="cuda")
    check(lib().qvz_hip_widen_u32(cnt.data_ptr(), wide.data_ptr(), cnt.numel(), None))
    assert torch.equal(wide, cnt.to(torch.int64).view(-1))
    for scale in (1, 1 << 22):
        big = (wide * scale).contiguous()
        if scale > 1:
            assert int(big.view(c, -1).sum(dim=1).max().item()) > 2 ** 32
        cb = K.Codebook(c)
        check(lib().qvz_hip_generate_codebook64(big.data_ptr(), c, dp, O.MODE_RATIO, 0.5, 0.0, cb.handle, None))
        assert cb.serialize() == narrow.serialize()
    # recalculate_means with 64-bit counts (cluster.c:104-128): 3 * 2^32 reads in a cluster
    k = 2
    counts = torch.tensor([3 << 32, 5], dtype=torch.int64, device="cuda")
    means = torch.tensor([[10, 20, 30], [1, 2, 3]], dtype=torch.uint8, device="cuda")
    acc = torch.tensor([[(3 << 32) * 12 + 7, (3 << 32) * 20, (3 << 32) * 29 + (3 << 32) - 1], [5 * 4, 5 * 2 + 4, 0]], dtype=torch.int64, device="cuda")
    moved = torch.zeros(k + 1, dtype=torch.float64, device="cuda")
    status = torch.zeros(1, dtype=torch.int32, device="cuda")
    check(lib().qvz_hip_kmeans_update64(acc.data_ptr(), counts.data_ptr(), k, 3, means.data_ptr(), moved.data_ptr(), status.data_ptr(), None))
    torch.cuda.synchronize()
    assert means.cpu().tolist() == [[12, 20, 29], [4, 2, 0]] and int(status.item()) == 0
    assert moved.cpu().tolist() == [18.0, 5.0, 18.0]                            # [max, cluster 0: 4 + 0 + 1, cluster 1: 9 + 0 + 9]

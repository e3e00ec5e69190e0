"""Pins the CPU oracle (oracle/qvz_oracle.c) against the reference's own outputs (tests/golden/)."""
import numpy as np
import pytest

import golden_util as G
import oracle_lib as O

FAST = [n for n, e in G.CASES.items() if e["input"]["n"] * e["input"]["c"] <= 120000]
SLOW = [n for n in G.CASES if n not in FAST]


@pytest.mark.parametrize("name", FAST + SLOW)
def test_encode_byte_identical(name):
    e = G.CASES[name]
    rows = G.case_rows(name)
    r = O.encode(rows, **G.case_opts(name))
    assert len(r["q"]) == e["q_size"]
    assert G.md5(r["q"]) == e["q_md5"], ".q differs from the reference"
    assert G.md5(r["u"]) == e["u_md5"], "-u differs from the reference"
    assert r["size"] == e["size"]
    assert f"{r['rate']:.4f}" == f"{e['rate']:.4f}"
    assert f"{r['distortion']:.4f}" == f"{e['distortion']:.4f}"
    if e["stored"]:
        assert r["q"] == G.gz(name + ".q.gz")
        assert r["u"] == G.gz(name + ".u.gz")


@pytest.mark.parametrize("name", sorted(G.SWEEPS))
def test_sweep_points_byte_identical(name):
    """The reference's rate-distortion sweep (generate_rd.sh:1-17; `-c 1 -f 0 ... .95`, and its commented `-c 3 -r 0 ... 1.90`): at
    every operating point the oracle writes the reference's .q and -u and prints its -s fields."""
    e = G.SWEEPS[name]
    rows = G.sweep_rows(name)
    k, mode, ratios = G.sweep_opts(name)
    assert len(ratios) == 20
    for pt, ratio in zip(e["points"], ratios):
        r = O.encode(rows, k=k, mode=mode, ratio=ratio, dist="M")
        assert (len(r["q"]), G.md5(r["q"]), G.md5(r["u"]), r["size"]) == (pt["q_size"], pt["q_md5"], pt["u_md5"], pt["size"]), pt["comp"]
        assert (f"{r['rate']:.4f}", f"{r['distortion']:.4f}") == (pt["rate"], pt["distortion"]), pt["comp"]


@pytest.mark.parametrize("name", [n for n, e in G.CASES.items() if e["stored"]])
def test_decode_matches_u(name):
    """The invariant the reference's own test.sh checks: decode(encode(x)) == the -u dump."""
    assert O.decode(G.gz(name + ".q.gz")) == G.gz(name + ".u.gz")


@pytest.mark.parametrize("name", [n for n, e in G.CASES.items() if e["phase"]])
def test_phase_intermediates(name):
    ph = G.phase(name)
    rows = G.case_rows(name)
    k = G.case_opts(name)["k"]
    labels, means, counts, iters = O.kmeans(rows, k)
    assert iters > 0
    np.testing.assert_array_equal(labels, ph["labels"])
    np.testing.assert_array_equal(means, ph["means"])
    np.testing.assert_array_equal(counts, ph["counts"])
    cnt = O.stats(rows, labels, k)
    np.testing.assert_array_equal(cnt, ph["cnt"])
    for j in range(k):
        m = O.marginals(cnt[j])
        assert m.tobytes() == ph["marg"][j].tobytes(), "marginals not bit-identical"


def test_well_known_answer():
    import ctypes as C
    kat = G.MANIFEST["well_kat"]

    class W(C.Structure):
        _fields_ = [("state", C.c_uint32 * 32), ("n", C.c_uint32), ("bit_output", C.c_uint32), ("bits_left", C.c_uint32)]

    lib = O.lib()
    lib.orc_well_word.restype = C.c_uint32
    lib.orc_well_bits.restype = C.c_uint32
    w = W()
    for i in range(32):
        w.state[i] = 0x55555555
    words = [lib.orc_well_word(C.byref(w)) for _ in range(4)]
    assert words == [int(x, 16) for x in kat["first_words"]]
    w = W()
    for i in range(32):
        w.state[i] = 0x55555555
    draws = [lib.orc_well_bits(C.byref(w), 7) for _ in range(12)]
    assert draws == kat["first_draws"]


def test_well_states_beyond_two_to_the_32_words():
    """tests/golden/well_far.json holds states of the REFERENCE's generator after up to 3.75e10 words (stepped once in the
    build container). The oracle's generator continues each of them correctly: its next words are the recorded ones, and
    stepping it from the vector at 2^32 - 1 words for 8 words lands on the vector at 2^32 + 7. (The jump-ahead of the product
    is tested against the same vectors in test_gpu_kernels.py.)"""
    far = G.well_far()
    assert [w for w, _, _ in far][:4] == [1000, (1 << 32) - 1, (1 << 32) + 7, (1 << 33) + 5] and far[-1][0] > 3.7e10
    assert np.array_equal(O.well_states_at([1000])[0], far[0][1])               # from the seed by stepping
    for words, state, nxt in far:
        after = O.well_states_at([1, 2, 3, 4], state0=state)
        # the word a call returns is the state word the new n points at = element 0 of the n = 0 form
        assert [int(a[0]) for a in after] == nxt, words
    assert np.array_equal(O.well_states_at([8], state0=far[1][1])[0], far[2][1])

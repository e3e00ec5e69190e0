"""Drop-in boundary tests (GPU).

1. oracle/_ref/qvz_dropin = the reference's UNMODIFIED src/main.c + src/util.c linked against qvz_amd/libqvzcompat.so
   (built by oracle/build_ref.sh where /root/reference exists; the binary travels to the GPU box). Its .q / -u outputs
   must be byte-identical to the reference's own.
2. The compat entry points called one by one through ctypes must leave in the reference's structs what the reference
   leaves there: labels per line, cluster means / counts / accumulators, conditional and marginal pmfs, quantizer lists.
"""
import ctypes as C
import os
import re
import subprocess

import numpy as np
import pytest

import golden_util as G
import oracle_lib as O

pytestmark = pytest.mark.gpu
pytest.importorskip("torch")

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
DROPIN = os.path.join(ROOT, "oracle", "_ref", "qvz_dropin")
COMPAT = os.path.join(ROOT, "qvz_amd", "libqvzcompat.so")


@pytest.mark.skipif(not os.path.exists(DROPIN), reason="qvz_dropin is built only where /root/reference is present")
@pytest.mark.parametrize("name", ["t2k_r10_M", "t2k_f05_L", "t2k_c3_f05_M", "twolevel_c2_f05", "cfg1_r10_M", "uniform72_r20_A",
                                  "t2k_D_asym_f06"])
def test_reference_main_on_compat_library_is_byte_identical(tmp_path, name):
    e = G.MANIFEST["cases"][name]
    inp = tmp_path / "in.txt"
    inp.write_bytes(O.qgen1_rows(e["input"]["seed"], e["input"]["n"], e["input"]["c"]).tobytes() if e.get("custom_dist")
                    else G.case_rows(name).tobytes())
    q, u = str(tmp_path / "d.q"), str(tmp_path / "d.u")
    env = dict(os.environ, QVZ_WELL_SEED="debug")
    out = subprocess.run([DROPIN] + G.cli_flags(e) + ["-s", "-v", "-u", u, str(inp), q], env=env, capture_output=True,
                         text=True, check=True).stdout
    assert G.md5(open(q, "rb").read()) == e["q_md5"]
    assert G.md5(open(u, "rb").read()) == e["u_md5"]
    m = re.search(r"rate, ([\d.]+), distortion, ([\d.]+), time, ([\d.]+), size, (\d+) ", out)
    assert m and m.group(1) == f"{e['rate']:.4f}" and m.group(2) == f"{e['distortion']:.4f}" and int(m.group(4)) == e["size"]
    assert "Clustering took" in out and "Stats and codebook generation took" in out


# ---- ctypes mirrors of the reference structs (include/qvz_compat.h) -------------------------------------------
class Alphabet(C.Structure):
    _fields_ = [("size", C.c_uint32), ("symbols", C.POINTER(C.c_uint8)), ("indexes", C.POINTER(C.c_uint32))]


class Pmf(C.Structure):
    _fields_ = [("pmf_ready", C.c_uint8), ("alphabet", C.POINTER(Alphabet)), ("pmf", C.POINTER(C.c_double)),
                ("counts", C.POINTER(C.c_uint32)), ("total", C.c_uint32)]


class PmfList(C.Structure):
    _fields_ = [("size", C.c_uint32), ("pmfs", C.POINTER(C.POINTER(Pmf)))]


class Distortion(C.Structure):
    _fields_ = [("distortion", C.POINTER(C.c_double)), ("symbols", C.c_uint8)]


class Well(C.Structure):
    _fields_ = [("state", C.c_uint32 * 32), ("n", C.c_uint32), ("bit_output", C.c_uint32), ("bits_left", C.c_uint32)]


class Quantizer(C.Structure):
    _fields_ = [("alphabet", C.POINTER(Alphabet)), ("output_alphabet", C.POINTER(Alphabet)), ("q", C.POINTER(C.c_uint8)),
                ("ratio", C.c_double), ("mse", C.c_double)]


class Options(C.Structure):
    _fields_ = [("verbose", C.c_uint8), ("stats", C.c_uint8), ("mode", C.c_uint8), ("clusters", C.c_uint8),
                ("uncompressed", C.c_uint8), ("distortion", C.c_uint8), ("dist_file", C.c_char_p),
                ("uncompressed_name", C.c_char_p), ("ratio", C.c_double), ("e_dist", C.c_double),
                ("cluster_threshold", C.c_double)]


class CondPmfList(C.Structure):
    _fields_ = [("columns", C.c_uint32), ("alphabet", C.POINTER(Alphabet)), ("pmfs", C.POINTER(C.POINTER(Pmf))),
                ("marginal_pmfs", C.POINTER(PmfList))]


class CondQuantizerList(C.Structure):
    _fields_ = [("columns", C.c_uint32), ("lines", C.c_uint32), ("input_alphabets", C.POINTER(C.POINTER(Alphabet))),
                ("q", C.POINTER(C.POINTER(C.POINTER(Quantizer)))), ("ratio", C.POINTER(C.POINTER(C.c_double))),
                ("qratio", C.POINTER(C.POINTER(C.c_uint8))), ("options", C.POINTER(Options))]


class Line(C.Structure):
    _fields_ = [("cluster", C.c_uint8), ("m_data", C.POINTER(C.c_uint8))]


class LineBlock(C.Structure):
    _fields_ = [("count", C.c_uint32), ("lines", C.POINTER(Line))]


class Cluster(C.Structure):
    _fields_ = [("id", C.c_uint8), ("count", C.c_uint32), ("mean", C.POINTER(C.c_uint8)), ("accumulator", C.POINTER(C.c_uint64)),
                ("training_stats", C.POINTER(CondPmfList)), ("qlist", C.POINTER(CondQuantizerList))]


class ClusterList(C.Structure):
    _fields_ = [("count", C.c_uint8), ("clusters", C.POINTER(Cluster)), ("distances", C.POINTER(C.c_double))]


class QualityFile(C.Structure):
    _fields_ = [("alphabet", C.POINTER(Alphabet)), ("path", C.c_char_p), ("lines", C.c_uint64), ("columns", C.c_uint32),
                ("block_count", C.c_uint32), ("blocks", C.POINTER(LineBlock)), ("cluster_count", C.c_uint8),
                ("clusters", C.POINTER(ClusterList)), ("dist", C.POINTER(Distortion)), ("opts", C.POINTER(Options)),
                ("well", Well)]


def test_struct_sizes_match_the_reference_abi():
    # sizes measured on the compiled reference
    assert C.sizeof(Line) == 16 and C.sizeof(Cluster) == 40 and C.sizeof(QualityFile) == 216
    assert C.sizeof(Options) == 48 and C.sizeof(Pmf) == 40 and C.sizeof(Quantizer) == 40 and C.sizeof(Well) == 140
    assert QualityFile.lines.offset == 16 and QualityFile.columns.offset == 24 and QualityFile.blocks.offset == 32
    assert QualityFile.cluster_count.offset == 40 and QualityFile.clusters.offset == 48 and QualityFile.dist.offset == 56
    assert QualityFile.opts.offset == 64 and QualityFile.well.offset == 72 and Options.ratio.offset == 24


@pytest.mark.parametrize("name", ["t2k_c3_f05_M", "twolevel_c2_f05"])
def test_entry_points_fill_the_reference_structs(tmp_path, name):
    lib = C.CDLL(COMPAT)
    lib.alloc_alphabet.restype = C.POINTER(Alphabet)
    lib.generate_distortion_matrix.restype = C.POINTER(Distortion)
    lib.generate_distortion_matrix.argtypes = [C.c_uint8, C.c_int]
    lib.load_file.restype = C.c_uint32
    lib.load_file.argtypes = [C.c_char_p, C.POINTER(QualityFile), C.c_uint64]
    lib.alloc_cluster_list.restype = C.POINTER(ClusterList)
    for fn in ("do_kmeans_clustering", "calculate_statistics", "generate_codebooks"):
        getattr(lib, fn).argtypes = [C.POINTER(QualityFile)]
        getattr(lib, fn).restype = None

    rows = G.case_rows(name)
    opts = G.case_opts(name)
    ph = G.phase(name)
    n, c, k = rows.shape[0], rows.shape[1] - 1, opts["k"]
    inp = tmp_path / "in.txt"
    inp.write_bytes(rows.tobytes())

    o = Options(mode=opts["mode"], clusters=k, distortion=O.DIST_BY_LETTER[opts["dist"]], ratio=opts["ratio"], cluster_threshold=4.0)
    info = QualityFile()
    info.alphabet = lib.alloc_alphabet(72)
    info.dist = lib.generate_distortion_matrix(72, O.DIST_BY_LETTER[opts["dist"]])
    info.cluster_count = k
    assert lib.load_file(str(inp).encode(), C.byref(info), 0) == 0
    assert info.lines == n and info.columns == c and info.block_count == 1
    assert lib.load_file(b"/nonexistent/file", C.byref(QualityFile()), 0) == 1        # LF_ERROR_NOT_FOUND
    info.clusters = lib.alloc_cluster_list(C.byref(info))
    info.opts = C.pointer(o)

    C.CDLL(None).srand(1)                       # fresh-process rand() state (cluster.c:199-200)
    lib.do_kmeans_clustering(C.byref(info))
    labels = np.array([info.blocks[0].lines[i].cluster for i in range(n)], dtype=np.uint8)
    np.testing.assert_array_equal(labels, ph["labels"])
    for j in range(k):
        cl = info.clusters.contents.clusters[j]
        assert cl.count == int(ph["counts"][j])
        np.testing.assert_array_equal(np.ctypeslib.as_array(cl.mean, (c,)), ph["means"][j])
        acc = rows[labels == j, :c].astype(np.uint64).sum(axis=0)                    # sums of raw bytes, cluster.c:98
        np.testing.assert_array_equal(np.ctypeslib.as_array(cl.accumulator, (c,)), acc)

    lib.calculate_statistics(C.byref(info))
    for j in range(k):
        ts = info.clusters.contents.clusters[j].training_stats.contents
        for s in (0, 1, c // 2, c - 1):
            for p in ((0,) if s == 0 else (0, 20, 38, 71)):
                pmf = ts.pmfs[0 if s == 0 else 1 + (s - 1) * 72 + p].contents
                cnt = np.ctypeslib.as_array(pmf.counts, (72,))
                np.testing.assert_array_equal(cnt, ph["cnt"][j, s, p])
                assert pmf.total == int(ph["cnt"][j, s, p].sum())
        for s in range(c):
            m = np.ctypeslib.as_array(ts.marginal_pmfs.contents.pmfs[s].contents.pmf, (72,))
            assert m.tobytes() == ph["marg"][j, s].tobytes(), "marginal pmf not bit-identical to the reference"

    lib.generate_codebooks(C.byref(info))
    d = O.distortion(opts["dist"])
    for j in range(k):
        ref = O.generate_codebook(ph["cnt"][j], d, opts["mode"], opts["ratio"])
        ql = info.clusters.contents.clusters[j].qlist.contents
        assert ql.columns == c
        for s in range(c):
            nctx = int(ref.nctx[s])
            ia = ql.input_alphabets[s].contents
            assert ia.size == nctx
            np.testing.assert_array_equal(np.ctypeslib.as_array(ia.symbols, (nctx,)), ref.ctx_sym[s, :nctx])
            for ctx in range(nctx):
                assert ql.ratio[s][ctx] == ref.ratio[s, ctx] and ql.qratio[s][ctx] == ref.qratio[s, ctx]
                for b in range(2):
                    qz = ql.q[s][2 * ctx + b].contents
                    np.testing.assert_array_equal(np.ctypeslib.as_array(qz.q, (72,)), ref.q[s, 2 * ctx + b])
                    S = int(ref.nstates[s, 2 * ctx + b])
                    oa = qz.output_alphabet.contents
                    assert oa.size == S
                    np.testing.assert_array_equal(np.ctypeslib.as_array(oa.symbols, (S,)), ref.recon[s, 2 * ctx + b, :S])
                    want_ratio = ref.ratio[s, ctx] if b == 0 else 1 - ref.ratio[s, ctx]
                    assert qz.ratio == want_ratio

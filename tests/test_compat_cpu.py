"""CPU-side tests of the decode direction and of the symbol-level compat interface (include/qvz_compat.h): no GPU needed,
the single serial stream is host work by definition. Every expectation is a file the REFERENCE binary wrote (tests/golden)."""
import ctypes as C
import os
import subprocess

import numpy as np
import pytest

import golden_util as G

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
PKG = os.path.join(ROOT, "qvz_amd")
QVZ = os.path.join(PKG, "bin", "qvz")
DROPIN = os.path.join(ROOT, "oracle", "_ref", "qvz_dropin")
SMALL = sorted(G.CASES)


@pytest.fixture(scope="module")
def harness(tmp_path_factory):
    out = str(tmp_path_factory.mktemp("h") / "compat_symbol_harness")
    subprocess.run(["gcc", "-O2", "-std=gnu11", "-I", os.path.join(ROOT, "include"), "-o", out,
                    os.path.join(ROOT, "tests", "compat_symbol_harness.c"), "-L", PKG, "-lqvzcompat", "-lqvzhip",
                    f"-Wl,-rpath,{PKG}", "-Wl,-rpath-link,/opt/rocm/lib"], check=True)
    return out


def _q_bytes(name):
    """The reference's .q of a fixture: stored, or (large ones) re-made by the CPU oracle and checked against the md5 of the
    file the reference wrote."""
    e = G.MANIFEST["cases"][name]
    if e["stored"]:
        return G.gz(name + ".q.gz")
    import oracle_lib as O
    q = O.encode(G.case_rows(name), want_u=False, **G.case_opts(name))["q"]
    assert G.md5(q) == e["q_md5"]
    return q


def _u_md5(name):
    return G.MANIFEST["cases"][name]["u_md5"]


def _q(tmp_path, name):
    p = tmp_path / (name + ".q")
    p.write_bytes(_q_bytes(name))
    return str(p)


@pytest.mark.parametrize("name", SMALL)
def test_qvz_x_decodes_the_references_file_without_a_gpu(tmp_path, name):
    """`qvz -x` on a reference-written .q: header, codebooks, WELL replay and the serial chain are host C."""
    out = str(tmp_path / "dec.txt")
    r = subprocess.run([QVZ, "-x", _q(tmp_path, name), out], capture_output=True, text=True)
    assert r.returncode == 0, r.stdout + r.stderr
    assert G.md5(open(out, "rb").read()) == _u_md5(name)


@pytest.mark.parametrize("name", ["t2k_f05_L", "t2k_c3_f05_M", "tiny5_r10"])
def test_reference_main_on_the_compat_library_decodes(tmp_path, name):
    """The reference's UNMODIFIED main.c linked against libqvzcompat.so (oracle/_ref/qvz_dropin): its decode() calls
    read_codebooks + start_qv_decompression (src/main.c:132-160)."""
    if not os.path.exists(DROPIN):
        pytest.skip("oracle/_ref/qvz_dropin is built only where /root/reference is present")
    out = str(tmp_path / "dec.txt")
    r = subprocess.run([DROPIN, "-x", "-v", _q(tmp_path, name), out], capture_output=True, text=True)
    assert r.returncode == 0, r.stdout + r.stderr
    assert open(out, "rb").read() == G.gz(name + ".u.gz")
    n = G.CASES[name]["input"]["n"]
    assert f"Decoded {n} lines in" in r.stdout                                   # main.c:158-160; info->lines as the reference leaves it


@pytest.mark.parametrize("name", ["t2k_f05_L", "t2k_c3_f05_M", "t2k_r10_M", "tiny5_r10", "c2_f08"])
def test_symbol_level_interface_round_trip(tmp_path, harness, name):
    """compress_qv / qv_write_cluster / decompress_qv / qv_read_cluster / initialize_qv_compressor / choose_quantizer driven
    symbol by symbol (the loops of src/qv_compressor.c:76-135,166-229): decoding gives the reference's -u text, re-encoding
    the original input with the decoded cluster ids gives the reference's .q byte for byte."""
    if name not in G.CASES:
        pytest.skip("fixture not present")
    q = _q(tmp_path, name)
    dec, lab, req = str(tmp_path / "dec.txt"), str(tmp_path / "labels.bin"), str(tmp_path / "re.q")
    r = subprocess.run([harness, "decode", q, dec, lab], capture_output=True, text=True)
    assert r.returncode == 0, r.stdout + r.stderr
    assert open(dec, "rb").read() == G.gz(name + ".u.gz")
    rows = G.case_rows(name)
    inp = str(tmp_path / "in.txt")
    open(inp, "wb").write(rows.tobytes())
    env = dict(os.environ, QVZ_WELL_SEED="debug")                                # the fixtures come from the -DDEBUG build
    r = subprocess.run([harness, "encode", q, inp, lab, req], capture_output=True, text=True, env=env)
    assert r.returncode == 0, r.stdout + r.stderr
    assert open(req, "rb").read() == G.gz(name + ".q.gz")


def test_corrupt_files_are_refused_not_crashed_on(tmp_path):
    """Truncated and bit-flipped .q files: QVZ_E_DATA (or a clean decode of garbage), never an out-of-bounds access. The
    codebook reader range-checks every table and ratio byte (the reference indexes with them unchecked)."""
    from qvz_amd._lib import lib
    good = G.gz("t2k_f05_L.q.gz")
    rng = np.random.default_rng(7)
    out = str(tmp_path / "o.txt").encode()
    cases = [good[:n] for n in (0, 5, 9, 10, 200, 5000, len(good) // 2, len(good) - 1)]
    for _ in range(60):
        b = bytearray(good)
        for pos in rng.integers(9, 200_000 if len(b) > 200_000 else len(b), size=4):
            b[int(pos)] = int(rng.integers(0, 256))
        cases.append(bytes(b))
    refused = 0
    for i, data in enumerate(cases):
        p = tmp_path / f"bad{i}.q"
        p.write_bytes(data)
        rc = lib().qvz_decode_file(str(p).encode(), out, 0, None)
        assert rc in (0, -3, -1), rc
        refused += rc != 0
    assert refused >= 8                                                          # the truncations at the very least


def test_shard_bounds_leave_no_shard_empty():
    from qvz_amd import host
    from qvz_amd._lib import lib
    for n, s in [(1000, 100), (1000, 250), (7, 1), (100_000_000, 8192), (12, 3), (1001, 250)]:
        assert s <= lib().qvz_shard_limit(n)
        first = host.shard_bounds(n, s)
        assert first[0] == 0 and first[-1] == n and all(f % 4 == 0 for f in first[:-1])
        assert all(b > a for a, b in zip(first, first[1:])), (n, s)
    assert lib().qvz_shard_limit(1000) == 250 and lib().qvz_shard_limit(3) == 1


def test_shardset_file_layout(tmp_path):
    """A shard set written from parsed codebooks: magic, header fields, one copy of the codebook bytes, streams back to
    back; qvz_shardset_stat reads it back."""
    from qvz_amd import host
    from qvz_amd._lib import lib
    L = lib()
    good = G.gz("t2k_c3_f05_M.q.gz")
    k, c = good[0], int.from_bytes(good[1:5], "big")
    buf = np.frombuffer(good, dtype=np.uint8)
    pos = C.c_size_t(9)
    cbs = []
    for _ in range(k):
        h = L.qvz_codebook_alloc(c)
        assert L.qvz_codebook_read(buf.ctypes.data, buf.size, C.byref(pos), h) == 0
        cbs.append(h)
    cb_bytes = good[9:pos.value]
    n, shards, stride = 2000, 5, 64
    streams = np.arange(shards * stride, dtype=np.uint8).reshape(shards, stride)
    sizes = (C.c_uint64 * shards)(10, 0, 64, 3, 17)
    well = (C.c_uint32 * 32)(*range(100, 132))
    libc = C.CDLL(None)
    libc.fopen.restype = C.c_void_p
    libc.fopen.argtypes = [C.c_char_p, C.c_char_p]
    libc.fclose.argtypes = [C.c_void_p]
    path = str(tmp_path / "set.qs")
    f = libc.fopen(path.encode(), b"wb")
    si = host.ShardsetInfo()
    arr = (C.c_void_p * k)(*cbs)
    assert L.qvz_shardset_write(f, k, c, n, shards, well, arr, streams.ctypes.data, stride, sizes, C.byref(si)) == 0
    libc.fclose(f)
    data = open(path, "rb").read()
    assert data[:8] == b"\0QVZSET1" and L.qvz_shardset_is(data, len(data)) == 1 and L.qvz_shardset_is(good, len(good)) == 0
    assert int.from_bytes(data[8:12], "little") == k and int.from_bytes(data[12:16], "little") == c
    assert int.from_bytes(data[16:24], "little") == n and int.from_bytes(data[24:28], "little") == shards
    assert data[32:160] == bytes(well)
    at = 160 + 8 * shards
    assert data[at:at + len(cb_bytes)] == cb_bytes                                # the reference's codebook text, once
    body = b"".join(streams[g, :sizes[g]].tobytes() for g in range(shards))
    assert data[at + len(cb_bytes):] == body
    st = host.shardset_stat(path)
    assert (st.clusters, st.columns, st.shards, st.lines) == (k, c, shards, n)
    assert (st.header_bytes, st.codebook_bytes, st.stream_bytes, st.file_bytes) == (at, len(cb_bytes), len(body), len(data))
    assert (si.file_bytes, si.codebook_bytes) == (len(data), len(cb_bytes))
    # A line count that makes N * (C + 1), C * ld or first_read * C wrap 64 bits (2^62 lines of c columns) must be
    # refused by everything that parses the header - stat, decode and split - before any size is computed from it.
    for lines in (1 << 62, (1 << 64) - 4, ((1 << 48) // (c + 1)) + 1):
        bad = bytearray(data)
        bad[16:24] = lines.to_bytes(8, "little")
        bp = str(tmp_path / "bad.qs")
        open(bp, "wb").write(bytes(bad))
        info = host.ShardsetInfo()
        assert L.qvz_shardset_stat(bp.encode(), C.byref(info)) != 0
        assert b"implausible" in L.qvz_encode_last_error()
        assert L.qvz_shardset_decode(bp.encode(), str(tmp_path / "bad.txt").encode(), 0, None) != 0
        assert L.qvz_shardset_split(bp.encode(), str(tmp_path / "piece").encode(), 0, 0, 0) != 0
    ok = bytearray(data)                                                           # the largest count that is accepted parses
    ok[16:24] = ((1 << 48) // (c + 1)).to_bytes(8, "little")
    open(bp, "wb").write(bytes(ok))
    assert L.qvz_shardset_stat(bp.encode(), C.byref(info)) == 0
    for h in cbs:
        L.qvz_codebook_free(h)


def test_compat_library_exports_every_reference_entry_point():
    """libqvzcompat.so carries every function include/qvz_compat.h declares under the reference's own name."""
    import re
    text = open(os.path.join(ROOT, "include", "qvz_compat.h")).read()
    text = re.sub(r"/\*.*?\*/", "", text, flags=re.S)
    names = set(re.findall(r"\b([a-z_0-9A-Z]+)\s*\([^;{]*\)\s*;", text))
    must = {"load_file", "alloc_cluster_list", "do_kmeans_clustering", "calculate_statistics", "generate_codebooks", "write_codebooks",
            "start_qv_compression", "read_codebooks", "start_qv_decompression", "compress_qv", "qv_write_cluster",
            "initialize_qv_compressor", "decompress_qv", "qv_read_cluster", "choose_quantizer", "encoder_last_step"}
    assert must <= names, must - names
    L = C.CDLL(os.path.join(PKG, "libqvzcompat.so"))
    for n in sorted(names):
        assert hasattr(L, n), f"{n} is declared in include/qvz_compat.h but not exported by libqvzcompat.so"

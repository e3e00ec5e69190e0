"""End-to-end GPU tests of the drop-in surface: the host C driver (qvz_encode_file) and the `qvz` binary must
produce byte-identical .q and -u files to the reference for the same WELL state, and print the reference's -s line."""
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
QVZ_BIN = os.path.join(ROOT, "qvz_amd", "bin", "qvz")


def fresh_process_rand():
    """Centroids come from libc rand() with whatever state the process is in (reference cluster.c:199-200): a new
    process starts as if srand(1) had been called, so in-process tests reset it the same way."""
    import ctypes
    ctypes.CDLL(None).srand(1)


def write_input(tmp_path, name):
    rows = G.case_rows(name)
    p = tmp_path / (name + ".txt")
    p.write_bytes(rows.tobytes())
    return str(p), rows


@pytest.mark.parametrize("name", sorted(G.CASES))
def test_encode_file_byte_identical(tmp_path, name):
    from qvz_amd import host
    e = G.CASES[name]
    inp, rows = write_input(tmp_path, name)
    opts = G.case_opts(name)
    q, u = str(tmp_path / "out.q"), str(tmp_path / "out.u")
    fresh_process_rand()
    r = host.encode_file(inp, q, clusters=opts["k"], mode=opts["mode"], ratio=opts["ratio"], dist=opts["dist"],
                         u_path=u, seed_mode=host.SEED_DEBUG)
    qb, ub = open(q, "rb").read(), open(u, "rb").read()
    assert len(qb) == e["q_size"] == r.file_bytes
    assert G.md5(ub) == e["u_md5"], "-u differs from the reference"
    assert G.md5(qb) == e["q_md5"], ".q differs from the reference"
    assert r.stream_bytes == e["size"] and r.lines == rows.shape[0] and r.columns == rows.shape[1] - 1
    assert f"{r.distortion:.4f}" == f"{e['distortion']:.4f}"
    # the reference decoder's invariant (test.sh): decode(.q) == -u ; the oracle's decoder stands in for qvz -x
    assert O.decode(qb) == ub


@pytest.mark.parametrize("chunk", [256, 768, 4096])
def test_chunking_does_not_change_the_bytes(tmp_path, chunk):
    from qvz_amd import host
    name = "t2k_c3_f05_M"
    inp, _ = write_input(tmp_path, name)
    opts = G.case_opts(name)
    q, u = str(tmp_path / "o.q"), str(tmp_path / "o.u")
    fresh_process_rand()
    host.encode_file(inp, q, clusters=opts["k"], mode=opts["mode"], ratio=opts["ratio"], dist=opts["dist"], u_path=u,
                     seed_mode=host.SEED_DEBUG, chunk_reads=chunk)
    assert open(q, "rb").read() == G.gz(name + ".q.gz")
    assert open(u, "rb").read() == G.gz(name + ".u.gz")


def test_explicit_well_state_roundtrip(tmp_path):
    """A random 128-byte state (what the reference draws from rand()): stored in the file, and the output decodes
    to the -u dump with the oracle's decoder."""
    from qvz_amd import host
    inp, rows = write_input(tmp_path, "t2k_f05_L")
    state = np.random.default_rng(5).integers(0, 2 ** 31, size=32, dtype=np.uint32)
    q, u = str(tmp_path / "o.q"), str(tmp_path / "o.u")
    host.encode_file(inp, q, mode=host.MODE_RATIO, ratio=0.5, dist="L", u_path=u, well_state=state)
    qb, ub = open(q, "rb").read(), open(u, "rb").read()
    ref = O.encode(rows, k=1, mode=O.MODE_RATIO, ratio=0.5, dist="L", well_state=state)
    assert qb == ref["q"] and ub == ref["u"]
    assert O.decode(qb) == ub


@pytest.mark.parametrize("name", ["t2k_r10_M", "t2k_c3_f05_M", "cfg1_f05_L"])
def test_cli_binary_matches_reference_output_and_stats_line(tmp_path, name):
    e = G.CASES[name]
    inp, _ = write_input(tmp_path, name)
    q, u = str(tmp_path / "cli.q"), str(tmp_path / "cli.u")
    env = dict(os.environ, QVZ_WELL_SEED="debug")
    out = subprocess.run([QVZ_BIN] + e["flags"].split() + ["-s", "-u", u, inp, q], env=env, capture_output=True,
                         text=True, check=True).stdout
    assert G.md5(open(q, "rb").read()) == e["q_md5"]
    assert G.md5(open(u, "rb").read()) == e["u_md5"]
    m = re.search(r"rate, ([\d.]+), distortion, ([\d.]+), time, ([\d.]+), size, (\d+) ", out)
    assert m, out
    assert m.group(1) == f"{e['rate']:.4f}" and m.group(2) == f"{e['distortion']:.4f}" and int(m.group(4)) == e["size"]


def test_cli_custom_distortion_file_matches_reference(tmp_path):
    """-D FILE (src/distortion.c:100-145, main.c:293-297): an asymmetric matrix; .q, -u and the -s line are the reference's
    (fixture made by the reference binary), the file decodes to the -u text, and a file holding the MSE matrix gives
    the bytes of -d M."""
    name = "t2k_D_asym_f06"
    e = G.CUSTOM_DIST_CASES[name]
    rows = O.qgen1_rows(e["input"]["seed"], e["input"]["n"], e["input"]["c"])
    assert G.md5(rows.tobytes()) == e["input_md5"]
    inp = str(tmp_path / "in.txt")
    open(inp, "wb").write(rows.tobytes())
    q, u, d = str(tmp_path / "c.q"), str(tmp_path / "c.u"), str(tmp_path / "c.dec")
    env = dict(os.environ, QVZ_WELL_SEED="debug")
    out = subprocess.run([QVZ_BIN] + G.cli_flags(e) + ["-s", "-v", "-u", u, inp, q], env=env, capture_output=True,
                         text=True, check=True).stdout
    assert G.md5(open(q, "rb").read()) == e["q_md5"] == G.md5(G.gz(name + ".q.gz"))
    assert G.md5(open(u, "rb").read()) == e["u_md5"]
    m = re.search(r"rate, ([\d.]+), distortion, ([\d.]+), time, ([\d.]+), size, (\d+) ", out)
    assert m and m.group(1) == f"{e['rate']:.4f}" and m.group(2) == f"{e['distortion']:.4f}" and int(m.group(4)) == e["size"]
    assert "A custom distortion metric stored in" in out and "Custom distortion: " in out
    subprocess.run([QVZ_BIN, "-x", q, d], env=env, check=True, capture_output=True)
    assert open(d, "rb").read() == open(u, "rb").read()
    mse = tmp_path / "mse.txt"
    mse.write_text("# (x - y)^2\n" + "\n".join(",".join(str((x - y) ** 2) for y in range(72)) for x in range(72)) + "\n")
    qa, qb = str(tmp_path / "a.q"), str(tmp_path / "b.q")
    subprocess.run([QVZ_BIN, "-c", "1", "-D", str(mse), "-f", "0.6", inp, qa], env=env, check=True, capture_output=True)
    subprocess.run([QVZ_BIN, "-c", "1", "-d", "M", "-f", "0.6", inp, qb], env=env, check=True, capture_output=True)
    assert open(qa, "rb").read() == open(qb, "rb").read()
    r = subprocess.run([QVZ_BIN, "-D", str(tmp_path / "missing.txt"), inp, qa], env=env, capture_output=True, text=True)
    assert r.returncode != 0


def test_cli_fastq_input_equals_quality_file_input(tmp_path):
    """A FASTQ file (four-line records) is detected and encoded exactly like the file of its quality lines; QVZ_INPUT=qual
    turns the detection off (the FASTQ text is then an ordinary, malformed, quality file)."""
    from test_host_cpu import make_fastq
    name = "t2k_f05_L"
    e = G.CASES[name]
    inp, rows = write_input(tmp_path, name)
    fq = tmp_path / "reads.fastq"
    fq.write_bytes(make_fastq(rows, seed=3))
    qa, qb, ua = str(tmp_path / "a.q"), str(tmp_path / "b.q"), str(tmp_path / "a.u")
    env = dict(os.environ, QVZ_WELL_SEED="debug")
    out = subprocess.run([QVZ_BIN] + e["flags"].split() + ["-v", "-u", ua, str(fq), qa], env=env, check=True, capture_output=True,
                         text=True).stdout
    assert f"FASTQ input: {rows.shape[0]} reads of {rows.shape[1] - 1} quality values" in out
    subprocess.run([QVZ_BIN] + e["flags"].split() + [inp, qb], env=env, check=True, capture_output=True)
    assert open(qa, "rb").read() == open(qb, "rb").read()
    assert G.md5(open(qa, "rb").read()) == e["q_md5"] and G.md5(open(ua, "rb").read()) == e["u_md5"]
    r = subprocess.run([QVZ_BIN] + e["flags"].split() + [str(fq), qa], env=dict(env, QVZ_INPUT="qual"), capture_output=True, text=True)
    assert r.returncode != 0


def test_cli_verbose_lines_and_errors(tmp_path):
    inp, _ = write_input(tmp_path, "t2k_c3_f05_M")
    env = dict(os.environ, QVZ_WELL_SEED="debug")
    out = subprocess.run([QVZ_BIN, "-c", "3", "-f", "0.5", "-v", inp, str(tmp_path / "v.q")], env=env,
                         capture_output=True, text=True, check=True).stdout
    for needle in ["will be encoded as", "Ratio mode selected, targeting 0.500000 compression ratio.",
                   "MSE will be used as a distortion metric.", "Compression will use 3 clusters, with a movement threshold of 4.",
                   "Chose block 0, line 886.", "Chose block 0, line 915.", "Chose block 0, line 335.",
                   "Total number of iterations: 4.", "Clustering took", "Stats and codebook generation took", "Line: 0M",
                   "MSE distortion: 0.597", "Lines: 2000", "Columns: 30", "Total bytes used: 11488", "Encoding took",
                   "Total time elapsed:"]:
        assert needle in out, (needle, out)
    # reference behaviour for bad command lines: message + usage + exit status 1 (main.c:298-309)
    r = subprocess.run([QVZ_BIN, "-z", inp, "x"], capture_output=True, text=True)
    assert r.returncode == 1 and "Unrecognized option -z." in r.stdout and "Usage:" in r.stdout
    r = subprocess.run([QVZ_BIN, inp], capture_output=True, text=True)
    assert r.returncode == 1 and "Missing required filenames." in r.stdout
    r = subprocess.run([QVZ_BIN, str(tmp_path / "missing.txt"), str(tmp_path / "o.q")], capture_output=True, text=True)
    assert r.returncode == 1 and "load_file returned error: 1" in r.stdout
    # out-of-alphabet bytes are undefined behaviour in the reference; here they are refused
    bad = tmp_path / "bad.txt"
    rows = G.case_rows("tiny5_r10").copy()
    rows[2, 3] = 10
    bad.write_bytes(rows.tobytes())
    r = subprocess.run([QVZ_BIN, str(bad), str(tmp_path / "o.q")], capture_output=True, text=True)
    assert r.returncode == 1 and "outside" in r.stdout


REF_BIN = os.path.join(ROOT, "oracle", "_ref", "qvz_ref")


def decode_with_reference_or_oracle(path, tmp_path):
    """`qvz -x` of the unmodified reference when its binary travelled with the repo (oracle/_ref), else the oracle's decoder."""
    if os.path.exists(REF_BIN):
        out = str(tmp_path / (os.path.basename(path) + ".dec"))
        subprocess.run([REF_BIN, "-x", path, out], check=True, capture_output=True)
        return open(out, "rb").read()
    return O.decode(open(path, "rb").read())


@pytest.mark.parametrize("name,shards", [("t2k_r10_M", 3), ("t2k_c3_f05_M", 2), ("cfg1_f05_L", 7), ("wide_c151", 2)])
def test_sharded_files_decode_with_the_reference_decoder(tmp_path, name, shards):
    """QVZ_SHARDS / qvz_encode_file_sharded: every <prefix>.<g> is a complete reference-format file; decoded one by one
    (by the reference binary itself where it is available) and concatenated they give the reference's -u text."""
    from qvz_amd import host
    e = G.CASES[name]
    inp, rows = write_input(tmp_path, name)
    opts = G.case_opts(name)
    prefix, u = str(tmp_path / "sh.q"), str(tmp_path / "sh.u")
    fresh_process_rand()
    r = host.encode_file_sharded(inp, prefix, shards, clusters=opts["k"], mode=opts["mode"], ratio=opts["ratio"],
                                 dist=opts["dist"], u_path=u, seed_mode=host.SEED_DEBUG)
    assert G.md5(open(u, "rb").read()) == e["u_md5"]
    files = [f"{prefix}.{g:04d}" for g in range(shards)]
    decoded = b"".join(decode_with_reference_or_oracle(f, tmp_path) for f in files)
    assert G.md5(decoded) == e["u_md5"]
    assert r.file_bytes == sum(os.path.getsize(f) for f in files)
    assert f"{r.distortion:.4f}" == f"{e['distortion']:.4f}"
    assert r.stream_bytes >= e["size"] * 0.98


@pytest.mark.parametrize("name", ["t2k_c3_f05_M", "wide_c151"])
def test_cli_text_written_through_preallocated_pages(tmp_path, name):
    """Big -u / decode outputs are written through a mapping into pages that a helper thread makes ahead of the writer
    (host/qvz_egress.c: qvz_prealloc_start; by default for outputs of 256 MB and more - the full-size CLI tests). Here the
    threshold is lowered so that small fixtures take that path: the same bytes as through pwrite, file length included."""
    e = G.CASES[name]
    inp, _ = write_input(tmp_path, name)
    outs = {}
    for tag, extra in (("pwrite", {"QVZ_PREALLOC": "0"}), ("mapped", {"QVZ_PREALLOC": "min=1"})):
        q, u, d = (str(tmp_path / f"{tag}.{x}") for x in ("qset", "u", "dec"))
        env = dict(os.environ, QVZ_WELL_SEED="debug", QVZ_SHARDS="5", **extra)
        subprocess.run([QVZ_BIN] + e["flags"].split() + ["-u", u, inp, q], env=env, check=True, capture_output=True)
        subprocess.run([QVZ_BIN, "-x", q, d], env=dict(os.environ, **extra), check=True, capture_output=True)
        outs[tag] = (open(q, "rb").read(), open(u, "rb").read(), open(d, "rb").read())
    assert outs["pwrite"] == outs["mapped"]
    assert G.md5(outs["mapped"][1]) == e["u_md5"] and outs["mapped"][2] == outs["mapped"][1]


def test_cli_qvz_shards_env(tmp_path):
    """QVZ_SHARDS=N: the default product is ONE shard-set file (codebooks once, include/qvz_shardset.h); `qvz -x` decodes
    it, qvz-shardset splits it into reference-format files the unmodified reference decoder reads; QVZ_SHARD_FILES=1 writes
    those files directly."""
    from qvz_amd import host
    name = "t2k_r10_M"
    e = G.CASES[name]
    inp, _ = write_input(tmp_path, name)
    q = str(tmp_path / "cli_sh.qset")
    env = dict(os.environ, QVZ_WELL_SEED="debug", QVZ_SHARDS="4")
    out = subprocess.run([QVZ_BIN] + e["flags"].split() + ["-s", inp, q], env=env, capture_output=True, text=True, check=True).stdout
    assert re.search(r"rate, [\d.]+, distortion, ([\d.]+)", out).group(1) == f"{e['distortion']:.4f}"
    blob = open(q, "rb").read()
    assert blob[:8] == b"\0QVZSET1"
    si = host.shardset_stat(q)
    c, n = e["input"]["c"], e["input"]["n"]
    assert (si.clusters, si.columns, si.lines, si.shards) == (1, c, n, 4)
    ref_q = G.gz(name + ".q.gz")
    assert si.codebook_bytes == len(ref_q) - 9 - 128 - e["size"]                 # the reference's codebook text, once
    assert si.file_bytes == len(blob) == 160 + 8 * 4 + si.codebook_bytes + si.stream_bytes
    # decode the set directly
    dec = str(tmp_path / "set.dec")
    subprocess.run([QVZ_BIN, "-x", q, dec], env=dict(os.environ), check=True, capture_output=True)
    assert G.md5(open(dec, "rb").read()) == e["u_md5"]
    # split it: every piece is a reference-format file
    tool = os.path.join(os.path.dirname(QVZ_BIN), "qvz-shardset")
    st = subprocess.run([tool, "stat", q], capture_output=True, text=True, check=True).stdout
    assert f"shards, 4" in st and "rate on disk" in st
    prefix = str(tmp_path / "piece")
    subprocess.run([tool, "split", q, prefix], check=True, capture_output=True)
    decoded = b"".join(decode_with_reference_or_oracle(f"{prefix}.{g:04d}", tmp_path) for g in range(4))
    assert G.md5(decoded) == e["u_md5"]
    # ... and the same files written directly
    q2 = str(tmp_path / "cli_files.q")
    subprocess.run([QVZ_BIN] + e["flags"].split() + [inp, q2], env=dict(env, QVZ_SHARD_FILES="1"), capture_output=True, check=True)
    for g in range(4):
        assert open(f"{q2}.{g:04d}", "rb").read() == open(f"{prefix}.{g:04d}", "rb").read()
    d2 = str(tmp_path / "files.dec")
    subprocess.run([QVZ_BIN, "-x", q2, d2], env=dict(env, QVZ_SHARD_FILES="1"), check=True, capture_output=True)
    assert G.md5(open(d2, "rb").read()) == e["u_md5"]


@pytest.mark.parametrize("name,shards", [("t2k_c3_f05_M", 3), ("cfg1_f05_L", 64), ("tiny5_r10", 1), ("r35_high_rate", 5)])
def test_shardset_round_trip_and_well_jumps(tmp_path, name, shards):
    """qvz_encode_file_shardset -> qvz_shardset_decode == the reference's -u; the split pieces carry the WELL state advanced to
    each shard's first draw (qvz_hip_well_jump_many == one qvz_hip_well_jump per shard) and decode with the reference."""
    from qvz_amd import host, kernels as K
    from qvz_amd._lib import lib
    e = G.CASES[name]
    inp, _ = write_input(tmp_path, name)
    opts = G.case_opts(name)
    q = str(tmp_path / "x.qset")
    fresh_process_rand()
    r = host.encode_file_shardset(inp, q, shards, clusters=opts["k"], mode=opts["mode"], ratio=opts["ratio"], dist=opts["dist"],
                                  seed_mode=host.SEED_DEBUG)
    si = host.shardset_stat(q)
    assert r.file_bytes == si.file_bytes == os.path.getsize(q) and r.stream_bytes == si.stream_bytes
    dec = str(tmp_path / "x.dec")
    assert lib().qvz_shardset_decode(q.encode(), dec.encode(), 0, None) == 0, lib().qvz_encode_last_error()
    assert G.md5(open(dec, "rb").read()) == e["u_md5"]
    prefix = str(tmp_path / "p")
    host.shardset_split(q, prefix)
    n, c = e["input"]["n"], e["input"]["c"]
    first = host.shard_bounds(n, shards)
    decoded = b""
    for g in range(shards):
        piece = open(f"{prefix}.{g:04d}", "rb").read()
        assert int.from_bytes(piece[5:9], "big") == first[g + 1] - first[g]
        well_at = len(piece) - 128 - int.from_bytes(open(q, "rb").read()[160 + 8 * g:168 + 8 * g], "little")
        want = K.well_jump(O.WELL_DEBUG_STATE, first[g] * c // 4)
        assert piece[well_at:well_at + 128] == want.tobytes(), f"WELL state of shard {g}"
        decoded += decode_with_reference_or_oracle(f"{prefix}.{g:04d}", tmp_path)
    assert G.md5(decoded) == e["u_md5"]


@pytest.mark.parametrize("name", ["t2k_r10_M", "t2k_f05_L", "t2k_c3_f05_M", "cfg1_r10_M", "const_f05", "uniform72_r20_A",
                                  "c2_f08", "tiny5_r10", "twolevel_c2_f05", "f00_zero_rate", "r35_high_rate"])
def test_decode_file_of_the_reference_encoders_q(tmp_path, name):
    """qvz_decode_file / `qvz -x` on the .q file the REFERENCE wrote (fixture) gives the reference's -u text."""
    import ctypes as C
    from qvz_amd._lib import lib
    q = tmp_path / (name + ".q")
    q.write_bytes(G.gz(name + ".q.gz"))
    out = str(tmp_path / "dec.txt")
    rc = lib().qvz_decode_file(str(q).encode(), out.encode(), 0, None)
    assert rc == 0, lib().qvz_encode_last_error().decode()
    assert open(out, "rb").read() == G.gz(name + ".u.gz")
    cli = str(tmp_path / "cli.txt")
    v = subprocess.run([QVZ_BIN, "-x", "-v", str(q), cli], capture_output=True, text=True, check=True).stdout
    assert open(cli, "rb").read() == G.gz(name + ".u.gz")
    assert re.search(rf"Decoded {G.CASES[name]['input']['n']} lines in [\d.]+ seconds\.", v)


def test_encode_then_decode_round_trips_through_files(tmp_path):
    """qvz (this build) -> .q -> qvz -x (this build) == -u, for one stream and for QVZ_SHARDS (one shard-set file)."""
    name = "cfg1_f05_L"
    e = G.CASES[name]
    inp, _ = write_input(tmp_path, name)
    q, u, d = str(tmp_path / "rt.q"), str(tmp_path / "rt.u"), str(tmp_path / "rt.dec")
    env = dict(os.environ)                                       # time-seeded WELL state, as the reference does by default
    subprocess.run([QVZ_BIN] + e["flags"].split() + ["-u", u, inp, q], env=env, check=True, capture_output=True)
    subprocess.run([QVZ_BIN, "-x", q, d], env=env, check=True, capture_output=True)
    assert open(d, "rb").read() == open(u, "rb").read()
    env5 = dict(os.environ, QVZ_SHARDS="5")
    subprocess.run([QVZ_BIN] + e["flags"].split() + ["-u", u, inp, q], env=env5, check=True, capture_output=True)
    subprocess.run([QVZ_BIN, "-x", q, d], env=env5, check=True, capture_output=True)
    assert open(d, "rb").read() == open(u, "rb").read()


def test_decode_rejects_damaged_files(tmp_path):
    name = "t2k_r10_M"
    blob = bytearray(G.gz(name + ".q.gz"))
    bad = tmp_path / "bad.q"
    bad.write_bytes(bytes(blob[:5000]))                          # cut inside the codebooks
    r = subprocess.run([QVZ_BIN, "-x", str(bad), str(tmp_path / "o.txt")], capture_output=True, text=True)
    assert r.returncode == 1 and "codebook" in r.stdout
    r = subprocess.run([QVZ_BIN, "-x", str(tmp_path / "missing.q"), str(tmp_path / "o.txt")], capture_output=True, text=True)
    assert r.returncode == 1


@pytest.mark.parametrize("name", ["t2k_c3_f05_M", "cfg1_f05_L", "rescale_140k_c4", "rescale_c2_100k_c3", "wide_c151", "tiny5_r10"])
def test_strict_stream_with_the_model_on_the_gpu(tmp_path, name, monkeypatch):
    """The single-stream .q with the adaptive model on the GPU (qvz_hip_model_session_records + qvz_coder_encode_records,
    the default) equals the one with model and chain on the host (QVZ_STRICT_MODEL=host) and the reference's file - with
    several clusters, a table rescale inside the stream, 151 columns, and chunks of a few hundred reads (model state carried
    from chunk to chunk)."""
    from qvz_amd import host
    e = G.CASES[name]
    inp, _ = write_input(tmp_path, name)
    opts = G.case_opts(name)
    kw = dict(clusters=opts["k"], mode=opts["mode"], ratio=opts["ratio"], dist=opts["dist"], seed_mode=host.SEED_DEBUG)
    files = {}
    for label, env, chunk in (("gpu", None, 0), ("gpu_small_chunks", None, 512), ("host", "host", 0)):
        if env:
            monkeypatch.setenv("QVZ_STRICT_MODEL", env)
        else:
            monkeypatch.delenv("QVZ_STRICT_MODEL", raising=False)
        q = str(tmp_path / f"{label}.q")
        fresh_process_rand()
        r = host.encode_file(inp, q, chunk_reads=chunk, **kw)
        assert r.stream_bytes == e["size"]
        files[label] = open(q, "rb").read()
    assert files["gpu"] == files["host"] == files["gpu_small_chunks"]
    assert G.md5(files["gpu"]) == e["q_md5"]


def test_model_session_records_equal_a_serial_model():
    """qvz_hip_model_session_records against a plain serial model in Python integers (tests/test_host_cpu.records_from_feed):
    the records - floor(2^62 * c / n) + 1 for c = cum_lo, cum_hi - must be equal to the last bit, chunk after chunk."""
    import ctypes as C
    import torch
    from qvz_amd import kernels as K
    from qvz_amd._lib import lib
    from test_host_cpu import oracle_side, product_codebook_from_oracle, records_from_feed
    rows, opts, labels, ocbs, ref = oracle_side("t2k_c3_f05_M")
    n, c = rows.shape[0], rows.shape[1] - 1
    k = opts["k"]
    cbs = [product_codebook_from_oracle(o) for o in ocbs]
    feed = (ref["qi"].astype(np.uint16) | (ref["st"].astype(np.uint16) << 8))
    want, pitch = records_from_feed(feed, labels, ocbs, k)
    sess = C.c_void_p()
    arr = (C.c_void_p * k)(*[cb.handle for cb in cbs])
    assert lib().qvz_hip_model_session_open(c, k, arr, C.byref(sess)) == 0
    assert lib().qvz_hip_model_session_pitch(sess) == pitch
    cuts = [0, 700, 702, n]
    pos = [0] + list(range(8, 8 + c))
    for a, b in zip(cuts[:-1], cuts[1:]):
        m = b - a
        ld = (m + 255) & ~255
        qi = torch.zeros((c, ld), dtype=torch.uint8, device="cuda")
        st = torch.zeros((c, ld), dtype=torch.uint8, device="cuda")
        qi[:, :m] = torch.from_numpy(np.ascontiguousarray(ref["qi"][a:b].T)).cuda()
        st[:, :m] = torch.from_numpy(np.ascontiguousarray(ref["st"][a:b].T)).cuda()
        lab = torch.zeros(ld, dtype=torch.uint8, device="cuda")
        lab[:m] = torch.from_numpy(labels[a:b]).cuda()
        rec = torch.zeros((((m + 1) // 2) * 2 * pitch, 2), dtype=torch.int64, device="cuda")
        assert lib().qvz_hip_model_session_records(sess, qi.data_ptr(), st.data_ptr(), ld, lab.data_ptr(), m, rec.data_ptr(), None) == 0
        torch.cuda.synchronize()
        got = rec.cpu().numpy().view(np.uint64)
        for i in range(m):
            for p_ in pos:
                src = (((a + i) >> 1) * (pitch >> 2) + (p_ >> 2)) * 8 + ((a + i) & 1) * 4 + (p_ & 3)
                dst = ((i >> 1) * (pitch >> 2) + (p_ >> 2)) * 8 + (i & 1) * 4 + (p_ & 3)
                assert got[dst, 0] == want[src, 0] and got[dst, 1] == want[src, 1], (a + i, p_)
    assert lib().qvz_hip_model_session_status(sess, None) == 0
    lib().qvz_hip_model_session_close(sess)


@pytest.mark.parametrize("name,devices,shards", [("t2k_c3_f05_M", [0, 0], 6), ("cfg1_f05_L", [0, 0, 0], 16), ("twolevel_c2_f05", [0, 0], 2),
                                                 ("t2k_c3_f05_M", [0, 1], 6), ("cfg1_f05_L", [0, 1], 16)])
def test_several_devices_write_the_single_device_shard_set(tmp_path, name, devices, shards):
    """qvz_encode_file_devices (host/qvz_multi.c): reads dealt to the devices in whole shards, k-means accumulators and PMF
    counters summed over the devices, centroid rows from wherever they lie, WELL offsets per part - and the shard set, the
    -u text and the distortion are those of the single-device encode, byte for byte / bit for bit. The same ordinal listed
    twice runs the parts on one GPU with the host-staged sum (what a one-GPU box can test); distinct ordinals use RCCL and
    need that many GPUs."""
    import torch
    from qvz_amd import host
    if max(devices) >= torch.cuda.device_count():
        pytest.skip(f"needs {max(devices) + 1} GPUs")
    e = G.CASES[name]
    inp, _ = write_input(tmp_path, name)
    opts = G.case_opts(name)
    kw = dict(clusters=opts["k"], mode=opts["mode"], ratio=opts["ratio"], dist=opts["dist"], seed_mode=host.SEED_DEBUG)
    one, many = str(tmp_path / "one.qset"), str(tmp_path / "many.qset")
    u1, u2 = str(tmp_path / "one.u"), str(tmp_path / "many.u")
    fresh_process_rand()
    a = host.encode_file_shardset(inp, one, shards, u_path=u1, **kw)
    fresh_process_rand()
    b = host.encode_file_devices(inp, many, devices, shards, u_path=u2, **kw)
    assert open(many, "rb").read() == open(one, "rb").read()
    assert open(u2, "rb").read() == open(u1, "rb").read() and G.md5(open(u2, "rb").read()) == e["u_md5"]
    assert (a.stream_bytes, a.file_bytes, a.kmeans_iterations) == (b.stream_bytes, b.file_bytes, b.kmeans_iterations)
    assert a.distortion == b.distortion                                          # the same sequential sum in file order


def test_cli_qvz_devices_env(tmp_path):
    name = "t2k_c3_f05_M"
    e = G.CASES[name]
    inp, _ = write_input(tmp_path, name)
    q, d = str(tmp_path / "m.qset"), str(tmp_path / "m.dec")
    env = dict(os.environ, QVZ_WELL_SEED="debug", QVZ_SHARDS="4", QVZ_DEVICES="0,0")
    out = subprocess.run([QVZ_BIN] + e["flags"].split() + ["-s", inp, q], env=env, capture_output=True, text=True, check=True).stdout
    assert re.search(r"rate, [\d.]+, distortion, ([\d.]+)", out).group(1) == f"{e['distortion']:.4f}"
    subprocess.run([QVZ_BIN, "-x", q, d], check=True, capture_output=True)
    assert G.md5(open(d, "rb").read()) == e["u_md5"]


@pytest.mark.parametrize("name,shards", [("t2k_c3_f05_M", 16), ("cfg1_f05_L", 64), ("r35_high_rate", 5)])
def test_decoding_in_slabs_gives_the_same_text(tmp_path, name, shards):
    """A file whose matrices do not fit the device - any file of 2^32 reads and more - is decoded in slabs of whole streams: draws
    from the generator advanced to the slab's first read, decode, text appended (qvz_decode_set_slab_reads). Forced on small
    files: one stream per slab, a few streams per slab with a ragged last slab, and the default (one slab) all give the
    reference's -u text, for a shard set and for shard files."""
    from qvz_amd import host
    from qvz_amd._lib import lib
    e = G.CASES[name]
    inp, _ = write_input(tmp_path, name)
    opts = G.case_opts(name)
    n = e["input"]["n"]
    q, prefix = str(tmp_path / "x.qset"), str(tmp_path / "p")
    fresh_process_rand()
    host.encode_file_shardset(inp, q, shards, clusters=opts["k"], mode=opts["mode"], ratio=opts["ratio"], dist=opts["dist"],
                              seed_mode=host.SEED_DEBUG)
    host.shardset_split(q, prefix)
    per = -(-n // shards)
    try:
        for slab in (1, per * 3 + 1, n - 1, n, 0, (0, 1), (0, shards - 1), (per * 5, 2)):
            if isinstance(slab, tuple):
                lib().qvz_decode_set_slab(*slab)                 # ... and a limit on the streams of a slab (their tables' memory)
            else:
                lib().qvz_decode_set_slab_reads(slab)
            dec = str(tmp_path / "x.slab.dec")
            assert lib().qvz_shardset_decode(q.encode(), dec.encode(), 0, None) == 0, lib().qvz_encode_last_error()
            assert G.md5(open(dec, "rb").read()) == e["u_md5"], f"shard set, slabs of {slab} reads / (reads, streams)"
            assert lib().qvz_decode_shard_files(prefix.encode(), shards, dec.encode(), 0, None) == 0, lib().qvz_encode_last_error()
            assert G.md5(open(dec, "rb").read()) == e["u_md5"], f"shard files, slabs of {slab} reads"
    finally:
        lib().qvz_decode_set_slab_reads(0)
    # the binary reports how many slabs it took (QVZ_TIMING) - the default on a small file is one
    env = dict(os.environ, QVZ_TIMING="1")
    err = subprocess.run([QVZ_BIN, "-x", q, str(tmp_path / "cli.dec")], env=env, capture_output=True, text=True, check=True).stderr
    assert re.search(r"qvz_timing: decode .* slabs=1\b", err), err


def test_sharded_streams_that_outgrow_their_first_allotment(tmp_path):
    """Near-uniform input over all 72 symbols at -f 1.0 needs more than the 6 bits per symbol a stream is given at first
    (log2 72 = 6.17): the GPU coder reports the overflow, the driver repeats the coding with twice the room, and the set
    decodes to the -u text."""
    from qvz_amd import host
    from qvz_amd._lib import lib
    rng = np.random.default_rng(5)
    n, c = 6000, 24
    rows = np.full((n, c + 1), 10, dtype=np.uint8)
    rows[:, :c] = rng.integers(33, 33 + 72, size=(n, c), dtype=np.uint8)
    inp = str(tmp_path / "uni.txt")
    open(inp, "wb").write(rows.tobytes())
    q, u, d = str(tmp_path / "uni.qset"), str(tmp_path / "uni.u"), str(tmp_path / "uni.dec")
    r = host.encode_file_shardset(inp, q, 7, mode=host.MODE_RATIO, ratio=1.0, dist="M", u_path=u, seed_mode=host.SEED_DEBUG)
    assert r.stream_bytes * 8 > 6.0 * n * c, "this input was meant to need more than 6 bits per symbol"
    assert lib().qvz_shardset_decode(q.encode(), d.encode(), 0, None) == 0, lib().qvz_encode_last_error()
    assert open(d, "rb").read() == open(u, "rb").read()
    assert open(u, "rb").read() == rows.tobytes()                                # -f 1.0 on this input keeps every value


def test_one_rank_rccl_communicator_in_the_c_driver(tmp_path, monkeypatch):
    """QVZ_MULTI_REDUCE=rccl with a single device: the k-means accumulators and the PMF counters go through ncclAllReduce on a
    one-rank communicator (librccl.so bound with dlopen, ncclCommInitAll, ncclUint64 / ncclSum by their numeric codes) and
    the output must still be the single-device shard set."""
    from qvz_amd import host
    name = "t2k_c3_f05_M"
    inp, _ = write_input(tmp_path, name)
    opts = G.case_opts(name)
    kw = dict(clusters=opts["k"], mode=opts["mode"], ratio=opts["ratio"], dist=opts["dist"], seed_mode=host.SEED_DEBUG)
    one, rc = str(tmp_path / "one.qset"), str(tmp_path / "rccl.qset")
    fresh_process_rand()
    host.encode_file_shardset(inp, one, 4, **kw)
    monkeypatch.setenv("QVZ_MULTI_REDUCE", "rccl")
    fresh_process_rand()
    host.encode_file_devices(inp, rc, [0], 4, **kw)
    assert open(rc, "rb").read() == open(one, "rb").read()


def test_cli_one_named_device_is_honoured(tmp_path):
    """QVZ_DEVICES with a single ordinal selects that GPU (it used to be ignored silently): ordinal 0 gives the usual file,
    an ordinal the box does not have fails instead of running on device 0; QVZ_SHARD_FILES with several devices is refused."""
    name = "t2k_r10_M"
    e = G.CASES[name]
    inp, _ = write_input(tmp_path, name)
    env = dict(os.environ, QVZ_WELL_SEED="debug", QVZ_SHARDS="4")
    a, b = str(tmp_path / "a.qset"), str(tmp_path / "b.qset")
    subprocess.run([QVZ_BIN] + e["flags"].split() + [inp, a], env=env, capture_output=True, check=True)
    subprocess.run([QVZ_BIN] + e["flags"].split() + [inp, b], env=dict(env, QVZ_DEVICES="0"), capture_output=True, check=True)
    assert open(a, "rb").read() == open(b, "rb").read()
    r = subprocess.run([QVZ_BIN] + e["flags"].split() + [inp, b], env=dict(env, QVZ_DEVICES="63"), capture_output=True, text=True)
    assert r.returncode != 0
    r = subprocess.run([QVZ_BIN] + e["flags"].split() + [inp, b], env=dict(env, QVZ_DEVICES="0,0", QVZ_SHARD_FILES="1"),
                       capture_output=True, text=True)
    assert r.returncode != 0 and "QVZ_SHARD_FILES" in r.stdout

"""CPU-side tests (no GPU): the C-ABI library loads and exports every symbol the headers declare, and the host C
logic (arithmetic coder, codebook serialisation, quantizer-table builder) matches the reference fixtures."""
import ctypes as C
import glob
import os
import re

import numpy as np
import pytest

import golden_util as G
import oracle_lib as O

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def declared_functions():
    names = set()
    for h in glob.glob(os.path.join(ROOT, "include", "*.h")):
        text = open(h).read()
        text = re.sub(r"/\*.*?\*/", "", text, flags=re.S)
        names.update(re.findall(r"\b(qvz_\w+)\s*\(", text))
    return sorted(names)


def test_library_exports_every_declared_symbol():
    from qvz_amd import _lib
    lib = _lib.lib()
    names = declared_functions()
    assert len(names) >= 40
    for n in names:
        assert hasattr(lib, n), f"{n} is declared in include/ but not exported by libqvzhip.so"
        assert n in _lib.SIGNATURES, f"{n} has no ctypes signature in qvz_amd/_lib.py"
    for n in _lib.SIGNATURES:
        assert n in names, f"{n} is bound in _lib.py but not declared in include/"


def test_missing_library_fails_loudly(monkeypatch):
    from qvz_amd import _lib
    monkeypatch.setattr(_lib, "_lib", None)
    monkeypatch.setattr(_lib, "LIB_PATH", "/nonexistent/libqvzhip.so")
    with pytest.raises(_lib.QvzHipError, match="no CPU fallback"):
        _lib.lib()


def product_codebook_from_oracle(ref):
    """Fill a product qvz_codebook_t (host struct) with an oracle codebook's contents."""
    from qvz_amd import kernels as K
    cb = K.Codebook(ref.c)
    for name in ("nctx", "ctx_sym", "ctx_idx", "qratio", "ratio", "q", "nstates", "recon", "out_idx"):
        getattr(cb, name)[...] = getattr(ref, name)
    return cb


def oracle_side(name):
    rows = G.case_rows(name)
    opts = G.case_opts(name)
    labels = O.kmeans(rows, opts["k"])[0]
    cnt = O.stats(rows, labels, opts["k"])
    d = O.distortion(opts["dist"])
    cbs = [O.generate_codebook(cnt[j], d, opts["mode"], opts["ratio"]) for j in range(opts["k"])]
    ref = O.encode_stream(rows, labels, cbs, d, want_feed=True)
    return rows, opts, labels, cbs, ref


STORED = sorted(n for n, e in G.CASES.items() if e["stored"] and e["input"]["n"] * e["input"]["c"] <= 200000)


@pytest.mark.parametrize("name", STORED + ["cfg1_r10_M"])
def test_coder_reproduces_reference_stream(name):
    from qvz_amd import host
    rows, opts, labels, ocbs, ref = oracle_side(name)
    n, c = rows.shape[0], rows.shape[1] - 1
    cbs = [product_codebook_from_oracle(o) for o in ocbs]
    feed = (ref["qi"].astype(np.uint16) | (ref["st"].astype(np.uint16) << 8))
    coder = host.Coder(c, [cb.handle for cb in cbs])
    # feed it in uneven pieces, draining the buffer in between: the stream must not depend on the chunking
    cuts = sorted(set([0, 1, n // 3, n // 2 + 1, n]))
    out = b""
    for a, b in zip(cuts[:-1], cuts[1:]):
        if b > a:
            coder.encode(feed[a:b], labels[a:b] if opts["k"] > 1 else None)
            out += coder.take()
    coder.finish()
    out += coder.take()
    q = G.gz(name + ".q.gz")
    size = G.CASES[name]["size"]
    assert len(out) == size == coder.total_bytes()
    assert out == q[-size:], "arithmetic stream differs from the reference's"
    # and the part in front of it: header + codebooks + 128-byte WELL state
    from qvz_amd import kernels as K
    blob = K.write_header(opts["k"], c, n) + b"".join(cb.serialize() for cb in cbs) + O.WELL_DEBUG_STATE.tobytes()
    assert blob + out == q


def records_from_feed(feed, labels, ocbs, k):
    """The records the GPU model leaves for one stream (include/qvz_coder.h), made here by a plain serial model in Python
    integers: floor(2^62 * c / n) + 1 for c = cum_lo, cum_hi; layout ((row/2)*(pitch/4) + pos/4)*8 + (row%2)*4 + pos%4."""
    n, c = feed.shape
    pos0 = 8 if k > 1 else 0
    pitch = (pos0 + c + 7) & ~7
    rec = np.zeros((((n + 1) // 2) * 2 * pitch, 2), dtype=np.uint64)
    tables = {}

    def step(key, card, x):
        t = tables.setdefault(key, [card, [1] * card])
        tot, cnts = t
        lo, f = sum(cnts[:x]), cnts[x]
        out = ((lo << 62) // tot + 1, ((lo + f) << 62) // tot + 1)
        cnts[x] += 8
        t[0] += 8
        if t[0] > (1 << 19):                                        # update_stats rescale, qv_stream.c:15-24
            t[1] = [(v >> 1) + 1 if v else 0 for v in cnts]
            t[0] = sum(t[1])
        return out

    def at(row, pos):
        return ((row >> 1) * (pitch >> 2) + (pos >> 2)) * 8 + (row & 1) * 4 + (pos & 3)

    for i in range(n):
        cl = int(labels[i]) if k > 1 else 0
        if k > 1:
            rec[at(i, 0)] = step(("cluster",), k, cl)
        for s_ in range(c):
            qi, st = int(feed[i, s_]) & 0xFF, int(feed[i, s_]) >> 8
            rec[at(i, pos0 + s_)] = step((cl, s_, qi), int(ocbs[cl].nstates[s_, qi]), st)
    return rec, pitch


@pytest.mark.parametrize("name", ["t2k_f05_L", "t2k_c3_f05_M", "const_r05", "tiny5_r10", "uniform72_r20_A", "rescale_140k_c4"])
def test_host_interval_chain_over_model_records(name):
    """Strict single-stream mode, host half: qvz_coder_encode_records over the fixed-point (cum_lo/n, cum_hi/n) records equals
    qvz_coder_encode over the feed, i.e. the reference's stream - including a stream long enough for update_stats to
    rescale, pending-bit runs (constant input) and chunked calls with an odd number of reads."""
    from qvz_amd import host
    rows, opts, labels, ocbs, ref = oracle_side(name)
    n, c = rows.shape[0], rows.shape[1] - 1
    cbs = [product_codebook_from_oracle(o) for o in ocbs]
    feed = (ref["qi"].astype(np.uint16) | (ref["st"].astype(np.uint16) << 8))
    coder = host.Coder(c, [cb.handle for cb in cbs])
    cuts = sorted(set([0, 1, (n // 3) | 1, n]))                     # chunks are laid out on their own: rows restart at 0
    state = {}
    out = b""
    for a, b in zip(cuts[:-1], cuts[1:]):
        if b <= a:
            continue
        rec_all, pitch = state.get("rec") or records_from_feed(feed, labels, ocbs, opts["k"])
        state["rec"] = (rec_all, pitch)
        # re-lay the chunk's rows from row 0
        pos0 = 8 if opts["k"] > 1 else 0
        m = b - a
        chunk = np.zeros((((m + 1) // 2) * 2 * pitch, 2), dtype=np.uint64)
        for i in range(m):
            for pos in ([0] if pos0 else []) + list(range(pos0, pos0 + c)):
                src = (((a + i) >> 1) * (pitch >> 2) + (pos >> 2)) * 8 + ((a + i) & 1) * 4 + (pos & 3)
                dst = ((i >> 1) * (pitch >> 2) + (pos >> 2)) * 8 + (i & 1) * 4 + (pos & 3)
                chunk[dst] = rec_all[src]
        coder.encode_records(chunk, m, pitch)
        out += coder.take()
    coder.finish()
    out += coder.take()
    assert out == ref["stream"], "interval chain over records differs from the reference's stream"


def test_coder_rejects_undefined_feed():
    from qvz_amd import host
    rows, opts, labels, ocbs, ref = oracle_side("t2k_f05_L")
    cbs = [product_codebook_from_oracle(o) for o in ocbs]
    feed = (ref["qi"].astype(np.uint16) | (ref["st"].astype(np.uint16) << 8))[:10].copy()
    feed[3, 5] = 143            # quantizer index beyond 2*nctx
    with pytest.raises(Exception):
        host.Coder(rows.shape[1] - 1, [cb.handle for cb in cbs]).encode(feed)
    feed[3, 5] = 0 | (71 << 8)  # state beyond the output alphabet
    with pytest.raises(Exception):
        host.Coder(rows.shape[1] - 1, [cb.handle for cb in cbs]).encode(feed)


def test_coder_long_pending_run_and_alignment():
    """const input: every symbol has probability ~1, so E3 underflow runs and the final padding byte are exercised."""
    for name in ("const_f05", "const_r05", "f00_zero_rate"):
        test_coder_reproduces_reference_stream(name)


def test_quantizer_tables_from_host_builder():
    from qvz_amd._lib import lib
    rows, opts, labels, ocbs, ref = oracle_side("t2k_c3_f05_M")
    for o in ocbs:
        cb = product_codebook_from_oracle(o)
        ct, qs = cb.tables()
        for s in range(o.c):
            for p in range(72):
                ctx = int(o.ctx_idx[s, p])
                want = 0xFF if ctx == 0xFF else ctx | (int(o.qratio[s, ctx]) << 8)
                assert int(ct[s, p]) == want
            for qi in range(2 * int(o.nctx[s])):
                qv = o.q[s, qi].astype(np.int64)
                want = o.q[s, qi].astype(np.uint16) | (o.out_idx[s, qi][qv].astype(np.uint16) << 8)
                np.testing.assert_array_equal(qs[s, qi], want)
        # every entry, the quantizers a column does not have included (rows of one value, which the converter writes without looking
        # each entry up), by the same formula over the product's own arrays - into a buffer that was NOT zero before
        qv = cb.q.astype(np.int64)
        want = cb.q.astype(np.uint16) | (np.take_along_axis(cb.out_idx, qv, axis=2).astype(np.uint16) << 8)
        ct2 = np.full((o.c, 72), 0xABCD, dtype=np.uint16)
        qs2 = np.full((o.c, 144, 72), 0xABCD, dtype=np.uint16)
        lib().qvz_codebook_tables(cb.handle, ct2.ctypes.data, qs2.ctypes.data)
        np.testing.assert_array_equal(qs2, want)
        np.testing.assert_array_equal(ct2, ct)
        dead = qs2[:, 2 * int(o.nctx.max()):]
        assert dead.size and (dead == dead[:, :, :1]).all()      # (one value per row: the quantizer does not exist)


def make_fastq(rows, seed=0, crlf=False, final_newline=True):
    """FASTQ text around the quality rows: headers of varying length, '+' or '+name' third lines, random bases."""
    rng = np.random.default_rng(seed)
    n, c = rows.shape[0], rows.shape[1] - 1
    eol = b"\r\n" if crlf else b"\n"
    out = []
    for i in range(n):
        name = b"@read%d/%s len=%d" % (i, b"x" * int(rng.integers(0, 40)), c)
        seq = bytes(rng.choice(np.frombuffer(b"ACGTN", dtype=np.uint8), size=c))
        plus = b"+" if i % 3 else b"+" + name[1:]
        out += [name, eol, seq, eol, plus, eol, rows[i, :c].tobytes(), eol]
    if not final_newline:
        out.pop()
    return b"".join(out)


@pytest.mark.parametrize("n,c,crlf,final_newline", [(7, 12, False, True), (300, 40, True, False), (9000, 150, False, True)])
def test_fastq_front_end_extracts_the_quality_lines(n, c, crlf, final_newline):
    """qvz_fastq_detect / qvz_fastq_qualities: the rows handed on are exactly the quality strings,
    also when they begin with '@' or '+', with CRLF line ends, without a final newline, and across the four scanner
    threads (inputs above 1 MiB)."""
    import ctypes as C
    from qvz_amd._lib import lib
    rows = O.qgen1_rows(21, n, c).copy()
    rows[::5, 0] = ord("@")                                          # quality strings that look like headers ...
    rows[1::7, 0] = ord("+")                                         # ... or like separator lines
    fq = make_fastq(rows, seed=n, crlf=crlf, final_newline=final_newline)
    buf = np.frombuffer(fq, dtype=np.uint8)
    assert lib().qvz_fastq_detect(buf.ctypes.data, buf.size) == 1
    plain = np.ascontiguousarray(rows).reshape(-1)
    assert lib().qvz_fastq_detect(plain.ctypes.data, plain.size) == 0     # a quality file is not FASTQ
    out, reads, cols = C.c_void_p(), C.c_uint64(), C.c_uint32()
    assert lib().qvz_fastq_qualities(buf.ctypes.data, buf.size, C.byref(out), C.byref(reads), C.byref(cols)) == 0
    assert (reads.value, cols.value) == (n, c)
    got = np.ctypeslib.as_array(C.cast(out, C.POINTER(C.c_uint8)), shape=(n * (c + 1),)).copy()
    C.CDLL(None).free(out)
    assert got.tobytes() == rows.tobytes()


def test_fastq_front_end_refuses_ragged_reads():
    import ctypes as C
    from qvz_amd._lib import lib
    rows = O.qgen1_rows(22, 50, 20)
    fq = make_fastq(rows)
    lines = fq.split(b"\n")
    lines[4 * 17 + 1] = lines[4 * 17 + 1][:-3]                       # read 17 is three bases / qualities short
    lines[4 * 17 + 3] = lines[4 * 17 + 3][:-3]
    bad = np.frombuffer(b"\n".join(lines), dtype=np.uint8)
    out, reads, cols = C.c_void_p(), C.c_uint64(), C.c_uint32()
    assert lib().qvz_fastq_qualities(bad.ctypes.data, bad.size, C.byref(out), C.byref(reads), C.byref(cols)) != 0
    assert b"read length differs" in lib().qvz_encode_last_error()


def test_fastq_front_end_refuses_a_bad_record_at_a_thread_boundary():
    """A malformed record right where a scanner thread's byte range begins must be an error too, not silently skipped by
    that thread's resynchronisation (inputs above 1 MiB are scanned by several threads)."""
    import ctypes as C
    from qvz_amd._lib import lib
    n, c = 9000, 150
    rows = O.qgen1_rows(23, n, c)
    fq = make_fastq(rows, seed=5)
    assert len(fq) > (1 << 20)
    nt = min(4, os.cpu_count() or 1)
    out, reads, cols = C.c_void_p(), C.c_uint64(), C.c_uint32()
    hits = 0
    for i in range(1, nt):
        lo = len(fq) // nt * i                                        # the range start qvz_fastq_qualities uses
        rec = fq.index(b"\n@read", lo) + 1                            # first record starting at or after it
        end_hdr = fq.index(b"\n", rec)
        seq_end = fq.index(b"\n", end_hdr + 1)
        bad = fq[:seq_end - 2] + fq[seq_end:]                         # that record's sequence loses two bases
        buf = np.frombuffer(bad, dtype=np.uint8)
        rc = lib().qvz_fastq_qualities(buf.ctypes.data, buf.size, C.byref(out), C.byref(reads), C.byref(cols))
        assert rc != 0, f"record at the start of range {i} was dropped silently ({reads.value} reads)"
        hits += 1
    if nt > 1:
        assert hits


def test_distortion_matrix_file_parser(tmp_path, capfd):
    """qvz_distortion_matrix_file = gen_custom_distortion (src/distortion.c:100-145): comma-separated atof fields,
    row x column y -> d[x + 72*y], '#' lines skipped, short rows / missing rows are 0.0 with the reference's warning."""
    import ctypes as C
    from qvz_amd._lib import lib
    rng = np.random.default_rng(3)
    m = np.round(rng.random((72, 72)) * 50, 3)
    text = ["# a comment"] + [",".join(repr(float(v)) for v in m[x]) for x in range(72)]
    text[5] = ",".join(text[5].split(",")[:10])                     # row 4 keeps 10 entries
    text = text[:60]                                                # rows 59.. are missing
    f = tmp_path / "d.txt"
    f.write_text("\n".join(text) + "\n")
    d = np.full(72 * 72, -1.0)
    assert lib().qvz_distortion_matrix_file(str(f).encode(), d.ctypes.data_as(C.POINTER(C.c_double))) == 0
    want = np.zeros((72, 72))
    want[:59] = m[:59]
    want[4, 10:] = 0.0
    np.testing.assert_array_equal(d.reshape(72, 72).T, want)       # d[x + 72*y]
    C.CDLL(None).fflush(None)                                       # the library prints through C stdio
    assert "Warning: one or more entries in the distortion matrix on line 4 were missing" in capfd.readouterr().out
    assert lib().qvz_distortion_matrix_file(str(tmp_path / "none.txt").encode(), d.ctypes.data_as(C.POINTER(C.c_double))) != 0


def test_distortion_matrix_host():
    from qvz_amd import kernels as K
    for kind in "MLA":
        assert K.distortion_matrix(kind).tobytes() == O.distortion(kind).tobytes()


def test_glibc_log2_restatement_equals_this_hosts_log2(tmp_path):
    """qvz_amd/csrc/qvz_log2_glibc.h (what the GPU evaluates entropies with) built for the host: bit-identical to libm's
    log2 on 2e7 arguments of the kinds the pmfs hold - count / total quotients, products, the close-to-1 branch and its
    edges, powers of two, arbitrary mantissas (1e8 when it was written). The constants are the library's own
    (tools/extract_glibc_log2.py); the product repeats this check on the device before it lets device entropies decide."""
    import subprocess
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    exe = str(tmp_path / "log2_check")
    subprocess.run(["gcc", "-O2", "-ffp-contract=off", "-std=gnu11", "-I", os.path.join(root, "qvz_amd", "csrc"), "-o", exe,
                    os.path.join(root, "tests", "log2_check.c"), "-lm"], check=True)
    for seed in (1, 2026):
        r = subprocess.run([exe, "10000000", str(seed)], capture_output=True, text=True)
        assert r.returncode == 0 and " 0 mismatches" in r.stdout, r.stdout[-2000:]


def test_parallel_file_write_and_copy_helpers(tmp_path):
    """The host halves of the slab pipelines (host/qvz_egress.c): a buffer written into a file at an offset by several threads
    lands byte for byte where one pwrite would put it (sizes below and above the threading threshold, an odd tail), and the
    threaded copy equals memcpy."""
    from qvz_amd._lib import lib
    L = lib()
    L.qvz_parallel_pwrite.argtypes = [C.c_int, C.c_void_p, C.c_size_t, C.c_int64]
    L.qvz_parallel_pwrite.restype = C.c_int
    L.qvz_parallel_memcpy.argtypes = [C.c_void_p, C.c_void_p, C.c_size_t]
    L.qvz_parallel_memcpy.restype = None
    rng = np.random.default_rng(5)
    path = str(tmp_path / "w.bin")
    with open(path, "wb") as f:
        f.write(b"\xAA" * 1000)
    fd = os.open(path, os.O_RDWR)
    try:
        at = 1000
        expect = bytearray(b"\xAA" * 1000)
        for n in (1, 4097, (8 << 20) - 1, (24 << 20) + 12345):
            buf = rng.integers(0, 256, size=n, dtype=np.uint8)
            assert L.qvz_parallel_pwrite(fd, buf.ctypes.data, n, at) == 0
            expect += buf.tobytes()
            at += n
            dst = np.zeros(n, dtype=np.uint8)
            L.qvz_parallel_memcpy(dst.ctypes.data, buf.ctypes.data, n)
            assert np.array_equal(dst, buf)
    finally:
        os.close(fd)
    assert open(path, "rb").read() == bytes(expect)


@pytest.mark.parametrize("estimate_of", ["exact", "too_much", "too_little", "none", "too_much_and_no_helper"])
def test_big_outputs_through_pages_made_ahead(tmp_path, estimate_of, monkeypatch):
    """host/qvz_egress.c: a big output whose size is known (or estimated) early gets its pages from a helper thread's
    fallocate and is written through a mapping, never ahead of the helper; what lies beyond an estimate goes through pwrite,
    what an estimate allocated too much is given back, and the file ends where the last byte went. The same bytes and the same
    length as a plain write, behind a header that stdio wrote first."""
    from qvz_amd._lib import lib
    L = lib()
    libc = C.CDLL(None)
    libc.fopen.restype = C.c_void_p
    libc.fopen.argtypes = [C.c_char_p, C.c_char_p]
    libc.fwrite.argtypes = [C.c_void_p, C.c_size_t, C.c_size_t, C.c_void_p]
    libc.fclose.argtypes = [C.c_void_p]
    L.qvz_host_write_slabs.argtypes = [C.c_void_p, C.c_void_p, C.c_uint64, C.c_uint64, C.c_uint64]
    L.qvz_host_write_slabs.restype = C.c_int
    monkeypatch.setenv("QVZ_PREALLOC", "min=1")
    rng = np.random.default_rng(11)
    head = b"header written by stdio\n" * 3                     # not page aligned: the mapping starts inside a page
    n = (21 << 20) + 4321
    body = rng.integers(0, 256, size=n, dtype=np.uint8)
    estimate = {"exact": n, "too_much": n + (9 << 20) + 17, "too_little": (5 << 20) + 3, "none": 0,
                "too_much_and_no_helper": n + (9 << 20) + 17}[estimate_of]
    if estimate_of == "too_much_and_no_helper":
        # the helper thread / table slot cannot be had AFTER the file was made longer for the mapping: the length is given
        # back at once (nobody would later - the over-estimate's zero bytes would stay behind the shard set)
        monkeypatch.setenv("QVZ_PREALLOC", "min=1,fail")
    path = str(tmp_path / "out.bin")
    f = libc.fopen(path.encode(), b"w")
    assert f
    assert libc.fwrite(head, 1, len(head), f) == len(head)
    assert L.qvz_host_write_slabs(f, body.ctypes.data, n, estimate, 3 << 20) == 0
    tail = b"and a trailer through stdio"
    assert libc.fwrite(tail, 1, len(tail), f) == len(tail)
    libc.fclose(f)
    got = open(path, "rb").read()
    assert len(got) == len(head) + n + len(tail)
    assert got[:len(head)] == head and got[len(head):len(head) + n] == body.tobytes() and got[len(head) + n:] == tail


def test_coder_segment_cut():
    """The cut of a stream into the coder's segments (qvz_coder_segment_list, the rule of qvz_hip_encode_planes): the segments
    tile the stream in order, none is longer than the bound, every start is a whole 64-read chunk; with overlapping kernels the
    first and the last are short (a quarter of the bound) and the middle ones differ by one chunk at most - no short left-over
    segment, which would cost every model workgroup a table save and load for a handful of chunks."""
    from qvz_amd._lib import lib
    L = lib()
    rng = np.random.default_rng(3)
    cases = [(12208, 1536, 1), (12208, 2432, 1), (1224, 192, 1), (12208, 2560, 0), (100, 64, 1), (255, 256, 1), (70000, 2432, 1)]
    cases += [(int(rng.integers(1, 200000)), int(rng.integers(1, 80)) * 64, int(rng.integers(0, 2))) for _ in range(300)]
    for longest, seg, overlap in cases:
        first = np.zeros(4096, dtype=np.uint64)
        reads = np.zeros(4096, dtype=np.uint64)
        n = L.qvz_coder_segment_list(longest, seg, overlap, first.ctypes.data, reads.ctypes.data, 4096)
        assert 0 < n <= 4096, (longest, seg, overlap, n)
        f, r = first[:n].astype(np.int64), reads[:n].astype(np.int64)
        assert f[0] == 0 and (f[1:] == f[:-1] + r[:-1]).all() and f[-1] + r[-1] == longest, (longest, seg, overlap)
        assert (r > 0).all() and (r <= seg).all() and (f % 64 == 0).all(), (longest, seg, overlap, r)
        if overlap and seg >= 256 and longest > seg:
            q = (seg // 4) // 64 * 64
            assert r[0] == q and q <= r[-1] < q + 64, (longest, seg, r)
            mid = r[1:-1]
            assert len(mid) == 0 or (mid.max() - mid.min() <= 64 and (mid % 64 == 0).all()), (longest, seg, mid)
        else:
            assert (r[:-1] == seg).all(), (longest, seg, r)
        # several clusters: a full first segment (the model kernels bound the phase: no short segment in front), the short one last
        n1 = L.qvz_coder_segment_list2(longest, seg, overlap, 1, first.ctypes.data, reads.ctypes.data, 4096)
        f, r = first[:n1].astype(np.int64), reads[:n1].astype(np.int64)
        assert 0 < n1 <= n and f[0] == 0 and (f[1:] == f[:-1] + r[:-1]).all() and f[-1] + r[-1] == longest
        assert (r > 0).all() and (r <= seg).all() and (f % 64 == 0).all(), (longest, seg, overlap, r)
        if overlap and seg >= 256 and longest > seg:
            q = (seg // 4) // 64 * 64
            assert r[0] == min(seg // 64 * 64, f[-1]) and q <= r[-1] < q + 64, (longest, seg, r)
            # the middle: equal segments, then one of about 0.72 of them (its coder kernel must not outlast the short last model kernel)
            mid = r[1:-1]
            if len(mid) > 1:
                assert (mid[:-1] == mid[0]).all() and 0 < mid[-1] <= mid[0], (longest, seg, mid)
                if mid[0] < seg // 64 * 64:                      # not capped by the buffer: the ratio is the rule's
                    assert 0.72 * mid[0] - 64 * len(mid) <= mid[-1] <= 0.72 * mid[0] + 64, (longest, seg, mid)

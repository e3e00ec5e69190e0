"""GPU tests of the shard decoder (qvz_hip_decode_shards): it must decode the reference
encoder's own streams to the reference's `-u` text, and invert the GPU coders shard by shard."""
import numpy as np
import pytest

import golden_util as G
import oracle_lib as O

pytestmark = pytest.mark.gpu
torch = pytest.importorskip("torch")

from test_gpu_pipeline import setup_case  # noqa: E402


def encode(name, shards, coder="gpu"):
    from qvz_amd import pipeline as P
    rows, opts, n, c, sym, means0 = setup_case(name)
    r = P.encode_pass(sym, n, 0, n, means0, clusters=opts["k"], mode=opts["mode"], ratio=opts["ratio"], dist=opts["dist"],
                      shards=shards, threads=2, keep=True, coder=coder)
    return rows, opts, n, c, r


def upload_streams(streams):
    cap = (max(len(s) for s in streams) + 7) & ~3
    buf = np.zeros((len(streams), cap), dtype=np.uint8)
    for g, s in enumerate(streams):
        buf[g, : len(s)] = np.frombuffer(s, dtype=np.uint8)
    return torch.from_numpy(buf).cuda().view(-1), cap


@pytest.mark.parametrize("name", ["t2k_r10_M", "t2k_f05_L", "t2k_c3_f05_M", "cfg1_r10_M", "const_f05", "const_r05",
                                  "uniform72_r20_A", "uniform72_f09_M", "c2_f08", "tiny5_r10", "twolevel_c2_f05",
                                  "f00_zero_rate", "r35_high_rate"])
def test_decodes_the_reference_encoders_stream(name):
    """Stream bytes straight out of the reference's .q file -> GPU decoder -> the reference's -u text."""
    from qvz_amd import host, kernels as K
    rows, opts, n, c, r = encode(name, 1, coder="host")          # codebooks only (bit-identical to the file's, tested elsewhere)
    e = G.CASES[name]
    stream = G.gz(name + ".q.gz")[-e["size"]:]
    coded, cap = upload_streams([stream])
    draws = K.well_draws(O.WELL_DEBUG_STATE, 0, n, c)
    qv, labels = host.decode_shards_gpu(coded, cap, [len(stream)], n, c, [cb.handle for cb in r["codebooks"]], 1, draws)
    text = K.untranspose_text(qv, n).cpu().numpy().tobytes()
    assert text == G.gz(name + ".u.gz")
    if opts["k"] > 1:
        assert torch.equal(labels[:n], r["labels"][:n])


@pytest.mark.parametrize("name,shards", [("t2k_r10_M", 3), ("t2k_c3_f05_M", 5), ("cfg1_f05_L", 64), ("cfg1_c3_r10_M", 17),
                                         ("wide_c151", 2), ("tiny5_r10", 2), ("rescale_140k_c4", 1), ("rescale_c2_100k_c3", 2),
                                         ("r35_high_rate", 4)])
def test_round_trip_of_gpu_coded_shards(name, shards):
    from qvz_amd import host, kernels as K
    rows, opts, n, c, r = encode(name, shards)
    coded, cap = upload_streams(r["streams"])
    draws = K.well_draws(O.WELL_DEBUG_STATE, 0, n, c)
    qv, labels = host.decode_shards_gpu(coded, cap, r["stream_bytes"], n, c, [cb.handle for cb in r["codebooks"]], shards, draws)
    assert torch.equal(qv[:, :n], r["qv"][:, :n])
    if opts["k"] > 1:
        assert torch.equal(labels[:n], r["labels"][:n])
    assert G.md5(K.untranspose_text(qv, n).cpu().numpy().tobytes()) == G.CASES[name]["u_md5"]


def test_round_trip_with_more_clusters_than_a_table_header_holds():
    """Fourteen clusters: the cluster-id table has more states than the twelve counts that come with a table's header (the search
    goes on in memory), its own values 8 .. 13 lie behind the counts. Round trip of the GPU coder's streams, and every shard's
    file - header, codebooks, WELL state, stream - decodes with the ORACLE's decoder to the same text."""
    from qvz_amd import host, kernels as K, pipeline as P
    n, c, k, shards = 48_000, 36, 14, 6
    sym = K.qgen1(11, n, c)
    means0 = torch.stack([sym[:, 977 * j + 5] for j in range(k)]).contiguous()
    r = P.encode_pass(sym, n, 0, n, means0, clusters=k, mode=host.MODE_RATIO, ratio=0.6, dist="M", shards=shards, coder="gpu", keep=True)
    assert len(set(r["labels"][:n].cpu().numpy().tolist())) == k, "every cluster should own reads"
    coded, cap = upload_streams(r["streams"])
    draws = K.well_draws(O.WELL_DEBUG_STATE, 0, n, c)
    qv, labels = host.decode_shards_gpu(coded, cap, r["stream_bytes"], n, c, [cb.handle for cb in r["codebooks"]], shards, draws)
    assert torch.equal(qv[:, :n], r["qv"][:, :n]) and torch.equal(labels[:n], r["labels"][:n])
    text = K.untranspose_text(qv, n).cpu().numpy().tobytes()
    first = host.shard_bounds(n, shards)
    books = b"".join(cb.serialize() for cb in r["codebooks"])
    for g in range(shards):
        image = (bytes([k]) + c.to_bytes(4, "big") + (first[g + 1] - first[g]).to_bytes(4, "big") + books +
                 K.well_jump(O.WELL_DEBUG_STATE, first[g] * c // 4).tobytes() + bytes(r["streams"][g]))
        assert O.decode(image) == text[first[g] * (c + 1):first[g + 1] * (c + 1)], f"shard {g}"


def test_decoder_rejects_a_foreign_stream():
    from qvz_amd import host, kernels as K
    rows, opts, n, c, r = encode("t2k_r10_M", 2)
    junk = [bytes([0xFF]) * len(s) for s in r["streams"]]
    coded, cap = upload_streams(junk)
    draws = K.well_draws(O.WELL_DEBUG_STATE, 0, n, c)
    with pytest.raises(Exception, match="does not decode"):
        host.decode_shards_gpu(coded, cap, [len(s) for s in junk], n, c, [cb.handle for cb in r["codebooks"]], 2, draws)


@pytest.mark.parametrize("name,shards", [("t2k_c3_f05_M", 5), ("t2k_r10_M", 3)])
def test_decoder_survives_mutated_streams(name, shards):
    """Streams somebody damaged: bit flips, zeroed and random runs, truncated and over-stated lengths, streams swapped
    between shards, a tag of all ones under the whole interval. A damaged stream still has ONE decode - the arithmetic is exact
    whatever the bits are -: every stream must decode to what the ORACLE's decoder (= the reference's) reads out of it, or be
    refused where the oracle gives up (a value that is no context). The decoder never reads
    outside its tables or its stream (the kernel takes every table offset from the validated codebooks, checks every state
    against the table's size and reads zeros behind a stream's stated end, as the reference's stream reader does,
    os_stream.c:36-52). The untouched streams decode to the quantizer's output before and after."""
    from qvz_amd import host, kernels as K
    rows, opts, n, c, r = encode(name, shards)
    handles = [cb.handle for cb in r["codebooks"]]
    draws = K.well_draws(O.WELL_DEBUG_STATE, 0, n, c)
    rng = np.random.default_rng(20260403)
    good = [bytes(s) for s in r["streams"]]
    outcomes = {"rejected": 0, "decoded": 0}

    first = host.shard_bounds(n, shards)
    books = b"".join(cb.serialize() for cb in r["codebooks"])
    # the codebooks as a READER has them (qvz_codebook_read: every context a previous value can name has its tables, as in the
    # reference's decoder; the encoder's own leave out contexts no read of the input reaches, and a damaged stream may reach one)
    import ctypes as C
    from qvz_amd._lib import lib
    parsed, at = [], C.c_size_t(0)
    image = np.frombuffer(books, dtype=np.uint8)
    for _ in r["codebooks"]:
        cb = K.Codebook(c)
        assert lib().qvz_codebook_read(image.ctypes.data, image.size, C.byref(at), cb.handle) == 0
        parsed.append(cb)
    handles = [cb.handle for cb in parsed]

    def oracle_says(g, stream):
        """what the ORACLE's decoder (= the reference's, start_qv_decompression) reads out of this stream as shard g's file:
        the text, or None where it meets a value that is no context / a cluster that does not exist"""
        lines = first[g + 1] - first[g]
        image = (bytes([opts["k"]]) + c.to_bytes(4, "big") + lines.to_bytes(4, "big") + books +
                 K.well_jump(O.WELL_DEBUG_STATE, first[g] * c // 4).tobytes() + bytes(stream))
        try:
            return O.decode(image)
        except AssertionError:
            return None

    def run(streams, sizes=None):
        coded, cap = upload_streams(streams)
        sz = sizes or [len(x) for x in streams]
        # a damaged stream still has ONE decode - the arithmetic is exact whatever the bits are -: the oracle's
        want = [oracle_says(g, bytes(streams[g][:sz[g]])) for g in range(shards)]   # (an over-stated size reads the slot's zeros: what the oracle reads behind a stream's end)
        try:
            qv, labels = host.decode_shards_gpu(coded, cap, sz, n, c, handles, shards, draws)
        except Exception as e:                                   # noqa: BLE001
            assert "does not decode" in str(e) or "longer than cap" in str(e), e
            assert None in want or "longer than cap" in str(e), "the oracle decodes every stream of a set the GPU decoder refused"
            outcomes["rejected"] += 1
            return None
        torch.cuda.synchronize()
        assert int(qv[:, :n].max()) < 72
        assert None not in want, "the GPU decoder accepted a stream the oracle's decoder gives up on"
        assert K.untranspose_text(qv, n).cpu().numpy().tobytes() == b"".join(want), "a damaged stream decodes differently"
        outcomes["decoded"] += 1
        return qv

    assert torch.equal(run(good)[:, :n], r["qv"][:, :n])
    for trial in range(48):
        streams = [bytearray(x) for x in good]
        g = int(rng.integers(shards))
        kind = trial % 8
        sizes = None
        if kind == 0:                                            # single bit flips, early in the stream (everything after depends on them)
            for _ in range(int(rng.integers(1, 4))):
                pos = int(rng.integers(min(64, len(streams[g]))))
                streams[g][pos] ^= 1 << int(rng.integers(8))
        elif kind == 1:                                          # ... anywhere
            pos = int(rng.integers(len(streams[g])))
            streams[g][pos] ^= 1 << int(rng.integers(8))
        elif kind == 2:                                          # a zeroed run
            a = int(rng.integers(len(streams[g])))
            streams[g][a:a + 37] = bytes(len(streams[g][a:a + 37]))
        elif kind == 3:                                          # random bytes from some point to the end
            a = int(rng.integers(len(streams[g])))
            streams[g][a:] = rng.integers(0, 256, len(streams[g]) - a, dtype=np.uint8).tobytes()
        elif kind == 4:                                          # truncated: the stated size ends early (zeros are read behind it)
            sizes = [len(x) for x in streams]
            sizes[g] = int(rng.integers(0, len(streams[g])))
        elif kind == 5:                                          # over-stated size: up to the slot's capacity
            sizes = [len(x) for x in streams]
            sizes[g] = (max(sizes) + 7) & ~3
        elif kind == 6:                                          # two shards' streams swapped
            h = (g + 1) % shards
            streams[g], streams[h] = streams[h], streams[g]
        else:                                                    # all ones / all zeros; or only the first bytes: tag = 2^22 - 1 under the whole interval
            if trial & 16:
                streams[g][:3 + trial % 3] = bytes([0xFF]) * (3 + trial % 3)
            else:
                streams[g] = bytearray([0xFF if trial & 8 else 0x00]) * len(streams[g])
        run([bytes(x) for x in streams], sizes)
    assert outcomes["decoded"] + outcomes["rejected"] == 49 and outcomes["decoded"] > 40, outcomes   # (rejection itself: test_decoder_rejects_a_foreign_stream)
    assert torch.equal(run(good)[:, :n], r["qv"][:, :n])


def test_round_trip_at_bench_stream_length():
    """12 208 reads per stream x 150 columns (the per-stream shape of BASELINE configs[1] at 8192 streams), 256 streams."""
    from qvz_amd import host, kernels as K, pipeline as P
    n, c, shards = 3_125_248, 150, 256
    sym = K.qgen1(3, n, c)
    means0 = K.qgen1(3, 4, c)[:, 0].contiguous().view(1, c)
    r = P.encode_pass(sym, n, 0, n, means0, mode=host.MODE_RATIO, ratio=0.5, dist="L", shards=shards, coder="gpu", keep=True)
    cap = (max(r["stream_bytes"]) + 7) & ~3
    coded = torch.zeros(shards * cap, dtype=torch.uint8, device="cuda").view(shards, cap)
    for g, s in enumerate(r["streams"]):
        coded[g, : len(s)] = torch.from_numpy(np.frombuffer(s, dtype=np.uint8).copy()).cuda()
    draws = K.well_draws(np.full(32, 0x55555555, dtype=np.uint32), 0, n, c)
    qv, _ = host.decode_shards_gpu(coded.view(-1), cap, r["stream_bytes"], n, c, [cb.handle for cb in r["codebooks"]], shards, draws)
    assert torch.equal(qv[:, :n], r["qv"][:, :n])


@pytest.mark.parametrize("name,shards", [("t2k_r10_M", 3), ("t2k_c3_f05_M", 7)])
def test_shard_files_cut_by_another_formula_decode(tmp_path, name, shards):
    """Shard FILES carry their own line counts, so `qvz -x` must decode them however they were cut - an earlier revision
    dealt ceil(N / S) reads rounded up to 4 per shard, qvz_shard_bounds deals groups of 4 evenly. Streams coded over the old
    cut (qvz_hip_encode_planes_bounds): the GPU decoder inverts them given the same cut, the files decode through the
    command line, and each file is what the oracle's (= the reference's) decoder reads."""
    import os
    import subprocess
    from qvz_amd import host, kernels as K
    from test_gpu_encode import QVZ_BIN
    rows, opts, n, c, r = encode(name, 1, coder="gpu")
    per = ((n + shards - 1) // shards + 3) & ~3
    bounds = [min(g * per, n) for g in range(shards)] + [n]
    assert bounds != host.shard_bounds(n, shards)
    handles = [cb.handle for cb in r["codebooks"]]
    sizes, streams, coded = host.encode_planes_gpu(r["qi"], r["st"], r["labels"] if opts["k"] > 1 else None, n, c, handles, shards,
                                                   bounds=bounds)
    draws = K.well_draws(O.WELL_DEBUG_STATE, 0, n, c)
    qv, labels = host.decode_shards_gpu(coded, coded.numel() // shards, sizes, n, c, handles, shards, draws, bounds=bounds)
    assert torch.equal(qv[:, :n], r["qv"][:, :n])
    with pytest.raises(Exception):                                              # a cut that is not on WELL word boundaries is refused
        host.decode_shards_gpu(coded, coded.numel() // shards, sizes, n, c, handles, shards, draws, bounds=[0, 2] + bounds[2:])
    u = K.untranspose_text(r["qv"], n).cpu().numpy().tobytes()
    prefix = str(tmp_path / "old.q")
    for g in range(shards):
        state = K.well_jump(O.WELL_DEBUG_STATE, bounds[g] * c // 4)
        image = host.shard_file_image(c, bounds[g + 1] - bounds[g], handles, state, streams[g])
        open(f"{prefix}.{g:04d}", "wb").write(image)
        assert O.decode(image) == u[bounds[g] * (c + 1):bounds[g + 1] * (c + 1)]
    dec = str(tmp_path / "old.dec")
    subprocess.run([QVZ_BIN, "-x", prefix, dec], env=dict(os.environ, QVZ_SHARDS=str(shards), QVZ_SHARD_FILES="1"), check=True,
                   capture_output=True)
    assert open(dec, "rb").read() == u

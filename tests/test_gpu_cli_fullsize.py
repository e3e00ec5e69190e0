"""The reference's encode() sequence (src/main.c:18-127) through the C boundary at BASELINE size, file to file:
`qvz -s -u out.u <flags> in.txt out.qset` on the 100M x 150 QGEN-1 text in a tmpfs file (configs[2] and configs[3] flags,
QVZ_WELL_SEED=debug, QVZ_SHARDS=8192) - the binary a user of the reference would run, driven by host C
(host/qvz_encode.c), not by the Python plumbing bench.py times. Checked:
  (a) the shard set's codebooks and all 8192 streams equal pipeline.encode_pass's (the path test_gpu_fullsize.py pins to
      the oracle) byte for byte, and the -u text equals its quantized plane;
  (b) `qvz -x out.qset` reproduces the -u text: the reference's one invariant (test.sh:7-9: decode == -u) at full size;
  (c) the distortion the binary prints is the file-order sequential sum (qv_compressor.c:127,139-140) of the per-read
      errors the quantizer produced.
The wall times of both directions are printed (pytest -s) and measured properly by bench.py's `e2e_cli`."""
import os
import re
import shutil
import subprocess
import time

import numpy as np
import pytest

pytestmark = pytest.mark.gpu
torch = pytest.importorskip("torch")

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
QVZ_BIN = os.path.join(ROOT, "qvz_amd", "bin", "qvz")
N, C, SHARDS, SEED = 100_000_000, 150, 8192, 3
CONFIGS = {
    "configs2_c1_L_f05": dict(clusters=1, mode="f", ratio=0.5, dist="L"),
    "configs3_c3_M_r10": dict(clusters=3, mode="r", ratio=1.0, dist="M"),
}


def _free_gpu():
    import gc
    from qvz_amd import host
    host.release_workspace()
    from qvz_amd._lib import lib
    lib().qvz_hip_encode_planes_release()
    gc.collect()
    torch.cuda.empty_cache()


@pytest.fixture(scope="module")
def workdir():
    base = "/dev/shm" if os.path.isdir("/dev/shm") else None
    need = 52 << 30                                  # input 15.1 GB + -u 15.1 + decoded 15.1 + shard set ~3
    if base is None or shutil.disk_usage(base).free < need:
        pytest.skip("needs ~52 GB of tmpfs under /dev/shm")
    avail = int(re.search(r"MemAvailable:\s+(\d+)", open("/proc/meminfo").read()).group(1)) << 10
    if avail < need + (16 << 30):
        pytest.skip("needs ~70 GB of free host memory")
    d = os.path.join(base, f"qvz_cli_fullsize_{os.getpid()}")
    os.makedirs(d, exist_ok=True)
    try:
        yield d
    finally:
        shutil.rmtree(d, ignore_errors=True)


@pytest.fixture(scope="module")
def input_text(workdir):
    """The QGEN-1 matrix as the text file load_file would map (lines.c:27-82): written once, shared by both configurations."""
    from qvz_amd import host, kernels as K
    _free_gpu()
    free, _ = torch.cuda.mem_get_info()
    if free < 250 * 2 ** 30:
        pytest.skip(f"needs ~250 GB of HBM, {free >> 30} GiB free")
    path = os.path.join(workdir, "in.txt")
    sym = K.qgen1(SEED, N, C)
    t = time.perf_counter()
    host.egress_rows(sym, N, path)
    print(f"\ninput text: {os.path.getsize(path) / 1e9:.1f} GB written in {time.perf_counter() - t:.2f} s")
    assert os.path.getsize(path) == N * (C + 1)
    # spot check against the oracle's generator: the first and the last rows
    import oracle_lib as O
    with open(path, "rb") as f:
        assert f.read(1000 * (C + 1)) == O.qgen1_rows(SEED, 1000, C).tobytes()
        f.seek((N - 1000) * (C + 1))
        assert f.read() == O.qgen1_rows(SEED, 1000, C, first=N - 1000).tobytes()
    del sym
    _free_gpu()
    return path


@pytest.mark.parametrize("name", sorted(CONFIGS))
def test_qvz_binary_file_to_file_at_full_size(name, workdir, input_text):
    from qvz_amd import host, kernels as K, pipeline as P
    cfg = CONFIGS[name]
    k = cfg["clusters"]
    qset, u, dec = (os.path.join(workdir, f"{name}.{ext}") for ext in ("qset", "u", "dec"))
    flags = ["-c", str(k), "-d", cfg["dist"], "-" + cfg["mode"], str(cfg["ratio"])]
    env = dict(os.environ, QVZ_WELL_SEED="debug", QVZ_SHARDS=str(SHARDS), QVZ_TIMING="1")
    _free_gpu()
    t = time.perf_counter()
    enc = subprocess.run([QVZ_BIN, "-s", "-u", u] + flags + [input_text, qset], env=env, capture_output=True, text=True)
    t_enc = time.perf_counter() - t
    assert enc.returncode == 0, enc.stdout[-2000:] + enc.stderr[-2000:]
    print(f"\n{name}: encode file -> file {t_enc:.2f} s; {enc.stderr.strip().splitlines()[-1]}")
    m = re.search(r"rate, ([\d.]+), distortion, ([\d.]+), time, ([\d.]+), size, (\d+)", enc.stdout)
    assert m, enc.stdout
    assert os.path.getsize(u) == N * (C + 1)

    # (b) decode == -u, through the binary
    t = time.perf_counter()
    subprocess.run([QVZ_BIN, "-x", qset, dec], env=dict(os.environ), check=True, capture_output=True)
    t_dec = time.perf_counter() - t
    print(f"{name}: decode file -> file {t_dec:.2f} s")
    assert subprocess.run(["cmp", "-s", dec, u]).returncode == 0, "qvz -x of the shard set differs from the -u text"
    os.remove(dec)
    if k == 1:
        # ... and in SLABS of whole streams, as a file whose matrices do not fit the device is decoded (2^32 reads and more:
        # always): three slabs of at most two fifths of the reads, the WELL generator advanced to each slab's first read
        from qvz_amd._lib import lib
        _free_gpu()
        lib().qvz_decode_set_slab(2 * N // 5, 0)
        try:
            t = time.perf_counter()
            assert lib().qvz_shardset_decode(qset.encode(), dec.encode(), 0, None) == 0, lib().qvz_encode_last_error()
            print(f"{name}: decode in three slabs {time.perf_counter() - t:.2f} s")
        finally:
            lib().qvz_decode_set_slab(0, 0)
        assert subprocess.run(["cmp", "-s", dec, u]).returncode == 0, "the decode in slabs differs from the -u text"
        os.remove(dec)

    # (a) the same encode through pipeline.encode_pass (pinned to the oracle by test_gpu_fullsize.py)
    sym = K.qgen1(SEED, N, C)
    mode = host.MODE_RATIO if cfg["mode"] == "f" else host.MODE_FIXED
    r = P.encode_pass(sym, N, 0, N, None, clusters=k, mode=mode, ratio=cfg["ratio"], dist=cfg["dist"], shards=SHARDS,
                      coder="gpu", keep_device=True)
    si = host.shardset_stat(qset)
    assert (si.clusters, si.columns, si.lines, si.shards) == (k, C, N, SHARDS)
    sizes = r["stream_bytes"]
    assert si.stream_bytes == sum(sizes) == int(m.group(4))
    head = 160 + 8 * SHARDS
    with open(qset, "rb") as f:
        blob = f.read(head)
        assert np.frombuffer(blob[160:], dtype="<u8").tolist() == sizes
        assert blob[32:160] == np.full(32, 0x55555555, dtype=np.uint32).tobytes()
        assert f.read(si.codebook_bytes) == b"".join(cb.serialize() for cb in r["codebooks"])
        packed = host.gather_streams(r["coded"], r["cap_per_shard"], sizes)
        step = 256 << 20
        for a in range(0, packed.numel(), step):
            mine = packed[a:a + step].cpu().numpy().tobytes()
            assert f.read(len(mine)) == mine, f"stream bytes differ near offset {a}"
        assert f.read() == b""
    del packed
    # ... and the -u text against the quantized plane, slab by slab on the device
    qv = r["qv"]
    slab = 1 << 21
    with open(u, "rb") as f:
        for a in range(0, N, slab):
            nrows = min(slab, N - a)
            text = torch.from_numpy(np.frombuffer(f.read(nrows * (C + 1)), dtype=np.uint8).reshape(nrows, C + 1).copy()).cuda()
            assert bool((text[:, C] == 10).all())
            assert torch.equal(text[:, :C].t() - 33, qv[:, a:a + nrows]), f"-u rows {a}.. differ from the quantizer's plane"
    # (c) distortion: the binary's is the sequential sum; encode_pass sums on the device (equal to ~1e-12 relative)
    assert abs(float(m.group(2)) - r["distortion"]) < 1e-4 + 1e-9 * r["distortion"]
    rate = si.stream_bytes * 8.0 / (N * C)
    assert abs(float(m.group(1)) - rate) < 1e-4
    os.remove(u)
    os.remove(qset)
    del r, sym, qv
    _free_gpu()

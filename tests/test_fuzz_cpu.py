"""Mutation fuzzing of the host-side parsers under AddressSanitizer + UBSan (tests/fuzz_host.c): the .q container, its
codebooks and the serial decoder behind them, the shard-set container, FASTQ text and -D distortion files. CPU only: the
host C sources are compiled with ROCm's clang and its sanitizer runtime and linked against the already built HIP objects
(no device code runs; entry points that need the GPU must fail with an error code)."""
import glob
import gzip
import os
import subprocess

import pytest

import golden_util as G

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CLANG = "/opt/rocm/lib/llvm/bin/clang"
HOST = ["qvz_coder", "qvz_decode", "qvz_fastq", "qvz_shardset", "qvz_shard", "qvz_encode", "qvz_multi"]
SAN = ["-O1", "-g", "-fsanitize=address,undefined", "-fno-sanitize-recover=undefined", "-fno-omit-frame-pointer"]


@pytest.fixture(scope="module")
def fuzz_bin():
    if not os.path.exists(CLANG):
        pytest.skip("ROCm clang not found")
    subprocess.run(["make", "-s"], cwd=ROOT, check=True)                       # the HIP objects the harness links against
    out = os.path.join(ROOT, "tests", "_build")
    os.makedirs(out, exist_ok=True)
    exe = os.path.join(out, "fuzz_host")
    srcs = [os.path.join(ROOT, "qvz_amd", "csrc", "host", f + ".c") for f in HOST] + [os.path.join(ROOT, "tests", "fuzz_host.c")]
    deps = srcs + glob.glob(os.path.join(ROOT, "include", "*.h")) + glob.glob(os.path.join(ROOT, "qvz_amd", "csrc", "*.hip.o"))
    if os.path.exists(exe) and all(os.path.getmtime(exe) >= os.path.getmtime(d) for d in deps):
        return exe
    inc = ["-I", os.path.join(ROOT, "include"), "-I", os.path.join(ROOT, "qvz_amd", "csrc", "host")]
    objs = []
    for src in srcs:
        obj = os.path.join(out, os.path.basename(src)[:-2] + ".san.o")
        subprocess.run([CLANG] + SAN + ["-ffp-contract=off", "-std=gnu11"] + inc + ["-c", src, "-o", obj], check=True)
        objs.append(obj)
    hip_objs = sorted(glob.glob(os.path.join(ROOT, "qvz_amd", "csrc", "*.hip.o")))
    subprocess.run(["hipcc", "--offload-arch=gfx950", "-fsanitize=address,undefined", "-o", exe] + objs + hip_objs + ["-lm", "-lpthread", "-ldl"],
                   check=True)
    return exe


def test_host_parsers_survive_mutated_files(fuzz_bin, tmp_path):
    """A few hundred mutants of five kinds of input per seed: every one must come back with a return code (no sanitizer
    report, no signal, no run-away loop: a mutated line count ends when the stream is used up)."""
    seeds = []
    for name in ("tiny5_r10", "c2_f08", "const_f05"):
        p = tmp_path / (name + ".q")
        p.write_bytes(G.gz(name + ".q.gz"))
        seeds.append(str(p))
    rows = gzip.decompress(open(os.path.join(G.GOLD, "t2k_c3_f05_M.u.gz"), "rb").read()).split(b"\n")[:200]
    fq = tmp_path / "s.fastq"
    fq.write_bytes(b"".join(b"@read%d\n" % i + b"A" * len(r) + b"\n+\n" + r + b"\n" for i, r in enumerate(rows) if r))
    env = dict(os.environ, ASAN_OPTIONS="detect_leaks=0:abort_on_error=0", UBSAN_OPTIONS="print_stacktrace=1")
    sset = tmp_path / "c2.set"
    subprocess.run([fuzz_bin, "--make-set", seeds[1], str(sset)], check=True, env=env)
    for seed in (1, 2):
        r = subprocess.run([fuzz_bin, str(seed), "300", str(tmp_path)] + seeds +
                           ["--fastq", str(fq), "--dist", os.path.join(G.GOLD, "dist_asym.txt"), "--set", str(sset)],
                           capture_output=True, text=True, env=env, timeout=600)
        assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-6000:]
        assert "fuzz_host: 300 iterations" in r.stdout
        counts = [int(t) for t in r.stdout.replace(";", " ").replace(":", " ").split() if t.isdigit()]
        assert counts[1] > 20 and counts[2] > 10, r.stdout                   # some mutants must get as far as the decoder

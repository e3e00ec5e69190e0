"""Mutation fuzzing of the host-side parsers under AddressSanitizer + UBSan (tests/fuzz_host.c): the .q container, its
codebooks and the serial decoder behind them, the shard-set container, FASTQ text and -D distortion files.

A CPU program by construction: the host C sources of qvz_amd/csrc/host + tests/fuzz_stubs.c (every qvz_hip_* entry point
they call, each failing with QVZ_E_HIP), one plain clang invocation - no hipcc, no offload architecture, no device
object, no HIP runtime. This file, the harness and its build directory are listed in .gpurunignore: they never travel
to a GPU box (the pool refuses sanitizer builds there), and `test_nothing_shipped_to_the_gpu_box_asks_for_a_sanitizer`
below holds every file that DOES travel to that."""
import fnmatch
import glob
import gzip
import os
import subprocess

import pytest

import golden_util as G

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CLANG = "/opt/rocm/lib/llvm/bin/clang"
HOST = ["qvz_coder", "qvz_decode", "qvz_fastq", "qvz_shardset", "qvz_shard", "qvz_encode", "qvz_multi", "qvz_codebook_host", "qvz_egress"]
SAN = ["-O1", "-g", "-fsanitize=address,undefined", "-fno-sanitize-recover=undefined", "-fno-omit-frame-pointer"]
# the harness must not see a GPU even where one exists (it links no HIP runtime; this is belt and braces)
CPU_ONLY = {"HIP_VISIBLE_DEVICES": "-1", "ROCR_VISIBLE_DEVICES": "", "CUDA_VISIBLE_DEVICES": "-1"}


@pytest.fixture(scope="module")
def fuzz_bin():
    if not os.path.exists(CLANG):
        pytest.skip("ROCm clang not found")
    out = os.path.join(ROOT, "tests", "_build")
    os.makedirs(out, exist_ok=True)
    exe = os.path.join(out, "fuzz_host")
    srcs = [os.path.join(ROOT, "qvz_amd", "csrc", "host", f + ".c") for f in HOST] + \
           [os.path.join(ROOT, "tests", "fuzz_host.c"), os.path.join(ROOT, "tests", "fuzz_stubs.c")]
    deps = srcs + glob.glob(os.path.join(ROOT, "include", "*.h")) + glob.glob(os.path.join(ROOT, "qvz_amd", "csrc", "host", "*.h")) + [__file__]
    if os.path.exists(exe) and all(os.path.getmtime(exe) >= os.path.getmtime(d) for d in deps):
        return exe
    inc = ["-I", os.path.join(ROOT, "include"), "-I", os.path.join(ROOT, "qvz_amd", "csrc", "host")]
    subprocess.run([CLANG] + SAN + ["-ffp-contract=off", "-std=gnu11"] + inc + srcs + ["-o", exe, "-lm", "-lpthread", "-ldl"], check=True)
    r = subprocess.run(["ldd", exe], capture_output=True, text=True)
    assert "amdhip" not in r.stdout and "hsa-runtime" not in r.stdout, r.stdout       # no GPU runtime in a sanitizer build
    return exe


def _shipped_files():
    """Every file a `gpurun` snapshot of the repository carries: the work tree minus .git/, gpurun_out/ and the
    patterns of .gpurunignore (a pattern names a path, a directory prefix or a glob, as in .gitignore)."""
    pats = [l.strip() for l in open(os.path.join(ROOT, ".gpurunignore")) if l.strip() and not l.startswith("#")]

    def ignored(rel):
        for p in pats:
            q = p.rstrip("/").rstrip("/*")
            if rel == q or rel.startswith(q + "/") or fnmatch.fnmatch(rel, p) or fnmatch.fnmatch(os.path.basename(rel), p):
                return True
        return False

    for base, dirs, files in os.walk(ROOT):
        rel_base = os.path.relpath(base, ROOT)
        dirs[:] = [d for d in dirs if not (rel_base == "." and d in (".git", "gpurun_out")) and
                   not ignored(os.path.normpath(os.path.join(rel_base, d)))]
        for f in files:
            rel = os.path.normpath(os.path.join(rel_base, f))
            if not ignored(rel):
                yield rel


def test_nothing_shipped_to_the_gpu_box_asks_for_a_sanitizer():
    """The GPU pool refuses a snapshot in which any file asks for a sanitizer or an XNACK-on build (round 2 lost its whole
    driver-side GPU suite to that): such lines may only live in files .gpurunignore keeps here."""
    needles = [b"-fsanitize", b"HSA_XNACK", b"xnack+"]
    assert "tests/test_fuzz_cpu.py" not in set(_shipped_files())
    bad = []
    for rel in _shipped_files():
        path = os.path.join(ROOT, rel)
        if os.path.islink(path) or os.path.getsize(path) > (64 << 20):
            continue
        with open(path, "rb") as fh:
            data = fh.read()
        if b"\0" in data[:4096]:                                               # built objects: flags are not in them as text
            continue
        for n in needles:
            if n in data:
                bad.append((rel, n.decode()))
    assert not bad, "files that travel to the GPU box mention sanitizer / XNACK flags: %r" % bad


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
    env = dict(os.environ, ASAN_OPTIONS="detect_leaks=0:abort_on_error=0", UBSAN_OPTIONS="print_stacktrace=1", **CPU_ONLY)
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

"""The error path of the multi-device C driver (host/qvz_multi.c) on the CPU: host C sources + a device layer in which every
call fails (tests/fuzz_stubs.c) + a fake librccl.so that notices any call on a communicator already freed
(tests/fake_rccl.c). ncclCommAbort frees its communicator, so a part that fails must abort each one at most once and the
driver must not destroy what was aborted (round-2 advisor finding; no GPU test can reach this path on a one-GPU box)."""
import os
import re
import subprocess

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
HOST = ["qvz_coder", "qvz_decode", "qvz_fastq", "qvz_shardset", "qvz_shard", "qvz_encode", "qvz_multi", "qvz_codebook_host", "qvz_egress"]


def test_a_failing_part_frees_every_communicator_exactly_once(tmp_path):
    stubs = os.path.join(ROOT, "tests", "fuzz_stubs.c")
    if not os.path.exists(stubs):
        pytest.skip("tests/fuzz_stubs.c does not travel to the GPU box")
    inc = ["-I", os.path.join(ROOT, "include"), "-I", os.path.join(ROOT, "qvz_amd", "csrc", "host")]
    fake = str(tmp_path / "librccl.so")
    subprocess.run(["gcc", "-O1", "-shared", "-fPIC", "-o", fake, os.path.join(ROOT, "tests", "fake_rccl.c")], check=True)
    exe = str(tmp_path / "multi_abort")
    srcs = [os.path.join(ROOT, "qvz_amd", "csrc", "host", f + ".c") for f in HOST] + [stubs, os.path.join(ROOT, "tests", "multi_abort_harness.c")]
    subprocess.run(["gcc", "-O1", "-g", "-std=gnu11", "-ffp-contract=off"] + inc + srcs + ["-o", exe, "-lm", "-lpthread", "-ldl"], check=True)
    env = dict(os.environ, LD_LIBRARY_PATH=str(tmp_path) + ":" + os.environ.get("LD_LIBRARY_PATH", ""))
    for _ in range(20):                                             # the two parts fail at about the same time: run it a few times
        r = subprocess.run([exe, str(tmp_path / "in.txt"), str(tmp_path / "out.qset")], env=env, capture_output=True, text=True, timeout=60)
        assert r.returncode == 0, r.stdout + r.stderr
        m = re.search(r"rc=(-?\d+) made=(\d+) aborts=(\d+) destroys=(\d+) violations=(\d+) live=(\d+)", r.stdout)
        assert m, r.stdout
        rc, made, aborts, destroys, violations, live = (int(x) for x in m.groups())
        assert rc != 0, "an encode without a device must fail"
        assert made == 2 and violations == 0 and live == 0 and aborts + destroys == 2, r.stdout

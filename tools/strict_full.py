#!/usr/bin/env python3
"""The byte-identical single-stream .q at BASELINE.json's full size, on the GPU box: QGEN-1 seed 3, 100M reads x 150 columns,
`qvz -c 1 -d L -f 0.5 -s in.txt out.q` (this repo's binary, WELL seed fixed) file to file, its md5 compared with what the
UNMODIFIED reference wrote for the same input in the build container (tests/golden/fullsize_cfg2.json, tools/ref_fullsize.py):
the one-to-one answer to the reference's start_qv_compression (src/qv_compressor.c:48-143) at the size the metric is quoted on.
Prints one JSON line (commit it as profiles/rNN_strict_full.txt).

    python tools/strict_full.py [--reads N]     (other sizes: no stored md5 to compare with)
"""
import hashlib
import json
import os
import re
import subprocess
import sys
import tempfile
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def md5_of(path):
    h = hashlib.md5()
    with open(path, "rb") as f:
        for blk in iter(lambda: f.read(1 << 26), b""):
            h.update(blk)
    return h.hexdigest()


def main():
    import torch
    from qvz_amd import host, kernels as K
    reads = int(sys.argv[sys.argv.index("--reads") + 1]) if "--reads" in sys.argv else 100_000_000
    cols, seed = 150, 3
    gold = os.path.join(ROOT, "tests", "golden", "fullsize_cfg2.json")
    ref = json.load(open(gold)) if os.path.exists(gold) and reads == 100_000_000 else None
    base = "/dev/shm" if os.path.isdir("/dev/shm") else tempfile.gettempdir()
    td = tempfile.mkdtemp(prefix="qvz_strict_", dir=base)
    try:
        inp, out = os.path.join(td, "in.txt"), os.path.join(td, "out.q")
        sym = K.qgen1(seed, reads, cols)
        host.egress_rows(sym, reads, inp)
        del sym
        torch.cuda.empty_cache()
        in_md5 = md5_of(inp) if ref else None
        env = dict(os.environ, QVZ_WELL_SEED="debug", QVZ_TIMING="1")
        env.pop("QVZ_SHARDS", None)
        t0 = time.perf_counter()
        res = subprocess.run([os.path.join(ROOT, "qvz_amd", "bin", "qvz"), "-c", "1", "-d", "L", "-f", "0.5", "-s", inp, out], env=env,
                             capture_output=True, text=True, timeout=1000)
        wall = time.perf_counter() - t0
        if res.returncode:
            print(json.dumps(dict(failed=(res.stdout + res.stderr)[-800:])))
            return 1
        sline = [ln for ln in res.stdout.splitlines() if ln.startswith("rate,")][0].strip()
        timing = [ln for ln in res.stderr.splitlines() if ln.startswith("qvz_timing: upload=")][-1]
        ph = {k: float(v) for k, v in re.findall(r"(\w+)=([\d.]+(?:e[-+]?\d+)?)", timing)}
        sym_count = float(reads) * cols
        line = dict(what="byte-identical single-stream .q, file to file, through the qvz binary (host C driver; the adaptive model on the GPU, "
                         "the interval chain on one host core)", workload=f"QGEN-1 seed {seed}, {reads} reads x {cols} cols, -c 1 -d L -f 0.5",
                    wall_seconds=round(wall, 2), symbols_per_second=round(sym_count / wall, 1),
                    encode_phase_seconds=ph.get("encode"), encode_phase_ns_per_symbol=round(ph.get("encode", 0.0) / sym_count * 1e9, 3),
                    phase_seconds={k: ph.get(k) for k in ("upload", "cluster", "stats", "codebook", "encode", "total")},
                    s_line=sline, q_bytes=os.path.getsize(out), q_md5=md5_of(out))
        if ref:
            f = sline.split(",")
            line.update(reference_q_md5=ref["q_md5"], md5_equals_reference=line["q_md5"] == ref["q_md5"], input_md5_equals_reference=in_md5 == ref["input_md5"],
                        s_line_fields_equal_reference=(f[1].strip(), f[3].strip(), int(f[7])) == (ref["rate"], ref["distortion"], ref["size"]),
                        reference_wall_seconds_build_container=ref["reference_wall_seconds"],
                        reference_source="tests/golden/fullsize_cfg2.json (tools/ref_fullsize.py: the unmodified reference, one core of the build container)")
        print(json.dumps(line), flush=True)
        return 0 if not ref or line["md5_equals_reference"] else 2
    finally:
        import shutil
        shutil.rmtree(td, ignore_errors=True)


if __name__ == "__main__":
    sys.exit(main())

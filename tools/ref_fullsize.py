#!/usr/bin/env python3
"""The UNMODIFIED reference (oracle/_ref/qvz_ref_fixedseed, built by oracle/build_ref.sh) run once at BASELINE.json's full size -
configs[2]: QGEN-1 seed 3, 100M reads x 150 columns, -c 1 -d L -f 0.5 - in the build container: ~15 minutes of one core.
Writes tests/golden/fullsize_cfg2.json: md5 and size of its .q, its -s line, its wall time (the reference's single-thread
figure on THIS host, whole encode). The GPU box runs the product's byte-identical single-stream path on the same input
(tools/strict_full.py) and compares the md5: the full-size answer to src/qv_compressor.c:48-143 one to one.

    python tools/ref_fullsize.py [--reads 100000000]
"""
import hashlib
import json
import os
import subprocess
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "tests"))

SEED, COLS = 3, 150
FLAGS = ["-c", "1", "-d", "L", "-f", "0.5"]


def piece(args):
    path, first, n = args
    import oracle_lib as O
    mm = np.memmap(path, dtype=np.uint8, mode="r+")
    step = 1 << 18
    for a in range(first, first + n, step):
        m = min(step, first + n - a)
        mm[a * (COLS + 1):(a + m) * (COLS + 1)] = O.qgen1_rows(SEED, m, COLS, first=a).reshape(-1)
    mm.flush()
    return n


def main():
    reads = int(sys.argv[sys.argv.index("--reads") + 1]) if "--reads" in sys.argv else 100_000_000
    base = "/dev/shm" if os.path.isdir("/dev/shm") else "/tmp"
    inp, out = os.path.join(base, "qvz_full_in.txt"), os.path.join(base, "qvz_full_out.q")
    with open(inp, "wb") as f:
        f.truncate(reads * (COLS + 1))
    workers = max(1, (os.cpu_count() or 2))
    per = -(-reads // workers)
    procs = []
    t0 = time.time()
    for a in range(0, reads, per):      # child processes of this script (no pool: the pieces are few and long)
        procs.append(subprocess.Popen([sys.executable, os.path.abspath(__file__), "--piece", inp, str(a), str(min(per, reads - a))]))
    for p in procs:
        assert p.wait() == 0
    print(f"input of {reads} reads written in {time.time() - t0:.1f} s", flush=True)
    h = hashlib.md5()
    with open(inp, "rb") as f:
        for blk in iter(lambda: f.read(1 << 26), b""):
            h.update(blk)
    t0 = time.time()
    res = subprocess.run([os.path.join(ROOT, "oracle", "_ref", "qvz_ref_fixedseed")] + FLAGS + ["-s", inp, out], check=True, capture_output=True, text=True)
    wall = time.time() - t0
    sline = [ln for ln in res.stdout.splitlines() if ln.startswith("rate,")][0]
    hq = hashlib.md5()
    with open(out, "rb") as f:
        for blk in iter(lambda: f.read(1 << 26), b""):
            hq.update(blk)
    f = sline.split(",")
    doc = dict(generator="tools/ref_fullsize.py", reference_binary="oracle/_ref/qvz_ref_fixedseed (WELL state 32 x 0x55555555)",
               input=dict(kind="qgen1", seed=SEED, n=reads, c=COLS), flags=" ".join(FLAGS), input_md5=h.hexdigest(), q_md5=hq.hexdigest(),
               q_size=os.path.getsize(out), rate=f[1].strip(), distortion=f[3].strip(), size=int(f[7]), s_line=sline.strip(),
               reference_wall_seconds=round(wall, 1), reference_symbols_per_second=round(reads * COLS / wall, 1),
               host=f"build container, {os.cpu_count()} CPUs visible, 1 used")
    name = "fullsize_cfg2.json" if reads == 100_000_000 else f"fullsize_cfg2_{reads}.json"
    json.dump(doc, open(os.path.join(ROOT, "tests", "golden", name), "w"), indent=1, sort_keys=True)
    print(json.dumps(doc), flush=True)
    os.remove(inp)
    os.remove(out)


if __name__ == "__main__":
    if "--piece" in sys.argv:
        i = sys.argv.index("--piece")
        piece((sys.argv[i + 1], int(sys.argv[i + 2]), int(sys.argv[i + 3])))
    else:
        main()

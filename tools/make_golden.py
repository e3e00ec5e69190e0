#!/usr/bin/env python3
"""Generate tests/golden/ fixtures by RUNNING the unmodified reference (oracle/_ref, built in place from
/root/reference by oracle/build_ref.sh). Only works in the build container (the reference does not travel);
the fixtures it writes are data: inputs, expected outputs, md5s and the reference's own `-s` numbers.

    python tools/make_golden.py            # regenerate everything

Layout of tests/golden/:
    manifest.json                    one entry per case (input recipe, flags, md5s, sizes, -s stats)
    <case>.q.gz  <case>.u.gz         reference outputs for the cases with store=True
    <case>.in.gz                     input text, only for inputs that are not QGEN-1
    <case>.phase.npz                 labels / means / counts / PMF counters / marginals from ref_harness
"""
import gzip
import hashlib
import json
import os
import subprocess
import sys
import tempfile

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "tests"))
import oracle_lib as O  # noqa: E402  (QGEN-1 generator only)

REF = os.path.join(ROOT, "oracle", "_ref")
GOLD = os.path.join(ROOT, "tests", "golden")


def sm64(x):
    x = (x + 0x9E3779B97F4A7C15) & (2 ** 64 - 1)
    x = ((x ^ (x >> 30)) * 0xBF58476D1CE4E5B9) & (2 ** 64 - 1)
    x = ((x ^ (x >> 27)) * 0x94D049BB133111EB) & (2 ** 64 - 1)
    return x ^ (x >> 31)


def input_rows(spec):
    kind = spec["kind"]
    n, c = spec["n"], spec["c"]
    if kind == "qgen1":
        return O.qgen1_rows(spec["seed"], n, c)
    rows = np.zeros((n, c + 1), dtype=np.uint8)
    rows[:, c] = 10
    if kind == "const":                      # every symbol the same: zero-entropy contexts
        rows[:, :c] = spec["q"] + 33
    elif kind == "uniform":                  # full 72-symbol alphabet, splitmix64 counter stream
        for i in range(n):
            for j in range(c):
                rows[i, j] = 33 + sm64(spec["seed"] * 1000003 + i * c + j) % 72
    elif kind == "twolevel":                 # two far-apart read populations (k-means separable)
        for i in range(n):
            base = 5 if sm64(spec["seed"] + i) % 3 == 0 else 60
            for j in range(c):
                rows[i, j] = 33 + base + sm64(spec["seed"] * 7919 + i * c + j) % 6
    else:
        raise ValueError(kind)
    return rows


# flags are exactly the reference CLI's (src/main.c:208-303)
CASES = [
    # the three survey-verified tiny cases
    dict(name="t2k_r10_M", inp=dict(kind="qgen1", seed=1, n=2000, c=30), flags="-c 1 -d M -r 1.0", store=True, phase=True),
    dict(name="t2k_f05_L", inp=dict(kind="qgen1", seed=1, n=2000, c=30), flags="-c 1 -d L -f 0.5", store=True),
    dict(name="t2k_c3_f05_M", inp=dict(kind="qgen1", seed=1, n=2000, c=30), flags="-c 3 -d M -f 0.5", store=True, phase=True),
    # BASELINE.json configs[0] and the flag sets of configs[2], configs[3] at its shape
    dict(name="cfg1_r10_M", inp=dict(kind="qgen1", seed=1, n=10000, c=100), flags="-c 1 -d M -r 1.0", store=True),
    dict(name="cfg1_f05_L", inp=dict(kind="qgen1", seed=1, n=10000, c=100), flags="-c 1 -d L -f 0.5", store=False),
    dict(name="cfg1_c3_r10_M", inp=dict(kind="qgen1", seed=1, n=10000, c=100), flags="-c 3 -d M -r 1.0", store=False, phase=True),
    # edge cases
    dict(name="const_f05", inp=dict(kind="const", q=40, n=300, c=12), flags="-c 1 -d M -f 0.5", store=True),
    dict(name="const_r05", inp=dict(kind="const", q=40, n=300, c=12), flags="-c 1 -d M -r 0.5", store=True),
    dict(name="uniform72_r20_A", inp=dict(kind="uniform", seed=5, n=1500, c=16), flags="-c 1 -d A -r 2.0", store=True),
    dict(name="uniform72_f09_M", inp=dict(kind="uniform", seed=6, n=1200, c=10), flags="-c 1 -d M -f 0.9", store=True),
    dict(name="c2_f08", inp=dict(kind="qgen1", seed=9, n=400, c=2), flags="-c 1 -d M -f 0.8", store=True),
    dict(name="tiny5_r10", inp=dict(kind="qgen1", seed=11, n=5, c=7), flags="-c 1 -d L -r 1.0", store=True),
    dict(name="twolevel_c2_f05", inp=dict(kind="twolevel", seed=3, n=900, c=20), flags="-c 2 -d M -f 0.5", store=True, phase=True),
    dict(name="f00_zero_rate", inp=dict(kind="qgen1", seed=2, n=500, c=25), flags="-c 1 -d M -f 0.0", store=True),
    dict(name="r35_high_rate", inp=dict(kind="qgen1", seed=2, n=800, c=25), flags="-c 1 -d M -r 3.5", store=True),
    dict(name="wide_c151", inp=dict(kind="qgen1", seed=4, n=700, c=151), flags="-c 1 -d M -f 0.4", store=False),
    # long enough for update_stats to rescale (a table passes n > 2^19 after 65536 hits, qv_stream.c:15-24)
    dict(name="rescale_140k_c4", inp=dict(kind="qgen1", seed=12, n=140000, c=4), flags="-c 1 -d M -f 0.5", store=False),
    dict(name="rescale_c2_100k_c3", inp=dict(kind="qgen1", seed=13, n=100000, c=3), flags="-c 2 -d M -r 0.8", store=False),
    # -D FILE (src/distortion.c:100-145): an asymmetric matrix, "@name" = tests/golden/name (written by dist_asym_text below)
    dict(name="t2k_D_asym_f06", inp=dict(kind="qgen1", seed=1, n=2000, c=30), flags="-c 1 -D @dist_asym.txt -f 0.6", store=True,
         custom_dist=True),
]


# Rate-distortion sweeps: the reference's generate_rd.sh:1-17 - twenty encodes of one file, `comp` spelled as `bc -l` prints
# idx * step ("0", ".05", ".10" ... / "1.00", "1.10" ...). The script's active line is `-c 1 -f $comp -s`, its commented one
# `-c 3 -r $comp -s` with step 0.10. Every point is one run of the reference binary; what is kept is the md5 of its .q and -u,
# and the fields of its -s line.
def bc_l(idx, step_hundredths):
    v = idx * step_hundredths
    return "0" if v == 0 else (f"{v // 100}" if v >= 100 else "") + f".{v % 100:02d}"


SWEEPS = [
    dict(name="rd_t2k", inp=dict(kind="qgen1", seed=1, n=2000, c=30), fixed="-c 1", flag="-f", comps=[bc_l(i, 5) for i in range(20)]),
    dict(name="rd_c3r_t2k", inp=dict(kind="qgen1", seed=1, n=2000, c=30), fixed="-c 3", flag="-r", comps=[bc_l(i, 10) for i in range(20)]),
]


def run_sweeps(manifest, td):
    manifest["sweeps"] = {}
    for sw in SWEEPS:
        rows = input_rows(sw["inp"])
        text = rows.tobytes()
        inp = os.path.join(td, sw["name"] + ".txt")
        open(inp, "wb").write(text)
        points = []
        for comp in sw["comps"]:
            q, u = os.path.join(td, "sw.q"), os.path.join(td, "sw.u")
            out = subprocess.run([os.path.join(REF, "qvz_ref_fixedseed")] + sw["fixed"].split() + [sw["flag"], comp, "-s", "-u", u, inp, q],
                                 check=True, capture_output=True, text=True).stdout
            sline = [ln for ln in out.splitlines() if ln.startswith("rate,")][0].split(",")
            qb, ub = open(q, "rb").read(), open(u, "rb").read()
            points.append(dict(comp=comp, q_md5=md5(qb), u_md5=md5(ub), q_size=len(qb), rate=sline[1].strip(), distortion=sline[3].strip(),
                               size=int(sline[7])))
            print(f"{sw['name']:12s} {sw['flag']} {comp:5s} rate={sline[1].strip()} dist={sline[3].strip()} size={int(sline[7])}")
        manifest["sweeps"][sw["name"]] = dict(input=sw["inp"], input_md5=md5(text), fixed=sw["fixed"], flag=sw["flag"], points=points,
                                              source="reference generate_rd.sh:1-17, one run of oracle/_ref/qvz_ref_fixedseed per point")


def dist_asym_text():
    """72 rows x 72 columns, row x column y = cost of reconstructing x as y: overshoot costs 1.5|x-y|, undershoot 2|x-y|.
    Short numbers: the reference reads rows through a 1024-byte line buffer (distortion.c:105,117)."""
    lines = ["# asymmetric distortion matrix for the -D fixture: row = true value, column = reconstruction"]
    for x in range(72):
        lines.append(",".join(repr(1.5 * (y - x) if y > x else 2.0 * (x - y)) for y in range(72)))
    return ("\n".join(lines) + "\n").encode()


def md5(b):
    return hashlib.md5(b).hexdigest()


def gz_write(path, data):
    with open(path, "wb") as f:
        with gzip.GzipFile(fileobj=f, mode="wb", mtime=0, compresslevel=9) as g:
            g.write(data)


def parse_phase(path):
    raw = open(path, "rb").read()
    n = int.from_bytes(raw[0:8], "little")
    c = int.from_bytes(raw[8:12], "little")
    k = int.from_bytes(raw[12:16], "little")
    off = 16
    labels = np.frombuffer(raw, np.uint8, n, off); off += n
    means = np.frombuffer(raw, np.uint8, k * c, off).reshape(k, c); off += k * c
    counts = np.frombuffer(raw, np.uint32, k, off); off += 4 * k
    cnt = np.frombuffer(raw, np.uint32, k * c * 72 * 72, off).reshape(k, c, 72, 72); off += 4 * k * c * 72 * 72
    marg = np.frombuffer(raw, np.float64, k * c * 72, off).reshape(k, c, 72)
    return labels, means, counts, cnt, marg


def main():
    subprocess.run(["make", "-s", "-C", os.path.join(ROOT, "oracle"), "all"], check=True)
    os.makedirs(GOLD, exist_ok=True)
    manifest = {"generator": "tools/make_golden.py", "reference_binary": "oracle/_ref/qvz_ref_fixedseed "
                "(reference sources + -DLINUX -DDEBUG -O3: WELL state = 32 x 0x55555555)", "cases": {}}
    only = sys.argv[sys.argv.index("--only") + 1] if "--only" in sys.argv else None      # regenerate one case, keep the rest
    if only:                                                                             # (--only sweeps: the sweeps alone)
        manifest = json.load(open(os.path.join(GOLD, "manifest.json")))
    open(os.path.join(GOLD, "dist_asym.txt"), "wb").write(dist_asym_text())
    with tempfile.TemporaryDirectory() as td:
        for case in CASES:
            if only and case["name"] != only:
                continue
            name, spec = case["name"], case["inp"]
            rows = input_rows(spec)
            text = rows.tobytes()
            inp = os.path.join(td, name + ".txt")
            open(inp, "wb").write(text)
            q, u = os.path.join(td, name + ".q"), os.path.join(td, name + ".u")
            flags = [os.path.join(GOLD, f[1:]) if f.startswith("@") else f for f in case["flags"].split()]
            out = subprocess.run([os.path.join(REF, "qvz_ref_fixedseed")] + flags +
                                 ["-s", "-u", u, inp, q], check=True, capture_output=True, text=True).stdout
            dec = os.path.join(td, name + ".dec")
            subprocess.run([os.path.join(REF, "qvz_ref_fixedseed"), "-x", q, dec], check=True, capture_output=True)
            qb, ub, db = open(q, "rb").read(), open(u, "rb").read(), open(dec, "rb").read()
            assert db == ub, f"{name}: reference decode != -u"
            sline = [ln for ln in out.splitlines() if ln.startswith("rate,")][0].split(",")
            entry = dict(input=spec, flags=case["flags"], input_md5=md5(text), q_md5=md5(qb), u_md5=md5(ub),
                         q_size=len(qb), rate=float(sline[1]), distortion=float(sline[3]), size=int(sline[7]),
                         stored=bool(case.get("store")), phase=bool(case.get("phase")))
            if case.get("custom_dist"):
                entry["custom_dist"] = True
            if spec["kind"] != "qgen1":
                gz_write(os.path.join(GOLD, name + ".in.gz"), text)
            if case.get("store"):
                gz_write(os.path.join(GOLD, name + ".q.gz"), qb)
                gz_write(os.path.join(GOLD, name + ".u.gz"), ub)
            if case.get("phase"):
                fl = case["flags"].split()
                k = fl[fl.index("-c") + 1]
                mode, ratio = ("f", fl[fl.index("-f") + 1]) if "-f" in fl else ("r", fl[fl.index("-r") + 1])
                dist = fl[fl.index("-d") + 1]
                pb = os.path.join(td, name + ".bin")
                subprocess.run([os.path.join(REF, "ref_harness"), inp, k, mode, ratio, dist, pb], check=True,
                               capture_output=True)
                labels, means, counts, cnt, marg = parse_phase(pb)
                np.savez_compressed(os.path.join(GOLD, name + ".phase.npz"), labels=labels, means=means,
                                    counts=counts, cnt=cnt, marg=marg)
            manifest["cases"][name] = entry
            print(f"{name:18s} q={len(qb):8d} rate={entry['rate']:.4f} dist={entry['distortion']:.4f} "
                  f"size={entry['size']}")
        if not only or only == "sweeps":
            run_sweeps(manifest, td)
    # WELL known-answer vector from the reference's own well.c (state 32 x 0x55555555)
    manifest["well_kat"] = dict(state="32 x 0x55555555", first_words=["0x82fd4280", "0x815ebfd5", "0x5303e800", "0x2e7c4815"],
                                first_draws=[0, 5, 117, 23, 85, 127, 122, 10, 0, 80, 15, 24],
                                source="SURVEY.md section 8c (checked against compiled reference well.c)")
    json.dump(manifest, open(os.path.join(GOLD, "manifest.json"), "w"), indent=1, sort_keys=True)


if __name__ == "__main__":
    main()

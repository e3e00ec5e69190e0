"""Access to tests/golden/ fixtures (produced by tools/make_golden.py from the compiled reference)."""
import gzip
import hashlib
import json
import os

import numpy as np

import oracle_lib as O

GOLD = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")
MANIFEST = json.load(open(os.path.join(GOLD, "manifest.json")))
CUSTOM_DIST_CASES = {n: e for n, e in MANIFEST["cases"].items() if e.get("custom_dist")}   # -D FILE: CLI-level tests only
CASES = {n: e for n, e in MANIFEST["cases"].items() if not e.get("custom_dist")}


def well_far():
    """WELL-1024a states far into the stream (tests/golden/well_far.json: the reference's own well_1024a stepped from
    32 x 0x55555555 by oracle/well_far.c). Returns [(words, state in the n = 0 form a file stores, next four words)]."""
    doc = json.load(open(os.path.join(GOLD, "well_far.json")))
    out = []
    for v in doc["vectors"]:
        st = np.array(v["state"], dtype=np.uint32)
        rot = np.array([st[(v["n"] + i) & 31] for i in range(32)], dtype=np.uint32)
        out.append((int(v["words"]), rot, [int(x) for x in v["next"]]))
    return out


def cli_flags(entry):
    """The reference CLI flags of a fixture as an argv list ("@name" = a file in tests/golden/)."""
    return [os.path.join(GOLD, f[1:]) if f.startswith("@") else f for f in entry["flags"].split()]


def md5(b):
    return hashlib.md5(b).hexdigest()


def gz(name):
    return gzip.open(os.path.join(GOLD, name), "rb").read()


def case_rows(name):
    """(n, c+1) uint8 text-layout rows of a fixture's input (QGEN-1 inputs are regenerated)."""
    e = CASES[name]
    spec = e["input"]
    if spec["kind"] == "qgen1":
        rows = O.qgen1_rows(spec["seed"], spec["n"], spec["c"])
    else:
        rows = O.rows_from_text(gz(name + ".in.gz"), spec["c"])
    assert md5(rows.tobytes()) == e["input_md5"], f"{name}: input md5 mismatch"
    return rows


def case_opts(name):
    """Reference CLI flags of a fixture -> dict(k, mode, ratio, dist)."""
    fl = CASES[name]["flags"].split()
    k = int(fl[fl.index("-c") + 1])
    if "-f" in fl:
        mode, ratio = O.MODE_RATIO, float(fl[fl.index("-f") + 1])
    else:
        mode, ratio = O.MODE_FIXED, float(fl[fl.index("-r") + 1])
    return dict(k=k, mode=mode, ratio=ratio, dist=fl[fl.index("-d") + 1])


def phase(name):
    return np.load(os.path.join(GOLD, name + ".phase.npz"))


SWEEPS = MANIFEST.get("sweeps", {})


def sweep_rows(name):
    """input rows of a rate-distortion sweep fixture (reference generate_rd.sh:1-17, one reference run per point)"""
    e = SWEEPS[name]
    spec = e["input"]
    rows = O.qgen1_rows(spec["seed"], spec["n"], spec["c"])
    assert md5(rows.tobytes()) == e["input_md5"], f"{name}: input md5 mismatch"
    return rows


def sweep_opts(name):
    """-> (k, oracle mode, [ratio as the reference's atof reads the script's spelling])"""
    e = SWEEPS[name]
    k = int(e["fixed"].split()[1])
    mode = O.MODE_RATIO if e["flag"] == "-f" else O.MODE_FIXED
    return k, mode, [float(p["comp"]) for p in e["points"]]

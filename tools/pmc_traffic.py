#!/usr/bin/env python3
"""Reduces two rocprofv3 PMC passes (FETCH_SIZE, WRITE_SIZE) of one bench.py step to HBM bytes per kernel and step.

    python tools/pmc_traffic.py <dir of the FETCH_SIZE pass> <dir of the WRITE_SIZE pass> > profiles/rNN_pmc_traffic.json

Both counters are in KiB. FETCH_SIZE is doubled (gfx950 counts a 128-byte request as 64 bytes; every kernel here reads
with wide coalesced loads - /opt/skills/guides/MI355X_MICROARCH.md, HBM section); WRITE_SIZE is taken as is.
"""
import csv, glob, json, os, sys

NAMES = {"k_model_ranks": "model_records", "k_model_columns": "model_records", "k_code_records": "code_records",
         "k_quantize": "quantize", "k_pmf_walk": "pmf_hist", "k_kmeans_accum": "kmeans_accumulate",
         "k_draws_transpose": "draws_transpose", "k_well_gen": "well_gen", "k_kmeans_assign": "kmeans_assign",
         "k_lm_window": "lm_window", "k_kmeans_step": "kmeans_step"}


# FETCH_SIZE x 2 is calibrated for 128-byte requests (wide coalesced reads). k_code_records' loader now asks for 64-byte
# halves of a line (16 bytes per lane, four instructions per half; the other half a read later): those are tallied at
# their size - the raw figure (225 GB) sits just under the algorithmic 243 GB (the rest are L2 hits), doubled it would
# claim twice the bytes the kernel can possibly ask for.
FETCH_FACTOR = {"code_records": 1.0}


def collect(d, counter):
    tot, launches = {}, {}
    for path in glob.glob(os.path.join(d, "**", "*counter_collection.csv"), recursive=True):
        with open(path, newline="") as f:
            for row in csv.DictReader(f):
                if row["Counter_Name"] != counter:
                    continue
                k = row["Kernel_Name"]
                k = k[5:] if k.startswith("void ") else k
                tot[k] = tot.get(k, 0.0) + float(row["Counter_Value"])
                launches.setdefault(k, set()).add(row["Dispatch_Id"])
    return tot, {k: len(v) for k, v in launches.items()}


METHOD = ("rocprofv3 --kernel-trace --pmc FETCH_SIZE (one pass) and --pmc WRITE_SIZE (second pass) around `python3 bench.py --steps 1 "
          "--warmup 0 --no-cpu-baseline --no-extras [--config N]` (tools/collect_profiles.sh, reduced by tools/pmc_traffic.py); counters are "
          "in KiB, summed over the launches of one step; FETCH_SIZE doubled (gfx950 counts 128-byte requests as 64, "
          "MI355X_MICROARCH.md HBM section) except for k_code_records, whose loader asks for 64-byte half lines; WRITE_SIZE as is")


def main():
    fetch, nl = collect(sys.argv[1], "FETCH_SIZE")
    write, _ = collect(sys.argv[2], "WRITE_SIZE")
    workload = json.loads(sys.argv[3]) if len(sys.argv) > 3 else None
    out = {}
    for kern in sorted(fetch, key=lambda k: -(2 * fetch[k] + write.get(k, 0.0))):
        short = next((v for p, v in NAMES.items() if kern.startswith(p)), None)
        if short is None:
            continue
        fb, wb = int(fetch[kern] * FETCH_FACTOR.get(short, 2.0) * 1024), int(write.get(kern, 0.0) * 1024)
        e = out.setdefault(short, {"kernel": kern.split("(")[0], "launches_per_step": 0, "fetch_bytes": 0, "write_bytes": 0})
        e["launches_per_step"] += nl[kern]
        e["fetch_bytes"] += fb
        e["write_bytes"] += wb
    for e in out.values():
        e["hbm_bytes"] = e["fetch_bytes"] + e["write_bytes"]
    json.dump({"method": METHOD, "workload": workload, "kernels": out} if workload else out, sys.stdout, indent=1)
    print()


if __name__ == "__main__":
    main()

#!/bin/bash
# Runs on the GPU box: SQ counters (issue / wait / LDS) of every kernel of one bench step, two passes of eight counters.
# usage: bash tools/sq_counters.sh <tag> <config>      -> gpurun_out/<tag>/sq_counters_cfg<config>.txt
tag=${1:-sq}; cfg=${2:-2}; root=$(pwd); out=$root/gpurun_out/$tag; mkdir -p $out; export TMPDIR=/tmp; cd /tmp
P1="SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS"
P2="SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAIT_INST_LDS"
i=0
for P in "$P1" "$P2"; do
  i=$((i+1))
  timeout -k 10 400 rocprofv3 --kernel-trace --pmc $P --output-format csv -d $out/sq$i -- python3 $root/bench.py --config $cfg --steps 1 --warmup 0 --no-cpu-baseline --no-extras > $out/sq$i.log 2>&1 || { tail -5 $out/sq$i.log; exit 1; }
done
cd $root
python3 - $out $cfg > $out/sq_counters_cfg$cfg.txt <<'PY'
import csv, glob, sys, collections
out, cfg = sys.argv[1], sys.argv[2]
tot = collections.defaultdict(lambda: collections.defaultdict(float))
calls = collections.Counter()
for i in (1, 2):
    for f in glob.glob(f"{out}/sq{i}/**/*counter_collection.csv", recursive=True):
        seen = set()
        for r in csv.DictReader(open(f)):
            name = r["Kernel_Name"].split("(")[0].replace("void ", "")[:44]
            tot[name][r["Counter_Name"]] += float(r["Counter_Value"])
            if i == 1 and r["Counter_Name"] == "SQ_WAVES": calls[name] += 1
print(f"# SQ counters summed over the dispatches of one bench step (config {cfg}); cycles are quad-cycles per the guide; per kernel")
keys = ["SQ_WAVES", "SQ_WAVE_CYCLES", "SQ_BUSY_CYCLES", "SQ_WAIT_ANY", "SQ_WAIT_INST_ANY", "SQ_ACTIVE_INST_ANY", "SQ_ACTIVE_INST_VALU", "SQ_ACTIVE_INST_LDS",
        "SQ_INSTS_VALU", "SQ_INSTS_SALU", "SQ_INSTS_LDS", "SQ_INSTS_VMEM_RD", "SQ_INSTS_VMEM_WR", "SQ_LDS_BANK_CONFLICT", "SQ_LDS_IDX_ACTIVE", "SQ_WAIT_INST_LDS"]
for name, c in sorted(tot.items(), key=lambda kv: -kv[1].get("SQ_WAVE_CYCLES", 0))[:12]:
    print(f"{name}  dispatches {calls[name]}")
    wc = c.get("SQ_WAVE_CYCLES", 0) or 1
    for k in keys:
        v = c.get(k, 0)
        extra = f"  ({v / wc:.3f} of wave cycles)" if k.startswith(("SQ_WAIT", "SQ_ACTIVE")) else ""
        print(f"    {k:24s} {v:18.0f}{extra}")
    if c.get("SQ_LDS_IDX_ACTIVE"): print(f"    LDS bank-conflict share of LDS cycles: {c['SQ_LDS_BANK_CONFLICT'] / c['SQ_LDS_IDX_ACTIVE']:.3f}")
    if c.get("SQ_WAVES"): print(f"    VALU instructions per wave: {c['SQ_INSTS_VALU'] / c['SQ_WAVES']:.0f}, LDS {c['SQ_INSTS_LDS'] / c['SQ_WAVES']:.0f}, SALU {c['SQ_INSTS_SALU'] / c['SQ_WAVES']:.0f}")
PY
rm -rf $out/sq1 $out/sq2
head -80 $out/sq_counters_cfg$cfg.txt

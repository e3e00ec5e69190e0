#!/bin/bash
# Runs on the GPU box: bench line, kernel-trace stats, and the two PMC passes of one BASELINE configuration.
# usage: bash tools/collect_profiles.sh <tag> <config 1|2|3>     (outputs under gpurun_out/<tag>/)
set -e
tag=${1:-prof}
cfg=${2:-2}
root=$(pwd)
out=$root/gpurun_out/$tag
mkdir -p $out
export TMPDIR=/tmp
timeout -k 10 400 python3 bench.py --config $cfg --steps 5 --warmup 2 > $out/bench.jsonl 2> $out/bench.err
# two more runs without the CPU baseline and the extras: the step time varies by a few per cent from run to run
timeout -k 10 300 python3 bench.py --config $cfg --steps 5 --warmup 2 --no-cpu-baseline --no-extras >> $out/bench.jsonl 2>> $out/bench.err
timeout -k 10 300 python3 bench.py --config $cfg --steps 5 --warmup 2 --no-cpu-baseline --no-extras >> $out/bench.jsonl 2>> $out/bench.err
echo "bench done"
cd /tmp
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $out/stats -- python3 $root/bench.py --config $cfg --steps 3 --warmup 1 --no-cpu-baseline --no-extras > $out/stats.log 2>&1
echo "stats done"
timeout -k 10 300 rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d $out/pmc_fetch -- python3 $root/bench.py --config $cfg --steps 1 --warmup 0 --no-cpu-baseline --no-extras > $out/pmc_fetch.log 2>&1
echo "fetch done"
timeout -k 10 300 rocprofv3 --kernel-trace --pmc WRITE_SIZE --output-format csv -d $out/pmc_write -- python3 $root/bench.py --config $cfg --steps 1 --warmup 0 --no-cpu-baseline --no-extras > $out/pmc_write.log 2>&1
echo "write done"
cd $root
python3 - "$cfg" > $out/workload.json <<'PY'
import json, sys
sys.path.insert(0, ".")
import bench
c = bench.CONFIGS[int(sys.argv[1])]
print(json.dumps({"config": int(sys.argv[1]), "reads": c["reads"], "columns": c["cols"], "flags": c["name"].split(", ")[-1], "coder": "gpu",
                  "shards": 8192, "record_workspace_GiB": 96, "qv_plane_written": True}))
PY
python3 tools/pmc_traffic.py $out/pmc_fetch $out/pmc_write "$(cat $out/workload.json)" > $out/pmc_traffic.json
find $out -name "*kernel_stats.csv" -exec cp {} $out/kernel_stats.csv \;
# the raw traces are large; keep the summaries
find $out -name "*kernel_trace.csv" -delete
find $out -name "*counter_collection.csv" -delete
rm -rf $out/stats $out/pmc_fetch $out/pmc_write
ls -la $out

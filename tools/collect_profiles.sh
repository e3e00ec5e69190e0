#!/bin/bash
# Runs on the GPU box: bench line, kernel-trace stats, and the two PMC passes of the default workload.
# usage: bash tools/collect_profiles.sh <tag>     (outputs under gpurun_out/<tag>/)
set -e
tag=${1:-prof}
root=$(pwd)
out=$root/gpurun_out/$tag
mkdir -p $out
export TMPDIR=/tmp
timeout -k 10 400 python3 bench.py --steps 3 --warmup 1 > $out/bench.jsonl 2> $out/bench.err
# two more runs without the CPU baseline: the step time varies by a few per cent from run to run and box to box
timeout -k 10 300 python3 bench.py --steps 3 --warmup 1 --no-cpu-baseline >> $out/bench.jsonl 2>> $out/bench.err
timeout -k 10 300 python3 bench.py --steps 3 --warmup 1 --no-cpu-baseline >> $out/bench.jsonl 2>> $out/bench.err
echo "bench done"
cd /tmp
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $out/stats -- python3 $root/bench.py --steps 3 --warmup 1 --no-cpu-baseline > $out/stats.log 2>&1
echo "stats done"
timeout -k 10 300 rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d $out/pmc_fetch -- python3 $root/bench.py --steps 1 --warmup 0 --no-cpu-baseline > $out/pmc_fetch.log 2>&1
echo "fetch done"
timeout -k 10 300 rocprofv3 --kernel-trace --pmc WRITE_SIZE --output-format csv -d $out/pmc_write -- python3 $root/bench.py --steps 1 --warmup 0 --no-cpu-baseline > $out/pmc_write.log 2>&1
echo "write done"
cd $root
python3 tools/pmc_traffic.py $out/pmc_fetch $out/pmc_write > $out/pmc_traffic.json
find $out -name "*kernel_stats.csv" -exec cp {} $out/kernel_stats.csv \;
# the raw traces are large; keep the summaries
find $out -name "*kernel_trace.csv" -delete
find $out -name "*counter_collection.csv" -size +20M -delete
ls -la $out

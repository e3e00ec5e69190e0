#!/bin/bash
# kernel-trace stats of one bench configuration: tools/gpu_prof_stats.sh <tag> <bench args...>
tag=$1; shift
root=$(pwd)
out=$root/gpurun_out/$tag
mkdir -p $out
export TMPDIR=/tmp
cd /tmp
timeout -k 10 400 rocprofv3 --kernel-trace --stats --output-format csv -d $out/stats -- python3 $root/bench.py --no-cpu-baseline --no-extras "$@" > $out/stats.log 2>&1
echo "rc=$?"
cd $root
find $out -name "*kernel_stats.csv" -exec cp {} $out/kernel_stats.csv \;
find $out -name "*kernel_trace.csv" -delete
head -40 $out/kernel_stats.csv
tail -3 $out/stats.log

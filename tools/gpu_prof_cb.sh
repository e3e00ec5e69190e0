#!/bin/bash
# kernel-trace stats of one bench configuration, codebook kernels only: bash tools/gpu_prof_cb.sh <tag> <config>
out=$(pwd)/gpurun_out/${1:-profcb}
cfg=${2:-2}
root=$(pwd)
mkdir -p $out
cd /tmp && export TMPDIR=/tmp
timeout -k 10 400 rocprofv3 --kernel-trace --stats --output-format csv -d $out/stats -- python3 $root/bench.py --config $cfg --steps 3 --warmup 1 --no-cpu-baseline --no-extras > $out/stats.log 2>&1
find $out -name "*kernel_stats.csv" -exec cp {} $out/kernel_stats.csv \;
find $out -name "*kernel_trace.csv" -delete
rm -rf $out/stats
grep -i "k_ctx\|k_lm\|k_cb\|k_cond\|k_marg\|Name" $out/kernel_stats.csv | cut -c1-60,150-400
tail -2 $out/stats.log | cut -c1-300

#!/bin/bash
# tools/cb_timeline.sh <tag> <config>: kernel trace of one bench run, reduced to the codebook generation's kernels
# (start, end, name) -> gpurun_out/<tag>/cb_trace.csv; development aid
tag=${1:?tag}; cfg=${2:-2}
root=$(pwd); out=$root/gpurun_out/$tag; mkdir -p $out
export TMPDIR=/tmp
cd /tmp
timeout -k 10 400 rocprofv3 --kernel-trace --output-format csv -d /tmp/cbtr -- python3 $root/bench.py --config $cfg --steps 2 --warmup 1 --no-cpu-baseline --no-extras > $out/trace.log 2>&1
f=$(find /tmp/cbtr -name "*kernel_trace.csv" | head -1)
python3 - "$f" > $out/cb_trace.csv <<'PY'
import csv, sys
rows = list(csv.DictReader(open(sys.argv[1])))
keep = ('k_lm_window', 'k_ctx_', 'k_cb_', 'k_marginals', 'k_cond_pmf', 'k_well_gen', 'k_draws_transpose', 'k_pmf_walk', 'k_quantize_lds')
rows = [r for r in rows if any(k in r['Kernel_Name'] for k in keep)]
rows.sort(key=lambda r: int(r['Start_Timestamp']))
t0 = int(rows[0]['Start_Timestamp']) if rows else 0
for r in rows:
    print(int(r['Start_Timestamp']) - t0, int(r['End_Timestamp']) - t0, r.get('Queue_Id', ''), r['Kernel_Name'][:28].replace(',', ' '), sep=',')
PY
rm -rf /tmp/cbtr
wc -l $out/cb_trace.csv

#!/bin/bash
# stream-count sweep of the default workload (profiles/r02_stream_sweep_cfg2.jsonl) + coder debug of config 3
out=gpurun_out/$1
mkdir -p $out
export TMPDIR=/tmp
: > $out/sweep.jsonl
for s in 1024 2048 4096 8192 16384; do
  timeout -k 10 300 python3 bench.py --shards $s --steps 2 --warmup 1 --no-cpu-baseline --no-extras >> $out/sweep.jsonl 2>> $out/sweep.err || exit 1
done
python3 - $out/sweep.jsonl <<'PY'
import json,sys
for l in open(sys.argv[1]):
    d=json.loads(l); print(d["config"]["coder"].split()[0], d["ms_per_step"], d["bits_per_symbol"], d["phase_seconds"]["coder"])
PY
QVZ_CODER_DEBUG=1 timeout -k 10 300 python3 bench.py --config 3 --steps 1 --warmup 0 --no-cpu-baseline --no-extras > /dev/null 2> $out/coder_debug_cfg3.txt
grep "qvz_hip_encode_planes" $out/coder_debug_cfg3.txt | head -3

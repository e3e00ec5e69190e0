#!/bin/bash
# Is the erratic 1.5 - 3.5 s stall of a fresh process (inside whichever large hipMalloc comes first) the driver scrubbing the
# HBM the PREVIOUS process freed? Same encode back to back, then with a pause before each run.
out=gpurun_out/${1:-e2e2}
mkdir -p $out
d=/dev/shm/qvz_e2e_$$
mkdir -p $d
python3 - $d <<'PY'
import sys
from qvz_amd import host, kernels as K
sym = K.qgen1(3, 100_000_000, 150)
host.egress_rows(sym, 100_000_000, sys.argv[1] + "/in.txt")
PY
export QVZ_WELL_SEED=debug QVZ_SHARDS=8192 QVZ_TIMING=1 QVZ_CODER_DEBUG=1
for pause in 0 0 0 0 6 6 6 6; do
  sleep $pause
  echo "pause $pause"
  qvz_amd/bin/qvz -s -c 1 -d L -f 0.5 $d/in.txt $d/out.qset 2>&1 | grep -E "qvz_timing|host side" | cut -c1-200
done > $out/runs_pause.txt
cat $out/runs_pause.txt
rm -rf $d

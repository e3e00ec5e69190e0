#!/bin/bash
# the C driver's encode tests, then the file-to-file encode at 100M x 150 on a settled GPU (6 s after the previous process)
out=gpurun_out/${1:-e2e3}
mkdir -p $out
export TMPDIR=/tmp
d=/dev/shm/qvz_e2e_$$
mkdir -p $d
python3 - $d <<'PY'
import sys
from qvz_amd import host, kernels as K
sym = K.qgen1(3, 100_000_000, 150)
host.egress_rows(sym, 100_000_000, sys.argv[1] + "/in.txt")
PY
export QVZ_WELL_SEED=debug QVZ_SHARDS=8192 QVZ_TIMING=1 QVZ_CODER_DEBUG=1
for how in thread thread thread; do
  sleep 6
  echo "== QVZ_WORK_EARLY=$how"
  QVZ_WORK_EARLY=$how qvz_amd/bin/qvz -s -c 1 -d L -f 0.5 $d/in.txt $d/out.qset 2>&1 | grep -E "qvz_timing|host side|wall" | cut -c1-260
done > $out/runs.txt
cat $out/runs.txt
rm -rf $d

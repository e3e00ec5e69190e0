#!/bin/bash
# file-to-file encode through the qvz binary at 100M x 150: host I/O rates of the box, then the encode several times
# (upload variance), with the -u text after / beside the coder kernels, and copy-thread counts
out=gpurun_out/${1:-e2e}
mkdir -p $out
export TMPDIR=/tmp
gcc -O2 -o /tmp/io_bench tools/io_bench.c -lpthread && /tmp/io_bench /dev/shm/iob 16 1 4 8 16 > $out/io_bench.txt 2>&1
cat $out/io_bench.txt
d=/dev/shm/qvz_e2e_$$
mkdir -p $d
python3 - $d <<'PY'
import sys, time
from qvz_amd import host, kernels as K
sym = K.qgen1(3, 100_000_000, 150)
t = time.perf_counter(); host.egress_rows(sym, 100_000_000, sys.argv[1] + "/in.txt"); print("input written in", round(time.perf_counter() - t, 2), "s")
PY
export QVZ_WELL_SEED=debug QVZ_SHARDS=8192 QVZ_TIMING=1
for i in 1 2 3; do
  qvz_amd/bin/qvz -s -u $d/out.u -c 1 -d L -f 0.5 $d/in.txt $d/out.qset 2>&1 | grep qvz_timing | cut -d' ' -f1-10
done > $out/encode_runs.txt
echo "--- overlap" >> $out/encode_runs.txt
QVZ_EGRESS_OVERLAP=1 QVZ_CODER_DEBUG=1 qvz_amd/bin/qvz -s -u $d/out.u -c 1 -d L -f 0.5 $d/in.txt $d/out.qset > $out/overlap_debug.txt 2>&1
grep qvz_timing $out/overlap_debug.txt | cut -d' ' -f1-10 >> $out/encode_runs.txt
for t in 8 16; do
  echo "--- copy threads $t" >> $out/encode_runs.txt
  QVZ_COPY_THREADS=$t qvz_amd/bin/qvz -s -u $d/out.u -c 1 -d L -f 0.5 $d/in.txt $d/out.qset 2>&1 | grep qvz_timing | cut -d' ' -f1-10 >> $out/encode_runs.txt
done
echo "--- c3" >> $out/encode_runs.txt
for i in 1 2; do
qvz_amd/bin/qvz -s -u $d/out.u -c 3 -d M -r 1.0 $d/in.txt $d/out.qset 2>&1 | grep qvz_timing | cut -d' ' -f1-10 >> $out/encode_runs.txt
done
echo "--- decode" >> $out/encode_runs.txt
( time qvz_amd/bin/qvz -x $d/out.qset $d/out.dec ) 2>&1 | grep real >> $out/encode_runs.txt
cat $out/encode_runs.txt
rm -rf $d

#!/bin/bash
# host I/O rates of the GPU box (tools/io_bench.c): tmpfs file, 16 GiB
out=gpurun_out/${1:-io}
mkdir -p $out
gcc -O2 -o /tmp/io_bench tools/io_bench.c -lpthread && /tmp/io_bench /dev/shm/iob 16 4 8 16 > $out/io_bench.txt 2>&1
cat $out/io_bench.txt

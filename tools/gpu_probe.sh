#!/bin/bash
# what the GPU box's host offers (cores, memory, tmpfs) - the file-to-file measurements are sized by it
out=gpurun_out/${1:-probe}
mkdir -p $out
{ nproc; free -g; df -h /dev/shm /tmp .; grep -m1 "model name" /proc/cpuinfo; ulimit -a | grep -i "open files\|locked"; } > $out/host.txt 2>&1
cat $out/host.txt
export TMPDIR=/tmp
timeout -k 10 900 python3 -m pytest tests -x -q -m gpu --durations=8 > $out/pytest.log 2>&1
echo "pytest rc=$?"
tail -15 $out/pytest.log

#!/bin/bash
# one pytest selection on the GPU box: bash tools/gpu_one.sh <tag> <pytest args...>
out=gpurun_out/$1
shift
mkdir -p $out
export TMPDIR=/tmp
timeout -k 10 1000 python3 -m pytest "$@" -x -q -s --durations=5 > $out/pytest.log 2>&1
echo "pytest rc=$?"
tail -25 $out/pytest.log

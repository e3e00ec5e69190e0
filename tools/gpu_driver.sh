#!/bin/bash
# What the driver runs at round end, in its order: the -m gpu suite with -x, smoke(), the default bench line.
# usage: gpurun -- bash tools/gpu_driver.sh <tag>
out=gpurun_out/${1:-driver}
mkdir -p $out
export TMPDIR=/tmp
timeout -k 10 1100 python3 -m pytest tests -x -q -m gpu --durations=15 > $out/pytest.log 2>&1
rc=$?
echo "pytest rc=$rc" | tee -a $out/pytest.log
grep -E "passed|failed|FAILED|ERROR" $out/pytest.log | tail -15
[ $rc -eq 0 ] || exit $rc
timeout -k 10 300 python3 -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" > $out/smoke.log 2>&1 || { tail -20 $out/smoke.log; exit 1; }
tail -2 $out/smoke.log
timeout -k 10 500 python3 bench.py > $out/bench.jsonl 2> $out/bench.err
echo "bench rc=$?"
tail -c 600 $out/bench.err

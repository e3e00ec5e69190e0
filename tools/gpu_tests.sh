#!/bin/bash
# the -m gpu suite (no -x: every failure is reported), then the default bench line
out=gpurun_out/$1
mkdir -p $out
export TMPDIR=/tmp
timeout -k 10 1000 python3 -m pytest tests -m gpu -q --durations=12 > $out/pytest.log 2>&1
echo "pytest rc=$?" | tee -a $out/pytest.log
grep -E "passed|failed|FAILED|ERROR" $out/pytest.log | tail -15
timeout -k 10 400 python3 bench.py --steps 5 --warmup 2 > $out/bench_cfg2.jsonl 2> $out/bench_cfg2.err
echo "bench rc=$?"
tail -c 400 $out/bench_cfg2.err

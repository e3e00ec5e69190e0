#!/bin/bash
# First GPU pass of a round: the -m gpu suite, then one bench line per BASELINE configuration.
out=gpurun_out/$1
mkdir -p $out
export TMPDIR=/tmp
timeout -k 10 900 python3 -m pytest tests -m gpu -x -q --durations=15 > $out/pytest.log 2>&1
echo "pytest rc=$?" | tee -a $out/pytest.log
tail -5 $out/pytest.log
timeout -k 10 300 python3 bench.py --steps 5 --warmup 2 > $out/bench_cfg2.jsonl 2> $out/bench_cfg2.err && \
timeout -k 10 300 python3 bench.py --config 3 --steps 3 --warmup 1 > $out/bench_cfg3.jsonl 2> $out/bench_cfg3.err && \
timeout -k 10 300 python3 bench.py --config 1 --steps 5 --warmup 2 > $out/bench_cfg1.jsonl 2> $out/bench_cfg1.err
echo "bench rc=$?"
tail -c 600 $out/bench_cfg2.err $out/bench_cfg3.err $out/bench_cfg1.err
head -c 1500 $out/bench_cfg2.jsonl

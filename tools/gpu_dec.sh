#!/bin/bash
# GPU decoder: parity tests, then the full-size round trip (bench.py --roundtrip prints decode seconds)
out=gpurun_out/${1:-dec}
mkdir -p $out
export TMPDIR=/tmp
timeout -k 10 600 python3 -m pytest tests/test_gpu_decoder.py -x -q > $out/pytest.log 2>&1
echo "pytest rc=$?"; tail -3 $out/pytest.log
for cfg in 2 3; do
timeout -k 10 400 python3 bench.py --config $cfg --steps 1 --warmup 1 --no-cpu-baseline --no-extras --roundtrip 2> $out/err$cfg.txt | python3 -c "
import json,sys
d=json.loads(sys.stdin.readline()); print('config $cfg', d['roundtrip'])"
done

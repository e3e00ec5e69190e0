#!/bin/bash
# full-size file-to-file tests through the qvz binary, then the bench line with e2e_cli
out=gpurun_out/${1:-cli}
mkdir -p $out
export TMPDIR=/tmp
timeout -k 10 900 python3 -m pytest tests/test_gpu_cli_fullsize.py -x -q -s > $out/pytest_cli.log 2>&1
echo "pytest rc=$?"
grep -E "encode file|decode file|input text|passed|failed|Error|error" $out/pytest_cli.log | tail -20
timeout -k 10 600 python3 bench.py > $out/bench.jsonl 2> $out/bench.err
echo "bench rc=$?"
tail -c 300 $out/bench.err
python3 -c "
import json
d=json.loads(open('$out/bench.jsonl').readline())
print(json.dumps(d.get('e2e_cli'), indent=1))
print(d['ms_per_step'], d['step_phase_ms'])
"

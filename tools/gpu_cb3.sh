#!/bin/bash
# codebook generation with the device's exact entropies: parity tests, then the phase split of the three configurations both ways
out=gpurun_out/${1:-cb3}
mkdir -p $out
export TMPDIR=/tmp
timeout -k 10 900 python3 -m pytest tests/test_gpu_codebook.py tests/test_gpu_encode.py tests/test_gpu_dropin.py -x -q > $out/pytest.log 2>&1
echo "pytest rc=$?"; tail -4 $out/pytest.log
for cfg in 2 1 3; do
  for how in default device-flow host-libm; do
    echo "== config $cfg, $how"
    unset QVZ_CB_LOG2 QVZ_CB_FLOW
    if [ $how = host-libm ]; then export QVZ_CB_LOG2=host; fi
    if [ $how = device-flow ]; then export QVZ_CB_FLOW=device; fi
    QVZ_CB_DEBUG=1 timeout -k 10 300 python3 bench.py --config $cfg --steps 4 --warmup 2 --no-cpu-baseline --no-extras 2> $out/err_${cfg}_$how.txt | python3 -c "
import json,sys
d=json.loads(sys.stdin.readline()); print('ms_per_step', d['ms_per_step'], 'codebook phase ms', d['step_phase_ms']['codebook'], 'bits/sym', d['bits_per_symbol'])"
    grep -m3 -i "codebook\|wait" $out/err_${cfg}_$how.txt | cut -c1-220
  done
done > $out/cb.txt 2>&1
cat $out/cb.txt

set -e
timeout -k 10 900 python -m pytest tests -x -q -m gpu > gpurun_out/t_m.log 2>&1 || { tail -30 gpurun_out/t_m.log; exit 1; }
tail -2 gpurun_out/t_m.log
for ov in 0 1; do
QVZ_CODER_OVERLAP=$ov timeout -k 10 300 python bench.py --steps 3 --warmup 1 --no-cpu-baseline > gpurun_out/b_p_$ov.log 2>&1
tail -1 gpurun_out/b_p_$ov.log | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('overlap $ov', d['ms_per_step'], d['bits_per_symbol'], d['phase_seconds'])"
done

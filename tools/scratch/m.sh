set -e
timeout -k 10 900 python -m pytest tests/test_gpu_pipeline.py tests/test_gpu_decoder.py tests/test_gpu_encode.py -x -q -m gpu > gpurun_out/t_m.log 2>&1 || { tail -30 gpurun_out/t_m.log; exit 1; }
tail -2 gpurun_out/t_m.log
for w in 48 64; do
QVZ_WORK_GIB=$w timeout -k 10 300 python bench.py --steps 3 --warmup 1 --no-cpu-baseline > gpurun_out/b_p_$w.log 2>&1
tail -1 gpurun_out/b_p_$w.log | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('work $w', d['ms_per_step'], d['bits_per_symbol'], d['phase_seconds'])"
done

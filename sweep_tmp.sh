mkdir -p gpurun_out
(timeout 1500 python -m pytest tests -m gpu -q -x 2>&1 | tail -8) > gpurun_out/test9.log 2>&1
cat gpurun_out/test9.log
timeout 300 python bench.py --steps 3 --warmup 1 --no-cpu --method ongrid 2>&1 | tail -1 | python3 -c "import sys,json; d=json.loads(sys.stdin.read()); print('ongrid:', d['value'], d['ms_per_step'], d['config']['refine_log'])"

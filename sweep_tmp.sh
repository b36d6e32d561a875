mkdir -p gpurun_out; rm -f gpurun_out/sweep.log
(timeout 1800 python -m pytest tests -m gpu -q -x 2>&1 | tail -12) > gpurun_out/test12.log 2>&1
cat gpurun_out/test12.log
for i in 1 2; do timeout 300 python bench.py --steps 5 --warmup 1 --no-cpu 2>&1 | tail -1 | python3 -c "import sys,json; d=json.loads(sys.stdin.read()); print(d['value'], d['ms_per_step'], d['roofline']['kernel_ms_avg'], d['roofline']['other_kernels_ms_avg'], d['config']['trapping_boxes'])" >> gpurun_out/sweep.log; done
cat gpurun_out/sweep.log

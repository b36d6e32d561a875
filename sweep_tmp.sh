mkdir -p gpurun_out; rm -f gpurun_out/sweep.log
(timeout 1200 python -m pytest tests -m gpu -q -x 2>&1 | tail -15) > gpurun_out/test5.log 2>&1
timeout 300 python bench.py --steps 3 --warmup 1 --no-cpu 2>&1 | tail -1 | python3 -c "import sys,json; d=json.loads(sys.stdin.read()); print(d['value'], d['ms_per_step'], d['roofline']['kernel_ms_avg'], d['roofline']['other_kernels_ms_avg'])" >> gpurun_out/sweep.log
export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/kt5 -- python3 bench.py --steps 3 --warmup 1 --no-cpu > /dev/null 2>&1
cat gpurun_out/test5.log gpurun_out/sweep.log; cut -d, -f1-4 gpurun_out/kt5/*/*kernel_stats.csv | head -20

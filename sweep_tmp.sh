rm -f gpurun_out/sweep.log
for t in 64; do for o in 0 2 1 3; do
  echo "TPB $t OPT $o" >> gpurun_out/sweep.log
  XB_OPT_TPB=$t XB_OPT_TRACE=$o timeout 300 python bench.py --steps 5 --warmup 1 --no-cpu 2>&1 | tail -1 | python3 -c "import sys,json; d=json.loads(sys.stdin.read()); print(d['value'], d['ms_per_step'], d['roofline']['kernel_ms_avg'])" >> gpurun_out/sweep.log
done; done
cat gpurun_out/sweep.log

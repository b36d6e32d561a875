export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/kt18 -- python3 bench.py --steps 3 --warmup 1 --no-cpu > /dev/null 2>&1
python3 - <<'PY'
import csv,glob
f=glob.glob('gpurun_out/kt18/*/*kernel_stats.csv')[0]
tot=0
for r in list(csv.DictReader(open(f)))[:16]:
    print(f"{r['Name'][:48]:50s} calls={r['Calls']:>4s} avg_us={float(r['AverageNs'])/1e3:10.1f} per_step_ms={float(r['TotalDurationNs'])/1e6/4:7.3f}")
PY

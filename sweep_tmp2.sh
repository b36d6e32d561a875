export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/kt14 -- python3 bench.py --steps 3 --warmup 1 --no-cpu > /dev/null 2>&1
rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d gpurun_out/pmc14a -- python3 bench.py --steps 1 --warmup 0 --no-cpu > /dev/null 2>&1
rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d gpurun_out/pmc14b -- python3 bench.py --steps 1 --warmup 0 --no-cpu > /dev/null 2>&1
python3 - <<'PY'
import csv,glob,collections
f=glob.glob('gpurun_out/kt14/*/*kernel_stats.csv')[0]
for r in list(csv.DictReader(open(f)))[:12]:
    print(f"{r['Name'][:48]:50s} calls={r['Calls']:>4s} avg_us={float(r['AverageNs'])/1e3:10.1f} pct={r['Percentage']}")
for d,cn in (('pmc14a','FETCH_SIZE'),('pmc14b','WRITE_SIZE')):
    f=glob.glob(f'gpurun_out/{d}/*/*counter_collection.csv')[0]
    agg=collections.defaultdict(float)
    for r in csv.DictReader(open(f)): agg[r['Kernel_Name'][:40]]+=float(r['Counter_Value'])
    for k,v in sorted(agg.items(), key=lambda kv:-kv[1])[:8]: print(cn, k, 'KB total over 1 step =', round(v))
PY

export TMPDIR=/tmp
mkdir -p gpurun_out/final
rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/final/kt -- python3 bench.py --steps 5 --warmup 1 --no-cpu > gpurun_out/final/bench_kt.log 2>&1
rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d gpurun_out/final/fetch -- python3 bench.py --steps 1 --warmup 0 --no-cpu > /dev/null 2>&1
rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d gpurun_out/final/write -- python3 bench.py --steps 1 --warmup 0 --no-cpu > /dev/null 2>&1
python3 - <<'PY'
import csv,glob,collections
f=glob.glob('gpurun_out/final/kt/*/*kernel_stats.csv')[0]
for r in list(csv.DictReader(open(f)))[:18]:
    print(f"{r['Name'][:48]:50s} calls={r['Calls']:>4s} avg_us={float(r['AverageNs'])/1e3:10.1f} per_step_ms={float(r['TotalDurationNs'])/1e6/6:7.3f}")
for d,cn in (('fetch','FETCH_SIZE'),('write','WRITE_SIZE')):
    f=glob.glob(f'gpurun_out/final/{d}/*/*counter_collection.csv')[0]
    agg=collections.defaultdict(float)
    for r in csv.DictReader(open(f)): agg[r['Kernel_Name'][:40]]+=float(r['Counter_Value'])
    print(cn, 'total KB per step (excl. synth):', round(sum(v for k,v in agg.items() if 'synth' not in k and 'k_fill' not in k)))
    for k,v in sorted(agg.items(), key=lambda kv:-kv[1])[:7]: print('   ', cn, k, round(v))
PY
tail -1 gpurun_out/final/bench_kt.log | cut -c1-400

#!/usr/bin/env python3
"""Sum rocprofv3 --pmc counter values per kernel: tools/pmc_summary.py <dir with fetch/ and write/> [--steady STEPS] > summary.txt

--steady STEPS: the passes ran STEPS steps of the workload (warm-up included); the dispatches of the FIRST step are dropped
(cold caches, first-touch page faults) and the rest is averaged per step -- the steady-state traffic of one step."""
import csv, glob, re, sys
from collections import defaultdict

def short(name):
    name = re.sub(r'^void ', '', name)
    return name.split('(')[0]

root = sys.argv[1]
steady = int(sys.argv[sys.argv.index('--steady') + 1]) if '--steady' in sys.argv else 0
# the kernel sources this summary was taken from (bench.py drops a `traffic` figure whose sources are not the running ones)
import os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
try:
    from bench import source_hash
    print(f'# sources {source_hash()}')
    if steady:
        print(f'# steady state: {steady} steps per pass, the first step dropped, KB per step')
except Exception as err:   # pragma: no cover
    print(f'# sources unknown ({err})')
for sub in ('fetch', 'write'):
    files = glob.glob(f'{root}/{sub}/**/*counter_collection.csv', recursive=True)
    tot, cnt = defaultdict(float), defaultdict(int)
    rows = defaultdict(list)
    for f in files:
        for r in csv.DictReader(open(f)):
            rows[(r['Counter_Name'], short(r['Kernel_Name']))].append((int(r.get('Dispatch_Id', 0)), float(r['Counter_Value'])))
    for k, v in rows.items():
        v.sort()
        if steady and len(v) % steady == 0:      # (a kernel launched the same number of times in every step)
            v = v[len(v) // steady:]
            tot[k] = sum(x for _, x in v) / (steady - 1)
            cnt[k] = 1
        elif steady:
            continue                              # one-off kernels (generator, first-call fills): not part of a steady step
        else:
            tot[k] = sum(x for _, x in v)
            cnt[k] = len(v)
    for (c, k), v in sorted(tot.items(), key=lambda kv: -kv[1]):
        if v >= 1000:
            print(f'{c:<11} {k:<30} launches={cnt[(c, k)]:3d} KB={int(v):12d}')

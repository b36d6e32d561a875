#!/usr/bin/env python3
"""Sum rocprofv3 --pmc counter values per kernel: tools/pmc_summary.py <dir with fetch/ and write/> > summary.txt"""
import csv, glob, re, sys
from collections import defaultdict

def short(name):
    name = re.sub(r'^void ', '', name)
    return name.split('(')[0]

root = sys.argv[1]
# the kernel sources this summary was taken from (bench.py drops a `traffic` figure whose sources are not the running ones)
import os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
try:
    from bench import source_hash
    print(f'# sources {source_hash()}')
except Exception as err:   # pragma: no cover
    print(f'# sources unknown ({err})')
for sub in ('fetch', 'write'):
    files = glob.glob(f'{root}/{sub}/**/*counter_collection.csv', recursive=True)
    tot, cnt = defaultdict(float), defaultdict(int)
    for f in files:
        for r in csv.DictReader(open(f)):
            k = (r['Counter_Name'], short(r['Kernel_Name']))
            tot[k] += float(r['Counter_Value'])
            cnt[k] += 1
    for (c, k), v in sorted(tot.items(), key=lambda kv: -kv[1]):
        if v >= 1000:
            print(f'{c:<11} {k:<30} launches={cnt[(c, k)]:3d} KB={int(v):12d}')

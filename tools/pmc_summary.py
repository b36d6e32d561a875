#!/usr/bin/env python3
"""Sum rocprofv3 --pmc counter values per kernel: tools/pmc_summary.py <dir with fetch/ and write/> > summary.txt"""
import csv, glob, re, sys
from collections import defaultdict

def short(name):
    name = re.sub(r'^void ', '', name)
    return name.split('(')[0]

root = sys.argv[1]
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

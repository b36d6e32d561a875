#!/usr/bin/env python3
"""Per kernel and counter, the sum over the dispatches of every rocprofv3 --pmc pass under a directory
(tools/pmc_pass.sh writes one sub-directory per pass):  tools/pmc_counters.py <dir> [kernel name filter ...] > summary.txt"""
import csv
import glob
import os
import re
import sys
from collections import defaultdict

root, wanted = sys.argv[1], sys.argv[2:]
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
try:
    from bench import source_hash
    print(f'# sources {source_hash()}')
except Exception as err:   # pragma: no cover
    print(f'# sources unknown ({err})')
tot, cnt = defaultdict(float), defaultdict(int)
for f in glob.glob(f'{root}/**/*counter_collection.csv', recursive=True):
    for r in csv.DictReader(open(f)):
        name = re.sub(r'^void ', '', r['Kernel_Name']).split('(')[0].split('<')[0]
        if wanted and not any(w in name for w in wanted):
            continue
        k = (name, r['Counter_Name'])
        tot[k] += float(r['Counter_Value'])
        cnt[k] += 1
for (name, c), v in sorted(tot.items()):
    if not name.startswith('__amd') and v:
        print(f'{name:<28} {c:<26} launches={cnt[(name, c)]:3d} value={v:.4g}')

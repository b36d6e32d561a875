#!/usr/bin/env python3
"""Sum every rocprofv3 --pmc counter per kernel over the pass directories of tools/pmc_pass.sh:
   tools/pmc_sq_summary.py gpurun_out/pmc_<tag> [kernel-name-prefix ...] > summary.txt"""
import csv, glob, re, sys
from collections import defaultdict

root, want = sys.argv[1], sys.argv[2:]
tot, cnt = defaultdict(float), defaultdict(int)
for f in glob.glob(f'{root}/**/*counter_collection.csv', recursive=True):
    for r in csv.DictReader(open(f)):
        k = re.sub(r'^void ', '', r['Kernel_Name']).split('(')[0].split('<')[0]
        if want and not any(k.startswith(w) for w in want):
            continue
        tot[(k, r['Counter_Name'])] += float(r['Counter_Value'])
        cnt[(k, r['Counter_Name'])] += 1
for (k, c), v in sorted(tot.items()):
    print(f'{k:<28} {c:<22} launches={cnt[(k, c)]:3d} value={v:.4g}')

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

# derived per kernel (MI355X: 1024 SIMDs, GRBM_GUI_ACTIVE counts every XCD: / 8 = the kernel's cycles; SQ_*_CYCLES in quad-cycles)
names = sorted({k for k, _ in tot})
print('# derived: valu_busy = SQ_ACTIVE_INST_VALU x 4 / 1024 SIMDs / (GRBM_GUI_ACTIVE / 8); lds_conflict_ratio = SQ_LDS_BANK_CONFLICT / '
      'SQ_ACTIVE_INST_LDS; lanes = SQ_THREAD_CYCLES_VALU / SQ_ACTIVE_INST_VALU; salu_per_valu = SQ_INSTS_SALU / SQ_INSTS_VALU')
for k in names:
    if k.startswith('__amd'):
        continue
    g = lambda c: tot.get((k, c), 0.0)
    if not g('GRBM_GUI_ACTIVE') or not g('SQ_ACTIVE_INST_VALU'):
        continue
    cyc = g('GRBM_GUI_ACTIVE') / 8.0
    line = f'{k:<28} derived                    valu_busy={g("SQ_ACTIVE_INST_VALU") * 4 / 1024 / cyc:.2f}'
    if g('SQ_ACTIVE_INST_LDS'):
        line += f' lds_conflict_ratio={g("SQ_LDS_BANK_CONFLICT") / g("SQ_ACTIVE_INST_LDS"):.2f}'
    line += f' lanes={g("SQ_THREAD_CYCLES_VALU") / g("SQ_ACTIVE_INST_VALU"):.1f}'
    if g('SQ_INSTS_VALU'):
        line += f' salu_per_valu={g("SQ_INSTS_SALU") / g("SQ_INSTS_VALU"):.2f}'
    if g('TCP_GATE_EN1_sum'):
        line += f' tcp_pending_stall={g("TCP_PENDING_STALL_CYCLES_sum") / g("TCP_GATE_EN1_sum"):.2f}'
    print(line)

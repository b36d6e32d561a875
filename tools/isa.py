#!/usr/bin/env python3
"""Device ISA of libbader_hip.so's kernels (cross-compiles here, no GPU needed): per kernel the register / occupancy
remarks and an instruction histogram; `--dump NAME` writes that kernel's assembly to /tmp/asm/NAME.s.

    python tools/isa.py [substring of a kernel name ...] [--dump]
"""
import collections
import os
import re
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
OUT = '/tmp/asm'


def main():
    args = [a for a in sys.argv[1:] if not a.startswith('--')]
    dump = '--dump' in sys.argv
    os.makedirs(OUT, exist_ok=True)
    src = os.path.join(ROOT, 'pybader_amd', 'csrc', 'bader_hip.hip')
    cmd = ['/opt/rocm/bin/hipcc', '--offload-arch=gfx950', '-O3', '-std=c++17', '-fPIC', '-ffp-contract=off', '-fno-fast-math',
           '-S', '--cuda-device-only', '-Rpass-analysis=kernel-resource-usage', '-o', f'{OUT}/bader.s', src]
    r = subprocess.run(cmd, stderr=subprocess.PIPE, text=True)
    if r.returncode:
        sys.exit(r.stderr[-4000:])
    usage = {}
    cur = None
    for line in r.stderr.splitlines():
        m = re.search(r'remark: Function Name: (\S+)', line)
        if m:
            cur = m.group(1)
            usage[cur] = {}
        m = re.search(r'remark:\s+(TotalSGPRs|VGPRs|ScratchSize \[bytes/lane\]|Occupancy \[waves/SIMD\]|LDS Size \[bytes/block\]|VGPRs Spill): (\d+)', line)
        if m and cur:
            usage[cur][m.group(1).replace(' [bytes/lane]', '').replace(' [waves/SIMD]', '').replace(' Size [bytes/block]', '')] = int(m.group(2))
    text = open(f'{OUT}/bader.s').read().splitlines()
    starts = {}
    for i, l in enumerate(text):
        m = re.match(r'^(_Z\w+):\s+; @', l)
        if m:
            starts[m.group(1)] = i
    demangle = subprocess.run(['c++filt'] + list(starts), stdout=subprocess.PIPE, text=True).stdout.splitlines()
    for (name, i), pretty in zip(starts.items(), demangle):
        if args and not any(a in pretty for a in args):
            continue
        j = i
        while 's_endpgm' not in text[j]:
            j += 1
        body = text[i:j + 1]
        ops = collections.Counter(l.split()[0] for l in body if re.match(r'^\s+[a-z]', l))
        valu = sum(n for o, n in ops.items() if o.startswith('v_'))
        salu = sum(n for o, n in ops.items() if o.startswith('s_'))
        mem = sum(n for o, n in ops.items() if o.startswith(('global_', 'buffer_', 'flat_', 'scratch_')))
        lds = sum(n for o, n in ops.items() if o.startswith('ds_'))
        f64 = sum(n for o, n in ops.items() if o.startswith('v_') and 'f64' in o)
        slow = sum(n for o, n in ops.items() if o in ('v_mad_u64_u32', 'v_mul_lo_u32', 'v_mul_hi_u32', 'v_mad_i64_i32', 'v_mul_hi_i32'))
        print(f'{pretty.split("(")[0]}: {usage.get(name, {})}\n    static: VALU {valu} (f64 {f64}, quarter-rate int {slow}) SALU {salu} VMEM {mem} LDS {lds}')
        if dump:
            short = re.sub(r'\W+', '_', pretty.split('(')[0])[:60]
            open(f'{OUT}/{short}.s', 'w').write('\n'.join(body) + '\n')
            print(f'    -> {OUT}/{short}.s')


if __name__ == '__main__':
    main()

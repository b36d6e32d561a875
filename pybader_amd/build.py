"""Builds libbader_hip.so in-tree with hipcc for gfx950 (cross-compiles without a GPU)."""
import os
import shutil
import subprocess

HERE = os.path.dirname(os.path.abspath(__file__))
SRC = os.path.join(HERE, 'csrc', 'bader_hip.hip')
import glob

DEPS = [SRC, os.path.join(os.path.dirname(HERE), 'include', 'bader_hip.h')] + glob.glob(os.path.join(HERE, 'csrc', '*.h'))
LIB = os.path.join(HERE, 'libbader_hip.so')

# -ffp-contract=off: the reference's float64 expressions are separate multiply/add;
# hipcc contracts to FMA by default, which would break bit-exact trajectories.
FLAGS = ['--offload-arch=gfx950', '-O3', '-std=c++17', '-fPIC', '-shared', '-ffp-contract=off',
         '-fno-fast-math', '-Wall', '-Wno-unused-result', '-Wno-unused-value']


def hipcc():
    for cand in (shutil.which('hipcc'), '/opt/rocm/bin/hipcc'):
        if cand and os.path.exists(cand):
            return cand
    raise RuntimeError('hipcc not found')


def stale():
    if not os.path.exists(LIB):
        return True
    t = os.path.getmtime(LIB)
    return any(os.path.getmtime(d) > t for d in DEPS)


def build_library(force=False, verbose=False, extra=()):
    if force or stale():
        cmd = [hipcc()] + FLAGS + list(extra) + ['-o', LIB, SRC]
        if verbose:
            print(' '.join(cmd))
        subprocess.check_call(cmd)
    return LIB


if __name__ == '__main__':
    print(build_library(force=True, verbose=True))

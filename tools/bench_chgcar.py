#!/usr/bin/env python3
"""Throughput of the CHGCAR density-block path on one GPU:

    python tools/bench_chgcar.py [--size 512]

Builds the text of a size^3 density block (VASP layout: 5 numbers per line, ' 0.dddddddddddE+ee'), parses
it with xb_parse_density_text (timed: PCIe upload of the text + device parse, result resident in HBM),
checks EVERY value against exact host arithmetic (mantissa / 10^k is one correctly rounded operation), and
times the reference's own route on a bounded sample (`array[a:b] = text.split()`, numpy's string -> float64).
(The oracle is test infrastructure and is not used here; tests/test_gpu_chgcar.py holds the parity checks.)"""
import argparse
import json
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))


def make_text(n, seed=3):
    rng = np.random.default_rng(seed)
    per_line, width = 5, 18
    lines = (n + per_line - 1) // per_line
    buf = np.full((lines, per_line * width + 1), ord(' '), dtype=np.uint8)
    buf[:, -1] = ord('\n')
    digits = rng.integers(0, 10, size=(lines * per_line, 11), dtype=np.uint8)
    digits[:, 0] = np.maximum(digits[:, 0], 1)                       # normalised mantissa 0.1 .. 0.999
    expo = rng.integers(0, 7, size=lines * per_line, dtype=np.int64)  # E-03 .. E+03
    tok = np.full((lines * per_line, width), ord(' '), dtype=np.uint8)
    tok[:, 1] = ord('0'); tok[:, 2] = ord('.')
    tok[:, 3:14] = digits + ord('0')
    tok[:, 14] = ord('E')
    tok[:, 15] = np.where(expo < 3, ord('-'), ord('+'))
    tok[:, 16] = ord('0')
    tok[:, 17] = np.abs(expo - 3) + ord('0')
    buf[:, :-1] = tok.reshape(lines, per_line * width)
    mant = np.zeros(lines * per_line, dtype=np.int64)
    for k in range(11):
        mant = mant * 10 + digits[:, k]
    return buf.reshape(-1), mant[:n], (expo - 3)[:n]


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument('--size', type=int, default=512)
    ap.add_argument('--cpu-sample', type=int, default=4_000_000)
    args = ap.parse_args()
    from pybader_amd import _lib
    shape = (args.size,) * 3
    n = args.size ** 3
    text, mant, expo = make_text(n)
    divisor = 216.0
    ctx = _lib.Context(0)
    ctx.set_grid(shape, np.zeros(27), np.zeros(9))
    times = []
    for _ in range(3):
        t0 = time.perf_counter()
        n_tok, n_host = ctx.parse_density_text(text, divisor)
        ctx.sync()
        times.append(time.perf_counter() - t0)
    dt = min(times)
    got = ctx.download_density()
    # exact expectation: value = mant * 10^(expo - 11); one correctly rounded operation on exact operands
    e10 = expo - 11
    p = np.power(10.0, np.abs(e10).astype(np.float64))            # exact powers of ten up to 1e14
    val = np.where(e10 < 0, mant.astype(np.float64) / p, mant.astype(np.float64) * p) / divisor
    want = np.ascontiguousarray(val.reshape(shape[::-1]).transpose(2, 1, 0))
    ok = bool(np.array_equal(got, want))
    # CPU: numpy's string -> float64 on a sample -- the reference's route (io/vasp.py:97-104)
    m = min(args.cpu_sample, n)
    sample = text[:(m // 5) * 91].tobytes()
    ms = (m // 5) * 5
    t0 = time.perf_counter()
    arr = np.zeros(ms)
    arr[:] = sample.decode().split()
    arr /= divisor
    t_numpy = time.perf_counter() - t0
    print(json.dumps({
        'workload': f'{args.size}^3 CHGCAR density block, {text.size / 1e9:.2f} GB of text, host (pageable) -> resident rho',
        'all_values_bit_exact': ok, 'tokens': int(n_tok), 'host_fallback_tokens': int(n_host),
        'gpu_seconds_incl_pcie_upload': dt, 'gpu_Mvalues_per_s': n / dt / 1e6, 'gpu_text_GB_per_s': text.size / dt / 1e9,
        'cpu_numpy_split_Mvalues_per_s': ms / t_numpy / 1e6,
        'cpu_sample_values': ms, 'cpu_cores': 1}))
    ctx.close()


if __name__ == '__main__':
    main()

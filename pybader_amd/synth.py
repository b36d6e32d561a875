"""Bit-reproducible synthetic charge densities (workload generator, not part of the hot path).

The reference ships no data files and no generator; BASELINE.json's configs are "synthetic
Gaussian-atom" grids.  Golden fixtures, the CPU oracle and the GPU bench must all see *the same
float64 bits*, produced by three independent implementations (this numpy one, the C one in
``oracle/bader_oracle.c:orc_synth_density`` and the HIP kernel ``k_synth_density``).  So the density
uses only IEEE-754 basic operations in a fixed order (no ``exp``, no library reductions, no FMA):

    f      = (i/nx, j/ny, k/nz)                      fractional voxel position
    d      = f - c_a ;  d -= rint(d)                 minimum image (round-half-even)
    x_m    = (d0*L[0][m] + d1*L[1][m]) + d2*L[2][m]  cartesian, m = 0..2
    r2     = (x0*x0 + x1*x1) + x2*x2
    t      = max(0, 1 - r2 / (2048*s_a*s_a)) ; t <- t*t ten times   (= t**1024 ~ exp(-r2/(2 s^2)))
    rho   += A_a * t                                 atoms in table order, rho starts at `background`

Layout is the reference's: C-order ``[x][y][z]``, z fastest (io/vasp.py:102-103).
"""
import hashlib

import numpy as np

# (frac_x, frac_y, frac_z, sigma [Angstrom], amplitude) -- 8 atoms, generic (no symmetric ties)
ATOMS8 = np.array([
    [0.2310, 0.2690, 0.2470, 0.42, 7.50],
    [0.7610, 0.2380, 0.2710, 0.36, 5.25],
    [0.2570, 0.7430, 0.2290, 0.47, 6.00],
    [0.7390, 0.7710, 0.2630, 0.33, 3.75],
    [0.2430, 0.2330, 0.7570, 0.39, 4.50],
    [0.7730, 0.2610, 0.7390, 0.45, 8.00],
    [0.2290, 0.7590, 0.7690, 0.31, 2.50],
    [0.7510, 0.7470, 0.7330, 0.50, 6.75],
], dtype=np.float64)

CUBIC6 = np.array([[6.0, 0.0, 0.0], [0.0, 6.0, 0.0], [0.0, 0.0, 6.0]], dtype=np.float64)
# the triclinic cell of SURVEY.md section 8(d)
TRICLINIC = np.array([[6.0, 0.0, 0.0], [1.5, 5.5, 0.0], [0.7, 1.1, 6.2]], dtype=np.float64)
BACKGROUND = 0.015625  # 2**-6, exact


def atoms_cartesian(atoms=ATOMS8, lattice=CUBIC6):
    """Cartesian atom positions as the reference's Bader() expects them (interface.py:116)."""
    return np.ascontiguousarray(np.dot(atoms[:, :3], lattice))


def synth_density(shape, lattice=CUBIC6, atoms=ATOMS8, background=BACKGROUND, x_range=None):
    """Return the float64 C-order density of `shape`; `x_range=(x0,x1)` builds only that slab."""
    nx, ny, nz = (int(s) for s in shape)
    x0, x1 = (0, nx) if x_range is None else x_range
    L = np.asarray(lattice, dtype=np.float64)
    fx = (np.arange(x0, x1, dtype=np.float64) / np.float64(nx))[:, None, None]
    fy = (np.arange(ny, dtype=np.float64) / np.float64(ny))[None, :, None]
    fz = (np.arange(nz, dtype=np.float64) / np.float64(nz))[None, None, :]
    rho = np.full((x1 - x0, ny, nz), background, dtype=np.float64)
    for a in np.asarray(atoms, dtype=np.float64):
        d0 = fx - a[0]
        d0 = d0 - np.rint(d0)
        d1 = fy - a[1]
        d1 = d1 - np.rint(d1)
        d2 = fz - a[2]
        d2 = d2 - np.rint(d2)
        r2 = None
        for m in range(3):
            xm = (d0 * L[0, m] + d1 * L[1, m]) + d2 * L[2, m]
            sq = xm * xm
            r2 = sq if r2 is None else (r2 + sq)
        denom = np.float64(2048.0) * a[3] * a[3]
        t = np.float64(1.0) - r2 / denom
        np.maximum(t, 0.0, out=t)
        for _ in range(10):
            t = t * t
        rho = rho + a[4] * t
    return rho


def sha256(a):
    return hashlib.sha256(np.ascontiguousarray(a).tobytes()).hexdigest()


def hash_noise(shape, seed):
    """Uniform [0, 1) float64 noise per voxel from integer arithmetic only (splitmix64 of the C-order voxel
    index + seed; top 53 bits scaled by 2**-53): bit-reproducible on any numpy, no RNG implementation involved."""
    n = int(np.prod(shape))
    with np.errstate(over='ignore'):
        z = np.arange(n, dtype=np.uint64) + np.uint64(seed) * np.uint64(0x9E3779B97F4A7C15)
        z = z + np.uint64(0x9E3779B97F4A7C15)
        z = (z ^ (z >> np.uint64(30))) * np.uint64(0xBF58476D1CE4E5B9)
        z = (z ^ (z >> np.uint64(27))) * np.uint64(0x94D049BB133111EB)
        z = z ^ (z >> np.uint64(31))
    return ((z >> np.uint64(11)).astype(np.float64) * np.float64(2.0 ** -53)).reshape(tuple(int(s) for s in shape))


def round_sig(rho, digits):
    """Round to `digits` significant decimal digits the way a '%.{digits-1}E' text file does (CHG files carry 5)."""
    flat = np.ascontiguousarray(rho, dtype=np.float64).reshape(-1)
    out = np.array([float(('%.' + str(int(digits) - 1) + 'E') % v) for v in flat], dtype=np.float64)
    return out.reshape(rho.shape)


def rough_density(shape, lattice=CUBIC6, atoms=ATOMS8, noise=0.0, seed=7, sig_digits=0, quantum=0.0):
    """The non-smooth test densities: the synthetic atoms + `noise` * hash_noise, optionally rounded to `sig_digits`
    significant digits (exact ties in every low-density region, like a CHG file) or to multiples of `quantum`
    (plateaus of exactly equal density)."""
    rho = synth_density(shape, lattice, atoms)
    if noise:
        rho = rho + np.float64(noise) * hash_noise(shape, seed)
    if sig_digits:
        rho = round_sig(rho, sig_digits)
    if quantum:
        rho = np.round(rho / np.float64(quantum)) * np.float64(quantum)
    return np.ascontiguousarray(rho)


def atoms_jittered_grid(k=6, seed=5):
    """k^3 atoms on a jittered k x k x k lattice of fractional positions with random widths and charges (rows as ATOMS8):
    the many-atom cell of bench.py's user leg and of the golden case c128_216atoms (k = 6: 216 atoms)."""
    rng = np.random.default_rng(seed)
    cells = np.stack(np.meshgrid(*(np.arange(k),) * 3, indexing='ij'), -1).reshape(-1, 3)
    frac = (cells + 0.5 + 0.18 * (rng.random(cells.shape) - 0.5)) / k
    return np.concatenate([frac, 0.09 + 0.04 * rng.random((len(frac), 1)), 2.0 + 6.0 * rng.random((len(frac), 1))], 1)

// k_text.h -- device kernels of libbader_hip.so: the density block of a CHGCAR / CHG file, text -> resident rho.
// Included by bader_hip.hip (one translation unit); see bader_kernels.h for the common device code.
#pragma once

// ---------------------------------------------------------------------------------------------
// io/vasp.py:90-104 reads the density block as whitespace separated decimal tokens, converts them with
// numpy's (correctly rounded) string -> float64, reshapes Fortran order (x fastest) into [x][y][z] and
// divides by the cell volume (io/vasp.py:147-149).  At 512^3 that is 1.3e8 tokens / 2.4 GB of text and
// dominates the end-to-end time once the partitioning takes milliseconds.
// Here the text is uploaded as it is and parsed on the device:
//   k_text_count   token starts per block (a token starts at a non-blank byte that follows a blank)
//   (exclusive scan of the block counts)
//   k_text_parse   every token start: decimal -> float64, divided, stored at its [x][y][z] position
// Decimal -> binary is exact here: a token with a mantissa m < 2^53 and a decimal exponent |e| <= 22 is
// m * 10^e or m / 10^-e with both operands exact doubles, i.e. ONE correctly rounded IEEE operation
// (Clinger's fast path) -- every number a VASP / pybader writer produces (<= 17 digits would not fit, 11-12
// do).  Anything else (longer mantissas, huge exponents, nan/inf, malformed tokens) is listed and converted
// by strtod on the host, so the result equals numpy's for every input the reference accepts.
// ---------------------------------------------------------------------------------------------
#define TXT_BYTES 16  // bytes per thread
__device__ __forceinline__ bool txt_blank(unsigned char ch) { return ch == ' ' || (ch >= 9 && ch <= 13); }

// bit k of the result: byte base+k starts a token
__device__ __forceinline__ unsigned int txt_starts(const unsigned char *__restrict__ t, long long base, long long n) {
    unsigned char b[TXT_BYTES];
    if (base + TXT_BYTES <= n) *reinterpret_cast<uint4 *>(b) = *reinterpret_cast<const uint4 *>(t + base);
    else {
#pragma unroll
        for (int k = 0; k < TXT_BYTES; k++) b[k] = base + k < n ? t[base + k] : (unsigned char)' ';
    }
    bool prev_blank = base == 0 ? true : txt_blank(t[base - 1]);
    unsigned int m = 0;
#pragma unroll
    for (int k = 0; k < TXT_BYTES; k++) {
        const bool bl = txt_blank(b[k]);
        if (!bl && prev_blank) m |= 1u << k;
        prev_blank = bl;
    }
    return m;
}
__global__ __launch_bounds__(TPB) void k_text_count(const unsigned char *__restrict__ t, long long n, int *block_count) {
    const long long base = ((long long)blockIdx.x * TPB + threadIdx.x) * TXT_BYTES;
    const int cnt = base < n ? __popc(txt_starts(t, base, n)) : 0;
    int total;
    block_scan_excl(cnt, total);
    if (threadIdx.x == 0) block_count[blockIdx.x] = total;
}
// exclusive scan of n ints, 2048 per block: out[i] = sum of in[0..i) within the block's range, block_sums[b] =
// the block's total (scanned in turn by the caller)
__global__ __launch_bounds__(TPB) void k_scan_2048(const int *__restrict__ in, int n, int *__restrict__ out, int *block_sums) {
    const int base = (blockIdx.x * TPB + threadIdx.x) * 8;
    int v[8], s = 0;
#pragma unroll
    for (int k = 0; k < 8; k++) { v[k] = base + k < n ? in[base + k] : 0; s += v[k]; }
    int total;
    int off = block_scan_excl(s, total);
#pragma unroll
    for (int k = 0; k < 8; k++) {
        if (base + k < n) out[base + k] = off;
        off += v[k];
    }
    if (threadIdx.x == 0) block_sums[blockIdx.x] = total;
}
__global__ __launch_bounds__(TPB) void k_scan_add(int *__restrict__ data, int n, const int *__restrict__ block_off) {
    const int base = (blockIdx.x * TPB + threadIdx.x) * 8;
    const int o = block_off[blockIdx.x];
#pragma unroll
    for (int k = 0; k < 8; k++)
        if (base + k < n) data[base + k] += o;
}

// one token -> double.  Returns false when the token needs the host (strtod) path.
__device__ __forceinline__ bool txt_parse(const unsigned char *__restrict__ p, const unsigned char *__restrict__ end,
                                          const double *__restrict__ p10, double &out) {
    bool neg = false;
    if (p < end && (*p == '-' || *p == '+')) { neg = *p == '-'; p++; }
    unsigned long long m = 0;
    int digits = 0, frac = 0, nd = 0;  // digits seen, digits after the point, significant digits in m
    bool point = false;
    for (; p < end; p++) {
        const unsigned char ch = *p;
        if (ch >= '0' && ch <= '9') {
            digits++;
            if (m != 0 || ch != '0') {
                if (++nd > 18) return false;  // would not fit / not exact: host
                m = m * 10ull + (unsigned long long)(ch - '0');
            }
            if (point) frac++;
        } else if (ch == '.' && !point) point = true;
        else break;
    }
    if (digits == 0) return false;  // nan, inf, garbage
    int ex = 0;
    if (p < end && (*p == 'e' || *p == 'E')) {
        p++;
        bool eneg = false;
        if (p < end && (*p == '-' || *p == '+')) { eneg = *p == '-'; p++; }
        int ed = 0;
        for (; p < end && *p >= '0' && *p <= '9'; p++) {
            if (++ed > 4) return false;
            ex = ex * 10 + (*p - '0');
        }
        if (ed == 0) return false;
        if (eneg) ex = -ex;
    }
    if (p < end && !txt_blank(*p)) return false;  // trailing characters: let the host decide (it will refuse)
    const int e10 = ex - frac;
    double v;
    if (m == 0) v = 0.;
    else if (m < (1ull << 53) && e10 >= -22 && e10 <= 22) {
        const double dm = (double)m;  // exact
        v = e10 < 0 ? dm / p10[-e10] : dm * p10[e10];  // one correctly rounded operation on exact operands
    } else
        return false;
    out = neg ? -v : v;
    return true;
}
// block_off: exclusive scan of the block counts.  Token i of the file (Fortran order, x fastest) is voxel
// (i % nx, (i / nx) % ny, i / (nx ny)); tokens beyond nx*ny*nz are ignored.  todo: (byte offset, token index)
// of the tokens left to the host.
__global__ __launch_bounds__(TPB) void k_text_parse(const unsigned char *__restrict__ t, long long n,
                                                    const int *__restrict__ block_off, const double *__restrict__ p10,
                                                    double divisor, int nx, int ny, int nz, double *__restrict__ rho,
                                                    long long *todo, int *n_todo, int todo_cap) {
    const long long base = ((long long)blockIdx.x * TPB + threadIdx.x) * TXT_BYTES;
    const unsigned int starts = base < n ? txt_starts(t, base, n) : 0u;
    int total;
    long long idx = (long long)block_off[blockIdx.x] + block_scan_excl(__popc(starts), total);
    const long long nvox = (long long)nx * ny * nz;
    for (unsigned int m = starts; m; m &= m - 1, idx++) {
        if (idx >= nvox) break;
        const long long at = base + (__ffs(m) - 1);
        double v;
        if (txt_parse(t + at, t + n, p10, v)) {
            const int x = (int)(idx % nx);
            const long long r = idx / nx;
            const int y = (int)(r % ny), z = (int)(r / ny);
            rho[((long long)x * ny + y) * nz + z] = v / divisor;  // io/vasp.py:147-149, true division
        } else {
            const int k = atomicAdd(n_todo, 1);
            if (k < todo_cap) { todo[2 * k] = at; todo[2 * k + 1] = idx; }
        }
    }
}

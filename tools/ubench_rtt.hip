// Round-trip times on one MI355X that DESIGN.md 4.2 quotes: a chain of dependent device-scope atomicAdd (with result) / sc1 loads on
// random 8-byte words of an 8 MB buffer, and pairs of workgroup barriers.  hipcc --offload-arch=gfx950 -O3 -o rtt tools/ubench_rtt.hip
// Measured (round 4): atomicAdd 0.40 us (1 wave), 0.71 us (256 waves), 5.6 us (2048 waves, saturated); sc1 load 0.34 / 0.36 / 1.08 us;
// two barriers 0.064 / 0.069 / 0.081 us for 64 / 256 / 1024 threads.
// Round 5: ONE workgroup with 64 / 256 / 640 / 1024 lanes, one device-scope atomic per lane and op: 0.50 / 0.44 / 1.08 / 1.73 us -- a
// compute unit's address path issues ~600 scattered atomics per microsecond (why k_ec_chase sheds long queues to other workgroups);
// with WORKGROUP scope (no sc1: the XCD's own L2) 0.40 / 0.43 / 1.07 us -- the scope is not what a round trip costs on this buffer.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
__global__ void chain_atomic(unsigned long long *p, int n, int mask, unsigned long long *out) {
    unsigned long long idx = threadIdx.x * 977 + blockIdx.x * 131;
    unsigned long long acc = 0;
    for (int i = 0; i < n; i++) {
        unsigned long long o = atomicAdd(p + (idx & mask), 1ull);
        acc += o;
        idx = idx * 6364136223846793005ull + 1442695040888963407ull + (o & 1);
        idx >>= 11;
    }
    if (acc == 12345) out[0] = acc;
}
// the same with WORKGROUP scope: the atomic executes in the XCD's own L2 (no sc1) -- only coherent among the workgroups of ONE XCD
__global__ void chain_atomic_l2(unsigned long long *p, int n, int mask, unsigned long long *out) {
    unsigned long long idx = threadIdx.x * 977 + blockIdx.x * 131;
    unsigned long long acc = 0;
    for (int i = 0; i < n; i++) {
        unsigned long long o = __hip_atomic_fetch_add(p + (idx & mask), 1ull, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
        acc += o;
        idx = idx * 6364136223846793005ull + 1442695040888963407ull + (o & 1);
        idx >>= 11;
    }
    if (acc == 12345) out[0] = acc;
}
__global__ void chain_load(const unsigned long long *p, int n, int mask, unsigned long long *out) {
    unsigned long long idx = threadIdx.x * 977 + blockIdx.x * 131;
    unsigned long long acc = 0;
    for (int i = 0; i < n; i++) {
        unsigned long long o = __hip_atomic_load(p + (idx & mask), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        acc += o;
        idx = idx * 6364136223846793005ull + 1442695040888963407ull + (o & 1);
        idx >>= 11;
    }
    if (acc == 12345) out[0] = acc;
}
__global__ void chain_barrier(int n, int *out) {
    __shared__ int s;
    int acc = 0;
    for (int i = 0; i < n; i++) { if (threadIdx.x == 0) s = i; __syncthreads(); acc += s; __syncthreads(); }
    if (acc == 12345) out[0] = acc;
}
int main() {
    const int words = 1 << 20;   // 8 MB
    unsigned long long *p, *out; int *iout;
    hipMalloc(&p, words * 8); hipMemset(p, 0, words * 8); hipMalloc(&out, 8); hipMalloc(&iout, 4);
    hipEvent_t a, b; hipEventCreate(&a); hipEventCreate(&b);
    const int n = 2000;
    for (int cfg = 0; cfg < 6; cfg++) {
        int blocks = cfg % 3 == 0 ? 1 : (cfg % 3 == 1 ? 256 : 2048), threads = 64;
        for (int rep = 0; rep < 2; rep++) {
            hipEventRecord(a);
            if (cfg < 3) chain_atomic<<<blocks, threads>>>(p, n, words - 1, out);
            else chain_load<<<blocks, threads>>>(p, n, words - 1, out);
            hipEventRecord(b); hipEventSynchronize(b);
            float ms; hipEventElapsedTime(&ms, a, b);
            if (rep) printf("%s blocks %4d x %d lanes: %.3f us per dependent op\n", cfg < 3 ? "atomicAdd(ret) device scope" : "sc1 load", blocks, threads, ms * 1e3 / n);
        }
    }
    // one workgroup, more waves: does ONE compute unit's address path limit a round of the chase (k_ec_chase: ~540 scattered atomics
    // from the workgroup that carries a dependency front)?
    for (int threads : {64, 256, 640, 1024}) {
        for (int rep = 0; rep < 2; rep++) {
            hipEventRecord(a); chain_atomic<<<1, threads>>>(p, n, words - 1, out); hipEventRecord(b); hipEventSynchronize(b);
            float ms; hipEventElapsedTime(&ms, a, b);
            if (rep) printf("atomicAdd(ret) device scope, ONE workgroup of %4d lanes: %.3f us per dependent op (every lane one atomic per op)\n", threads, ms * 1e3 / n);
        }
    }
    for (int threads : {64, 256, 640}) {
        for (int rep = 0; rep < 2; rep++) {
            hipEventRecord(a); chain_atomic_l2<<<1, threads>>>(p, n, words - 1, out); hipEventRecord(b); hipEventSynchronize(b);
            float ms; hipEventElapsedTime(&ms, a, b);
            if (rep) printf("atomicAdd(ret) WORKGROUP scope (the XCD's L2), one workgroup of %4d lanes: %.3f us per dependent op\n", threads, ms * 1e3 / n);
        }
    }
    for (int threads : {64, 256, 1024}) {
        for (int rep = 0; rep < 2; rep++) {
            hipEventRecord(a); chain_barrier<<<256, threads>>>(20000, iout); hipEventRecord(b); hipEventSynchronize(b);
            float ms; hipEventElapsedTime(&ms, a, b);
            if (rep) printf("2 barriers, %d threads: %.3f us per pair\n", threads, ms * 1e3 / 20000);
        }
    }
    return 0;
}

// What a device-wide barrier inside ONE kernel costs on an MI355X (round 5: is folding the dozen small dependent launches of the
// region growth into one persistent kernel worth it?).  G workgroups, all resident; per barrier every workgroup writes a word of
// global memory that its neighbour workgroup reads after the barrier (checked: the barrier must make it visible across XCDs).
//   hipcc --offload-arch=gfx950 -O3 -o gridbar tools/ubench_gridbar.hip
// Variants: one counter line for everybody / one counter per XCD-sized group of workgroups + a second level.
#include <hip/hip_runtime.h>
#include <cstdio>
__device__ __forceinline__ void grid_barrier(unsigned *bar, unsigned target) {
    __syncthreads();
    if (threadIdx.x == 0) {
        __threadfence();   // release: this workgroup's global writes, device-wide
        atomicAdd(bar, 1u);
        int spins = 0;
        while (__hip_atomic_load(bar, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) < target) {
            __builtin_amdgcn_s_sleep(1);
            if (++spins > (1 << 22)) break;   // (never: every workgroup is resident)
        }
        __threadfence();   // acquire
    }
    __syncthreads();
}
__global__ void k_bar(unsigned *bar, int *data, int n, int *bad) {
    const int G = gridDim.x;
    int errs = 0;
    for (int i = 0; i < n; i++) {
        if (threadIdx.x == 0) data[blockIdx.x * 32] = i * 1000 + blockIdx.x;
        grid_barrier(bar, (unsigned)(i + 1) * G);
        const int nb = (blockIdx.x + 37) % G;
        const int v = __hip_atomic_load(&data[nb * 32], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        const int w = data[nb * 32];   // (a plain load: what the phases of a fused kernel would use)
        if (v != i * 1000 + nb || w != i * 1000 + nb) errs++;
        grid_barrier(bar + 64, (unsigned)(i + 1) * G);   // (so that nobody overwrites data before its reader has read it)
    }
    if (errs && threadIdx.x == 0) atomicAdd(bad, errs);
}
int main() {
    unsigned *bar; int *data, *bad;
    hipMalloc(&bar, 1024); hipMalloc(&data, 4096 * 32 * 4); hipMalloc(&bad, 4);
    hipEvent_t a, b; hipEventCreate(&a); hipEventCreate(&b);
    const int n = 200;
    for (int G : {8, 32, 64, 128, 256, 512, 1024})
        for (int threads : {64, 512}) {
            if (G * threads > 256 * 2048) continue;
            float best = 1e9f; int hbad = 0;
            for (int rep = 0; rep < 3; rep++) {
                hipMemset(bar, 0, 1024); hipMemset(bad, 0, 4);
                hipEventRecord(a); k_bar<<<G, threads>>>(bar, data, n, bad); hipEventRecord(b); hipEventSynchronize(b);
                float ms; hipEventElapsedTime(&ms, a, b);
                best = ms < best ? ms : best;
                hipMemcpy(&hbad, bad, 4, hipMemcpyDeviceToHost);
            }
            printf("%4d workgroups x %3d threads: %.2f us per barrier (2 per round), stale reads %d\n", G, threads, best * 1e3 / (2 * n), hbad);
        }
    return 0;
}

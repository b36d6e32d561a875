// k_table.h -- device side of libbader_hip.so: tile / brick geometry shared by the kernels of the brick pipeline.
// Included by bader_hip.hip (one translation unit); see bader_kernels.h for the common device code.
//
// Round 4 retired what used to live here: k_grad_field (round 1's 32-byte record for EVERY voxel), the closed seed cubes
// around the maxima (k_box_shells / k_box_stamp, at most 1023 of them), the host-driven brick growth (k_brick_seed,
// k_brick_grow) and k_og_pointer_tiled.  Every grid now goes through passes A / B (k_masks.h) and the device-driven growth
// (k_fused.h); where trapping regions are not built -- grids below 16 voxels on an axis, slabs that cut bricks or whose grid
// is not made of whole bricks -- the records come from k_brick_records over every brick of the table window and the
// trajectories are traced in full.
#pragma once

// the density tile of passes A (k_brick_masks) and of the ongrid pass (k_og_masks): GT_Z / 8 whole 8^3 bricks in a row along z
#define GT_X 8
#define GT_Y 8
#define GT_Z 32

#define BRK 8   // edge of a brick (voxels)
#define BG 8    // edge of the chunk of bricks a workgroup of the region growth iterates in LDS (k_brick_grow_dev)

__device__ __forceinline__ int wrap_any(int v, int n) { v %= n; return v < 0 ? v + n : v; }

// brick_rec[b] := flag for the bricks whose x-brick index lies in [b0, b0 + nb) modulo nb0 (the table window of a slab; the
// whole lattice on one GPU), 0 elsewhere: k_brick_records(walk == nullptr) then writes the records of exactly those bricks
__global__ void k_flag_window_bricks(int nb0, int per_plane, int b0, int nb, unsigned char flag, unsigned char *brick_rec) {
    const int b = blockIdx.x * blockDim.x + threadIdx.x;
    if (b >= nb0 * per_plane) return;
    int d = b / per_plane - b0;
    if (d < 0) d += nb0;
    brick_rec[b] = d < nb ? flag : (unsigned char)0;
}

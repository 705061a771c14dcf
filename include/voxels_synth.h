/* voxels_synth.h — synthetic procedural inputs for the polygonization path (C ABI, host code, OpenMP).
 *
 * The reference ships no terrain generator: surfaces are supplied by the application through VoxelSurface
 * (reference include/VoxelSurface.h:13-41).  These functions play that role for tests and bench.py (SURVEY.md
 * §8(d)): they fill dense Z-up grids (index (z*n + y)*n + x) with exactly the bytes Grid::Create would store —
 * distances quantised by the reference rule sign(v)*ceil(|v|) clamped to +-4 (reference src/VoxelGrid.cpp:37-50).
 * Deterministic: integer hash noise, fixed operation order, no libm in the hash.
 */
#ifndef VOXELS_SYNTH_H
#define VOXELS_SYNTH_H
#include <stdint.h>
#ifdef __cplusplus
extern "C" {
#endif

/* Noise terrain of the global n^3 world, z-planes [z0, z1): fBm height field (4 octaves, base wavelength n/4
 * voxels) + a 3-D noise term for overhangs, distances clamped to +-100 before quantisation; materials 0..2 by
 * dithered height band, blend = smoothstep across the band.  Outputs hold (z1-z0)*n*n bytes each; mat/blend
 * may be NULL. */
void vxs_terrain(uint32_t n, uint32_t z0, uint32_t z1, uint32_t seed, int8_t* dist, uint8_t* mat, uint8_t* blend);
/* The same with a surface style: 0 = the terrain above; 1 = "caves": the 3-D noise term dominates and an isosurface
 * network fills a band of about +-380 voxels around the terrain height (|z - h| / 64 < 6, vx_terrain_math.h) (a workload with surface in a large share of all
 * blocks, for throughput figures that do not depend on a sparse surface). */
void vxs_terrain_ex(uint32_t n, uint32_t z0, uint32_t z1, uint32_t seed, uint32_t style, int8_t* dist, uint8_t* mat, uint8_t* blend);

/* Ball of radius r_frac*n centred in the grid (the survey's known-answer input), same quantisation. */
void vxs_sphere(uint32_t n, uint32_t z0, uint32_t z1, float r_frac, int8_t* dist);

/* BF_Empty of every 16^3 block of a dense n x n x planes field (planes multiple of 16), exactly as the
 * reference's block codec derives it (src/VoxelGrid.cpp:610-672): uniform strict sign of all 4096 voxels AND the
 * run-length form not longer than the raw block.  flags[(n/16)^2 * planes/16] in block-id order. */
void vxs_block_empty_flags(uint32_t n, uint32_t planes, const int8_t* dist, uint8_t* flags);

#ifdef __cplusplus
}
#endif
#endif

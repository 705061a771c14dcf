// vx_terrain_math.h — the synthetic noise terrain (include/voxels_synth.h) as functions of a voxel coordinate, shared by
// the host generator (vx_synth.cpp, g++) and the device generator (k_terrain_*, vx_hip.hip): the same operations in the
// same order, plain fp32 without contraction on both sides, so both produce the same bytes.
//
// This is the VoxelSurface -> Grid step in front of the path (reference src/VoxelGrid.cpp:79-132: sample the surface;
// :37-50: quantise) for the one surface the benchmark uses.
#pragma once
#include <math.h>
#include <stdint.h>

#if defined(__HIPCC__)
#define VXT_HD __host__ __device__ __forceinline__
#else
#define VXT_HD inline
#endif

namespace vxt {

VXT_HD uint32_t hash3(uint32_t x, uint32_t y, uint32_t z, uint32_t seed)
{
	uint32_t h = seed * 0x9E3779B1u + x * 0x85EBCA77u + y * 0xC2B2AE3Du + z * 0x27D4EB2Fu;
	h ^= h >> 15; h *= 0x2C1B3C6Du; h ^= h >> 12; h *= 0x297A2D39u; h ^= h >> 15;
	return h;
}

VXT_HD float lattice(uint32_t x, uint32_t y, uint32_t z, uint32_t seed)
{
	return (float)(hash3(x, y, z, seed) >> 8) * (2.0f / 16777216.0f) - 1.0f; // [-1, 1)
}

VXT_HD float fade(float t) { return t * t * (3.0f - 2.0f * t); }

// value noise, period-free, coordinates in lattice units
VXT_HD float noise2(float x, float y, uint32_t seed)
{
	const float fx = floorf(x), fy = floorf(y);
	const uint32_t ix = (uint32_t)(int)fx, iy = (uint32_t)(int)fy;
	const float u = fade(x - fx), v = fade(y - fy);
	const float a = lattice(ix, iy, 0, seed), b = lattice(ix + 1, iy, 0, seed);
	const float c = lattice(ix, iy + 1, 0, seed), d = lattice(ix + 1, iy + 1, 0, seed);
	const float ab = a + (b - a) * u, cd = c + (d - c) * u;
	return ab + (cd - ab) * v;
}

VXT_HD float noise3(float x, float y, float z, uint32_t seed)
{
	const float fx = floorf(x), fy = floorf(y), fz = floorf(z);
	const uint32_t ix = (uint32_t)(int)fx, iy = (uint32_t)(int)fy, iz = (uint32_t)(int)fz;
	const float u = fade(x - fx), v = fade(y - fy), w = fade(z - fz);
	float r[2];
	for (int k = 0; k < 2; ++k) {
		const float a = lattice(ix, iy, iz + k, seed), b = lattice(ix + 1, iy, iz + k, seed);
		const float c = lattice(ix, iy + 1, iz + k, seed), d = lattice(ix + 1, iy + 1, iz + k, seed);
		const float ab = a + (b - a) * u, cd = c + (d - c) * u;
		r[k] = ab + (cd - ab) * v;
	}
	return r[0] + (r[1] - r[0]) * w;
}

// reference quantisation (src/VoxelGrid.cpp:37-50) followed by the grid's +-4 clamp
VXT_HD int8_t quantise(float value)
{
	float a = ceilf(fabsf(value));
	float b = a * (float)(value > 0 ? 1 : -1);
	if (b > 127.f) b = 127.f;
	int v = (int)b;
	return (int8_t)(v > 4 ? 4 : (v < -4 ? -4 : v));
}

// terrain height above voxel column (x,y) of the n^3 world: 4 octaves of value noise, base wavelength n/4
VXT_HD float height(uint32_t n, uint32_t x, uint32_t y, uint32_t seed)
{
	const float fn = (float)n;
	const float base = fn / 4.0f;
	float amp = 1.0f, freq = 1.0f / base, sum = 0.0f, norm = 0.0f;
	for (int o = 0; o < 4; ++o) {
		sum += amp * noise2((float)x * freq, (float)y * freq, seed + 31u * (uint32_t)o);
		norm += amp; amp *= 0.5f; freq *= 2.0f;
	}
	return fn * 0.5f + 0.25f * fn * (sum / norm);
}

// one voxel: quantised distance, material id, blend
// style 0: the height-field terrain with overhangs (the benchmark's headline workload); style 1: "caves" — the same
// heights, but the distance falls off 64 times slower and the 3-D noise term dominates, so that an isosurface network
// fills a band of about +-380 voxels around the terrain height: a workload where a large share of ALL blocks carries
// surface (the reference has no generator at all: surfaces are application callbacks, include/VoxelSurface.h)
VXT_HD void voxel(uint32_t x, uint32_t y, uint32_t z, float h, uint32_t seed, int8_t& dist, uint8_t& mat, uint8_t& blend, uint32_t style = 0)
{
	const float CAVE_AMP = style ? 6.0f : 5.0f;
	const float caveFreq = 1.0f / 24.0f;
	float d = (float)z - h;
	if (style) d = d * (1.0f / 64.0f) - CAVE_AMP * noise3((float)x * caveFreq, (float)y * caveFreq, (float)z * caveFreq, seed + 977u);
	else if (d > 4.0f + CAVE_AMP) d = 100.f;       // far above: clamped to +4 anyway
	else if (d < -(4.0f + CAVE_AMP)) d = -100.f;   // far below
	else d = d - CAVE_AMP * noise3((float)x * caveFreq, (float)y * caveFreq, (float)z * caveFreq, seed + 977u);
	if (d > 100.f) d = 100.f;
	if (d < -100.f) d = -100.f;
	dist = quantise(d);
	// three bands around the local terrain height, boundary dithered by +-4 voxels
	// (the hash is skipped where the band saturates anyway: |z - h| > 13)
	const float rel = (float)z - h;
	const float jitter = (rel > 13.0f || rel < -13.0f) ? 0.0f : 4.0f * lattice(x, y, z, seed + 4242u);
	const float band = ((float)z - h + jitter) / 6.0f + 1.5f;
	const float bc = band < 0.f ? 0.f : (band > 2.999f ? 2.999f : band);
	const int id = (int)bc;
	const float fr = bc - (float)id;
	mat = (uint8_t)id;
	blend = (uint8_t)(255.0f * fade(fr));
}

} // namespace vxt

// vx_synth.cpp — synthetic VoxelSurface stand-ins (include/voxels_synth.h).  Host code; feeds byte-identical
// inputs to the oracle and to the HIP path.  Not part of the polygonization hot path.
#include "../../include/voxels_synth.h"

#include <cmath>
#include <cstring>
#include <vector>

namespace {

inline uint32_t hash3(uint32_t x, uint32_t y, uint32_t z, uint32_t seed)
{
	uint32_t h = seed * 0x9E3779B1u + x * 0x85EBCA77u + y * 0xC2B2AE3Du + z * 0x27D4EB2Fu;
	h ^= h >> 15; h *= 0x2C1B3C6Du; h ^= h >> 12; h *= 0x297A2D39u; h ^= h >> 15;
	return h;
}

inline float lattice(uint32_t x, uint32_t y, uint32_t z, uint32_t seed)
{
	return (float)(hash3(x, y, z, seed) >> 8) * (2.0f / 16777216.0f) - 1.0f; // [-1, 1)
}

inline float fade(float t) { return t * t * (3.0f - 2.0f * t); }

// value noise, period-free, coordinates in lattice units
inline float noise2(float x, float y, uint32_t seed)
{
	const float fx = floorf(x), fy = floorf(y);
	const uint32_t ix = (uint32_t)(int)fx, iy = (uint32_t)(int)fy;
	const float u = fade(x - fx), v = fade(y - fy);
	const float a = lattice(ix, iy, 0, seed), b = lattice(ix + 1, iy, 0, seed);
	const float c = lattice(ix, iy + 1, 0, seed), d = lattice(ix + 1, iy + 1, 0, seed);
	const float ab = a + (b - a) * u, cd = c + (d - c) * u;
	return ab + (cd - ab) * v;
}

inline float noise3(float x, float y, float z, uint32_t seed)
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

// reference quantisation (src/VoxelGrid.cpp:37-50)
inline int8_t quantise(float value)
{
	float a = ceilf(fabsf(value));
	float b = a * (float)(value > 0 ? 1 : -1);
	if (b > 127.f) b = 127.f;
	int v = (int)b;
	return (int8_t)(v > 4 ? 4 : (v < -4 ? -4 : v));
}

const float CAVE_AMP = 5.0f;

} // namespace

extern "C" {

void vxs_terrain(uint32_t n, uint32_t z0, uint32_t z1, uint32_t seed, int8_t* dist, uint8_t* mat, uint8_t* blend)
{
	const float fn = (float)n;
	const float base = fn / 4.0f; // base wavelength in voxels
	std::vector<float> height((size_t)n * n);
	#pragma omp parallel for schedule(static)
	for (int64_t y = 0; y < (int64_t)n; ++y)
	for (uint32_t x = 0; x < n; ++x) {
		float amp = 1.0f, freq = 1.0f / base, sum = 0.0f, norm = 0.0f;
		for (int o = 0; o < 4; ++o) {
			sum += amp * noise2((float)x * freq, (float)y * freq, seed + 31u * (uint32_t)o);
			norm += amp; amp *= 0.5f; freq *= 2.0f;
		}
		height[(size_t)y * n + x] = fn * 0.5f + 0.25f * fn * (sum / norm);
	}
	const float caveFreq = 1.0f / 24.0f;
	#pragma omp parallel for schedule(dynamic, 1)
	for (int64_t zz = (int64_t)z0; zz < (int64_t)z1; ++zz) {
		const uint32_t z = (uint32_t)zz;
		const size_t plane = (size_t)(z - z0) * n * n;
		for (uint32_t y = 0; y < n; ++y)
		for (uint32_t x = 0; x < n; ++x) {
			const size_t i = plane + (size_t)y * n + x;
			const float h = height[(size_t)y * n + x];
			float d = (float)z - h;
			if (d > 4.0f + CAVE_AMP) d = 100.f;            // far above: clamped to +4 anyway
			else if (d < -(4.0f + CAVE_AMP)) d = -100.f;   // far below
			else d = d - CAVE_AMP * noise3((float)x * caveFreq, (float)y * caveFreq, (float)z * caveFreq, seed + 977u);
			if (d > 100.f) d = 100.f;
			if (d < -100.f) d = -100.f;
			dist[i] = quantise(d);
			if (mat || blend) {
				// three bands around the local terrain height, boundary dithered by +-4 voxels
				// (the hash is skipped where the band saturates anyway: |z - h| > 13)
				const float rel = (float)z - h;
				const float jitter = (rel > 13.0f || rel < -13.0f) ? 0.0f : 4.0f * lattice(x, y, z, seed + 4242u);
				const float band = ((float)z - h + jitter) / 6.0f + 1.5f;
				const float bc = band < 0.f ? 0.f : (band > 2.999f ? 2.999f : band);
				const int id = (int)bc;
				const float fr = bc - (float)id;
				if (mat) mat[i] = (uint8_t)id;
				if (blend) blend[i] = (uint8_t)(255.0f * fade(fr));
			}
		}
	}
}

void vxs_sphere(uint32_t n, uint32_t z0, uint32_t z1, float r_frac, int8_t* dist)
{
	const float c = (float)n / 2, r = r_frac * (float)n;
	#pragma omp parallel for schedule(static)
	for (int64_t zz = (int64_t)z0; zz < (int64_t)z1; ++zz) {
		const float dz = (float)zz - c;
		for (uint32_t y = 0; y < n; ++y)
		for (uint32_t x = 0; x < n; ++x) {
			const float dx = (float)x - c, dy = (float)y - c;
			float d = sqrtf((dx * dx + dy * dy) + dz * dz) - r;
			if (d > 100.f) d = 100.f;
			if (d < -100.f) d = -100.f;
			dist[((size_t)(zz - z0) * n + y) * n + x] = quantise(d);
		}
	}
}

void vxs_block_empty_flags(uint32_t n, uint32_t planes, const int8_t* dist, uint8_t* flags)
{
	const uint32_t nb = n / 16, nbz = planes / 16;
	#pragma omp parallel for schedule(static)
	for (int64_t id = 0; id < (int64_t)nb * nb * nbz; ++id) {
		const uint32_t bx = (uint32_t)(id % nb), by = (uint32_t)((id / nb) % nb), bz = (uint32_t)(id / ((int64_t)nb * nb));
		// walk the block in codec order (x, then y, then z), counting run boundaries like CompressBlock does
		const int8_t first = dist[((size_t)(bz * 16) * n + by * 16) * n + bx * 16];
		int8_t last = first;
		unsigned counter = 0, size = 1;
		bool empty = true, effective = true;
		for (uint32_t z = 0; z < 16 && effective; ++z)
		for (uint32_t y = 0; y < 16 && effective; ++y) {
			const int8_t* row = dist + ((size_t)(bz * 16 + z) * n + by * 16 + y) * n + bx * 16;
			for (uint32_t x = 0; x < 16; ++x) {
				const int8_t cur = row[x];
				if (last == cur && counter < 0xFF) { ++counter; continue; }
				size += 2; counter = 1; last = cur;
				if ((int)first * (int)last <= 0) empty = false;
				if (size > 4096) { effective = false; break; }
			}
		}
		flags[id] = (empty && effective) ? 1 : 0;
	}
}

} // extern "C"

// vx_synth.cpp — synthetic VoxelSurface stand-ins (include/voxels_synth.h).  Host code; feeds byte-identical
// inputs to the oracle and to the HIP path.  Not part of the polygonization hot path.
#include "../../include/voxels_synth.h"

#include <cmath>
#include <cstring>
#include <vector>

#include "vx_terrain_math.h"

namespace {

using vxt::quantise;

} // namespace

extern "C" {

void vxs_terrain(uint32_t n, uint32_t z0, uint32_t z1, uint32_t seed, int8_t* dist, uint8_t* mat, uint8_t* blend)
{
	vxs_terrain_ex(n, z0, z1, seed, 0, dist, mat, blend);
}

void vxs_terrain_ex(uint32_t n, uint32_t z0, uint32_t z1, uint32_t seed, uint32_t style, int8_t* dist, uint8_t* mat, uint8_t* blend)
{
	std::vector<float> height((size_t)n * n);
	#pragma omp parallel for schedule(static)
	for (int64_t y = 0; y < (int64_t)n; ++y)
	for (uint32_t x = 0; x < n; ++x) height[(size_t)y * n + x] = vxt::height(n, x, (uint32_t)y, seed);
	#pragma omp parallel for schedule(dynamic, 1)
	for (int64_t zz = (int64_t)z0; zz < (int64_t)z1; ++zz) {
		const uint32_t z = (uint32_t)zz;
		const size_t plane = (size_t)(z - z0) * n * n;
		for (uint32_t y = 0; y < n; ++y)
		for (uint32_t x = 0; x < n; ++x) {
			const size_t i = plane + (size_t)y * n + x;
			int8_t d; uint8_t m, b;
			vxt::voxel(x, y, z, height[(size_t)y * n + x], seed, d, m, b, style);
			dist[i] = d;
			if (mat) mat[i] = m;
			if (blend) blend[i] = b;
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

// tests/cpp/brick_layout_test.cpp — host-side check of the address arithmetic of the grid's mirrors (tv_core.h brick_local /
// brick_offset, tv_block.h pyramid_offset): the maps have to be bijections into the allocations vx_host.inl makes for
// them, a voxel row has to stay 16 contiguous bytes and a 128-byte line has to hold 8 rows (4 along y x 2 along z).
#include <cstdio>
#include <cstdlib>
#include <vector>

#include "../../voxels_amd/csrc/tv_block.h"

using namespace tv;

static int fail(const char* what) { fprintf(stderr, "FAIL: %s\n", what); return 1; }

int main()
{
	// 1. inside a brick
	std::vector<int> seen(BRICK_BYTES, 0);
	for (u32 z = 0; z < 16; ++z) for (u32 y = 0; y < 16; ++y) for (u32 x = 0; x < 16; ++x) {
		const u32 o = brick_local(x, y, z);
		if (o >= BRICK_BYTES || seen[o]++) return fail("brick_local is not a bijection");
		if (o != brick_local(0, y, z) + x) return fail("a voxel row is not contiguous");
		if ((o >> 7) != (brick_local(0, y & ~3u, z & ~1u) >> 7)) return fail("a 128-byte line is not a 16 x 4 x 2 tile");
	}
	// 2. whole fields: a y-slab of a 64^3 grid with one halo block row on either side
	GridView g;
	g.n = 64; g.bYb0 = 1; g.bZb0 = 0; g.bRowsY = 3; // resident block rows 1..3, all block planes
	const size_t bytes = (size_t)(g.n / 16) * g.bRowsY * (g.n / 16) * BRICK_BYTES;
	std::vector<unsigned char> hit(bytes, 0);
	for (int z = 0; z < 64; ++z) for (int y = 16; y < 64; ++y) for (int x = 0; x < 64; ++x) {
		const size_t o = brick_offset(g, x, y, z);
		if (o >= bytes || hit[o]++) return fail("brick_offset leaves the allocation or collides");
	}
	for (int bz = 0; bz < 4; ++bz) for (int by = 1; by < 4; ++by) for (int bx = 0; bx + 1 < 4; ++bx)
		if (brick_base(g, bx + 1, by, bz) != brick_base(g, bx, by, bz) + BRICK_BYTES) return fail("x-neighbour bricks do not follow each other");
	// 3. lattice copies: entries [0, extent >> L] per axis, sized like ensure_level_tables (vx_host.inl)
	for (int L = 1; L < PYRAMID_LEVELS; ++L) {
		const int n = 256, yBegin = 128, yEnd = 256, zBegin = 0, zEnd = 256;
		PyramidLevel P;
		P.data = nullptr;
		P.bricksX = ((n >> L) >> 4) + 1;
		P.bricksY = (((yEnd - yBegin) >> L) >> 4) + 1;
		P.yOrigin = yBegin >> L; P.zOrigin = zBegin >> L;
		const size_t bricksZ = (((zEnd - zBegin) >> L) >> 4) + 1;
		const size_t cap = (size_t)P.bricksX * P.bricksY * bricksZ * BRICK_BYTES;
		std::vector<unsigned char> used(cap, 0);
		for (int Z = zBegin >> L; Z <= zEnd >> L; ++Z) for (int Y = yBegin >> L; Y <= yEnd >> L; ++Y) for (int X = 0; X <= n >> L; ++X) {
			const size_t o = pyramid_offset(P, X, Y, Z);
			if (o >= cap || used[o]++) return fail("pyramid_offset leaves the allocation or collides");
		}
		// a block's sample row: 16 contiguous entries, the 17th in the same row of the next brick
		if (pyramid_offset(P, 16, P.yOrigin + 3, 5) != pyramid_offset(P, 0, P.yOrigin + 3, 5) + BRICK_BYTES) return fail("sample 16 of a lattice row is not in the next brick");
	}
	printf("OK\n");
	return 0;
}

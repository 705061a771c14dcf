// tests/cpp/dropin_test.cpp — application-level test written ONLY against the public API of the reference
// (Voxels.h).  The same source is compiled twice:
//   * against the reference's own headers and the unmodified reference library (oracle/_ref/dropin_ref), and
//   * against this repo's include/ and libVoxels.so (tests/cpp/dropin_ours, runs the HIP kernels),
// and both binaries dump the complete result (every level, block, vertex byte, index, statistic) to a file that
// tests/test_dropin_cpp.py compares.  Scenario: Grid::Create from a procedural VoxelSurface with materials,
// Polygonizer::Execute, Grid::InjectSurface (sphere carve), Execute with a Modification, PackForSave/Load round trip.
#include <cmath>
#include <cstddef>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <vector>

#include <Voxels.h> // the reference's Library.h uses size_t without including a header for it

using namespace Voxels;

static unsigned Hash(unsigned x, unsigned y, unsigned z)
{
	unsigned h = x * 0x85EBCA77u + y * 0xC2B2AE3Du + z * 0x27D4EB2Fu + 0x9E3779B1u;
	h ^= h >> 15; h *= 0x2C1B3C6Du; h ^= h >> 12; h *= 0x297A2D39u; h ^= h >> 15;
	return h;
}

// rolling hills + a floating ball, three materials by height with hashed boundaries
struct Hills : public VoxelSurface
{
	float n;
	float relief = 1.f;
	void GetSurface(float xStart, float xEnd, float xStep, float yStart, float yEnd, float yStep,
	                float zStart, float zEnd, float zStep, float* output, unsigned char* materialid, unsigned char* blend) override
	{
		size_t o = 0;
		for (float z = zStart; z < zEnd; z += zStep)
		for (float y = yStart; y < yEnd; y += yStep)
		for (float x = xStart; x < xEnd; x += xStep) {
			// (DROPIN_GENTLE: the same hills at a fifth of their height - no block with more than 640 non-trivial cells, the kind of
			// surface a multi-device Execute of libVoxels.so can split into a partial run of the primary + the helpers' levels)
			const float h = n * 0.5f + relief * (n * 0.12f * (sinf(x * 0.21f) + cosf(y * 0.17f)) + 0.04f * n * sinf((x + y) * 0.45f));
			float d = z - h;
			const float bx = x - n * 0.3f, by = y - n * 0.7f, bz = z - n * 0.8f;
			const float ball = sqrtf(bx * bx + by * by + bz * bz) - n * 0.1f;
			if (ball < d) d = ball;
			if (d > 100.f) d = 100.f;
			if (d < -100.f) d = -100.f;
			output[o] = d;
			if (materialid) materialid[o] = (unsigned char)(((unsigned)(z + (Hash((unsigned)x, (unsigned)y, (unsigned)z) & 7)) * 3) / (unsigned)(n + 8));
			if (blend) blend[o] = (unsigned char)(Hash((unsigned)z, (unsigned)x, (unsigned)y) & 0xFF);
			++o;
		}
	}
};

struct Ball : public VoxelSurface
{
	float r;
	void GetSurface(float xStart, float xEnd, float xStep, float yStart, float yEnd, float yStep,
	                float zStart, float zEnd, float zStep, float* output, unsigned char*, unsigned char*) override
	{
		size_t o = 0;
		for (float z = zStart; z < zEnd; z += zStep)
		for (float y = yStart; y < yEnd; y += yStep)
		for (float x = xStart; x < xEnd; x += xStep) output[o++] = sqrtf(x * x + y * y + z * z) - r;
	}
};

struct Mats : public MaterialMap
{
	mutable Material table[256];
	Mats()
	{
		for (int i = 0; i < 256; ++i)
			for (int k = 0; k < 3; ++k) { table[i].DiffuseIds0[k] = (unsigned char)(i * 6 + k); table[i].DiffuseIds1[k] = (unsigned char)(i * 6 + 3 + k); }
	}
	Material* GetMaterial(unsigned char id) const override { return id == 200 ? nullptr : &table[id]; }
};

static void Put(FILE* f, const void* p, size_t n) { fwrite(p, 1, n, f); }
static void PutU(FILE* f, unsigned v) { Put(f, &v, 4); }

static void Dump(FILE* f, const PolygonSurface* s)
{
	const float3 e = s->GetExtents();
	Put(f, &e, 12);
	PutU(f, s->GetLevelsCount());
	for (unsigned l = 0; l < s->GetLevelsCount(); ++l) {
		PutU(f, s->GetBlocksForLevelCount(l));
		for (unsigned i = 0; i < s->GetBlocksForLevelCount(l); ++i) {
			const BlockPolygons* b = s->GetBlockForLevel(l, i);
			PutU(f, b->GetId());
			const float3 mn = b->GetMinimalCorner(), mx = b->GetMaximalCorner();
			Put(f, &mn, 12); Put(f, &mx, 12);
			unsigned c = 0;
			const PolygonVertex* v = b->GetVertices(&c); PutU(f, c); Put(f, v, size_t(c) * sizeof(PolygonVertex));
			const unsigned* ix = b->GetIndices(&c); PutU(f, c); Put(f, ix, size_t(c) * 4);
			for (int face = 0; face < 6; ++face) {
				const PolygonVertex* tv = b->GetTransitionVertices((BlockPolygons::TransitionFaceId)face, &c); PutU(f, c); Put(f, tv, size_t(c) * sizeof(PolygonVertex));
				const unsigned* ti = b->GetTransitionIndices((BlockPolygons::TransitionFaceId)face, &c); PutU(f, c); Put(f, ti, size_t(c) * 4);
			}
		}
	}
	const PolygonizationStatistics* st = s->GetStatistics();
	Put(f, st, sizeof(*st));
	PutU(f, s->GetCacheSizeBytes());
	PutU(f, s->GetPolygonDataSizeBytes()); // (incl. the reference's quirk: the transition meshes count as 12 vector objects per block, src/TransVoxelImpl.cpp:222-235)
}

static void LogSink(LogSeverity sev, const char* msg) { if (sev >= LS_Error) fprintf(stderr, "[voxels] %s\n", msg); }

int main(int argc, char** argv)
{
	if (argc < 3) { fprintf(stderr, "usage: %s <grid edge> <out file> [B]\n", argv[0]); return 2; }
	// "B": the Modification is executed by a SECOND Polygonizer, after the first one has been reused for a full run on
	// another grid (the reference keeps a surface's caches in the PolygonSurface, so any Polygonizer can update it)
	const bool second = argc > 3 && argv[3][0] == 'B';
	const unsigned n = (unsigned)atoi(argv[1]);
	FILE* f = fopen(argv[2], "wb");
	if (!f) return 2;
	if (InitializeVoxels(VOXELS_VERSION, &LogSink, nullptr) != IE_Ok) return 3;
	PutU(f, GetBuildVersion());

	Hills hills; hills.n = (float)n;
	if (getenv("DROPIN_GENTLE")) hills.relief = 0.2f;
	Grid* grid = Grid::Create(n, n, n, 0.f, 0.f, 0.f, 1.f, &hills);
	if (!grid) return 4;
	PutU(f, grid->GetWidth()); PutU(f, grid->GetBlockExtent()); PutU(f, grid->GetGridBlocksMemorySize());

	Mats mats;
	Polygonizer poly;
	PolygonSurface* surface = poly.Execute(*grid, &mats);
	if (!surface) { fprintf(stderr, "Execute failed\n"); return 5; }
	Dump(f, surface);

	// sphere carve + incremental update
	Ball ball; ball.r = n * 0.11f;
	const float3pair box = grid->InjectSurface(float3(n * 0.5f, n * 0.45f, n * 0.52f), float3(n * 0.3f, n * 0.3f, n * 0.3f), &ball, IT_Subtract);
	Put(f, &box, sizeof(box));
	Modification* mod = Modification::Create();
	mod->Map = surface;
	mod->MinCornerModified = box.first;
	mod->MaxCornerModified = box.second;
	Polygonizer polyB;
	if (second) {
		Ball other; other.r = n * 0.2f;
		Grid* g2 = Grid::Create(n / 2, n / 2, n / 2, 0.f, 0.f, 0.f, 1.f, &other);
		if (!g2) return 4;
		PolygonSurface* s2 = poly.Execute(*g2, &mats);
		if (!s2) return 5;
		Dump(f, s2);
		s2->Destroy();
		g2->Destroy();
	}
	PolygonSurface* updated = (second ? polyB : poly).Execute(*grid, &mats, mod);
	if (updated != surface) { fprintf(stderr, "Modification did not update the surface in place\n"); return 6; }
	unsigned mc = 0;
	const unsigned* ids = mod->GetModifiedBlocks(&mc);
	PutU(f, mc); Put(f, ids, size_t(mc) * 4);
	Dump(f, surface);
	mod->Destroy();

	// DROPIN_EDITS=k: k further edits of the same surface - overlapping the first and each other, adding and carving in turn,
	// updated by either Polygonizer in turn (blocks rebuilt twice, blocks dropped again, lists that keep growing)
	const int further = getenv("DROPIN_EDITS") ? atoi(getenv("DROPIN_EDITS")) : 0;
	for (int e = 0; e < further; ++e) {
		Ball brush; brush.r = n * (0.06f + 0.02f * (float)(e % 3));
		const float3 at(n * (0.42f + 0.09f * (float)e), n * (0.47f - 0.03f * (float)(e % 2)), n * (0.55f - 0.06f * (float)e));
		const float3pair touched = grid->InjectSurface(at, float3(n * 0.2f, n * 0.2f, n * 0.2f), &brush, (e & 1) ? IT_Subtract : IT_Add);
		Put(f, &touched, sizeof(touched));
		Modification* m2 = Modification::Create();
		m2->Map = surface;
		m2->MinCornerModified = touched.first;
		m2->MaxCornerModified = touched.second;
		if (((e & 1) ? polyB : poly).Execute(*grid, &mats, m2) != surface) { fprintf(stderr, "Modification %d did not update the surface in place\n", e + 2); return 6; }
		unsigned c2 = 0;
		const unsigned* ids2 = m2->GetModifiedBlocks(&c2);
		PutU(f, c2); Put(f, ids2, size_t(c2) * 4);
		Dump(f, surface);
		m2->Destroy();
	}

	// material edit, block accessors, save/load round trip, fresh full polygonization of the loaded grid
	grid->InjectMaterial(float3(n * 0.4f, n * 0.5f, n * 0.5f), float3(n * 0.25f, n * 0.25f, n * 0.25f), 5, true);
	std::vector<char> dist(4096);
	std::vector<unsigned char> m(4096), b(4096);
	grid->GetBlockDistanceData(float3(1, 1, 1), dist.data());
	grid->GetBlockMaterialData(float3(1, 1, 1), m.data(), b.data());
	Put(f, dist.data(), 4096); Put(f, m.data(), 4096); Put(f, b.data(), 4096);
	Grid::PackedGrid* pack = grid->PackForSave();
	PutU(f, pack->GetSize()); Put(f, pack->GetData(), pack->GetSize());
	Grid* loaded = Grid::Load(pack->GetData(), pack->GetSize());
	pack->Destroy();
	PolygonSurface* again = poly.Execute(*loaded, &mats);
	if (!again) return 7;
	Dump(f, again);

	again->Destroy();
	surface->Destroy();
	loaded->Destroy();
	grid->Destroy();
	DeinitializeVoxels();
	fclose(f);
	return 0;
}

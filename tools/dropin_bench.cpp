// tools/dropin_bench.cpp — end-to-end time of Voxels::Polygonizer::Execute through the drop-in C++ API (libVoxels.so):
// grid upload (the packed file travels and is expanded on the device), all kernels, block lists, and the download of
// every level into PolygonBlock vectors — what an application that links against Voxels.h waits for.
// Usage: dropin_bench <grid file written by Grid::PackForSave / vx_grid_pack> [runs]
#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <vector>

#include <Voxels.h>

using namespace Voxels;

struct Mats : public MaterialMap
{
	mutable Material table[256];
	Mats()
	{
		for (int i = 0; i < 256; ++i)
			for (int k = 0; k < 3; ++k) { table[i].DiffuseIds0[k] = (unsigned char)((i * 6 + k) % 251); table[i].DiffuseIds1[k] = (unsigned char)((i * 6 + 3 + k) % 251); }
	}
	Material* GetMaterial(unsigned char id) const override { return &table[id]; }
};

static void Quiet(LogSeverity, const char*) {}

int main(int argc, char** argv)
{
	if (argc < 2) { fprintf(stderr, "usage: %s grid.bin [runs]\n", argv[0]); return 2; }
	const int runs = argc > 2 ? atoi(argv[2]) : 3;
	FILE* f = fopen(argv[1], "rb");
	if (!f) { perror(argv[1]); return 2; }
	fseek(f, 0, SEEK_END);
	const long size = ftell(f);
	fseek(f, 0, SEEK_SET);
	std::vector<char> blob((size_t)size);
	if (fread(blob.data(), 1, (size_t)size, f) != (size_t)size) { fclose(f); return 2; }
	fclose(f);
	const auto i0 = std::chrono::steady_clock::now();
	if (InitializeVoxels(VOXELS_VERSION, &Quiet, nullptr) != IE_Ok) return 3;
	const double initMs = std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - i0).count(); // (one-time costs: device context, optional arena)
	Mats mats;
	Polygonizer poly;
	double best = 1e30, first = 0;
	unsigned long long verts = 0, indices = 0;
	unsigned levels = 0;
	for (int r = 0; r < runs; ++r) {
		// a fresh Grid per run: Execute then includes bringing the grid to the device (as its packed file)
		Grid* g = Grid::Load(blob.data(), (unsigned)size);
		if (!g) return 4;
		const auto t0 = std::chrono::steady_clock::now();
		PolygonSurface* s = poly.Execute(*g, &mats);
		const auto t1 = std::chrono::steady_clock::now();
		if (!s) return 5;
		const double ms = std::chrono::duration<double, std::milli>(t1 - t0).count();
		if (r == 0) first = ms;
		if (ms < best) best = ms;
		levels = s->GetLevelsCount();
		verts = indices = 0;
		for (unsigned l = 0; l < levels; ++l)
			for (unsigned b = 0; b < s->GetBlocksForLevelCount(l); ++b) {
				const BlockPolygons* bp = s->GetBlockForLevel(l, b);
				unsigned c = 0;
				bp->GetVertices(&c); verts += c;
				bp->GetIndices(&c); indices += c;
			}
		s->Destroy();
		g->Destroy();
	}
	printf("{\"execute_ms_best\": %.3f, \"execute_ms_first\": %.3f, \"initialize_ms\": %.3f, \"prewarm_mb\": %d, \"runs\": %d, \"levels\": %u, \"verts\": %llu, \"indices\": %llu, \"grid_file_bytes\": %ld}\n",
	       best, first, initMs, getenv("VOXELS_PREWARM_MB") ? atoi(getenv("VOXELS_PREWARM_MB")) : 0, runs, levels, verts, indices, size);
	DeinitializeVoxels();
	return 0;
}

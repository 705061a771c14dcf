// oracle/ref_harness.cpp — drives the UNMODIFIED reference library (compiled from
// /root/reference/src/{TransVoxelImpl,VoxelGrid,Voxels}.cpp where they lie, see oracle/Makefile)
// through its public API only (include/Voxels.h of the reference) and exposes the results through
// oracle/vxo_api.h.  TEST INFRASTRUCTURE ONLY — never linked into the product.
//
// Reference entry points exercised (all /root/reference):
//   InitializeVoxels                     include/Library.h:57, src/Voxels.cpp:35-60
//   Grid::Create / Load / PackForSave    include/Grid.h:57-85, src/VoxelGrid.cpp:759-806
//   Grid::Modify*Data / Get*Data         include/Grid.h:127-145, src/VoxelGrid.cpp:885-909
//   Grid::InjectSurface / InjectMaterial include/Grid.h:103-117, src/VoxelGrid.cpp:854-878
//   Polygonizer::Execute                 include/Polygonizer.h:230, src/TransVoxelImpl.cpp:74-79
//   PolygonSurface / BlockPolygons       include/Polygonizer.h:52-178
#include "vxo_api.h"

#include <cmath>
#include <cstring>
#include <vector>
#include <memory>
#include <utility>
#include <atomic>
#include <omp.h>
#include <glm/glm.hpp>

#include <Voxels.h>          // reference public API (-I/root/reference/include)
#include <VoxelGrid.h>       // IsBlockEmpty only (-I/root/reference/src); needs glm + std first

using namespace Voxels;

static_assert(sizeof(PolygonVertex) == 48, "PolygonVertex must be 48 bytes");
static_assert(sizeof(vxo_vertex) == 48, "vxo_vertex must be 48 bytes");

struct vxo_grid { Grid* g; uint32_t n; };
struct vxo_surface { PolygonSurface* s; };

static std::atomic<uint32_t> g_logErrors(0);

static void LogSink(LogSeverity sev, const char*)
{
	if (sev >= LS_Error) ++g_logErrors;
}

static void EnsureInit()
{
	static bool done = false;
	if (!done) {
		InitializeVoxels(VOXELS_VERSION, &LogSink, nullptr);
		done = true;
	}
}

namespace {

// VoxelSurface backed by dense arrays; the grid asks for boxes [start,end) with the given step.
struct ArraySurface : public VoxelSurface
{
	uint32_t n;
	const float* values;
	const uint8_t* mat;
	const uint8_t* blend;

	void GetSurface(float xStart, float xEnd, float xStep,
		float yStart, float yEnd, float yStep,
		float zStart, float zEnd, float zStep,
		float* output, unsigned char* materialid, unsigned char* blendOut) override
	{
		size_t o = 0;
		for (float z = zStart; z < zEnd; z += zStep)
		for (float y = yStart; y < yEnd; y += yStep)
		for (float x = xStart; x < xEnd; x += xStep) {
			const size_t id = (size_t(z) * n + size_t(y)) * n + size_t(x);
			output[o] = values[id];
			if (materialid) materialid[o] = mat ? mat[id] : 0;
			if (blendOut) blendOut[o] = blend ? blend[id] : 0;
			++o;
		}
	}
};

// Ball brush: d = |p| - r, p in the coordinates the grid passes (relative to the inject position).
struct BallSurface : public VoxelSurface
{
	float r;
	void GetSurface(float xStart, float xEnd, float xStep,
		float yStart, float yEnd, float yStep,
		float zStart, float zEnd, float zStep,
		float* output, unsigned char* materialid, unsigned char* blendOut) override
	{
		size_t o = 0;
		for (float z = zStart; z < zEnd; z += zStep)
		for (float y = yStart; y < yEnd; y += yStep)
		for (float x = xStart; x < xEnd; x += xStep) {
			output[o] = sqrtf(x * x + y * y + z * z) - r;
			if (materialid) materialid[o] = 0;
			if (blendOut) blendOut[o] = 0;
			++o;
		}
	}
};

struct LutMaterials : public MaterialMap
{
	mutable Material table[256];
	uint8_t valid[256];
	Material* GetMaterial(unsigned char id) const override
	{
		return valid[id] ? &table[id] : nullptr;
	}
};

void FillLut(LutMaterials& m, const uint8_t* lut, const uint8_t* valid)
{
	for (int i = 0; i < 256; ++i) {
		for (int k = 0; k < 3; ++k) {
			m.table[i].DiffuseIds0[k] = lut[i * 6 + k];
			m.table[i].DiffuseIds1[k] = lut[i * 6 + 3 + k];
		}
		m.valid[i] = valid ? valid[i] : 1;
	}
}

} // namespace

extern "C" {

const char* vxo_kind(void) { return "reference"; }

vxo_grid* vxo_grid_from_dense(uint32_t n, const int8_t* dist, const uint8_t* mat, const uint8_t* blend)
{
	EnsureInit();
	Grid* g = Grid::Create(n, n, n);
	const uint32_t nb = n / 16;
	std::vector<char> d(4096);
	std::vector<unsigned char> m(4096), b(4096);
	for (uint32_t bz = 0; bz < nb; ++bz)
	for (uint32_t by = 0; by < nb; ++by)
	for (uint32_t bx = 0; bx < nb; ++bx) {
		for (uint32_t z = 0; z < 16; ++z)
		for (uint32_t y = 0; y < 16; ++y) {
			const size_t src = (size_t(bz * 16 + z) * n + (by * 16 + y)) * n + bx * 16;
			const size_t dst = z * 256 + y * 16;
			memcpy(&d[dst], dist + src, 16);
			if (mat) memcpy(&m[dst], mat + src, 16); else memset(&m[dst], 0, 16);
			if (blend) memcpy(&b[dst], blend + src, 16); else memset(&b[dst], 0, 16);
		}
		const float3 c((float)bx, (float)by, (float)bz);
		g->ModifyBlockDistanceData(c, d.data());
		g->ModifyBlockMaterialData(c, m.data(), b.data());
	}
	return new vxo_grid{ g, n };
}

vxo_grid* vxo_grid_from_float(uint32_t n, const float* values, const uint8_t* mat, const uint8_t* blend)
{
	EnsureInit();
	ArraySurface surf;
	surf.n = n; surf.values = values; surf.mat = mat; surf.blend = blend;
	Grid* g = Grid::Create(n, n, n, 0.f, 0.f, 0.f, 1.f, &surf);
	return new vxo_grid{ g, n };
}

vxo_grid* vxo_grid_from_heightmap(uint32_t n, const char* heightmap)
{
	EnsureInit();
	Grid* g = Grid::Create(n, heightmap);
	return g ? new vxo_grid{ g, n } : nullptr;
}

void vxo_grid_destroy(vxo_grid* g)
{
	if (!g) return;
	g->g->Destroy();
	delete g;
}

uint32_t vxo_grid_size(const vxo_grid* g) { return g->n; }

void vxo_grid_read_dense(const vxo_grid* g, int8_t* dist, uint8_t* mat, uint8_t* blend)
{
	const uint32_t n = g->n, nb = n / 16;
	std::vector<char> d(4096);
	std::vector<unsigned char> m(4096), b(4096);
	for (uint32_t bz = 0; bz < nb; ++bz)
	for (uint32_t by = 0; by < nb; ++by)
	for (uint32_t bx = 0; bx < nb; ++bx) {
		const float3 c((float)bx, (float)by, (float)bz);
		g->g->GetBlockDistanceData(c, d.data());
		g->g->GetBlockMaterialData(c, m.data(), b.data());
		for (uint32_t z = 0; z < 16; ++z)
		for (uint32_t y = 0; y < 16; ++y) {
			const size_t dst = (size_t(bz * 16 + z) * n + (by * 16 + y)) * n + bx * 16;
			const size_t src = z * 256 + y * 16;
			if (dist) memcpy(dist + dst, &d[src], 16);
			if (mat) memcpy(mat + dst, &m[src], 16);
			if (blend) memcpy(blend + dst, &b[src], 16);
		}
	}
}

void vxo_grid_block_flags(const vxo_grid* g, uint8_t* out)
{
	const uint32_t nb = g->n / 16;
	const VoxelGrid* vg = g->g->GetInternalRepresentation();
	size_t o = 0;
	for (uint32_t bz = 0; bz < nb; ++bz)
	for (uint32_t by = 0; by < nb; ++by)
	for (uint32_t bx = 0; bx < nb; ++bx)
		out[o++] = vg->IsBlockEmpty(glm::vec3(bx, by, bz)) ? 1 : 0;
}

uint32_t vxo_grid_memory_size(vxo_grid* g) { return g->g->GetGridBlocksMemorySize(); }

void vxo_grid_inject_ball(vxo_grid* g, const float pos[3], const float ext[3], float radius, int type,
	float out_min[3], float out_max[3])
{
	BallSurface ball;
	ball.r = radius;
	float3pair r = g->g->InjectSurface(float3(pos[0], pos[1], pos[2]), float3(ext[0], ext[1], ext[2]),
		&ball, (InjectionType)type);
	out_min[0] = r.first.x; out_min[1] = r.first.y; out_min[2] = r.first.z;
	out_max[0] = r.second.x; out_max[1] = r.second.y; out_max[2] = r.second.z;
}

void vxo_grid_inject_material(vxo_grid* g, const float pos[3], const float ext[3], uint8_t material,
	int add, float out_min[3], float out_max[3])
{
	float3pair r = g->g->InjectMaterial(float3(pos[0], pos[1], pos[2]), float3(ext[0], ext[1], ext[2]),
		material, add != 0);
	out_min[0] = r.first.x; out_min[1] = r.first.y; out_min[2] = r.first.z;
	out_max[0] = r.second.x; out_max[1] = r.second.y; out_max[2] = r.second.z;
}

size_t vxo_grid_pack(const vxo_grid* g, char* out, size_t cap)
{
	Grid::PackedGrid* p = g->g->PackForSave();
	const size_t sz = p->GetSize();
	if (out) memcpy(out, p->GetData(), sz < cap ? sz : cap);
	p->Destroy();
	return sz;
}

vxo_grid* vxo_grid_load(const char* blob, size_t size)
{
	EnsureInit();
	Grid* g = Grid::Load(blob, (unsigned)size);
	if (!g) return nullptr;
	return new vxo_grid{ g, g->GetWidth() };
}

vxo_surface* vxo_execute(const vxo_grid* g, const uint8_t* lut, const uint8_t* valid, int threads)
{
	EnsureInit();
	if (threads > 0) omp_set_num_threads(threads);
	LutMaterials mats;
	FillLut(mats, lut, valid);
	Polygonizer poly;
	PolygonSurface* s = poly.Execute(*g->g, &mats, nullptr);
	return s ? new vxo_surface{ s } : nullptr;
}

uint32_t vxo_execute_modify(const vxo_grid* g, const uint8_t* lut, const uint8_t* valid, int threads,
	vxo_surface* prev, const float min_corner[3], const float max_corner[3],
	uint32_t* modified_ids, uint32_t cap)
{
	EnsureInit();
	if (threads > 0) omp_set_num_threads(threads);
	LutMaterials mats;
	FillLut(mats, lut, valid);
	Modification* mod = Modification::Create();
	mod->Map = prev->s;
	mod->MinCornerModified = float3(min_corner[0], min_corner[1], min_corner[2]);
	mod->MaxCornerModified = float3(max_corner[0], max_corner[1], max_corner[2]);
	Polygonizer poly;
	PolygonSurface* s = poly.Execute(*g->g, &mats, mod);
	prev->s = s;
	unsigned count = 0;
	const unsigned* ids = mod->GetModifiedBlocks(&count);
	for (unsigned i = 0; i < count && i < cap; ++i) modified_ids[i] = ids[i];
	mod->Destroy();
	return count;
}

void vxo_surface_destroy(vxo_surface* s)
{
	if (!s) return;
	s->s->Destroy();
	delete s;
}

uint32_t vxo_surface_levels(const vxo_surface* s) { return s->s->GetLevelsCount(); }

void vxo_surface_extents(const vxo_surface* s, float out[3])
{
	const float3 e = s->s->GetExtents();
	out[0] = e.x; out[1] = e.y; out[2] = e.z;
}

uint32_t vxo_surface_blocks(const vxo_surface* s, uint32_t level) { return s->s->GetBlocksForLevelCount(level); }

void vxo_surface_level_totals(const vxo_surface* s, uint32_t level, uint64_t totals[4])
{
	totals[0] = totals[1] = totals[2] = totals[3] = 0;
	const unsigned nb = s->s->GetBlocksForLevelCount(level);
	for (unsigned i = 0; i < nb; ++i) {
		const BlockPolygons* b = s->s->GetBlockForLevel(level, i);
		unsigned c = 0;
		b->GetVertices(&c); totals[0] += c;
		b->GetIndices(&c); totals[1] += c;
		for (int f = 0; f < 6; ++f) {
			b->GetTransitionVertices((BlockPolygons::TransitionFaceId)f, &c); totals[2] += c;
			b->GetTransitionIndices((BlockPolygons::TransitionFaceId)f, &c); totals[3] += c;
		}
	}
}

void vxo_surface_dump_level(const vxo_surface* s, uint32_t level, vxo_block_info* infos,
	vxo_vertex* verts, uint32_t* idx, vxo_vertex* tverts, uint32_t* tidx)
{
	const unsigned nb = s->s->GetBlocksForLevelCount(level);
	size_t ov = 0, oi = 0, otv = 0, oti = 0;
	for (unsigned i = 0; i < nb; ++i) {
		const BlockPolygons* b = s->s->GetBlockForLevel(level, i);
		vxo_block_info& info = infos[i];
		info.id = b->GetId();
		unsigned c = 0;
		const PolygonVertex* v = b->GetVertices(&c);
		info.n_verts = c;
		if (c && verts) memcpy(verts + ov, v, size_t(c) * 48);
		ov += c;
		const unsigned* ix = b->GetIndices(&c);
		info.n_idx = c;
		if (c && idx) memcpy(idx + oi, ix, size_t(c) * 4);
		oi += c;
		for (int f = 0; f < 6; ++f) {
			const PolygonVertex* tv = b->GetTransitionVertices((BlockPolygons::TransitionFaceId)f, &c);
			info.n_tverts[f] = c;
			if (c && tverts) memcpy(tverts + otv, tv, size_t(c) * 48);
			otv += c;
			const unsigned* ti = b->GetTransitionIndices((BlockPolygons::TransitionFaceId)f, &c);
			info.n_tidx[f] = c;
			if (c && tidx) memcpy(tidx + oti, ti, size_t(c) * 4);
			oti += c;
		}
		const float3 mn = b->GetMinimalCorner(), mx = b->GetMaximalCorner();
		info.min_corner[0] = mn.x; info.min_corner[1] = mn.y; info.min_corner[2] = mn.z;
		info.max_corner[0] = mx.x; info.max_corner[1] = mx.y; info.max_corner[2] = mx.z;
	}
}

void vxo_surface_stats(const vxo_surface* s, uint32_t stats[20])
{
	const PolygonizationStatistics* st = s->s->GetStatistics();
	stats[0] = st->BlocksCalculated;
	stats[1] = st->TrivialCells;
	stats[2] = st->NonTrivialCells;
	stats[3] = st->DegenerateTrianglesRemoved;
	for (int i = 0; i < 16; ++i) stats[4 + i] = st->PerCaseCellsCount[i];
}

uint32_t vxo_surface_cache_bytes(const vxo_surface* s) { return s->s->GetCacheSizeBytes(); }
uint32_t vxo_surface_polygon_bytes(const vxo_surface* s) { return s->s->GetPolygonDataSizeBytes(); }
uint32_t vxo_log_errors(void) { return g_logErrors.load(); }

} // extern "C"

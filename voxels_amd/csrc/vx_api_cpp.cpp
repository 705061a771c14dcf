// vx_api_cpp.cpp — libVoxels.so: the reference's public C++ API (include/Voxels.h) implemented on the host and
// backed by the HIP kernels of libvoxels_hip.so through the C ABI (include/voxels_hip.h).
//
// Replaces (by interface, not by code):
//   Grid::* forwarding            reference src/VoxelGrid.cpp:755-919      -> Voxels::VoxelGrid (vx_grid_host.cpp)
//   Polygonizer::Execute          reference src/TransVoxelImpl.cpp:74-79   -> vx_grid_upload/update + vx_polygonize[_dirty]
//   PolygonMap / PolygonBlock     reference src/TransVoxelImpl.h:33-142    -> SurfaceImpl / BlockImpl below (host copies of
//                                                                             the meshes the kernels wrote to the device pools)
//   Initialize/DeinitializeVoxels reference src/Voxels.cpp:35-76
// There is no CPU polygonizer here: without a HIP device Execute logs an error and returns nullptr.
#include "../../include/Voxels.h"
#include "../../include/voxels_hip.h"
#include "vx_grid_host.h"

#include <algorithm>
#include <chrono>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <memory>
#include <mutex>
#include <thread>
#include <vector>

namespace Voxels
{

static_assert(sizeof(PolygonVertex) == 48, "PolygonVertex must stay 48 bytes (include/Polygonizer.h of the reference)");
static_assert(sizeof(PolygonVertex) == sizeof(vx_vertex), "C ABI vertex layout");

namespace
{
LogMessage g_Logger = nullptr;
bool g_Initialized = false;

void Log(LogSeverity s, const char* msg)
{
	if (g_Logger) g_Logger(s, msg);
}
} // namespace

const unsigned PolygonSurface::INVALID_ID = 0xFFFFFFFF;

// ------------------------------------------------------------------------------------------------ Grid
Grid::Grid(VoxelGrid* impl) : m_InternalGrid(impl) {}
Grid::~Grid() { delete m_InternalGrid; }

static bool CubeOfBlocks(unsigned w, unsigned d, unsigned h)
{
	// like the reference, only cubes work (VoxelGrid.cpp:103, TransVoxelImpl.cpp:490); unlike it we say so
	if (w == d && d == h && w >= 16 && (w % 16) == 0) return true;
	Log(LS_Error, "Grid: only cubic grids with an edge that is a multiple of 16 are supported");
	return false;
}

Grid* Grid::Create(unsigned w, unsigned d, unsigned h, float startX, float startY, float startZ, float step, VoxelSurface* surface)
{
	if (!CubeOfBlocks(w, d, h) || !surface) return nullptr;
	return new Grid(VoxelGrid::FromSurface(w, startX, startY, startZ, step, surface));
}

Grid* Grid::Create(unsigned w, unsigned d, unsigned h)
{
	if (!CubeOfBlocks(w, d, h)) return nullptr;
	return new Grid(new VoxelGrid(w));
}

Grid* Grid::Create(unsigned w, const char* heightmap)
{
	if (!CubeOfBlocks(w, w, w) || !heightmap) return nullptr;
	return new Grid(VoxelGrid::FromHeightmap(w, heightmap));
}

Grid* Grid::Load(const char* blob, unsigned size)
{
	VoxelGrid* g = blob ? VoxelGrid::Load(blob, size) : nullptr;
	if (!g) { Log(LS_Error, "Voxel grid file version not supported!"); return nullptr; }
	return new Grid(g);
}

void Grid::Destroy() { delete this; }

namespace
{
struct PackedGridImpl : public Grid::PackedGrid
{
	std::vector<char> Data;
	void Destroy() override { delete this; }
	unsigned GetSize() const override { return (unsigned)Data.size(); }
	const char* GetData() const override { return Data.data(); }
};
} // namespace

Grid::PackedGrid* Grid::PackForSave() const
{
	PackedGridImpl* p = new PackedGridImpl;
	m_InternalGrid->Pack(p->Data);
	return p;
}

unsigned Grid::GetWidth() const { return m_InternalGrid->Size(); }
unsigned Grid::GetDepth() const { return m_InternalGrid->Size(); }
unsigned Grid::GetHeight() const { return m_InternalGrid->Size(); }
unsigned Grid::GetBlockExtent() const { return VoxelGrid::BLOCK; }
unsigned Grid::GetGridBlocksMemorySize() { return (unsigned)m_InternalGrid->MemoryForBlocks(); }
VoxelGrid* Grid::GetInternalRepresentation() const { return m_InternalGrid; }

float3pair Grid::InjectSurface(const float3& position, const float3& extents, VoxelSurface* surface, InjectionType type)
{
	const float pos[3] = { position.x, position.y, position.z }, ext[3] = { extents.x, extents.y, extents.z };
	float mn[3], mx[3];
	m_InternalGrid->InjectSurface(pos, ext, surface, (int)type, mn, mx);
	float3pair r;
	r.first = float3(mn[0], mn[1], mn[2]);
	r.second = float3(mx[0], mx[1], mx[2]);
	return r;
}

float3pair Grid::InjectMaterial(const float3& position, const float3& extents, MaterialId material, bool addSubtractBlend)
{
	const float pos[3] = { position.x, position.y, position.z }, ext[3] = { extents.x, extents.y, extents.z };
	float mn[3], mx[3];
	m_InternalGrid->InjectMaterial(pos, ext, material, addSubtractBlend, mn, mx);
	float3pair r;
	r.first = float3(mn[0], mn[1], mn[2]);
	r.second = float3(mx[0], mx[1], mx[2]);
	return r;
}

bool Grid::GetBlockDistanceData(const float3& c, char* output) const
{
	m_InternalGrid->GetBlock((unsigned)c.x, (unsigned)c.y, (unsigned)c.z, (int8_t*)output, nullptr, nullptr);
	return true;
}

void Grid::ModifyBlockDistanceData(const float3& c, const char* distances)
{
	m_InternalGrid->SetBlockDistances((unsigned)c.x, (unsigned)c.y, (unsigned)c.z, (const int8_t*)distances);
}

bool Grid::GetBlockMaterialData(const float3& c, MaterialId* materials, BlendFactor* blends) const
{
	m_InternalGrid->GetBlock((unsigned)c.x, (unsigned)c.y, (unsigned)c.z, nullptr, materials, blends);
	return true;
}

void Grid::ModifyBlockMaterialData(const float3& c, const MaterialId* materials, const BlendFactor* blends)
{
	m_InternalGrid->SetBlockMaterials((unsigned)c.x, (unsigned)c.y, (unsigned)c.z, materials, blends);
}

// ------------------------------------------------------------------------------------------------ surface
struct DeviceState; // below: the device context a surface shares with the Polygonizer that made it

namespace
{

// A block's seven meshes are views into the surface's page-locked copy of the two pools (vx_host_meshes_acquire): no
// per-block arrays are allocated or filled.
struct BlockImpl : public BlockPolygons
{
	unsigned Id = 0;
	float3 MinCorner, MaxCorner;
	const PolygonVertex* Vertices = nullptr;
	const unsigned* Indices = nullptr;
	unsigned VertexCount = 0, IndexCount = 0;
	const PolygonVertex* TVertices[6] = { nullptr, nullptr, nullptr, nullptr, nullptr, nullptr };
	const unsigned* TIndices[6] = { nullptr, nullptr, nullptr, nullptr, nullptr, nullptr };
	unsigned TVertexCount[6] = { 0, 0, 0, 0, 0, 0 }, TIndexCount[6] = { 0, 0, 0, 0, 0, 0 };

	unsigned GetId() const override { return Id; }
	const PolygonVertex* GetVertices(unsigned* count) const override
	{
		if (count) *count = VertexCount;
		return VertexCount ? Vertices : nullptr;
	}
	const unsigned* GetIndices(unsigned* count) const override
	{
		if (count) *count = IndexCount;
		return IndexCount ? Indices : nullptr;
	}
	const PolygonVertex* GetTransitionVertices(TransitionFaceId face, unsigned* count) const override
	{
		if (count) *count = TVertexCount[face];
		return TVertexCount[face] ? TVertices[face] : nullptr;
	}
	const unsigned* GetTransitionIndices(TransitionFaceId face, unsigned* count) const override
	{
		if (count) *count = TIndexCount[face];
		return TIndexCount[face] ? TIndices[face] : nullptr;
	}
	float3 GetMinimalCorner() const override { return MinCorner; }
	float3 GetMaximalCorner() const override { return MaxCorner; }
};

struct SurfaceImpl : public PolygonSurface
{
	float3 Extents;
	std::vector<std::vector<BlockImpl> > Levels;
	vx_host_meshes Meshes = { nullptr, nullptr, 0, 0, nullptr }; // owned: released with the surface
	// a surface made by several devices (VOXELS_DEVICES): one page-locked copy per helper device for the finer levels, plain
	// arrays for the few blocks of the levels coarser than a slab (from the primary device)
	std::vector<vx_host_meshes> ShardMeshes;
	struct CoarseLevel { std::vector<PolygonVertex> V, TV; std::vector<unsigned> I, TI; };
	std::vector<CoarseLevel> Coarse;
	// Levels [0, HelperLevels) of such a surface were meshed by the helpers ONLY (the primary ran vx_polygonize_from: caches and
	// bitmaps of every level, meshes from HelperLevels up).  HelperBlocks[l] = the helper-made blocks of level l that no
	// Modification has replaced yet (views into ShardMeshes, which then stay with the surface); a Modification drops the ones
	// inside its box and appends what the primary rebuilt.
	unsigned HelperLevels = 0;
	std::vector<std::vector<BlockImpl> > HelperBlocks;
	PolygonizationStatistics Stats;
	unsigned GridSize = 0;
	std::shared_ptr<DeviceState> Device; // the context whose device caches describe this surface: kept alive by the surface, usable by any Polygonizer (Modification)

	void ReleaseShardCopies()
	{
		for (vx_host_meshes& m : ShardMeshes) if (m.arena) vx_host_meshes_release(m.arena);
		ShardMeshes.clear();
		Coarse.clear();
	}
	~SurfaceImpl() { if (Meshes.arena) vx_host_meshes_release(Meshes.arena); ReleaseShardCopies(); }
	float3 GetExtents() const override { return Extents; }
	unsigned GetLevelsCount() const override { return (unsigned)Levels.size(); }
	unsigned GetBlocksForLevelCount(unsigned level) const override { return (unsigned)Levels[level].size(); }
	const BlockPolygons* GetBlockForLevel(unsigned level, unsigned id) const override
	{
		if (id >= Levels[level].size()) return nullptr;
		return &Levels[level][id];
	}
	const PolygonizationStatistics* GetStatistics() const override { return &Stats; }
	unsigned GetCacheSizeBytes() const override
	{
		// what the reference's caches would hold (TransVoxelImpl.cpp:196-220): one bit per level-0 cell plus two
		// bytes per cell of every coarser level
		size_t total = 0;
		unsigned blocks = GridSize / 16;
		total += (size_t(blocks) * blocks * blocks * 4096) >> 3;
		for (unsigned l = 1; l < Levels.size(); ++l) {
			const unsigned b = (GridSize / 16) >> l;
			total += size_t(b) * b * b * 4096 * 2;
		}
		return (unsigned)total;
	}
	unsigned GetPolygonDataSizeBytes() const override
	{
		size_t r = 0;
		for (const auto& lvl : Levels)
			for (const auto& b : lvl) r += (size_t)b.VertexCount * sizeof(PolygonVertex) + (size_t)b.IndexCount * sizeof(unsigned) + 12 * sizeof(std::vector<unsigned>);
		return (unsigned)r;
	}
	void Destroy() override { delete this; }
};

struct ModificationImpl : public Modification
{
	std::vector<unsigned> ModifiedBlocks;
	const unsigned* GetModifiedBlocks(unsigned* count) const override
	{
		if (count) *count = (unsigned)ModifiedBlocks.size();
		return ModifiedBlocks.empty() ? nullptr : ModifiedBlocks.data();
	}
	void Destroy() override { delete this; }
};

// Brings the surface's host copy of the pools up to date (one DMA into page-locked memory; after an incremental run only
// what the run appended) and rebuilds the per-block views of all levels.  The views are built aside and swapped in only
// when everything succeeded; on a failure the surface keeps NO views (vx_host_meshes_acquire may already have exchanged
// the arena the old views pointed into for a larger one: a stale view would be a use-after-free waiting to happen).
bool FetchSurface(vx_ctx* ctx, unsigned levels, SurfaceImpl& s)
{
	std::vector<std::vector<BlockImpl>> built(levels);
	const auto fail = [&]() { s.Levels.clear(); s.Levels.resize(levels); return false; };
	if (vx_host_meshes_acquire(ctx, &s.Meshes) != VX_OK) return fail();
	const PolygonVertex* pv = (const PolygonVertex*)s.Meshes.verts;
	const unsigned* pi = s.Meshes.indices;
	std::vector<vx_block_info> infos;
	std::vector<vx_block_ranges> ranges;
	for (unsigned level = 0; level < levels; ++level) {
		uint32_t nb = 0;
		if (vx_level_counts(ctx, level, &nb, nullptr) != VX_OK) return fail();
		infos.resize(nb);
		ranges.resize(nb);
		if (nb && (vx_download_level(ctx, level, infos.data(), nullptr, nullptr, nullptr, nullptr) != VX_OK
		           || vx_level_ranges(ctx, level, ranges.data()) != VX_OK)) return fail();
		std::vector<BlockImpl>& out = built[level];
		out.resize(nb);
		for (uint32_t k = 0; k < nb; ++k) {
			const vx_block_info& in = infos[k];
			const vx_block_ranges& r = ranges[k];
			BlockImpl& b = out[k];
			b.Id = in.id;
			b.MinCorner = float3(in.min_corner[0], in.min_corner[1], in.min_corner[2]);
			b.MaxCorner = float3(in.max_corner[0], in.max_corner[1], in.max_corner[2]);
			b.Vertices = pv + r.v_off; b.VertexCount = in.n_verts;
			b.Indices = pi + r.i_off; b.IndexCount = in.n_idx;
			for (int f = 0; f < 6; ++f) {
				b.TVertices[f] = pv + r.tv_off[f]; b.TVertexCount[f] = in.n_tverts[f];
				b.TIndices[f] = pi + r.ti_off[f]; b.TIndexCount[f] = in.n_tidx[f];
			}
		}
	}
	s.Levels.swap(built);
	return true;
}

} // namespace

Modification* Modification::Create()
{
	ModificationImpl* m = new ModificationImpl;
	m->Map = nullptr;
	return m;
}

Modification::~Modification() {}

// One device context with the grid mirrored in it and the caches of the last surface it produced (the reference keeps
// those caches - consistency bitmaps, material caches, slot maps - inside the PolygonSurface, src/TransVoxelImpl.h:81-95;
// here they live in HBM, owned jointly by the Polygonizer that made the surface and by the surface itself).
struct DeviceState
{
	vx_ctx* Ctx = nullptr;        // the primary context: the whole grid, every level, the caches a Modification continues from
	uint64_t ResidentGridUid = 0; // VoxelGrid::Uid of the grid mirrored in HBM (0 = none; never compare addresses: they get reused)
	uint64_t ResidentGeneration = 0;
	~DeviceState()
	{
		if (Ctx) vx_ctx_destroy(Ctx);
	}

	// InitializeVoxels brings one context up ahead of time (HIP runtime, code objects, the context's constant tables: ~0.3 s that
	// would otherwise land on the application's first Execute); the first Polygonizer adopts it.
	static std::mutex& PrewarmLock() { static std::mutex m; return m; }
	static std::shared_ptr<DeviceState>& Prewarmed() { static std::shared_ptr<DeviceState> d; return d; }

	static std::shared_ptr<DeviceState> Create()
	{
		{
			std::lock_guard<std::mutex> g(PrewarmLock());
			if (Prewarmed()) { std::shared_ptr<DeviceState> d; d.swap(Prewarmed()); return d; }
		}
		return CreateFresh(LS_CriticalError);
	}
	// (severity: what a failure is worth to the caller - an Execute cannot go on without a device; InitializeVoxels' warm-up can)
	static std::shared_ptr<DeviceState> CreateFresh(LogSeverity severity)
	{
		std::shared_ptr<DeviceState> d(new DeviceState);
		if (vx_ctx_create(0, &d->Ctx) != VX_OK) {
			d->Ctx = nullptr;
			Log(severity, "Voxels: no usable HIP device (libvoxels_hip has no CPU fallback)");
			return nullptr;
		}
		return d;
	}

	// mirror the host grid into HBM: whole grid the first time, edited blocks afterwards
	bool SyncGrid(VoxelGrid& g)
	{
		static const bool trace = getenv("VOXELS_TRACE") != nullptr;
		auto t0 = std::chrono::steady_clock::now();
		auto lap = [&](const char* what) {
			if (!trace) return;
			const auto t1 = std::chrono::steady_clock::now();
			fprintf(stderr, "[Voxels]   %-26s %8.3f ms\n", what, std::chrono::duration<double, std::milli>(t1 - t0).count());
			t0 = t1;
		};
		std::vector<uint8_t> flags;
		if (ResidentGridUid != g.Uid()) {
			// a grid that came from Grid::Load and was not edited since travels as its (much smaller) file and is
			// expanded on the device (its BF_Empty flags are in the file); anything else as dense fields
			const std::vector<char>* file = g.PristineFile();
			if (!file) g.EmptyFlags(flags);
			const int rc = file ? vx_grid_upload_packed(Ctx, file->data(), file->size())
			                    : vx_grid_upload(Ctx, g.Size(), g.Distances(), g.Materials(), g.Blends(), flags.data());
			lap(file ? "packed file to device" : "dense fields to device");
			if (rc != VX_OK) return false;
			// (the grid keeps its file copy - a thirtieth of the dense fields: handing 87 MB back to the system inside Execute
			// cost 7.6 ms, more than bringing the grid to the device)
			ResidentGridUid = g.Uid();
		} else if (ResidentGeneration != g.Generation()) {
			g.EmptyFlags(flags);
			std::vector<uint32_t> ids;
			g.DirtySince(ResidentGeneration, ids); // per-consumer: another mirror of the same grid is not affected
			std::vector<int8_t> d(ids.size() * 4096);
			std::vector<uint8_t> m(ids.size() * 4096), b(ids.size() * 4096);
			const uint32_t nb = g.BlocksPerAxis();
			for (size_t i = 0; i < ids.size(); ++i)
				g.GetBlock(ids[i] % nb, (ids[i] / nb) % nb, ids[i] / (nb * nb), d.data() + i * 4096, m.data() + i * 4096, b.data() + i * 4096);
			if (vx_grid_update_blocks(Ctx, (uint32_t)ids.size(), ids.data(), d.data(), m.data(), b.data(), flags.data()) != VX_OK) return false;
		}
		ResidentGeneration = g.Generation();
		return true;
	}

	bool SyncMaterials(const MaterialMap* materials)
	{
		uint8_t lut[256 * 6], valid[256];
		memset(lut, 0, sizeof(lut));
		for (int id = 0; id < 256; ++id) {
			MaterialMap::Material* m = materials ? materials->GetMaterial((unsigned char)id) : nullptr;
			valid[id] = m ? 1 : 0;
			if (m) { memcpy(lut + id * 6, m->DiffuseIds0, 3); memcpy(lut + id * 6 + 3, m->DiffuseIds1, 3); }
		}
		return vx_material_lut(Ctx, lut, valid) == VX_OK;
	}
};

// ------------------------------------------------------------------------------------------------ polygonizer
class TransVoxelImpl
{
public:
	std::shared_ptr<DeviceState> Device; // where this Polygonizer's full runs happen
	// Helper contexts, one per further device (VOXELS_DEVICES = N; on a box with fewer devices they share them): each holds a
	// slab of rows of the grid, polygonizes the levels whose blocks fit the slab and delivers those meshes over its own link
	// - of a 1024^3 Execute 10 of 16 ms are the meshes on their way to the host.  They belong to the Polygonizer, not to
	// the device state of one surface: what a helper produced lives on in host arenas the surface owns, so the next
	// Execute reuses the same N contexts however many older surfaces are still alive.
	std::vector<vx_ctx*> Helpers;

	~TransVoxelImpl()
	{
		for (vx_ctx* h : Helpers) if (h) vx_ctx_destroy(h);
	}

	// helper contexts for slabs of n / count rows (created on first use, kept); false: the grid is not cut that way
	bool EnsureHelpers(unsigned n, unsigned count)
	{
		if (count < 2 || n < 16 * count || (n / 16) % count) return false; // (every helper needs at least one block layer)
		int devices = 0;
		if (vx_device_count(&devices) != VX_OK || devices < 1) return false;
		while (Helpers.size() < count) {
			vx_ctx* h = nullptr;
			if (vx_ctx_create((int)(Helpers.size() % (size_t)devices), &h) != VX_OK) return false;
			Helpers.push_back(h);
		}
		return true;
	}

	static void FillStats(vx_ctx* ctx, PolygonizationStatistics& st)
	{
		uint32_t s[20];
		if (vx_stats(ctx, s) != VX_OK) memset(s, 0, sizeof(s));
		st.BlocksCalculated = s[0]; st.TrivialCells = s[1]; st.NonTrivialCells = s[2]; st.DegenerateTrianglesRemoved = s[3];
		for (int i = 0; i < 16; ++i) st.PerCaseCellsCount[i] = s[4 + i];
	}

	static unsigned RequestedDevices()
	{
		const char* e = getenv("VOXELS_DEVICES");
		const int v = e ? atoi(e) : 1;
		return v > 1 ? (unsigned)v : 1u;
	}

	// One helper's share of a full Execute: its slab of the host grid to its device, the levels whose blocks fit the slab,
	// every mesh into a page-locked copy of its own, and the block tables of those levels.
	struct HelperResult {
		bool Ok = false;
		vx_host_meshes Meshes = { nullptr, nullptr, 0, 0, nullptr };
		std::vector<std::vector<vx_block_info> > Infos;
		std::vector<std::vector<vx_block_ranges> > Ranges;
		uint32_t Stats[20] = {};
	};
	static void RunHelper(vx_ctx* ctx, const VoxelGrid* g, const uint8_t* flags, const uint8_t* lut, const uint8_t* valid, unsigned y0, unsigned y1, unsigned levels, HelperResult* out)
	{
		vx_exec_info info;
		if (vx_grid_upload_slab_y(ctx, g->Size(), y0, y1, g->Distances(), g->Materials(), g->Blends(), flags) != VX_OK) return;
		if (vx_material_lut(ctx, lut, valid) != VX_OK) return;
		if (vx_polygonize(ctx, levels, &info) != VX_OK) return;
		if (vx_host_meshes_acquire(ctx, &out->Meshes) != VX_OK) return;
		if (vx_stats(ctx, out->Stats) != VX_OK) return;
		out->Infos.resize(levels); out->Ranges.resize(levels);
		for (unsigned l = 0; l < levels; ++l) {
			uint32_t nb = 0;
			if (vx_level_counts(ctx, l, &nb, nullptr) != VX_OK) return;
			out->Infos[l].resize(nb); out->Ranges[l].resize(nb);
			if (nb && (vx_download_level(ctx, l, out->Infos[l].data(), nullptr, nullptr, nullptr, nullptr) != VX_OK || vx_level_ranges(ctx, l, out->Ranges[l].data()) != VX_OK)) return;
		}
		out->Ok = true;
	}

	static void FillBlock(BlockImpl& b, const vx_block_info& in, const vx_block_ranges& r, const PolygonVertex* pv, const unsigned* pi)
	{
		b.Id = in.id;
		b.MinCorner = float3(in.min_corner[0], in.min_corner[1], in.min_corner[2]);
		b.MaxCorner = float3(in.max_corner[0], in.max_corner[1], in.max_corner[2]);
		b.Vertices = pv + r.v_off; b.VertexCount = in.n_verts;
		b.Indices = pi + r.i_off; b.IndexCount = in.n_idx;
		for (int f = 0; f < 6; ++f) {
			b.TVertices[f] = pv + r.tv_off[f]; b.TVertexCount[f] = in.n_tverts[f];
			b.TIndices[f] = pi + r.ti_off[f]; b.TIndexCount[f] = in.n_tidx[f];
		}
	}

	// A full Execute on several devices.  The primary context holds the whole grid and runs from the first level the helpers
	// do not cover (vx_polygonize_from): it builds the caches of every level - what a later Modification continues from - and
	// meshes the few blocks of the levels coarser than a slab, which read voxels of every slab.  The meshes of the finer
	// levels - nearly all bytes of the result - are the helpers', each for its slab, and travel to the host over the helpers'
	// own links, side by side.  (A dense surface makes the primary mesh everything, as until round 4: info.first_meshed_level.)
	PolygonSurface* ExecuteOnDevices(VoxelGrid* g, const MaterialMap* materials, unsigned devices)
	{
		const unsigned n = g->Size(), rows = n / devices;
		unsigned helperLevels = 1;
		for (unsigned m = rows / 16; m && !(m & 1u); m >>= 1) ++helperLevels; // coarsest block that divides the slab (and its origin); (rows >= 16: EnsureHelpers)
		unsigned refLevels = 1;
		for (unsigned v = n / 16; v >>= 1;) ++refLevels;
		if (helperLevels > refLevels) helperLevels = refLevels;
		std::vector<uint8_t> flags;
		g->EmptyFlags(flags);
		uint8_t lut[256 * 6], valid[256];
		memset(lut, 0, sizeof(lut));
		for (int id = 0; id < 256; ++id) { // (MaterialMap::GetMaterial is only ever called from the calling thread)
			MaterialMap::Material* m = materials ? materials->GetMaterial((unsigned char)id) : nullptr;
			valid[id] = m ? 1 : 0;
			if (m) { memcpy(lut + id * 6, m->DiffuseIds0, 3); memcpy(lut + id * 6 + 3, m->DiffuseIds1, 3); }
		}
		std::vector<HelperResult> res(devices);
		// (the helpers read `flags`, `lut`, `valid` and `g`: whatever happens below - a thread that cannot be started, an
		// allocation that fails - every started thread is joined before those go out of scope)
		struct Joiner {
			std::vector<std::thread> threads;
			void join() { for (std::thread& t : threads) if (t.joinable()) t.join(); }
			~Joiner() { join(); }
		} helpers;
		std::vector<std::thread>& threads = helpers.threads;
		threads.reserve(devices);
		for (unsigned i = 0; i < devices; ++i)
			threads.emplace_back(&TransVoxelImpl::RunHelper, Helpers[i], (const VoxelGrid*)g, (const uint8_t*)flags.data(), (const uint8_t*)lut, (const uint8_t*)valid, i * rows, (i + 1) * rows, helperLevels, &res[i]);
		// the primary, on the calling thread
		vx_ctx* ctx = Device->Ctx;
		vx_exec_info info;
		// (the primary leaves the meshes of the levels the helpers cover to them: caches, bitmaps and slot maps of every level -
		// what a later Modification continues from - and the meshes of the coarser levels; info.first_meshed_level says whether
		// the run could be partial - it cannot for dense surfaces - or meshed everything as until round 4)
		bool ok = Device->SyncGrid(*g) && vx_material_lut(ctx, lut, valid) == VX_OK && vx_polygonize_from(ctx, 0, helperLevels, &info) == VX_OK;
		std::unique_ptr<SurfaceImpl> s(new SurfaceImpl);
		s->GridSize = n;
		s->Device = Device;
		s->Extents = float3((float)n, (float)n, (float)n);
		std::vector<std::vector<vx_block_info> > coarseInfos;
		if (ok) {
			s->Levels.resize(info.levels);
			s->Coarse.resize(info.levels);
			coarseInfos.resize(info.levels);
			for (unsigned l = helperLevels; ok && l < info.levels; ++l) {
				uint32_t nb = 0;
				uint64_t tot[4] = { 0, 0, 0, 0 };
				ok = vx_level_counts(ctx, l, &nb, tot) == VX_OK;
				if (!ok) break;
				SurfaceImpl::CoarseLevel& c = s->Coarse[l];
				coarseInfos[l].resize(nb);
				c.V.resize(tot[0]); c.I.resize(tot[1]); c.TV.resize(tot[2]); c.TI.resize(tot[3]);
				ok = !nb || vx_download_level(ctx, l, coarseInfos[l].data(), (vx_vertex*)c.V.data(), c.I.data(), (vx_vertex*)c.TV.data(), c.TI.data()) == VX_OK;
			}
			FillStats(ctx, s->Stats);
		}
		helpers.join();
		if (ok && info.first_meshed_level) {
			// the statistics of the levels the primary did not mesh are the helpers' (every counter adds up over slabs and levels)
			bool all = true;
			for (unsigned i = 0; i < devices; ++i) all = all && res[i].Ok;
			if (all) for (unsigned i = 0; i < devices; ++i) {
				const uint32_t* h = res[i].Stats;
				s->Stats.BlocksCalculated += h[0]; s->Stats.TrivialCells += h[1]; s->Stats.NonTrivialCells += h[2]; s->Stats.DegenerateTrianglesRemoved += h[3];
				for (int k = 0; k < 16; ++k) s->Stats.PerCaseCellsCount[k] += h[4 + k];
			}
		}
		for (unsigned i = 0; i < devices; ++i) { s->ShardMeshes.push_back(res[i].Meshes); ok = ok && res[i].Ok; } // (the surface owns the copies from here on)
		if (!ok) { Log(LS_Error, vx_last_error(ctx)); for (vx_ctx* h : Helpers) if (h && *vx_last_error(h)) Log(LS_Error, vx_last_error(h)); return nullptr; }
		// the finer levels: every helper's blocks, merged by id (ids number the blocks of the whole grid: = GetBlockForLevel order)
		for (unsigned l = 0; l < helperLevels && l < info.levels; ++l) {
			struct Ref { unsigned id, helper, k; };
			std::vector<Ref> order;
			for (unsigned i = 0; i < devices; ++i) for (unsigned k = 0; k < res[i].Infos[l].size(); ++k) order.push_back({ res[i].Infos[l][k].id, i, k });
			std::sort(order.begin(), order.end(), [](const Ref& a, const Ref& b) { return a.id < b.id; });
			std::vector<BlockImpl>& out = s->Levels[l];
			out.resize(order.size());
			for (size_t q = 0; q < order.size(); ++q) {
				const Ref& r = order[q];
				FillBlock(out[q], res[r.helper].Infos[l][r.k], res[r.helper].Ranges[l][r.k], (const PolygonVertex*)res[r.helper].Meshes.verts, res[r.helper].Meshes.indices);
			}
		}
		if (getenv("VOXELS_TRACE")) fprintf(stderr, "[Voxels] %u devices: helpers mesh the levels below %u, the primary meshed from level %u (device %.3f ms)\n", devices, helperLevels, info.first_meshed_level, info.device_ms);
		if (info.first_meshed_level) {
			s->HelperLevels = std::min<unsigned>(info.first_meshed_level, helperLevels);
			s->HelperBlocks.assign(s->Levels.begin(), s->Levels.begin() + std::min<size_t>(s->HelperLevels, s->Levels.size()));
		}
		// the coarser levels: compact arrays in block order (vx_download_level's layout)
		for (unsigned l = helperLevels; l < info.levels; ++l) {
			const SurfaceImpl::CoarseLevel& c = s->Coarse[l];
			std::vector<BlockImpl>& out = s->Levels[l];
			out.resize(coarseInfos[l].size());
			size_t ov = 0, oi = 0, otv = 0, oti = 0;
			for (size_t k = 0; k < out.size(); ++k) {
				const vx_block_info& in = coarseInfos[l][k];
				vx_block_ranges r;
				r.v_off = (uint32_t)ov; r.i_off = (uint32_t)oi;
				ov += in.n_verts; oi += in.n_idx;
				BlockImpl& b = out[k];
				b.Id = in.id;
				b.MinCorner = float3(in.min_corner[0], in.min_corner[1], in.min_corner[2]);
				b.MaxCorner = float3(in.max_corner[0], in.max_corner[1], in.max_corner[2]);
				b.Vertices = c.V.data() + r.v_off; b.VertexCount = in.n_verts;
				b.Indices = c.I.data() + r.i_off; b.IndexCount = in.n_idx;
				for (int f = 0; f < 6; ++f) {
					b.TVertices[f] = c.TV.data() + otv; b.TVertexCount[f] = in.n_tverts[f]; otv += in.n_tverts[f];
					b.TIndices[f] = c.TI.data() + oti; b.TIndexCount[f] = in.n_tidx[f]; oti += in.n_tidx[f];
				}
			}
		}
		return s.release();
	}

	PolygonSurface* Execute(const Grid& grid, const MaterialMap* materials, Modification* modification)
	{
		VoxelGrid* g = grid.GetInternalRepresentation();
		if (!g) return nullptr;
		// VOXELS_TRACE=1: where a call spends its time (stderr)
		static const bool trace = getenv("VOXELS_TRACE") != nullptr;
		auto t0 = std::chrono::steady_clock::now();
		auto lap = [&](const char* what) {
			if (!trace) return;
			const auto t1 = std::chrono::steady_clock::now();
			fprintf(stderr, "[Voxels] %-28s %8.3f ms\n", what, std::chrono::duration<double, std::milli>(t1 - t0).count());
			t0 = t1;
		};
		if (!modification) {
			// A surface that is still alive shares the device state it was made in (its caches are there, and a later
			// Modification needs them): a new full run then gets a context of its own instead of overwriting them.
			if (!Device || Device.use_count() > 1) Device = DeviceState::Create();
			if (!Device) return nullptr;
			const unsigned devices = RequestedDevices();
			if (devices > 1 && EnsureHelpers(g->Size(), devices)) {
				PolygonSurface* r = ExecuteOnDevices(g, materials, devices);
				lap("Execute on several devices");
				return r;
			}
			vx_ctx* ctx = Device->Ctx;
			if (!Device->SyncGrid(*g) || !Device->SyncMaterials(materials)) { Log(LS_Error, vx_last_error(ctx)); return nullptr; }
			lap("grid + materials to device");
			vx_exec_info info;
			if (vx_polygonize(ctx, 0, &info) != VX_OK) { Log(LS_Error, vx_last_error(ctx)); return nullptr; }
			lap("vx_polygonize");
			SurfaceImpl* s = new SurfaceImpl;
			s->GridSize = g->Size();
			s->Device = Device;
			s->Extents = float3((float)g->Size(), (float)g->Size(), (float)g->Size());
			if (!FetchSurface(ctx, info.levels, *s)) { Log(LS_Error, vx_last_error(ctx)); delete s; return nullptr; }
			lap("meshes to host + block views");
			FillStats(ctx, s->Stats);
			return s;
		}
		// Incremental run over the modification's dirty box: the same Map object is updated in place, in the device state
		// the surface carries - whichever Polygonizer is asked to do it (the reference's caches travel with the
		// PolygonSurface the same way, src/TransVoxelImpl.cpp:362-364).
		SurfaceImpl* s = static_cast<SurfaceImpl*>(modification->Map);
		ModificationImpl* mod = static_cast<ModificationImpl*>(modification);
		if (!s || !s->Device) {
			Log(LS_Error, "Modification: no surface to update (Modification::Map)");
			return nullptr;
		}
		DeviceState& dev = *s->Device;
		vx_ctx* ctx = dev.Ctx;
		if (!dev.SyncGrid(*g) || !dev.SyncMaterials(materials)) { Log(LS_Error, vx_last_error(ctx)); return nullptr; }
		const float mn[3] = { modification->MinCornerModified.x, modification->MinCornerModified.y, modification->MinCornerModified.z };
		const float mx[3] = { modification->MaxCornerModified.x, modification->MaxCornerModified.y, modification->MaxCornerModified.z };
		// room for every block of every level: the run cannot report more, so it never has to be repeated for space
		size_t allBlocks = 0;
		for (uint32_t cnt = g->Size() / 16; cnt; cnt >>= 1) allBlocks += (size_t)cnt * cnt * cnt;
		std::vector<uint32_t> ids(allBlocks + 1);
		uint32_t count = 0;
		vx_exec_info info;
		int rc = vx_polygonize_dirty(ctx, mn, mx, &info, ids.data(), (uint32_t)ids.size(), &count);
		if (rc == VX_OK && count > ids.size()) {
			Log(LS_Error, "Modification: too many modified blocks");
			return nullptr;
		}
		if (rc != VX_OK) { Log(LS_Error, vx_last_error(ctx)); return nullptr; }
		mod->ModifiedBlocks.insert(mod->ModifiedBlocks.end(), ids.begin(), ids.begin() + count);
		if (!FetchSurface(ctx, std::min<unsigned>(info.levels, (unsigned)s->Levels.size()), *s)) { Log(LS_Error, vx_last_error(ctx)); return nullptr; }
		if (s->HelperLevels) {
			// The finer levels of this surface were meshed by helper devices only: the primary's lists hold just what Modifications
			// rebuilt.  A level is then the helper-made blocks that no Modification has replaced - minus the ones whose minimal
			// corner lies in this one's box (src/TransVoxelImpl.cpp:433-450: touched blocks plus one ring, per level) - followed by
			// the primary's blocks, which is the order the reference's vector ends up in (:443-464: erase in place, append).
			for (unsigned l = 0; l < s->HelperLevels && l < s->Levels.size(); ++l) {
				const float bm = (float)(16u << l), ext = (float)s->GridSize;
				float lo[3], hi[3];
				for (int k = 0; k < 3; ++k) {
					lo[k] = std::floor(mn[k] / bm - 1.0f) * bm; hi[k] = std::floor(mx[k] / bm + 2.0f) * bm;
					lo[k] = std::min(std::max(lo[k], 0.f), ext); hi[k] = std::min(std::max(hi[k], 0.f), ext);
				}
				std::vector<BlockImpl>& kept = s->HelperBlocks[l];
				kept.erase(std::remove_if(kept.begin(), kept.end(), [&](const BlockImpl& b) {
					return b.MinCorner.x >= lo[0] && b.MinCorner.y >= lo[1] && b.MinCorner.z >= lo[2] && b.MinCorner.x < hi[0] && b.MinCorner.y < hi[1] && b.MinCorner.z < hi[2]; }), kept.end());
				std::vector<BlockImpl> merged;
				merged.reserve(kept.size() + s->Levels[l].size());
				merged.insert(merged.end(), kept.begin(), kept.end());
				merged.insert(merged.end(), s->Levels[l].begin(), s->Levels[l].end());
				s->Levels[l].swap(merged);
			}
			s->Coarse.clear(); // (the coarser levels' views point into the primary's page-locked copy now)
		} else {
			// (a surface made by several devices whose primary meshed everything continues on that context, which holds every
			// level and every cache: all views now point into the primary's copy, the helpers' copies can go)
			s->ReleaseShardCopies();
		}
		FillStats(ctx, s->Stats);
		return s;
	}
};

Polygonizer::Polygonizer() : m_Impl(new TransVoxelImpl) {}
Polygonizer::~Polygonizer() { delete m_Impl; }

PolygonSurface* Polygonizer::Execute(const Grid& grid, const MaterialMap* materials, Modification* modification)
{
	return m_Impl->Execute(grid, materials, modification);
}

} // namespace Voxels

// ------------------------------------------------------------------------------------------------ library init
extern "C" Voxels::InitError InitializeVoxels(int version, Voxels::LogMessage logger, Voxels::VoxelsAllocators*)
{
	if ((VOXELS_VERSION & 0xFFFF) != (version & 0xFFFF)) return Voxels::IE_VersionMismatch; // low 16 bits, like Voxels.cpp:35-40
	Voxels::g_Logger = logger;
	Voxels::g_Initialized = true;
	char buffer[128];
	snprintf(buffer, sizeof(buffer), "Voxels library initialized - ver. %#010x (MI355X / %s)", VOXELS_VERSION, vx_backend());
	Voxels::Log(Voxels::LS_Info, buffer);
	// One-time costs belong here, not into the application's first Execute (408 ms at 1024^3 without this: HIP runtime and
	// code objects ~0.25 s, page-locking the first mesh arena ~0.13 s).  VOXELS_NO_PREWARM=1 skips the context,
	// VOXELS_PREWARM_MB=<n> also page-locks an arena for n MB of meshes (vertices : indices as 5 : 1 by bytes).
	if (!getenv("VOXELS_NO_PREWARM")) {
		// (a host without a usable device still initialises - IE_Ok like the reference - and is told so as a warning, not as
		// an error; the first Execute reports the missing device as the error it then is)
		std::shared_ptr<Voxels::DeviceState> d = Voxels::DeviceState::CreateFresh(Voxels::LS_Warning);
		if (d) {
			// the kernels are loaded on first use, a few milliseconds each: one small synthetic grid through every call an
			// Execute makes (terrain on the device -> polygonize -> write as a file -> expand the file -> polygonize -> meshes
			// to the host) pays that here as well
			{
				vx_exec_info info;
				vx_host_meshes hm = { nullptr, nullptr, 0, 0, nullptr };
				uint64_t size = 0;
				uint8_t lut[256 * 6], valid[256];
				memset(lut, 0, sizeof(lut)); memset(valid, 1, sizeof(valid));
				bool ok = vx_grid_create_terrain(d->Ctx, 64, 1) == VX_OK && vx_material_lut(d->Ctx, lut, valid) == VX_OK && vx_polygonize(d->Ctx, 0, &info) == VX_OK;
				std::vector<char> file;
				if (ok && vx_grid_pack(d->Ctx, nullptr, 0, &size) == VX_OK) {
					file.resize((size_t)size);
					ok = vx_grid_pack(d->Ctx, file.data(), size, &size) == VX_OK && vx_grid_upload_packed(d->Ctx, file.data(), size) == VX_OK && vx_polygonize(d->Ctx, 0, &info) == VX_OK;
				}
				if (ok && vx_host_meshes_acquire(d->Ctx, &hm) == VX_OK) vx_host_meshes_release(hm.arena);
				(void)vx_grid_invalidate(d->Ctx); // (whatever the application uploads next replaces this grid; nothing of it is kept)
				(void)vx_ctx_forget_hints(d->Ctx); // (... nor what the toy terrain taught the context about capacity classes: the application's first grid may be dense)
				if (!ok) Voxels::Log(Voxels::LS_Warning, "Voxels: the warm-up run of InitializeVoxels failed (the first Execute pays the one-time costs instead)");
			}
			const char* mb = getenv("VOXELS_PREWARM_MB");
			const uint64_t bytes = mb ? (uint64_t)atoll(mb) << 20 : 0;
			if (bytes) (void)vx_host_meshes_reserve(d->Ctx, bytes * 5 / 6 / sizeof(Voxels::PolygonVertex), bytes / 6 / 4);
			std::lock_guard<std::mutex> g(Voxels::DeviceState::PrewarmLock());
			Voxels::DeviceState::Prewarmed() = d;
		}
	}
	return Voxels::IE_Ok;
}

extern "C" void DeinitializeVoxels()
{
	{
		std::lock_guard<std::mutex> g(Voxels::DeviceState::PrewarmLock());
		Voxels::DeviceState::Prewarmed().reset();
	}
	vx_host_meshes_trim();
	Voxels::Log(Voxels::LS_Info, "Voxels library deinitialized");
	Voxels::g_Logger = nullptr;
	Voxels::g_Initialized = false;
}

extern "C" unsigned GetBuildVersion() { return VOXELS_VERSION; }

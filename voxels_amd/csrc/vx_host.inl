// vx_host.inl — host orchestration behind include/voxels_hip.h, written against a small backend interface
// (memory + the five pipeline stages).  voxels_amd/csrc/vx_hip.hip instantiates it with the HIP backend (the
// product, gfx950 kernels); tests/emu/emu.cpp instantiates it with a CPU emulation of the same phases to test
// the formulation and this host logic without a GPU.  There is NO runtime switch between the two: each shared
// library is compiled with exactly one backend.
//
// The including file must define, before including this file:
//   struct Backend { init/alloc/free/fill/h2d/d2h/sync/begin_timing/end_timing_ms + run_* stages }  (see below)
//   VX_BACKEND_NAME
#include "../../include/voxels_hip.h"
#include "tv_block.h"
#include "tv_fast0.h"
#include "tv_fast1.h"

#include <algorithm>
#include <atomic>
#include <chrono>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <memory>
#include <mutex>
#include <string>
#include <thread>
#include <vector>

namespace {

struct HostArena;

#include "tv_tables.inc"

using namespace tv;

struct ExecParams {
	Globals G;
	LevelDesc levels[MAX_LEVELS];
	Pools P;
	const u8* tables; // TAB_BYTES image in device memory
};

// the meshes of a listed block live in the output pools (offsets in rec): full runs rewrite the pools, incremental runs
// append to them
typedef ListedBlock EmittedBlock;

// largest grid edge: 2048 = 8 LOD levels (MAX_LEVELS) and 32-bit element offsets inside a block neighbourhood
enum { VX_MAX_GRID = 2048 };
enum { HDR_WORDS = 576, HDR_LISTS = 8, HDR_CURSORS = 32, HDR_STATS = 128, HDR_WORK = 160, HDR_LARGE = 176, HDR_SLOW = 224, HDR_UPPER = 256, HDR_GIVEUP = 288, HDR_PUBLISHED = 289 /* workgroups of the list pass that are done */, HDR_L0HEAD = 320 /* eight heads, one line each: k_main's level-0 queue per XCD */, HDR_PARTIALS = 32768 }; // counters spread over 128-byte lines

} // namespace

struct vx_ctx {
	Backend be;
	std::string err;
	// grid
	u32 n = 0, zBegin = 0, zEnd = 0;
	u32 yBegin = 0, yEnd = 0;            // owned rows (slabs cut along y); [0, n) otherwise
	int distY0 = 0, matY0 = 0;           // global y of row 0 of every resident plane
	u32 distRows = 0, matRows = 0;       // rows per resident plane (n for a whole grid or a z-slab)
	bool ownsGrid = false;
	bool ownsSlab = false;               // the attached slab buffers are the context's own (vx_grid_upload_slab_y)
	void *dDist = nullptr, *dMat = nullptr, *dBlend = nullptr, *dFlags = nullptr;
	void* dBlockClass = nullptr;                           // per level-0 block scratch of the classify pass
	void* dFlatItems = nullptr;                            // active blocks of the levels >= 1 in level order (Globals::flatItems)
	void* dSlowItems[2] = { nullptr, nullptr };            // blocks handed from the fast regular passes to the general one (level 0 | levels >= 1)
	PyramidLevel pyr[PYRAMID_LEVELS];                    // lattice copies of the distance field for levels 1..3
	XPlanes xp[XPLANE_LEVELS];                           // yz-planes of the lattices 0..2 at every 32nd x
	// brick mirrors of the three fields (tv_core.h GridView): resident block rows [brickYb0, +brickRowsY) of the block
	// planes [brickZb0, +brickPlanesZ); stale = everything has to be copied again before the next polygonization
	void* dBrick[3] = { nullptr, nullptr, nullptr };
	void* dBlockSign = nullptr;          // per level-0 block: sign summary of its samples (MirrorState)
	u32 brickN = 0, brickYb0 = 0, brickZb0 = 0, brickRowsY = 0, brickPlanesZ = 0;
	bool bricksStale = true;
	void* dListCounts = nullptr;                         // listed blocks per LIST_WG block coordinates, all levels (LevelDesc::listCounts)
	// Two sets of what a full run needs in its start state - the header's counters, the block -> slot maps of the levels >= 1,
	// the list counts: a run works on one set, and its last kernel (k_tail) resets the other for the run behind it, which then
	// starts without k_reset.  dHeader / dListCounts / lv[L].slotOf, nActive, listCounts always name the current set.
	void* headerSet[2] = { nullptr, nullptr };
	void* listCountSet[2] = { nullptr, nullptr };
	int* upperMaps[2][MAX_LEVELS] = {};
	u32 runSet = 0;
	bool otherSetClean = false;
	u32 listWgs = 0;
	void* haloBuf[4] = { nullptr, nullptr, nullptr, nullptr }; // staging of the halo messages: send below, send above, receive from below, receive from above
	size_t haloCap[4] = { 0, 0, 0, 0 };
	int slabAxis = 0;                                    // how the grid was attached: 0 = not attached, 1 = slab of z-planes, 2 = slab of y-rows
	int commRanks = 0, commRank = 0;                     // RCCL communicator joined by vx_comm_init (0 = none)
	bool deviceLists = false;                            // the device block tables describe the current surface (full run; not after incremental runs)
	int distZ0 = 0, matZ0 = 0;
	// constant device data
	void *dLut = nullptr, *dTables = nullptr, *dHeader = nullptr; // header (HDR_WORDS u32): slot counts | vertex cursor | index cursor | overflow | stats[20] | workCount[8], one line each
	void *dDirty = nullptr, *dWork = nullptr, *dGather = nullptr;  // incremental runs: dirty block coords, work items, gathered records
	u32 dirtyCap = 0;
	void* dDirtyTicket = nullptr; // incremental runs as three launches: k_dirty_head's count of finished workgroups over all its launches
	u32 dirtyTickets = 0;         // ... and what the host knows it to be
	bool editRoomFailed = false;  // the first incremental run's reservation (pools for twice the live meshes, spare pair) failed: not retried per edit
	bool dirtyLargeHint = false;  // the last incremental run met blocks beyond the first capacity class on the levels >= 1 (their launches are made)
	bool dirtyLarge0Hint = false; // ... on level 0
	// level tables
	u32 tablesN = 0, tablesZb0 = 0, tablesZb1 = 0, tablesYb0 = 0, tablesYb1 = 0;
	u32 refLevels = 0;
	LevelDesc lv[MAX_LEVELS];
	std::vector<void*> levelAllocs;
	// pools
	void *dVerts = nullptr, *dIdx = nullptr;
	u32 vertCap = 0, idxCap = 0;
	void *dVertsSpare = nullptr, *dIdxSpare = nullptr; // vx_compact_pools: the pair the live meshes are packed into (kept; swaps roles with the pools)
	u32 spareVertCap = 0, spareIdxCap = 0;
	void* dSeg = nullptr;                              // ... and its copy list
	size_t segCap = 0;
	// results
	u32 levelsRun = 0;
	bool haveSurface = false;
	u32 nextId = 0;
	std::vector<EmittedBlock> blocks[MAX_LEVELS];
	HostArena* hostArena = nullptr; // page-locked host copy of the pools (vx_download_level; handed out by vx_host_meshes_acquire)
	uint64_t poolLineage = 0;             // changes whenever the pools are rewritten (a host copy of another lineage is useless)
	bool listsReady = false;
	bool liveKnown = false; uint64_t liveVerts = 0, liveIdx = 0; // what the block lists sum up to (live_totals), kept across incremental runs
	std::vector<unsigned char> blockDead[MAX_LEVELS]; // per level: empty, or one flag per list entry - blocks an incremental run dropped and nobody has asked for the lists since
	u32 deadBlocks = 0;                               // (the entries leave when the lists are next read: erasing from the middle of a list moved 400 KB per edit at 512^3)
	u32 poolVerts = 0, poolIdx = 0;
	u32 stats[20];
	u32 hdr[HDR_WORDS];
	u32* hdrPinned = nullptr; // pinned landing buffer of the header read-back
	void* dScratch = nullptr;  // small reusable device buffer (block id lists of edits)
	size_t scratchCap = 0;
	BlockRecord* hRecs = nullptr; // pinned staging for record read-back
	void *dBlobStage = nullptr, *hBlobStage = nullptr, *dWhereStage = nullptr, *hWhereStage = nullptr; // vx_grid_upload_packed: the file and its block offsets on their way to the device (kept across calls)
	size_t blobCap = 0, whereCap = 0;
	size_t hRecCap = 0;
	bool largeHint = true;      // launch the 4096-cell capacity class of the regular pass (unknown before the first run)
	bool stagedMain = false;    // the last run with stage timing used the single-stream form (vx_stage_layout)
	u32 runEpoch = 0;           // tag of the current full run in LevelDesc::matDone (Globals::epoch)
	u32 poolSlack = 1u << 16;   // VX_POOL_SLACK (read once at context creation; tests): the vertices of headroom the pool rules add - smallest pool, packing threshold, room for edits (indices: four times as many)
	bool hostTiming = false;    // VX_HOST_TIMING (read once at context creation): print where a vx_polygonize call spends host time
};

// every entry point makes the context's device the calling thread's current device (the HIP current device is per thread)
#define VX_ENTER(c) do { if (c) (c)->be.make_current(); } while (0)

namespace {

int fail(vx_ctx* c, int code, const std::string& msg)
{
	if (c) c->err = msg;
	return code;
}

u32 ref_levels(u32 n)
{
	u32 l = 0;
	for (u32 v = n >> 4; v >>= 1;) ++l;
	return l + 1; // TransVoxelImpl.cpp:490
}

void free_level_tables(vx_ctx* c)
{
	for (void* p : c->levelAllocs) c->be.free(p);
	c->levelAllocs.clear();
	c->tablesN = 0;
}

void release_grid(vx_ctx* c)
{
	if (c->ownsGrid || c->ownsSlab) {
		c->be.free(c->dDist); c->be.free(c->dMat); c->be.free(c->dBlend); c->be.free(c->dFlags);
	}
	c->dDist = c->dMat = c->dBlend = c->dFlags = nullptr;
	c->ownsGrid = false;
	c->ownsSlab = false;
	c->slabAxis = 0;
	c->bricksStale = true;
}

void free_bricks(vx_ctx* c)
{
	for (void*& b : c->dBrick) { c->be.free(b); b = nullptr; }
	c->be.free(c->dBlockSign); c->dBlockSign = nullptr;
	c->brickN = 0;
	c->bricksStale = true;
}

// layers of the dense fields that are resident, in global coordinates: { z0, z1, y0, y1 } for the distance field (dr) and
// for material / blend (mr).  A slab carries 1 distance layer below and 2 above, 1 material layer above
// (include/voxels_hip.h), clamped to the grid.
void resident_ranges(const vx_ctx* c, int dr[4], int mr[4])
{
	const int N = (int)c->n;
	const bool alongY = c->slabAxis == 2, slab = c->slabAxis != 0;
	const int zb = (int)c->zBegin, ze = (int)c->zEnd, yb = (int)c->yBegin, ye = (int)c->yEnd;
	dr[0] = (slab && !alongY) ? std::max(zb - 1, 0) : 0; dr[1] = (slab && !alongY) ? std::min(ze + 2, N) : N;
	dr[2] = (slab && alongY) ? std::max(yb - 1, 0) : 0;  dr[3] = (slab && alongY) ? std::min(ye + 2, N) : N;
	mr[0] = (slab && !alongY) ? zb : 0; mr[1] = (slab && !alongY) ? std::min(ze + 1, N) : N;
	mr[2] = (slab && alongY) ? yb : 0;  mr[3] = (slab && alongY) ? std::min(ye + 1, N) : N;
}

GridView resident_view(const vx_ctx* c)
{
	GridView g;
	g.dist = (const i8*)c->dDist; g.mat = (const u8*)c->dMat; g.blend = (const u8*)c->dBlend;
	g.n = (int)c->n; g.zOrigin = c->distZ0; g.zOriginMat = c->matZ0; g.yOrigin = c->distY0; g.yOriginMat = c->matY0;
	g.pitchY = (int)c->distRows; g.pitchYMat = (int)c->matRows;
	g.bDist = (const i8*)c->dBrick[0]; g.bMat = (const u8*)c->dBrick[1]; g.bBlend = (const u8*)c->dBrick[2];
	g.bYb0 = (int)c->brickYb0; g.bZb0 = (int)c->brickZb0; g.bRowsY = (int)c->brickRowsY;
	return g;
}

// what the kernels that keep the mirrors current need beside the view
MirrorState mirror_state(const vx_ctx* c)
{
	MirrorState ms;
	for (u32 L = 0; L < PYRAMID_LEVELS; ++L) ms.pyr[L] = c->pyr[L];
	for (u32 L = 0; L < XPLANE_LEVELS; ++L) ms.xp[L] = c->xp[L];
	ms.blockSign = (u16*)c->dBlockSign;
	ms.yBegin = (int)c->yBegin; ms.yEnd = (int)c->yEnd; ms.zBegin = (int)c->zBegin; ms.zEnd = (int)c->zEnd;
	return ms;
}

// Brick mirrors allocated for the resident block layers (halo layers included) and brought up to date.  Called at the
// start of every polygonization; between runs the mutation entry points re-copy what they touched (rebrick_*), so a run on
// an unchanged grid finds nothing to do here.
bool ensure_bricks(vx_ctx* c)
{
	if (!c->be.wants_bricks()) return true;
	const u32 nb = c->n / 16;
	int dr[4], mr[4];
	resident_ranges(c, dr, mr);
	const u32 zb0 = (u32)dr[0] / 16, zb1 = ((u32)dr[1] + 15) / 16, yb0 = (u32)dr[2] / 16, yb1 = ((u32)dr[3] + 15) / 16;
	if (!c->dBrick[0] || c->brickN != c->n || c->brickYb0 != yb0 || c->brickZb0 != zb0 || c->brickRowsY != yb1 - yb0 || c->brickPlanesZ != zb1 - zb0) {
		free_bricks(c);
		const size_t bytes = (size_t)nb * (yb1 - yb0) * (zb1 - zb0) * BRICK_BYTES;
		for (void*& b : c->dBrick) b = c->be.alloc(bytes);
		c->dBlockSign = c->be.alloc((size_t)nb * nb * nb * 2);
		if (!c->dBrick[0] || !c->dBrick[1] || !c->dBrick[2] || !c->dBlockSign) { free_bricks(c); return false; }
		if (!c->be.fill(c->dBlockSign, 0, (size_t)nb * nb * nb * 2)) { free_bricks(c); return false; } // blocks outside the resident range: unknown
		c->brickN = c->n; c->brickYb0 = yb0; c->brickZb0 = zb0; c->brickRowsY = yb1 - yb0; c->brickPlanesZ = zb1 - zb0;
	}
	if (c->bricksStale) {
		const int box[4] = { (int)yb0, (int)yb1, (int)zb0, (int)zb1 };
		c->be.run_rebrick(resident_view(c), dr, mr, mirror_state(c), box, nullptr, 0);
		c->bricksStale = false;
	}
	return true;
}

// the mirrors of the listed blocks (device id list) follow a change of the dense fields; nothing to do while the mirrors
// wait for a full copy anyway (the rows of a halo exchange are written into the mirrors by its unpack kernel)
void rebrick_blocks(vx_ctx* c, const u32* dIds, u32 count)
{
	if (!c->be.wants_bricks() || c->bricksStale || !c->dBrick[0] || !count) return;
	int dr[4], mr[4];
	resident_ranges(c, dr, mr);
	c->be.run_rebrick(resident_view(c), dr, mr, mirror_state(c), nullptr, dIds, count);
}


// the set of counters, upper-level maps and list counts the next full run works on (vx_ctx::headerSet)
void select_run_set(vx_ctx* c, u32 set)
{
	c->runSet = set;
	c->dHeader = c->headerSet[set];
	c->dListCounts = c->listCountSet[set];
	size_t at = 0;
	for (u32 L = 0; L < c->refLevels && L < MAX_LEVELS; ++L) {
		LevelDesc& d = c->lv[L];
		d.nActive = (u32*)c->dHeader + L;
		if (L) d.slotOf = c->upperMaps[set][L];
		d.listCounts = (u32*)c->dListCounts + at;
		at += ((size_t)d.cnt * d.cnt * d.cnt + LIST_WG - 1) / LIST_WG;
	}
}

bool ensure_level_tables(vx_ctx* c)
{
	const u32 zb0 = c->zBegin / 16, zb1 = c->zEnd / 16, yb0 = c->yBegin / 16, yb1 = c->yEnd / 16;
	if (c->tablesN == c->n && c->tablesZb0 == zb0 && c->tablesZb1 == zb1 && c->tablesYb0 == yb0 && c->tablesYb1 == yb1) return true;
	free_level_tables(c);
	c->refLevels = ref_levels(c->n);
	auto alloc = [&](size_t bytes) -> void* {
		void* p = c->be.alloc(bytes ? bytes : 16);
		if (p) c->levelAllocs.push_back(p);
		return p;
	};
	for (u32 L = 0; L < MAX_LEVELS; ++L) memset(&c->lv[L], 0, sizeof(LevelDesc));
	memset(c->pyr, 0, sizeof(c->pyr));
	for (u32 L = 0; L < c->refLevels && L < MAX_LEVELS; ++L) {
		LevelDesc& d = c->lv[L];
		d.mult = 1u << L;
		d.cnt = (c->n >> 4) / d.mult;
		d.zb0 = zb0 >> L;
		d.zb1 = (zb1 + d.mult - 1) >> L;
		if (d.zb1 > d.cnt) d.zb1 = d.cnt;
		d.yb0 = yb0 >> L;
		d.yb1 = (yb1 + d.mult - 1) >> L;
		if (d.yb1 > d.cnt) d.yb1 = d.cnt;
		d.hasTransitions = (L > 0 && L != c->refLevels - 1) ? 1 : 0;
		const size_t total = (size_t)d.cnt * d.cnt * d.cnt;
		const size_t cap = (size_t)d.cnt * (d.yb1 > d.yb0 ? d.yb1 - d.yb0 : 0) * (d.zb1 > d.zb0 ? d.zb1 - d.zb0 : 0);
		d.cap = (u32)cap;
		d.slotOf = (int*)alloc(total * 4);
		c->upperMaps[0][L] = L ? d.slotOf : nullptr;
		c->upperMaps[1][L] = L ? (int*)alloc(total * 4) : nullptr;
		if (L && !c->upperMaps[1][L]) return false;
		// (level 0 has one map: runs without a reset write every entry of their block range, the entries outside it - a
		// rank's slab - stay as they are set here)
		if (!L && d.slotOf && !c->be.fill(d.slotOf, 0xFF, total * 4)) return false;
		d.slotCoord = (u32*)alloc(cap * 4);
		d.ntBits = (u32*)alloc(cap * 512);
		d.consBits = L ? nullptr : (u32*)alloc(cap * 512);
		d.cache = L ? (u16*)alloc(cap * BLOCK_CELLS * 2) : nullptr;
		d.skip = L ? nullptr : (u8*)alloc(cap);
		d.ntCount = (u16*)alloc(cap * 2);
		d.records = (BlockRecord*)alloc(cap * sizeof(BlockRecord));
		d.listed = (ListedBlock*)alloc(cap * sizeof(ListedBlock));
		if (L) {
			d.matDone = (unsigned long long*)alloc(cap * 8);
			if (!d.matDone || !c->be.fill(d.matDone, 0, cap * 8)) return false; // (no run's tag is 0)
		}
#if defined(VX_CASE_DUMP)
		d.caseDump = (u8*)alloc(cap * BLOCK_CELLS);
		d.trCaseDump = (u16*)alloc(cap * TR_CELLS * 2);
		if (!d.caseDump || !d.trCaseDump) return false;
#endif
		d.nActive = (u32*)c->dHeader + L;
		if (!d.slotOf || !d.slotCoord || !d.ntBits || !d.records || !d.listed || !d.ntCount || (L && !d.cache) || (!L && !d.skip)) return false;
		if (!L) {
			c->dBlockClass = alloc(total);
			c->dSlowItems[0] = alloc(cap * 4 + 16);
			if (!c->dSlowItems[0]) return false;
			if (!c->dBlockClass) return false;
		}
		// lattice copy of the distance samples of this level over the rank's rows / planes (one more than it owns: the far
		// samples of its last block layer)
		if (L >= 1 && L < PYRAMID_LEVELS && c->be.wants_pyramid()) {
			PyramidLevel& P = c->pyr[L];
			// entries [0, extent >> L] along every axis, in bricks of 16^3 entries
			P.bricksX = ((c->n >> L) >> 4) + 1;
			P.bricksY = (((c->yEnd - c->yBegin) >> L) >> 4) + 1;
			P.yOrigin = (int)(c->yBegin >> L); P.zOrigin = (int)(c->zBegin >> L);
			const size_t bricksZ = (((c->zEnd - c->zBegin) >> L) >> 4) + 1;
			P.data = (i8*)alloc((size_t)P.bricksX * P.bricksY * bricksZ * BRICK_BYTES + 64);
			if (!P.data) return false;
		}
	}
	// x-plane copies of the lattices 0..2 (whole extent in y and z: 35 MB at 1024^3; a slab only ever fills and reads its
	// own rows).  Part of the mirrors: filled by the next rebrick.
	for (u32 l = 0; l < XPLANE_LEVELS; ++l) {
		XPlanes& X = c->xp[l];
		X.data = nullptr; X.rows = X.stride = 0;
		const u32 nl = c->n >> l;
		if (!c->be.wants_pyramid() || nl < 32 || l + 1 >= c->refLevels) continue;
		X.rows = nl + 1;
		X.stride = (nl + 1 + 15u) & ~15u;
		X.data = (i8*)alloc((size_t)(nl / 32 + 1) * X.rows * X.stride + 64);
		if (!X.data) return false;
	}
	{
		size_t coarse = 0;
		for (u32 L = 1; L < c->refLevels && L < MAX_LEVELS; ++L) coarse += c->lv[L].cap;
		c->dSlowItems[1] = alloc(coarse * 4 + 16);
		if (!c->dSlowItems[1]) return false;
		c->dFlatItems = alloc(coarse * 16 + 16);
		if (!c->dFlatItems) return false;
	}
	{
		size_t wgs = 0;
		for (u32 L = 0; L < c->refLevels && L < MAX_LEVELS; ++L) wgs += ((size_t)c->lv[L].cnt * c->lv[L].cnt * c->lv[L].cnt + LIST_WG - 1) / LIST_WG;
		c->dListCounts = alloc(wgs * 4 + 16);
		c->listCountSet[0] = c->dListCounts;
		c->listCountSet[1] = alloc(wgs * 4 + 16);
		if (!c->dListCounts || !c->listCountSet[1]) return false;
		c->listWgs = (u32)wgs;
		size_t at = 0; // every level's segment of the counts (the list kernel's workgroups are laid out the same way, ListPlan::wgStart)
		for (u32 L = 0; L < c->refLevels && L < MAX_LEVELS; ++L) {
			c->lv[L].listCounts = (u32*)c->dListCounts + at;
			at += ((size_t)c->lv[L].cnt * c->lv[L].cnt * c->lv[L].cnt + LIST_WG - 1) / LIST_WG;
		}
	}
	c->tablesN = c->n; c->tablesZb0 = zb0; c->tablesZb1 = zb1; c->tablesYb0 = yb0; c->tablesYb1 = yb1;
	c->otherSetClean = false;
	select_run_set(c, 0);
	return true;
}

bool ensure_pools(vx_ctx* c, u32 needVerts, u32 needIdx)
{
	if (needVerts > c->vertCap) {
		c->be.free(c->dVerts);
		c->dVerts = c->be.alloc((size_t)needVerts * sizeof(PolyVertex));
		c->vertCap = c->dVerts ? needVerts : 0;
	}
	if (needIdx > c->idxCap) {
		c->be.free(c->dIdx);
		c->dIdx = c->be.alloc((size_t)needIdx * 4);
		c->idxCap = c->dIdx ? needIdx : 0;
	}
	return c->dVerts && c->dIdx;
}

void fill_params(vx_ctx* c, ExecParams& p, u32 levels)
{
	memset(&p, 0, sizeof(p));
	p.G.grid = resident_view(c);
	p.G.emptyFlags = (const u8*)c->dFlags;
	p.G.lut = (const u8*)c->dLut;
	p.G.stats = (u32*)c->dHeader + HDR_STATS;
	p.G.workCount = (u32*)c->dHeader + HDR_WORK;
	p.G.largeBlocks = (u32*)c->dHeader + HDR_LARGE;
	p.G.blockClass = (u8*)c->dBlockClass;
	p.G.blockSign = (const u16*)c->dBlockSign;
	p.G.slowItems[0] = (u32*)c->dSlowItems[0];
	p.G.slowItems[1] = (u32*)c->dSlowItems[1];
	p.G.slowCount = (u32*)c->dHeader + HDR_SLOW;
	p.G.flatItems = (FlatItem*)c->dFlatItems;
	p.G.slotCounts = (const u32*)c->dHeader;
	p.G.epoch = c->runEpoch;
	p.G.upperHead = (u32*)c->dHeader + HDR_UPPER;
	p.G.giveUp = (u32*)c->dHeader + HDR_GIVEUP;
	p.G.level0Head = (u32*)c->dHeader + HDR_L0HEAD;
	for (u32 L = 0; L < PYRAMID_LEVELS; ++L) p.G.pyr[L] = c->pyr[L];
	for (u32 L = 0; L < XPLANE_LEVELS; ++L) p.G.xp[L] = c->xp[L];
	p.G.levels = levels;
	p.G.refLevels = c->refLevels;
	for (u32 L = 0; L < MAX_LEVELS; ++L) p.levels[L] = c->lv[L];
	// listed blocks are counted where their records are written - on small block ranges, and in every run that is one stream
	// with k_main (a launch less); k_list_count otherwise
	if (!c->be.ancestors_with_classification(p, levels)) for (u32 L = 0; L < MAX_LEVELS; ++L) p.levels[L].listCounts = nullptr;
	p.P.verts = (PolyVertex*)c->dVerts;
	p.P.idx = (u32*)c->dIdx;
	p.P.cursors = (u32*)c->dHeader + HDR_CURSORS;
	p.P.vertCap = c->vertCap;
	p.P.idxCap = c->idxCap;
	p.tables = (const u8*)c->dTables;
}

void build_table_image(std::vector<u8>& img)
{
	img.assign(TAB_F0_BYTES, 0);
	memcpy(&img[TAB_REG_CLASS], TVT_REG_CLASS, 256);
	memcpy(&img[TAB_REG_CELL], TVT_REG_CELL, 256);
	memcpy(&img[TAB_TR_CLASS], TVT_TR_CLASS, 512);
	memcpy(&img[TAB_TR_CORNER], TVT_TR_CORNER, 16);
	memcpy(&img[TAB_TR_CELL], TVT_TR_CELL, 56 * 40);
	// vertex words -> 16-entry word table + 4-bit indices (padding entries, never read, index 0)
	auto pack = [&](const unsigned short* words, u32 cases, u32 edgeOff, u32 vertOff) {
		u16 distinct[16];
		u32 nDistinct = 0;
		for (u32 i = 0; i < cases * 12; ++i) {
			const u16 w = words[i];
			u32 idx = 0;
			if (w) {
				for (idx = 0; idx < nDistinct && distinct[idx] != w; ++idx) {}
				if (idx == nDistinct) {
					if (nDistinct == 16) abort(); // the Transvoxel tables have 12 / 16 distinct edge words
					distinct[nDistinct++] = w;
				}
			}
			img[vertOff + i / 2] |= (u8)(idx << ((i & 1u) * 4u));
		}
		memcpy(&img[edgeOff], distinct, nDistinct * 2);
	};
	pack(TVT_REG_VERT, 256, TAB_REG_EDGE, TAB_REG_VERT);
	for (u32 code = 0; code < 256; ++code) { // reuse slots a case owns: its vertices with reuse direction 8
		const u32 nv = TVT_REG_CELL[TVT_REG_CLASS[code] * 16] >> 4;
		u8 m = 0;
		for (u32 vi = 0; vi < nv; ++vi) { const u32 w = TVT_REG_VERT[code * 12 + vi]; if ((w >> 12) == 8u) m |= (u8)(1u << ((w >> 8) & 15)); }
		img[TAB_REG_OWN + code] = m;
	}
	for (u32 code = 0; code < 512; ++code) {
		const u32 nv = TVT_TR_CELL[(TVT_TR_CLASS[code] & 0x7F) * 40] >> 4;
		u16 m = 0;
		for (u32 vi = 0; vi < nv; ++vi) { const u32 w = TVT_TR_VERT[code * 12 + vi]; if ((w >> 12) == 8u) m |= (u16)(1u << ((w >> 8) & 15)); }
		memcpy(&img[TAB_TR_OWN + code * 2], &m, 2);
	}
	pack(TVT_TR_VERT, 512, TAB_TR_EDGE, TAB_TR_VERT);
	static_assert((u32)TAB_F0_CASE == (u32)TAB_BYTES, "the fast-pass tables follow the image of tv_core.h");
	f0_build_tables(img.data(), TVT_REG_CLASS, TVT_REG_CELL, TVT_REG_VERT, (const u16*)&img[TAB_REG_EDGE]);
}

// one pass of the device pipeline over the blocks listed in the level tables
void run_pipeline(vx_ctx* c, const ExecParams& p, u32 levels)
{
	const bool overlapped = !c->be.stage_timing_on();
	const bool ancestorsDone = c->be.ancestors_with_classification(p, levels);
	c->be.run_classify(p, overlapped && ancestorsDone); // k_run_head, stage_mark(1), k_classify: slot 0 = the head, slot 1 = the classify pass alone
	c->be.stage_mark(2);
	// (the HIP backend's classify pass also activates the ancestors of the blocks it finds; the hierarchy pass remains the
	// second step of an incremental run, and of the CPU emulation)
	if (overlapped) {
		// normal operation: independent stages overlap on side streams (per-stage times are then meaningless)
		if (!ancestorsDone) c->be.run_hierarchy(p, levels, true); // (carries the event that releases side stream A)
		c->be.run_overlapped_tail(p, levels);
		return;
	}
	if (!ancestorsDone) c->be.run_hierarchy(p, levels, false);
	c->be.stage_mark(3);
	c->stagedMain = false;
	if (c->be.single_stream(p, levels)) {
		// stage timing of the product path: [3] = k_main, [4] / [5] = what follows it for level 0 / for the levels >= 1 (what the
		// table-driven passes hand on, the upper capacity classes, levels beyond the lattice copies), [6] = nothing
		c->be.run_main_staged(p, levels);
		c->stagedMain = true;
		return;
	}
	for (u32 L = 1; L < levels; ++L) c->be.run_material(p, L);
	c->be.stage_mark(4);
	c->be.run_regular(p, levels);
	c->be.stage_mark(5);
	c->be.run_transition(p, levels);
	c->be.stage_mark(6);
}

void block_corners(const LevelDesc& d, u32 coordId, float mn[3], float mx[3])
{
	u32 bx, by, bz;
	block_coords(coordId, d.cnt, bx, by, bz);
	const float ext = (float)(d.mult * 16);
	mn[0] = (float)bx * ext; mn[1] = (float)bz * ext; mn[2] = (float)by * ext; // output is Y-up
	mx[0] = mn[0] + ext; mx[1] = mn[1] + ext; mx[2] = mn[2] + ext;
}

// host mirror of the output pools; between full runs the pools only grow, so only the new tail is copied
// ---- page-locked host copies of the pools ("arenas") --------------------------------------------------------------
// One block of page-locked memory: vertices first, indices behind them.  Arenas outlive contexts (a PolygonSurface of
// the drop-in API owns one) and are recycled through a process-wide list: page-locking some 100 MB costs far more than
// copying into them.
struct HostArena {
	void* mem = nullptr;
	PolyVertex* verts = nullptr;
	u32* idx = nullptr;
	size_t capVerts = 0, capIdx = 0;
	u32 haveVerts = 0, haveIdx = 0; // prefix of the pools it holds...
	uint64_t lineage = 0;           // ...of this vx_ctx::poolLineage
	void (*release_mem)(void*) = nullptr;
};

std::mutex g_arenaLock;
std::vector<HostArena*> g_arenaFree;
uint64_t g_lineage = 0;

uint64_t next_lineage()
{
	std::lock_guard<std::mutex> g(g_arenaLock);
	return ++g_lineage;
}

void arena_destroy(HostArena* a)
{
	if (!a) return;
	if (a->mem && a->release_mem) a->release_mem(a->mem);
	delete a;
}

void arena_recycle(HostArena* a)
{
	if (!a) return;
	a->haveVerts = a->haveIdx = 0; a->lineage = 0;
	HostArena* drop = nullptr;
	{
		std::lock_guard<std::mutex> g(g_arenaLock);
		g_arenaFree.push_back(a);
		if (g_arenaFree.size() > 4) { // keep the larger ones
			size_t k = 0;
			for (size_t i = 1; i < g_arenaFree.size(); ++i) if (g_arenaFree[i]->capVerts < g_arenaFree[k]->capVerts) k = i;
			drop = g_arenaFree[k];
			g_arenaFree.erase(g_arenaFree.begin() + k);
		}
	}
	arena_destroy(drop);
}

HostArena* arena_get(vx_ctx* c, size_t needVerts, size_t needIdx)
{
	{
		std::lock_guard<std::mutex> g(g_arenaLock);
		size_t best = g_arenaFree.size();
		for (size_t i = 0; i < g_arenaFree.size(); ++i) {
			const HostArena* a = g_arenaFree[i];
			if (a->capVerts < needVerts || a->capIdx < needIdx) continue;
			if (best == g_arenaFree.size() || a->capVerts < g_arenaFree[best]->capVerts) best = i;
		}
		if (best != g_arenaFree.size()) {
			HostArena* a = g_arenaFree[best];
			g_arenaFree.erase(g_arenaFree.begin() + best);
			return a;
		}
	}
	// some slack: the next run of a similar grid, or an incremental run's appended blocks, fit without a new arena
	const size_t capV = needVerts + needVerts / 8 + 4096, capI = needIdx + needIdx / 8 + 16384;
	const size_t vBytes = (capV * sizeof(PolyVertex) + 255) & ~size_t(255);
	std::unique_ptr<HostArena> a(new HostArena);
	a->mem = c->be.alloc_pinned(vBytes + capI * 4);
	if (!a->mem) return nullptr;
	a->release_mem = &Backend::release_pinned;
	a->verts = (PolyVertex*)a->mem;
	a->idx = (u32*)((char*)a->mem + vBytes);
	a->capVerts = capV; a->capIdx = capI;
	return a.release();
}

// make `a` hold the pools of the current surface; only what it does not hold yet travels
bool arena_fill(vx_ctx* c, HostArena*& a)
{
	if (a && (a->lineage != c->poolLineage || a->capVerts < c->poolVerts || a->capIdx < c->poolIdx)) {
		if (a->lineage == c->poolLineage && a->haveVerts <= c->poolVerts && a->haveIdx <= c->poolIdx) {
			// grown beyond its capacity: move what it holds to a larger one (host copy; no second trip over the bus)
			HostArena* b = arena_get(c, (size_t)c->poolVerts + c->poolVerts / 4, (size_t)c->poolIdx + c->poolIdx / 4);
			if (!b) return false;
			memcpy(b->verts, a->verts, (size_t)a->haveVerts * sizeof(PolyVertex));
			memcpy(b->idx, a->idx, (size_t)a->haveIdx * 4);
			b->haveVerts = a->haveVerts; b->haveIdx = a->haveIdx; b->lineage = a->lineage;
			arena_recycle(a);
			a = b;
		} else {
			a->haveVerts = a->haveIdx = 0;
			a->lineage = c->poolLineage;
			if (a->capVerts < c->poolVerts || a->capIdx < c->poolIdx) { arena_recycle(a); a = nullptr; }
		}
	}
	if (!a) {
		a = arena_get(c, c->poolVerts, c->poolIdx);
		if (!a) return false;
		a->lineage = c->poolLineage;
	}
	void* dst[2] = { a->verts + a->haveVerts, a->idx + a->haveIdx };
	const void* src[2] = { (const PolyVertex*)c->dVerts + a->haveVerts, (const u32*)c->dIdx + a->haveIdx };
	const size_t bytes[2] = { (size_t)(c->poolVerts - a->haveVerts) * sizeof(PolyVertex), (size_t)(c->poolIdx - a->haveIdx) * 4 };
	if (!c->be.d2h_bulk(dst, src, bytes, 2)) return false;
	a->haveVerts = c->poolVerts; a->haveIdx = c->poolIdx;
	return true;
}

bool fetch_pools(vx_ctx* c) { return arena_fill(c, c->hostArena); }

// grow the pools of an incremental run: what earlier runs wrote stays valid
bool grow_pools_keeping(vx_ctx* c, u32 needVerts, u32 needIdx)
{
	if (needVerts > c->vertCap) {
		void* nv = c->be.alloc((size_t)needVerts * sizeof(PolyVertex));
		if (!nv) return false;
		if (c->poolVerts && !c->be.d2d(nv, c->dVerts, (size_t)c->poolVerts * sizeof(PolyVertex))) { c->be.free(nv); return false; }
		c->be.free(c->dVerts);
		c->dVerts = nv; c->vertCap = needVerts;
	}
	if (needIdx > c->idxCap) {
		void* ni = c->be.alloc((size_t)needIdx * 4);
		if (!ni) return false;
		if (c->poolIdx && !c->be.d2d(ni, c->dIdx, (size_t)c->poolIdx * 4)) { c->be.free(ni); return false; }
		c->be.free(c->dIdx);
		c->dIdx = ni; c->idxCap = needIdx;
	}
	return true;
}

// Host copy of the block lists of a full run (every block with at least one regular vertex, in coordinate order; ids
// number ALL blocks of all levels in level-major order, TransVoxelImpl.cpp:395-401).  The lists themselves are part of
// the device run; this is only their download, done on first access.
// the blocks incremental runs dropped leave the lists (order kept: what the reference's erase leaves, TransVoxelImpl.cpp:443-450)
void purge_dead_blocks(vx_ctx* c)
{
	if (!c->deadBlocks) return;
	for (u32 L = 0; L < MAX_LEVELS; ++L) {
		std::vector<unsigned char>& dead = c->blockDead[L];
		if (dead.empty()) continue;
		std::vector<EmittedBlock>& list = c->blocks[L];
		size_t keep = 0;
		for (size_t i = 0; i < list.size(); ++i) if (!dead[i]) { if (keep != i) list[keep] = list[i]; ++keep; }
		list.resize(keep);
		dead.clear();
	}
	c->deadBlocks = 0;
}

// purge = false: the caller (an incremental run) only appends and marks; every reader of the lists takes them purged
int ensure_lists(vx_ctx* c, bool purge = true)
{
	if (c->listsReady) { if (purge) purge_dead_blocks(c); return VX_OK; }
	for (u32 L = 0; L < MAX_LEVELS; ++L) c->blockDead[L].clear();
	c->deadBlocks = 0;
	// the tables were written by the run itself (k_list_write): one copy per level, nothing to sort
	for (u32 L = 0; L < c->levelsRun; ++L) {
		const u32 count = c->hdr[HDR_LISTS + L];
		std::vector<EmittedBlock>& out = c->blocks[L];
		out.resize(count);
		if (count && !c->be.d2h_async(out.data(), c->lv[L].listed, (size_t)count * sizeof(EmittedBlock))) return fail(c, VX_ERR_DEVICE, "block lists: download failed: " + c->be.error());
	}
	if (!c->be.sync_ok()) return fail(c, VX_ERR_DEVICE, "block lists: download failed: " + c->be.error());
	c->listsReady = true;
	return VX_OK;
}

} // namespace

extern "C" {

const char* vx_backend(void) { return VX_BACKEND_NAME; }

int vx_device_count(int* count)
{
	if (!count) return VX_ERR_INVALID;
	*count = Backend::device_count();
	return *count > 0 ? VX_OK : VX_ERR_DEVICE;
}

int vx_ctx_create(int device_index, vx_ctx** out)
{
	if (!out) return VX_ERR_INVALID;
	*out = nullptr;
	vx_ctx* c = new vx_ctx;
	memset(c->stats, 0, sizeof(c->stats));
	memset(c->pyr, 0, sizeof(c->pyr));
	std::string e;
	if (!c->be.init(device_index, e)) { delete c; return VX_ERR_DEVICE; }
	c->hostTiming = getenv("VX_HOST_TIMING") != nullptr;
	if (const char* e = getenv("VX_POOL_SLACK")) c->poolSlack = std::max<u32>(64u, (u32)atoll(e));
	std::vector<u8> img;
	build_table_image(img);
	c->dTables = c->be.alloc(TAB_F0_BYTES);
	c->dLut = c->be.alloc(256 * 8);
	c->dHeader = c->be.alloc((HDR_WORDS + HDR_PARTIALS) * 4); // header + the block-class partial sums of k_run_head (one word per workgroup)
	c->headerSet[0] = c->dHeader;
	c->headerSet[1] = c->be.alloc((HDR_WORDS + HDR_PARTIALS) * 4);
	if (!c->dTables || !c->dLut || !c->dHeader || !c->headerSet[1] || !c->be.h2d(c->dTables, img.data(), TAB_F0_BYTES)) {
		vx_ctx_destroy(c);
		return VX_ERR_DEVICE;
	}
	// default material map: every id valid, all texture ids 0
	std::vector<u8> lut(256 * 8, 0);
	for (int i = 0; i < 256; ++i) lut[i * 8 + 6] = 1;
	c->be.h2d(c->dLut, lut.data(), lut.size());
	*out = c;
	return VX_OK;
}

void vx_ctx_destroy(vx_ctx* c)
{
	VX_ENTER(c);
	if (!c) return;
	c->be.sync();
	release_grid(c);
	free_bricks(c);
	free_level_tables(c);
	c->be.free(c->dVerts); c->be.free(c->dIdx);
	c->be.free(c->dVertsSpare); c->be.free(c->dIdxSpare); c->be.free(c->dSeg);
	c->be.free(c->dTables); c->be.free(c->dLut); c->be.free(c->headerSet[0]); c->be.free(c->headerSet[1]);
	c->be.free(c->dDirty); c->be.free(c->dWork); c->be.free(c->dGather);
	c->be.free_pinned(c->hRecs);
	c->be.free(c->dDirtyTicket);
	c->be.free(c->dBlobStage); c->be.free_pinned(c->hBlobStage); c->be.free(c->dWhereStage); c->be.free_pinned(c->hWhereStage);
	arena_recycle(c->hostArena);
	for (void* hb : c->haloBuf) c->be.free(hb);
	c->be.comm_destroy();
	c->be.free_pinned(c->hdrPinned);
	c->be.free(c->dScratch);
	c->be.shutdown();
	delete c;
}

const char* vx_last_error(const vx_ctx* c) { return c ? c->err.c_str() : "null context"; }

int vx_set_stream(vx_ctx* c, void* stream)
{
	VX_ENTER(c);
	if (!c) return VX_ERR_INVALID;
	c->be.set_stream(stream);
	return VX_OK;
}

int vx_grid_upload(vx_ctx* c, uint32_t n, const int8_t* dist, const uint8_t* mat, const uint8_t* blend, const uint8_t* flags)
{
	VX_ENTER(c);
	if (!c || !dist || !flags || n < 16 || (n & 15) || n > VX_MAX_GRID) return fail(c, VX_ERR_INVALID, "vx_grid_upload: n must be a multiple of 16 up to 2048, dist and empty_flags non-null");
	const size_t tot = (size_t)n * n * n, nb = (size_t)(n / 16) * (n / 16) * (n / 16);
	if (!(c->ownsGrid && c->n == n && c->zBegin == 0 && c->zEnd == n)) {
		release_grid(c);
		c->dDist = c->be.alloc(tot); c->dMat = c->be.alloc(tot); c->dBlend = c->be.alloc(tot); c->dFlags = c->be.alloc(nb);
		c->ownsGrid = true;
		if (!c->dDist || !c->dMat || !c->dBlend || !c->dFlags) { release_grid(c); return fail(c, VX_ERR_DEVICE, "vx_grid_upload: device allocation failed: " + c->be.error()); }
	}
	c->n = n; c->zBegin = 0; c->zEnd = n; c->distZ0 = 0; c->matZ0 = 0;
	c->yBegin = 0; c->yEnd = n; c->distY0 = 0; c->matY0 = 0; c->distRows = n; c->matRows = n;
	bool ok = c->be.h2d(c->dDist, dist, tot) && c->be.h2d(c->dFlags, flags, nb);
	ok = ok && (mat ? c->be.h2d(c->dMat, mat, tot) : c->be.fill(c->dMat, 0, tot));
	ok = ok && (blend ? c->be.h2d(c->dBlend, blend, tot) : c->be.fill(c->dBlend, 0, tot));
	c->haveSurface = false;
	c->bricksStale = true;
	return ok ? VX_OK : fail(c, VX_ERR_DEVICE, "vx_grid_upload: copy failed: " + c->be.error());
}

int vx_grid_create_heightmap(vx_ctx* c, uint32_t n, const int8_t* heightmap)
{
	VX_ENTER(c);
	if (!c || !heightmap || n < 16 || (n & 15) || n > VX_MAX_GRID) return fail(c, VX_ERR_INVALID, "vx_grid_create_heightmap: w must be a multiple of 16 up to 2048, heightmap non-null");
	const u32 nb = n / 16;
	const size_t blocks = (size_t)nb * nb * nb, tot = (size_t)n * n * n;
	if (!(c->ownsGrid && c->n == n && c->zBegin == 0 && c->zEnd == n && c->yBegin == 0 && c->yEnd == n)) {
		release_grid(c);
		c->dDist = c->be.alloc(tot); c->dMat = c->be.alloc(tot); c->dBlend = c->be.alloc(tot); c->dFlags = c->be.alloc(blocks);
		c->ownsGrid = true;
		if (!c->dDist || !c->dMat || !c->dBlend || !c->dFlags) { release_grid(c); return fail(c, VX_ERR_DEVICE, "vx_grid_create_heightmap: device allocation failed: " + c->be.error()); }
	}
	c->n = n; c->zBegin = 0; c->zEnd = n; c->distZ0 = 0; c->matZ0 = 0;
	c->yBegin = 0; c->yEnd = n; c->distY0 = 0; c->matY0 = 0; c->distRows = n; c->matRows = n;
	c->haveSurface = false;
	c->bricksStale = true;
	void* dMap = c->be.alloc((size_t)n * n);
	bool ok = dMap && c->be.h2d(dMap, heightmap, (size_t)n * n) && c->be.fill(c->dMat, 0, tot) && c->be.fill(c->dBlend, 0, tot);
	if (ok) {
		c->be.run_heightmap(resident_view(c), (const i8*)dMap, (u8*)c->dFlags);
		ok = c->be.sync_ok();
	}
	c->be.free(dMap);
	return ok ? VX_OK : fail(c, VX_ERR_DEVICE, "vx_grid_create_heightmap: device pass failed: " + c->be.error());
}

namespace {

// the synthetic terrain over everything resident (own layers + halo, clamped to the grid) + flags of the own blocks
int generate_terrain(vx_ctx* c, u32 seed, u32 style, const char* what)
{
	const u32 n = c->n, nb = n / 16;
	const int zb = (int)c->zBegin, ze = (int)c->zEnd, yb = (int)c->yBegin, ye = (int)c->yEnd;
	int dr[4], mr[4];
	resident_ranges(c, dr, mr);
	std::vector<u32> ids;
	for (u32 z = (u32)zb / 16; z < (u32)ze / 16; ++z) for (u32 y = (u32)yb / 16; y < (u32)ye / 16; ++y) for (u32 x = 0; x < nb; ++x) ids.push_back((z * nb + y) * nb + x);
	void* dHeight = c->be.alloc((size_t)n * n * 4);
	void* dIds = c->be.alloc(ids.size() * 4 + 16);
	bool ok = dHeight && dIds && c->be.h2d(dIds, ids.data(), ids.size() * 4);
	if (ok) {
		c->be.run_terrain(resident_view(c), seed, (float*)dHeight, dr, mr, (u8*)c->dFlags, (const u32*)dIds, (u32)ids.size(), style);
		ok = c->be.sync_ok();
	}
	c->be.free(dHeight); c->be.free(dIds);
	c->haveSurface = false;
	c->bricksStale = true;
	return ok ? VX_OK : fail(c, VX_ERR_DEVICE, std::string(what) + ": device pass failed: " + c->be.error());
}

} // namespace

int vx_grid_create_terrain(vx_ctx* c, uint32_t n, uint32_t seed) { return vx_grid_create_terrain_ex(c, n, seed, 0); }

int vx_grid_create_terrain_ex(vx_ctx* c, uint32_t n, uint32_t seed, uint32_t style)
{
	VX_ENTER(c);
	if (!c || n < 16 || (n & 15) || n > VX_MAX_GRID) return fail(c, VX_ERR_INVALID, "vx_grid_create_terrain: n must be a multiple of 16 up to 2048");
	const size_t tot = (size_t)n * n * n, blocks = (size_t)(n / 16) * (n / 16) * (n / 16);
	if (!(c->ownsGrid && c->n == n && c->zBegin == 0 && c->zEnd == n && c->yBegin == 0 && c->yEnd == n)) {
		release_grid(c);
		c->dDist = c->be.alloc(tot); c->dMat = c->be.alloc(tot); c->dBlend = c->be.alloc(tot); c->dFlags = c->be.alloc(blocks);
		c->ownsGrid = true;
		if (!c->dDist || !c->dMat || !c->dBlend || !c->dFlags) { release_grid(c); return fail(c, VX_ERR_DEVICE, "vx_grid_create_terrain: device allocation failed: " + c->be.error()); }
	}
	c->n = n; c->zBegin = 0; c->zEnd = n; c->distZ0 = 0; c->matZ0 = 0;
	c->yBegin = 0; c->yEnd = n; c->distY0 = 0; c->matY0 = 0; c->distRows = n; c->matRows = n;
	return generate_terrain(c, seed, style, "vx_grid_create_terrain");
}

int vx_grid_fill_terrain(vx_ctx* c, uint32_t seed)
{
	VX_ENTER(c);
	if (!c || !c->n || !c->dDist || !c->slabAxis) return fail(c, VX_ERR_INVALID, "vx_grid_fill_terrain: needs a slab attached with vx_grid_attach / vx_grid_attach_y");
	return generate_terrain(c, seed, 0, "vx_grid_fill_terrain");
}

int vx_grid_upload_packed(vx_ctx* c, const void* blobPtr, uint64_t size)
{
	VX_ENTER(c);
	if (!c || !blobPtr || size < 16) return fail(c, VX_ERR_INVALID, "vx_grid_upload_packed: null or truncated blob");
	const u8* blob = (const u8*)blobPtr;
	auto rd32 = [&](uint64_t off) { u32 v; memcpy(&v, blob + off, 4); return v; };
	if (rd32(0) != 1) return fail(c, VX_ERR_INVALID, "vx_grid_upload_packed: not a version-1 grid file");
	const u32 n = rd32(4);
	if (n < 16 || (n & 15) || n > VX_MAX_GRID || rd32(8) != n || rd32(12) != n) return fail(c, VX_ERR_INVALID, "vx_grid_upload_packed: the grid must be a cube with an edge that is a multiple of 16 up to 2048");
	const u32 nb = n / 16;
	const size_t blocks = (size_t)nb * nb * nb, tot = (size_t)n * n * n;
	const uint64_t tableEnd = 16 + (uint64_t)blocks * 12;
	if (size < tableEnd) return fail(c, VX_ERR_INVALID, "vx_grid_upload_packed: truncated size table");
	// The file travels through a page-locked staging buffer of the context (a copy from pageable memory is staged by the
	// runtime anyway, by one thread, piece by piece): a few host threads fill it in pieces, and every piece goes on its way to
	// the device as soon as it and its predecessors are complete, while the calling thread walks the size table (per block:
	// offset of its record {flags, 3 streams} and the three stream sizes).  Device-side staging is kept across calls as well.
	// Nothing of the context's grid is touched before the file has been found sound.
	if (size + 16 > c->blobCap) {
		c->be.free(c->dBlobStage); c->be.free_pinned(c->hBlobStage);
		c->blobCap = size + size / 8 + 4096;
		c->dBlobStage = c->be.alloc(c->blobCap);
		c->hBlobStage = c->be.alloc_pinned(c->blobCap);
		if (!c->dBlobStage || !c->hBlobStage) { c->blobCap = 0; return fail(c, VX_ERR_DEVICE, "vx_grid_upload_packed: staging allocation failed: " + c->be.error()); }
	}
	if (blocks * 16 > c->whereCap) {
		c->be.free(c->dWhereStage); c->be.free_pinned(c->hWhereStage);
		c->whereCap = blocks * 16;
		c->dWhereStage = c->be.alloc(c->whereCap);
		c->hWhereStage = c->be.alloc_pinned(c->whereCap);
		if (!c->dWhereStage || !c->hWhereStage) { c->whereCap = 0; return fail(c, VX_ERR_DEVICE, "vx_grid_upload_packed: staging allocation failed: " + c->be.error()); }
	}
	const auto tp0 = std::chrono::steady_clock::now();
	enum { PIECES = 16 };
	const uint64_t step = ((size + PIECES - 1) / PIECES + 255) & ~uint64_t(255);
	std::atomic<unsigned> next(0);
	std::atomic<unsigned char> done[PIECES];
	for (auto& d : done) d.store(0);
	auto filler = [&]() {
		for (;;) {
			const unsigned k = next.fetch_add(1);
			if (k >= (unsigned)PIECES) return;
			const uint64_t at = (uint64_t)k * step;
			if (at < size) memcpy((u8*)c->hBlobStage + at, blob + at, std::min(step, size - at));
			done[k].store(1, std::memory_order_release);
		}
	};
	std::thread helpers[3];
	for (std::thread& t : helpers) t = std::thread(filler);
	uint64_t* where = (uint64_t*)c->hWhereStage;
	uint64_t off = tableEnd;
	bool corrupt = false;
	for (size_t i = 0; i < blocks; ++i) {
		const u32 sd = rd32(16 + i * 12), sm = rd32(16 + i * 12 + 4), sb = rd32(16 + i * 12 + 8);
		if (sd > 4096 || sm > 4096 || sb > 4096 || (sd & 1) || (sm & 1) || (sb & 1)) { corrupt = true; break; }
		where[i * 2] = off;
		where[i * 2 + 1] = (uint64_t)sd | ((uint64_t)sm << 16) | ((uint64_t)sb << 32);
		off += 4 + (uint64_t)sd + sm + sb;
		if (off > size) { corrupt = true; break; }
	}
	if (corrupt) {
		for (std::thread& t : helpers) t.join();
		return fail(c, VX_ERR_INVALID, "vx_grid_upload_packed: corrupt or truncated block data");
	}
	if (!(c->ownsGrid && c->n == n && c->zBegin == 0 && c->zEnd == n)) {
		release_grid(c);
		c->dDist = c->be.alloc(tot); c->dMat = c->be.alloc(tot); c->dBlend = c->be.alloc(tot); c->dFlags = c->be.alloc(blocks);
		c->ownsGrid = true;
		if (!c->dDist || !c->dMat || !c->dBlend || !c->dFlags) {
			for (std::thread& t : helpers) t.join();
			release_grid(c);
			return fail(c, VX_ERR_DEVICE, "vx_grid_upload_packed: device allocation failed: " + c->be.error());
		}
	}
	c->n = n; c->zBegin = 0; c->zEnd = n; c->distZ0 = 0; c->matZ0 = 0;
	c->yBegin = 0; c->yEnd = n; c->distY0 = 0; c->matY0 = 0; c->distRows = n; c->matRows = n;
	c->haveSurface = false;
	c->bricksStale = true;
	const auto tp1 = std::chrono::steady_clock::now();
	bool ok = c->be.h2d_async(c->dWhereStage, c->hWhereStage, blocks * 16);
	for (unsigned k = 0; k < (unsigned)PIECES; ++k) {
		if (!done[k].load(std::memory_order_acquire)) filler(); // (lend a hand while there is something to wait for)
		while (!done[k].load(std::memory_order_acquire)) std::this_thread::yield();
		const uint64_t at = (uint64_t)k * step;
		if (ok && at < off) ok = c->be.h2d_async((u8*)c->dBlobStage + at, (const u8*)c->hBlobStage + at, std::min(step, off - at));
	}
	for (std::thread& t : helpers) t.join();
	const auto tp2 = std::chrono::steady_clock::now();
	if (ok && c->hostTiming) ok = c->be.sync_ok(); // (so that the copies and the decode are timed apart)
	const auto tp3 = std::chrono::steady_clock::now();
	if (ok) {
		c->be.run_decode_grid((const u8*)c->dBlobStage, (const uint64_t*)c->dWhereStage, n, (i8*)c->dDist, (u8*)c->dMat, (u8*)c->dBlend, (u8*)c->dFlags);
		ok = c->be.sync_ok();
	}
	if (c->hostTiming) {
		const auto tp4 = std::chrono::steady_clock::now();
		auto ms = [](std::chrono::steady_clock::time_point a, std::chrono::steady_clock::time_point b) { return std::chrono::duration<double, std::milli>(b - a).count(); };
		fprintf(stderr, "[vx host] upload_packed: size table + start of staging %.2f ms, rest of staging + copies queued %.2f ms, copies done %.2f ms, decode %.2f ms\n", ms(tp0, tp1), ms(tp1, tp2), ms(tp2, tp3), ms(tp3, tp4));
	}
	return ok ? VX_OK : fail(c, VX_ERR_DEVICE, "vx_grid_upload_packed: device decode failed: " + c->be.error());
}

int vx_grid_pack(vx_ctx* c, void* out, uint64_t capacity, uint64_t* size)
{
	VX_ENTER(c);
	if (!c || !size) return fail(c, VX_ERR_INVALID, "vx_grid_pack: null argument");
	if (!c->n || !c->dDist || (c->zBegin != 0 || c->zEnd != c->n || c->yBegin != 0 || c->yEnd != c->n)) return fail(c, VX_ERR_INVALID, "vx_grid_pack: needs a whole grid resident");
	const u32 n = c->n, nb = n / 16;
	const size_t blocks = (size_t)nb * nb * nb;
	const GridView g = resident_view(c);
	// pass 1: stream sizes + flags of every block
	void* dMeta = c->be.alloc(blocks * 16);
	std::vector<u32> meta(blocks * 4);
	bool ok = dMeta != nullptr;
	if (ok) {
		c->be.run_encode_grid(g, (u32*)dMeta, nullptr, nullptr);
		ok = c->be.d2h(meta.data(), dMeta, blocks * 16);
	}
	const uint64_t tableEnd = 16 + (uint64_t)blocks * 12;
	std::vector<uint64_t> where(blocks);
	uint64_t off = 0; // relative to the end of the size table
	for (size_t i = 0; ok && i < blocks; ++i) { where[i] = off; off += 4 + (uint64_t)meta[i * 4] + meta[i * 4 + 1] + meta[i * 4 + 2]; }
	*size = tableEnd + off;
	if (!ok) { c->be.free(dMeta); return fail(c, VX_ERR_DEVICE, "vx_grid_pack: device pass failed: " + c->be.error()); }
	if (!out) { c->be.free(dMeta); return VX_OK; }
	if (capacity < *size) { c->be.free(dMeta); return fail(c, VX_ERR_INVALID, "vx_grid_pack: output buffer too small"); }
	// pass 2: the block records {flags, distance, material, blend streams} at their offsets
	void* dWhere = c->be.alloc(blocks * 8);
	void* dBlob = c->be.alloc(off + 16);
	ok = dWhere && dBlob && c->be.h2d(dWhere, where.data(), blocks * 8);
	if (ok) {
		c->be.run_encode_grid(g, (u32*)dMeta, (const uint64_t*)dWhere, (u8*)dBlob);
		ok = c->be.d2h((u8*)out + tableEnd, dBlob, off);
	}
	c->be.free(dMeta); c->be.free(dWhere); c->be.free(dBlob);
	if (!ok) return fail(c, VX_ERR_DEVICE, "vx_grid_pack: device pass failed: " + c->be.error());
	u8* o = (u8*)out;
	const u32 header[4] = { 1u, n, n, n };
	memcpy(o, header, 16);
	for (size_t i = 0; i < blocks; ++i) memcpy(o + 16 + i * 12, &meta[i * 4], 12);
	return VX_OK;
}

int vx_grid_read_block(vx_ctx* c, uint32_t id, int8_t* dist, uint8_t* mat, uint8_t* blend, uint8_t* emptyFlag)
{
	VX_ENTER(c);
	if (!c || !c->n || !c->dDist) return fail(c, VX_ERR_INVALID, "vx_grid_read_block: no grid resident");
	const u32 n = c->n, nb = n / 16;
	if (id >= nb * nb * nb) return fail(c, VX_ERR_INVALID, "vx_grid_read_block: block id out of range");
	const u32 bx = id % nb, by = (id / nb) % nb, bz = id / (nb * nb);
	if (bz * 16 < c->zBegin || bz * 16 >= c->zEnd || by * 16 < c->yBegin || by * 16 >= c->yEnd) return fail(c, VX_ERR_INVALID, "vx_grid_read_block: block outside the resident slab");
	bool ok = true;
	for (u32 z = 0; z < 16 && ok; ++z)
	for (u32 y = 0; y < 16 && ok; ++y) {
		const size_t dst = (size_t)z * 256 + y * 16;
		const size_t srcD = ((size_t)((int)(bz * 16 + z) - c->distZ0) * c->distRows + ((int)(by * 16 + y) - c->distY0)) * n + bx * 16;
		const size_t srcM = ((size_t)((int)(bz * 16 + z) - c->matZ0) * c->matRows + ((int)(by * 16 + y) - c->matY0)) * n + bx * 16;
		if (dist) ok = ok && c->be.d2h_async(dist + dst, (const u8*)c->dDist + srcD, 16);
		if (mat) ok = ok && c->be.d2h_async(mat + dst, (const u8*)c->dMat + srcM, 16);
		if (blend) ok = ok && c->be.d2h_async(blend + dst, (const u8*)c->dBlend + srcM, 16);
	}
	if (emptyFlag) ok = ok && c->be.d2h_async(emptyFlag, (const u8*)c->dFlags + id, 1);
	ok = ok && c->be.sync_ok();
	return ok ? VX_OK : fail(c, VX_ERR_DEVICE, "vx_grid_read_block: copy failed: " + c->be.error());
}

int vx_grid_attach(vx_ctx* c, uint32_t n, uint32_t z_begin, uint32_t z_end, const void* d_dist, int32_t dist_z0,
                   const void* d_mat, const void* d_blend, int32_t mat_z0, const void* d_flags)
{
	VX_ENTER(c);
	if (!c || !d_dist || !d_mat || !d_blend || !d_flags || n < 16 || (n & 15) || n > VX_MAX_GRID || z_begin >= z_end || z_end > n || (z_begin & 15) || (z_end & 15))
		return fail(c, VX_ERR_INVALID, "vx_grid_attach: bad arguments (slab bounds must be multiples of 16)");
	release_grid(c);
	c->n = n; c->zBegin = z_begin; c->zEnd = z_end;
	c->dDist = (void*)d_dist; c->dMat = (void*)d_mat; c->dBlend = (void*)d_blend; c->dFlags = (void*)d_flags;
	c->distZ0 = dist_z0; c->matZ0 = mat_z0;
	c->yBegin = 0; c->yEnd = n; c->distY0 = 0; c->matY0 = 0; c->distRows = n; c->matRows = n;
	c->haveSurface = false;
	c->bricksStale = true;
	c->slabAxis = 1;
	return VX_OK;
}

int vx_grid_attach_y(vx_ctx* c, uint32_t n, uint32_t y_begin, uint32_t y_end, const void* d_dist, int32_t dist_y0, uint32_t dist_rows,
                     const void* d_mat, const void* d_blend, int32_t mat_y0, uint32_t mat_rows, const void* d_flags)
{
	VX_ENTER(c);
	if (!c || !d_dist || !d_mat || !d_blend || !d_flags || n < 16 || (n & 15) || n > VX_MAX_GRID || y_begin >= y_end || y_end > n || (y_begin & 15) || (y_end & 15) || !dist_rows || !mat_rows)
		return fail(c, VX_ERR_INVALID, "vx_grid_attach_y: bad arguments (slab bounds must be multiples of 16)");
	release_grid(c);
	c->n = n; c->zBegin = 0; c->zEnd = n; c->distZ0 = 0; c->matZ0 = 0;
	c->yBegin = y_begin; c->yEnd = y_end;
	c->dDist = (void*)d_dist; c->dMat = (void*)d_mat; c->dBlend = (void*)d_blend; c->dFlags = (void*)d_flags;
	c->distY0 = dist_y0; c->matY0 = mat_y0; c->distRows = dist_rows; c->matRows = mat_rows;
	c->haveSurface = false;
	c->bricksStale = true;
	c->slabAxis = 2;
	return VX_OK;
}

int vx_grid_upload_slab_y(vx_ctx* c, uint32_t n, uint32_t y_begin, uint32_t y_end, const int8_t* dist, const uint8_t* mat, const uint8_t* blend, const uint8_t* flags)
{
	VX_ENTER(c);
	if (!c || !dist || !flags || n < 16 || (n & 15) || n > VX_MAX_GRID || y_begin >= y_end || y_end > n || (y_begin & 15) || (y_end & 15))
		return fail(c, VX_ERR_INVALID, "vx_grid_upload_slab_y: bad arguments (slab bounds must be multiples of 16)");
	const u32 rows = y_end - y_begin, dRows = rows + 3, mRows = rows + 1;
	const size_t nb = (size_t)(n / 16) * (n / 16) * (n / 16);
	const bool reuse = c->ownsSlab && c->slabAxis == 2 && c->n == n && c->yBegin == y_begin && c->yEnd == y_end;
	if (!reuse) {
		release_grid(c);
		c->dDist = c->be.alloc((size_t)n * dRows * n); c->dMat = c->be.alloc((size_t)n * mRows * n); c->dBlend = c->be.alloc((size_t)n * mRows * n); c->dFlags = c->be.alloc(nb);
		c->ownsSlab = true;
		if (!c->dDist || !c->dMat || !c->dBlend || !c->dFlags) { release_grid(c); return fail(c, VX_ERR_DEVICE, "vx_grid_upload_slab_y: device allocation failed: " + c->be.error()); }
	}
	c->n = n; c->zBegin = 0; c->zEnd = n; c->distZ0 = 0; c->matZ0 = 0;
	c->yBegin = y_begin; c->yEnd = y_end;
	c->distY0 = (int)y_begin - 1; c->matY0 = (int)y_begin; c->distRows = dRows; c->matRows = mRows;
	c->slabAxis = 2;
	c->haveSurface = false;
	c->bricksStale = true;
	// rows [a, b) of every plane: (b - a) * n contiguous bytes per plane, n planes, n * n bytes apart in the host array
	auto rows_of = [&](void* dDst, u32 dstRows, int dstRow0, const void* src, int a, int b) -> bool {
		a = std::max(a, 0); b = std::min(b, (int)n);
		if (!src) return c->be.fill(dDst, 0, (size_t)n * dstRows * n);
		return c->be.h2d_2d((u8*)dDst + (size_t)(a - dstRow0) * n, (size_t)dstRows * n, (const u8*)src + (size_t)a * n, (size_t)n * n, (size_t)(b - a) * n, n);
	};
	bool ok = rows_of(c->dDist, dRows, c->distY0, dist, (int)y_begin - 1, (int)y_end + 2)
	       && rows_of(c->dMat, mRows, c->matY0, mat, (int)y_begin, (int)y_end + 1)
	       && rows_of(c->dBlend, mRows, c->matY0, blend, (int)y_begin, (int)y_end + 1)
	       && c->be.h2d(c->dFlags, flags, nb);
	return ok ? VX_OK : fail(c, VX_ERR_DEVICE, "vx_grid_upload_slab_y: copy failed: " + c->be.error());
}

int vx_grid_invalidate(vx_ctx* c)
{
	VX_ENTER(c);
	if (!c || !c->n || !c->dDist) return fail(c, VX_ERR_INVALID, "vx_grid_invalidate: no grid resident");
	// the caller rewrote attached memory in place: the brick mirrors, lattice copies and sign summaries are rebuilt by the
	// next polygonization, and whatever surface the context holds no longer describes the grid
	c->bricksStale = true;
	c->haveSurface = false;
	return VX_OK;
}

int vx_ctx_forget_hints(vx_ctx* c)
{
	VX_ENTER(c);
	if (!c) return VX_ERR_INVALID;
	c->largeHint = true;
	c->dirtyLargeHint = c->dirtyLarge0Hint = false;
	c->editRoomFailed = false;
	c->be.upperItemsHint = 0;
	c->be.slowHint[0] = c->be.slowHint[1] = ~0u;
	return VX_OK;
}

namespace {

// The four halo messages of an attached slab (see include/voxels_hip.h): which layers of which field go where.
// Layer = z-plane (slabs of planes) or y-row of every plane (slabs of rows).
struct HaloPlan {
	HaloMove sendLo, sendHi, recvLo, recvHi; // below = towards smaller coordinates
	bool hasLo, hasHi;
};

bool halo_plan(vx_ctx* c, HaloPlan& pl, std::string& why)
{
	if (!c->n || !c->dDist || c->ownsGrid || !c->slabAxis) { why = "needs a slab attached with vx_grid_attach / vx_grid_attach_y"; return false; }
	const bool alongY = c->slabAxis == 2;
	const int b = (int)(alongY ? c->yBegin : c->zBegin), e = (int)(alongY ? c->yEnd : c->zEnd);
	const int n = (int)c->n, cnt = n / 16;
	const int dOrigin = alongY ? c->distY0 : c->distZ0, mOrigin = alongY ? c->matY0 : c->matZ0;
	if (dOrigin != b - 1 || mOrigin != b) { why = "the attached buffers must start one distance layer below the slab and at the slab's first material layer"; return false; }
	if (alongY && (c->distRows != (u32)(e - b) + 3 || c->matRows != (u32)(e - b) + 1)) { why = "y-slab buffers must hold the slab's rows plus 3 (distances) / plus 1 (materials)"; return false; }
	memset(&pl, 0, sizeof(pl));
	pl.hasLo = b > 0; pl.hasHi = e < n;
	auto field = [&](void* base, int first, int layers, int origin, u32 rowsPerPlane) {
		HaloPiece p;
		p.field = (u8*)base; p.stagingOffset = 0; p.firstLayer = first; p.layers = layers; p.origin = origin;
		p.strideLayer = alongY ? 1u : rowsPerPlane; p.strideA = alongY ? rowsPerPlane : 1u;
		p.rowBytes = (u32)n; p.rows = (u32)n;
		return p;
	};
	auto flags = [&](int blockLayer) {
		HaloPiece p;
		p.field = (u8*)c->dFlags; p.stagingOffset = 0; p.firstLayer = blockLayer; p.layers = 1; p.origin = 0;
		p.strideLayer = alongY ? 1u : (u32)cnt; p.strideA = alongY ? (u32)cnt : 1u;
		p.rowBytes = (u32)cnt; p.rows = (u32)cnt;
		return p;
	};
	auto pack = [&](HaloMove& mv, std::initializer_list<HaloPiece> pieces, u32 unpack, void* staging) {
		u32 off = 0;
		mv.count = 0;
		for (HaloPiece p : pieces) { p.stagingOffset = off; off += (u32)halo_piece_bytes(p); mv.piece[mv.count++] = p; }
		mv.unpack = unpack; mv.staging = (u8*)staging;
		return (size_t)off;
	};
	const u32 dRows = alongY ? c->distRows : (u32)n, mRows = alongY ? c->matRows : (u32)n;
	const size_t loBytes = 4 * (size_t)n * n + (size_t)cnt * cnt, hiBytes = (size_t)n * n + (size_t)cnt * cnt;
	const size_t need[4] = { pl.hasLo ? loBytes : 0, pl.hasHi ? hiBytes : 0, pl.hasLo ? hiBytes : 0, pl.hasHi ? loBytes : 0 };
	for (int i = 0; i < 4; ++i) if (need[i] > c->haloCap[i]) {
		c->be.free(c->haloBuf[i]);
		c->haloBuf[i] = c->be.alloc(need[i]);
		c->haloCap[i] = c->haloBuf[i] ? need[i] : 0;
		if (!c->haloBuf[i]) { why = "staging allocation failed"; return false; }
	}
	if (pl.hasLo) {
		pack(pl.sendLo, { field(c->dDist, b, 2, dOrigin, dRows), field(c->dMat, b, 1, mOrigin, mRows), field(c->dBlend, b, 1, mOrigin, mRows), flags(b / 16) }, 0, c->haloBuf[0]);
		pack(pl.recvLo, { field(c->dDist, b - 1, 1, dOrigin, dRows), flags(b / 16 - 1) }, 1, c->haloBuf[2]);
	}
	if (pl.hasHi) {
		pack(pl.sendHi, { field(c->dDist, e - 1, 1, dOrigin, dRows), flags(e / 16 - 1) }, 0, c->haloBuf[1]);
		pack(pl.recvHi, { field(c->dDist, e, 2, dOrigin, dRows), field(c->dMat, e, 1, mOrigin, mRows), field(c->dBlend, e, 1, mOrigin, mRows), flags(e / 16) }, 1, c->haloBuf[3]);
	}
	return true;
}

// the view an unpack writes through: with the brick mirrors while they are current (they then follow the received rows),
// without them while they wait for a full copy anyway
GridView halo_view(const vx_ctx* c)
{
	GridView g = resident_view(c);
	if (!c->be.wants_bricks() || c->bricksStale || !c->dBrick[0]) { g.bDist = nullptr; g.bMat = nullptr; g.bBlend = nullptr; }
	return g;
}

// Behind an unpack that found the mirrors current: the halo block layers once more through the mirror pass.  The unpack has
// written the received rows into the bricks and the lattice copies already; what it cannot keep current is the layers'
// sign summaries (blockSign), which k_run_head reads for the blocks of the last owned layer - exact for the plane that is
// resident since round 4 (rebrick_row), so they have to follow the rows.  Two block layers: ~20 us at 1024^3.
void refresh_halo_layers(vx_ctx* c)
{
#if defined(VX_NO_HALO_REFRESH) // (tools builds: shows that tests/test_gpu_parity.py::test_hip_halo_exchange_after_a_neighbour_changed needs it)
	return;
#endif
	if (!c->be.wants_bricks() || c->bricksStale || !c->dBrick[0] || c->slabAxis == 0) return;
	int dr[4], mr[4];
	resident_ranges(c, dr, mr);
	const bool alongY = c->slabAxis == 2;
	const int yb0 = (int)c->brickYb0, yb1 = yb0 + (int)c->brickRowsY, zb0 = (int)c->brickZb0, zb1 = zb0 + (int)c->brickPlanesZ;
	const int own0 = (int)(alongY ? c->yBegin : c->zBegin) / 16, own1 = (int)(alongY ? c->yEnd : c->zEnd) / 16;
	auto layers = [&](int l0, int l1) {
		if (l1 <= l0) return;
		const int box[4] = { alongY ? l0 : yb0, alongY ? l1 : yb1, alongY ? zb0 : l0, alongY ? zb1 : l1 };
		c->be.run_rebrick(resident_view(c), dr, mr, mirror_state(c), box, nullptr, 0);
	};
	layers(alongY ? yb0 : zb0, own0);
	layers(own1, alongY ? yb1 : zb1);
}

size_t halo_move_bytes(const HaloMove& mv)
{
	size_t s = 0;
	for (u32 i = 0; i < mv.count; ++i) s += halo_piece_bytes(mv.piece[i]);
	return s;
}

} // namespace

int vx_comm_unique_id(void* id)
{
	if (!id) return VX_ERR_INVALID;
	return Backend::comm_unique_id(id) ? VX_OK : VX_ERR_DEVICE;
}

int vx_comm_init(vx_ctx* c, int nranks, int rank, const void* id)
{
	VX_ENTER(c);
	if (!c || !id || nranks < 1 || rank < 0 || rank >= nranks) return fail(c, VX_ERR_INVALID, "vx_comm_init: bad arguments");
	if (!c->be.comm_init(nranks, rank, id)) return fail(c, VX_ERR_DEVICE, "vx_comm_init: " + c->be.error());
	c->commRanks = nranks; c->commRank = rank;
	return VX_OK;
}

int vx_comm_destroy(vx_ctx* c)
{
	VX_ENTER(c);
	if (!c) return VX_ERR_INVALID;
	c->be.comm_destroy();
	c->commRanks = 0;
	return VX_OK;
}

int vx_halo_exchange(vx_ctx* c)
{
	VX_ENTER(c);
	if (!c) return VX_ERR_INVALID;
	if (!c->commRanks) return fail(c, VX_ERR_INVALID, "vx_halo_exchange: call vx_comm_init first");
	HaloPlan pl;
	std::string why;
	if (!halo_plan(c, pl, why)) return fail(c, VX_ERR_INVALID, "vx_halo_exchange: " + why);
	// rank r owns the r-th slab: a rank that posts no message towards a neighbour that expects one (or the reverse) would
	// leave its peer blocked in the grouped send / receive for ever
	if (pl.hasLo != (c->commRank > 0) || pl.hasHi != (c->commRank + 1 < c->commRanks)) return fail(c, VX_ERR_INVALID, "vx_halo_exchange: rank r must own the r-th slab (first rank starts at 0, last rank ends at n, no other rank does)");
	// pack -> one grouped send/recv batch -> unpack, all queued on the context's stream: nothing waits on the host
	const bool alongY = c->slabAxis == 2;
	c->be.run_halo_moves(pl.hasLo ? &pl.sendLo : nullptr, pl.hasHi ? &pl.sendHi : nullptr, halo_view(c), mirror_state(c), alongY);
	const bool ok = c->be.comm_exchange(pl.hasLo ? c->commRank - 1 : -1, pl.hasLo ? c->haloBuf[0] : nullptr, pl.hasLo ? halo_move_bytes(pl.sendLo) : 0,
	                                    pl.hasLo ? c->haloBuf[2] : nullptr, pl.hasLo ? halo_move_bytes(pl.recvLo) : 0,
	                                    pl.hasHi ? c->commRank + 1 : -1, pl.hasHi ? c->haloBuf[1] : nullptr, pl.hasHi ? halo_move_bytes(pl.sendHi) : 0,
	                                    pl.hasHi ? c->haloBuf[3] : nullptr, pl.hasHi ? halo_move_bytes(pl.recvHi) : 0);
	if (!ok) return fail(c, VX_ERR_DEVICE, "vx_halo_exchange: " + c->be.error());
	c->be.run_halo_moves(pl.hasLo ? &pl.recvLo : nullptr, pl.hasHi ? &pl.recvHi : nullptr, halo_view(c), mirror_state(c), alongY);
	refresh_halo_layers(c);
	c->haveSurface = false;
	return VX_OK;
}

int vx_halo_exchange_group(vx_ctx* const* ctxs, int count)
{
	if (!ctxs || count < 1) return VX_ERR_INVALID;
	std::vector<HaloPlan> plans((size_t)count);
	std::string why;
	for (int i = 0; i < count; ++i) {
		vx_ctx* c = ctxs[i];
		VX_ENTER(c);
		if (!c) return VX_ERR_INVALID;
		if (!halo_plan(c, plans[(size_t)i], why)) return fail(c, VX_ERR_INVALID, "vx_halo_exchange_group: " + why);
		if (plans[(size_t)i].hasLo != (i > 0) || plans[(size_t)i].hasHi != (i + 1 < count)) return fail(c, VX_ERR_INVALID, "vx_halo_exchange_group: contexts must be the slabs of one grid in order");
		c->be.run_halo_moves(plans[(size_t)i].hasLo ? &plans[(size_t)i].sendLo : nullptr, plans[(size_t)i].hasHi ? &plans[(size_t)i].sendHi : nullptr, halo_view(c), mirror_state(c), c->slabAxis == 2);
	}
	for (int i = 0; i < count; ++i) { VX_ENTER(ctxs[i]); if (!ctxs[i]->be.sync_ok()) return fail(ctxs[i], VX_ERR_DEVICE, "vx_halo_exchange_group: pack failed: " + ctxs[i]->be.error()); }
	for (int i = 0; i + 1 < count; ++i) {
		vx_ctx *lo = ctxs[i], *hi = ctxs[i + 1];
		// upward: lo's last layer -> hi's layer below; downward: hi's first layers -> lo's layers above
		VX_ENTER(hi);
		if (!hi->be.copy_from_peer(hi->haloBuf[2], lo->be, lo->haloBuf[1], halo_move_bytes(plans[(size_t)i].sendHi))) return fail(hi, VX_ERR_DEVICE, "vx_halo_exchange_group: peer copy failed: " + hi->be.error());
		VX_ENTER(lo);
		if (!lo->be.copy_from_peer(lo->haloBuf[3], hi->be, hi->haloBuf[0], halo_move_bytes(plans[(size_t)i + 1].sendLo))) return fail(lo, VX_ERR_DEVICE, "vx_halo_exchange_group: peer copy failed: " + lo->be.error());
	}
	for (int i = 0; i < count; ++i) {
		vx_ctx* c = ctxs[i];
		VX_ENTER(c);
		if (!c->be.sync_ok()) return fail(c, VX_ERR_DEVICE, "vx_halo_exchange_group: copy failed: " + c->be.error());
		c->be.run_halo_moves(plans[(size_t)i].hasLo ? &plans[(size_t)i].recvLo : nullptr, plans[(size_t)i].hasHi ? &plans[(size_t)i].recvHi : nullptr, halo_view(c), mirror_state(c), c->slabAxis == 2);
		refresh_halo_layers(c);
		c->haveSurface = false;
	}
	return VX_OK;
}

int vx_grid_update_blocks(vx_ctx* c, uint32_t count, const uint32_t* ids, const int8_t* dist, const uint8_t* mat,
                          const uint8_t* blend, const uint8_t* flags)
{
	VX_ENTER(c);
	if (!c || !c->ownsGrid || !c->n) return fail(c, VX_ERR_INVALID, "vx_grid_update_blocks: needs a grid uploaded with vx_grid_upload");
	const u32 n = c->n, nb = n / 16;
	bool ok = true;
	for (u32 i = 0; i < count; ++i) if (ids[i] >= nb * nb * nb) return fail(c, VX_ERR_INVALID, "vx_grid_update_blocks: block id out of range");
	if (count && (dist || mat || blend)) {
		// the blocks travel as they are (4096 contiguous bytes each) and are scattered into the dense fields on the device
		const size_t bytes = (size_t)count * 4096;
		void* dIds = c->be.alloc((size_t)count * 4);
		void* stage[3] = { dist ? c->be.alloc(bytes) : nullptr, mat ? c->be.alloc(bytes) : nullptr, blend ? c->be.alloc(bytes) : nullptr };
		const void* src[3] = { dist, mat, blend };
		ok = dIds && c->be.h2d(dIds, ids, (size_t)count * 4);
		for (int k = 0; k < 3 && ok; ++k) if (src[k]) ok = stage[k] && c->be.h2d(stage[k], src[k], bytes);
		if (ok) {
			c->be.run_scatter_blocks((const u32*)dIds, count, n, (const u8*)stage[0], (const u8*)stage[1], (const u8*)stage[2], (u8*)c->dDist, (u8*)c->dMat, (u8*)c->dBlend);
			rebrick_blocks(c, (const u32*)dIds, count);
			ok = c->be.sync_ok();
		}
		c->be.free(dIds);
		for (int k = 0; k < 3; ++k) c->be.free(stage[k]);
	}
	if (flags) ok = ok && c->be.h2d(c->dFlags, flags, (size_t)nb * nb * nb);
	return ok ? VX_OK : fail(c, VX_ERR_DEVICE, "vx_grid_update_blocks: copy failed: " + c->be.error());
}

namespace {

// blocks the reference refreshes for an edit at pos +- ext (VoxelGrid.cpp:341-366: the test uses the FULL extents on
// both sides, twice the edited box) and the box it reports (:477-487, output order)
// The blocks an edit touches (Grid::InjectSurface / InjectMaterial walk every block and test its box against the brush's,
// src/VoxelGrid.cpp): the test is one comparison pair per axis and monotone in the block's coordinate, so the touched blocks
// are a box of blocks - first[k] .. first[k] + count[k] - 1 per axis (internal axes), found with 3 * nb tests instead of nb^3.
// The float expressions are the reference's, per axis.
bool edit_touched_box(u32 n, const float pos[3], const float ext[3], u32 first[3], u32 count[3])
{
	const u32 nb = n / 16;
	for (int k = 0; k < 3; ++k) {
		u32 lo = nb, hi = 0, hits = 0;
		for (u32 b = 0; b < nb; ++b) {
			const float bmin = (float)(b * 16), bmax = (bmin + 8.f) + 8.f;
			if (pos[k] - ext[k] > bmax || bmin > pos[k] + ext[k]) continue;
			lo = std::min(lo, b); hi = b; ++hits;
		}
		if (!hits) return false;
		if (hi - lo + 1 != hits) return false; // (cannot happen: the test is monotone; nothing is edited rather than the wrong blocks)
		first[k] = lo; count[k] = hits;
	}
	return true;
}

void edit_modified_box(u32 n, const float pos[3], const float ext[3], float outMin[3], float outMax[3])
{
	const float p[3] = { pos[0] - ext[0] / 2.0f, pos[1] - ext[1] / 2.0f, pos[2] - ext[2] / 2.0f };
	outMin[0] = std::max(0.f, p[0]); outMin[1] = std::max(0.f, p[2]); outMin[2] = std::max(0.f, p[1]);
	outMax[0] = std::min((float)n, outMin[0] + ext[0]);
	outMax[1] = std::min((float)n, outMin[1] + ext[2]);
	outMax[2] = std::min((float)n, outMin[2] + ext[1]);
}

int run_edit(vx_ctx* c, const char* what, const float pos[3], const float ext[3], const EditParams& e, float outMin[3], float outMax[3])
{
	if (!c || !pos || !ext || !outMin || !outMax) return fail(c, VX_ERR_INVALID, std::string(what) + ": null argument");
	if (!c->ownsGrid || !c->n || (c->zBegin != 0 || c->zEnd != c->n || c->yBegin != 0 || c->yEnd != c->n)) return fail(c, VX_ERR_INVALID, std::string(what) + ": needs a whole grid owned by the context (vx_grid_upload / vx_grid_upload_packed)");
	u32 first[3], count[3];
	edit_modified_box(c->n, pos, ext, outMin, outMax);
	if (!edit_touched_box(c->n, pos, ext, first, count)) return VX_OK;
	const size_t touched = (size_t)count[0] * count[1] * count[2];
	if (touched * 4 > c->scratchCap) {
		c->be.free(c->dScratch);
		c->scratchCap = touched * 4 + 4096;
		c->dScratch = c->be.alloc(c->scratchCap);
		if (!c->dScratch) { c->scratchCap = 0; return fail(c, VX_ERR_DEVICE, std::string(what) + ": allocation failed"); }
	}
	// the box's block ids are written on the device (z-major like the reference's walk): nothing is uploaded, the call's one
	// wait is the edit's completion (the brush's box the caller gets back does not depend on it)
	u32* dIds = (u32*)c->dScratch;
	c->be.run_box_ids(dIds, first, count, c->n / 16);
	c->be.run_edit(resident_view(c), (u8*)c->dFlags, dIds, (u32)touched, e);
	rebrick_blocks(c, dIds, (u32)touched);
	return c->be.sync_ok() ? VX_OK : fail(c, VX_ERR_DEVICE, std::string(what) + ": device edit failed: " + c->be.error());
}

} // namespace

int vx_grid_inject_ball(vx_ctx* c, const float pos[3], const float ext[3], float radius, int type, float outMin[3], float outMax[3])
{
	VX_ENTER(c);
	if (type < 0 || type > 2) return fail(c, VX_ERR_INVALID, "vx_grid_inject_ball: unknown injection type");
	EditParams e;
	memset(&e, 0, sizeof(e));
	if (pos && ext) for (int k = 0; k < 3; ++k) { e.pos[k] = pos[k]; e.ext[k] = ext[k]; }
	e.kind = EDIT_BALL; e.type = type; e.radius = radius;
	return run_edit(c, "vx_grid_inject_ball", pos, ext, e, outMin, outMax);
}

int vx_grid_inject_material(vx_ctx* c, const float pos[3], const float ext[3], uint8_t material, int add, float outMin[3], float outMax[3])
{
	VX_ENTER(c);
	EditParams e;
	memset(&e, 0, sizeof(e));
	if (pos && ext) for (int k = 0; k < 3; ++k) { e.pos[k] = pos[k]; e.ext[k] = ext[k]; }
	e.kind = EDIT_MATERIAL; e.material = material; e.add = add ? 1 : 0;
	return run_edit(c, "vx_grid_inject_material", pos, ext, e, outMin, outMax);
}

int vx_material_lut(vx_ctx* c, const uint8_t* lut, const uint8_t* valid)
{
	VX_ENTER(c);
	if (!c || !lut) return fail(c, VX_ERR_INVALID, "vx_material_lut: null argument");
	std::vector<u8> img(256 * 8, 0);
	for (int i = 0; i < 256; ++i) {
		memcpy(&img[i * 8], lut + i * 6, 6);
		img[i * 8 + 6] = valid ? (valid[i] ? 1 : 0) : 1;
	}
	return c->be.h2d(c->dLut, img.data(), img.size()) ? VX_OK : fail(c, VX_ERR_DEVICE, "vx_material_lut: copy failed");
}

int vx_polygonize(vx_ctx* c, uint32_t num_levels, vx_exec_info* info) { return vx_polygonize_from(c, num_levels, 0, info); }

int vx_polygonize_from(vx_ctx* c, uint32_t num_levels, uint32_t first_meshed_level, vx_exec_info* info)
{
	VX_ENTER(c);
	if (!c || !c->n || !c->dDist) return fail(c, VX_ERR_INVALID, "vx_polygonize: no grid resident (call vx_grid_upload / vx_grid_attach first)");
	if (!ensure_level_tables(c)) return fail(c, VX_ERR_DEVICE, "vx_polygonize: level table allocation failed: " + c->be.error());
	float mirrorMs = 0.f;
	{
		const bool stale = c->bricksStale;
		if (stale) c->be.begin_timing();
		if (!ensure_bricks(c)) return fail(c, VX_ERR_DEVICE, "vx_polygonize: brick mirror allocation failed: " + c->be.error());
		if (stale) mirrorMs = c->be.end_timing_ms(); // (waits for the copy: only a run on a changed grid pays this)
		// (a changed grid: what the run before handed to the general passes says nothing about this one - they are launched at
		// full width once, ~10 us, instead of leaving thousands of zero-sample blocks of, say, a height map to two workgroups)
		if (stale) c->be.slowHint[0] = c->be.slowHint[1] = ~0u;
	}
	const u32 levels = (num_levels == 0 || num_levels > c->refLevels) ? c->refLevels : num_levels;
	const u32 slabPlanes = c->zEnd - c->zBegin, slabRows = c->yEnd - c->yBegin;
	{
		// a slab must hold whole blocks of the coarsest level: thickness AND origin multiples of its block size
		const u32 coarse = 16u << (levels - 1);
		const bool zSlab = c->zBegin != 0 || c->zEnd != c->n, ySlab = c->yBegin != 0 || c->yEnd != c->n;
		if (levels > 1 && ((zSlab && ((slabPlanes % coarse) || (c->zBegin % coarse))) || (ySlab && ((slabRows % coarse) || (c->yBegin % coarse)))))
			return fail(c, VX_ERR_INVALID, "vx_polygonize: slab bounds must be multiples of the coarsest block size");
	}
	if (!c->vertCap) {
		// first guess: ~3 vertices and ~12 indices per surface voxel column; grown on demand (exact need is known after a run)
		const u32 area = c->n * c->n;
		if (!ensure_pools(c, std::max(c->poolSlack, c->poolSlack == (1u << 16) ? area * 6 : 0u), std::max(c->poolSlack * 4u, c->poolSlack == (1u << 16) ? area * 24 : 0u))) return fail(c, VX_ERR_DEVICE, "vx_polygonize: pool allocation failed");
	}
	u32 retries = 0, emitFrom = 0;
	float ms = 0.f;
	const bool hostTiming = c->hostTiming;
	// (a partial run exists on the single-stream path only: it is tried first whatever the run before - of another grid,
	// perhaps - met; blocks beyond the first capacity class repeat the run as the chain, with every level meshed)
	c->be.largeClass = first_meshed_level ? false : c->largeHint;
	auto tNow = []() { return std::chrono::steady_clock::now(); };
	auto tUs = [](std::chrono::steady_clock::time_point a, std::chrono::steady_clock::time_point b) { return (double)std::chrono::duration_cast<std::chrono::nanoseconds>(b - a).count() * 1e-3; };
	const auto t0 = tNow();
	auto t1 = t0, t2 = t0, t3 = t0;
	for (;;) {
		// every attempt gets a tag of its own for the dependency flags of the upper pass; on wrap-around the flags start over
		if (++c->runEpoch == 0) {
			for (u32 L = 1; L < c->refLevels && L < MAX_LEVELS; ++L) if (c->lv[L].matDone && !c->be.fill(c->lv[L].matDone, 0, (size_t)c->lv[L].cap * 8)) return fail(c, VX_ERR_DEVICE, "vx_polygonize: flag reset failed");
			c->runEpoch = 1;
		}
		// the set the previous run's last kernel left in its start state, if it did (then this run has no k_reset)
		const bool clean = c->otherSetClean;
		if (clean) select_run_set(c, c->runSet ^ 1u);
		c->otherSetClean = false;
		ExecParams p;
		fill_params(c, p, levels);
		// a partial run (first_meshed_level > 0) exists on the single-stream path only; anything else meshes every level and says so
		emitFrom = (first_meshed_level && c->be.partial_applies(p, levels)) ? std::min<u32>(first_meshed_level, levels) : 0u;
		c->be.emitFrom = emitFrom;
		c->be.begin_timing();
		c->be.stage_mark(0);
		c->be.tailDone = (u32*)c->dHeader + HDR_PUBLISHED + 1; // (k_tail's count of finished general workgroups)
		c->be.run_reset(p, levels, (u32*)c->dHeader, HDR_WORDS, (u32*)c->dListCounts, c->listWgs, clean); // header = 0, slot maps = -1, list counts = 0
		{
			u32 ids[MAX_LEVELS];
			for (u32 L = 0; L < MAX_LEVELS; ++L) ids[L] = L < c->refLevels ? c->lv[L].cnt * c->lv[L].cnt * c->lv[L].cnt : 0u;
			c->be.set_next_reset((u32*)c->headerSet[c->runSet ^ 1u], HDR_WORDS, (u32*)c->listCountSet[c->runSet ^ 1u], c->listWgs, c->upperMaps[c->runSet ^ 1u], ids);
		}
#if defined(VX_CASE_DUMP)
		for (u32 L = 0; L < levels; ++L) { c->be.fill(c->lv[L].caseDump, 0, (size_t)c->lv[L].cap * BLOCK_CELLS); c->be.fill(c->lv[L].trCaseDump, 0, (size_t)c->lv[L].cap * TR_CELLS * 2); }
#endif
		run_pipeline(c, p, levels);
		u32 partials = 0;
		bool published = false;
		{
			ListPlan plan;
			memset(&plan, 0, sizeof(plan));
			u32 wg = 0, idBase = 0;
			for (u32 L = 0; L <= MAX_LEVELS; ++L) {
				plan.wgStart[L] = wg;
				if (L < levels) {
					const u32 ids = c->lv[L].cnt * c->lv[L].cnt * c->lv[L].cnt;
					plan.idBase[L] = idBase;
					idBase += ids;
					wg += (ids + LIST_WG - 1) / LIST_WG;
				}
			}
			plan.counts = (u32*)c->dListCounts;
			plan.totals = (u32*)c->dHeader + HDR_LISTS;
			// the header reaches the host with the run's last kernel (HeaderPublish) - or as a copy behind it
			if (!c->hdrPinned) c->hdrPinned = (u32*)c->be.alloc_pinned((HDR_WORDS + HDR_PARTIALS) * 4);
			if (!c->hdrPinned) { c->be.sync(); c->be.end_overlapped(); return fail(c, VX_ERR_DEVICE, "vx_polygonize: pinned allocation failed"); }
			partials = std::min<u32>(c->be.head_partials(), (u32)HDR_PARTIALS);
			c->hdrPinned[HDR_PUBLISHED] = 0;
			published = c->be.lists_publish_header(c->hdrPinned, (const u32*)c->dHeader, HDR_WORDS + partials, (u32*)c->dHeader + HDR_PUBLISHED);
			published = c->be.run_block_lists(p, plan, levels) && published;
			c->otherSetClean = c->be.tailCleaned;
			c->be.stage_mark(7);
		}
		c->be.end_timing_record();
		bool okRun = published || c->be.d2h_async(c->hdrPinned, c->dHeader, (HDR_WORDS + partials) * 4); // (one host wait per run either way)
		t1 = tNow();
		okRun = okRun && c->be.sync_ok();
		c->be.end_overlapped(); // (the tail of an overlapped run was queued on a side stream)
		t2 = tNow();
		if (!okRun) return fail(c, VX_ERR_DEVICE, "vx_polygonize: device run failed: " + c->be.error());
		ms = c->be.elapsed_ms();
		if (published && c->hdrPinned[HDR_PUBLISHED] == 0) return fail(c, VX_ERR_DEVICE, "vx_polygonize: the run's header did not arrive (internal error)");
		memcpy(c->hdr, c->hdrPinned, HDR_WORDS * 4);
		if (partials) { // the block-class statistics arrive as per-workgroup partial sums behind the header (k_run_head)
			u32 readers = 0, calculated = 0;
			for (u32 i = 0; i < partials; ++i) { readers += c->hdrPinned[HDR_WORDS + i] & 0xFFFFu; calculated += c->hdrPinned[HDR_WORDS + i] >> 16; }
			c->hdr[HDR_LARGE + 1] = readers;
			c->hdr[HDR_STATS + 2] = calculated;
		}
		t3 = tNow();
		const u32 usedV = c->hdr[HDR_CURSORS], usedI = c->hdr[HDR_CURSORS + CUR_I], overflow = c->hdr[HDR_CURSORS + CUR_OVF];
		if (c->hdr[HDR_GIVEUP]) return fail(c, VX_ERR_DEVICE, "vx_polygonize: a dependency wait inside the upper pass timed out (internal error)");
		if (c->hdr[HDR_LARGE] && !c->be.largeClass) { c->be.largeClass = true; continue; } // blocks of the large class showed up: once more, with it
		if (!overflow) break;
		if (++retries > 3) return fail(c, VX_ERR_OVERFLOW, "vx_polygonize: output pools keep overflowing");
		if (!ensure_pools(c, usedV + usedV / 8 + 1024, usedI + usedI / 8 + 4096)) return fail(c, VX_ERR_OVERFLOW, "vx_polygonize: cannot grow output pools");
	}
	c->be.emitFrom = 0;
	c->largeHint = c->hdr[HDR_LARGE] != 0;
	{
		// what the next run of this context can expect on the levels >= 1 (sizes the launch of k_main): a material item per
		// active block, a regular one on the levels with a lattice copy, a transition one on the levels with transition cells
		u32 items = 0;
		for (u32 L = 1; L < levels; ++L) items += c->hdr[L] * (1u + (L < (u32)PYRAMID_LEVELS ? 1u : 0u) + (c->lv[L].hasTransitions ? 1u : 0u));
		c->be.upperItemsHint = items ? items : 1u;
		c->be.slowHint[0] = c->hdr[HDR_SLOW]; c->be.slowHint[1] = c->hdr[HDR_SLOW + 1];
	}
	c->levelsRun = levels;
	c->poolVerts = c->hdr[HDR_CURSORS]; c->poolIdx = c->hdr[HDR_CURSORS + CUR_I];
	c->poolLineage = next_lineage(); // the pools were rewritten
	c->haveSurface = true;
	c->editRoomFailed = false; // (a new surface: its first incremental run asks for room again)
	c->listsReady = false; // the host copy of the block lists is fetched on first access
	c->liveKnown = false;
	c->deviceLists = true;
	u32 idBase = 0;
	u32 blocksCalculated = 0, trivial = 0;
	for (u32 L = 0; L < levels; ++L) {
		const LevelDesc& d = c->lv[L];
		const u32 owned = d.cnt * (d.yb1 - d.yb0) * (d.zb1 - d.zb0);
		idBase += d.cnt * d.cnt * d.cnt;
		if (L < emitFrom) continue; // (a partial run: the statistics are those of the levels it meshed)
		blocksCalculated += owned;
		trivial += BLOCK_CELLS * (L == 0 ? c->hdr[HDR_STATS + 2] : owned);
	}
#if defined(VX_MAIN_PROFILE)
	{
		static const char* names[8] = { "barrier before dequeue", "dequeue", "level-0 batch", "material block", "regular block (levels >= 1)", "transition block", "exit", "-" };
		unsigned long long sum = 0;
		for (int i = 0; i < 8; ++i) sum += c->hdr[HDR_LARGE + 4 + i];
		for (int i = 0; i < 7; ++i) fprintf(stderr, "[main profile] %-28s %10u x64 cycles  %5.1f %%\n", names[i], c->hdr[HDR_LARGE + 4 + i], 100.0 * c->hdr[HDR_LARGE + 4 + i] / (double)(sum ? sum : 1));
	}
#endif
#if defined(VX_MAIN_TRACE)
	{
		u32 cnt = 0;
		if (hipMemcpyFromSymbol(&cnt, HIP_SYMBOL(g_mainTraceN), sizeof(cnt)) == hipSuccess && cnt && cnt <= 4096u) {
			std::vector<unsigned long long> v((size_t)cnt * 18);
			if (hipMemcpyFromSymbol(v.data(), HIP_SYMBOL(g_mainTrace), v.size() * 8) == hipSuccess) {
				unsigned long long t0 = ~0ull;
				for (u32 i = 0; i < cnt; ++i) t0 = std::min(t0, v[18 * i + 2]);
				static const char* kinds[4] = { "level0", "mat", "reg", "tr" };
				for (u32 i = 0; i < cnt; ++i) {
					const unsigned long long* e = &v[18 * i];
					const u32 what = (u32)e[0];
					fprintf(stderr, "[main trace] %-6s L%u slot %5u wg %4u  ticket %7.2f  start %7.2f  waited %7.2f  end %7.2f us\n", kinds[what >> 28], (what >> 24) & 15u, what & 0xFFFFFFu, (u32)e[1],
					        (e[2] - t0) * 0.01, (e[3] - t0) * 0.01, e[4] ? (double)(long long)(e[4] - t0) * 0.01 : 0.0, (e[5] - t0) * 0.01);
					if ((what >> 28) >= 1u) { fprintf(stderr, "             marks:"); for (int m = 0; m < 12; ++m) if (e[6 + m]) fprintf(stderr, " %d:%.2f", m, (double)(long long)(e[6 + m] - t0) * 0.01); fprintf(stderr, "\n"); }
				}
			}
		}
		cnt = 0;
		(void)hipMemcpyToSymbol(HIP_SYMBOL(g_mainTraceN), &cnt, sizeof(cnt));
	}
#endif
#if defined(VX_F0_PROFILE)
	{
		static const char* names[9] = { "between blocks", "top barrier", "deposit + barrier", "own bitmap + barrier", "prefix + list + barrier", "cells + barrier", "bases + reserve + describe + barrier", "vertices + triangles", "record" };
		unsigned long long v[12], sum = 0;
		if (hipMemcpyFromSymbol(v, HIP_SYMBOL(g_f0prof), sizeof(v)) == hipSuccess) {
			for (int i = 0; i < 9; ++i) sum += v[i];
			for (int i = 0; i < 9; ++i) fprintf(stderr, "[f0 profile] %-36s %12llu x64 cycles  %5.1f %%\n", names[i], v[i], 100.0 * (double)v[i] / (double)(sum ? sum : 1));
			memset(v, 0, sizeof(v));
			(void)hipMemcpyToSymbol(HIP_SYMBOL(g_f0prof), v, sizeof(v));
		}
	}
#endif
#if defined(VX_CLS_PROFILE)
	{
		static const char* names[5] = { "class bytes + init", "loads + sign masks", "classification", "slot allocation (+ ancestors)", "bitmap stores" };
		unsigned long long sum = 0;
		for (int i = 0; i < 5; ++i) sum += c->hdr[HDR_LARGE + 16 + i];
		for (int i = 0; i < 5; ++i) fprintf(stderr, "[classify profile] %-30s %10u x16 cycles  %5.1f %%\n", names[i], c->hdr[HDR_LARGE + 16 + i], 100.0 * c->hdr[HDR_LARGE + 16 + i] / (double)(sum ? sum : 1));
	}
#endif
#if defined(VX_TR_PROFILE)
	{
		static const char* names[11] = { "requests issued", "sign summaries + barrier", "planes to LDS + barrier", "classification + barrier", "batch bits + scan", "material wait", "list + barrier", "count + barrier", "scans + reservation", "describe + vertices", "indices" };
		unsigned long long sum = 0;
		for (int i = 0; i < 11; ++i) sum += c->hdr[HDR_LARGE + 16 + i];
		for (int i = 0; i < 11; ++i) fprintf(stderr, "[transition profile] %-28s %10u x64 cycles  %5.1f %%\n", names[i], c->hdr[HDR_LARGE + 16 + i], 100.0 * c->hdr[HDR_LARGE + 16 + i] / (double)(sum ? sum : 1));
	}
#endif
#if defined(VX_REG_PROFILE)
	{
		static const char* names[16] = { "next item", "top barrier", "begin+stage+barrier", "prefix scan", "list+barrier", "cells+barrier", "count+barrier", "vertex scan+reserve", "describe+barrier", "emit vertices", "barrier", "keep+barrier", "index scan+reserve", "stage indices+barrier", "flush indices", "record" };
		unsigned long long sum = 0;
		for (int i = 0; i < 16; ++i) sum += c->hdr[HDR_LARGE + 16 + i];
		for (int i = 0; i < 16; ++i) fprintf(stderr, "[regular profile, levels >= 1] %-24s %10u kcycles  %5.1f %%\n", names[i], c->hdr[HDR_LARGE + 16 + i], 100.0 * c->hdr[HDR_LARGE + 16 + i] / (double)(sum ? sum : 1));
	}
#endif
#if defined(VX_R0_PROFILE)
	{
		static const char* names[10] = { "top barrier", "deposit+barrier", "prefix+list+barrier", "cells", "scan barrier", "reserve+describe", "barrier", "vertices+indices", "record", "next item (drain)" };
		unsigned long long sum = 0;
		for (int i = 0; i < 10; ++i) sum += c->hdr[HDR_LARGE + 4 + i];
		for (int i = 0; i < 10; ++i) fprintf(stderr, "[r0 profile] %-22s %10u kcycles  %5.1f %%\n", names[i], c->hdr[HDR_LARGE + 4 + i], 100.0 * c->hdr[HDR_LARGE + 4 + i] / (double)(sum ? sum : 1));
	}
#endif
	if (hostTiming) {
		const auto t5 = tNow();
		fprintf(stderr, "[vx host] enqueue %.0f us, wait+events %.0f us, header %.0f us, after-header %.0f us, device %.0f us\n",
		        tUs(t0, t1), tUs(t1, t2), tUs(t2, t3), tUs(t3, t5), ms * 1e3);
	}
	c->nextId = idBase;
	c->stats[0] = blocksCalculated;
	c->stats[2] = c->hdr[HDR_STATS + 0];
	c->stats[1] = trivial - c->stats[2];
	c->stats[3] = c->hdr[HDR_STATS + 1];
	for (int i = 0; i < 16; ++i) c->stats[4 + i] = c->hdr[HDR_STATS + 4 + i];
	if (info) {
		memset(info, 0, sizeof(*info));
		info->levels = levels;
		info->retries = retries;
		info->device_ms = ms;
		info->total_verts = c->poolVerts;
		info->total_indices = c->poolIdx;
		for (u32 L = 0; L < levels && L < 8; ++L) info->active_blocks[L] = c->hdr[L];
		info->algorithmic_bytes = (uint64_t)c->n * slabRows * slabPlanes + 2ull * 4096 * c->hdr[0] + 48ull * c->poolVerts + 4ull * c->poolIdx;
		info->blocks_read = c->hdr[HDR_LARGE + 1];
		info->mirror_ms = mirrorMs;
		info->first_meshed_level = emitFrom;
	}
	return VX_OK;
}

// Incremental re-polygonization (TransVoxelRun::Execute with a Modification, TransVoxelImpl.cpp:429-465): per level the
// blocks of the dirty box plus one ring are dropped and rebuilt; the material caches keep their old contents.
namespace {

// live vertices / indices according to the block lists
void add_block_totals(const EmittedBlock& e, uint64_t& verts, uint64_t& idx)
{
	verts += e.rec.vCount; idx += e.rec.iCount;
	for (int f = 0; f < 6; ++f) { verts += e.rec.tvCount[f]; idx += e.rec.tiCount[f]; }
}

// (summed once per surface; an incremental run then subtracts what it drops and adds what it appends - a walk over every block
// of every level per edit was a third of the call's host time at 512^3)
void live_totals(vx_ctx* c, uint64_t& verts, uint64_t& idx)
{
	if (!c->liveKnown) {
		c->liveVerts = c->liveIdx = 0;
		for (u32 L = 0; L < c->levelsRun; ++L) for (size_t i = 0; i < c->blocks[L].size(); ++i) if (c->blockDead[L].empty() || !c->blockDead[L][i]) add_block_totals(c->blocks[L][i], c->liveVerts, c->liveIdx);
		c->liveKnown = true;
	}
	verts = c->liveVerts; idx = c->liveIdx;
}

// the second pair of pools vx_compact_pools packs into: at least `needV` vertices / `needI` indices, allocated with the pools'
// own capacity when that is more
bool ensure_spare_pools(vx_ctx* c, uint64_t needV, uint64_t needI)
{
	if (c->dVertsSpare && c->dIdxSpare && c->spareVertCap >= needV && c->spareIdxCap >= needI) return true;
	const uint64_t capV = std::max<uint64_t>(needV, c->vertCap), capI = std::max<uint64_t>(needI, c->idxCap);
	if (capV > 0xFFFFFFFFull || capI > 0xFFFFFFFFull) return false;
	c->be.free(c->dVertsSpare); c->be.free(c->dIdxSpare);
	c->dVertsSpare = c->be.alloc((size_t)capV * sizeof(PolyVertex));
	c->dIdxSpare = c->be.alloc((size_t)capI * 4);
	if (!c->dVertsSpare || !c->dIdxSpare) {
		c->be.free(c->dVertsSpare); c->be.free(c->dIdxSpare);
		c->dVertsSpare = c->dIdxSpare = nullptr; c->spareVertCap = c->spareIdxCap = 0;
		return false;
	}
	c->spareVertCap = (u32)capV; c->spareIdxCap = (u32)capI;
	return true;
}

} // namespace

int vx_compact_pools(vx_ctx* c)
{
	VX_ENTER(c);
	if (!c || !c->haveSurface) return fail(c, VX_ERR_INVALID, "vx_compact_pools: no surface");
	if (ensure_lists(c) != VX_OK) return VX_ERR_DEVICE;
	uint64_t liveV, liveI;
	live_totals(c, liveV, liveI);
	if (liveV == c->poolVerts && liveI == c->poolIdx) return VX_OK; // nothing dead
	// copy list: (source offset, destination offset, count) per mesh, vertices first then indices
	std::vector<u32> segV, segI;
	u32 nv = 0, ni = 0;
	auto moveV = [&](u32& off, u32 count) { if (count) { segV.push_back(off); segV.push_back(nv); segV.push_back(count); } off = nv; nv += count; };
	auto moveI = [&](u32& off, u32 count) { if (count) { segI.push_back(off); segI.push_back(ni); segI.push_back(count); } off = ni; ni += count; };
	std::vector<EmittedBlock> moved[MAX_LEVELS];
	for (u32 L = 0; L < c->levelsRun; ++L) {
		moved[L] = c->blocks[L];
		for (EmittedBlock& e : moved[L]) {
			moveV(e.rec.vOff, e.rec.vCount); moveI(e.rec.iOff, e.rec.iCount);
			for (int f = 0; f < 6; ++f) { moveV(e.rec.tvOff[f], e.rec.tvCount[f]); moveI(e.rec.tiOff[f], e.rec.tiCount[f]); }
		}
	}
	// The live meshes move into a second pair of pools, which the context keeps from then on (the pools swap roles at every
	// compaction): allocating and freeing pool-sized device buffers cost 15-55 ms per compaction - every eight or so edits
	// of BASELINE config 5 - against 0.3 ms for the copy itself.
	// It has to hold the live meshes and room for the next edits' blocks; the first incremental run of a surface allocates it
	// with the pools' capacity (ensure_spare_pools), so that no edit in a sequence pays for the allocation.
	if (!ensure_spare_pools(c, (uint64_t)nv + nv / 2 + c->poolSlack, (uint64_t)ni + ni / 2 + 4ull * c->poolSlack)) return fail(c, VX_ERR_DEVICE, "vx_compact_pools: allocation failed");
	const size_t segWords = segV.size() + segI.size() + 4;
	if (segWords * 4 > c->segCap) {
		c->be.free(c->dSeg);
		c->segCap = segWords * 4 + segWords;
		c->dSeg = c->be.alloc(c->segCap);
		if (!c->dSeg) { c->segCap = 0; return fail(c, VX_ERR_DEVICE, "vx_compact_pools: allocation failed"); }
	}
	void* dSeg = c->dSeg;
	bool ok = true;
	if (ok && !segV.empty()) ok = c->be.h2d(dSeg, segV.data(), segV.size() * 4);
	if (ok && !segI.empty()) ok = c->be.h2d((u32*)dSeg + segV.size(), segI.data(), segI.size() * 4);
	if (ok) {
		c->be.run_copy_segments((const u32*)dSeg, (u32)(segV.size() / 3), c->dVerts, c->dVertsSpare, (u32)sizeof(PolyVertex));
		c->be.run_copy_segments((const u32*)dSeg + segV.size(), (u32)(segI.size() / 3), c->dIdx, c->dIdxSpare, 4u);
		ok = c->be.sync_ok();
	}
	if (!ok) return fail(c, VX_ERR_DEVICE, "vx_compact_pools: device copy failed: " + c->be.error());
	std::swap(c->dVerts, c->dVertsSpare); std::swap(c->dIdx, c->dIdxSpare);
	std::swap(c->vertCap, c->spareVertCap); std::swap(c->idxCap, c->spareIdxCap);
	c->deviceLists = false;
	c->poolVerts = nv; c->poolIdx = ni;
	c->poolLineage = next_lineage(); // host copies describe the old layout
	for (u32 L = 0; L < c->levelsRun; ++L) c->blocks[L].swap(moved[L]);
	return VX_OK;
}

// An incremental run found the pools too small (its cursors ended at usedV / usedI; nothing it wrote is referenced yet).  When a
// third of what the pools hold is dead, packing them (into the spare pair: no allocation) makes the room; otherwise, or if that
// is still not enough, they grow to one and a half times what the run needs.
static int make_room_for_edit(vx_ctx* c, u32 usedV, u32 usedI)
{
	const auto t0 = std::chrono::steady_clock::now();
	const u32 addedV = usedV > c->poolVerts ? usedV - c->poolVerts : 0, addedI = usedI > c->poolIdx ? usedI - c->poolIdx : 0;
	uint64_t liveV, liveI;
	live_totals(c, liveV, liveI);
	bool packed = false;
	if ((uint64_t)c->poolVerts * 2 > liveV * 3 || (uint64_t)c->poolIdx * 2 > liveI * 3) {
		const int rc = vx_compact_pools(c);
		if (rc != VX_OK) return rc;
		packed = true;
	}
	const uint64_t needV = (uint64_t)c->poolVerts + addedV, needI = (uint64_t)c->poolIdx + addedI;
	bool grown = false;
	if (needV > c->vertCap || needI > c->idxCap) {
		const uint64_t wantV = needV + needV / 2 + 1024, wantI = needI + needI / 2 + 4096;
		if (wantV > 0xFFFFFFFFull || wantI > 0xFFFFFFFFull || !grow_pools_keeping(c, (u32)wantV, (u32)wantI)) return fail(c, VX_ERR_OVERFLOW, "vx_polygonize_dirty: cannot grow output pools");
		grown = true;
	}
	if (c->hostTiming) fprintf(stderr, "[vx host, dirty] the run did not fit: pools%s%s: %.0f us\n", packed ? " packed" : "", grown ? " grown" : "",
	                           (double)std::chrono::duration_cast<std::chrono::nanoseconds>(std::chrono::steady_clock::now() - t0).count() * 1e-3);
	return VX_OK;
}

int vx_polygonize_dirty(vx_ctx* c, const float min_corner[3], const float max_corner[3], vx_exec_info* info,
                        uint32_t* modified_ids, uint32_t cap, uint32_t* count)
{
	VX_ENTER(c);
	if (!c || !min_corner || !max_corner) return fail(c, VX_ERR_INVALID, "vx_polygonize_dirty: null argument");
	if (!c->haveSurface) return fail(c, VX_ERR_INVALID, "vx_polygonize_dirty: run vx_polygonize first");
	if (c->zBegin != 0 || c->zEnd != c->n || c->yBegin != 0 || c->yEnd != c->n) return fail(c, VX_ERR_INVALID, "vx_polygonize_dirty: not supported on slabs");
	const u32 levels = c->levelsRun;
	auto tNow = []() { return std::chrono::steady_clock::now(); };
	auto tUs = [](std::chrono::steady_clock::time_point a, std::chrono::steady_clock::time_point b) { return (double)std::chrono::duration_cast<std::chrono::nanoseconds>(b - a).count() * 1e-3; };
	const auto t0 = tNow(); // (VX_HOST_TIMING: where the call's host time goes)
	auto t1 = t0, t2 = t0, t3 = t0, t4 = t0;
	if (ensure_lists(c, false) != VX_OK) return VX_ERR_DEVICE;
	if (!ensure_bricks(c)) return fail(c, VX_ERR_DEVICE, "vx_polygonize_dirty: brick mirror allocation failed: " + c->be.error());
	// the new blocks' meshes are appended behind what the pools already hold: every kept block stays where it is;
	// once more than half of the pools is dead they are packed first (into the spare pair, which the first incremental run of
	// a context allocates: 15-55 ms that would otherwise hit whichever edit triggers the first compaction; a failure here is
	// not an error - vx_compact_pools asks again when it needs the pair)
	// The same call gives the pools room for twice what is alive: appended blocks then fill them up to the point where they
	// are packed, and a sequence of edits allocates nothing (an allocation of this size can take tens of milliseconds when
	// the driver has to reclaim memory first; pools that still overflow - the surface grew - grow by half, as before).
	if (!c->dVertsSpare && !c->editRoomFailed) {
		uint64_t liveV, liveI;
		live_totals(c, liveV, liveI);
		const uint64_t wantV = 2 * liveV + 2ull * c->poolSlack, wantI = 2 * liveI + 8ull * c->poolSlack;
		bool got = true;
		if (wantV <= 0xFFFFFFFFull && wantI <= 0xFFFFFFFFull) got = grow_pools_keeping(c, (u32)wantV, (u32)wantI);
		got = ensure_spare_pools(c, 0, 0) && got;
		//  a reservation the device could not give is not asked for again by every later edit - each attempt is a
		// large allocation that can take tens of milliseconds; vx_compact_pools asks when it needs the pair, a full run starts over
		if (!got) { c->editRoomFailed = true; if (c->hostTiming) fprintf(stderr, "[vx host, dirty] room for edits could not be reserved (not asked for again before the next full run)\n"); }
		if (c->hostTiming) fprintf(stderr, "[vx host, dirty] room for edits (pools for twice the live meshes, spare pair): %.0f us\n", tUs(t0, tNow()));
	}
	{
		uint64_t liveV, liveI;
		live_totals(c, liveV, liveI);
		if ((uint64_t)c->poolVerts > 2 * liveV + c->poolSlack || (uint64_t)c->poolIdx > 2 * liveI + 4ull * c->poolSlack) {
			const auto tc = tNow();
			const int rc = vx_compact_pools(c);
			if (rc != VX_OK) return rc;
			if (c->hostTiming) fprintf(stderr, "[vx host, dirty] pools packed (more than half dead): %.0f us\n", tUs(tc, tNow()));
		}
	}
	// ---- block lists (everything in output, Y-up, coordinates like the reference) ----------------------------
	// Nothing of the context's surface state (block lists, id counter) is touched before the device run has succeeded:
	// the lists without the dropped blocks are built aside and swapped in at the end.
	std::vector<u32> coords, ids;
	std::vector<EmittedBlock> fresh[MAX_LEVELS];                 // the rebuilt blocks, in list order
	float dropLo[MAX_LEVELS][3] = {}, dropHi[MAX_LEVELS][3] = {}; // per level: blocks whose minimal corner lies in here are dropped (:443-450)
	u32 nextId = c->nextId;
	u32 start[MAX_LEVELS + 1] = { 0 }, cnt[MAX_LEVELS] = { 0 };
	u32 boxLo[MAX_LEVELS][3] = {}, boxHi[MAX_LEVELS][3] = {}; // the levels' dirty boxes in block coordinates (output axes)
	const float ext = (float)c->n;
	for (u32 L = 0; L < levels; ++L) {
		const LevelDesc& d = c->lv[L];
		const float bm = (float)(d.mult * 16);
		float lo[3], hi[3];
		for (int k = 0; k < 3; ++k) {
			lo[k] = std::floor(min_corner[k] / bm - 1.0f) * bm;
			hi[k] = std::floor(max_corner[k] / bm + 2.0f) * bm;
			lo[k] = std::min(std::max(lo[k], 0.f), ext);
			hi[k] = std::min(std::max(hi[k], 0.f), ext);
		}
		for (int k = 0; k < 3; ++k) { boxLo[L][k] = (u32)(lo[k] / bm); boxHi[L][k] = std::max((u32)(hi[k] / bm), boxLo[L][k]); }
		for (int k = 0; k < 3; ++k) { dropLo[L][k] = lo[k]; dropHi[L][k] = hi[k]; }
		start[L] = (u32)coords.size();
		for (u32 z = (u32)(lo[1] / bm); z < (u32)(hi[1] / bm); ++z)      // internal z = output y
		for (u32 y = (u32)(lo[2] / bm); y < (u32)(hi[2] / bm); ++y)
		for (u32 x = (u32)(lo[0] / bm); x < (u32)(hi[0] / bm); ++x) {
			coords.push_back(block_coord_id(x, y, z, d.cnt));
			ids.push_back(nextId++);
		}
		cnt[L] = (u32)coords.size() - start[L];
	}
	start[levels] = (u32)coords.size();
	const u32 total = (u32)coords.size();
	if (count) *count = total;
	for (u32 k = 0; k < total && k < cap && modified_ids; ++k) modified_ids[k] = ids[k];
	if (total > c->dirtyCap) {
		c->be.free(c->dDirty); c->be.free(c->dWork); c->be.free(c->dGather);
		c->dirtyCap = total + total / 2 + 64;
		c->dDirty = c->be.alloc((size_t)c->dirtyCap * 4);
		c->dWork = c->be.alloc((size_t)c->dirtyCap * 4);
		c->dGather = c->be.alloc((size_t)c->dirtyCap * sizeof(BlockRecord));
		if (!c->dDirty || !c->dWork || !c->dGather) { c->dirtyCap = 0; return fail(c, VX_ERR_DEVICE, "vx_polygonize_dirty: allocation failed"); }
	}
	u32 prevActive[MAX_LEVELS];
	for (u32 L = 0; L < MAX_LEVELS; ++L) prevActive[L] = c->hdr[L];
	u32 retries = 0;
	float ms = 0.f;
	t1 = tNow();
	std::vector<BlockRecord> recsChain;       // (the chain's records are downloaded into this; the three-launch path reads the page-locked landing buffer in place)
	const BlockRecord* recs = nullptr;
	bool recordsInListOrder = false;          // the three-launch path: a level's records follow the order of its box's coordinates
	// Three launches where the table-driven passes apply (k_dirty_head | k_main<true> | k_dirty_tail, vx_hip.hip): the box goes
	// out as launch arguments, records and header come back through page-locked memory written by the last kernel.  A run
	// that meets a block beyond the first capacity class says so in its header and is repeated as the chain of launches.
	bool fused = false, uploaded = false;
	{
		ExecParams p0;
		fill_params(c, p0, levels);
		fused = total && cnt[0] && c->be.dirty_fused_applies(p0, levels, c->dirtyLargeHint || c->dirtyLarge0Hint);
	}
	if (fused) {
		if (!c->dDirtyTicket) {
			c->dDirtyTicket = c->be.alloc(64);
			if (!c->dDirtyTicket || !c->be.fill(c->dDirtyTicket, 0, 64)) return fail(c, VX_ERR_DEVICE, "vx_polygonize_dirty: allocation failed");
			c->dirtyTickets = 0;
		}
		if (total > c->hRecCap) {
			c->be.free_pinned(c->hRecs);
			c->hRecCap = (size_t)c->dirtyCap;
			c->hRecs = (BlockRecord*)c->be.alloc_pinned(c->hRecCap * sizeof(BlockRecord));
			if (!c->hRecs) { c->hRecCap = 0; return fail(c, VX_ERR_DEVICE, "vx_polygonize_dirty: pinned allocation failed"); }
		}
		if (!c->hdrPinned) c->hdrPinned = (u32*)c->be.alloc_pinned((HDR_WORDS + HDR_PARTIALS) * 4);
		if (!c->hdrPinned) return fail(c, VX_ERR_DEVICE, "vx_polygonize_dirty: pinned allocation failed");
	}
	for (;;) {
		ExecParams p;
		if (fused && ++c->runEpoch == 0) {
			// (k_main's dependency flags carry the run's tag; on wrap-around they start over)
			for (u32 L = 1; L < c->refLevels && L < MAX_LEVELS; ++L) if (c->lv[L].matDone && !c->be.fill(c->lv[L].matDone, 0, (size_t)c->lv[L].cap * 8)) return fail(c, VX_ERR_DEVICE, "vx_polygonize_dirty: flag reset failed");
			c->runEpoch = 1;
		}
		fill_params(c, p, levels);
		p.G.dirty = 1;
		c->be.largeClass = true;
		for (u32 L = 0; L < levels; ++L) { p.G.workItems[L] = (const u32*)c->dWork + start[L]; p.G.prevActive[L] = prevActive[L]; }
		if (fused) {
			Backend::DirtyLaunch q;
			memset(&q, 0, sizeof(q));
			for (u32 L = 0; L < levels; ++L) {
				// (boxLo / boxHi are output coordinates, Y up: internal y = output z, internal z = output y)
				q.lo[L][0] = boxLo[L][0]; q.lo[L][1] = boxLo[L][2]; q.lo[L][2] = boxLo[L][1];
				q.hi[L][0] = boxHi[L][0]; q.hi[L][1] = boxHi[L][2]; q.hi[L][2] = boxHi[L][1];
			}
			for (u32 L = 0; L <= MAX_LEVELS; ++L) q.start[L] = start[L < levels ? L : levels];
			q.work = (u32*)c->dWork; q.info = (u32*)c->dDirty;
			q.ticket = (u32*)c->dDirtyTicket; c->dirtyTickets += cnt[0]; q.ticketTarget = c->dirtyTickets;
			q.header = (u32*)c->dHeader; q.resetFrom = HDR_CURSORS; q.resetTo = HDR_WORDS; q.poolVerts = c->poolVerts; q.poolIdx = c->poolIdx;
			q.roleTicket = (u32*)c->dHeader + HDR_PUBLISHED + 2; q.slowDone = (u32*)c->dHeader + HDR_PUBLISHED + 1;
			q.hostRecs = c->hRecs; q.hostHeader = c->hdrPinned; q.headerWords = HDR_WORDS; q.publishedWord = HDR_PUBLISHED;
			c->hdrPinned[HDR_PUBLISHED] = 0;
			c->be.begin_timing();
			c->be.run_dirty_fused(p, levels, q, c->dirtyLarge0Hint, c->dirtyLargeHint);
			c->be.end_timing_record();
			t2 = tNow();
			//  a run that failed may not have counted all its workgroups into the device's ticket: the host's count and
			// the device's start over together, or no later run would ever elect its last workgroup
			auto ticketLost = [&]() {
				if (c->dDirtyTicket && !c->be.fill(c->dDirtyTicket, 0, 64)) { c->be.free(c->dDirtyTicket); c->dDirtyTicket = nullptr; }
				c->dirtyTickets = 0;
			};
			if (!c->be.sync_ok()) { const std::string why = c->be.error(); ticketLost(); return fail(c, VX_ERR_DEVICE, "vx_polygonize_dirty: device run failed: " + why); }
			t3 = tNow();
			ms = c->be.elapsed_ms();
			if (c->hdrPinned[HDR_PUBLISHED] == 0) { ticketLost(); (void)c->be.sync_ok(); return fail(c, VX_ERR_DEVICE, "vx_polygonize_dirty: the run's header did not arrive (internal error)"); }
			memcpy(c->hdr, c->hdrPinned, HDR_WORDS * 4);
			if (c->hdr[HDR_GIVEUP]) { ticketLost(); (void)c->be.sync_ok(); return fail(c, VX_ERR_DEVICE, "vx_polygonize_dirty: a dependency wait inside the run timed out (internal error)"); }
			{
				// blocks beyond the first capacity class, on level 0 (counted by k_dirty_head) and above it: unannounced, the run is
				// repeated once with their launches made
				const u32 large0 = c->hdr[HDR_LARGE + 2], largeUpper = c->hdr[HDR_LARGE] > large0 ? c->hdr[HDR_LARGE] - large0 : 0u;
				const bool missed = (large0 && !c->dirtyLarge0Hint) || (largeUpper && !c->dirtyLargeHint);
				c->dirtyLarge0Hint = large0 != 0 || (missed && c->dirtyLarge0Hint);
				c->dirtyLargeHint = largeUpper != 0 || (missed && c->dirtyLargeHint);
				if (missed) continue;
			}
			const u32 usedV = c->hdr[HDR_CURSORS], usedI = c->hdr[HDR_CURSORS + CUR_I], overflow = c->hdr[HDR_CURSORS + CUR_OVF];
			if (!overflow) { recs = c->hRecs; recordsInListOrder = true; break; }
			if (++retries > 3) return fail(c, VX_ERR_OVERFLOW, "vx_polygonize_dirty: output pools keep overflowing");
			// (appended blocks fill the pools' slack edit after edit: growing by half keeps overflows - a repeated run, a copy of
			// the pools and an allocation, ~1 ms - rare; vx_compact_pools gives dead ranges back)
			{ const int rc = make_room_for_edit(c, usedV, usedI); if (rc != VX_OK) return rc; }
			continue;
		}
		if (!uploaded && total && !c->be.h2d(c->dDirty, coords.data(), (size_t)total * 4)) return fail(c, VX_ERR_DEVICE, "vx_polygonize_dirty: upload failed");
		uploaded = true;
		c->be.begin_timing();
		c->be.stage_mark(0);
		{
			// cursors (continuing at the pools' current ends), stats, work counts — NOT the slot counts
			u32 tail[HDR_WORDS - HDR_CURSORS];
			memset(tail, 0, sizeof(tail));
			tail[CUR_V] = c->poolVerts; tail[CUR_I] = c->poolIdx;
			if (!c->be.h2d((u32*)c->dHeader + HDR_CURSORS, tail, sizeof(tail))) return fail(c, VX_ERR_DEVICE, "vx_polygonize_dirty: header upload failed");
		}
		c->be.stage_mark(1);
		c->be.run_classify_blocks(p, (const u32*)c->dDirty + start[0], cnt[0]);
		c->be.stage_mark(2);
		c->be.run_hierarchy(p, levels);
		c->be.run_build_worklist(p, (const u32*)c->dDirty, start, cnt, levels, (u32*)c->dWork);
		c->be.stage_mark(3);
		for (u32 L = 1; L < levels; ++L) c->be.run_material(p, L);
		c->be.stage_mark(4);
		c->be.run_regular(p, levels);
		c->be.stage_mark(5);
		c->be.run_transition(p, levels);
		c->be.run_gather_records(p, levels, start, (BlockRecord*)c->dGather);
		c->be.stage_mark(6);
		c->be.stage_mark(7);
		ms = c->be.end_timing_ms();
		if (!c->be.d2h(c->hdr, c->dHeader, HDR_WORDS * 4)) return fail(c, VX_ERR_DEVICE, "vx_polygonize_dirty: device run failed: " + c->be.error());
		const u32 usedV = c->hdr[HDR_CURSORS], usedI = c->hdr[HDR_CURSORS + CUR_I], overflow = c->hdr[HDR_CURSORS + CUR_OVF];
		if (!overflow) {
			recsChain.resize(total);
			if (total && !c->be.d2h(recsChain.data(), c->dGather, (size_t)total * sizeof(BlockRecord))) return fail(c, VX_ERR_DEVICE, "vx_polygonize_dirty: record download failed");
			recs = recsChain.data();
			break;
		}
		if (++retries > 3) return fail(c, VX_ERR_OVERFLOW, "vx_polygonize_dirty: output pools keep overflowing");
		{ const int rc = make_room_for_edit(c, usedV, usedI); if (rc != VX_OK) return rc; }
	}
	// ---- new blocks, appended in list order (TransVoxelImpl.cpp:1274-1293) -------------------------------------
	u32 trivialBlocks = c->hdr[HDR_STATS + 2];
	for (u32 L = 0; L < levels; ++L) {
		const LevelDesc& d = c->lv[L];
		const u32 nWork = c->hdr[HDR_WORK + L];
		// The three-launch path lists a level's active slots in the order of the box's coordinates (k_dirty_head), which is the
		// order of `coords`: one walk over both.  The chain appends to its work lists with atomics (arbitrary order): its
		// records are indexed by block coordinate first.
		std::vector<std::pair<u32, u32> > byCoord;
		bool inOrder = recordsInListOrder;
	again:
		if (!inOrder) {
			byCoord.reserve(nWork);
			for (u32 k = 0; k < nWork; ++k) byCoord.push_back(std::make_pair(recs[start[L] + k].coordId, start[L] + k));
			std::sort(byCoord.begin(), byCoord.end());
		}
		u32 w = 0;
		fresh[L].reserve(nWork);
		for (u32 k = 0; k < cnt[L]; ++k) {
			const u32 coord = coords[start[L] + k];
			const BlockRecord* rp = nullptr;
			if (inOrder) {
				if (w < nWork && recs[start[L] + w].coordId == coord) rp = &recs[start[L] + w++];
			} else {
				auto it = std::lower_bound(byCoord.begin(), byCoord.end(), std::make_pair(coord, 0u));
				if (it != byCoord.end() && it->first == coord) rp = &recs[it->second];
			}
			if (!rp) continue; // no surface in this block
			const BlockRecord& r = *rp;
			if (!r.vCount) continue;
			fresh[L].emplace_back();
			EmittedBlock& e = fresh[L].back();
			e.rec = r;
			e.id = ids[start[L] + k];
			block_corners(d, coord, e.minc, e.maxc);
		}
		//  records that do not follow the box's coordinate order are looked up by coordinate, like the chain's: nothing
		// of the context has been committed yet, and nothing fails here
		if (inOrder && w != nWork) { inOrder = false; fresh[L].clear(); goto again; }
		if (L) trivialBlocks += cnt[L];
	}
	c->poolVerts = c->hdr[HDR_CURSORS]; c->poolIdx = c->hdr[HDR_CURSORS + CUR_I]; // from here on nothing can fail
	c->deviceLists = false;
	// the lists change in place, now that nothing can fail any more: the dropped blocks leave (everything behind the first of
	// them moves up, nothing is copied in front of it), the rebuilt ones are appended (:1274-1293)
	for (u32 L = 0; L < levels; ++L) {
		std::vector<EmittedBlock>& list = c->blocks[L];
		const float* lo = dropLo[L];
		const float* hi = dropHi[L];
		uint64_t goneV = 0, goneI = 0;
		std::vector<unsigned char>& dead = c->blockDead[L];
		for (size_t i = 0; i < list.size(); ++i) {
			const EmittedBlock& e = list[i];
			if (!dead.empty() && dead[i]) continue;
			if (e.minc[0] >= lo[0] && e.minc[1] >= lo[1] && e.minc[2] >= lo[2] && e.minc[0] < hi[0] && e.minc[1] < hi[1] && e.minc[2] < hi[2]) {
				if (dead.empty()) dead.assign(list.size(), 0);
				dead[i] = 1; ++c->deadBlocks;
				add_block_totals(e, goneV, goneI);
			}
		}
		list.insert(list.end(), fresh[L].begin(), fresh[L].end());
		if (!dead.empty()) dead.resize(list.size(), 0);
		if (c->liveKnown) {
			c->liveVerts -= goneV; c->liveIdx -= goneI;
			for (const EmittedBlock& e : fresh[L]) add_block_totals(e, c->liveVerts, c->liveIdx);
		}
	}
	t4 = tNow();
	if (c->hostTiming) fprintf(stderr, "[vx host, dirty] lists + box %.0f us, enqueue %.0f us, wait %.0f us, records + lists %.0f us, device %.0f us, stream idle before the run %.0f us\n", tUs(t0, t1), tUs(t1, t2), tUs(t2, t3), tUs(t3, t4), ms * 1e3, (double)c->be.idle_before_ms() * 1e3);
	c->nextId = nextId;
	c->stats[0] = total;
	c->stats[2] = c->hdr[HDR_STATS + 0];
	c->stats[1] = BLOCK_CELLS * trivialBlocks - c->stats[2];
	c->stats[3] = c->hdr[HDR_STATS + 1];
	for (int i = 0; i < 16; ++i) c->stats[4 + i] = c->hdr[HDR_STATS + 4 + i];
	if (info) {
		memset(info, 0, sizeof(*info));
		info->levels = levels;
		info->retries = retries;
		info->device_ms = ms;
		info->total_verts = c->poolVerts;
		info->total_indices = c->poolIdx;
		for (u32 L = 0; L < levels && L < 8; ++L) info->active_blocks[L] = c->hdr[HDR_WORK + L];
		info->algorithmic_bytes = (uint64_t)4096 * 3 * cnt[0] + 48ull * c->poolVerts + 4ull * c->poolIdx;
	}
	return VX_OK;
}

int vx_level_counts(vx_ctx* c, uint32_t level, uint32_t* n_blocks, uint64_t totals[4])
{
	VX_ENTER(c);
	if (!c || !c->haveSurface || level >= c->levelsRun) return fail(c, VX_ERR_INVALID, "vx_level_counts: no such level");
	if (ensure_lists(c) != VX_OK) return VX_ERR_DEVICE;
	uint64_t t[4] = { 0, 0, 0, 0 };
	for (const EmittedBlock& e : c->blocks[level]) {
		t[0] += e.rec.vCount; t[1] += e.rec.iCount;
		for (int f = 0; f < 6; ++f) { t[2] += e.rec.tvCount[f]; t[3] += e.rec.tiCount[f]; }
	}
	if (n_blocks) *n_blocks = (u32)c->blocks[level].size();
	if (totals) memcpy(totals, t, sizeof(t));
	return VX_OK;
}

int vx_download_level(vx_ctx* c, uint32_t level, vx_block_info* infos, vx_vertex* verts, uint32_t* idx, vx_vertex* tverts, uint32_t* tidx)
{
	VX_ENTER(c);
	if (!c || !c->haveSurface || level >= c->levelsRun) return fail(c, VX_ERR_INVALID, "vx_download_level: no such level");
	if (ensure_lists(c) != VX_OK) return VX_ERR_DEVICE;
	if (verts || idx || tverts || tidx) {
		// A level that is a small part of the pools (the coarse levels; a caller that wants them alone) is gathered on the device
		// and copied by itself while no host copy of the pools exists; otherwise both pools travel once (fetch_pools) and every
		// level is served from that copy.
		uint64_t lv = 0, li = 0;
		for (const EmittedBlock& e : c->blocks[level]) {
			lv += e.rec.vCount; li += e.rec.iCount;
			for (int f = 0; f < 6; ++f) { lv += e.rec.tvCount[f]; li += e.rec.tiCount[f]; }
		}
		const bool haveCopy = c->hostArena && c->hostArena->lineage == c->poolLineage && c->hostArena->haveVerts >= c->poolVerts && c->hostArena->haveIdx >= c->poolIdx;
		if (!haveCopy && (lv + li) * 8 < (uint64_t)c->poolVerts + c->poolIdx) {
			std::vector<u32> segV, segI; // (source offset, destination offset, count): vertices first, transition vertices behind them; indices likewise
			u32 nv = 0, ntv = 0, ni = 0, nti = 0;
			for (const EmittedBlock& e : c->blocks[level]) { nv += e.rec.vCount; ni += e.rec.iCount; }
			u32 av = 0, atv = nv, ai = 0, ati = ni;
			for (const EmittedBlock& e : c->blocks[level]) {
				const BlockRecord& r = e.rec;
				if (r.vCount) { segV.push_back(r.vOff); segV.push_back(av); segV.push_back(r.vCount); av += r.vCount; }
				if (r.iCount) { segI.push_back(r.iOff); segI.push_back(ai); segI.push_back(r.iCount); ai += r.iCount; }
				for (int f = 0; f < 6; ++f) {
					if (r.tvCount[f]) { segV.push_back(r.tvOff[f]); segV.push_back(atv); segV.push_back(r.tvCount[f]); atv += r.tvCount[f]; ntv += r.tvCount[f]; }
					if (r.tiCount[f]) { segI.push_back(r.tiOff[f]); segI.push_back(ati); segI.push_back(r.tiCount[f]); ati += r.tiCount[f]; nti += r.tiCount[f]; }
				}
			}
			void* dSeg = c->be.alloc((segV.size() + segI.size() + 4) * 4);
			void* dV = c->be.alloc((size_t)(nv + ntv) * sizeof(PolyVertex) + 16);
			void* dI = c->be.alloc((size_t)(ni + nti) * 4 + 16);
			bool ok = dSeg && dV && dI;
			if (ok && !segV.empty()) ok = c->be.h2d(dSeg, segV.data(), segV.size() * 4);
			if (ok && !segI.empty()) ok = c->be.h2d((u32*)dSeg + segV.size(), segI.data(), segI.size() * 4);
			if (ok) {
				c->be.run_copy_segments((const u32*)dSeg, (u32)(segV.size() / 3), c->dVerts, dV, (u32)sizeof(PolyVertex));
				c->be.run_copy_segments((const u32*)dSeg + segV.size(), (u32)(segI.size() / 3), c->dIdx, dI, 4u);
				if (verts && nv) ok = ok && c->be.d2h(verts, dV, (size_t)nv * sizeof(PolyVertex));
				if (tverts && ntv) ok = ok && c->be.d2h(tverts, (const PolyVertex*)dV + nv, (size_t)ntv * sizeof(PolyVertex));
				if (idx && ni) ok = ok && c->be.d2h(idx, dI, (size_t)ni * 4);
				if (tidx && nti) ok = ok && c->be.d2h(tidx, (const u32*)dI + ni, (size_t)nti * 4);
				ok = ok && c->be.sync_ok();
			}
			c->be.free(dSeg); c->be.free(dV); c->be.free(dI);
			if (!ok) return fail(c, VX_ERR_DEVICE, "vx_download_level: gathered download failed: " + c->be.error());
			verts = nullptr; idx = nullptr; tverts = nullptr; tidx = nullptr; // (done; the loop below only fills the block infos)
		} else if (!fetch_pools(c)) return fail(c, VX_ERR_DEVICE, "vx_download_level: pool download failed: " + c->be.error());
	}
	size_t ov = 0, oi = 0, otv = 0, oti = 0, k = 0;
	for (const EmittedBlock& e : c->blocks[level]) {
		const BlockRecord& r = e.rec;
		if (infos) {
			vx_block_info& b = infos[k];
			b.id = e.id; b.n_verts = r.vCount; b.n_idx = r.iCount;
			for (int f = 0; f < 6; ++f) { b.n_tverts[f] = r.tvCount[f]; b.n_tidx[f] = r.tiCount[f]; }
			memcpy(b.min_corner, e.minc, 12); memcpy(b.max_corner, e.maxc, 12);
		}
		++k;
		if (verts && r.vCount) memcpy(verts + ov, c->hostArena->verts + r.vOff, (size_t)r.vCount * 48);
		ov += r.vCount;
		if (idx && r.iCount) memcpy(idx + oi, c->hostArena->idx + r.iOff, (size_t)r.iCount * 4);
		oi += r.iCount;
		for (int f = 0; f < 6; ++f) {
			if (tverts && r.tvCount[f]) memcpy(tverts + otv, c->hostArena->verts + r.tvOff[f], (size_t)r.tvCount[f] * 48);
			otv += r.tvCount[f];
			if (tidx && r.tiCount[f]) memcpy(tidx + oti, c->hostArena->idx + r.tiOff[f], (size_t)r.tiCount[f] * 4);
			oti += r.tiCount[f];
		}
	}
	return VX_OK;
}

int vx_device_meshes(vx_ctx* c, const vx_vertex** dVerts, const uint32_t** dIdx, uint64_t* nVerts, uint64_t* nIdx)
{
	VX_ENTER(c);
	if (!c || !c->haveSurface) return fail(c, VX_ERR_INVALID, "vx_device_meshes: no surface");
	if (dVerts) *dVerts = (const vx_vertex*)c->dVerts;
	if (dIdx) *dIdx = (const uint32_t*)c->dIdx;
	if (nVerts) *nVerts = c->poolVerts;
	if (nIdx) *nIdx = c->poolIdx;
	return VX_OK;
}

int vx_export_meshes(vx_ctx* c, vx_ipc_meshes* out)
{
	VX_ENTER(c);
	if (!c || !c->haveSurface || !out) return fail(c, VX_ERR_INVALID, "vx_export_meshes: no surface");
	memset(out, 0, sizeof(*out));
	static_assert(sizeof(out->verts_handle) == 64 && sizeof(out->indices_handle) == 64, "hipIpcMemHandle_t is 64 bytes");
	if (!c->be.ipc_handle(c->dVerts, out->verts_handle) || !c->be.ipc_handle(c->dIdx, out->indices_handle))
		return fail(c, VX_ERR_DEVICE, "vx_export_meshes: " + c->be.error());
	out->n_verts = c->poolVerts; out->n_indices = c->poolIdx;
	out->verts_capacity = c->vertCap; out->indices_capacity = c->idxCap;
	out->generation = c->poolLineage;
	return VX_OK;
}

int vx_host_meshes_acquire(vx_ctx* c, vx_host_meshes* m)
{
	VX_ENTER(c);
	if (!c || !c->haveSurface || !m) return fail(c, VX_ERR_INVALID, "vx_host_meshes_acquire: no surface");
	HostArena* a = (HostArena*)m->arena;
	if (!a) { a = c->hostArena; c->hostArena = nullptr; } // what vx_download_level may have fetched already is handed over
	const bool ok = arena_fill(c, a);
	m->arena = a;
	m->verts = a ? (const vx_vertex*)a->verts : nullptr;
	m->indices = a ? a->idx : nullptr;
	m->n_verts = a ? a->haveVerts : 0;
	m->n_indices = a ? a->haveIdx : 0;
	return ok ? VX_OK : fail(c, VX_ERR_DEVICE, "vx_host_meshes_acquire: " + (c->be.error().empty() ? std::string("page-locked allocation failed") : c->be.error()));
}

void vx_host_meshes_release(void* arena) { arena_recycle((HostArena*)arena); }

int vx_host_meshes_reserve(vx_ctx* c, uint64_t n_verts, uint64_t n_indices)
{
	VX_ENTER(c);
	if (!c) return VX_ERR_INVALID;
	HostArena* a = arena_get(c, (size_t)n_verts, (size_t)n_indices);
	if (!a) return fail(c, VX_ERR_DEVICE, "vx_host_meshes_reserve: page-locked allocation failed");
	arena_recycle(a);
	return VX_OK;
}

void vx_host_meshes_trim(void)
{
	std::vector<HostArena*> all;
	{
		std::lock_guard<std::mutex> g(g_arenaLock);
		all.swap(g_arenaFree);
	}
	for (HostArena* a : all) arena_destroy(a);
}

int vx_level_ranges(vx_ctx* c, uint32_t level, vx_block_ranges* ranges)
{
	VX_ENTER(c);
	if (!c || !c->haveSurface || level >= c->levelsRun || !ranges) return fail(c, VX_ERR_INVALID, "vx_level_ranges: no such level");
	if (ensure_lists(c) != VX_OK) return VX_ERR_DEVICE;
	size_t k = 0;
	for (const EmittedBlock& e : c->blocks[level]) {
		vx_block_ranges& r = ranges[k++];
		r.v_off = e.rec.vOff; r.i_off = e.rec.iOff;
		for (int f = 0; f < 6; ++f) { r.tv_off[f] = e.rec.tvOff[f]; r.ti_off[f] = e.rec.tiOff[f]; }
	}
	return VX_OK;
}

int vx_device_block_table(vx_ctx* c, uint32_t level, const vx_listed_block** dTable, uint32_t* nBlocks)
{
	VX_ENTER(c);
	static_assert(sizeof(vx_listed_block) == sizeof(ListedBlock), "vx_listed_block layout");
	if (!c || !c->haveSurface || level >= c->levelsRun || !dTable || !nBlocks) return fail(c, VX_ERR_INVALID, "vx_device_block_table: no such level");
	if (!c->deviceLists) {
		// incremental runs and pool compaction edit the lists on the host: bring the device tables up to date
		if (ensure_lists(c) != VX_OK) return VX_ERR_DEVICE;
		for (u32 L = 0; L < c->levelsRun; ++L) {
			const std::vector<EmittedBlock>& b = c->blocks[L];
			if (b.size() > c->lv[L].cap) return fail(c, VX_ERR_DEVICE, "vx_device_block_table: list larger than the level's table");
			if (!b.empty() && !c->be.h2d(c->lv[L].listed, b.data(), b.size() * sizeof(EmittedBlock))) return fail(c, VX_ERR_DEVICE, "vx_device_block_table: upload failed: " + c->be.error());
		}
		c->deviceLists = true;
	}
	*dTable = (const vx_listed_block*)c->lv[level].listed;
	*nBlocks = c->listsReady ? (u32)c->blocks[level].size() : c->hdr[HDR_LISTS + level];
	return VX_OK;
}

#if defined(VX_CASE_DUMP)
// test builds only: the case codes the last full run looked up, per active block of a level (slot order)
int vx_debug_case_dump(vx_ctx* c, uint32_t level, uint32_t cap, uint32_t* coords, uint8_t* cases, uint16_t* trCases, uint32_t* count)
{
	VX_ENTER(c);
	if (!c || !c->haveSurface || level >= c->levelsRun || !count) return VX_ERR_INVALID;
	*count = c->hdr[level];
	if (*count > cap) return VX_OK;
	const LevelDesc& d = c->lv[level];
	bool ok = true;
	if (*count) {
		ok = c->be.d2h(coords, d.slotCoord, (size_t)*count * 4) && c->be.d2h(cases, d.caseDump, (size_t)*count * BLOCK_CELLS)
		  && c->be.d2h(trCases, d.trCaseDump, (size_t)*count * TR_CELLS * 2);
	}
	return ok ? VX_OK : fail(c, VX_ERR_DEVICE, "vx_debug_case_dump: download failed");
}
#endif

int vx_selftest(vx_ctx* c, uint32_t results[16])
{
	VX_ENTER(c);
	if (!c || !results) return VX_ERR_INVALID;
	void* d = c->be.alloc(16 * 4);
	const bool ok = d && c->be.run_selftest((u32*)d) && c->be.d2h(results, d, 16 * 4);
	c->be.free(d);
	return ok ? VX_OK : fail(c, VX_ERR_DEVICE, "vx_selftest: " + c->be.error());
}

int vx_set_stage_timing(vx_ctx* c, int enable)
{
	VX_ENTER(c);
	if (!c) return VX_ERR_INVALID;
	c->be.stage_enable(enable != 0);
	return VX_OK;
}

int vx_stage_layout(vx_ctx* c, int* layout)
{
	if (!c || !layout) return VX_ERR_INVALID;
	*layout = c->stagedMain ? 1 : 0;
	return VX_OK;
}

int vx_debug_header(vx_ctx* c, uint32_t* out, uint32_t count)
{
	if (!c || !out || count > (uint32_t)HDR_WORDS) return VX_ERR_INVALID;
	return c->be.d2h_side(out, c->dHeader, (size_t)count * 4) ? VX_OK : VX_ERR_DEVICE;
}

int vx_stage_times(vx_ctx* c, float ms[8]) /* reset, classify, hierarchy, material, regular level 0, regular levels >= 1, transition, block lists */
{
	VX_ENTER(c);
	if (!c || !ms) return VX_ERR_INVALID;
	return c->be.stage_ms(ms) ? VX_OK : fail(c, VX_ERR_INVALID, "vx_stage_times: stage timing was not enabled for the last run");
}

int vx_stats(vx_ctx* c, uint32_t stats[20])
{
	VX_ENTER(c);
	if (!c || !c->haveSurface) return fail(c, VX_ERR_INVALID, "vx_stats: nothing polygonized yet");
	memcpy(stats, c->stats, 80);
	return VX_OK;
}

} // extern "C"

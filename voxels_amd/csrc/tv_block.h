// tv_block.h — per-block phases of the polygonizer, written as "for (i = tid; i < count; i += nthreads)" loops
// over workgroup-shared state.  On the GPU the state lives in LDS, tid = threadIdx.x and the kernels in
// vx_hip.hip put __syncthreads() + wavefront scans between the phases; tests/emu runs the same phases with
// tid = 0, nthreads = 1 on a CPU to check the parallel formulation against the oracle.
//
// Pipeline (one polygonization):
//   classify (stream, all level-0 blocks)  -> non-trivial bitmaps + active block slots per level
//   material (levels 1..Lmax, serial)      -> per-cell material cache (vote over children), level bitmaps
//   regular  (all levels at once)          -> vertices + indices of regular cells
//   transition (levels 1..last-1)          -> vertices + indices of the 6 transition faces
#pragma once

#include "tv_core.h"

#if defined(__HIPCC__)
#define TV_ATOMIC_ADD(p, v) atomicAdd((p), (v))
#define TV_ATOMIC_OR(p, v) atomicOr((p), (v))
#define TV_POPC(x) __popc(x)
#else
namespace tv {
template <typename T> inline T host_atomic_add(T* p, T v) { T o = *p; *p = o + v; return o; }
template <typename T> inline T host_atomic_or(T* p, T v) { T o = *p; *p = o | v; return o; }
}
#define TV_ATOMIC_ADD(p, v) tv::host_atomic_add((p), (v))
#define TV_ATOMIC_OR(p, v) tv::host_atomic_or((p), (v))
#define TV_POPC(x) __builtin_popcount(x)
#endif

// A load of data that another workgroup of the SAME launch may have written (k_main: material caches, published with
// write-through stores): past the CU's L1, which no other CU's store ever refreshes - an agent-scope relaxed atomic load
// is an `sc1` load on gfx950.  A plain load elsewhere (the CPU emulation; data of earlier launches).
#if defined(__HIP_DEVICE_COMPILE__)
#define TV_LOAD_THROUGH(p) __hip_atomic_load((p), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT)
#else
#define TV_LOAD_THROUGH(p) (*(p))
#endif

namespace tv {

enum { MAX_LEVELS = 8, BLOCK_CELLS = 4096, SAMPLES = 17 * 17 * 17, PLANE = 33 * 33 };
enum { CUR_V = 0, CUR_I = 1, CUR_OVF = 64 }; // the two pool cursors share an aligned 64-bit word: one atomic reserves both ranges of a mesh
enum { LIST_WG = 256 }; // block coordinates per workgroup of the list kernels
enum { LARGE_THRESHOLD = 640 }; // blocks with more non-trivial cells use the 4096-cell LDS class of the regular pass // hot device counters live in separate cache lines (atomics serialise per line)

// One emitted block (regular mesh + 6 transition meshes) inside the shared vertex/index pools
struct BlockRecord {
	u32 coordId;
	u32 vOff, vCount, iOff, iCount;
	u32 tvOff[6], tvCount[6], tiOff[6], tiCount[6];
	u32 degenerate;
	u32 ntCells;
	u32 pad;
};

// A block as the result lists it (PushBlocksToResult, src/TransVoxelImpl.cpp:1266-1293): where its meshes are in the
// pools, its id (:149-152: all blocks of all levels numbered level-major in coordinate order) and its corners (:1283-1293,
// output axes: Y up).  One table per level, in coordinate order (= PolygonSurface::GetBlockForLevel order), blocks
// without a regular vertex left out (:1274) — written on the device at the end of a full run.
struct ListedBlock {
	BlockRecord rec;
	u32 id;
	float minc[3], maxc[3];
};

// Per-LOD-level device tables
struct LevelDesc {
	u32 cnt;            // blocks per axis of this level (global)
	u32 mult;           // cell size in voxels
	u32 zb0, zb1;       // block layers [zb0, zb1) owned by this rank
	u32 yb0, yb1;       // block rows [yb0, yb1) owned by this rank (slabs are cut along z or along y)
	int* slotOf;        // [cnt^3] block coordinate id -> active slot, -1 = none
	u32* slotCoord;     // [cap] slot -> coordinate id
	u32* nActive;       // number of slots in use
	u32* listCounts;    // [ceil(cnt^3 / LIST_WG)] blocks with a regular mesh per LIST_WG block coordinates: counted where a block's
	                    //          record is written, consumed by the list kernel of a full run (zeroed with the run's counters)
	u32* ntBits;        // [cap][128] non-trivial cell bitmap of the block (current geometry)
	u32* consBits;      // level 0: [cap][128] the reference's Level0ConsistencyCache: bits of every cell that was ever
	                    //          polygonized as non-trivial (only set, never cleared, TransVoxelImpl.cpp:757)
	u16* cache;         // [cap][4096] per-cell material cache (levels >= 1)
	u8* skip;           // [cap] level 0: block skipped by the emptiness rule
	u16* ntCount;       // [cap] number of non-trivial cells of the block (picks the LDS capacity class)
	BlockRecord* records; // [cap]
	ListedBlock* listed;  // [cap] the level's block list (see ListedBlock)
	unsigned long long* matDone; // [cap] levels >= 1, full runs of the GPU backend: Globals::epoch << 32 | non-trivial cells once the block's
	                      //       material cache, bitmap and cell count are in memory (the dependency flags of k_main: one 8-byte word per block,
	                      //       written with one store, polled by whoever needs the block - its parent's vote, its own regular and transition cells)
#if defined(VX_CASE_DUMP)
	// test builds only (libvoxels_hip_casedump.so): the case code of every cell the passes looked up in Lengyel's tables,
	// for tests/test_case_codes.py (case codes are not part of the public result)
	u8* caseDump;         // [cap][4096] regular cells: CalcCaseCode of the non-trivial cells, 0 elsewhere
	u16* trCaseDump;      // [cap][6 * 256] transition cells: the 9-bit code of the non-trivial ones, 0 elsewhere
#endif
	u32 cap;
	u32 hasTransitions; // 0 < level < levelsCount - 1
};

struct Pools {
	PolyVertex* verts;
	u32* idx;
	u32* cursors;       // [CUR_V] vertices used, [CUR_I] indices used (one 64-bit word), [CUR_OVF] overflow flag (its own 128-byte line)
	u32 vertCap, idxCap;
};

// The distance samples of LOD level L (the lattice of voxels whose coordinates are multiples of 2^L) as an array of
// their own: entry (X,Y,Z) = dist(min(X << L, n-1), min(Y << L, n-1), min(Z << L, n-1)), X,Y,Z in [0, n >> L] — the last
// index is the reference's clamped far sample.  A mirror of the grid like the bricks: written where the grid changes
// (k_rebrick, the unpack of a halo exchange), complete over the rank's range, never touched by a polygonization;
// levels >= 1 read 17 contiguous bytes per sample row instead of 17 bytes that are 2^L apart.  Same brick
// layout as the grid's mirrors (tv_core.h brick_local): the 16^3 lattice samples of a level-L block are 4 KB of
// consecutive addresses; the far samples live in one more brick along every axis.
enum { PYRAMID_LEVELS = 7 }; // levels 1..6 have a lattice copy (every level of a grid up to 1024^3: the table-driven passes cover them all; round 4 stopped at 3 and left the coarser levels - few blocks, but 100 us each in the general pass - to gather from the grid)
struct PyramidLevel {
	i8* data;             // nullptr: no copy of this level
	u32 bricksX, bricksY; // bricks per row of bricks, rows of bricks per plane of bricks
	int yOrigin, zOrigin; // lattice coordinates of row 0 / plane 0 (slabs)
};

TV_HD size_t pyramid_offset(const PyramidLevel& P, int X, int Y, int Z)
{
	const u32 y = (u32)(Y - P.yOrigin), z = (u32)(Z - P.zOrigin);
	return (((size_t)(z >> 4) * P.bricksY + (size_t)(y >> 4)) * P.bricksX + (size_t)((u32)X >> 4)) * BRICK_BYTES + brick_local((u32)X & 15u, y & 15u, z & 15u);
}

// The yz-planes X = 32 k of the level-l lattice (l = 0..2: the planes the x faces of the transition pass of level l + 1
// lie in; k = 0 .. (n >> l) / 32, the last one being the clamped far plane), bytes [k][z][y] with y fastest, z and y over
// 0 .. n >> l (far entries included).  In brick order such a plane costs one access per sample (one byte of every
// 16-byte voxel row); here a face row is 33 contiguous bytes like the rows of the other four faces.  A fourth mirror,
// written by the same kernels as the lattice copies.
enum { XPLANE_LEVELS = 3 };
struct XPlanes {
	i8* data;          // nullptr: no copy of this level
	u32 rows, stride;  // entries along z; bytes per row (a multiple of 16, >= entries along y)
};
TV_HD size_t xplane_offset(const XPlanes& P, u32 plane, u32 Z, u32 Y) { return ((size_t)plane * P.rows + Z) * P.stride + Y; }

struct __attribute__((aligned(16))) FlatItem { u32 where, coordId, ntCells, pad; }; // Globals::flatItems

// stats[0] = non-trivial cells, [1] = degenerate triangles removed, [2] = level-0 blocks processed,
// stats[4..19] = per-class cell counts
struct Globals {
	GridView grid;
	const u8* emptyFlags;   // [cnt0^3] BF_Empty of every level-0 block (host-computed property of the Grid)
	const u8* lut;          // 256 x 8 material LUT
	u32* stats;
	u32 levels;             // number of levels being polygonized (0..levels-1)
	u32 refLevels;          // the reference's levelsCount = log2(N/16)+1 (decides which levels get transitions)
	// incremental (Modification) runs: only the listed blocks are re-polygonized, caches keep their old contents
	u32 dirty;                        // 0 = full run
	const u32* workItems[MAX_LEVELS]; // dirty: active slots to process per level
	u32* workCount;                   // dirty: [MAX_LEVELS] number of items per level
	u32 prevActive[MAX_LEVELS];       // dirty: slots >= prevActive[level] were created by this run
	u32* largeBlocks;                 // number of classified blocks whose non-trivial cell count exceeds LARGE_THRESHOLD
	// scratch, rebuilt by every full run from emptyFlags + one sample per empty block (level-0 blocks, [cnt0^3]):
	u8* blockClass;                   // BC_* bits: what the classify pass may assume without reading the block
	const u16* blockSign;             // per level-0 block, kept with the grid's mirrors: eight 2-bit sign summaries (MirrorState)
	PyramidLevel pyr[PYRAMID_LEVELS]; // [1..]: lattice copies of the distance field for the coarser levels (GPU backend)
	XPlanes xp[XPLANE_LEVELS];        // yz-planes of the lattices 0..2 at every 32nd x (the x faces of the transition pass)
	// full runs: blocks the fast regular passes (tv_fast0.h, tv_fast1.h) hand on to the general pass (a zero sample, a LOD
	// chain ending on a voxel).  [0]: level-0 slots; [1]: level << 24 | slot for the levels >= 1
	u32* slowItems[2];
	u32* slowCount;                   // [2]
	// full runs: the active blocks of the levels >= 1 as one list in level order, written by the material pass (which
	// visits every one of them): { level << 24 | slot, block coordinate id, non-trivial cells, 0 }.  The passes over
	// those blocks index it by work item instead of turning the item into (level, slot) and then fetching the slot's
	// coordinate and cell count - one dependent round trip less at the head of every block.
	struct FlatItem* flatItems;
	const u32* slotCounts;            // [MAX_LEVELS] active blocks per level (LevelDesc::nActive of all levels, contiguous)
	// full runs of the GPU backend: everything behind the classification is ONE launch (k_main) whose upper work items - material blocks of the levels
	// 1, 2, ... in that order, then their regular and transition blocks - are dequeued in order and wait for what they
	// depend on through LevelDesc::matDone
	u32 epoch;                        // this run's tag in matDone (never 0)
	u32* upperHead;                   // queue head (zeroed with the run's counters)
	u32* level0Head;                  // head of the queue of level-0 slots of the same launch
	u32* giveUp;                      // set when a dependency wait ran out of patience (a bug, never data-dependent): the host fails the run
};

// BF_Empty (VoxelGrid.cpp:455-476 / CompressBlock) means: every sample of the block is non-zero and has the sign of the
// first one.  A block whose 27-neighbourhood is BF_Empty is skipped by the reference (TransVoxelImpl.cpp:520).  Apart
// from that: if a block and the parts of the seven blocks its cells reach into (the first plane / line / voxel of the +x,
// +y, +z neighbours) are of one and the same sign (blockSign), no cell of the block can be non-trivial, so the classify
// pass does not have to read it ("quiet").

// What the mirrors of a grid carry beside the bricks (handed to the kernels that keep them current)
struct MirrorState {
	PyramidLevel pyr[PYRAMID_LEVELS];
	XPlanes xp[XPLANE_LEVELS];
	u16* blockSign;                 // per level-0 block: field f = dx | dy << 1 | dz << 2 (2 bits each) summarises the voxels with x = 0 (dx), y = 0 (dy), z = 0 (dz): 1 = all >= 0, 2 = all < 0, 0 = mixed or not all resident
	int yBegin, yEnd, zBegin, zEnd; // the rank's own rows / planes
};
enum { BC_SKIPPED = 1, BC_QUIET = 2, BC_NEGATIVE = 4 };

TV_HD u32 block_coord_id(u32 bx, u32 by, u32 bz, u32 cnt) { return (bz * cnt + by) * cnt + bx; }

// One piece of a halo message: `layers` consecutive layers of a voxel field (n x n bytes each) or one block layer of
// the flag array (cnt x cnt bytes), between the field and a contiguous staging buffer.  Element (layer l, a, x) of a
// field lives at ((l - origin) * strideLayer + a * strideA) * rowBytes + x: for slabs of z-planes strideLayer = rows per
// plane and strideA = 1; for slabs of y-rows strideLayer = 1 and strideA = rows per plane.
struct HaloPiece {
	u8* field;
	u32 stagingOffset;   // bytes
	int firstLayer, layers, origin;
	u32 strideLayer, strideA;
	u32 rowBytes;        // n for voxel fields, cnt for flags
	u32 rows;            // a runs over [0, rows): n or cnt
};
enum { HALO_MAX_PIECES = 4 };
struct HaloMove {
	HaloPiece piece[HALO_MAX_PIECES];
	u32 count;
	u8* staging;
	u32 unpack;          // 0: field -> staging, 1: staging -> field
};

TV_HD size_t halo_piece_bytes(const HaloPiece& p) { return (size_t)p.layers * p.rows * p.rowBytes; }

// byte offset inside the field of (layer l, row a), x = 0
TV_HD size_t halo_field_offset(const HaloPiece& p, int l, u32 a) { return ((size_t)(l - p.origin) * p.strideLayer + (size_t)a * p.strideA) * p.rowBytes; }

// the work of the list kernels: workgroup w handles the block coordinates [(w - wgStart[l]) * LIST_WG, ...) of its level l
struct ListPlan {
	u32 wgStart[MAX_LEVELS + 1];
	u32 idBase[MAX_LEVELS];   // id of block coordinate 0 of the level
	u32* counts;              // [wgStart[levels]] listed blocks per workgroup (scratch)
	u32* totals;              // [MAX_LEVELS] out: listed blocks per level
};

TV_HD void listed_block_fill(ListedBlock& out, const LevelDesc& L, u32 coordId, u32 slot, u32 idBase)
{
	out.rec = L.records[slot];
	out.id = idBase + coordId;
	u32 bx, by, bz;
	bx = coordId % L.cnt; by = (coordId / L.cnt) % L.cnt; bz = coordId / (L.cnt * L.cnt);
	const float ext = (float)(L.mult * 16);
	out.minc[0] = (float)bx * ext; out.minc[1] = (float)bz * ext; out.minc[2] = (float)by * ext; // output is Y-up
	out.maxc[0] = out.minc[0] + ext; out.maxc[1] = out.minc[1] + ext; out.maxc[2] = out.minc[2] + ext;
}

// does block coordinate `id` of level L appear in the level's list?  (-1: no, else its slot)
// a block whose record says it has a regular mesh will be listed (listed_block_slot): its group of LIST_WG coordinates counts it
TV_HD void count_listed_block(const LevelDesc& L, u32 coordId, u32 vCount)
{
	if (vCount && L.listCounts) TV_ATOMIC_ADD(&L.listCounts[coordId / LIST_WG], 1u);
}

TV_HD int listed_block_slot(const LevelDesc& L, u32 id)
{
	if (id >= L.cnt * L.cnt * L.cnt) return -1;
	const int slot = L.slotOf[id];
	if (slot < 0) return -1;
	return L.records[slot].vCount ? slot : -1;
}

TV_HD void block_coords(u32 id, u32 cnt, u32& bx, u32& by, u32& bz) { bx = id % cnt; by = (id / cnt) % cnt; bz = id / (cnt * cnt); }

// ---------------------------------------------------------------------------------------------------------
// shared helpers
// ---------------------------------------------------------------------------------------------------------
// corner samples of a block: sample (i,j,k), 0..16, at global position (block*16 + ijk) * mult
TV_HD void stage_samples(const GridView& g, u32 bx, u32 by, u32 bz, u32 mult, i8* samp, int tid, int nth)
{
	for (int s = tid; s < SAMPLES; s += nth) {
		const int i = s % 17, j = (s / 17) % 17, k = s / 289;
		samp[s] = (i8)dist_at(g, (int)((bx * 16 + i) * mult), (int)((by * 16 + j) * mult), (int)((bz * 16 + k) * mult));
	}
}

TV_HD void cell_values(const i8* samp, int cx, int cy, int cz, i8 V[8])
{
	const int o = (cz * 17 + cy) * 17 + cx;
	V[0] = samp[o]; V[1] = samp[o + 1]; V[2] = samp[o + 17]; V[3] = samp[o + 18];
	V[4] = samp[o + 289]; V[5] = samp[o + 290]; V[6] = samp[o + 306]; V[7] = samp[o + 307];
}

TV_HD u32 bit_get(const u32* bits, u32 i) { return (bits[i >> 5] >> (i & 31)) & 1u; }

// rank of set bit i in a bitmap with exclusive per-word popcount prefix
TV_HD u32 bit_rank(const u32* bits, const u16* prefix, u32 i)
{
	return prefix[i >> 5] + TV_POPC(bits[i >> 5] & ((1u << (i & 31)) - 1u));
}

// ---------------------------------------------------------------------------------------------------------
// Classification of level-0 blocks (portable form; the GPU kernel uses a bit-parallel streaming version)
// ---------------------------------------------------------------------------------------------------------
TV_HD bool block_skipped_by_emptiness(const u8* emptyFlags, u32 cnt, u32 bx, u32 by, u32 bz)
{
	for (int z = -1; z < 2; ++z)
	for (int y = -1; y < 2; ++y)
	for (int x = -1; x < 2; ++x) {
		const u32 cx = (u32)clampi((int)bx + x, 0, (int)cnt - 1), cy = (u32)clampi((int)by + y, 0, (int)cnt - 1), cz = (u32)clampi((int)bz + z, 0, (int)cnt - 1);
		if (!emptyFlags[block_coord_id(cx, cy, cz, cnt)]) return false;
	}
	return true;
}

// Publish the classification of one level-0 block: slot (existing, or new when the block has non-trivial cells),
// current bitmap, emptiness-skip flag, accumulated consistency bits.  `bits` = 128 words, one caller per block.
TV_HD void publish_level0_block(const Globals& G, const LevelDesc& L, u32 bx, u32 by, u32 bz, const u32* bits, u32 ntCells, bool accumulate)
{
	const bool skipped = block_skipped_by_emptiness(G.emptyFlags, L.cnt, bx, by, bz);
	if (!skipped) TV_ATOMIC_ADD(&G.stats[2], 1u);
	const u32 id = block_coord_id(bx, by, bz, L.cnt);
	int slot = L.slotOf[id];
	bool fresh = false;
	if (slot < 0) {
		if (!ntCells) return;
		slot = (int)TV_ATOMIC_ADD(L.nActive, 1u);
		L.slotOf[id] = slot;
		L.slotCoord[slot] = id;
		fresh = true;
	}
	L.skip[slot] = skipped ? 1 : 0;
	L.ntCount[slot] = (u16)ntCells;
	if (ntCells > LARGE_THRESHOLD) TV_ATOMIC_ADD(G.largeBlocks, 1u);
	u32* nt = L.ntBits + (size_t)slot * 128;
	u32* cons = L.consBits + (size_t)slot * 128;
	for (int w = 0; w < 128; ++w) {
		nt[w] = bits[w];
		const u32 add = skipped ? 0u : bits[w];
		cons[w] = (accumulate && !fresh) ? (cons[w] | add) : add;
	}
}

// ---------------------------------------------------------------------------------------------------------
// Material pass state (levels >= 1)
// ---------------------------------------------------------------------------------------------------------
struct MatState {
	i8 samp[SAMPLES + 7];
	u32 ntBits[128];
	int childSlot[8];          // active slot of the 2x2x2 child blocks (level - 1), -1 = none / outside / skipped
	u32 childBits[8][128];     // level 1 only: non-trivial bitmaps of the child blocks (the level-0 consistency cache)
	u16 voteList[BLOCK_CELLS]; // cells that have at least one child entry to look at (compact, any order)
	u32 voteCount;
};

// stage what the vote needs to know about the 8 child blocks
TV_HD void mat_phase_children(MatState& st, const LevelDesc* levels, u32 level, u32 bx, u32 by, u32 bz, int tid, int nth)
{
	const LevelDesc& C = levels[level - 1];
	for (int i = tid; i < 8; i += nth) {
		const u32 cx = bx * 2 + (i & 1), cy = by * 2 + ((i >> 1) & 1), cz = bz * 2 + (i >> 2);
		int slot = -1;
		if (cx < C.cnt && cy < C.cnt && cz < C.cnt) slot = C.slotOf[block_coord_id(cx, cy, cz, C.cnt)];
		st.childSlot[i] = slot;
	}
	if (level == 1) {
		for (int q = tid; q < 8 * 128; q += nth) {
			const int i = q >> 7;
			const u32 cx = bx * 2 + (i & 1), cy = by * 2 + ((i >> 1) & 1), cz = bz * 2 + (i >> 2);
			u32 v = 0;
			if (cx < C.cnt && cy < C.cnt && cz < C.cnt) {
				const int slot = C.slotOf[block_coord_id(cx, cy, cz, C.cnt)];
				if (slot >= 0) v = C.consBits[(size_t)slot * 128 + (q & 127)];
			}
			st.childBits[i][q & 127] = v;
		}
	}
}

TV_HD void mat_phase_classify(MatState& st, int tid, int nth)
{
	for (int c = tid; c < BLOCK_CELLS; c += nth) {
		i8 V[8];
		cell_values(st.samp, c & 15, (c >> 4) & 15, c >> 8, V);
		const u32 code = reg_case_code(V);
		if (code != 0 && code != 255) TV_ATOMIC_OR(&st.ntBits[c >> 5], 1u << (c & 31));
	}
}

// does the transition pass of this block visit cell (lx,ly,lz)?  (boundary cell on a face with a neighbour block)
TV_HD bool cell_on_transition_face(const LevelDesc& L, u32 bx, u32 by, u32 bz, int lx, int ly, int lz)
{
	if (!L.hasTransitions) return false;
	return (lz == 0 && bz > 0) || (ly == 0 && by > 0) || (lx == 0 && bx > 0)
	    || (lz == 15 && bz + 1 < L.cnt) || (ly == 15 && by + 1 < L.cnt) || (lx == 15 && bx + 1 < L.cnt);
}

// child cell i (x fastest) of cell (lx,ly,lz): child block index (0..7) and cell id inside it
TV_HD void child_location(int lx, int ly, int lz, u32 i, u32& cb, u32& local)
{
	const u32 ccx = (u32)lx * 2 + (i & 1), ccy = (u32)ly * 2 + ((i >> 1) & 1), ccz = (u32)lz * 2 + (i >> 2);
	cb = (ccx >> 4) | ((ccy >> 4) << 1) | ((ccz >> 4) << 2);
	local = ((ccz & 15) << 8) | ((ccy & 15) << 4) | (ccx & 15);
}

// Pass 1 (LDS only): which cells need a vote and have anything to vote on; everything else is final already
TV_HD void mat_phase_select(MatState& st, const Globals& G, const LevelDesc* levels, u32 level, u32 slot,
                            u32 bx, u32 by, u32 bz, int tid, int nth)
{
	const LevelDesc& L = levels[level];
	u16* out = L.cache + (size_t)slot * BLOCK_CELLS;
	const bool defineAll = !G.dirty || slot >= G.prevActive[level];
	for (int c = tid; c < BLOCK_CELLS; c += nth) {
		const int lx = c & 15, ly = (c >> 4) & 15, lz = c >> 8;
		bool candidate = false;
		if (bit_get(st.ntBits, (u32)c) || cell_on_transition_face(L, bx, by, bz, lx, ly, lz)) {
			for (u32 i = 0; i < 8 && !candidate; ++i) {
				u32 cb, local;
				child_location(lx, ly, lz, i, cb, local);
				if (st.childSlot[cb] < 0) continue;
				candidate = (level == 1) ? bit_get(st.childBits[cb], local) != 0 : true;
			}
		}
		if (candidate) st.voteList[TV_ATOMIC_ADD(&st.voteCount, 1u)] = (u16)c;
		else if (defineAll) out[c] = (u16)EMPTY_MATINFO; // incremental runs keep the old entry (TransVoxelImpl.cpp:829)
	}
	u32* bitsOut = L.ntBits + (size_t)slot * 128;
	for (int w = tid; w < 128; w += nth) bitsOut[w] = st.ntBits[w];
	if (tid == 0) {
		u32 cnt = 0;
		for (int w = 0; w < 128; ++w) cnt += TV_POPC(st.ntBits[w]);
		L.ntCount[slot] = (u16)cnt;
		if (cnt > LARGE_THRESHOLD) TV_ATOMIC_ADD(G.largeBlocks, 1u);
	}
}

// Pass 2: dense over the selected cells, the eight child fetches of a cell in flight together
TV_HD void mat_phase_vote(const MatState& st, const Globals& G, const LevelDesc* levels, u32 level, u32 slot,
                          u32 bx, u32 by, u32 bz, int tid, int nth)
{
	const LevelDesc& L = levels[level];
	const LevelDesc& C = levels[level - 1];
	u16* out = L.cache + (size_t)slot * BLOCK_CELLS;
	const bool defineAll = !G.dirty || slot >= G.prevActive[level];
	const int n = (int)st.voteCount;
	for (int k = tid; k < n; k += nth) {
		const int c = st.voteList[k];
		const int lx = c & 15, ly = (c >> 4) & 15, lz = c >> 8;
		const u32 entry = vote_material([&](u32 i) -> u32 {
			u32 cb, local;
			child_location(lx, ly, lz, i, cb, local);
			const int cslot = st.childSlot[cb];
			if (cslot < 0) return EMPTY_MATINFO;
			if (level == 1) {
				if (!bit_get(st.childBits[cb], local)) return EMPTY_MATINFO;
				return mat_at(G.grid, (int)((bx * 16 + lx) * 2 + (i & 1)), (int)((by * 16 + ly) * 2 + ((i >> 1) & 1)), (int)((bz * 16 + lz) * 2 + (i >> 2)));
			}
			return C.cache[(size_t)cslot * BLOCK_CELLS + local];
		});
		if (defineAll || entry != EMPTY_MATINFO) out[c] = (u16)entry;
	}
}

// ---------------------------------------------------------------------------------------------------------
// Regular-cell pass
//
// LDS sample layout: 19 x 19 rows of 24 bytes; sample (i,j,k), i,j,k in [-1,17] relative to the block origin (in
// cells of the block's level) lives at (k+1)*SPLANE + (j+1)*SROW + (i+4), so every row starts 4-byte aligned at
// local x = -4 and level-0 rows are filled with six aligned dword loads.  Level 0 stages the full [-1,17]^3
// neighbourhood (central-difference normals then never leave LDS); levels >= 1 stage the 17^3 corner samples only
// (their normals / LOD chain read level-0 voxels from HBM).
// ---------------------------------------------------------------------------------------------------------
enum { SROW = 24, SPLANE = 19 * SROW, SAMP_BYTES = 19 * SPLANE, VDESC_CAP = 1024 };

TV_HD int samp_index(int i, int j, int k) { return (k + 1) * SPLANE + (j + 1) * SROW + (i + 4); }

TV_HD void reg_stage_level0(const GridView& g, u32 bx, u32 by, u32 bz, i8* samp, int tid, int nth)
{
	const int n = g.n, gx0 = (int)bx * 16 - 4;
	for (int q = tid; q < 361 * 6; q += nth) {
		const int r = q / 6, j = q - r * 6;
		const int jj = r % 19, kk = r / 19;
		const int y = clampi((int)by * 16 + jj - 1, 0, n - 1);
		const int z = clampi((int)bz * 16 + kk - 1, 0, n - 1);
		const int x = clampi(gx0 + 4 * j, 0, n - 4);
		const u32 v = *(const u32*)(g.dist + dist_offset(g, x, y, z));
		*(u32*)(samp + kk * SPLANE + jj * SROW + 4 * j) = v;
	}
}

// x-clamp of the level-0 rows of border blocks (the dword loads above clamp whole dwords, not samples)
TV_HD void reg_fix_borders(u32 bx, u32 cnt, i8* samp, int tid, int nth)
{
	if (bx == 0) for (int r = tid; r < 361; r += nth) { i8* row = samp + (r / 19) * SPLANE + (r % 19) * SROW; row[3] = row[4]; }
	if (bx + 1 == cnt) for (int r = tid; r < 361; r += nth) { i8* row = samp + (r / 19) * SPLANE + (r % 19) * SROW; row[20] = row[19]; row[21] = row[19]; }
}

TV_HD void reg_stage_strided(const GridView& g, u32 bx, u32 by, u32 bz, u32 mult, i8* samp, int tid, int nth)
{
	for (int s = tid; s < SAMPLES; s += nth) {
		const int i = s % 17, j = (s / 17) % 17, k = s / 289;
		samp[samp_index(i, j, k)] = (i8)dist_at(g, (int)((bx * 16 + i) * mult), (int)((by * 16 + j) * mult), (int)((bz * 16 + k) * mult));
	}
}

TV_HD void reg_cell_values(const i8* samp, int cx, int cy, int cz, i8 V[8])
{
	const i8* p = samp + samp_index(cx, cy, cz);
	V[0] = p[0]; V[1] = p[1]; V[2] = p[SROW]; V[3] = p[SROW + 1];
	V[4] = p[SPLANE]; V[5] = p[SPLANE + 1]; V[6] = p[SPLANE + SROW]; V[7] = p[SPLANE + SROW + 1];
}

// level-0 distance sampler over the staged neighbourhood (global voxel coordinates in, like GlobalDist)
struct LocalDist {
	const i8* samp;
	int ox, oy, oz;
	TV_HD int operator()(int x, int y, int z) const { return samp[samp_index(x - ox, y - oy, z - oz)]; }
};

template <int CAP>
struct RegStateT {
	i8 samp[SAMP_BYTES + 8];
	u32 ntBits[128];
	u16 wordPrefix[130];      // exclusive popcount prefix, [128] = number of non-trivial cells
	u16 cellOf[CAP];          // compact index -> cell id
	u16 cellMat[CAP];         // compact: id | blend << 8
	u16 cellBits[CAP];        // compact: case code | (corner sample == 0) mask << 8 — all that the index logic needs of the samples
	u32 info[CAP];            // compact: bits 0-15 slot ordinals (4 x 4), 16-19 slot valid, 20-23 new vertex count, 24-28 kept triangles
	u16 vbase[CAP];           // compact: exclusive scan of new vertex counts
	u16 ibase[CAP];           // compact: exclusive scan of kept index counts
	u16 newMask[CAP];         // compact: table vertices this cell creates
	u16 atV0Mask[CAP];        // compact: created corner vertices placed at edge corner v0 (TransVoxelImpl.cpp:1637)
	u16 invalidMask[CAP];     // compact: table vertices that resolve to INVALID_INDEX
	u16 vdesc[VDESC_CAP];     // one chunk of new-vertex descriptors: compact cell | table vertex << 12
	u32 suspect[128];         // per cell id: touches a vertex that is not strictly inside its edge (may be degenerate)
	u32 perCase[16];
	u32 vOff, iOff, vTotal, iTotal, degenerate;
};

struct RegBlockCtx {
	u32 level, slot, bx, by, bz, mult;
};

template <typename ST>
TV_HD void reg_phase_begin(ST& st, const LevelDesc& L, u32 slot, int tid, int nth)
{
	const u32* src = L.ntBits + (size_t)slot * 128;
	for (int w = tid; w < 128; w += nth) st.ntBits[w] = src[w];
	for (int i = tid; i < 16; i += nth) st.perCase[i] = 0;
	for (int w = tid; w < 128; w += nth) st.suspect[w] = 0;
	if (tid == 0) st.degenerate = 0;
}

template <typename ST>
TV_HD void reg_phase_stage(ST& st, const Globals& G, const LevelDesc& L, const RegBlockCtx& b, int tid, int nth)
{
	if (b.level == 0) reg_stage_level0(G.grid, b.bx, b.by, b.bz, st.samp, tid, nth);
	else reg_stage_strided(G.grid, b.bx, b.by, b.bz, b.mult, st.samp, tid, nth);
}

// after wordPrefix is known: compact list of the non-trivial cells, in cell order (one 16-cell row per step)
template <typename ST>
TV_HD void reg_phase_list(ST& st, const LevelDesc& L, const RegBlockCtx& b, int tid, int nth)
{
	if (b.level == 0) reg_fix_borders(b.bx, L.cnt, st.samp, tid, nth);
	for (int row = tid; row < 256; row += nth) {
		u32 bits = (st.ntBits[row >> 1] >> ((row & 1) * 16)) & 0xFFFFu;
		if (!bits) continue;
		u32 k = bit_rank(st.ntBits, st.wordPrefix, (u32)row * 16);
		while (bits) {
			const u32 x = (u32)__builtin_ctz(bits);
			bits &= bits - 1;
			st.cellOf[k++] = (u16)(row * 16 + x);
		}
	}
}

// dense over the compact list: cell material, slot-valid mask, per-class statistics
template <typename ST>
TV_HD void reg_phase_cells(ST& st, const Tables& T, const Globals& G, const LevelDesc& L, const RegBlockCtx& b, int tid, int nth)
{
	const int nt = st.wordPrefix[128];
	for (int k = tid; k < nt; k += nth) {
		const int c = st.cellOf[k];
		const int cx = c & 15, cy = (c >> 4) & 15, cz = c >> 8;
		i8 V[8];
		reg_cell_values(st.samp, cx, cy, cz, V);
		const u32 code = reg_case_code(V);
		const u32 zeroMask = reg_zero_mask(V);
		st.cellBits[k] = (u16)(code | (zeroMask << 8));
#if defined(VX_CASE_DUMP)
		L.caseDump[(size_t)b.slot * BLOCK_CELLS + (u32)c] = (u8)code;
#endif
		st.info[k] = reg_slot_valid(T, zeroMask, code) << 16;
		u32 m;
		if (b.level == 0) m = mat_at(G.grid, (int)(b.bx * 16 + cx), (int)(b.by * 16 + cy), (int)(b.bz * 16 + cz));
		else m = L.cache[(size_t)b.slot * BLOCK_CELLS + c];
		st.cellMat[k] = (u16)m;
		TV_ATOMIC_ADD(&st.perCase[T.regClass(code)], 1u);
	}
}

template <typename ST>
struct RegNeighbour {
	const ST* st;
	int cx, cy, cz;
	TV_HD void operator()(int dx, int dy, int dz, u32 slot, bool& valid, u32& mat) const
	{
		const u32 c = (u32)(((cz - dz) << 8) | ((cy - dy) << 4) | (cx - dx));
		valid = false; mat = 0;
		if (!bit_get(st->ntBits, c)) return;
		const u32 k = bit_rank(st->ntBits, st->wordPrefix, c);
		valid = ((st->info[k] >> (16 + slot)) & 1u) != 0;
		mat = st->cellMat[k] & 0xFFu;
	}
};

// reuseValidityMask of cell c from the bitmap: x-bit, y-bit, z-bit
template <typename ST>
TV_HD u32 reg_mask3(const ST& st, int cx, int cy, int cz)
{
	const u32 row = (u32)((cz << 4) | cy);              // 16 cells = half a word
	const u32 word = st.ntBits[row >> 1];
	const u32 rowBits = (word >> ((row & 1) * 16)) & 0xFFFFu;
	u32 m = (rowBits & ((1u << cx) - 1u)) ? 1u : 0u;
	const u32 sliceStart = (u32)cz * 256, before = (u32)cz * 256 + (u32)cy * 16;
	const u32 nBeforeRow = bit_rank(st.ntBits, st.wordPrefix, before) - st.wordPrefix[sliceStart >> 5];
	if (nBeforeRow) m |= 2u;
	if (st.wordPrefix[sliceStart >> 5]) m |= 4u;
	return m;
}

template <typename ST>
TV_HD void reg_cell_setup(const ST& st, const RegBlockCtx& b, u32 k, int& cx, int& cy, int& cz, i8 V[8], CellGeom& geo)
{
	const u32 c = st.cellOf[k];
	cx = c & 15; cy = (c >> 4) & 15; cz = c >> 8;
	reg_cell_values(st.samp, cx, cy, cz, V);
	geo.mult = (int)b.mult; geo.level = (int)b.level;
	geo.local[0] = cx; geo.local[1] = cy; geo.local[2] = cz;
	geo.base[0] = (int)((b.bx * 16 + cx) * b.mult); geo.base[1] = (int)((b.by * 16 + cy) * b.mult); geo.base[2] = (int)((b.bz * 16 + cz) * b.mult);
}

// resolution of every table vertex of every non-trivial cell, stored compactly: new-vertex count, slot ordinals,
// which vertices are created / invalid / placed at v0, and the (direction, slot) each reused vertex comes from
template <typename ST>
TV_HD void reg_phase_count(ST& st, const Tables& T, const RegBlockCtx& b, int tid, int nth)
{
	const int nt = st.wordPrefix[128];
	for (int k = tid; k < nt; k += nth) {
		const u32 c = st.cellOf[k];
		const int cx = (int)(c & 15), cy = (int)((c >> 4) & 15), cz = (int)(c >> 8);
		const u32 bits = st.cellBits[k], code = bits & 0xFFu, zeroMask = bits >> 8;
		const u32 nv = (u32)T.regCell(T.regClass(code))[0] >> 4;
		const u32 mask3 = reg_mask3(st, cx, cy, cz);
		const u32 myMat = st.cellMat[k] & 0xFFu;
		RegNeighbour<ST> nb{ &st, cx, cy, cz };
		u32 count = 0, ords = 0, newMask = 0, atV0 = 0, invalid = 0;
		for (u32 vi = 0; vi < nv; ++vi) {
			const u32 w = T.regVert(code, vi);
			const Resolution r = reg_resolve(zeroMask, w, mask3, myMat, nb);
			if (r.kind == RK_NEW_EDGE || r.kind == RK_NEW_CORNER) {
				if (r.store != NO_SLOT) ords = (ords & ~(0xFu << (r.store * 4))) | (count << (r.store * 4));
				++count;
				newMask |= 1u << vi;
				if (r.kind == RK_NEW_CORNER && r.a == ((w >> 4) & 15)) atV0 |= 1u << vi;
			} else if (r.kind != RK_REUSE) {
				invalid |= 1u << vi;
			}
		}
		st.info[k] = (st.info[k] & 0x000F0000u) | (count << 20) | ords;
		st.vbase[k] = (u16)count;
		st.newMask[k] = (u16)newMask; st.atV0Mask[k] = (u16)atV0; st.invalidMask[k] = (u16)invalid;
	}
}

// Per cell, after vbase is scanned: publish descriptors of this chunk's new vertices
template <typename ST>
TV_HD void reg_phase_describe(ST& st, u32 chunkBase, int tid, int nth)
{
	const int nt = st.wordPrefix[128];
	for (int k = tid; k < nt; k += nth) {
		u32 m = st.newMask[k];
		u32 j = st.vbase[k];
		if (j >= chunkBase + VDESC_CAP || j + 12 <= chunkBase) continue;
		while (m) {
			const u32 vi = (u32)__builtin_ctz(m);
			m &= m - 1;
			if (j >= chunkBase && j < chunkBase + VDESC_CAP) st.vdesc[j - chunkBase] = (u16)((u32)k | (vi << 12));
			++j;
		}
	}
}

// every cell that may reference a vertex created by cell (cx,cy,cz) lies at +0/+1 offsets from it
template <typename ST>
TV_HD void reg_mark_suspect(ST& st, int cx, int cy, int cz)
{
	for (int o = 0; o < 8; ++o) {
		const int x = cx + (o & 1), y = cy + ((o >> 1) & 1), z = cz + (o >> 2);
		if (x > 15 || y > 15 || z > 15) continue;
		const u32 c = (u32)((z << 8) | (y << 4) | x);
		TV_ATOMIC_OR(&st.suspect[c >> 5], 1u << (c & 31));
	}
}

// what a new vertex needs from HBM, requested before anything is computed
struct VertexFetch {
	u32 key;                  // compact cell | table vertex << 12 | table word << 16
	u32 m0, m1;               // mat_at() of the two end points (level 0: the cell corners themselves)
	unsigned long long lut;   // LUT row of the cell's material id (every vertex of a cell carries that id)
};

struct FetchedMaterials {
	u32 m0, m1;
	TV_HD u32 operator()(int which, const int*) const { return which ? m1 : m0; }
};

enum { EMIT_BATCH = 4 };     // vertices per lane whose fetches are in flight together
enum { LOD_BATCH = 4 };      // ... whose LOD chains advance in lockstep (levels >= 1)

// mat_at() for the corners of a level-0 block's cells: one base address per block, 32-bit offsets per corner.
// The only clamp that can bite is the far corner layer of the last block of an axis.
struct BlockMaterials {
	const u8* mat;
	const u8* blend;
	int n, pitch, maxX, maxY, maxZ; // pitch = rows per z-plane of the resident material field
	TV_HD u32 at(int lx, int ly, int lz) const
	{
		lx = lx > maxX ? maxX : lx; ly = ly > maxY ? maxY : ly; lz = lz > maxZ ? maxZ : lz;
		const u32 off = (u32)((lz * pitch + ly) * n + lx);
		return (u32)mat[off] | ((u32)blend[off] << 8);
	}
};

TV_HD BlockMaterials block_materials(const GridView& g, const RegBlockCtx& b)
{
	BlockMaterials m;
	const size_t origin = mat_offset(g, (int)(b.bx * 16), (int)(b.by * 16), (int)(b.bz * 16));
	m.mat = g.mat + origin; m.blend = g.blend + origin; m.n = g.n; m.pitch = g.pitchYMat;
	m.maxX = g.n - 1 - (int)(b.bx * 16); m.maxY = g.n - 1 - (int)(b.by * 16); m.maxZ = g.n - 1 - (int)(b.bz * 16);
	return m;
}

// One lane = one new vertex of the chunk: uniform work, consecutive 48-byte stores.  The global reads of a vertex
// (end-point materials, LUT row) do not depend on its arithmetic, so a lane requests them for all of its vertices
// first: one memory round trip per chunk instead of two per vertex.  Levels >= 1 read their materials at the
// LOD-shifted end points, which only the chain knows; they still get the LUT row ahead.
template <typename ST, typename D, bool LEVEL0>
TV_HD void reg_vertices_emit_with(ST& st, const D& d, const Tables& T, const Globals& G, const Pools& P, const RegBlockCtx& b, u32 chunkBase, int tid, int nth)
{
	const bool room = st.vOff + st.vTotal <= P.vertCap;
	const u32 end = (st.vTotal - chunkBase < (u32)VDESC_CAP) ? st.vTotal - chunkBase : (u32)VDESC_CAP;
	const BlockMaterials bm = block_materials(G.grid, b);
	for (u32 j0 = (u32)tid; j0 < end; j0 += (u32)nth * EMIT_BATCH) {
		VertexFetch f[EMIT_BATCH];
#pragma unroll
		for (int r = 0; r < EMIT_BATCH; ++r) {
			const u32 j = j0 + (u32)r * (u32)nth;
			if (j >= end) continue;
			const u32 desc = st.vdesc[j];
			const u32 k = desc & 0xFFFu, vi = desc >> 12;
			const u32 bits = st.cellBits[k];
			const u32 w = T.regVert(bits & 0xFFu, vi);
			f[r].key = desc | (w << 16);
			f[r].lut = lut_row(G.lut, st.cellMat[k]);
			if (LEVEL0) {
				const u32 c = st.cellOf[k];
				const int cx = (int)(c & 15), cy = (int)((c >> 4) & 15), cz = (int)(c >> 8);
				const int v0 = (w >> 4) & 15, v1 = w & 15;
				const int e = edge_end_bits(bits >> 8, v0, v1);
				int a = v0, bb = v1; // the corners whose materials the vertex reads
				if (e != 1) { a = ((st.atV0Mask[k] >> vi) & 1u) ? v0 : ((e == 0) ? v1 : v0); bb = a; }
				f[r].m0 = bm.at(cx + (a & 1), cy + ((a >> 1) & 1), cz + (a >> 2));
				f[r].m1 = bm.at(cx + (bb & 1), cy + ((bb >> 1) & 1), cz + (bb >> 2));
			}
		}
#pragma unroll
		for (int r = 0; r < EMIT_BATCH; ++r) {
			const u32 j = j0 + (u32)r * (u32)nth;
			if (j >= end) continue;
			const u32 k = f[r].key & 0xFFFu, vi = (f[r].key >> 12) & 15u, w = f[r].key >> 16;
			const u32 c = st.cellOf[k];
			const int cx = (int)(c & 15), cy = (int)((c >> 4) & 15), cz = (int)(c >> 8);
			CellGeom geo;
			geo.mult = (int)b.mult; geo.level = (int)b.level;
			geo.local[0] = cx; geo.local[1] = cy; geo.local[2] = cz;
			geo.base[0] = (int)((b.bx * 16 + cx) * b.mult); geo.base[1] = (int)((b.by * 16 + cy) * b.mult); geo.base[2] = (int)((b.bz * 16 + cz) * b.mult);
			const int v0 = (w >> 4) & 15, v1 = w & 15;
			const int val0 = st.samp[samp_index(cx + (v0 & 1), cy + ((v0 >> 1) & 1), cz + (v0 >> 2))];
			const int val1 = st.samp[samp_index(cx + (v1 & 1), cy + ((v1 >> 1) & 1), cz + (v1 >> 2))];
			const u32 cellMat = st.cellMat[k];
			const int e = edge_end(val0, val1);
			RawVertex rv;
			bool interior = false;
			if (e != 1) {
				const int corner = ((st.atV0Mask[k] >> vi) & 1u) ? v0 : ((e == 0) ? v1 : v0);
				if (LEVEL0) reg_corner_vertex(d, FetchedMaterials{ f[r].m0, f[r].m1 }, geo, corner, cellMat, rv);
				else reg_corner_vertex(d, GridMaterials{ &G.grid }, geo, corner, cellMat, rv);
			} else {
				const int t = edge_t(val0, val1);
				if (LEVEL0) interior = reg_edge_vertex(d, FetchedMaterials{ f[r].m0, f[r].m1 }, geo, v0, v1, t, val0, val1, cellMat, rv);
				else interior = reg_edge_vertex(d, GridMaterials{ &G.grid }, geo, v0, v1, t, val0, val1, cellMat, rv);
			}
			if (!interior) reg_mark_suspect(st, cx, cy, cz);
			if (room) pack_vertex_row(rv, f[r].lut, P.verts + st.vOff + chunkBase + j);
		}
	}
}

// Levels >= 1: the LOD chain (FindBestVertexInLODChain, TransVoxelImpl.cpp:1484-1509) is a sequence of dependent
// fetches — one per level — before a vertex knows its level-0 edge.  A lane walks the chains of all its vertices of the
// chunk in lockstep (one memory round trip per chain step for the whole batch, not per vertex), then finishes them one
// after the other with everything a vertex reads around its end points requested together (reg_edge_finish).
template <typename ST, typename D>
TV_HD void reg_vertices_emit_lod(ST& st, const D& d, const Tables& T, const Globals& G, const Pools& P, const RegBlockCtx& b, u32 chunkBase, int tid, int nth)
{
	const bool room = st.vOff + st.vTotal <= P.vertCap;
	const u32 end = (st.vTotal - chunkBase < (u32)VDESC_CAP) ? st.vTotal - chunkBase : (u32)VDESC_CAP;
	const int ox = (int)(b.bx * 16 * b.mult), oy = (int)(b.by * 16 * b.mult), oz = (int)(b.bz * 16 * b.mult);
	enum { NO_VERTEX = 0x200, EDGE_VERTEX = 0x100 }; // state word: sample p0 (byte 0) | sample p1 (byte 1) | corner id or one of these << 16
	for (u32 j0 = (u32)tid; j0 < end; j0 += (u32)nth * LOD_BATCH) {
		// per vertex of the batch: descriptor + table word, the two end points, the state word — nothing else lives across
		// the chain (registers: the batch is what a lane keeps in flight)
		u32 key[LOD_BATCH];            // compact cell | table vertex << 12 | table word << 16
		int P0[LOD_BATCH][3], P1[LOD_BATCH][3];
		u32 state[LOD_BATCH];
#pragma unroll
		for (int r = 0; r < LOD_BATCH; ++r) {
			const u32 j = j0 + (u32)r * (u32)nth;
			key[r] = 0; state[r] = (u32)NO_VERTEX << 16;
			P0[r][0] = P1[r][0] = ox; P0[r][1] = P1[r][1] = oy; P0[r][2] = P1[r][2] = oz; // a vertex-less slot reads the block's origin
			if (j >= end) continue;
			const u32 desc = st.vdesc[j];
			const u32 k = desc & 0xFFFu, vi = desc >> 12;
			const u32 w = T.regVert(st.cellBits[k] & 0xFFu, vi);
			key[r] = desc | (w << 16);
			const u32 c = st.cellOf[k];
			const int cx = (int)(c & 15), cy = (int)((c >> 4) & 15), cz = (int)(c >> 8);
			const int v0 = (w >> 4) & 15, v1 = w & 15;
			const int s0 = st.samp[samp_index(cx + (v0 & 1), cy + ((v0 >> 1) & 1), cz + (v0 >> 2))];
			const int s1 = st.samp[samp_index(cx + (v1 & 1), cy + ((v1 >> 1) & 1), cz + (v1 >> 2))];
			const int e = edge_end(s0, s1);
			const int corner = (e != 1) ? (((st.atV0Mask[k] >> vi) & 1u) ? v0 : ((e == 0) ? v1 : v0)) : (int)EDGE_VERTEX;
			state[r] = ((u32)s0 & 0xFFu) | (((u32)s1 & 0xFFu) << 8) | ((u32)corner << 16);
			const int a = (e != 1) ? corner : v0, bb = (e != 1) ? corner : v1;
			const int mult = (int)b.mult;
			P0[r][0] = ox + (cx + (a & 1)) * mult; P0[r][1] = oy + (cy + ((a >> 1) & 1)) * mult; P0[r][2] = oz + (cz + (a >> 2)) * mult;
			P1[r][0] = ox + (cx + (bb & 1)) * mult; P1[r][1] = oy + (cy + ((bb >> 1) & 1)) * mult; P1[r][2] = oz + (cz + (bb >> 2)) * mult;
		}
		// the chains, one step of every vertex at a time (a slot without an edge vertex has P0 == P1: it reads that point and
		// changes nothing)
		for (int lev = (int)b.level; lev > 0; --lev) {
			int mv[LOD_BATCH];
#pragma unroll
			for (int r = 0; r < LOD_BATCH; ++r)
				mv[r] = d(P0[r][0] + (P1[r][0] - P0[r][0]) / 2, P0[r][1] + (P1[r][1] - P0[r][1]) / 2, P0[r][2] + (P1[r][2] - P0[r][2]) / 2);
#pragma unroll
			for (int r = 0; r < LOD_BATCH; ++r) {
				if ((state[r] >> 16) != (u32)EDGE_VERTEX) continue;
				const int mx = P0[r][0] + (P1[r][0] - P0[r][0]) / 2, my = P0[r][1] + (P1[r][1] - P0[r][1]) / 2, mz = P0[r][2] + (P1[r][2] - P0[r][2]) / 2;
				const int v0s = (int)(i8)(state[r] & 0xFFu);
				if (v0s * mv[r] <= 0) { P1[r][0] = mx; P1[r][1] = my; P1[r][2] = mz; state[r] = (state[r] & 0xFFFF00FFu) | (((u32)mv[r] & 0xFFu) << 8); }
				else { P0[r][0] = mx; P0[r][1] = my; P0[r][2] = mz; state[r] = (state[r] & 0xFFFFFF00u) | ((u32)mv[r] & 0xFFu); }
			}
		}
#pragma unroll
		for (int r = 0; r < LOD_BATCH; ++r) {
			const u32 kind = state[r] >> 16;
			if (kind == (u32)NO_VERTEX) continue;
			const u32 j = j0 + (u32)r * (u32)nth;
			const u32 k = key[r] & 0xFFFu, w = key[r] >> 16;
			const u32 c = st.cellOf[k];
			const int cx = (int)(c & 15), cy = (int)((c >> 4) & 15), cz = (int)(c >> 8);
			CellGeom geo;
			geo.mult = (int)b.mult; geo.level = (int)b.level;
			geo.local[0] = cx; geo.local[1] = cy; geo.local[2] = cz;
			geo.base[0] = ox + cx * (int)b.mult; geo.base[1] = oy + cy * (int)b.mult; geo.base[2] = oz + cz * (int)b.mult;
			const u32 cellMat = st.cellMat[k];
			const unsigned long long lut = lut_row(G.lut, cellMat); // requested with the end-point samples below
			RawVertex rv;
			bool interior = false;
			if (kind != (u32)EDGE_VERTEX) {
				reg_corner_vertex(d, GridMaterials{ &G.grid }, geo, (int)kind, cellMat, rv);
			} else {
				const int p0 = (int)(i8)(state[r] & 0xFFu), p1 = (int)(i8)((state[r] >> 8) & 0xFFu);
				const int t = (p0 != p1) ? edge_t(p0, p1) : 0;
				interior = reg_edge_finish(d, GridMaterials{ &G.grid }, geo, (int)((w >> 4) & 15), (int)(w & 15), P0[r], P1[r], p0, p1, t, cellMat, rv);
			}
			if (!interior) reg_mark_suspect(st, cx, cy, cz);
			if (room) pack_vertex_row(rv, lut, P.verts + st.vOff + chunkBase + j);
		}
	}
}

template <typename ST>
TV_HD void reg_phase_emit_vertices(ST& st, const Tables& T, const Globals& G, const Pools& P, const RegBlockCtx& b, u32 chunkBase, int tid, int nth)
{
	if (b.level == 0) {
		const LocalDist d{ st.samp, (int)(b.bx * 16), (int)(b.by * 16), (int)(b.bz * 16) };
		reg_vertices_emit_with<ST, LocalDist, true>(st, d, T, G, P, b, chunkBase, tid, nth);
	} else {
		const GlobalDist d{ &G.grid };
		reg_vertices_emit_lod<ST, GlobalDist>(st, d, T, G, P, b, chunkBase, tid, nth);
	}
}

// Kept-triangle mask (PushBlocksToResult's degenerate filter).  A triangle whose three vertices lie strictly inside
// three distinct cell edges cannot be degenerate, and for cell sizes <= 16 the reference's fp32 test is exact integer
// arithmetic, so only "suspect" cells (marked during vertex emission) evaluate it; larger cells always do.
template <typename ST, typename D>
TV_HD void reg_keep_with(ST& st, const D& d, const Tables& T, const RegBlockCtx& b, int tid, int nth)
{
	const int nt = st.wordPrefix[128];
	for (int k = tid; k < nt; k += nth) {
		const u32 code = st.cellBits[k] & 0xFFu;
		const u8* cd = T.regCell(T.regClass(code));
		const u32 ntri = (u32)cd[0] & 15;
		u32 keepMask = (1u << ntri) - 1u, kept = ntri;
		if (b.mult > 16 || bit_get(st.suspect, (u32)st.cellOf[k])) {
			int cx, cy, cz; i8 V[8]; CellGeom geo;
			reg_cell_setup(st, b, (u32)k, cx, cy, cz, V, geo);
			const u32 invalidMask = st.invalidMask[k], atV0Mask = st.atV0Mask[k];
			keepMask = 0; kept = 0;
			for (u32 tr = 0; tr < ntri; ++tr) {
				const u32 a = cd[1 + tr * 3], bb = cd[2 + tr * 3], c = cd[3 + tr * 3];
				bool keepIt = true;
				if (!(((invalidMask >> a) | (invalidMask >> bb) | (invalidMask >> c)) & 1u)) {
					float pa[3], pb[3], pc[3];
					reg_vertex_position(d, geo, V, T.regVert(code, a), ((atV0Mask >> a) & 1u) != 0, pa);
					reg_vertex_position(d, geo, V, T.regVert(code, bb), ((atV0Mask >> bb) & 1u) != 0, pb);
					reg_vertex_position(d, geo, V, T.regVert(code, c), ((atV0Mask >> c) & 1u) != 0, pc);
					keepIt = !triangle_degenerate(pa, pb, pc);
				}
				if (keepIt) { keepMask |= 1u << tr; ++kept; }
			}
			if (kept != ntri) TV_ATOMIC_ADD(&st.degenerate, ntri - kept);
		}
		st.info[k] = (st.info[k] & 0x00FFFFFFu) | (keepMask << 24);
		st.ibase[k] = (u16)(kept * 3);
	}
}

template <typename ST>
TV_HD void reg_phase_keep(ST& st, const Tables& T, const Globals& G, const RegBlockCtx& b, int tid, int nth)
{
	if (b.level == 0) {
		const LocalDist d{ st.samp, (int)(b.bx * 16), (int)(b.by * 16), (int)(b.bz * 16) };
		reg_keep_with(st, d, T, b, tid, nth);
	} else {
		const GlobalDist d{ &G.grid };
		reg_keep_with(st, d, T, b, tid, nth);
	}
}

// Index list, one lane per index.  Phase 1 (per cell, cheap): a descriptor for every kept triangle corner of the chunk
// goes to st.vdesc (free after vertex emission) at its position in the block's index list (ibase scanned):
// compact cell | (triangle * 3 + corner) << 12.
template <typename ST>
TV_HD void reg_phase_stage_indices(ST& st, const Tables& T, u32 chunkBase, int tid, int nth)
{
	const int nt = st.wordPrefix[128];
	for (int k = tid; k < nt; k += nth) {
		u32 keepMask = st.info[k] >> 24;
		u32 pos = st.ibase[k];
		const u32 count = 3u * (u32)TV_POPC(keepMask);
		if (pos >= chunkBase + VDESC_CAP || pos + count <= chunkBase) continue;
		while (keepMask) {
			const u32 tr = (u32)__builtin_ctz(keepMask);
			keepMask &= keepMask - 1;
			for (u32 e = 0; e < 3; ++e, ++pos)
				if (pos >= chunkBase && pos < chunkBase + VDESC_CAP) st.vdesc[pos - chunkBase] = (u16)((u32)k | ((tr * 3 + e) << 12));
		}
	}
}

// Phase 2: every lane resolves its own index — nothing is re-resolved: a created vertex is the cell's vbase plus its
// rank among the cell's created vertices, a reused one comes from the stored slot ordinal of the owner cell — and
// stores it: consecutive lanes write consecutive dwords of the index pool.
template <typename ST>
TV_HD void reg_phase_flush_indices(const ST& st, const Tables& T, const Pools& P, u32 chunkBase, int tid, int nth)
{
	if (st.iOff + st.iTotal > P.idxCap) return;
	const u32 end = (st.iTotal - chunkBase < (u32)VDESC_CAP) ? st.iTotal - chunkBase : (u32)VDESC_CAP;
	u32* out = P.idx + st.iOff + chunkBase;
	for (u32 j = (u32)tid; j < end; j += (u32)nth) {
		const u32 desc = st.vdesc[j];
		const u32 k = desc & 0xFFFu, corner = desc >> 12;
		const u32 bits = st.cellBits[k], code = bits & 0xFFu;
		const u32 vi = T.regCell(T.regClass(code))[1 + corner];
		const u32 newMask = st.newMask[k];
		u32 id;
		if ((newMask >> vi) & 1u) {
			id = (u32)st.vbase[k] + (u32)TV_POPC(newMask & ((1u << vi) - 1u));
		} else if ((st.invalidMask[k] >> vi) & 1u) {
			id = INVALID_INDEX;
		} else {
			u32 dir, slot;
			reg_reuse_source(bits >> 8, T.regVert(code, vi), dir, slot);
			const u32 c = st.cellOf[k];
			const u32 c2 = c - ((dir & 1u) + (((dir >> 1) & 1u) << 4) + (((dir >> 2) & 1u) << 8));
			const u32 k2 = bit_rank(st.ntBits, st.wordPrefix, c2);
			id = (u32)st.vbase[k2] + ((st.info[k2] >> (slot * 4)) & 0xFu);
		}
		TV_STREAM_STORE(&out[j], id);
	}
}

TV_HD void reg_write_empty_record(const LevelDesc& L, u32 slot)
{
	BlockRecord& r = L.records[slot];
	r.coordId = L.slotCoord[slot];
	r.vOff = r.vCount = r.iOff = r.iCount = 0;
	// the transition fields belong to the transition pass wherever it runs (it may run concurrently on another stream)
	if (!L.hasTransitions) for (int f = 0; f < 6; ++f) { r.tvOff[f] = r.tvCount[f] = r.tiOff[f] = r.tiCount[f] = 0; }
	r.degenerate = 0; r.ntCells = 0; r.pad = 0;
}

// `acc` = 20 counters (layout of Globals::stats) private to the caller: the HIP kernels keep them in LDS for the whole
// launch and flush them once per workgroup — per-block global atomics on one cache line serialise chip-wide.
template <typename ST>
TV_HD void reg_phase_record(const ST& st, u32* acc, const LevelDesc& L, const RegBlockCtx& b, const Pools& P, int tid)
{
	if (tid != 0) return;
	BlockRecord& r = L.records[b.slot];
	r.coordId = L.slotCoord[b.slot];
	const bool ok = st.vOff + st.vTotal <= P.vertCap && st.iOff + st.iTotal <= P.idxCap;
	r.vOff = st.vOff; r.vCount = ok ? st.vTotal : 0; r.iOff = st.iOff; r.iCount = ok ? st.iTotal : 0;
	count_listed_block(L, r.coordId, r.vCount);
	if (!L.hasTransitions) for (int f = 0; f < 6; ++f) { r.tvOff[f] = 0; r.tvCount[f] = 0; r.tiOff[f] = 0; r.tiCount[f] = 0; }
	r.degenerate = st.degenerate;
	r.ntCells = st.wordPrefix[128];
	r.pad = 0;
	if (!ok) TV_ATOMIC_OR(&P.cursors[CUR_OVF], 1u);
	acc[0] += (u32)st.wordPrefix[128];
	if (st.vTotal) acc[1] += st.degenerate;
	for (int i = 0; i < 16; ++i) acc[4 + i] += st.perCase[i];
}

// ---------------------------------------------------------------------------------------------------------
// Transition pass state: all 6 faces of a block at once; cell id = f * 256 + row * 16 + col
// ---------------------------------------------------------------------------------------------------------
enum { TR_CELLS = 6 * 256, TR_CAP = 512, TR_PROW = 48 }; // TR_PROW: bytes per staged plane row (33 samples; 16-byte pieces stay aligned)

// The six faces of a block are independent (reuse never crosses a face, every face has its own output ranges), so
// the per-cell state is sized for TR_CAP non-trivial cells and a block is handled in batches of consecutive faces
// whose non-trivial cells fit: nearly always one batch; any two faces (2 x 256 cells) always fit.
struct TrState {
	__attribute__((aligned(16))) i8 plane[6][33 * TR_PROW]; // 33 x 33 full-resolution samples of each boundary plane, index v * TR_PROW + u
	u32 ntAll[48];            // non-trivial transition cells of all faces
	u32 ntBits[48];           // ... of the faces of the current batch
	u16 wordPrefix[50];       // [48] = number of non-trivial transition cells of the batch
	u16 cellOf[TR_CAP];       // compact -> cell id
	u16 cellMat[TR_CAP];      // compact: low-res cell material
	u16 valid[TR_CAP];        // compact: slot valid mask (10 bits)
	u32 cellBits[TR_CAP];     // compact: case code (9 bits) | (expanded sample == 0) mask << 9 — what the index logic needs of the samples
	unsigned long long ords[TR_CAP]; // compact: ordinal (4 bits) of the vertex stored in each of the 10 slots
	u16 vbase[TR_CAP];        // compact: exclusive scan of new vertex counts (flat over the batch's faces)
	u16 ibase[TR_CAP];        // compact: exclusive scan of index counts (flat over the batch's faces)
	u16 newMask[TR_CAP];      // compact: table vertices this cell creates
	u16 vdesc[VDESC_CAP];     // one chunk of new-vertex descriptors: compact cell | table vertex << 11
	u16 faceMat[TR_CELLS];    // GPU passes: the material entries of the low-res cells behind ALL transition cells, requested with the planes
	                          // (one round trip less in front of the list phase) when the block's material cache is known to be complete
	u32 faceOn;               // bit f = face has a neighbour block
	u32 vOff, iOff, vTotal, iTotal;
};

// end of the batch of faces starting at f0: as many consecutive faces as fit TR_CAP cells (at least one)
TV_HD int tr_batch_end(const TrState& st, int f0)
{
	u32 sum = 0;
	int f = f0;
	for (; f < 6; ++f) {
		u32 cnt = 0;
		for (int w = 0; w < 8; ++w) cnt += (u32)TV_POPC(st.ntAll[f * 8 + w]);
		if (f > f0 && sum + cnt > (u32)TR_CAP) break;
		sum += cnt;
	}
	return f;
}

// working bitmap of the batch [f0, f1) and its per-word counts (to be scanned into wordPrefix)
TV_HD void tr_phase_batch_bits(TrState& st, int f0, int f1, int tid, int nth)
{
	for (int w = tid; w < 48; w += nth) {
		const u32 bits = ((w >> 3) >= f0 && (w >> 3) < f1) ? st.ntAll[w] : 0u;
		st.ntBits[w] = bits;
		st.wordPrefix[w] = (u16)TV_POPC(bits);
	}
}

TV_HD void tr_phase_load(TrState& st, const Globals& G, const LevelDesc& L, const RegBlockCtx& b, int tid, int nth)
{
	u32 on = 0;
	const u32 bc[3] = { b.bx, b.by, b.bz };
	for (int f = 0; f < 6; ++f) {
		const FaceGeom fg = face_geom(f);
		if (fg.positive ? (bc[fg.axis] + 1 < L.cnt) : (bc[fg.axis] > 0)) on |= 1u << f;
	}
	if (tid == 0) st.faceOn = on;
	for (int w = tid; w < 48; w += nth) st.ntAll[w] = 0;
	const int half = (int)b.mult >> 1;
	for (int s = tid; s < 6 * PLANE; s += nth) {
		const int f = s / PLANE, r = s % PLANE;
		if (!((on >> f) & 1u)) continue;
		const FaceGeom fg = face_geom(f);
		int p[3];
		p[fg.ua] = (int)(bc[fg.ua] * 16 * b.mult) + (r % 33) * half;
		p[fg.va] = (int)(bc[fg.va] * 16 * b.mult) + (r / 33) * half;
		p[fg.axis] = (int)((bc[fg.axis] * 16 + (fg.positive ? 16 : 0)) * b.mult);
		st.plane[f][(r / 33) * TR_PROW + r % 33] = (i8)dist_at(G.grid, p[0], p[1], p[2]);
	}
}

TV_HD void tr_cell_values(const TrState& st, int f, int row, int col, i8 v9[9])
{
	const i8* p = st.plane[f] + (row * 2) * TR_PROW + col * 2;
#pragma unroll
	for (int j = 0; j < 3; ++j)
#pragma unroll
		for (int i = 0; i < 3; ++i) v9[j * 3 + i] = p[j * TR_PROW + i];
}

// sign bits of one staged plane row (33 samples): bits 0..16 = samples 0..16 (returned), `hi` = samples 16..32
TV_HD u32 tr_row_signs(const i8* row, u32& hi)
{
#if defined(__HIP_DEVICE_COMPILE__)
	// (rows are 16-byte aligned: TR_PROW = 48).  The sign bits of four bytes as a nibble: the bits 7, 15, 23, 31 times
	// (1 + 2^7 + 2^14 + 2^21) meet in the bits 28..31 - no two partial products share a bit, nothing carries
	const uint4 a = *(const uint4*)row, b = *(const uint4*)(row + 16);
	const auto s4 = [](u32 d) { return ((d & 0x80808080u) * 0x00204081u) >> 28; };
	const u32 lo16 = s4(a.x) | (s4(a.y) << 4) | (s4(a.z) << 8) | (s4(a.w) << 12), hi16 = s4(b.x) | (s4(b.y) << 4) | (s4(b.z) << 8) | (s4(b.w) << 12);
#else
	u32 lo16 = 0, hi16 = 0;
	for (int i = 0; i < 16; ++i) { lo16 |= ((u32)(row[i] >> 7) & 1u) << i; hi16 |= ((u32)(row[16 + i] >> 7) & 1u) << i; }
#endif
	hi = hi16 | (((u32)(row[32] >> 7) & 1u) << 16);
	return lo16 | ((hi16 & 1u) << 16);
}

// non-trivial transition cells among the 8 cells whose 3 x 3 samples lie in the bits 2c .. 2c + 2 of three rows: A = AND of the
// rows' sign masks, O = their OR (17 bits each).  A cell is trivial iff its nine samples agree in sign (case code 0 or 511, :1923).
TV_HD u32 tr_cells8(u32 A, u32 O)
{
	const u32 all = A & (A >> 1) & (A >> 2), any = O | (O >> 1) | (O >> 2);
	u32 x = any & ~all & 0x5555u;                       // bit 2c: cell c
	x = (x | (x >> 1)) & 0x3333u; x = (x | (x >> 2)) & 0x0F0Fu; x = (x | (x >> 4)) & 0x00FFu;
	return x;
}

// One lane per (face, cell row): the 16 cells of the row from the sign masks of its three sample rows, bit-parallel (round 6:
// one lane per cell with nine byte reads and a case code each was a tenth of a transition block's vector instructions).
TV_HD void tr_phase_classify(TrState& st, int tid, int nth)
{
	for (int cr = tid; cr < 96; cr += nth) {
		const int f = cr >> 4, row = cr & 15;
		u32 bits = 0;
		if ((st.faceOn >> f) & 1u) {
			const i8* p = st.plane[f] + (row * 2) * TR_PROW;
			u32 h0, h1, h2;
			const u32 l0 = tr_row_signs(p, h0), l1 = tr_row_signs(p + TR_PROW, h1), l2 = tr_row_signs(p + 2 * TR_PROW, h2);
			bits = tr_cells8(l0 & l1 & l2, l0 | l1 | l2) | (tr_cells8(h0 & h1 & h2, h0 | h1 | h2) << 8);
		}
		// cell c = f << 8 | row << 4 | col is bit c & 31 of word c >> 5: the row's 16 cells are one half-word
		((u16*)st.ntAll)[cr] = (u16)bits;
	}
}

// local (x,y,z) of the low-res cell behind transition cell (f,row,col)
TV_HD void tr_low_local(const FaceGeom& fg, int row, int col, int local[3])
{
	face_scatter(fg, col, row, fg.positive ? 15 : 0, local);
}

// compact -> cell id of the batch's non-trivial transition cells (behind the scan of the batch's word counts): one lane per
// (face, cell row) half-word of the working bitmap
TV_HD void tr_phase_cells_of(TrState& st, int tid, int nth)
{
	for (int cr = tid; cr < 96; cr += nth) {
		u32 bits = ((const u16*)st.ntBits)[cr];
		if (!bits) continue;
		u32 k = st.wordPrefix[cr >> 1];
		if (cr & 1) k += (u32)TV_POPC(st.ntBits[cr >> 1] & 0xFFFFu);
		while (bits) {
			const u32 col = (u32)__builtin_ctz(bits);
			bits &= bits - 1;
			st.cellOf[k++] = (u16)(((u32)cr << 4) | col);
		}
	}
}

// preMat: TrState::faceMat filled by the caller (per transition cell), or nullptr: the entries are fetched here.
// One lane per COMPACT cell (tr_phase_cells_of ran, a barrier passed): until round 6 every lane walked six of the 1 536 cells
// and skipped the trivial ones - with a few non-trivial cells in every wave the whole body ran for nearly every group of 64.
TV_HD void tr_phase_list(TrState& st, const Tables& T, const LevelDesc& L, const RegBlockCtx& b, int tid, int nth, const u16* preMat = nullptr)
{
	const int nt = st.wordPrefix[48];
	for (int k = tid; k < nt; k += nth) {
		const int c = st.cellOf[k];
		const int f = c >> 8, row = (c >> 4) & 15, col = c & 15;
		i8 v9[9], v[13];
		tr_cell_values(st, f, row, col, v9);
		tr_expand_values(v9, v);
		const u32 code = tr_case_code(v9);
#if defined(VX_CASE_DUMP)
		L.trCaseDump[(size_t)b.slot * TR_CELLS + (u32)c] = (u16)code;
#endif
		st.cellBits[k] = code | (tr_zero_mask(v) << 9);
		st.valid[k] = (u16)tr_slot_valid(T, v, code);
		int local[3];
		tr_low_local(face_geom(f), row, col, local);
		st.cellMat[k] = preMat ? preMat[c] : TV_LOAD_THROUGH(&L.cache[(size_t)b.slot * BLOCK_CELLS + (u32)((local[2] << 8) | (local[1] << 4) | local[0])]);
	}
}

struct TrNeighbour {
	const TrState* st;
	int f, row, col;
	TV_HD void operator()(int dcol, int drow, u32 slot, bool& valid, u32& mat) const
	{
		const u32 c = (u32)((f << 8) | ((row - drow) << 4) | (col - dcol));
		valid = false; mat = 0;
		if (!bit_get(st->ntBits, c)) return;
		const u32 k = bit_rank(st->ntBits, st->wordPrefix, c);
		valid = ((st->valid[k] >> slot) & 1u) != 0;
		mat = st->cellMat[k] & 0xFFu;
	}
};

TV_HD u32 tr_mask2(const TrState& st, int f, int row, int col)
{
	const u32 r = (u32)((f << 4) | row);
	const u32 rowBits = (st.ntBits[r >> 1] >> ((r & 1) * 16)) & 0xFFFFu;
	u32 m = (rowBits & ((1u << col) - 1u)) ? 1u : 0u;
	if (row > 0) m |= 2u;
	return m;
}

TV_HD void tr_phase_count(TrState& st, const Tables& T, int tid, int nth)
{
	const int nt = st.wordPrefix[48];
	for (int k = tid; k < nt; k += nth) {
		const u32 c = st.cellOf[k];
		const int f = (int)(c >> 8), row = (int)((c >> 4) & 15), col = (int)(c & 15);
		const u32 bits = st.cellBits[k], code = bits & 0x1FFu, zeroMask = bits >> 9;
		const u8* cd = T.trCell(T.trClass(code) & 0x7F);
		const u32 nv = (u32)cd[0] >> 4, ntri = (u32)cd[0] & 15;
		const u32 mask2 = tr_mask2(st, f, row, col);
		TrNeighbour nb{ &st, f, row, col };
		u32 count = 0, newMask = 0;
		unsigned long long ords = 0;
		for (u32 vi = 0; vi < nv; ++vi) {
			const TrResolution r = tr_resolve(T, zeroMask, T.trVert(code, vi), mask2, st.cellMat[k] & 0xFFu, nb);
			if (r.kind == RK_NEW_EDGE) {
				if (r.store != NO_SLOT) ords = (ords & ~(0xFull << (r.store * 4))) | ((unsigned long long)count << (r.store * 4));
				++count;
				newMask |= 1u << vi;
			}
		}
		st.ords[k] = ords;
		st.vbase[k] = (u16)count;
		st.ibase[k] = (u16)(ntri * 3);
		st.newMask[k] = (u16)newMask;
	}
}

TV_HD void tr_cell_geom(const FaceGeom& fg, const RegBlockCtx& b, int row, int col, TrCellGeom& geo)
{
	tr_low_local(fg, row, col, geo.local);
	geo.mult = (int)b.mult; geo.level = (int)b.level;
	geo.lowBase[0] = (int)((b.bx * 16 + geo.local[0]) * b.mult);
	geo.lowBase[1] = (int)((b.by * 16 + geo.local[1]) * b.mult);
	geo.lowBase[2] = (int)((b.bz * 16 + geo.local[2]) * b.mult);
}

// descriptors of this chunk's new vertices (after vbase is scanned)
TV_HD void tr_phase_describe(TrState& st, u32 chunkBase, int tid, int nth)
{
	const int nt = st.wordPrefix[48];
	for (int k = tid; k < nt; k += nth) {
		u32 m = st.newMask[k];
		u32 j = st.vbase[k];
		if (j >= chunkBase + VDESC_CAP || j + 12 <= chunkBase) continue;
		while (m) {
			const u32 vi = (u32)__builtin_ctz(m);
			m &= m - 1;
			if (j >= chunkBase && j < chunkBase + VDESC_CAP) st.vdesc[j - chunkBase] = (u16)((u32)k | (vi << 11));
			++j;
		}
	}
}

// one lane = one new transition vertex of the chunk; `smp` reads the voxels around it (tr_new_vertex)
template <typename SMP>
TV_HD void tr_phase_emit_vertices(TrState& st, const Tables& T, const Globals& G, const SMP& smp, const Pools& P, const RegBlockCtx& b, u32 chunkBase, int tid, int nth)
{
	if (st.vOff + st.vTotal > P.vertCap || st.iOff + st.iTotal > P.idxCap) return;
	const u32 end = (st.vTotal - chunkBase < (u32)VDESC_CAP) ? st.vTotal - chunkBase : (u32)VDESC_CAP;
	for (u32 j = (u32)tid; j < end; j += (u32)nth) {
		const u32 desc = st.vdesc[j];
		const u32 k = desc & 0x7FFu, vi = desc >> 11;
		const u32 c = st.cellOf[k];
		const int f = (int)(c >> 8), row = (int)((c >> 4) & 15), col = (int)(c & 15);
		const FaceGeom fg = face_geom(f);
		// case code and zero mask are the classification's (cellBits); of the cell's samples the vertex needs its two end points:
		// expanded sample i sits at plane offset (dv, du) = (i / 3, i % 3) for i < 9, the low-resolution corners 9..12 are the
		// plane's corners 0, 2, 6, 8
		const u32 bits = st.cellBits[k];
		const u32 w = T.trVert(bits & 0x1FFu, vi);
		TrResolution r;
		int corner; u32 dir, slot; bool endpoint;
		tr_vertex_dir_slot_z(T, bits >> 9, w, r.t, dir, slot, endpoint, corner);
		r.endpoint = endpoint ? 1 : 0; r.dir = (u8)dir; r.slot = (u8)slot; r.kind = RK_NEW_EDGE; r.store = NO_SLOT;
		const i8* cellSamples = st.plane[f] + (row * 2) * TR_PROW + col * 2;
		const u32 e0 = (w >> 4) & 15u, e1 = w & 15u;
		// small tables in constants: plane sample g (0..8) of expanded sample i (nibbles); row g / 3 and column g % 3 (2 bits each)
		const u32 g0 = (u32)(0x8620876543210ull >> (4u * e0)) & 15u, g1 = (u32)(0x8620876543210ull >> (4u * e1)) & 15u;
		const int p0 = cellSamples[((0x2A540u >> (2u * g0)) & 3u) * TR_PROW + ((0x24924u >> (2u * g0)) & 3u)];
		const int p1 = cellSamples[((0x2A540u >> (2u * g1)) & 3u) * TR_PROW + ((0x24924u >> (2u * g1)) & 3u)];
		TrCellGeom geo;
		tr_cell_geom(fg, b, row, col, geo);
		// the vertex's material id is the low-res cell's whatever the end points hold (tr_new_vertex), so its row of the
		// material table is requested before the voxel fetches instead of behind them (one round trip less)
		const unsigned long long lut = lut_row(G.lut, st.cellMat[k]);
		RawVertex rv;
		tr_new_vertex(smp, fg, geo, p0, p1, w, r, st.cellMat[k], rv);
		pack_vertex_row(rv, lut, P.verts + st.vOff + chunkBase + j);
	}
}

// Index lists of the batch's faces, one lane per TRIANGLE (its three indices share every look-up of their cell).  A chunk is
// VDESC_CAP triangles = TR_INDEX_CHUNK indices; chunkBase counts indices.  Phase 1, per cell: a descriptor for every triangle
// of the chunk at its position in the flat triangle list: compact cell | triangle << 9.
enum { TR_INDEX_CHUNK = 3 * VDESC_CAP };
TV_HD void tr_phase_stage_indices(TrState& st, const Tables& T, u32 chunkBase, int tid, int nth)
{
	const int nt = st.wordPrefix[48];
	const u32 chunkTri = chunkBase / 3u;
	for (int k = tid; k < nt; k += nth) {
		const u32 first = (u32)st.ibase[k] / 3u; // (index counts are multiples of three, and so are their prefix sums)
		const u32 count = (u32)T.trCell(T.trClass(st.cellBits[k] & 0x1FFu) & 0x7F)[0] & 15u;
		if (first >= chunkTri + VDESC_CAP || first + count <= chunkTri) continue;
		for (u32 tr = 0; tr < count; ++tr) {
			const u32 pos = first + tr;
			if (pos >= chunkTri && pos < chunkTri + VDESC_CAP) st.vdesc[pos - chunkTri] = (u16)((u32)k | (tr << 9));
		}
	}
}

// Phase 2: each lane resolves and stores its triangle's three indices (relative to its face's first vertex; winding flipped per class/face)
TV_HD void tr_phase_flush_indices(const TrState& st, const Tables& T, const Pools& P, u32 chunkBase, int tid, int nth)
{
	if (st.vOff + st.vTotal > P.vertCap || st.iOff + st.iTotal > P.idxCap) return;
	const u32 left = (st.iTotal - chunkBase) / 3u;
	const u32 end = left < (u32)VDESC_CAP ? left : (u32)VDESC_CAP;
	u32* out = P.idx + st.iOff + chunkBase;
	for (u32 j = (u32)tid; j < end; j += (u32)nth) {
		const u32 desc = st.vdesc[j];
		const u32 k = desc & 0x1FFu, tr = desc >> 9;
		const u32 c = st.cellOf[k];
		const u32 f = c >> 8;
		const u32 bits = st.cellBits[k], code = bits & 0x1FFu;
		const u32 cls = T.trClass(code);
		const u8* cd = T.trCell(cls & 0x7F);
		const bool flip = ((cls >> 7) ^ (f & 1u)) != 0; // reverseWinding = {0,1,0,1,0,1}
		const u32 faceVBase = st.vbase[st.wordPrefix[f * 8]]; // the face's first cell exists: cell k is in it
		const u32 newMask = st.newMask[k], own = (u32)st.vbase[k] - faceVBase;
#pragma unroll
		for (u32 e = 0; e < 3; ++e) {
			const u32 vi = cd[1 + tr * 3 + ((flip && e) ? 3 - e : e)];
			u32 id;
			if ((newMask >> vi) & 1u) {
				id = own + (u32)TV_POPC(newMask & ((1u << vi) - 1u));
			} else {
				int t, corner; u32 dir, slot; bool endpoint;
				tr_vertex_dir_slot_z(T, bits >> 9, T.trVert(code, vi), t, dir, slot, endpoint, corner);
				const u32 c2 = c - ((dir & 1u) + (((dir >> 1) & 1u) << 4));
				const u32 k2 = bit_rank(st.ntBits, st.wordPrefix, c2);
				id = (u32)st.vbase[k2] - faceVBase + ((u32)(st.ords[k2] >> (slot * 4)) & 0xFu);
			}
			TV_STREAM_STORE(&out[j * 3u + e], id);
		}
	}
}

// a block of a transition level without any transition geometry
TV_HD void tr_write_empty_record(const LevelDesc& L, u32 slot)
{
	BlockRecord& r = L.records[slot];
	for (int f = 0; f < 6; ++f) { r.tvOff[f] = r.tvCount[f] = r.tiOff[f] = r.tiCount[f] = 0; }
}

// output ranges of the faces [f0, f1) of the current batch (zero ranges when the pools overflowed or nothing came out)
TV_HD void tr_phase_record(const TrState& st, const LevelDesc& L, const RegBlockCtx& b, const Pools& P, int f0, int f1, int tid)
{
	if (tid != 0) return;
	BlockRecord& r = L.records[b.slot];
	const u32 nt = st.wordPrefix[48];
	const bool ok = st.vOff + st.vTotal <= P.vertCap && st.iOff + st.iTotal <= P.idxCap;
	if (!ok) TV_ATOMIC_OR(&P.cursors[CUR_OVF], 1u);
	for (int f = f0; f < f1; ++f) {
		if (!ok || !nt) { r.tvOff[f] = r.tvCount[f] = r.tiOff[f] = r.tiCount[f] = 0; continue; }
		const u32 k0 = st.wordPrefix[f * 8], k1 = st.wordPrefix[f * 8 + 8];
		const u32 v0 = (k0 < nt) ? st.vbase[k0] : st.vTotal, v1 = (k1 < nt) ? st.vbase[k1] : st.vTotal;
		const u32 i0 = (k0 < nt) ? st.ibase[k0] : st.iTotal, i1 = (k1 < nt) ? st.ibase[k1] : st.iTotal;
		r.tvOff[f] = st.vOff + v0; r.tvCount[f] = v1 - v0;
		r.tiOff[f] = st.iOff + i0; r.tiCount[f] = i1 - i0;
	}
}

// ---------------------------------------------------------------------------------------------------------
// Edits on the resident grid: Grid::InjectSurface with the analytic ball brush, Grid::InjectMaterial
// (src/VoxelGrid.cpp:388-584), and the codec's BF_Empty rule for the touched blocks (:610-672)
// ---------------------------------------------------------------------------------------------------------
enum { EDIT_BALL = 0, EDIT_MATERIAL = 1 };

struct EditParams {
	float pos[3], ext[3];
	int kind, type;     // type = InjectionType for EDIT_BALL
	float radius;
	u32 material, add;
};

// the section of block (bx,by,bz) an edit rewrites: per axis first local coordinate and iteration count of the
// reference's `for (x = start; x < end; ++x)` loops (:368-386; coordinates may be fractional, the voxel is (unsigned)x)
struct EditSection {
	float start[3];
	int count[3];
	float bmin[3];
	// the brush is sampled by ITS OWN float loops from sstart (one step per sample) while the grid consumes the samples
	// with a running index over its loops; float rounding can give the two a different trip count along an axis
	// (e.g. 5 rows in the grid loop, 6 samples in the brush loop), and then the reference pairs voxels with shifted
	// samples.  scount = trip counts of the brush loops.
	float sstart[3];
	int scount[3];
};

TV_HD float edit_clampf(float v, float lo, float hi) { return v < lo ? lo : (v > hi ? hi : v); }

TV_HD EditSection edit_section(const EditParams& e, u32 bx, u32 by, u32 bz)
{
	EditSection s;
	const u32 bc[3] = { bx, by, bz };
	for (int k = 0; k < 3; ++k) {
		s.bmin[k] = (float)(bc[k] * 16);
		const float p = e.pos[k] - e.ext[k] / 2;
		const float b0 = edit_clampf(p, s.bmin[k], s.bmin[k] + 16.f) - s.bmin[k];
		const float b1 = edit_clampf(p + e.ext[k], s.bmin[k], s.bmin[k] + 16.f) - s.bmin[k];
		s.start[k] = b0;
		int cnt = 0;
		for (float x = b0; x < b1; ++x) ++cnt; // at most 17 steps; the loop itself is the specification
		s.count[k] = cnt;
		const float s0 = (s.bmin[k] + b0) - e.pos[k], s1 = (s.bmin[k] + b1) - e.pos[k]; // surfaceCoordStart / End (:412-413)
		s.sstart[k] = s0;
		int sc = 0;
		for (float x = s0; x < s1; x += 1.0f) ++sc;
		s.scount[k] = sc;
	}
	return s;
}

// :37-50
TV_HD i8 edit_round_distance(float v)
{
	const float a = ceilf(fabsf(v));
	float b = a * (float)(v > 0 ? 1 : -1);
	if (b > 127.f) b = 127.f;
	return (i8)(int)b;
}

// value of a float loop variable after k steps of `+= 1` (the reference iterates floats; start + k may round differently)
TV_HD float edit_step(float start, int k)
{
	float x = start;
	for (int q = 0; q < k; ++q) x += 1.0f;
	return x;
}

// one voxel of the section: (ix,iy,iz) = loop indices along x,y,z
TV_HD void edit_voxel(const GridView& g, const EditParams& e, const EditSection& s, int ix, int iy, int iz)
{
	const float x = edit_step(s.start[0], ix), y = edit_step(s.start[1], iy), z = edit_step(s.start[2], iz);
	const size_t i = ((size_t)((u32)s.bmin[2] + (u32)z) * g.n + ((u32)s.bmin[1] + (u32)y)) * g.n + ((u32)s.bmin[0] + (u32)x);
	if (e.kind == EDIT_BALL) {
		// the voxel's sample is the one its running index meets in the brush's output (:405-441), see EditSection
		const int vi = (iz * s.count[1] + iy) * s.count[0] + ix;
		const int plane = s.scount[0] * s.scount[1];
		float sv = 0.f; // past the last sample the brush wrote (the reference reads unwritten floats there)
		if (plane > 0 && vi < plane * s.scount[2]) {
			const int jz = vi / plane, rem = vi - jz * plane, jy = rem / s.scount[0], jx = rem - jy * s.scount[0];
			const float sx = edit_step(s.sstart[0], jx), sy = edit_step(s.sstart[1], jy), sz = edit_step(s.sstart[2], jz);
			sv = sqrtf((sx * sx + sy * sy) + sz * sz) - e.radius;
		}
		i8* dist = const_cast<i8*>(g.dist);
		const float value = (float)dist[i];
		float r;
		if (e.type == 0) r = value < sv ? value : sv;        // IT_Add: min
		else if (e.type == 1) r = value > sv ? value : sv;   // IT_SubtractAddInner: max
		else r = (-sv > value) ? -sv : value;                // IT_Subtract: max(-surface, value)
		dist[i] = edit_round_distance(r);
	} else {
		u8* mat = const_cast<u8*>(g.mat);
		u8* blend = const_cast<u8*>(g.blend);
		const float coeff = (e.ext[0] / 2.0f) * 0.75f;
		const float cx = (x + s.bmin[0]) - e.pos[0], cy = (y + s.bmin[1]) - e.pos[1], cz = (z + s.bmin[2]) - e.pos[2];
		const float d = sqrtf((cx * cx + cy * cy) + cz * cz) / coeff;
		float w = 1 - d;
		w = w > 0.f ? w : 0.f; w = w < 1.f ? w : 1.f;
		const u8 outBlend = (u8)(w * 255.f);
		if (mat[i] == (u8)e.material) {
			int v = (e.add ? 1 : -1) * (int)outBlend + (int)blend[i];
			v = v < 255 ? v : 255; v = v > 0 ? v : 0;
			blend[i] = (u8)v;
		} else {
			mat[i] = (u8)e.material;
			blend[i] = outBlend;
		}
	}
}

// BF_Empty of a block by the codec's rule: the RLE (runs of at most 255) fits 4096 bytes, and every run value has
// strictly the sign of the first sample (walk in codec order: x, then y, then z)
TV_HD u8 edit_block_empty(const GridView& g, u32 bx, u32 by, u32 bz)
{
	const i8* base = g.dist + dist_offset(g, (int)(bx * 16), (int)(by * 16), (int)(bz * 16)); // whole grids and slabs alike
	const i8 first = base[0];
	i8 last = first;
	u32 counter = 0, size = 1;
	bool empty = true;
	for (u32 z = 0; z < 16; ++z)
	for (u32 y = 0; y < 16; ++y) {
		const i8* row = base + ((size_t)z * g.pitchY + y) * g.n;
		for (u32 x = 0; x < 16; ++x) {
			const i8 cur = row[x];
			if (last == cur && counter < 0xFF) { ++counter; continue; }
			size += 2; counter = 1; last = cur;
			if ((int)first * (int)last <= 0) empty = false;
			if (size > 4096) return 0;
		}
	}
	return empty ? 1 : 0;
}

// VoxelGrid(unsigned w, const char* heightmap), src/VoxelGrid.cpp:159-213: one distance sample
TV_HD i8 heightmap_distance(int z, int height)
{
	int h = (z - 127) - height;
	h = h < -127 ? -127 : (h > 127 ? 127 : h);
	return (i8)(h > 4 ? 4 : (h < -4 ? -4 : h)); // toGridDistValue (:42-50)
}

// CompressBlock (src/VoxelGrid.cpp:610-672) on one 4096-byte stream given as 256 rows of 16 bytes with stride
// `rowStride(row)`: returns the coded size (2 bytes per run, runs of at most 255) or 4096 when the code would not fit
// (then the stream is stored raw); with `out` the bytes are written too.
template <typename RowFn>
TV_HD u32 encode_stream_serial(const RowFn& rowPtr, u8* out, bool& raw)
{
	u32 runs = 1;
	{
		u8 last = rowPtr(0)[0];
		u32 counter = 0;
		for (u32 r = 0; r < 256; ++r) { const u8* row = rowPtr(r); for (u32 x = 0; x < 16; ++x) { const u8 cur = row[x]; if (last == cur && counter < 0xFF) { ++counter; continue; } ++runs; counter = 1; last = cur; } }
	}
	raw = runs > 2048;
	if (!out) return raw ? 4096u : 2u * runs;
	if (raw) { for (u32 r = 0; r < 256; ++r) { const u8* row = rowPtr(r); for (u32 x = 0; x < 16; ++x) out[r * 16 + x] = row[x]; } return 4096u; }
	u8 last = rowPtr(0)[0];
	u32 counter = 0, w = 0;
	for (u32 r = 0; r < 256; ++r) { const u8* row = rowPtr(r); for (u32 x = 0; x < 16; ++x) {
		const u8 cur = row[x];
		if (last == cur && counter < 0xFF) { ++counter; continue; }
		out[w++] = (u8)counter; out[w++] = last;
		counter = 1; last = cur;
	} }
	out[w++] = (u8)counter; out[w++] = last;
	return w;
}

} // namespace tv

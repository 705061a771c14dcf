// vx_hip.hip — gfx950 (MI355X / CDNA4) kernels and HIP backend of libvoxels_hip.so.
//
// A full run of a terrain-like grid is THREE launches on one stream:
//   k_run_head    what the BF_Empty flags and the sign summaries say about a block; the blocks that are not quiet get their
//                 level-0 slots and their ancestors the slots of the levels above (workgroup = an aligned 8 x 8 x 4 box)
//   k_main        (vx_main.inl) every block of every level: two queues handed out to persistent workgroups - the level-0
//                 slots, and [material | regular | transition] blocks of the levels >= 1 in dependency order, waiting for
//                 each other through per-block flags (write-through stores, polled loads, no fences)
//   k_tail        the general passes over what the table-driven blocks handed on, the result's block tables, the header
//                 into page-locked host memory, and the start state of the counters / maps the NEXT run will use
// (k_reset in front of the first run of a context).  Dense surfaces (blocks beyond the first capacity class), incremental
// runs and grids beyond 1024^3 run the same per-block bodies as a chain of kernels on three to five streams, no host round
// trip in between (work lists and output offsets live in device memory):
//   k_run_head    counters / slot maps reset; what the BF_Empty flags and the sign summaries say about a block
//   k_classify    stream over the density field, skipping blocks the flags prove quiet: 16-byte coalesced loads, sign
//                 bits packed to bit-masks in LDS, cells classified bit-parallel (256 per lane); emits the
//                 non-trivial-cell bitmap and an active slot for every surface-bearing level-0 block (small block
//                 ranges: also the ancestors' slots).
//   k_hierarchy   marks the ancestors of active blocks on the coarser LOD levels (large ranges, incremental runs).
//   k_material    (per level >= 1, serial) per-cell material vote over the 8 children -> material cache; flat list of
//                 the active blocks of the levels >= 1.
//   k_regular0_fast, k_regular1_fast (vx_fast0.inl, vx_fast1.inl)   the regular cells of level 0 / of the levels 1..3 for
//                 blocks without a zero sample: table-driven cells and triangles, one lane per vertex and per triangle,
//                 one 64-bit atomicAdd per mesh to reserve its ranges of the output pools (per-block stream
//                 compaction), streaming stores.  Level 0 runs beside the material chain on a side stream.
//   k_regular0, k_regular (vx_regular0.inl, here)   the general pass: what the table-driven passes hand on (zero samples,
//                 LOD chains ending on a voxel), the 4096-cell class, levels beyond the lattice copies, incremental runs.
//   k_transition  the 6 x 16 x 16 transition cells of the blocks of levels 1..last-1 (third stream).
//   k_list_count, k_list_write   the result's block tables.
//   k_classify_blocks, k_build_worklist, k_gather_records   incremental (Modification) runs.
//   k_decode_grid Grid file format v1 -> dense fields (vx_grid_upload_packed).
// The per-cell logic is tv_core.h / tv_block.h (shared with the CPU emulation used by the tests).
//
// No MFMA: this is table-driven integer/byte work bound by HBM bandwidth and latency.
#include <hip/hip_runtime.h>
#include <hip/hip_ext.h>

#include <dlfcn.h>

#include <algorithm>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <string>
#include <type_traits>
#include <vector>

#include "tv_block.h"
#include "tv_fast0.h"
#include "tv_fast1.h"
#include "vx_terrain_math.h"

#define VX_BACKEND_NAME "hip:gfx950"

namespace {
using namespace tv;

struct ExecParamsDev {
	Globals G;
	LevelDesc levels[MAX_LEVELS];
	Pools P;
	const u8* tables;
};

// The kernel's first argument read afresh through the kernarg segment: a pointer the compiler cannot see through, so nothing
// loaded from the parameters BEFORE this point is kept alive across it (a persistent kernel otherwise holds every pointer it
// ever uses in scalar registers for its whole life - 250 of them spilled to vector lanes and read back with v_readlane in
// k_main) and everything behind it is loaded again, 16 dwords per s_load.  Only inside kernels whose FIRST parameter is the
// ExecParamsDev (k_main, k_regular0_fast, k_dirty_regular0_fast: the callers of f0_walk).  Round 6: 963 -> 375 v_readlane in
// k_main, 126.2M -> 120.8M vector instructions per launch at 1024^3, the step -1.3 %.
__device__ __forceinline__ const ExecParamsDev& kernarg_params()
{
	typedef const __attribute__((address_space(4))) ExecParamsDev* KP;
	KP kp = (KP)__builtin_amdgcn_kernarg_segment_ptr();
	asm volatile("" : "+s"(kp));
	return *(const ExecParamsDev*)kp;
}

// tools builds (tools/ab_build.py x=-DVX_ABL=<bits>): parts of the work switched off to see what they cost in time and
// instructions (tools/exp/r06_ablate.sh); the results of such a build are wrong by construction.  0 in the product.
#if !defined(VX_ABL)
#define VX_ABL 0
#endif

constexpr int WG = 256;
constexpr int REG_CAP_SMALL = LARGE_THRESHOLD; // LDS capacity class of the regular pass that covers ordinary surfaces
constexpr int REG_CAP_MID = 1536;               // second class of the table-driven passes (dense surfaces: three workgroups per CU instead of one in the 4096-cell class)
constexpr int REG_CAP_BIG = 4096;               // third class of the table-driven passes: every block (two workgroups per CU; round 4 left blocks beyond 1536 cells to the general pass at one workgroup per CU)

// ------------------------------------------------------------------------------------------------------
// workgroup helpers
// ------------------------------------------------------------------------------------------------------
__device__ __forceinline__ u32 wave_inclusive_scan(u32 v)
{
	const int lane = threadIdx.x & 63;
#pragma unroll
	for (int off = 1; off < 64; off <<= 1) {
		const u32 t = __shfl_up(v, off, 64);
		if (lane >= off) v += t;
	}
	return v;
}

// in-place exclusive scan of a[0..n) (LDS), all 256 threads must call; returns the total
__device__ u32 block_exclusive_scan_u16(u16* a, u32 n, u32* scratch)
{
	u32 tid = threadIdx.x;
	asm volatile("" : "+v"(tid)); // (inside a persistent kernel's block loop: the lane's addresses are formed here, not ahead of the loop)
	const u32 per = (n + WG - 1) / WG;
	u32 beg = tid * per, end = beg + per;
	if (beg > n) beg = n;
	if (end > n) end = n;
	u32 sum = 0;
	for (u32 i = beg; i < end; ++i) sum += a[i];
	const u32 incl = wave_inclusive_scan(sum);
	if ((tid & 63) == 63) scratch[tid >> 6] = incl;
	__syncthreads();
	u32 waveBase = 0, total = 0;
#pragma unroll
	for (u32 w = 0; w < WG / 64; ++w) {
		const u32 s = scratch[w];
		if (w < (tid >> 6)) waveBase += s;
		total += s;
	}
	u32 run = waveBase + incl - sum;
	for (u32 i = beg; i < end; ++i) { const u32 v = a[i]; a[i] = (u16)run; run += v; }
	__syncthreads();
	return total;
}

// both output ranges of a mesh with ONE returning atomic: the cursors are the halves of an aligned 64-bit word (the pools
// hold fewer than 2^32 elements, so the vertex half never carries into the index half).  Every block of every pass
// reserves through the same two addresses — about 30 reservations per microsecond during a 1024^3 run, a third of
// what one address sustains — so the number of round trips matters more than the bytes.
__device__ __forceinline__ void reserve_both(u32* cursors, u32 verts, u32 indices, u32& vOff, u32& iOff)
{
	const unsigned long long r = atomicAdd((unsigned long long*)cursors, (unsigned long long)verts | ((unsigned long long)indices << 32));
	vOff = (u32)r; iOff = (u32)(r >> 32);
}

// ---- dependencies between workgroups of ONE launch (k_main): LevelDesc::matDone ----------------------------------------
// Producer: the payload leaves through write-through (sc1) stores, every storing wave drains them, the workgroup meets and
// ONE lane stores the 8-byte word epoch << 32 | payload.  Consumer: ONE lane polls the word (relaxed, agent scope, s_sleep
// between polls), the workgroup meets, and the payload is read with write-through loads (TV_LOAD_THROUGH / load16_through:
// past the CU's L1, which no other CU's store refreshes).  Results do not depend on dispatch order, timing or placement;
// every wait is bounded (Globals::giveUp fails the run instead of hanging the device).
// (Why not plain stores + an agent-scope release fence: the fence writes back EVERY dirty line of the XCD's L2 - beside
// level-0 blocks that dirty hundreds of MB per run, thousands of fences per run cost more than the launches they replace.)
typedef unsigned int v4u32 __attribute__((ext_vector_type(4)));
__device__ __forceinline__ void store16_through(void* uniformBase, u32 byteOffset, uint4 v)
{
	const auto rsrc = __builtin_amdgcn_make_buffer_rsrc(uniformBase, 0, 0x7FFFFFFF, 0x00020000);
	v4u32 x = { v.x, v.y, v.z, v.w };
	__builtin_amdgcn_raw_buffer_store_b128(x, rsrc, (int)byteOffset, 0, /* aux: sc1 */ 16);
}
// -DVX_CONSERVATIVE_SYNC (libvoxels_hip_conservative.so, a test build of the same sources): the textbook form of the same
// protocol - every producer wave releases at agent scope before the barrier, the flag is a release store, the poll an acquire
// load, and every consumer wave acquires behind the barrier - so that nothing rests on which stores and loads were written
// as write-through ones.  tests/test_gpu_parity.py runs both libraries on the same grids and compares every byte: a payload
// access the fast protocol forgot to route past the caches shows as a difference there (or in the stress runs).
__device__ __forceinline__ void publish_done_through(unsigned long long* flag, u32 epoch, u32 payload)
{
#if defined(VX_CONSERVATIVE_SYNC)
	__threadfence();
	__syncthreads();
	if (threadIdx.x == 0) __hip_atomic_store(flag, ((unsigned long long)epoch << 32) | payload, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_AGENT);
#else
	asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); // every storing wave: its write-through stores have left
	__syncthreads();
	if (threadIdx.x == 0) __hip_atomic_store(flag, ((unsigned long long)epoch << 32) | payload, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
#endif
}

enum { WAIT_SPINS = 1u << 17 }; // x ~1 us per poll: a tenth of a second, against runs of a millisecond

#if defined(VX_MAIN_TRACE)
// tools builds: a timeline of k_main's items (100 MHz clock) - per item {kind << 28 | level << 24 | slot, workgroup, ticket drawn,
// started, last wait over, done}; printed by vx_polygonize after the run (small grids)
__device__ unsigned long long g_mainTrace[18 * 8192];
__device__ u32 g_mainTraceN;
__device__ unsigned long long g_waitEnd[4096];
__device__ unsigned long long g_marks[4096 * 12];
#define TRACE_MARK(i) do { if (threadIdx.x == 0) g_marks[(blockIdx.x & 4095u) * 12u + (i)] = __builtin_amdgcn_s_memrealtime(); } while (0)
#else
#define TRACE_MARK(i) do { } while (0)
#endif
// one lane: poll until the word carries this run's tag; returns its payload (0 after giving up).  Once one wait has given
// up every other one does so at its next look (the run is lost; it must end, not hang the device).
__device__ __forceinline__ u32 wait_done(const unsigned long long* flag, u32 epoch, u32* giveUp)
{
	for (u32 spins = 0;; ++spins) {
#if defined(VX_CONSERVATIVE_SYNC)
		const unsigned long long v = __hip_atomic_load(flag, __ATOMIC_ACQUIRE, __HIP_MEMORY_SCOPE_AGENT);
#else
		const unsigned long long v = __hip_atomic_load(flag, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
#endif
#if defined(VX_MAIN_TRACE)
		if ((u32)(v >> 32) == epoch) { g_waitEnd[blockIdx.x & 4095u] = __builtin_amdgcn_s_memrealtime(); return (u32)v; }
#endif
		if ((u32)(v >> 32) == epoch) return (u32)v;
		if (spins > (u32)WAIT_SPINS || ((spins & 255u) == 255u && __hip_atomic_load(giveUp, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) != 0u)) { atomicOr(giveUp, 1u); return 0u; }
		__builtin_amdgcn_s_sleep(4);
	}
}

// behind the polls of ONE wave (the others wait at the barrier).  What the producers published left through write-through
// stores and is read through TV_LOAD_THROUGH / load16_through (past the L1), so no acquire - an invalidation of the whole
// L1 of the CU, also under the other workgroups running there - is needed (measured: plain stores + release fence + acquire
// 0.553 ms per step at 1024^3, write-through stores + acquire 0.461, write-through stores and loads 0.445).
__device__ __forceinline__ void acquire_and_meet(bool polled)
{
	(void)polled;
	__syncthreads();
#if defined(VX_CONSERVATIVE_SYNC)
	__threadfence(); // every wave that is about to read what the producers published
#endif
}

__device__ __forceinline__ uint4 load16_through(const void* uniformBase, u32 byteOffset)
{
	const auto rsrc = __builtin_amdgcn_make_buffer_rsrc(const_cast<void*>(uniformBase), 0, 0x7FFFFFFF, 0x00020000);
	const v4u32 x = __builtin_amdgcn_raw_buffer_load_b128(rsrc, (int)byteOffset, 0, /* aux: sc1 */ 16);
	return make_uint4(x.x, x.y, x.z, x.w);
}

__device__ __forceinline__ void stage_tables(u8* dst, const u8* src)
{
	const uint4* s = (const uint4*)src;
	uint4* d = (uint4*)dst;
	for (u32 i = threadIdx.x; i < TAB_BYTES / 16; i += WG) d[i] = s[i];
}

// Gather COUNT elements with 256 threads so that ALL loads of a thread are in flight before the first use: the
// plain "for (q = tid; q < COUNT; q += 256) lds[q] = load(q)" form is compiled into one dependent round trip per
// iteration (load, s_waitcnt, ds_write), which made every staging loop latency-bound.
template <int COUNT, typename T, int BATCH, typename LoadFn, typename StoreFn>
__device__ __forceinline__ void batched_gather(LoadFn load, StoreFn store)
{
	constexpr int ITER = (COUNT + WG - 1) / WG;
#pragma unroll 1
	for (int i0 = 0; i0 < ITER; i0 += BATCH) {
		T v[BATCH];
#pragma unroll
		for (int i = 0; i < BATCH; ++i) {
			const int q = (int)threadIdx.x + (i0 + i) * WG;
			if (q < COUNT) v[i] = load(q);
		}
#pragma unroll
		for (int i = 0; i < BATCH; ++i) {
			const int q = (int)threadIdx.x + (i0 + i) * WG;
			if (q < COUNT) store(q, v[i]);
		}
	}
}

// 17 consecutive lattice samples of level `level` starting at lattice point (X0, Y, Z), X0 a multiple of 16
struct PyramidRow { uint4 lo; u32 far; };
__device__ __forceinline__ PyramidRow pyramid_row17(const PyramidLevel& P, int X0, int Y, int Z)
{
	const i8* src = P.data + pyramid_offset(P, X0, Y, Z);
	PyramidRow r;
	r.lo = *(const uint4*)src;
	r.far = *(const u8*)(src + BRICK_BYTES); // sample X0 + 16: the same row of the next brick
	return r;
}

// GPU forms of the staging phases of tv_block.h (same results, batched loads)
__device__ __forceinline__ void gpu_stage_samples17(const GridView& g, u32 bx, u32 by, u32 bz, u32 mult, i8* samp)
{
	batched_gather<SAMPLES, i8, 10>(
		[&](int s) { const int i = s % 17, j = (s / 17) % 17, k = s / 289;
		             return (i8)dist_at(g, (int)((bx * 16 + i) * mult), (int)((by * 16 + j) * mult), (int)((bz * 16 + k) * mult)); },
		[&](int s, i8 v) { samp[s] = v; });
}

__device__ __forceinline__ void gpu_reg_stage(const Globals& G, const RegBlockCtx& b, i8* samp)
{
	const GridView& g = G.grid;
	if (b.level >= 1 && b.level < PYRAMID_LEVELS && G.pyr[b.level].data) {
		// the level's lattice copy: 17 contiguous samples per row
		const PyramidLevel& P = G.pyr[b.level];
		for (int r = (int)threadIdx.x; r < 289; r += WG) {
			const int k = r / 17, j = r - k * 17;
			const PyramidRow row = pyramid_row17(P, (int)(b.bx * 16), (int)(b.by * 16) + j, (int)(b.bz * 16) + k);
			u32* dst = (u32*)(samp + samp_index(0, j, k));
			dst[0] = row.lo.x; dst[1] = row.lo.y; dst[2] = row.lo.z; dst[3] = row.lo.w;
			samp[samp_index(16, j, k)] = (i8)row.far;
		}
	} else {
		batched_gather<SAMPLES, i8, 5>(
			[&](int s) { const int i = s % 17, j = (s / 17) % 17, k = s / 289;
			             return (i8)dist_at(g, (int)((b.bx * 16 + i) * b.mult), (int)((b.by * 16 + j) * b.mult), (int)((b.bz * 16 + k) * b.mult)); },
			[&](int s, i8 v) { const int i = s % 17, j = (s / 17) % 17, k = s / 289; samp[samp_index(i, j, k)] = v; });
	}
}

// ------------------------------------------------------------------------------------------------------
// k_classify
// ------------------------------------------------------------------------------------------------------
__device__ __forceinline__ u32 sign_nibble(u32 d) { return (((d & 0x80808080u) >> 7) * 0x01020408u) >> 24; }

// ------------------------------------------------------------------------------------------------------
// k_decode_grid: Grid file format v1 -> dense fields (DecompressBlock, src/VoxelGrid.cpp:674-694, for all blocks at
// once).  One workgroup per EIGHT x-neighbour blocks, one stream (distance, material, blend) after the other.
//   * Streams of up to 32 runs - nearly every stream of a terrain: a run is at most 255 voxels long, so a CONSTANT block is
//     17 runs, which is why "a handful of runs" has to mean 32 and not 16 - are walked from LDS: the workgroup fetches the
//     records' places, then the flag words and the first 64 bytes of all 24 streams (two dependent round trips for eight
//     blocks), and lane = (block tid & 7, row group tid >> 3) fills its 8 voxel rows - a row inside one run without a walk -
//     so that the eight lanes of a row write the 128-byte line the eight blocks share in the dense field.
//   * The other streams, one after the other with the whole workgroup: run lengths -> exclusive scan = run starts; lane t
//     owns the 16-byte voxel row t of the block (y = t & 15, z = t >> 4), finds the run that covers its first voxel by
//     binary search and walks on from there; raw streams are copied.
// ------------------------------------------------------------------------------------------------------
__device__ __forceinline__ void decode_stream_whole_block(const u8* src, u32 sz, bool raw, u8* out, u16* starts, u8* vals, u32* scanScratch, const u32 tid)
{
	u32 b[4] = { 0, 0, 0, 0 };
	if (raw) {
		// raw: 4096 bytes, row t at offset 16 t (the stream itself is not aligned)
#pragma unroll
		for (u32 i = 0; i < 16; ++i) if (tid * 16 + i < sz) b[i >> 2] |= (u32)src[tid * 16 + i] << ((i & 3) * 8);
	} else {
		const u32 pairs = sz >> 1; // <= 2048
#pragma unroll
		for (u32 i = 0; i < 8; ++i) {
			const u32 q = tid * 8 + i;
			u32 len = 0, val = 0;
			if (q < pairs) { len = src[2 * q]; val = src[2 * q + 1]; }
			starts[q] = (u16)len;
			vals[q] = (u8)val;
		}
		__syncthreads();
		const u32 total = block_exclusive_scan_u16(starts, pairs, scanScratch);
		// last run whose start is <= the row's first voxel
		const u32 q0 = tid * 16;
		u32 lo = 0, hi = pairs; // invariant: starts[lo] <= q0 (starts[0] == 0), answer in [lo, hi)
		while (hi - lo > 1) {
			const u32 mid = (lo + hi) >> 1;
			if (starts[mid] <= q0) lo = mid; else hi = mid;
		}
		u32 r = lo;
#pragma unroll
		for (u32 i = 0; i < 16; ++i) {
			const u32 pos = q0 + i;
			while (r + 1 < pairs && starts[r + 1] <= pos) ++r;
			const u32 v = (pairs && pos < total) ? vals[r] : 0u;
			b[i >> 2] |= v << ((i & 3) * 8);
		}
		__syncthreads(); // the run tables are reused by the next stream
	}
	*(uint4*)out = make_uint4(b[0], b[1], b[2], b[3]);
}

constexpr u32 DECODE_SHORT_RUNS = 32; // streams up to this many runs are walked from LDS (a constant block: 17 runs of <= 255 voxels)

__global__ __launch_bounds__(WG) void k_decode_grid(const u8* blob, const unsigned long long* where, u32 n, i8* dist, u8* mat, u8* blend, u8* flags)
{
	__shared__ u16 starts[2048 + 8];
	__shared__ u8 vals[2048];
	__shared__ u32 scanScratch[8];
	__shared__ unsigned long long recAt[8], recSizes[8]; // per block of this workgroup: offset of its record, its three stream sizes
	__shared__ u32 recFlags[8];
	__shared__ u8 runs[8][3][2 * DECODE_SHORT_RUNS];     // the first runs of every stream (all of a short one)
	__shared__ u32 wholeMask[3];                         // per stream: the blocks that take the whole-workgroup path
	const u32 nb = n >> 4, tid = threadIdx.x, groups = (nb + 7u) >> 3;
	const u32 gx = blockIdx.x % groups, by = (blockIdx.x / groups) % nb, bz = blockIdx.x / (groups * nb);
	const u32 firstId = (bz * nb + by) * nb + gx * 8u, count = min(8u, nb - gx * 8u);
	// two dependent round trips in all: the records' places, then their flags and the heads of their streams
	if (tid < 16) { const u32 b = tid >> 1; const unsigned long long v = b < count ? where[2 * (size_t)(firstId + b) + (tid & 1u)] : 0ull; if (tid & 1u) recSizes[b] = v; else recAt[b] = v; }
	if (tid < 3) wholeMask[tid] = 0;
	__syncthreads();
	constexpr u32 PER_BLOCK = 4u + 3u * 2u * DECODE_SHORT_RUNS;
	for (u32 q = tid; q < 8u * PER_BLOCK; q += WG) {
		const u32 b = q / PER_BLOCK, j = q % PER_BLOCK; // j < 4: a byte of the record's flag word; else stream (j - 4) / 64, byte (j - 4) % 64
		if (b >= count) continue;
		const u8* rec = blob + recAt[b];
		if (j < 4u) { ((u8*)&recFlags[b])[j] = rec[j]; continue; }
		const u32 st = (j - 4u) / (2u * DECODE_SHORT_RUNS), at = (j - 4u) % (2u * DECODE_SHORT_RUNS);
		const unsigned long long sizes = recSizes[b];
		u32 before = 4;
		for (u32 t = 0; t < st; ++t) before += (u32)((sizes >> (16 * t)) & 0xFFFFu);
		if (at < (u32)((sizes >> (16 * st)) & 0xFFFFu)) runs[b][st][at] = rec[before + at];
	}
	__syncthreads();
	if (tid < 8 && tid < count) {
		flags[firstId + tid] = (u8)(recFlags[tid] & 1u);
		for (u32 st = 0; st < 3; ++st) {
			const u32 sz = (u32)((recSizes[tid] >> (16 * st)) & 0xFFFFu);
			if (((recFlags[tid] >> (st + 1)) & 1u) || (sz >> 1) > DECODE_SHORT_RUNS) atomicOr(&wholeMask[st], 1u << tid);
		}
	}
	__syncthreads();
	const u32 blk = tid & 7u, bx = gx * 8u + blk;
#pragma unroll 1
	for (u32 s = 0; s < 3; ++s) {
		u8* field = s == 0 ? (u8*)dist : (s == 1 ? mat : blend);
		const u32 whole = wholeMask[s];
		if (blk < count && !((whole >> blk) & 1u)) {
			// the lane walks the runs (LDS, uniform among the lanes of a block) for its rows it * 32 + (tid >> 3), ascending
			const u32 pairs = (u32)((recSizes[blk] >> (16 * s)) & 0xFFFFu) >> 1;
			const u8* rl = runs[blk][s];
			u32 r = 0, end = pairs ? (u32)rl[0] : 0u; // run r ends in front of voxel `end`
#pragma unroll 1
			for (u32 it = 0; it < 8; ++it) {
				const u32 row = it * 32u + (tid >> 3), q0 = row * 16u;
				u32 b[4] = { 0, 0, 0, 0 };
				while (r < pairs && q0 >= end) { ++r; if (r < pairs) end += (u32)rl[2 * r]; }
				if (r >= pairs || q0 + 16u <= end) {
					// the whole row lies in one run (nearly every row of such a block), or behind the last one
					const u32 x = r < pairs ? (u32)rl[2 * r + 1] * 0x01010101u : 0u;
					b[0] = b[1] = b[2] = b[3] = x;
				} else {
#pragma unroll
					for (u32 i = 0; i < 16; ++i) {
						const u32 pos = q0 + i;
						while (r < pairs && pos >= end) { ++r; if (r < pairs) end += (u32)rl[2 * r]; }
						const u32 x = r < pairs ? (u32)rl[2 * r + 1] : 0u;
						b[i >> 2] |= x << ((i & 3) * 8);
					}
				}
				*(uint4*)(field + ((size_t)(bz * 16u + (row >> 4)) * n + by * 16u + (row & 15u)) * n + bx * 16u) = make_uint4(b[0], b[1], b[2], b[3]);
			}
		}
		// the blocks with longer run lists or raw streams: one at a time, all lanes (uniform loop: the mask is the workgroup's)
		u32 todo = whole;
		while (todo) {
			const u32 k = (u32)__builtin_ctz(todo);
			todo &= todo - 1u;
			const unsigned long long sizesk = recSizes[k];
			u32 before = 4;
			for (u32 t = 0; t < s; ++t) before += (u32)((sizesk >> (16 * t)) & 0xFFFFu);
			const size_t rowOff = ((size_t)(bz * 16u + (tid >> 4)) * n + by * 16u + (tid & 15u)) * n + (gx * 8u + k) * 16u;
			decode_stream_whole_block(blob + recAt[k] + before, (u32)((sizesk >> (16 * s)) & 0xFFFFu), ((recFlags[k] >> (s + 1)) & 1u) != 0u, field + rowOff, starts, vals, scanScratch, tid);
		}
	}
}

// ------------------------------------------------------------------------------------------------------
// k_edit / k_edit_flags: Grid::InjectSurface (analytic ball) and Grid::InjectMaterial on the resident grid, one
// workgroup per touched block; then BF_Empty of the touched blocks, one lane per block (the codec walk is serial).
// ------------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(WG) void k_edit(GridView g, const u32* ids, EditParams e)
{
	const u32 nb = (u32)g.n / 16, id = ids[blockIdx.x];
	const EditSection s = edit_section(e, id % nb, (id / nb) % nb, id / (nb * nb));
	const int total = s.count[0] * s.count[1] * s.count[2];
	for (int v = threadIdx.x; v < total; v += WG) {
		const int ix = v % s.count[0], iy = (v / s.count[0]) % s.count[1], iz = v / (s.count[0] * s.count[1]);
		edit_voxel(g, e, s, ix, iy, iz);
	}
}

// The ids of a box of blocks (an edit's touched blocks, run_edit of vx_host.inl), z-major: id = (z * nb + y) * nb + x.
__global__ __launch_bounds__(WG) void k_box_ids(u32* out, u32 x0, u32 y0, u32 z0, u32 nx, u32 ny, u32 total, u32 nb)
{
	const u32 i = blockIdx.x * WG + threadIdx.x;
	if (i >= total) return;
	const u32 x = x0 + i % nx, y = y0 + (i / nx) % ny, z = z0 + i / (nx * ny);
	out[i] = (z * nb + y) * nb + x;
}

// BF_Empty by the codec's rule (edit_block_empty of tv_block.h walks the block serially), one workgroup per block.
// A run starts where the value changes and every 255 voxels inside a constant stretch; the RLE stays "effective"
// while it has at most 2048 runs; empty <=> effective and every sample has strictly the sign of the first.
__global__ __launch_bounds__(WG) void k_edit_flags(GridView g, u8* flags, const u32* ids, u32 count)
{
	__shared__ i8 lastOfRow[WG];
	__shared__ int waveMax[WG / 64];
	__shared__ u32 runs;
	const u32 nb = (u32)g.n / 16, id = ids ? ids[blockIdx.x] : blockIdx.x, t = threadIdx.x; // no list = every block
	const u32 bx = id % nb, by = (id / nb) % nb, bz = id / (nb * nb);
	const i8* base = g.dist + dist_offset(g, (int)(bx * 16), (int)(by * 16), (int)(bz * 16)); // whole grids and slabs alike
	const uint4 raw = *(const uint4*)(base + ((size_t)(t >> 4) * g.pitchY + (t & 15)) * g.n); // row t = (y = t & 15, z = t >> 4): codec order
	i8 v[16];
	memcpy(v, &raw, 16);
	lastOfRow[t] = v[15];
	if (t == 0) runs = 0;
	__syncthreads();
	const i8 first = base[0];
	i8 prev = t ? lastOfRow[t - 1] : (i8)~v[0]; // voxel 0 always starts a run
	int lastStart = -1;                         // last position in this row where the value changes
	bool sameSign = true;
#pragma unroll
	for (int j = 0; j < 16; ++j) {
		if (v[j] != prev) lastStart = (int)t * 16 + j;
		prev = v[j];
		sameSign = sameSign && ((int)first * (int)v[j] > 0);
	}
	// start of the stretch that is open when this row begins = max of lastStart over the earlier rows
	int incl = lastStart;
#pragma unroll
	for (int d = 1; d < 64; d <<= 1) { const int o = __shfl_up(incl, d); if ((int)(t & 63) >= d) incl = max(incl, o); }
	if ((t & 63) == 63) waveMax[t >> 6] = incl;
	__syncthreads();
	int carry = __shfl_up(incl, 1);
	if ((t & 63) == 0) carry = -1;
	for (u32 w = 0; w < (t >> 6); ++w) carry = max(carry, waveMax[w]);
	u32 myRuns = 0;
	int start = carry;
	prev = t ? lastOfRow[t - 1] : (i8)~v[0];
#pragma unroll
	for (int j = 0; j < 16; ++j) {
		const int pos = (int)t * 16 + j;
		if (v[j] != prev) start = pos;
		prev = v[j];
		if ((pos - start) % 255 == 0) ++myRuns;
	}
	if (myRuns) atomicAdd(&runs, myRuns);
	const int allSame = __syncthreads_and(sameSign ? 1 : 0);
	if (t == 0) flags[id] = (allSame && runs <= 2048u) ? 1 : 0;
}

// k_copy_segments: pool compaction — segment i moves `count` elements from src[srcOff] to dst[dstOff]; elements are
// 48-byte vertices or 4-byte indices, moved as dwords (both pools are at least 4-byte aligned at any element)
__global__ __launch_bounds__(WG) void k_copy_segments(const u32* seg, const u32* src, u32* dst, u32 dwordsPerElem)
{
	const u32 s = seg[blockIdx.x * 3] * dwordsPerElem, d = seg[blockIdx.x * 3 + 1] * dwordsPerElem, n = seg[blockIdx.x * 3 + 2] * dwordsPerElem;
	for (u32 i = threadIdx.x; i < n; i += WG) dst[d + i] = src[s + i];
}

// ------------------------------------------------------------------------------------------------------
// k_encode_grid: dense fields -> Grid file format v1 (CompressBlock, src/VoxelGrid.cpp:610-672, for every stream of every
// block).  One workgroup per EIGHT x-neighbour blocks, like k_decode_grid: a voxel row of the eight blocks is one 128-byte line
// of the dense field, and lane (block tid & 7, row group tid >> 3) fetches eight of its block's rows - whole lines per
// request, where one block per workgroup read an eighth of every line it touched (and its seven x-neighbours, dealt to the
// other seven XCDs, fetched the same lines again: 3.45 ms per pass at 1024^3 for 3.2 GB of fields).
//   * A stream whose 4096 bytes are one value - most streams of a terrain - is 16 runs of 255 and one of 16: written without
//     a scan.  Pass 1 leaves "constant, value" per stream in the block's fourth meta word; pass 2 does not read such a
//     stream again.
//   * Any other stream goes through the codec's rule with the whole workgroup, block by block, out of LDS: lane t owns the
//     16-byte row t (codec order).  Run starts = value changes and every 255 voxels inside a constant stretch (start of the
//     stretch by a workgroup-wide max-scan); run ids by an exclusive scan of the start counts; a run's length is the
//     distance to the next start; beyond 2048 runs the stream is stored raw.
// Pass 1 (blob == nullptr) writes sizes + flags, pass 2 writes the records at the offsets the host derived from them.
// ------------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(WG) void k_encode_grid(GridView g, u32* meta, const unsigned long long* where, u8* blob)
{
	__shared__ __attribute__((aligned(16))) u8 rowsLds[8][4096]; // the current stream of the eight blocks, codec order
	__shared__ u8 lastOfRow[WG];
	__shared__ int waveMax[WG / 64];
	__shared__ u32 waveSum[WG / 64];
	__shared__ u16 startPos[2048 + 2];
	__shared__ u8 vals[2048];
	__shared__ u32 notConst[8];  // per block: the current stream holds more than one value
	__shared__ u32 blockFlags[8]; // per block: BF_Empty | raw bits
	__shared__ u32 recOff[8];     // per block: bytes of the record's streams written so far
	__shared__ u32 hint[8];       // pass 2: the fourth meta word pass 1 left
	const u32 n = (u32)g.n, nb = n >> 4, t = threadIdx.x, groups = (nb + 7u) >> 3;
	const u32 gx = blockIdx.x % groups, by = (blockIdx.x / groups) % nb, bz = blockIdx.x / (groups * nb);
	const u32 firstId = (bz * nb + by) * nb + gx * 8u, count = min(8u, nb - gx * 8u);
	const u32 blk = t & 7u, rg = t >> 3;
	if (t < 8) { blockFlags[t] = 0; recOff[t] = 0; hint[t] = (blob && t < count) ? meta[(firstId + t) * 4u + 3u] : 0u; }
	__syncthreads();
#pragma unroll 1
	for (u32 s = 0; s < 3; ++s) {
		const u8* field = s == 0 ? (const u8*)g.dist : (s == 1 ? g.mat : g.blend);
		// pass 2: a stream pass 1 found constant is not read again
		const u32 myHint = hint[blk];
		const bool known = blob && ((myHint >> (4u + s)) & 1u);
		uint4 r[8];
#pragma unroll
		for (u32 it = 0; it < 8; ++it) {
			const u32 row = it * 32u + rg;
			r[it] = make_uint4(0, 0, 0, 0);
			if (blk < count && !known) r[it] = *(const uint4*)(field + ((size_t)(bz * 16u + (row >> 4)) * n + by * 16u + (row & 15u)) * n + (gx * 8u + blk) * 16u);
		}
		__syncthreads(); // the rows of the previous stream are no longer read
#pragma unroll
		for (u32 it = 0; it < 8; ++it) *(uint4*)&rowsLds[blk][(it * 32u + rg) * 16u] = r[it];
		if (t < 8) notConst[t] = 0;
		__syncthreads();
		const u32 first = known ? ((myHint >> (8u + 8u * s)) & 0xFFu) : (u32)rowsLds[blk][0];
		{
			const u32 f4 = first * 0x01010101u;
			u32 diff = 0;
#pragma unroll
			for (u32 it = 0; it < 8; ++it) diff |= (r[it].x ^ f4) | (r[it].y ^ f4) | (r[it].z ^ f4) | (r[it].w ^ f4);
			const unsigned long long m = __ballot(diff != 0u && !known && blk < count);
			if ((t & 63u) < 8u && ((m >> (t & 63u)) & 0x0101010101010101ull)) notConst[t & 63u] = 1u; // (lanes b, b + 8, ... of a wave belong to block b)
		}
		__syncthreads();
		// ---- constant streams: 16 x { 255, v } + { 16, v }; BF_Empty <=> the value is not zero -------------------------
		if (t < 8 && t < count && !notConst[t]) {
			if (!blob) {
				meta[(firstId + t) * 4u + s] = 34u;
				hint[t] |= (1u << (4u + s)) | (first << (8u + 8u * s)); // (pass 1: collected here, stored with the flags)
			}
			if (s == 0 && first != 0u) blockFlags[t] |= 1u;
		}
		if (blob && t < 8u * 17u) {
			const u32 b = t / 17u, k = t % 17u;
			if (b < count && !notConst[b]) {
				const u32 h = hint[b];
				const u32 v = ((h >> (4u + s)) & 1u) ? ((h >> (8u + 8u * s)) & 0xFFu) : (u32)rowsLds[b][0];
				u8* dst = blob + where[firstId + b] + 4 + recOff[b];
				dst[2 * k] = (u8)(k < 16u ? 255u : 16u);
				dst[2 * k + 1] = (u8)v;
			}
		}
		__syncthreads();
		if (t < 8 && t < count && !notConst[t]) recOff[t] += 34u;
		// ---- the other streams: one block at a time, all lanes (uniform loop: the flags are the workgroup's) -------------
		for (u32 b = 0; b < count; ++b) {
			if (!notConst[b]) continue;
			__syncthreads(); // LDS scratch of the previous block is free
			u8 v[16];
			{
				const uint4 rawRow = *(const uint4*)&rowsLds[b][t * 16u];
				memcpy(v, &rawRow, 16);
			}
			u8* dst = blob ? blob + where[firstId + b] + 4 + recOff[b] : nullptr;
			lastOfRow[t] = v[15];
			__syncthreads();
			u8 prev = t ? lastOfRow[t - 1] : (u8)~v[0]; // voxel 0 always starts a run
			int lastStart = -1;
#pragma unroll
			for (int j = 0; j < 16; ++j) { if (v[j] != prev) lastStart = (int)t * 16 + j; prev = v[j]; }
			int incl = lastStart;
#pragma unroll
			for (int d = 1; d < 64; d <<= 1) { const int o = __shfl_up(incl, d); if ((int)(t & 63) >= d) incl = max(incl, o); }
			if ((t & 63) == 63) waveMax[t >> 6] = incl;
			__syncthreads();
			int start = __shfl_up(incl, 1);
			if ((t & 63) == 0) start = -1;
			for (u32 w = 0; w < (t >> 6); ++w) start = max(start, waveMax[w]);
			u32 startMask = 0; // which of this row's voxels start a run
			prev = t ? lastOfRow[t - 1] : (u8)~v[0];
#pragma unroll
			for (int j = 0; j < 16; ++j) {
				const int pos = (int)t * 16 + j;
				if (v[j] != prev) start = pos;
				prev = v[j];
				if ((pos - start) % 255 == 0) startMask |= 1u << j;
			}
			// exclusive scan of the start counts -> id of this row's first run; total = number of runs
			const u32 mine = (u32)__popc(startMask);
			u32 inclSum = mine;
#pragma unroll
			for (int d = 1; d < 64; d <<= 1) { const u32 o = __shfl_up(inclSum, d); if ((int)(t & 63) >= d) inclSum += o; }
			if ((t & 63) == 63) waveSum[t >> 6] = inclSum;
			__syncthreads();
			u32 base = inclSum - mine, runs = 0;
			for (u32 w = 0; w < WG / 64; ++w) { if (w < (t >> 6)) base += waveSum[w]; runs += waveSum[w]; }
			const bool raw = runs > 2048u;
			const u32 sz = raw ? 4096u : 2u * runs;
			if (!blob) {
				if (t == 0) meta[(firstId + b) * 4u + s] = sz;
			} else if (raw) {
#pragma unroll
				for (int j = 0; j < 16; ++j) dst[t * 16 + j] = v[j];
			} else {
				u32 q = base;
				u32 m = startMask;
				while (m) {
					const int j = __builtin_ctz(m);
					m &= m - 1;
					startPos[q] = (u16)(t * 16 + j);
					vals[q] = v[j];
					++q;
				}
				if (t == 0) startPos[runs] = 4096;
				__syncthreads();
				for (u32 k = t; k < runs; k += WG) {
					dst[2 * k] = (u8)(startPos[k + 1] - startPos[k]);
					dst[2 * k + 1] = vals[k];
				}
			}
			bool same = true;
			if (s == 0) {
				const int firstSample = (int)(i8)rowsLds[b][0];
#pragma unroll
				for (int j = 0; j < 16; ++j) same = same && (firstSample * (int)(i8)v[j] > 0);
			}
			const int allSame = __syncthreads_and(same ? 1 : 0);
			if (t == 0) {
				if (raw) blockFlags[b] |= 2u << s;
				if (s == 0 && allSame && !raw) blockFlags[b] |= 1u;
				recOff[b] += sz;
			}
		}
	}
	__syncthreads();
	if (t < 8 && t < count) {
		// (pass 2 takes BF_Empty of a stream it did not read from pass 1's word)
		if (!blob) meta[(firstId + t) * 4u + 3u] = blockFlags[t] | (hint[t] & ~0xFu);
		else {
			const u32 fl = blockFlags[t] | (((hint[t] >> 4) & 1u) ? (hint[t] & 1u) : 0u);
			u8* rec = blob + where[firstId + t];
			rec[0] = (u8)fl; rec[1] = 0; rec[2] = 0; rec[3] = 0;
		}
	}
}

// k_heightmap: Grid::Create(w, heightmap) — one 16-byte store per lane (16 voxels of a row share y and z)
__global__ __launch_bounds__(WG) void k_heightmap(GridView g, const i8* map)
{
	const u32 n = (u32)g.n, segs = n >> 4;
	const size_t q = (size_t)blockIdx.x * WG + threadIdx.x; // 16-voxel segment index: x fastest
	if (q >= (size_t)n * n * segs) return;
	const u32 sx = (u32)(q % segs), y = (u32)((q / segs) % n), z = (u32)(q / ((size_t)segs * n));
	i8 h[16], v[16];
	memcpy(h, map + (size_t)y * n + sx * 16, 16);
#pragma unroll
	for (int i = 0; i < 16; ++i) v[i] = heightmap_distance((int)z, (int)h[i]);
	uint4 out;
	memcpy(&out, v, 16);
	*(uint4*)(const_cast<i8*>(g.dist) + ((size_t)z * n + y) * n + sx * 16) = out;
}

// k_terrain_height / k_terrain_fill: the benchmark's synthetic terrain (include/voxels_synth.h, vx_terrain_math.h)
// evaluated where the grid lives — the VoxelSurface -> Grid step (src/VoxelGrid.cpp:79-132, :37-50) for this one surface.
// Heights once per voxel column; then one lane per 16-voxel row segment, 16-byte stores.
struct TerrainRange { int z0, z1, y0, y1; }; // resident layers of a field inside the grid: planes [z0, z1), rows [y0, y1)

__global__ __launch_bounds__(WG) void k_terrain_height(u32 n, u32 seed, float* height)
{
	const u32 i = blockIdx.x * WG + threadIdx.x;
	if (i < n * n) height[i] = vxt::height(n, i % n, i / n, seed);
}

__global__ __launch_bounds__(WG) void k_terrain_fill(GridView g, u32 seed, const float* height, TerrainRange dr, TerrainRange mr, u32 style)
{
	const u32 n = (u32)g.n, segs = n >> 4;
	const u32 rows = (u32)(dr.y1 - dr.y0);
	const size_t q = (size_t)blockIdx.x * WG + threadIdx.x;
	if (q >= (size_t)segs * rows * (u32)(dr.z1 - dr.z0)) return;
	const u32 sx = (u32)(q % segs), y = (u32)dr.y0 + (u32)((q / segs) % rows), z = (u32)dr.z0 + (u32)(q / ((size_t)segs * rows));
	i8 d[16]; u8 m[16], b[16];
#pragma unroll 4
	for (u32 i = 0; i < 16; ++i) vxt::voxel(sx * 16 + i, y, z, height[(size_t)y * n + sx * 16 + i], seed, d[i], m[i], b[i], style);
	uint4 v;
	memcpy(&v, d, 16);
	*(uint4*)(const_cast<i8*>(g.dist) + dist_offset(g, (int)(sx * 16), (int)y, (int)z)) = v;
	if ((int)z >= mr.z0 && (int)z < mr.z1 && (int)y >= mr.y0 && (int)y < mr.y1) {
		const size_t o = mat_offset(g, (int)(sx * 16), (int)y, (int)z);
		memcpy(&v, m, 16); *(uint4*)(const_cast<u8*>(g.mat) + o) = v;
		memcpy(&v, b, 16); *(uint4*)(const_cast<u8*>(g.blend) + o) = v;
	}
}

// k_scatter_blocks: edited 16^3 blocks (4096 contiguous bytes each) into the dense fields; lane t owns voxel row t
__global__ __launch_bounds__(WG) void k_scatter_blocks(const u32* ids, u32 n, const u8* sd, const u8* sm, const u8* sb, u8* dist, u8* mat, u8* blend)
{
	const u32 nb = n >> 4, id = ids[blockIdx.x], t = threadIdx.x;
	const u32 bx = id % nb, by = (id / nb) % nb, bz = id / (nb * nb);
	const size_t rowOff = ((size_t)(bz * 16 + (t >> 4)) * n + by * 16 + (t & 15)) * n + bx * 16;
	const size_t srcOff = (size_t)blockIdx.x * 4096 + t * 16;
	if (sd) *(uint4*)(dist + rowOff) = *(const uint4*)(sd + srcOff);
	if (sm) *(uint4*)(mat + rowOff) = *(const uint4*)(sm + srcOff);
	if (sb) *(uint4*)(blend + rowOff) = *(const uint4*)(sb + srcOff);
}

// ---- lattice copies of the distance field for the coarser levels (PyramidLevel, tv_block.h) -----------------------------
// One 16-voxel segment [xs, xs + 16) of the voxel row (y,z): its samples on the level-L lattice go to the level's copy.  y and z may be the first row beyond the grid (y == n: the loaded data then is the
// clamped row n - 1, and y >> L is exactly the index of the level's clamped far entry); the voxel n - 1 of a row is also
// stored as the far entry of its lattice row.
template <int FIRST = 1, bool FAR_X = true>
__device__ __forceinline__ void pyramid_write_segment(const PyramidLevel* pyr, int n, int xs, int y, int z, uint4 d)
{
#pragma unroll
	for (int l = FIRST; l < 4; ++l) {
		const PyramidLevel& P = pyr[l];
		if (!P.data || ((y | z) & ((1 << l) - 1))) continue;
		i8* at = P.data + pyramid_offset(P, xs >> l, y >> l, z >> l); // 8 / 4 / 2 lattice samples: inside one brick row
		if (l == 1) {
			uint2 v;
			v.x = __builtin_amdgcn_perm(d.y, d.x, 0x06040200u);
			v.y = __builtin_amdgcn_perm(d.w, d.z, 0x06040200u);
			*(uint2*)at = v;
		} else if (l == 2) {
			*(u32*)at = __builtin_amdgcn_perm(d.y, d.x, 0x0C0C0400u) | __builtin_amdgcn_perm(d.w, d.z, 0x04000C0Cu);
		} else {
			*(u16*)at = (u16)((d.x & 0xFFu) | ((d.z & 0xFFu) << 8));
		}
		if (FAR_X && xs + 16 == n) P.data[pyramid_offset(P, n >> l, y >> l, z >> l)] = (i8)(d.w >> 24);
	}
	// levels >= 4: at most one lattice sample per segment (its first voxel), on one row in 256 - a loop that is not unrolled
	// (k_rebrick is a copy kernel: the registers of three more unrolled levels cost it a wave per SIMD)
	if ((y | z) & 15) return;
#pragma unroll 1
	for (int l = 4; l < PYRAMID_LEVELS; ++l) {
		const PyramidLevel& P = pyr[l];
		if (!P.data || ((y | z) & ((1 << l) - 1))) break;
		if (!(xs & ((1 << l) - 1))) P.data[pyramid_offset(P, xs >> l, y >> l, z >> l)] = (i8)(d.x & 0xFFu);
		if (FAR_X && xs + 16 == n) P.data[pyramid_offset(P, n >> l, y >> l, z >> l)] = (i8)(d.w >> 24);
	}
}

// What the mirrors keep beside the bricks (all of it a function of the grid alone, brought up to date with them):
//   * the lattice copies of levels 1..3 (PyramidLevel): every distance row with even y and z of the rank's range
//     [yBegin, yEnd] x [zBegin, zEnd] carries lattice samples; the grid's last row / plane (n - 1) also is the clamped far
//     sample n of every level;
//   * blockSign[id]: 1 = every distance sample of the block is >= 0, 2 = every sample is < 0, 0 = mixed or not all resident.

// The same segment for the x-plane copies (XPlanes, tv_block.h): its first voxel if the segment starts on a plane
// X = 32 k of a lattice, its last voxel (n - 1) as the clamped far plane.  y / z may be n like above.
template <int FIRST = 0, bool FAR_X = true>
__device__ __forceinline__ void xplane_write_segment(const XPlanes* xp, int n, int xs, int y, int z, uint4 d)
{
#pragma unroll
	for (int l = FIRST; l < XPLANE_LEVELS; ++l) {
		const XPlanes& P = xp[l];
		if (!P.data || ((y | z) & ((1 << l) - 1))) continue;
		if (!(xs & ((32 << l) - 1))) P.data[xplane_offset(P, (u32)(xs >> l) >> 5, (u32)(z >> l), (u32)(y >> l))] = (i8)(d.x & 0xFFu);
		if (FAR_X && xs + 16 == n) P.data[xplane_offset(P, (u32)(n >> l) >> 5, (u32)(z >> l), (u32)(y >> l))] = (i8)(d.w >> 24);
	}
}

// the clamped far entries along x of every lattice copy and x-plane copy that row (y, z) belongs to: the row's last voxel (n - 1)
__device__ __forceinline__ void lattice_far_x(const MirrorState& X, int n, int y, int z, u32 last)
{
#pragma unroll 1
	for (int l = 0; l < PYRAMID_LEVELS; ++l) {
		if ((y | z) & ((1 << l) - 1)) break;
		if (l >= 1 && X.pyr[l].data) X.pyr[l].data[pyramid_offset(X.pyr[l], n >> l, y >> l, z >> l)] = (i8)last;
		if (l < XPLANE_LEVELS && X.xp[l].data) X.xp[l].data[xplane_offset(X.xp[l], (u32)(n >> l) >> 5, (u32)(z >> l), (u32)(y >> l))] = (i8)last;
	}
}

__device__ __forceinline__ void lattice_rows_of(const MirrorState& X, int n, int xs, int y, int z, uint4 d)
{
	if (y < X.yBegin || y > X.yEnd || z < X.zBegin || z > X.zEnd) return;
	const bool yFar = y == n - 1, zFar = z == n - 1;
	if (X.xp[0].data || X.xp[1].data || X.xp[2].data) {
		xplane_write_segment(X.xp, n, xs, y, z, d);
		if (yFar) xplane_write_segment(X.xp, n, xs, n, z, d);
		if (zFar) xplane_write_segment(X.xp, n, xs, y, n, d);
		if (yFar && zFar) xplane_write_segment(X.xp, n, xs, n, n, d);
	}
	if (!X.pyr[1].data) return; // (the copies of the coarser levels exist where this one does)
	if (!((y | z) & 1)) pyramid_write_segment(X.pyr, n, xs, y, z, d);
	if (yFar && !(z & 1)) pyramid_write_segment(X.pyr, n, xs, n, z, d);
	if (zFar && !(y & 1)) pyramid_write_segment(X.pyr, n, xs, y, n, d);
	if (yFar && zFar) pyramid_write_segment(X.pyr, n, xs, n, n, d);
}

// k_rebrick: dense fields -> mirrors (tv_core.h GridView).  Box mode (ids == nullptr): a workgroup copies the 8
// x-neighbour blocks that share the 128-byte lines of their voxel rows, 8 consecutive lanes per line read and one whole
// 128-byte tile per brick written by a wave's store; list mode: one block per workgroup, one voxel row per lane.  Only rows that are resident in the dense fields are copied (a slab's halo
// block layers hold a few planes / rows each).
struct RebrickRanges { int dz0, dz1, dy0, dy1, mz0, mz1, my0, my1; };

// blockSign fields (2 bits each, field f = dx | dy << 1 | dz << 2): the signs found in the part of the block that the
// cells of its -x / -y / -z neighbours reach into — all voxels (f = 0), the plane x = 0 (f = 1), y = 0 (f = 2), the line
// x = y = 0 (f = 3), the plane z = 0 (f = 4), ... the voxel (0,0,0) (f = 7).  While collecting: bit 0 = a sample >= 0 was
// seen, bit 1 = a sample < 0; a row that is not resident sets every bit (unknown).
__device__ __forceinline__ u32 rebrick_row(const GridView& g, const RebrickRanges& r, const MirrorState& X, int bx, int gy, int gz)
{
	const size_t dst = brick_base(g, bx, gy >> 4, gz >> 4) + brick_local(0u, (u32)gy & 15u, (u32)gz & 15u);
	const bool y0 = (gy & 15) == 0, z0 = (gz & 15) == 0;
	// (a row that is not resident makes the fields it belongs to unknown - those alone: of a slab's halo block layer the
	// plane y = 0 is resident, and that plane is all the cells of the last owned block layer read of it)
	u32 fields = 0xFu | (y0 ? 0xF0u : 0u) | (z0 ? 0xF00u : 0u) | (y0 && z0 ? 0xF000u : 0u);
	if (gz >= r.dz0 && gz < r.dz1 && gy >= r.dy0 && gy < r.dy1) {
		const uint4 d = *(const uint4*)(g.dist + dist_offset(g, bx * 16, gy, gz));
		*(uint4*)(const_cast<i8*>(g.bDist) + dst) = d;
		lattice_rows_of(X, g.n, bx * 16, gy, gz, d);
		const u32 any = (d.x | d.y | d.z | d.w) & 0x80808080u, all = (d.x & d.y & d.z & d.w) & 0x80808080u;
		const u32 rowAll = (any ? 2u : 0u) | (all != 0x80808080u ? 1u : 0u);
		const u32 rowX0 = (d.x & 0x80u) ? 2u : 1u;
		const u32 pair = rowAll | (rowX0 << 2);                   // fields 0 and 1 of this row
		fields = pair | (y0 ? pair << 4 : 0u) | (z0 ? pair << 8 : 0u) | (y0 && z0 ? pair << 12 : 0u);
	}
	if (gz >= r.mz0 && gz < r.mz1 && gy >= r.my0 && gy < r.my1) {
		const size_t src = mat_offset(g, bx * 16, gy, gz);
		*(uint4*)(const_cast<u8*>(g.bMat) + dst) = *(const uint4*)(g.mat + src);
		*(uint4*)(const_cast<u8*>(g.bBlend) + dst) = *(const uint4*)(g.blend + src);
	}
	return fields;
}

// collected field bits -> blockSign: per field 1 = all >= 0, 2 = all < 0, 0 = mixed or unknown
__device__ __forceinline__ u16 block_sign_word(u32 collected)
{
	u32 w = 0;
#pragma unroll
	for (int f = 0; f < 8; ++f) {
		const u32 b = (collected >> (2 * f)) & 3u;
		w |= ((b == 1u || b == 2u) ? b : 0u) << (2 * f);
	}
	return (u16)w;
}

__global__ __launch_bounds__(WG) __attribute__((amdgpu_waves_per_eu(4))) void k_rebrick(GridView g, RebrickRanges r, MirrorState X, int yb0, int ybCount, int zb0, const u32* ids)
{
	__shared__ u32 blockSigns[8];
	// what the workgroup's rows contribute to the finest lattice copies and x-plane copies, collected here and written in whole
	// 16-byte rows behind the copy (as stores of their own they were 8-, 4- and 1-byte pieces by a quarter of the lanes: more
	// store instructions than the copy itself, 8 % of its time)
	__shared__ __attribute__((aligned(16))) struct {
		u8 p1[8][8][64];  // level-1 lattice: [Z1][Y1][X1] of the 64 x 8 x 8 samples under the 8 blocks
		u8 p2[4][4][32];  // level-2 lattice
		u8 x0[4][16][16]; // level-0 x planes X = 32 k: [k][z][y]
		u8 x1[2][8][8];   // level-1 x planes
		u8 p3[2][2][16];  // level-3 lattice
		u8 x2[4][4];      // level-2 x plane (one per 8 blocks)
	} lat;
	const int nb = g.n >> 4, tid = (int)threadIdx.x;
	if (tid < 8) blockSigns[tid] = 0;
	__syncthreads();
	if (ids) {
		const u32 id = ids[blockIdx.x];
		const int bx = (int)(id % (u32)nb), by = (int)((id / (u32)nb) % (u32)nb), bz = (int)(id / (u32)(nb * nb));
		atomicOr(&blockSigns[0], rebrick_row(g, r, X, bx, by * 16 + (tid & 15), bz * 16 + (tid >> 4)));
		__syncthreads();
		if (tid == 0 && X.blockSign) X.blockSign[id] = block_sign_word(blockSigns[0]);
		return;
	}
	const int groups = (nb + 7) >> 3;
	const int gx = (int)blockIdx.x % groups, by = yb0 + ((int)blockIdx.x / groups) % ybCount, bz = zb0 + (int)blockIdx.x / (groups * ybCount);
	const int bxl = tid & 7, bx = gx * 8 + bxl;
	u32 signs = 0;
	// A wave takes one 128-byte tile of each of its 8 bricks per trip: rows y = 4 w .. 4 w + 3 of the planes z = 2 it, 2 it + 1
	// (brick_local: the tile's 8 pieces in lane order), so every store instruction writes 8 whole lines.
	const int sub = tid >> 3, ry = ((sub >> 3) << 2) | (sub & 3), rz = (sub >> 2) & 1;
	const int gy = by * 16 + ry, gz0 = bz * 16 + rz;
	// (uniform) the rule: all 8 blocks exist, every row of theirs is resident in all three dense fields and owned by this rank, and
	// none is the grid's last row or plane (whose samples are also the lattices' clamped far entries) - a slab's halo block
	// layers, the grid's far block layers and grids narrower than 8 blocks take the general row by row form below
	const bool whole = gx * 8 + 8 <= nb
	                && by * 16 >= r.dy0 && by * 16 + 16 <= r.dy1 && bz * 16 >= r.dz0 && bz * 16 + 16 <= r.dz1
	                && by * 16 >= r.my0 && by * 16 + 16 <= r.my1 && bz * 16 >= r.mz0 && bz * 16 + 16 <= r.mz1
	                && by * 16 >= X.yBegin && by * 16 + 15 <= X.yEnd && bz * 16 >= X.zBegin && bz * 16 + 15 <= X.zEnd
	                && by + 1 < nb && bz + 1 < nb;
	if (whole) {
		// all requests of a field are in flight before the first piece is stored (a load behind a range test is waited for on the spot)
		const size_t dst = brick_base(g, bx, by, bz) + brick_local(0u, (u32)ry, (u32)rz);
		const size_t srcD = dist_offset(g, bx * 16, gy, gz0), stepD = (size_t)2 * g.pitchY * g.n;
		const size_t srcM = mat_offset(g, bx * 16, gy, gz0), stepM = (size_t)2 * g.pitchYMat * g.n;
		uint4 d[8];
#pragma unroll
		for (int it = 0; it < 8; ++it) d[it] = *(const uint4*)(g.mat + srcM + it * stepM);
#pragma unroll
		for (int it = 0; it < 8; ++it) *(uint4*)(const_cast<u8*>(g.bMat) + dst + (it << 9)) = d[it];
#pragma unroll
		for (int it = 0; it < 8; ++it) d[it] = *(const uint4*)(g.blend + srcM + it * stepM);
#pragma unroll
		for (int it = 0; it < 8; ++it) *(uint4*)(const_cast<u8*>(g.bBlend) + dst + (it << 9)) = d[it];
#pragma unroll
		for (int it = 0; it < 8; ++it) d[it] = *(const uint4*)(g.dist + srcD + it * stepD);
		const bool y0 = ry == 0;
		const bool planes = X.xp[0].data || X.xp[1].data || X.xp[2].data, lattices = X.pyr[1].data != nullptr;
		const bool even = !((ry | rz) & 1), fourth = !((ry & 3) | rz);   // this lane's rows with even it / it a multiple of 2 lie on the level-1 / level-2 lattice
#pragma unroll
		for (int it = 0; it < 8; ++it) {
			*(uint4*)(const_cast<i8*>(g.bDist) + dst + (it << 9)) = d[it];
			const u32 any = (d[it].x | d[it].y | d[it].z | d[it].w) & 0x80808080u, all = (d[it].x & d[it].y & d[it].z & d[it].w) & 0x80808080u;
			const u32 pair = (any ? 2u : 0u) | (all != 0x80808080u ? 1u : 0u) | ((d[it].x & 0x80u) ? 8u : 4u); // fields 0 and 1 of this row
			const bool z0 = it == 0 && rz == 0;
			signs |= pair | (y0 ? pair << 4 : 0u) | (z0 ? pair << 8 : 0u) | (y0 && z0 ? pair << 12 : 0u);
			// the lattice samples of the row (one row in four carries any; no far entries along y and z in here)
			if (planes) {
				if (!(bxl & 1)) lat.x0[bxl >> 1][2 * it + rz][ry] = (u8)d[it].x;
				if (!(bxl & 3) && even) lat.x1[bxl >> 2][it][ry >> 1] = (u8)d[it].x;
				if (!bxl && fourth && !(it & 1)) lat.x2[it >> 1][ry >> 2] = (u8)d[it].x;
			}
			if (lattices && even) {
				*(uint2*)&lat.p1[it][ry >> 1][bxl * 8] = make_uint2(__builtin_amdgcn_perm(d[it].y, d[it].x, 0x06040200u), __builtin_amdgcn_perm(d[it].w, d[it].z, 0x06040200u));
				if (fourth && !(it & 1)) *(u32*)&lat.p2[it >> 1][ry >> 2][bxl * 4] = __builtin_amdgcn_perm(d[it].y, d[it].x, 0x0C0C0400u) | __builtin_amdgcn_perm(d[it].w, d[it].z, 0x04000C0Cu);
				if (!(ry & 7) && !(it & 3)) *(u16*)&lat.p3[it >> 2][ry >> 3][bxl * 2] = (u16)((d[it].x & 0xFFu) | ((d[it].z & 0xFFu) << 8));
			}
			if (bx + 1 == nb) lattice_far_x(X, g.n, gy, gz0 + 2 * it, d[it].w >> 24);
		}
		if (lattices && !(ry | rz)) pyramid_write_segment<4, false>(X.pyr, g.n, bx * 16, gy, gz0, d[0]); // (levels >= 4: the block's first row)
		atomicOr(&blockSigns[bxl], signs);
		__syncthreads();
		if (lattices) {
			// level 1: 8 x 8 rows of 64 samples = 256 pieces of 16 bytes (a lattice row inside its brick), lanes ordered so that 8
			// consecutive ones write one 128-byte tile when the rank's lattice origin is tile-aligned
			const int seg = (tid >> 3) & 3, y1 = (tid & 3) | ((tid >> 5) & 1) << 2, z1 = ((tid >> 2) & 1) | (tid >> 6) << 1;
			*(uint4*)(X.pyr[1].data + pyramid_offset(X.pyr[1], (gx * 4 + seg) * 16, by * 8 + y1, bz * 8 + z1)) = *(const uint4*)&lat.p1[z1][y1][seg * 16];
			if (tid < 32 && X.pyr[2].data) {
				const int s2 = tid & 1, y2 = (tid >> 1) & 3, z2 = tid >> 3;
				*(uint4*)(X.pyr[2].data + pyramid_offset(X.pyr[2], (gx * 2 + s2) * 16, by * 4 + y2, bz * 4 + z2)) = *(const uint4*)&lat.p2[z2][y2][s2 * 16];
			}
		}
		if (tid >= 64 && tid < 128 && X.xp[0].data) {
			const int k = (tid >> 4) & 3, z = tid & 15;
			*(uint4*)(X.xp[0].data + xplane_offset(X.xp[0], (u32)(gx * 4 + k), (u32)(bz * 16 + z), (u32)(by * 16))) = *(const uint4*)&lat.x0[k][z][0];
		}
		if (tid >= 128 && tid < 144 && X.xp[1].data) {
			const int k = (tid >> 3) & 1, z = tid & 7;
			*(uint2*)(X.xp[1].data + xplane_offset(X.xp[1], (u32)(gx * 2 + k), (u32)(bz * 8 + z), (u32)(by * 8))) = *(const uint2*)&lat.x1[k][z][0];
		}
		if (tid >= 192 && tid < 196) {
			const int a = tid & 1, b = (tid >> 1) & 1;
			if (lattices && X.pyr[3].data) *(uint4*)(X.pyr[3].data + pyramid_offset(X.pyr[3], gx * 16, by * 2 + a, bz * 2 + b)) = *(const uint4*)&lat.p3[b][a][0];
			if (X.xp[2].data) *(u32*)(X.xp[2].data + xplane_offset(X.xp[2], (u32)gx, (u32)(bz * 4 + (tid & 3)), (u32)(by * 4))) = *(const u32*)&lat.x2[tid & 3][0];
		}
	} else {
		if (bx < nb) {
#pragma unroll 1
			for (int it = 0; it < 8; ++it) signs |= rebrick_row(g, r, X, bx, gy, gz0 + 2 * it);
			atomicOr(&blockSigns[bxl], signs);
		}
		__syncthreads();
	}
	if (tid < 8 && gx * 8 + tid < nb && X.blockSign)
		X.blockSign[block_coord_id((u32)(gx * 8 + tid), (u32)by, (u32)bz, (u32)nb)] = block_sign_word(blockSigns[tid]);
}

// ---- start of a full run: header = 0, block -> slot maps = -1 (one launch instead of a memset per array) -------------
struct ResetRanges {
	u32* header;
	u32 headerWords;
	u32* partials;            // [workgroups of k_run_head]: block-class statistics (behind the header, copied with it)
	u32 partialCount;
	u32* listCounts;          // [listWgs] listed blocks per LIST_WG block coordinates (LevelDesc::listCounts): zeroed here
	u32 listWgs;
	u32 start[MAX_LEVELS + 1]; // word ranges of the flat index space: [0, headerWords) header, then slotOf of level 0, 1, ...
	int* maps[MAX_LEVELS];     // the block -> slot maps (of the run's own set, or - k_tail - of the set the next run will use)
};

__device__ __forceinline__ void reset_words(const ResetRanges& r, u32 lane)
{
	const u32 total = r.start[MAX_LEVELS];
	// four consecutive words per lane; ranges are multiples of 4 words except possibly tiny coarse levels
	const u32 first = lane * 4;
#pragma unroll
	for (u32 k = 0; k < 4; ++k) {
		const u32 i = first + k;
		if (i >= total) return;
		if (i < r.headerWords) { r.header[i] = 0; continue; }
		u32 l = 0;
#pragma unroll
		for (u32 q = 1; q < MAX_LEVELS; ++q) if (i >= r.start[q]) l = q;
		int* map = r.maps[0];
#pragma unroll
		for (u32 q = 1; q < MAX_LEVELS; ++q) if (l == q) map = r.maps[q];
		map[i - r.start[l]] = -1;
	}
}

// ---- head of a full run, one launch: the run's counters = 0 and block -> slot maps = -1 (four words per lane), and per
//      level-0 block what the emptiness flags and the sign summaries already say about it (one block per lane).  The two
//      halves touch different memory, so one kernel can do both; the statistics of the classes leave as partial sums. ---------
// (the reset alone: in front of a k_run_head that hands out slots, which needs the counters and the maps of the levels >= 1
// in their start state before its first atomic)
__global__ __launch_bounds__(WG) void k_reset(ExecParamsDev p, ResetRanges r)
{
	const u32 i = blockIdx.x * WG + threadIdx.x;
	reset_words(r, i);
	if (i < r.listWgs) r.listCounts[i] = 0;
}

// allocate: the blocks that are not quiet get their level-0 slots here, and their ancestors the slots of the levels above
// (what k_classify + k_hierarchy do for a run whose classification is a pass of its own).  Not quiet means: the 17^3 samples
// the block's cells read are not of one sign, so at least one cell is non-trivial - exactly the blocks k_classify finds
// active.  Their bitmaps are then the business of whoever polygonizes them (k_main: f0_walk<.., SELF>, mat_block).
__global__ __launch_bounds__(WG) void k_run_head(ExecParamsDev p, ResetRanges r, u32 allocate)
{
	const LevelDesc& L = p.levels[0];
	const u32 i = blockIdx.x * WG + threadIdx.x;
	if (r.header) reset_words(r, i);
	if (r.header && i < r.listWgs) r.listCounts[i] = 0;
	const u32 rowsY = L.yb1 - L.yb0;
	bool inRange = i < L.cnt * rowsY * (L.zb1 - L.zb0);
	const u32 ii = inRange ? i : 0u;
	u32 bx = ii % L.cnt, by = L.yb0 + (ii / L.cnt) % rowsY, bz = L.zb0 + ii / (L.cnt * rowsY);
	// allocate: a workgroup is an 8 x 8 x 4 box of blocks at an aligned place (lane = x | y << 3 | z << 6), so that the ancestors
	// of its blocks on the levels 1 and 2 are its own business and the ancestor on level 3 that of two workgroups
	u32 boxX = 0, boxY = 0, boxZ = 0;
	if (allocate) {
		const u32 nx = (L.cnt + 7u) >> 3, y0 = L.yb0 >> 3, ny = ((L.yb1 + 7u) >> 3) - y0, z0 = L.zb0 >> 2;
		boxX = blockIdx.x % nx; boxY = y0 + (blockIdx.x / nx) % ny; boxZ = z0 + blockIdx.x / (nx * ny);
		const u32 x = boxX * 8u + (threadIdx.x & 7u), y = boxY * 8u + ((threadIdx.x >> 3) & 7u), z = boxZ * 4u + (threadIdx.x >> 6);
		inRange = x < L.cnt && y >= L.yb0 && y < L.yb1 && z >= L.zb0 && z < L.zb1;
		bx = min(x, L.cnt - 1u); by = min(y, L.cnt - 1u); bz = min(z, L.cnt - 1u); // (lanes outside the range read a block that exists)
	}
	const u32 id = block_coord_id(bx, by, bz, L.cnt);
	u32 all = 1u; // AND over the 27 BF_Empty flags (neighbour coordinates clamped like the reference's)
#pragma unroll
	for (int k = 0; k < 27; ++k) {
		const u32 cx = (u32)clampi((int)bx + (k % 3) - 1, 0, (int)L.cnt - 1);
		const u32 cy = (u32)clampi((int)by + ((k / 3) % 3) - 1, 0, (int)L.cnt - 1);
		const u32 cz = (u32)clampi((int)bz + (k / 9) - 1, 0, (int)L.cnt - 1);
		const u32 s = p.G.emptyFlags[block_coord_id(cx, cy, cz, L.cnt)] ? 1u : 0u; // BF_Empty of the neighbour (the flag array covers the whole grid)
		all &= s;
	}
	u32 c = (all & 1u) ? (u32)BC_SKIPPED : 0u;
	// quiet: the block itself and the parts of its +x / +y / +z neighbours that its cells reach into (their first plane,
	// line or voxel: the fields of blockSign) are of one sign, so none of its cells can be non-trivial.  At the grid's far
	// side the samples are clamped, i.e. the block's own.
	u32 signAll = 3u, signAny = 0u;
#pragma unroll
	for (u32 k = 0; k < 8; ++k) {
		u32 f = k;
		u32 cx = bx + (k & 1u), cy = by + ((k >> 1) & 1u), cz = bz + (k >> 2);
		if (cx >= L.cnt) { cx = bx; f &= ~1u; }
		if (cy >= L.cnt) { cy = by; f &= ~2u; }
		if (cz >= L.cnt) { cz = bz; f &= ~4u; }
		const u32 sg = ((u32)p.G.blockSign[block_coord_id(cx, cy, cz, L.cnt)] >> (2u * f)) & 3u;
		signAll &= sg; signAny |= sg;
	}
	if (signAll == signAny && signAll != 0u) c |= BC_QUIET | (signAll == 2u ? (u32)BC_NEGATIVE : 0u);
	if (inRange) p.G.blockClass[id] = (u8)c;
	if (allocate) {
		// Level 0: ballot-compacted slots, one reservation per workgroup.  The ancestors on the levels 1 and 2: which of the box's
		// 4 x 4 x 2 / 2 x 2 x 1 ancestors have an active block below them is a mask in LDS; their slots are reserved with one
		// atomic per level and workgroup and handed out by rank - no claim is needed, nobody else has blocks below them.
		// Levels >= 3: one lane per workgroup claims (compare-and-swap, all levels at once) - two workgroups share an
		// ancestor on level 3, 16 on level 4.  All reservations of a workgroup travel together: one round trip.
		// (One compare-and-swap per active block and level, as k_classify does it, was 18 of this kernel's 27 us on the 1024^3
		// bench terrain.)
		__shared__ u32 waveActive[WG / 64];
		__shared__ u32 below[3];      // [1], [2]: bit per ancestor of the box on that level; [0]: any active block
		__shared__ u32 firstSlot[3];  // first slot of the box's blocks on the levels 0, 1, 2
		if (threadIdx.x < 3) below[threadIdx.x] = 0;
		__syncthreads();
		const bool active = inRange && !(c & BC_QUIET);
		const unsigned long long m = __ballot(active);
		const u32 lane = threadIdx.x & 63u, wave = threadIdx.x >> 6;
		if (lane == 0) waveActive[wave] = (u32)__popcll(m);
		if (active) {
			const u32 lx = threadIdx.x & 7u, ly = (threadIdx.x >> 3) & 7u, lz = threadIdx.x >> 6;
			atomicOr(&below[1], 1u << ((lx >> 1) | ((ly >> 1) << 2) | ((lz >> 1) << 4)));
			atomicOr(&below[2], 1u << ((lx >> 2) | ((ly >> 2) << 1)));
			below[0] = 1u;
		}
		__syncthreads();
		const u32 levelsRun = p.G.levels;
		if (threadIdx.x == 0) {
			u32 tot = 0;
			for (u32 w = 0; w < (u32)(WG / 64); ++w) tot += waveActive[w];
			firstSlot[0] = tot ? atomicAdd(L.nActive, tot) : 0u;
		}
		if ((threadIdx.x == 1u || threadIdx.x == 2u) && threadIdx.x < levelsRun) {
			const u32 cnt = (u32)__popc(below[threadIdx.x]);
			firstSlot[threadIdx.x] = cnt ? atomicAdd(p.levels[threadIdx.x].nActive, cnt) : 0u;
		}
		if (threadIdx.x == 64u && below[0]) {
			// the box's one ancestor per level >= 3
			int won[MAX_LEVELS];
#pragma unroll
			for (u32 l = 3; l < (u32)MAX_LEVELS; ++l) {
				won[l] = 0;
				if (l < levelsRun) {
					const LevelDesc& A = p.levels[l];
					const u32 px = (boxX * 8u) >> l, py = (boxY * 8u) >> l, pz = (boxZ * 4u) >> l;
					if (px < A.cnt && py < A.cnt && pz < A.cnt) won[l] = atomicCAS(&A.slotOf[block_coord_id(px, py, pz, A.cnt)], -1, -2) == -1 ? 1 : 0;
				}
			}
			u32 aslot[MAX_LEVELS];
#pragma unroll
			for (u32 l = 3; l < (u32)MAX_LEVELS; ++l) if (won[l]) aslot[l] = atomicAdd(p.levels[l].nActive, 1u);
#pragma unroll
			for (u32 l = 3; l < (u32)MAX_LEVELS; ++l) {
				if (!won[l]) continue;
				const LevelDesc& A = p.levels[l];
				const u32 aid = block_coord_id((boxX * 8u) >> l, (boxY * 8u) >> l, (boxZ * 4u) >> l, A.cnt);
				A.slotCoord[aslot[l]] = aid;
				A.slotOf[aid] = (int)aslot[l]; // visible to the next kernel
			}
		}
		__syncthreads();
		int slotOrNone = -1;
		if (active) {
			u32 slot = firstSlot[0] + (u32)__popcll(m & ((1ull << lane) - 1ull));
			for (u32 w = 0; w < wave; ++w) slot += waveActive[w];
			slotOrNone = (int)slot;
			L.slotCoord[slot] = id;
			L.skip[slot] = (c & BC_SKIPPED) ? 1 : 0;
			L.ntCount[slot] = 1; // (not known yet: whoever walks the slot counts its cells; 0 would mean "no geometry")
		}
		if (threadIdx.x < 32u && 1u < levelsRun && ((below[1] >> threadIdx.x) & 1u)) {
			const LevelDesc& A = p.levels[1];
			const u32 j = threadIdx.x, aslot = firstSlot[1] + (u32)__popc(below[1] & ((1u << j) - 1u));
			const u32 aid = block_coord_id(boxX * 4u + (j & 3u), boxY * 4u + ((j >> 2) & 3u), boxZ * 2u + (j >> 4), A.cnt);
			A.slotCoord[aslot] = aid;
			A.slotOf[aid] = (int)aslot;
		}
		if (threadIdx.x >= 32u && threadIdx.x < 36u && 2u < levelsRun && ((below[2] >> (threadIdx.x - 32u)) & 1u)) {
			const LevelDesc& A = p.levels[2];
			const u32 j = threadIdx.x - 32u, aslot = firstSlot[2] + (u32)__popc(below[2] & ((1u << j) - 1u));
			const u32 aid = block_coord_id(boxX * 2u + (j & 1u), boxY * 2u + (j >> 1), boxZ, A.cnt);
			A.slotCoord[aslot] = aid;
			A.slotOf[aid] = (int)aslot;
		}
		if (inRange) L.slotOf[id] = slotOrNone; // (level 0's map is written here for every block of the range: a run that finds its counters
		                                        // and the maps of the levels above reset - k_tail of the run before - needs no k_reset)
	}
	// Two statistics, as per-workgroup partial sums that travel with the header (no atomics: the header is being zeroed by
	// this very launch): blocks the classify pass will read - what "every distance sample once" amounts to for this grid
	// (reported, bench.py) - and the reference's "blocks calculated" on level 0: every block its emptiness rule does not
	// skip (:1511-1527), whether the classify pass has to read it or not.
	const int readers = __syncthreads_count(inRange && !(c & BC_QUIET));
	const int calculated = __syncthreads_count(inRange && !(c & BC_SKIPPED));
	if (threadIdx.x == 0 && allocate) {
		// (the counters were reset before this launch: the sums go straight into the header, which then is all the host reads)
		if (readers) atomicAdd(p.G.largeBlocks + 1, (u32)readers);
		if (calculated) atomicAdd(p.G.stats + 2, (u32)calculated);
	} else if (threadIdx.x == 0 && r.partials) r.partials[blockIdx.x] = (u32)readers | ((u32)calculated << 16);
}

constexpr int TB = 16;            // level-0 blocks per classify tile along x (256 voxels = two 128-byte lines per row)
constexpr int TW = TB / 2;        // 32-bit words of sign bits per tile row

__global__ __launch_bounds__(WG) void k_classify(ExecParamsDev p, u32 rowGroup, u32 activateAncestors)
{
	__shared__ __attribute__((aligned(16))) u16 sgn[289 * TB];  // sign masks: row r = rz*17+ry, TB x 16 voxels
	__shared__ u8 halo[292];                                     // sign of the voxel right of the tile, per row
	__shared__ __attribute__((aligned(16))) u16 blockBits[TB * 256];
	__shared__ u32 blockAny[TB];
	__shared__ u32 blockCnt[TB];
	__shared__ u32 blockSkipped[TB];
	__shared__ u32 blockCls[TB];
	__shared__ int blockSlot[TB];

	const LevelDesc& L = p.levels[0];
	const GridView& g = p.G.grid;
	const int n = g.n;
	const u32 tilesX = (L.cnt + TB - 1) / TB;
	// XCD-aware tile order: workgroup b runs on XCD b % 8 (observed dispatch order; used for speed only).  An XCD walks
	// whole block rows (all tiles of a row back to back), R consecutive rows at a time, so its L2 sees all address bits
	// (all of its channels) and consecutive rows share their halo lines in that L2.
	u32 tile = blockIdx.x;
	{
		if (rowGroup) {
			// groups of R block rows go round-robin over the XCDs: the surface usually sits in a narrow z band, and
			// tiles of quiet blocks cost nothing, so contiguous per-XCD ranges would leave most XCDs idle
			const u32 R = rowGroup;
			const u32 xcd = tile & 7u, m = tile >> 3;
			const u32 lr = m / tilesX;
			tile = (((lr / R) * 8u + xcd) * R + lr % R) * tilesX + m % tilesX;
		}
	}
	const u32 rowsY = L.yb1 - L.yb0;
	const u32 tx = tile % tilesX, by = L.yb0 + (tile / tilesX) % rowsY, bz = L.zb0 + tile / (tilesX * rowsY);
	const int x0 = (int)tx * 16 * TB;
	const int validCells = (n - x0) < 16 * TB ? (n - x0) : 16 * TB; // multiple of 16
	const int tid = threadIdx.x;

	// what the flags say (the classes of the tile's 16 blocks, written by k_run_head): quiet blocks are not read at all, a
	// tile of quiet blocks is done - most tiles of a terrain
	u32 myClass = BC_SKIPPED | BC_QUIET;
	if ((L.cnt & (TB - 1u)) == 0u) {
		// whole tiles: the 16 classes are 16 aligned bytes at a workgroup-uniform address (a scalar load)
		const uint4 cls = *(const uint4*)(p.G.blockClass + block_coord_id(tx * TB, by, bz, L.cnt));
		const u32 quietAll = cls.x & cls.y & cls.z & cls.w & 0x01010101u * (u32)BC_QUIET;
		if (quietAll == 0x01010101u * (u32)BC_QUIET) return;
		const u32 word = ((u32)tid & 12u) == 0u ? cls.x : (((u32)tid & 12u) == 4u ? cls.y : (((u32)tid & 12u) == 8u ? cls.z : cls.w));
		myClass = (word >> (((u32)tid & 3u) * 8u)) & 0xFFu;
	} else {
		const u32 bx = tx * TB + ((u32)tid & (TB - 1u));
		if (bx < L.cnt) myClass = p.G.blockClass[block_coord_id(bx, by, bz, L.cnt)];
		if (__ballot(!(myClass & BC_QUIET)) == 0ull) return; // (every wave holds the 16 classes four times over: uniform over the workgroup)
	}
#if defined(VX_CLS_PROFILE)
	unsigned long long clsTick = __builtin_readcyclecounter();
#define CLS_TICK(i) do { const unsigned long long now_ = __builtin_readcyclecounter(); if (tid == 0) atomicAdd(&p.G.largeBlocks[16 + (i)], (u32)((now_ - clsTick) >> 4)); clsTick = now_; } while (0)
#else
#define CLS_TICK(i) do { } while (0)
#endif
	if (tid < TB) { blockAny[tid] = 0; blockCnt[tid] = 0; blockSlot[tid] = -1; blockCls[tid] = myClass; }
	__syncthreads();
	CLS_TICK(0);

	// ---- load from the brick mirror: the TB blocks of a tile are 64 KB of consecutive addresses, lane t takes the 16-byte
	//      voxel row t (memory order) of every block, so a wave reads 1 KB at a stretch (the dense field would hand out the
	//      same bytes as 289 pieces of 256 bytes, 1 KB apart: one DRAM page per piece).  Then the rows y = 16 and z = 16
	//      from the neighbour bricks and the voxel right of the tile.  Several loads of a thread are in flight before the
	//      first sign mask is formed. --------------------------------------------------------------------------------------
	const auto keep = [&](int seg, int ry, int rz, uint4 d) {
		sgn[(rz * 17 + ry) * TB + seg] = (u16)(sign_nibble(d.x) | (sign_nibble(d.y) << 4) | (sign_nibble(d.z) << 8) | (sign_nibble(d.w) << 12));
	};
	const auto quiet_fill = [&](u32 cls) {
		return (cls & BC_NEGATIVE) ? make_uint4(0x80808080u, 0x80808080u, 0x80808080u, 0x80808080u) : make_uint4(0, 0, 0, 0);
	};
	{
		// Two rounds of loads: the rows of blocks 0..7; then the rows of blocks 8..15 together with the rows y = 16 / z = 16
		// (16 + 17 per block: 528 requests over 256 lanes) and the voxel right of the tile (289 bytes) — everything of a
		// round is in flight before its first sign mask is formed.
		const i8* tileBase = g.bDist + brick_base(g, (int)(tx * TB), (int)by, (int)bz);
		const int row = tid; // the lane's voxel row of every block, in memory order
		const int ry = ((row >> 3) & 3) * 4 + (row & 3), rz = (row >> 5) * 2 + ((row >> 2) & 1); // brick_local, inverted
		const auto main_row = [&](int seg) {
			const u32 cls = blockCls[seg];
			uint4 d = make_uint4(0, 0, 0, 0);
			if (cls & BC_QUIET) d = quiet_fill(cls);
			else if (seg * 16 < validCells) d = *(const uint4*)(tileBase + (size_t)seg * BRICK_BYTES + (size_t)row * 16);
			return d;
		};
		const auto far_where = [&](int q, int& seg, int& fy, int& fz) {
			if (q < 16 * TB) { seg = q >> 4; fy = 16; fz = q & 15; }
			else { const int e = q - 16 * TB; seg = e / 17; fy = e - seg * 17; fz = 16; }
		};
		const auto far_row = [&](int q) {
			int seg, fy, fz;
			far_where(min(q, 16 * TB + 17 * TB - 1), seg, fy, fz);
			const u32 cls = blockCls[seg];
			uint4 d = make_uint4(0, 0, 0, 0);
			if (cls & BC_QUIET) d = quiet_fill(cls);
			else if (seg * 16 < validCells) {
				const int y = clampi((int)by * 16 + fy, 0, n - 1);
				const int z = clampi((int)bz * 16 + fz, 0, n - 1);
				d = *(const uint4*)(g.bDist + brick_offset(g, x0 + seg * 16, y, z));
			}
			return d;
		};
		const auto right_voxel = [&](int r) {
			r = min(r, 288);
			const int hy = r % 17, hz = r / 17;
			const int y = clampi((int)by * 16 + hy, 0, n - 1);
			const int z = clampi((int)bz * 16 + hz, 0, n - 1);
			const int x = clampi(x0 + validCells, 0, n - 1);
			return g.bDist[brick_offset(g, x, y, z)];
		};
		uint4 m[8];
#pragma unroll
		for (int i = 0; i < 8; ++i) m[i] = main_row(i);
#pragma unroll
		for (int i = 0; i < 8; ++i) keep(i, ry, rz, m[i]);
		uint4 f[3];
		i8 xv[2];
#pragma unroll
		for (int i = 0; i < 8; ++i) m[i] = main_row(8 + i);
#pragma unroll
		for (int i = 0; i < 3; ++i) f[i] = far_row(tid + i * WG);
#pragma unroll
		for (int i = 0; i < 2; ++i) xv[i] = right_voxel(tid + i * WG);
#pragma unroll
		for (int i = 0; i < 8; ++i) keep(8 + i, ry, rz, m[i]);
#pragma unroll
		for (int i = 0; i < 3; ++i) {
			const int q = tid + i * WG;
			if (q < 16 * TB + 17 * TB) { int seg, fy, fz; far_where(q, seg, fy, fz); keep(seg, fy, fz, f[i]); }
		}
#pragma unroll
		for (int i = 0; i < 2; ++i) { const int r = tid + i * WG; if (r < 289) halo[r] = (u8)((u32)(xv[i] >> 7) & 1u); }
	}
	__syncthreads();
	CLS_TICK(1);

	// ---- classify the 16*TB cells of one (y,z) row per thread, bit-parallel -----------------------------
	{
		const int y = tid & 15, z = tid >> 4;
		const int r00 = z * 17 + y, r01 = r00 + 1, r10 = r00 + 17, r11 = r00 + 18;
		const u32* s00 = (const u32*)(sgn + r00 * TB);
		const u32* s01 = (const u32*)(sgn + r01 * TB);
		const u32* s10 = (const u32*)(sgn + r10 * TB);
		const u32* s11 = (const u32*)(sgn + r11 * TB);
		u32 A[TW + 1], O[TW + 1];
#pragma unroll
		for (int i = 0; i < TW; ++i) {
			const u32 a = s00[i], b = s01[i], c = s10[i], d = s11[i];
			A[i] = a & b & c & d;
			O[i] = a | b | c | d;
		}
		A[TW] = 0; O[TW] = 0;
		const u32 hAnd = halo[r00] & halo[r01] & halo[r10] & halo[r11];
		const u32 hOr = halo[r00] | halo[r01] | halo[r10] | halo[r11];
		// the sample right of the last valid cell sits at bit `validCells` of the row
#pragma unroll
		for (int i = 0; i <= TW; ++i) {
			if (i == (validCells >> 5)) { A[i] |= hAnd << (validCells & 31); O[i] |= hOr << (validCells & 31); }
		}
#pragma unroll
		for (int i = 0; i < TW; ++i) {
			const u32 sA = (A[i] >> 1) | (A[i + 1] << 31), sO = (O[i] >> 1) | (O[i + 1] << 31);
			const u32 nt = (O[i] | sO) & ~(A[i] & sA);
#pragma unroll
			for (int h = 0; h < 2; ++h) {
				const int j = 2 * i + h;
				u32 bits = (nt >> (h * 16)) & 0xFFFFu;
				if (j * 16 >= validCells) bits = 0;
				blockBits[j * 256 + tid] = (u16)bits;
				if (bits) { blockAny[j] = 1; atomicAdd(&blockCnt[j], (u32)__popc(bits)); }
			}
		}
	}
	__syncthreads();
	CLS_TICK(2);

	// ---- one lane per block: emptiness rule, slot allocation (one reservation per tile) -------------------
	if (tid < 64) { // the first wave; lanes >= TB only take part in the ballots
		const bool mine = tid < TB && tid * 16 < validCells;
		const u32 bx = tx * TB + (u32)tid;
		const bool skipped = mine && (blockCls[tid] & BC_SKIPPED) != 0;
		const bool active = mine && blockAny[tid] != 0;
		const unsigned long long actMask = __ballot(active);
		u32 base = 0;
		if (tid == 0) {
			if (actMask) base = atomicAdd(L.nActive, (u32)__popcll(actMask));
		}
		base = __shfl(base, 0);
		if (active) {
			const u32 slot = base + (u32)__popcll(actMask & ((1ull << tid) - 1ull));
			const u32 id = block_coord_id(bx, by, bz, L.cnt);
			L.slotOf[id] = (int)slot;
			L.slotCoord[slot] = id;
			L.skip[slot] = skipped ? 1 : 0;
			L.ntCount[slot] = (u16)blockCnt[tid];
			if (blockCnt[tid] > (u32)LARGE_THRESHOLD) atomicAdd(p.G.largeBlocks, 1u);
			blockSkipped[tid] = skipped ? 1u : 0u;
			blockSlot[tid] = (int)slot;
			// The block's ancestors become active on the levels 1 .. levels-1 right here: whoever marks an ancestor first gives
			// it its slot.  All levels are tried at once - the claims of the different levels are independent addresses, so the
			// compare-and-swaps travel together and the slot counters of the levels won follow in a second round trip: two
			// dependent round trips per block instead of two per level (as a chain that stopped at the first ancestor somebody
			// else owned this cost k_classify +21 us at 1024^3 and was left to a launch of its own, k_hierarchy, there).
			if (activateAncestors) {
				int won[MAX_LEVELS];
#pragma unroll
				for (u32 l = 1; l < (u32)MAX_LEVELS; ++l) {
					won[l] = 0;
					if (l < p.G.levels) {
						const LevelDesc& A = p.levels[l];
						const u32 px = bx >> l, py = by >> l, pz = bz >> l;
						if (px < A.cnt && py < A.cnt && pz < A.cnt) won[l] = atomicCAS(&A.slotOf[block_coord_id(px, py, pz, A.cnt)], -1, -2) == -1 ? 1 : 0;
					}
				}
				u32 aslot[MAX_LEVELS];
#pragma unroll
				for (u32 l = 1; l < (u32)MAX_LEVELS; ++l) if (won[l]) aslot[l] = atomicAdd(p.levels[l].nActive, 1u);
#pragma unroll
				for (u32 l = 1; l < (u32)MAX_LEVELS; ++l) {
					if (!won[l]) continue;
					const LevelDesc& A = p.levels[l];
					const u32 aid = block_coord_id(bx >> l, by >> l, bz >> l, A.cnt);
					A.slotCoord[aslot[l]] = aid;
					A.slotOf[aid] = (int)aslot[l]; // visible to the next kernel
				}
			}
		}
	}
	__syncthreads();
	CLS_TICK(3);
#pragma unroll
	for (int j = 0; j < TB; ++j) {
		const int slot = blockSlot[j];
		if (slot >= 0) {
			const u16 bits = blockBits[j * 256 + tid];
			((u16*)(L.ntBits + (size_t)slot * 128))[tid] = bits;
			((u16*)(L.consBits + (size_t)slot * 128))[tid] = blockSkipped[j] ? (u16)0 : bits;
		}
	}
	CLS_TICK(4);
}

// ------------------------------------------------------------------------------------------------------
// k_hierarchy: ancestors of active level-0 blocks become active on levels 1..levels-1
// ------------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(WG) void k_hierarchy(ExecParamsDev p, u32 levels)
{
	const LevelDesc& L0 = p.levels[0];
	const u32 s = blockIdx.x * WG + threadIdx.x;
	if (s >= *L0.nActive) return;
	u32 bx, by, bz;
	block_coords(L0.slotCoord[s], L0.cnt, bx, by, bz);
	for (u32 l = 1; l < levels; ++l) {
		const LevelDesc& L = p.levels[l];
		const u32 px = bx >> l, py = by >> l, pz = bz >> l;
		if (px >= L.cnt || py >= L.cnt || pz >= L.cnt) break;
		const u32 id = block_coord_id(px, py, pz, L.cnt);
		const int old = atomicCAS(&L.slotOf[id], -1, -2);
		if (old != -1) break; // somebody else owns this ancestor chain
		const u32 slot = atomicAdd(L.nActive, 1u);
		L.slotCoord[slot] = id;
		L.slotOf[id] = (int)slot; // visible to the next kernel
	}
}

// ------------------------------------------------------------------------------------------------------
// k_material: LevelMaterialCache of one level >= 1 (same results as the mat_phase_* functions of tv_block.h,
// which the CPU emulation runs).  The block's cells are classified and selected bit-parallel, one (y,z) cell row
// per lane; only the sign of a sample is kept.  Global reads happen in four waves per block: block coordinates;
// samples + child slots; child bitmaps; child entries of the voted cells.  The 8 KB cache block is assembled in
// LDS and written with 16-byte stores.
// ------------------------------------------------------------------------------------------------------
struct MatLds {
	u32 rowMask[292];          // sign bits of the 17 samples of sample row r = k * 17 + j
	u16 ntRow[256];            // non-trivial cells of cell row (y,z) = the block's ntBits
	u32 childBits[8][128];     // level 1: consistency bitmaps of the 2x2x2 child blocks
	int childSlot[8];
	u32 childSkip[8];          // level 1, children without a classification pass: the emptiness rule skips the child
	u32 voteCount, ntTotal;
	u16 voteList[BLOCK_CELLS];
	__attribute__((aligned(16))) u16 out[BLOCK_CELLS];
};

// majority vote over eight id | blend << 8 entries (255 = no entry): vote_entries() of tv_core.h without
// dynamically indexed arrays — an entry's count is compared in order of first appearance, first maximum wins
__device__ __forceinline__ u32 vote8(const u32 e[8])
{
	// (the blend sum is formed for the winning id alone, behind the counting: the 64 comparisons of the counting carry one
	// accumulation instead of two - a material block votes on every cell of its six boundary layers, ~1 700 votes, and this
	// function was an eighth of k_main's vector instructions)
	u32 ids[8];
#pragma unroll
	for (int i = 0; i < 8; ++i) ids[i] = e[i] & 0xFFu;
	u32 bestCnt = 0, bestId = 0;
#pragma unroll
	for (int i = 0; i < 8; ++i) {
		u32 cnt = 0;
#pragma unroll
		for (int j = 0; j < 8; ++j) cnt += ids[j] == ids[i] ? 1u : 0u;
		if (ids[i] != EMPTY_MATERIAL && cnt > bestCnt) { bestCnt = cnt; bestId = ids[i]; }
	}
	if (!bestCnt) return EMPTY_MATINFO;
	u32 bestBl = 0;
#pragma unroll
	for (int j = 0; j < 8; ++j) bestBl += ids[j] == bestId ? (e[j] >> 8) : 0u;
	// bl <= 8 * 255, cnt <= 8: the quotient is an integer or at least 1/8 away from one, fp32 division is exact enough
	const u32 avg = (u32)((float)bestBl / (float)bestCnt);
	return bestId | ((avg & 0xFFu) << 8);
}

// The same vote where it is usually trivial: on a terrain nearly every voted cell has children of ONE material (or none), so the
// 64 comparisons of the general count are spent on finding out that there is nothing to decide.  Every lane looks for its first
// entry and whether all its other entries carry that id or none; if that holds for every voting lane of the wave (`active`: the
// others pass anything) the winner is that id, its count the entries that carry it, its blend their mean - what vote8 computes
// for such a lane, with a third of the instructions.  One wave-uniform branch; a wave with a mixed lane takes the general vote.
__device__ __forceinline__ u32 vote8_mostly_uniform(const u32 e[8], bool active)
{
	u32 first = EMPTY_MATERIAL;
#pragma unroll
	for (int i = 7; i >= 0; --i) first = (e[i] & 0xFFu) != (u32)EMPTY_MATERIAL ? (e[i] & 0xFFu) : first;
	bool simple = true;
#pragma unroll
	for (int i = 0; i < 8; ++i) simple = simple && ((e[i] & 0xFFu) == first || (e[i] & 0xFFu) == (u32)EMPTY_MATERIAL);
	if (__ballot(active && !simple) != 0ull) return vote8(e);
	u32 cnt = 0, bl = 0;
#pragma unroll
	for (int i = 0; i < 8; ++i) { const bool mine = (e[i] & 0xFFu) == first; cnt += mine ? 1u : 0u; bl += mine ? (e[i] >> 8) : 0u; }
	if (first == (u32)EMPTY_MATERIAL) return EMPTY_MATINFO;
	const u32 avg = (u32)((float)bl / (float)cnt);
	return first | ((avg & 0xFFu) << 8);
}

// One block of one level >= 1.  GATED (k_main): the children's caches come from other workgroups of the same launch - waited
// for right in front of the vote, the only phase that reads them - and the block's own completion is published.
template <bool GATED, bool PARTIAL = false>
// selfChild (level 1): the children's consistency bitmaps are not read but formed here, from the children's own samples -
// the run has no classification pass (k_run_head<allocate>), and the level-0 blocks that form their bitmaps themselves run
// beside this block, in no order.
// boxLo / boxHi (incremental runs inside k_main): only the children inside this box of block coordinates are part of the run and
// publish; the others' caches are what earlier launches left.
// PARTIAL (vx_polygonize_from, emitFrom >= 1: no level-0 walk, no meshes below emitFrom): a level-1 block also writes what the
// walks of its level-0 children would have left - bitmap, consistency bits, cell count, an empty record - and every block of a
// level below emitFrom its own empty record (all seven meshes): the caches are complete for a later Modification, the lists
// hold nothing of these levels.
__device__ __forceinline__ void mat_block(const ExecParamsDev& p, u32 level, u32 slot, MatLds& st, const int tid, const bool selfChild = false, const u32* boxLo = nullptr, const u32* boxHi = nullptr, const u32 emitFrom = 0u)
{
	const LevelDesc& L = p.levels[level];
	const LevelDesc& C = p.levels[level - 1];
	const GridView& g = p.G.grid;
	const int n = g.n;
	const int mult = (int)L.mult;
	constexpr int PER = (SAMPLES + WG - 1) / WG; // 20 samples per lane
	constexpr int MAT_BATCH = 10;                 // of which this many are in flight together
	constexpr bool THROUGH = GATED;               // what other workgroups of the launch wait for leaves through write-through stores
	u32 bx, by, bz;
	block_coords(L.slotCoord[slot], L.cnt, bx, by, bz);
	const bool defineAll = !p.G.dirty || slot >= p.G.prevActive[level];
	u16* cacheOut = L.cache + (size_t)slot * BLOCK_CELLS;
	__syncthreads(); // the previous block of this workgroup is done with the LDS state
	TRACE_MARK(0);

	// ---- requests: child slots, old cache contents (incremental runs), samples ----------------------
	int cs = -1;
	if (tid < 8) {
		const u32 cx = bx * 2 + (tid & 1), cy = by * 2 + ((tid >> 1) & 1), cz = bz * 2 + (tid >> 2);
		if (cx < C.cnt && cy < C.cnt && cz < C.cnt) cs = C.slotOf[block_coord_id(cx, cy, cz, C.cnt)];
	}
	uint4 old0, old1;
	if (!defineAll) { old0 = ((const uint4*)cacheOut)[tid]; old1 = ((const uint4*)cacheOut)[tid + WG]; }
	const int x0 = (int)(bx * 16) * mult, y0 = (int)(by * 16) * mult, z0 = (int)(bz * 16) * mult;
	const i8* base = g.dist + dist_offset(g, x0, y0, z0); // uniform; lanes add 32-bit offsets
	const int pitch = g.pitchY;
	const PyramidLevel& pyr = p.G.pyr[level < PYRAMID_LEVELS ? level : 0];
	const bool lattice = level < PYRAMID_LEVELS && pyr.data != nullptr;
	for (int r = tid; r < 292; r += WG) st.rowMask[r] = 0;
	if (tid < 8) st.childSlot[tid] = cs;
	if (tid == 0) { st.voteCount = 0; st.ntTotal = 0; }
	if (defineAll) {
		const u32 e2 = (u32)EMPTY_MATINFO | ((u32)EMPTY_MATINFO << 16);
		old0 = make_uint4(e2, e2, e2, e2); old1 = old0;
	}
	((uint4*)st.out)[tid] = old0; ((uint4*)st.out)[tid + WG] = old1;
	__syncthreads();
	TRACE_MARK(1);
	if (lattice) {
		// the level's lattice copy (complete: it is kept with the grid's mirrors): one 16-byte load + one byte per sample row
		for (int r = tid; r < 289; r += WG) {
			const int k = r / 17, j = r - k * 17;
			const PyramidRow row = pyramid_row17(pyr, (int)(bx * 16), (int)(by * 16) + j, (int)(bz * 16) + k);
			st.rowMask[r] = sign_nibble(row.lo.x) | (sign_nibble(row.lo.y) << 4) | (sign_nibble(row.lo.z) << 8) | (sign_nibble(row.lo.w) << 12) | (((row.far >> 7) & 1u) << 16);
		}
	} else
#pragma unroll 1
	for (int q0 = 0; q0 < PER; q0 += MAT_BATCH) {
		i8 v[MAT_BATCH];
#pragma unroll
		for (int q = 0; q < MAT_BATCH; ++q) {
			const int sIdx = tid + (q0 + q) * WG;
			if (sIdx < SAMPLES) {
				const int i = sIdx % 17, j = (sIdx / 17) % 17, k = sIdx / 289;
				const int dx = min(x0 + i * mult, n - 1) - x0, dy = min(y0 + j * mult, n - 1) - y0, dz = min(z0 + k * mult, n - 1) - z0;
				v[q] = base[(u32)((dz * pitch + dy) * n + dx)];
			}
		}
#pragma unroll
		for (int q = 0; q < MAT_BATCH; ++q) {
			const int sIdx = tid + (q0 + q) * WG;
			if (sIdx < SAMPLES) atomicOr(&st.rowMask[sIdx / 17], (((u32)(v[q] >> 7)) & 1u) << (sIdx % 17));
		}
	}
	TRACE_MARK(2);
	// ---- child bitmaps (level 1) requested while the rows are classified -----------------------------
	u32 cb4[4] = { 0, 0, 0, 0 };
	if (level == 1 && selfChild) {
		// The 2 x 2 x 2 children span 33 x 33 sample rows (Y, Z) of 33 samples; a row is two 16-byte pieces of the brick mirror
		// (x half h = the child column) and the sample behind them.  Sign masks: 16 bits per piece, one byte per far sample,
		// on top of the vote list (written after the bitmaps are complete).  A piece only some absent child would read is
		// not resident on a rank that owns a slab of the grid: it is read from a resident address instead and masked.
		static_assert(sizeof(st.voteList) >= 1089 * 4 + 1092, "row masks of the children on top of the vote list");
		u32* rowW = (u32*)st.voteList;                  // [row]: sign bits of the samples 0..31
		u8* farBit = (u8*)st.voteList + 1089 * 4;       // [row]: sign bit of sample 32
		// Addresses relative to the first child's brick, in 32 bits (k_main runs on mirrors below 4 GiB): the brick index is
		// linear in the block coordinates and brick_local's bit fields are disjoint per axis, so a row's offset is
		// ((Z >> 4) * bricks per plane + (Y >> 4) * bricks per row + h) << 12 | local(Y & 15, Z & 15).
		const u32 nbU = (u32)n >> 4, planeBricks = (u32)g.bRowsY * nbU;
		const size_t origin = brick_base(g, (int)(bx * 2), (int)(by * 2), (int)(bz * 2));
		u32 ex = 0, residentRel = 0;
#pragma unroll
		for (int c = 7; c >= 0; --c) if (st.childSlot[c] >= 0) { ex |= 1u << c; residentRel = (((u32)(c >> 2) * planeBricks + (u32)((c >> 1) & 1) * nbU + (u32)(c & 1)) << 12); }
		if (tid < 8) st.childSkip[tid] = st.childSlot[tid] >= 0 ? (u32)C.skip[st.childSlot[tid]] : 1u;
		const u32 exPair = (ex | (ex >> 1)) & 0x55u;
		const auto needed = [&](u32 Y, u32 Z) {
			// (the second piece of a row also holds sample 16 of the first child column: x is never partitioned)
			const u32 sel = ((Y <= 16u ? 0x11u : 0u) | (Y >= 16u ? 0x44u : 0u)) & ((Z <= 16u ? 0x0Fu : 0u) | (Z >= 16u ? 0xF0u : 0u));
			return (exPair & sel) != 0u;
		};
		const u32 yMax = (u32)(n - 1 - (int)(by * 32)), zMax = (u32)(n - 1 - (int)(bz * 32)); // (>= 31: the clamp binds on row / plane 32 of the grid's last blocks)
		const auto rowRel = [&](u32 Y, u32 Z) {
			const u32 yc = min(Y, yMax), zc = min(Z, zMax);
			return ((__umul24(zc >> 4, planeBricks) + __umul24(yc >> 4, nbU)) << 12) | (((zc & 15u) >> 1) << 9) | (((yc & 15u) >> 2) << 7) | ((zc & 1u) << 6) | ((yc & 3u) << 4);
		};
		const i8* base = g.bDist + origin;
		// the sample behind the second piece: x = min(X0 + 32, n - 1) - the first byte of the third brick along x, or, at the
		// grid's far side, the last byte of the second
		const bool lastX = (int)(bx * 32) + 32 > n - 1;
		const u32 farRel = lastX ? ((1u << 12) | 15u) : (2u << 12);
		// the sign bits of four bytes as a nibble: the bits 7, 15, 23, 31 times (1 + 2^7 + 2^14 + 2^21) meet in the bits 28..31 (no two
		// partial products share a bit, nothing carries)
		const auto sign4 = [](u32 d) { return ((d & 0x80808080u) * 0x00204081u) >> 28; };
		const auto sign16 = [&](uint4 d) { return sign4(d.x) | (sign4(d.y) << 4) | (sign4(d.z) << 8) | (sign4(d.w) << 12); };
		// A lane takes the rows tid, tid + 256, ... (row = Z * 33 + Y; 256 = 7 * 33 + 25: the next row is 7 planes and 25 rows
		// on) and reads BOTH pieces and the far sample of each: three loads per row from one offset, all of a batch in flight
		// before the first mask is formed.  (Round 6: the pieces used to be dealt out one by one, 2 178 of them with an address
		// of their own each, and the consumer below picked its half per lane - together a fifth of a material block's vector
		// instructions at 1024^3.)
		u32 Z = ((u32)tid * 1986u) >> 16, Y = (u32)tid - Z * 33u; // (tid / 33, exact below 1280)
#pragma unroll
		for (int batch = 0; batch < 2; ++batch) {
			const int rows = batch ? 2 : 3;
			uint4 a[3], b[3];
			i8 f[3];
			u32 rowIdx[3];
#pragma unroll
			for (int q = 0; q < 3; ++q) {
				if (q < rows) {
					const u32 r = Z * 33u + Y;
					rowIdx[q] = r;
					u32 off = residentRel, offB = residentRel, offF = residentRel;
					if (r <= 1088u && needed(Y, Z)) { off = rowRel(Y, Z); offB = off + (1u << 12); offF = off + farRel; }
					a[q] = *(const uint4*)(base + off);
					b[q] = *(const uint4*)(base + offB);
					f[q] = base[offF];
					Y += 25u; Z += 7u;
					if (Y >= 33u) { Y -= 33u; Z += 1u; }
				}
			}
#pragma unroll
			for (int q = 0; q < 3; ++q) {
				if (q < rows && rowIdx[q] <= 1088u) {
					rowW[rowIdx[q]] = sign16(a[q]) | (sign16(b[q]) << 16);
					farBit[rowIdx[q]] = (u8)(((u32)(f[q] >> 7)) & 1u);
				}
			}
		}
	} else if (level == 1) {
#pragma unroll
		for (int q = 0; q < 4; ++q) {
			const int w = tid + q * WG;
			const int c = st.childSlot[w >> 7];
			if (c >= 0) cb4[q] = C.consBits[(size_t)c * 128 + (w & 127)];
		}
	}
	__syncthreads();
	TRACE_MARK(3);
	const int y = tid & 15, z = tid >> 4;
	u32 nt;
	{
		const u32 a = st.rowMask[z * 17 + y], b2 = st.rowMask[z * 17 + y + 1], c = st.rowMask[(z + 1) * 17 + y], d = st.rowMask[(z + 1) * 17 + y + 1];
		const u32 A = a & b2 & c & d, O = a | b2 | c | d;
		nt = ((O | (O >> 1)) & ~(A & (A >> 1))) & 0xFFFFu;
		st.ntRow[tid] = (u16)nt;
		if (!THROUGH) ((u16*)(L.ntBits + (size_t)slot * 128))[tid] = (u16)nt;
		if (nt) atomicAdd(&st.ntTotal, (u32)__popc(nt));
	}
	if (level == 1 && selfChild) {
		const u32* rowW = (const u32*)st.voteList;
		const u8* farBit = (const u8*)st.voteList + 1089 * 4;
		const bool lastHalf = (int)(bx * 32) + 16 >= n; // the grid ends behind the first child column: sample 16 = sample 15
		// cell row (Y, Z) of the 32 x 32 child cell rows: both x halves from the same four sample rows (AND / OR over the rows
		// first, the halves cut out of the results)
#pragma unroll
		for (int q = 0; q < 4; ++q) {
			const int Yc = tid & 31, Zc = (tid >> 5) + q * 8;
			const int r00 = Zc * 33 + Yc;
			const u32 w0 = rowW[r00], w1 = rowW[r00 + 1], w2 = rowW[r00 + 33], w3 = rowW[r00 + 34];
			const u32 f0 = farBit[r00], f1 = farBit[r00 + 1], f2 = farBit[r00 + 33], f3 = farBit[r00 + 34];
			const u32 A = w0 & w1 & w2 & w3, O = w0 | w1 | w2 | w3, Af = f0 & f1 & f2 & f3, Of = f0 | f1 | f2 | f3;
			const u32 A0 = lastHalf ? ((A & 0xFFFFu) | ((A & 0x8000u) << 1)) : (A & 0x1FFFFu), O0 = lastHalf ? ((O & 0xFFFFu) | ((O & 0x8000u) << 1)) : (O & 0x1FFFFu);
			const u32 A1 = (A >> 16) | (Af << 16), O1 = (O >> 16) | (Of << 16);
			u32 bits0 = ((O0 | (O0 >> 1)) & ~(A0 & (A0 >> 1))) & 0xFFFFu, bits1 = ((O1 | (O1 >> 1)) & ~(A1 & (A1 >> 1))) & 0xFFFFu;
			const int cb = ((Yc >> 4) << 1) | ((q >> 1) << 2);
			if (st.childSkip[cb]) bits0 = 0;              // (absent children count as skipped)
			if (st.childSkip[cb | 1]) bits1 = 0;
			const int at = ((Zc & 15) << 4) | (Yc & 15);
			((u16*)st.childBits[cb])[at] = (u16)bits0;
			((u16*)st.childBits[cb | 1])[at] = (u16)bits1;
		}
	} else if (level == 1) {
#pragma unroll
		for (int q = 0; q < 4; ++q) { const int w = tid + q * WG; st.childBits[w >> 7][w & 127] = cb4[q]; }
	}
	__syncthreads();
	TRACE_MARK(4);
	if (PARTIAL && level == 1u && selfChild) {
		// (what f0_self_bits / f0_next<SELF> leave behind a level-0 block: here for all eight children at once)
#pragma unroll
		for (int q = 0; q < 4; ++q) {
			const int w = tid + q * WG, c = w >> 7, cs = st.childSlot[c];
			if (cs >= 0) {
				const u32 bits = st.childBits[c][w & 127];
				C.ntBits[(size_t)cs * 128 + (w & 127)] = bits;
				C.consBits[(size_t)cs * 128 + (w & 127)] = bits;
			}
		}
		if (tid < 8 && st.childSlot[tid] >= 0) {
			u32 cells = 0;
			for (int w = 0; w < 128; ++w) cells += (u32)__popc(st.childBits[tid][w]);
			C.ntCount[st.childSlot[tid]] = (u16)cells;
			reg_write_empty_record(C, (u32)st.childSlot[tid]);
		}
	}
	if (PARTIAL && level < emitFrom && tid == 0) {
		reg_write_empty_record(L, slot);
		BlockRecord& r = L.records[slot];
		for (int f = 0; f < 6; ++f) { r.tvOff[f] = r.tvCount[f] = r.tiOff[f] = r.tiCount[f] = 0; }
	}
	if (THROUGH && tid < 32) store16_through(L.ntBits + (size_t)slot * 128, (u32)tid * 16u, ((const uint4*)st.ntRow)[tid]); // the bitmap, 16 bytes per lane
	// ---- selection: cells that need an entry (non-trivial, or visited by the transition pass) and have
	//      any child entry to vote on.  All eight children of a cell live in one child block. -----------
	{
		u32 want = nt;
		if (L.hasTransitions) {
			const bool wholeRow = (z == 0 && bz > 0) || (y == 0 && by > 0) || (z == 15 && bz + 1 < L.cnt) || (y == 15 && by + 1 < L.cnt);
			if (wholeRow) want = 0xFFFFu;
			else { if (bx > 0) want |= 1u; if (bx + 1 < L.cnt) want |= 0x8000u; }
		}
		const u32 cbLo = (u32)(((y >> 3) << 1) | ((z >> 3) << 2));
		u32 have = 0;
#pragma unroll
		for (u32 h = 0; h < 2; ++h) {
			const u32 cb = cbLo | h;
			if (st.childSlot[cb] < 0) continue;
			u32 r = 0xFFFFu;
			if (level == 1) {
				r = 0;
#pragma unroll
				for (int dd = 0; dd < 4; ++dd) {
					const u32 row = (u32)((((2 * z + (dd >> 1)) & 15) << 4) | ((2 * y + (dd & 1)) & 15));
					r |= (st.childBits[cb][row >> 1] >> ((row & 1u) * 16u)) & 0xFFFFu;
				}
				r |= r >> 1;                       // child pair (2x', 2x'+1) -> even bit
				r &= 0x5555u; r = (r | (r >> 1)) & 0x3333u; r = (r | (r >> 2)) & 0x0F0Fu; r = (r | (r >> 4)) & 0x00FFu;
			} else {
				r = 0xFFu;
			}
			have |= (r & 0xFFu) << (8u * h);
		}
		u32 cand = want & have;
		if (cand) {
			u32 pos = atomicAdd(&st.voteCount, (u32)__popc(cand));
			const u32 rowBase = (u32)tid << 4;
			while (cand) {
				const u32 x = (u32)__builtin_ctz(cand);
				cand &= cand - 1;
				st.voteList[pos++] = (u16)(rowBase | x);
			}
		}
		if (tid == 0) {
			const u32 cnt = st.ntTotal;
			L.ntCount[slot] = (u16)cnt;
			if (cnt > (u32)LARGE_THRESHOLD) atomicAdd(p.G.largeBlocks, 1u);
			if (!p.G.dirty) { // (the slot counts of all levels are final since k_hierarchy)
				u32 before = 0;
				for (u32 l = 1; l < level; ++l) before += p.G.slotCounts[l];
				FlatItem e;
				e.where = (level << 24) | slot; e.coordId = L.slotCoord[slot]; e.ntCells = cnt; e.pad = 0;
				p.G.flatItems[before + slot] = e;
			}
		}
	}
	__syncthreads();
	TRACE_MARK(5);
	if (GATED && level >= 2u) {
		// the children's cache blocks are written by other workgroups of this launch (the level below comes first in the queue)
		if (tid < 8) {
			const int c = st.childSlot[tid];
			bool inRun = true;
			if (boxLo) {
				const u32 cx = bx * 2 + (tid & 1), cy = by * 2 + ((tid >> 1) & 1), cz = bz * 2 + (tid >> 2);
				inRun = cx >= boxLo[0] && cx < boxHi[0] && cy >= boxLo[1] && cy < boxHi[1] && cz >= boxLo[2] && cz < boxHi[2];
			}
			if (c >= 0 && inRun) (void)wait_done(C.matDone + c, p.G.epoch, p.G.giveUp);
		}
		acquire_and_meet(tid < 64);
	}
	TRACE_MARK(6);
	// ---- vote.  A lane takes VB cells per trip and requests ALL their child entries before it looks at any: children
	//      that are neighbours along x come in one load (two u16 entries / two material bytes), and no load is
	//      conditional (a load with a default value is waited for on the spot) — the children of a cell are always
	//      inside the grid; entries the consistency bits rule out are masked after the fact. ----------------------------
	{
		const int nVote = (VX_ABL & 16) ? 0 : (int)st.voteCount;
		// the children's materials come from the brick mirrors: the 2 x 2 x 2 child blocks are 8 consecutive-in-x pairs of
		// 4 KB bricks, a cell's 8 children sit in 2 lines per field (4 in the dense fields)
		const size_t childOrigin = brick_base(g, (int)(bx * 2), (int)(by * 2), (int)(bz * 2));
		const u8* matBase = g.bMat + childOrigin;
		const u8* blendBase = g.bBlend + childOrigin;
		const u32 brickRow = (u32)(g.n >> 4) * BRICK_BYTES, brickPlane = (u32)g.bRowsY * brickRow; // next child block along y / z
		if (level == 1) {
			constexpr int VB = 2;
			for (int k0 = tid; k0 < nVote; k0 += WG * VB) {
				u32 mat2[VB][4], bl2[VB][4];
#pragma unroll
				for (int v = 0; v < VB; ++v) {
					const u32 c = st.voteList[min(k0 + v * WG, nVote - 1)];
					const int lx = (int)(c & 15), ly = (int)((c >> 4) & 15), lz = (int)(c >> 8);
#pragma unroll
					for (int q = 0; q < 4; ++q) {
						const u32 cx = 2u * (u32)lx, cy = 2u * (u32)ly + (u32)(q & 1), cz = 2u * (u32)lz + (u32)(q >> 1);
						const u32 off = (cz >> 4) * brickPlane + (cy >> 4) * brickRow + (cx >> 4) * BRICK_BYTES + brick_local(cx & 15u, cy & 15u, cz & 15u);
						mat2[v][q] = *(const u16*)(matBase + off);
						bl2[v][q] = *(const u16*)(blendBase + off);
					}
				}
#pragma unroll
				for (int v = 0; v < VB; ++v) {
					if (k0 + v * WG >= nVote) continue;
					const u32 c = st.voteList[k0 + v * WG];
					const int lx = (int)(c & 15), ly = (int)((c >> 4) & 15), lz = (int)(c >> 8);
					const u32 cb = (u32)((lx >> 3) | ((ly >> 3) << 1) | ((lz >> 3) << 2));
					u32 e[8];
#pragma unroll
					for (int i = 0; i < 8; ++i) {
						const int cxx = 2 * lx + (i & 1), cyy = 2 * ly + ((i >> 1) & 1), czz = 2 * lz + (i >> 2);
						const u32 local = (u32)(((czz & 15) << 8) | ((cyy & 15) << 4) | (cxx & 15));
						const u32 sh = (u32)(i & 1) * 8u;
						const u32 entry = ((mat2[v][i >> 1] >> sh) & 0xFFu) | (((bl2[v][i >> 1] >> sh) & 0xFFu) << 8);
						e[i] = ((st.childBits[cb][local >> 5] >> (local & 31u)) & 1u) ? entry : (u32)EMPTY_MATINFO;
					}
					const u32 entry = vote8_mostly_uniform(e, true);
					if (defineAll || entry != (u32)EMPTY_MATINFO) st.out[c] = (u16)entry;
				}
			}
		} else {
			constexpr int VB = 4;
			for (int k0 = tid; k0 < nVote; k0 += WG * VB) {
				u32 pair[VB][4];
#pragma unroll
				for (int v = 0; v < VB; ++v) {
					const u32 c = st.voteList[min(k0 + v * WG, nVote - 1)];
					const int lx = (int)(c & 15), ly = (int)((c >> 4) & 15), lz = (int)(c >> 8);
					const u32 cb = (u32)((lx >> 3) | ((ly >> 3) << 1) | ((lz >> 3) << 2));
					const u16* child = C.cache + (size_t)st.childSlot[cb] * BLOCK_CELLS;
#pragma unroll
					for (int q = 0; q < 4; ++q)
						pair[v][q] = TV_LOAD_THROUGH((const u32*)(child + ((((2 * lz + (q >> 1)) & 15) << 8) | (((2 * ly + (q & 1)) & 15) << 4) | ((2 * lx) & 15))));
				}
#pragma unroll
				for (int v = 0; v < VB; ++v) {
					if (k0 + v * WG >= nVote) continue;
					const u32 c = st.voteList[k0 + v * WG];
					u32 e[8];
#pragma unroll
					for (int i = 0; i < 8; ++i) e[i] = (pair[v][i >> 1] >> ((u32)(i & 1) * 16u)) & 0xFFFFu;
					const u32 entry = vote8_mostly_uniform(e, true);
					if (defineAll || entry != (u32)EMPTY_MATINFO) st.out[c] = (u16)entry;
				}
			}
		}
	}
	__syncthreads();
	TRACE_MARK(7);
	if (THROUGH) {
		store16_through(cacheOut, (u32)tid * 16u, ((const uint4*)st.out)[tid]);
		store16_through(cacheOut, (u32)(tid + WG) * 16u, ((const uint4*)st.out)[tid + WG]);
		publish_done_through(L.matDone + slot, p.G.epoch, st.ntTotal);
	} else {
		((uint4*)cacheOut)[tid] = ((const uint4*)st.out)[tid];
		((uint4*)cacheOut)[tid + WG] = ((const uint4*)st.out)[tid + WG];
	}
}

__global__ __launch_bounds__(WG) __attribute__((amdgpu_waves_per_eu(6))) void k_material(ExecParamsDev p, u32 level)
{
	__shared__ MatLds st;
	const u32 nItems = p.G.dirty ? p.G.workCount[level] : *p.levels[level].nActive;
	for (u32 it = blockIdx.x; it < nItems; it += gridDim.x) mat_block<false>(p, level, p.G.dirty ? p.G.workItems[level][it] : it, st, (int)threadIdx.x);
}


// ------------------------------------------------------------------------------------------------------
// k_regular / k_transition: workgroups striding over (level, slot) work items
// ------------------------------------------------------------------------------------------------------
struct WorkList {
	u32 start[MAX_LEVELS + 1];
};

// Slots are handed out in tile order, so neighbouring item numbers are x-neighbour blocks sharing cache lines.
// Workgroup b runs on XCD b % 8 (observed; speed only): transpose every group of 64 items so that 8 consecutive
// items go to workgroups of the same XCD and meet in that XCD's L2 (grids are multiples of 8 workgroups).
__device__ __forceinline__ u32 xcd_item(u32 i) { return (i & ~63u) | ((i & 7u) << 3) | ((i >> 3) & 7u); }

__device__ __forceinline__ void decode_item(const WorkList& wl, u32 levels, u32 item, u32& level, u32& slot)
{
	level = 0;
	for (u32 l = 1; l < levels; ++l) if (item >= wl.start[l]) level = l;
	slot = item - wl.start[level];
}

extern __shared__ __attribute__((aligned(16))) unsigned char smem[];

// LDS images of the table subsets each kernel needs
constexpr u32 REG_TAB_LDS = 512 + 1600 + 256;           // regClass + regCell | edge words + regVert rows | regOwn
constexpr u32 TR_TAB_LDS = 2832 + 3072 + 1024;          // trClass + trCorner + trCell + edge words | trVert rows | trOwn

// (`tid`: the lane's index; a caller inside a loop hands in an opaque copy so that the lane's addresses are not hoisted out of
// the loop and kept in registers through everything else)
__device__ __forceinline__ void copy16(u8* dst, const u8* src, u32 bytes, u32 tid)
{
	const uint4* s = (const uint4*)src;
	uint4* d = (uint4*)dst;
	for (u32 i = tid; i < bytes / 16; i += WG) d[i] = s[i];
}
__device__ __forceinline__ void copy16(u8* dst, const u8* src, u32 bytes) { copy16(dst, src, bytes, threadIdx.x); }

__device__ __forceinline__ Tables stage_regular_tables(u8* lds, const u8* image)
{
	copy16(lds, image + TAB_REG_CLASS, 512);
	copy16(lds + 512, image + TAB_REG_EDGE, 1600); // both edge word tables + the regular rows (contiguous in the image)
	copy16(lds + 2112, image + TAB_REG_OWN, 256);
	Tables T;
	T.regClassP = lds; T.regCellP = lds + 256; T.regEdgeP = (const u16*)(lds + 512); T.regVertP = lds + 512 + 64; T.regOwnP = lds + 2112;
	T.trClassP = nullptr; T.trCornerP = nullptr; T.trCellP = nullptr; T.trVertP = nullptr; T.trEdgeP = nullptr; T.trOwnP = nullptr;
	return T;
}

__device__ __forceinline__ Tables stage_transition_tables(u8* lds, const u8* image, u32 tid)
{
	copy16(lds, image + TAB_TR_CLASS, 2832, tid);       // class, corner, cell tables + both edge word tables
	copy16(lds + 2832, image + TAB_TR_VERT, 3072, tid);
	copy16(lds + 5904, image + TAB_TR_OWN, 1024, tid);
	Tables T;
	T.regClassP = nullptr; T.regCellP = nullptr; T.regVertP = nullptr; T.regEdgeP = nullptr; T.regOwnP = nullptr;
	T.trClassP = lds; T.trCornerP = lds + 512; T.trCellP = lds + 528; T.trEdgeP = (const u16*)(lds + 2768 + 32); T.trVertP = lds + 2832; T.trOwnP = (const u16*)(lds + 5904);
	return T;
}

} // namespace

#include "vx_regular0.inl"
#include "vx_fast0.inl"
#include "vx_fast1.inl"

namespace {

// One capacity class of the regular pass of levels >= 1: blocks with lo < non-trivial cells <= CAP
// 4 waves per SIMD: the lockstep LOD chains of the vertex emission keep a batch of vertices in registers; at 5 waves (96
// registers) the pass spilled them to scratch — and produced wrong vertices now and then (drop-in byte-dump test)
#if !defined(VX_REG_WAVES)
#define VX_REG_WAVES 4
#endif
// MODE 0: the slots of the levels [levelBegin, levels) (or the work lists of an incremental run); MODE 2: the blocks the
// fast pass of the levels >= 1 (vx_fast1.inl) handed on (Globals::slowItems[1])
// (the pass as workgroup `first` of `stride`: a launch of its own - k_regular below - or the second group of workgroups of
// k_tail; returns whether the workgroup wrote anything)
template <int CAP, int MODE>
__device__ __forceinline__ bool regular_pass(const ExecParamsDev& p, u32 levelBegin, u32 levels, u32 lo, const u32 first, const u32 stride)
{
	typedef RegStateT<CAP> ST;
	if (lo && *p.G.largeBlocks == 0) return false; // nothing for the 4096-cell class (uniform over the grid)
	u8* tab = smem;
	ST& st = *(ST*)(smem + REG_TAB_LDS);
	__shared__ WorkList wl;
	__shared__ u32 scanScratch[8];
	__shared__ u32 wgStats[20]; // statistics of every block this workgroup handles, flushed once at the end

	if (threadIdx.x < 20) wgStats[threadIdx.x] = 0;
	if (threadIdx.x == 0) {
		u32 run = 0;
		for (u32 l = 0; l < levels; ++l) { wl.start[l] = run; if (l >= levelBegin) run += p.G.dirty ? p.G.workCount[l] : *p.levels[l].nActive; }
		for (u32 l = levels; l <= MAX_LEVELS; ++l) wl.start[l] = run;
		if (MODE == 2) wl.start[MAX_LEVELS] = p.G.slowCount[1];
	}
	__syncthreads();
	const u32 total = wl.start[MAX_LEVELS];
	if (first >= ((total + 63u) & ~63u)) return false; // the grid is sized before the block counts are known
	const Tables T = stage_regular_tables(tab, p.tables); // visible after the first barrier of the item loop
	const int tid = threadIdx.x;
#if defined(VX_REG_PROFILE)
	u32 prof[16] = { 0 };
	unsigned long long tick = __builtin_readcyclecounter();
#define RG_TICK(i) do { const unsigned long long now_ = __builtin_readcyclecounter(); prof[i] += (u32)(now_ - tick); tick = now_; } while (0)
#else
#define RG_TICK(i) do { } while (0)
#endif

	for (u32 it = first; it < ((total + 63u) & ~63u); it += stride) {
		const u32 item = xcd_item(it);
		if (item >= total) continue;
		RegBlockCtx b;
		if (MODE == 2) {
			const u32 packed = p.G.slowItems[1][item];
			b.level = packed >> 24; b.slot = packed & 0xFFFFFFu;
		} else {
			decode_item(wl, levels, item, b.level, b.slot);
			if (p.G.dirty) b.slot = p.G.workItems[b.level][b.slot];
		}
		const LevelDesc& L = p.levels[b.level];
		const u32 ntc = L.ntCount[b.slot];
		if ((lo && ntc <= lo) || ntc > (u32)CAP) continue;   // the first class (lo == 0) also owns empty blocks
		b.mult = L.mult;
		block_coords(L.slotCoord[b.slot], L.cnt, b.bx, b.by, b.bz);
		if (ntc == 0 || (b.level == 0 && L.skip[b.slot])) {
			if (tid == 0) reg_write_empty_record(L, b.slot);
			continue;
		}
		RG_TICK(0);
		__syncthreads();
		RG_TICK(1);
		reg_phase_begin(st, L, b.slot, tid, WG);
		gpu_reg_stage(p.G, b, st.samp);
		__syncthreads();
		RG_TICK(2);
		for (int w = tid; w < 128; w += WG) st.wordPrefix[w] = (u16)__popc(st.ntBits[w]);
		__syncthreads();
		{
			const u32 nt = block_exclusive_scan_u16(st.wordPrefix, 128, scanScratch);
			if (tid == 0) st.wordPrefix[128] = (u16)nt;
		}
		__syncthreads();
		RG_TICK(3);
		reg_phase_list(st, L, b, tid, WG);
		__syncthreads();
		RG_TICK(4);
		reg_phase_cells(st, T, p.G, L, b, tid, WG);
		__syncthreads();
		RG_TICK(5);
		reg_phase_count(st, T, b, tid, WG);
		__syncthreads();
		RG_TICK(6);
		{
			const u32 vt = block_exclusive_scan_u16(st.vbase, st.wordPrefix[128], scanScratch);
			if (tid == 0) { st.vTotal = vt; st.vOff = atomicAdd(&p.P.cursors[CUR_V], vt); }
		}
		__syncthreads();
		RG_TICK(7);
		for (u32 chunk = 0; chunk == 0 || chunk < st.vTotal; chunk += VDESC_CAP) {
			if (chunk) __syncthreads();
			reg_phase_describe(st, chunk, tid, WG);
			__syncthreads();
			RG_TICK(8);
			reg_phase_emit_vertices(st, T, p.G, p.P, b, chunk, tid, WG);
			RG_TICK(9);
		}
		__syncthreads();
		RG_TICK(10);
		reg_phase_keep(st, T, p.G, b, tid, WG);
		__syncthreads();
		RG_TICK(11);
		{
			const u32 it = block_exclusive_scan_u16(st.ibase, st.wordPrefix[128], scanScratch);
			if (tid == 0) { st.iTotal = it; st.iOff = atomicAdd(&p.P.cursors[CUR_I], it); }
		}
		__syncthreads();
		RG_TICK(12);
		for (u32 chunk = 0; chunk < st.iTotal; chunk += VDESC_CAP) {
			if (chunk) __syncthreads();
			reg_phase_stage_indices(st, T, chunk, tid, WG);
			__syncthreads();
			RG_TICK(13);
			reg_phase_flush_indices(st, T, p.P, chunk, tid, WG);
			RG_TICK(14);
		}
		reg_phase_record(st, wgStats, L, b, p.P, tid);
		RG_TICK(15);
	}
#if defined(VX_REG_PROFILE)
	if (tid == 0 && !lo) for (int i = 0; i < 16; ++i) if (prof[i]) atomicAdd(&p.G.largeBlocks[16 + i], prof[i] >> 10); // header words 192..207, units of 1024 cycles
#endif
	__syncthreads();
	if (threadIdx.x < 20 && wgStats[threadIdx.x]) atomicAdd(&p.G.stats[threadIdx.x], wgStats[threadIdx.x]);
	return true;
}

template <int CAP, int MODE>
__global__ __launch_bounds__(WG) __attribute__((amdgpu_waves_per_eu(CAP > REG_CAP_SMALL ? 1 : VX_REG_WAVES))) void k_regular(ExecParamsDev p, u32 levelBegin, u32 levels, u32 lo)
{
	(void)regular_pass<CAP, MODE>(p, levelBegin, levels, lo, blockIdx.x, gridDim.x);
}

// Samples of one boundary plane of a block (tr_phase_load of tv_block.h for a compile-time face): the plane is an
// affine image of (u,v), so a lane adds two scaled 32-bit terms to one base address per face.  Faces are only staged
// where a neighbour block exists, so the plane itself is inside the grid; in-plane coordinates clamp at the far edge.
template <int F>
__device__ __forceinline__ void tr_face_request(const GridView& g, const RegBlockCtx& b, u32 on, int tid, i8 (&v)[5])
{
	// a face without a neighbour block is requested all the same (its plane clamped into the grid) and zeroed by
	// tr_face_store: a branch around the loads would end with a wait for them, one round trip per face
	const FaceGeom fg = face_geom(F);
	const int n = g.n, mult = (int)b.mult, half = mult >> 1;
	int o[3] = { (int)(b.bx * 16) * mult, (int)(b.by * 16) * mult, (int)(b.bz * 16) * mult };
	const int maxU = n - 1 - o[fg.ua], maxV = n - 1 - o[fg.va];
	if (fg.positive) o[fg.axis] = min(o[fg.axis] + 16 * mult, n - 1);
	// no load is conditional (a load with a default value is waited for on the spot: 15 round trips instead of one);
	// lanes beyond the plane re-read its last sample and tr_face_store drops it.  The samples come from the brick mirror:
	// a plane x = const puts 8 of its samples into every 128-byte line it touches (one in the dense field).
#pragma unroll
	for (int q = 0; q < 5; ++q) {
		const int r = min(tid + q * WG, PLANE - 1);
		const int vv = r / 33, uu = r - vv * 33;
		int c[3];
		c[fg.axis] = o[fg.axis];
		c[fg.ua] = o[fg.ua] + min(uu * half, maxU);
		c[fg.va] = o[fg.va] + min(vv * half, maxV);
		v[q] = g.bDist[brick_offset(g, c[0], c[1], c[2])];
	}
}

__device__ __forceinline__ void tr_face_store(i8* plane, int tid, const i8 (&v)[5], bool faceOn)
{
#pragma unroll
	for (int q = 0; q < 5; ++q) { const int r = tid + q * WG; if (r < PLANE) { const int vv = r / 33; plane[vv * TR_PROW + (r - vv * 33)] = faceOn ? v[q] : (i8)0; } }
}

// The boundary planes of a level-L block are 33 x 33 points of the level L-1 lattice, and that lattice is resident in
// brick order: the lattice copy of level L-1, or the grid's own mirror for L = 1.  Addresses are sums of one term per
// axis (tv_core.h brick_local, pyramid_offset).  `last` = largest valid coordinate: n - 1 in the grid's mirror, n >> l in
// a lattice copy (which holds the clamped far sample).
template <typename OFF>
struct TrLatticeT {
	const i8* base;
	u32 bricksX, bricksY;
	int yOrg, zOrg, last;
	const i8* xp;        // the lattice's x-plane copy (XPlanes), or nullptr: the x faces are gathered from the bricks
	u32 xpRows, xpStride;
	__device__ __forceinline__ OFF tx(int X) const { return ((OFF)((u32)X >> 4) << 12) | ((u32)X & 15u); }
	__device__ __forceinline__ OFF ty(int Y) const
	{
		const u32 y = (u32)(Y - yOrg);
		return ((OFF)__umul24(y >> 4, bricksX) << 12) | (((y >> 2) & 3u) << 7) | ((y & 3u) << 4);
	}
	__device__ __forceinline__ OFF tz(int Z) const
	{
		const u32 z = (u32)(Z - zOrg);
		return ((OFF)__umul24(__umul24(z >> 4, bricksY), bricksX) << 12) | (((z >> 1) & 7u) << 9) | ((z & 1u) << 6);
	}
};

// false: the lattice the block's planes lie in has no resident copy (levels beyond the lattice copies; shards without them)
template <typename OFF>
__device__ __forceinline__ bool tr_lattice_of(const Globals& G, u32 level, TrLatticeT<OFF>& lat)
{
	const GridView& g = G.grid;
	if (level == 1) {
		lat.base = g.bDist; lat.bricksX = (u32)g.n >> 4; lat.bricksY = (u32)g.bRowsY;
		lat.yOrg = g.bYb0 * 16; lat.zOrg = g.bZb0 * 16; lat.last = g.n - 1;
		lat.xp = G.xp[0].data; lat.xpRows = G.xp[0].rows; lat.xpStride = G.xp[0].stride;
		return true;
	}
	if (level - 1u >= (u32)PYRAMID_LEVELS || !G.pyr[level - 1u].data) return false;
	const PyramidLevel& P = G.pyr[level - 1u];
	lat.base = P.data; lat.bricksX = P.bricksX; lat.bricksY = P.bricksY;
	lat.yOrg = P.yOrigin; lat.zOrg = P.zOrigin; lat.last = g.n >> (level - 1u);
	const bool haveXp = level - 1u < (u32)XPLANE_LEVELS;
	lat.xp = haveXp ? G.xp[level - 1u].data : nullptr;
	lat.xpRows = haveXp ? G.xp[level - 1u].rows : 0u; lat.xpStride = haveXp ? G.xp[level - 1u].stride : 0u;
	return true;
}

// All planes of one block: the four faces whose rows run along x (z and y faces) are read row by row - two 16-byte pieces
// and the far sample per row, 132 rows, one lane each; the two x faces sample by sample (one byte of every 16-byte voxel
// row they cross).  Nothing is conditional on a face being on (a branch around loads ends with a wait for them; and which
// faces are on is only known once the sign summaries, requested right behind the planes, have arrived): faces that turn
// out to be off were read from clamped addresses and are not stored.
template <typename OFF>
__device__ __forceinline__ void tr_planes_request(const TrLatticeT<OFF>& lat, const RegBlockCtx& b, int tid, uint4& r0, uint4& r1, i8& rFar, i8 (&g)[9])
{
	// (the lane's row / sample indices do not depend on the block: left alone the compiler computes them once before the
	// item loop and keeps some twenty registers occupied through every other phase - which then spill)
	asm volatile("" : "+v"(tid));
	const int X0 = (int)b.bx * 32, Y0 = (int)b.by * 32, Z0 = (int)b.bz * 32;
	// ---- rows: lanes 0..131 the faces 0, 1, 3, 4 (rows along x in the bricks); with an x-plane copy of the lattice lanes
	//      132..197 the faces 2 and 5 (rows along y of the planes X0 / 32 and X0 / 32 + 1) ----
	{
		const int row = min(tid, lat.xp ? 197 : 131), fi = row / 33, rowV = row - fi * 33;
		const i8 *src, *hi, *far;
		if (fi < 4) {
			const bool zFace = (fi & 1) == 0, positive = fi >= 2;
			const int A = min((zFace ? Z0 : Y0) + (positive ? 32 : 0), lat.last);
			const int V = min((zFace ? Y0 : Z0) + rowV, lat.last);
			const OFF yz = lat.ty(zFace ? V : A) + lat.tz(zFace ? A : V);
			src = lat.base + yz + lat.tx(X0);
			hi = src + BRICK_BYTES;
			far = lat.base + yz + lat.tx(min(X0 + 32, lat.last));
		} else {
			// (a plane of the copy covers the lattice's whole y / z extent with its far entries: Y0 + 32 and Z0 + rowV exist)
			src = lat.xp + ((size_t)((u32)(X0 >> 5) + (fi == 5 ? 1u : 0u)) * lat.xpRows + (u32)(Z0 + rowV)) * lat.xpStride + (u32)Y0;
			hi = src + 16;
			far = src + 32;
		}
		r0 = *(const uint4*)src;
		r1 = *(const uint4*)hi;
		rFar = *far;
	}
	if (lat.xp) return; // (uniform)
	// ---- the x faces (2: x = X0, 5: x = X0 + 32) sample by sample from the bricks: samples (u, v) = (y, z) ----
	const OFF xNeg = lat.tx(X0), xPos = lat.tx(min(X0 + 32, lat.last));
#pragma unroll
	for (int q = 0; q < 9; ++q) {
		const int t = min(tid + q * WG, 2 * PLANE - 1);
		const int fi = t >= PLANE ? 1 : 0, r = t - fi * PLANE;
		const int vv = r / 33, uu = r - vv * 33;
		g[q] = lat.base[(fi ? xPos : xNeg) + lat.ty(min(Y0 + uu, lat.last)) + lat.tz(min(Z0 + vv, lat.last))];
	}
}

__device__ __forceinline__ void tr_planes_store(const uint4& r0, const uint4& r1, i8 rFar, const i8 (&g)[9], bool xRows, u32 on, int tid, TrState& st)
{
	asm volatile("" : "+v"(tid));
	if (tid < (xRows ? 198 : 132)) {
		const int fi = tid / 33, rowV = tid - fi * 33, rowFace = fi < 4 ? fi + (fi >= 2 ? 1 : 0) : (fi == 4 ? 2 : 5);
		if ((on >> rowFace) & 1u) {
			i8* dst = st.plane[rowFace] + rowV * TR_PROW;
			*(uint4*)dst = r0;
			*(uint4*)(dst + 16) = r1;
			dst[32] = rFar;
		}
	}
	if (xRows) return; // (uniform)
#pragma unroll
	for (int q = 0; q < 9; ++q) {
		const int t = tid + q * WG;
		if (t < 2 * PLANE) {
			const int fi = t >= PLANE ? 1 : 0, r = t - fi * PLANE;
			const int vv = r / 33, uu = r - vv * 33;
			if ((on >> (fi ? 5 : 2)) & 1u) st.plane[fi ? 5 : 2][vv * TR_PROW + uu] = g[q];
		}
	}
}

#if !defined(VX_TR_WAVES)
#define VX_TR_WAVES 5
#endif
// WIDE: a brick mirror of 4 GiB or more (grids beyond 1024^3): 64-bit voxel offsets around the vertices
// (three waves per SIMD there: the 64-bit address terms do not fit the 128 registers of four)
// One block of a level with transition cells.  GATED (k_main): the block's material cache comes from another workgroup of the
// same launch and is waited for where it is first read (planes, sign summaries, cell classification and scans need none of it).
template <bool WIDE, bool GATED>
__device__ __forceinline__ void tr_block(const ExecParamsDev& p, RegBlockCtx b, u32 coordId, TrState& st, const Tables& T, u32* scanScratch, u32* quietFaces, u32& quietParity,
                                         const BrickSamplerT<typename std::conditional<WIDE, size_t, u32>::type>& smp, int tid, const bool matKnown = true)
{
	// matKnown: the block's material cache is known to be complete (the stand-alone pass: earlier launches wrote it): its entries
	// behind the transition cells are then requested with the planes.  (Inside k_main that would need a launch-wide "all
	// material blocks published" counter: measured, the counter's hot line cost the material blocks more than the round trip
	// saved here.)
	bool matReady = !GATED || matKnown;
	const bool preMat = matReady;
#if defined(VX_TR_PROFILE)
	// tools builds: where a transition block's time goes (cycles / 64 as thread 0 sees them; header words 192..)
	unsigned long long trTick = __builtin_readcyclecounter();
#define TRB_TICK(i) do { const unsigned long long now_ = __builtin_readcyclecounter(); if (tid == 0) atomicAdd(&p.G.largeBlocks[16 + (i)], (u32)((now_ - trTick) >> 6)); trTick = now_; } while (0)
#else
#define TRB_TICK(i) do { } while (0)
#endif
	const LevelDesc& L = p.levels[b.level];
	b.mult = L.mult;
	block_coords(__builtin_amdgcn_readfirstlane(coordId), L.cnt, b.bx, b.by, b.bz);
	

	{
		u32 on = 0;
		const u32 bc[3] = { b.bx, b.by, b.bz };
		for (int f = 0; f < 6; ++f) {
			const FaceGeom fg = face_geom(f);
			if (fg.positive ? (bc[fg.axis] + 1 < L.cnt) : (bc[fg.axis] > 0)) on |= 1u << f;
		}
		// the planes are requested before the sign summaries below are looked at: both arrive in one round trip, and what
		// the summaries say only decides which planes are stored
		// (32-bit offsets: a lattice copy is at most an eighth of the grid; the grid's own mirror - level-1 planes - only
		// qualifies while it is smaller than 4 GiB)
		TrLatticeT<u32> lat;
		uint4 rowLo = { 0, 0, 0, 0 }, rowHi = { 0, 0, 0, 0 };
		i8 rowFar = 0, xFace[9] = { 0, 0, 0, 0, 0, 0, 0, 0, 0 };
		const bool haveLattice = !(WIDE && b.level == 1u) && tr_lattice_of(p.G, b.level, lat); // (uniform)
		if (haveLattice) tr_planes_request(lat, b, tid, rowLo, rowHi, rowFar, xFace);
		// the material entries of the low-res cells behind the 6 x 256 transition cells (lane t: cell t of every face)
		u16 fm[6] = { 0, 0, 0, 0, 0, 0 };
		if (preMat) { // (uniform)
			const u16* cache = L.cache + (size_t)b.slot * BLOCK_CELLS;
#pragma unroll
			for (int f = 0; f < 6; ++f) {
				int local[3];
				tr_low_local(face_geom(f), tid >> 4, tid & 15, local);
				fm[f] = TV_LOAD_THROUGH(&cache[(u32)((local[2] << 8) | (local[1] << 4) | local[0])]);
			}
		}
		TRB_TICK(0);
		
		// A boundary plane whose samples all have one sign holds no transition cell.  The sign summaries of the level-0
		// blocks (MirrorState::blockSign: "every voxel of the block's plane x = 0 / y = 0 / z = 0 is >= 0 / < 0") decide
		// that without reading the plane: the face covers (mult + 1)^2 of those block planes, its far edge included.
		// Faces found quiet are treated like faces without a neighbour block: not staged, no cells.
		if (p.G.blockSign && b.mult <= 8u) {
			const u32 cnt0 = p.levels[0].cnt, m1 = b.mult + 1u; // m1 <= 9: lane -> (du, dv) = (lane & 15, lane >> 4 + 4 * pass)
			u32 quiet = 0;
#pragma unroll
			for (int f = 0; f < 6; ++f) { // (unrolled: the face's axes are compile-time constants)
				if ((f & 3) != (tid >> 6) || !((on >> f) & 1u)) continue; // one wave per face; uniform per wave
				const FaceGeom fg = face_geom(f);
				const u32 field = fg.axis == 0 ? 1u : (fg.axis == 1 ? 2u : 4u);
				const u32 du = (u32)tid & 15u;
				const u32 qa = (fg.positive ? bc[fg.axis] + 1u : bc[fg.axis]) * b.mult;
				const u32 qu = min(bc[fg.ua] * b.mult + du, cnt0 - 1u);
				u32 sg[3];
#pragma unroll
				for (u32 pass = 0; pass < 3; ++pass) { // all (at most three) summaries of a lane are requested together
					const u32 dv = (((u32)tid >> 4) & 3u) + 4u * pass;
					const u32 qv = min(bc[fg.va] * b.mult + dv, cnt0 - 1u);
					int q[3];
					face_scatter(fg, (int)qu, (int)qv, (int)qa, q);
					const u32 id = block_coord_id((u32)q[0], (u32)q[1], (u32)q[2], cnt0);
					const u32 word = p.G.blockSign[id]; // (the coordinates are clamped: every lane reads a valid entry)
					sg[pass] = (du < m1 && dv < m1) ? (word >> (2u * field)) & 3u : 3u; // 3: outside the face
				}
				u32 seen = 0; // bit 0: a plane of samples >= 0, bit 1: a plane of samples < 0, bit 2: a mixed or unknown plane
#pragma unroll
				for (u32 pass = 0; pass < 3; ++pass) seen |= sg[pass] == 1u ? 1u : (sg[pass] == 2u ? 2u : (sg[pass] == 3u ? 0u : 4u));
				const bool pos = __ballot((seen & 1u) != 0) != 0, neg = __ballot((seen & 2u) != 0) != 0, mixed = __ballot((seen & 4u) != 0) != 0;
				if (!mixed && !(pos && neg)) quiet |= 1u << f;
			}
			if ((tid & 63) == 0 && quiet) atomicOr(&quietFaces[quietParity], quiet);
		}
		__syncthreads();
		TRB_TICK(1);
		on &= ~quietFaces[quietParity];
		if (tid == 0) quietFaces[quietParity ^ 1u] = 0; // the other word is next written behind this barrier and read behind the next item's
		quietParity ^= 1u;
		if (tid == 0) st.faceOn = on;
		for (int w = tid; w < 48; w += WG) st.ntAll[w] = 0;
		if (preMat) {
#pragma unroll
			for (int f = 0; f < 6; ++f) st.faceMat[f * 256 + tid] = fm[f];
		}
		if (haveLattice) {
			tr_planes_store(rowLo, rowHi, rowFar, xFace, lat.xp != nullptr, on, tid, st);
		} else {
			// no resident lattice for these planes: sample by sample from the grid's mirror; 33 x 33 samples per face, three
			// faces (15 loads per lane) in flight together (a face that is off - uniform over the workgroup - is neither
			// requested nor stored: nothing reads its plane)
			i8 v[3][5];
			if (on & 1u) tr_face_request<0>(p.G.grid, b, on, tid, v[0]);
			if (on & 2u) tr_face_request<1>(p.G.grid, b, on, tid, v[1]);
			if (on & 4u) tr_face_request<2>(p.G.grid, b, on, tid, v[2]);
			if (on & 1u) tr_face_store(st.plane[0], tid, v[0], true);
			if (on & 2u) tr_face_store(st.plane[1], tid, v[1], true);
			if (on & 4u) tr_face_store(st.plane[2], tid, v[2], true);
			if (on & 8u) tr_face_request<3>(p.G.grid, b, on, tid, v[0]);
			if (on & 16u) tr_face_request<4>(p.G.grid, b, on, tid, v[1]);
			if (on & 32u) tr_face_request<5>(p.G.grid, b, on, tid, v[2]);
			if (on & 8u) tr_face_store(st.plane[3], tid, v[0], true);
			if (on & 16u) tr_face_store(st.plane[4], tid, v[1], true);
			if (on & 32u) tr_face_store(st.plane[5], tid, v[2], true);
		}
	}
	__syncthreads();
	TRB_TICK(2);
	tr_phase_classify(st, tid, WG);
	__syncthreads();
	if (VX_ABL & 2048) { for (int w = tid; w < 48; w += WG) st.ntAll[w] = 0; __syncthreads(); }
	TRB_TICK(3);
	for (int f0 = 0; f0 < 6;) {
		const int f1 = tr_batch_end(st, f0); // uniform
		__syncthreads();
		tr_phase_batch_bits(st, f0, f1, tid, WG);
		if (tid == 0) { st.vTotal = st.iTotal = st.vOff = st.iOff = 0; }
		__syncthreads();
		{
			const u32 nt = block_exclusive_scan_u16(st.wordPrefix, 48, scanScratch);
			if (tid == 0) st.wordPrefix[48] = (u16)nt;
		}
		__syncthreads();
		TRB_TICK(4);
		if (st.wordPrefix[48] != 0) {
			tr_phase_cells_of(st, tid, WG); // (the compact list needs no material: formed in front of the wait, one barrier for both)
			if (GATED && !matReady) {
				// the block's material cache (tr_phase_list reads the cells behind the faces) comes from another workgroup of this launch
				if (tid == 0) (void)wait_done(L.matDone + b.slot, p.G.epoch, p.G.giveUp);
				acquire_and_meet(tid < 64);
				matReady = true;
			} else
				__syncthreads();
			TRB_TICK(5);
			tr_phase_list(st, T, L, b, tid, WG, preMat ? st.faceMat : nullptr);
			__syncthreads();
			TRB_TICK(6);
			tr_phase_count(st, T, tid, WG);
			__syncthreads();
			TRB_TICK(7);
			{
				// the reservation is requested here and first looked at behind the descriptors of the first chunk (which need
				// none of it): one round trip off the block's chain
				const u32 vt = block_exclusive_scan_u16(st.vbase, st.wordPrefix[48], scanScratch);
				const u32 it2 = block_exclusive_scan_u16(st.ibase, st.wordPrefix[48], scanScratch);
				u32 resV = 0, resI = 0;
				if (tid == 0) {
					st.vTotal = vt; st.iTotal = it2;
					reserve_both(p.P.cursors, vt, it2, resV, resI);
				}
				tr_phase_describe(st, 0, tid, WG);
				if (tid == 0) { st.vOff = resV; st.iOff = resI; }
			}
			__syncthreads();
			TRB_TICK(8);
			for (u32 chunk = 0; chunk == 0 || chunk < st.vTotal; chunk += VDESC_CAP) {
				if (chunk) {
					__syncthreads();
					tr_phase_describe(st, chunk, tid, WG);
					__syncthreads();
				}
				if (!(VX_ABL & 512)) tr_phase_emit_vertices(st, T, p.G, smp, p.P, b, chunk, tid, WG);
			}
			TRB_TICK(9);
			for (u32 chunk = 0; chunk < ((VX_ABL & 1024) ? 0u : st.iTotal); chunk += TR_INDEX_CHUNK) {
				__syncthreads();
				tr_phase_stage_indices(st, T, chunk, tid, WG);
				__syncthreads();
				tr_phase_flush_indices(st, T, p.P, chunk, tid, WG);
			}
		}
		TRB_TICK(10);
		tr_phase_record(st, L, b, p.P, f0, f1, tid);
		f0 = f1;
	}
	__syncthreads();
}

template <bool WIDE>
__global__ __launch_bounds__(WG) __attribute__((amdgpu_waves_per_eu(WIDE ? 3 : VX_TR_WAVES))) void k_transition(ExecParamsDev p, u32 levels)
{
	u8* tab = smem;
	TrState& st = *(TrState*)(smem + TR_TAB_LDS);
	__shared__ WorkList wl;
	__shared__ u32 scanScratch[8];
	__shared__ u32 quietFaces[2];
	u32 quietParity = 0;

	// (requested first: the copy is in flight while thread 0 fetches the level counts)
	const Tables T = stage_transition_tables(tab, p.tables, threadIdx.x); // visible after the first barrier of the item loop
	if (threadIdx.x < 2) quietFaces[threadIdx.x] = 0;
	// Full runs: the work items are the leading entries of the run's list of active blocks of the levels >= 1
	// (Globals::flatItems, level order) - the levels with transition cells are 1 .. refLevels - 2.  Incremental runs walk
	// their per-level work lists.
	u32 total = 0;
	if (p.G.dirty) { // (uniform)
		if (threadIdx.x == 0) {
			u32 run = 0;
			for (u32 l = 0; l < MAX_LEVELS; ++l) {
				wl.start[l] = run;
				if (l < levels && p.levels[l].hasTransitions) run += p.G.workCount[l];
			}
			wl.start[MAX_LEVELS] = run;
		}
		__syncthreads();
		total = wl.start[MAX_LEVELS];
	} else {
		for (u32 l = 1; l < levels; ++l) if (p.levels[l].hasTransitions) total += p.G.slotCounts[l];
	}
	total = __builtin_amdgcn_readfirstlane(total);
	if (blockIdx.x >= ((total + 63u) & ~63u)) return;
	const int tid0 = threadIdx.x;
	const GridView& gv = p.G.grid;
	const BrickSamplerT<typename std::conditional<WIDE, size_t, u32>::type> smp = { gv.bDist, gv.bMat, gv.bBlend, gv.n - 1, (u32)gv.n >> 4, (u32)gv.bRowsY, gv.bYb0, gv.bZb0 };

	for (u32 it = blockIdx.x; it < ((total + 63u) & ~63u); it += gridDim.x) {
		const u32 item = xcd_item(it);
		if (item >= total) continue;
		// Opaque per iteration: everything a lane derives from its index alone (row and sample coordinates, addresses in the
		// LDS state ...) is the same for every item, and the compiler would compute it all once in front of this loop and
		// keep it in registers through every phase - dozens of them, in a kernel at its register limit, for a loop that
		// nearly always runs once.
		int tid = tid0;
		asm volatile("" : "+v"(tid));
		RegBlockCtx b;
		u32 coordId;
		if (p.G.dirty) { // (uniform)
			b.level = 0;
			for (u32 l = 1; l < MAX_LEVELS; ++l) if (item >= wl.start[l] && wl.start[l + 1] > wl.start[l]) b.level = l;
			b.slot = p.G.workItems[b.level][item - wl.start[b.level]];
			coordId = p.levels[b.level].slotCoord[b.slot];
		} else {
			const FlatItem fi = p.G.flatItems[item];
			b.level = fi.where >> 24; b.slot = fi.where & 0xFFFFFFu;
			coordId = fi.coordId;
		}
		b.level = __builtin_amdgcn_readfirstlane(b.level); b.slot = __builtin_amdgcn_readfirstlane(b.slot);
		tr_block<WIDE, false>(p, b, coordId, st, T, scanScratch, quietFaces, quietParity, smp, tid);
	}
}


} // namespace

#include "vx_main.inl"

namespace {

// ------------------------------------------------------------------------------------------------------
// incremental (Modification) runs: classify only the dirty level-0 blocks, list the slots to rebuild
// ------------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(WG) void k_classify_blocks(ExecParamsDev p, const u32* coords, u32 count)
{
	__shared__ MatState st; // samp + ntBits are all that is used here
	__shared__ u32 ntCells;
	const LevelDesc& L = p.levels[0];
	const int tid = threadIdx.x;
	for (u32 k = blockIdx.x; k < count; k += gridDim.x) {
		u32 bx, by, bz;
		block_coords(coords[k], L.cnt, bx, by, bz);
		__syncthreads();
		gpu_stage_samples17(p.G.grid, bx, by, bz, 1, st.samp);
		for (int w = tid; w < 128; w += WG) st.ntBits[w] = 0;
		if (tid == 0) ntCells = 0;
		__syncthreads();
		mat_phase_classify(st, tid, WG);
		// whatever a full run knew about the block without reading it no longer holds
		if (tid == 0) p.G.blockClass[coords[k]] = 0;
		__syncthreads();
		if (tid < 128) atomicAdd(&ntCells, (u32)__popc(st.ntBits[tid]));
		__syncthreads();
		if (tid == 0) publish_level0_block(p.G, L, bx, by, bz, st.ntBits, ntCells, true);
	}
}

// ------------------------------------------------------------------------------------------------------
// k_list_write: the result's block lists (ListedBlock, tv_block.h) built where the records are.  Coordinate order =
// index order of the block -> slot maps, so the lists are an ordered compaction of them: the passes that write a block's
// record count it into its group of LIST_WG coordinates (LevelDesc::listCounts); every workgroup here adds up the counts
// before it (at most ~1200 of them) and writes its blocks.
// ------------------------------------------------------------------------------------------------------
__device__ __forceinline__ u32 list_level_of(const ListPlan& plan, u32 levels, u32 w)
{
	u32 l = 0;
	for (u32 q = 1; q < levels; ++q) if (w >= plan.wgStart[q]) l = q;
	return l;
}

// (large block ranges: the counts come from a launch of their own - an atomic per block record in the regular passes
// cost the level-0 pass more there than this launch costs the tail)
__global__ __launch_bounds__(LIST_WG) void k_list_count(ExecParamsDev p, ListPlan plan, u32 levels)
{
	const u32 w = blockIdx.x, l = list_level_of(plan, levels, w);
	const u32 id = (w - plan.wgStart[l]) * LIST_WG + threadIdx.x;
	const int n = __syncthreads_count(listed_block_slot(p.levels[l], id) >= 0 ? 1 : 0);
	if (threadIdx.x == 0) plan.counts[w] = (u32)n;
}

// publish: the workgroup that finishes last copies the run's header (counters, totals, statistics; the block-class partial
// sums behind it) into page-locked host memory - the host then needs no copy behind the run's last kernel (a blit kernel of
// its own: ~3.5 us, and ~5.5 us until it starts), only the wait it does anyway.
struct HeaderPublish {
	u32* done;         // counter of finished workgroups (a header word: zeroed with the run's counters)
	u32* host;         // page-locked destination (nullptr: no publication)
	const u32* dev;    // the header
	u32 words;
};

// (workgroup w of listWgs: a launch of its own, or the workgroups of k_tail behind the general passes)
__device__ __forceinline__ void list_write_pass(const ExecParamsDev& p, const ListPlan& plan, u32 levels, const HeaderPublish& pub, const u32 w, const u32 listWgs, const bool countsThrough)
{
	__shared__ u32 waveSum[LIST_WG / 64];
	__shared__ u32 baseShared;
	const u32 l = list_level_of(plan, levels, w), tid = threadIdx.x;
	const LevelDesc& L = p.levels[l];
	const u32 id = (w - plan.wgStart[l]) * LIST_WG + tid;
	// (countsThrough = inside k_tail: records and counts may come from the general workgroups of this very launch, on another
	// XCD - every read of them goes past the caches that no other XCD's store refreshes)
	int slot = listed_block_slot(L, id);
	if (countsThrough) {
		slot = id < L.cnt * L.cnt * L.cnt ? L.slotOf[id] : -1;
		if (slot >= 0 && TV_LOAD_THROUGH(&L.records[slot].vCount) == 0u) slot = -1;
	}
	// listed blocks of this level in the workgroups before this one
	u32 before = 0;
	for (u32 q = plan.wgStart[l] + tid; q < w; q += LIST_WG) before += countsThrough ? TV_LOAD_THROUGH(plan.counts + q) : plan.counts[q];
	for (int off = 32; off > 0; off >>= 1) before += __shfl_down(before, off, 64);
	if ((tid & 63) == 0) waveSum[tid >> 6] = before;
	__syncthreads();
	if (tid == 0) { u32 b = 0; for (u32 q = 0; q < LIST_WG / 64; ++q) b += waveSum[q]; baseShared = b; }
	__syncthreads();
	const u32 base = baseShared;
	// rank of this block among the workgroup's listed ones
	const unsigned long long mask = __ballot(slot >= 0);
	__syncthreads();
	if ((tid & 63) == 0) waveSum[tid >> 6] = (u32)__popcll(mask);
	__syncthreads();
	u32 rank = (u32)__popcll(mask & ((1ull << (tid & 63)) - 1ull));
	for (u32 q = 0; q < (tid >> 6); ++q) rank += waveSum[q];
	if (slot >= 0) {
		ListedBlock& out = L.listed[base + rank];
		listed_block_fill(out, L, id, (u32)slot, plan.idBase[l]);
		if (countsThrough) {
			static_assert(sizeof(BlockRecord) == 128 && offsetof(ListedBlock, rec) == 0, "a record is eight 16-byte pieces at the head of a listed block");
#pragma unroll
			for (u32 q = 0; q < 8; ++q) { // (list entries are 156 bytes apart: dword stores)
				const uint4 v = load16_through(L.records, (u32)slot * (u32)sizeof(BlockRecord) + q * 16u); // (the base is the level's, uniform: it travels in scalar registers)
				u32* dst = (u32*)&out.rec + q * 4u;
				dst[0] = v.x; dst[1] = v.y; dst[2] = v.z; dst[3] = v.w;
			}
		}
	}
	// The header is complete when every level's total is written (everything else in it was final before the list pass):
	// the workgroups that write a total count themselves - one returning atomic per level, not per workgroup, which on a
	// 1024^3 grid were 1170 serialised round trips - and the last of them publishes.
	const bool levelFinal = w + 1 == plan.wgStart[l + 1];
	if (levelFinal && tid == 0) {
		u32 total = base;
		for (u32 q = 0; q < LIST_WG / 64; ++q) total += waveSum[q];
		__hip_atomic_store(&plan.totals[l], total, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); // (read by the publishing workgroup, possibly on another XCD)
		if (pub.host) {
			asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); // the total is on its way out before this level counts as finished
			baseShared = atomicAdd(pub.done, 1u) + 1u == levels ? 1u : 0u;
		}
	}
	if (pub.host && levelFinal) {
		__syncthreads();
		if (baseShared) for (u32 i = tid; i < pub.words; i += LIST_WG) pub.host[i] = TV_LOAD_THROUGH(pub.dev + i);
	}
	(void)listWgs;
}

__global__ __launch_bounds__(LIST_WG) void k_list_write(ExecParamsDev p, ListPlan plan, u32 levels, HeaderPublish pub)
{
	list_write_pass(p, plan, levels, pub, blockIdx.x, gridDim.x, false);
}

// ------------------------------------------------------------------------------------------------------
// k_tail: what follows k_main in a single-stream run, as ONE launch (a launch behind k_main costs ~4.5 us on this chip
// whether it finds work or not - three of them were a tenth of a 128^3 run): the general passes over what the table-driven
// level-0 blocks and the table-driven blocks of the levels >= 1 handed on (blocks with a zero sample: none, or a handful)
// and the block lists.  The list workgroups need every record: they wait until the general workgroups - which wait for
// nobody - have counted themselves done (roles: see the kernel).  A general workgroup that wrote something makes it visible
// device-wide first (records and list counts are read by list workgroups on other XCDs: through write-through loads where
// the list pass reads what this launch wrote).
// ------------------------------------------------------------------------------------------------------
struct TailPlan {
	u32 wgs0, wgs1, listWgs; // workgroups: general pass of level 0 | of the levels >= 1 | lists
	u32 levels;
	u32* slowDone;           // finished general workgroups (a header word: zeroed with the run's counters)
	u32* roleTicket;         // (the header word behind it) roles handed out in order of arrival, see below
	ResetRanges next;        // the counters and maps of the OTHER set, which the next run will use: put into their start state here
};

// Roles.  Nothing was handed on (the rule on a terrain; two header words say so): the general workgroups find nothing, the list
// workgroups wait for nobody, and a workgroup's role is its place in the grid.  Something was handed on: the list workgroups
// wait for the general ones - and a wait may only depend on workgroups that are already running, whatever order the hardware
// dispatches them in - so every workgroup draws a ticket and the first `general` tickets are the general passes (one returning
// atomic per workgroup on one address: ~10 us at 1024^3, paid only by runs that met a block with a zero sample).
__global__ __launch_bounds__(WG) __attribute__((amdgpu_waves_per_eu(3))) void k_tail(ExecParamsDev p, ListPlan plan, HeaderPublish pub, TailPlan t)
{
	static_assert(LIST_WG == WG, "one workgroup shape for the passes of k_tail");
	__shared__ u32 roleSh[2];
	const u32 general = t.wgs0 + t.wgs1;
	if (threadIdx.x == 0) {
		const u32 handed = TV_LOAD_THROUGH(p.G.slowCount) + TV_LOAD_THROUGH(p.G.slowCount + 1); // (blocks handed on, both passes)
		roleSh[0] = handed;
		roleSh[1] = handed ? atomicAdd(t.roleTicket, 1u) : blockIdx.x;
	}
	__syncthreads();
	const bool handedOn = r0_uniform(roleSh[0]) != 0u;
	const u32 role = r0_uniform(roleSh[1]);
	if (role < general) {
		const bool wrote = role < t.wgs0 ? regular0_pass<REG_CAP_SMALL, 2>(p, 0u, role, t.wgs0)
		                                 : regular_pass<REG_CAP_SMALL, 2>(p, 1u, t.levels, 0u, role - t.wgs0, t.wgs1);
		if (wrote) __threadfence();
		__syncthreads();
		if (threadIdx.x == 0) atomicAdd(t.slowDone, 1u);
		return;
	}
	if (threadIdx.x == 0 && handedOn) {
		// Patience in proportion to the work handed on: the general workgroups were sized by the run before (two per pass when
		// it handed on nothing), and a grid whose every surface block holds a zero sample - a height map's - can leave them
		// thousands of blocks at 50-100 us each.  (A fixed bound of 2^17 polls gave up on exactly that run now and then.)
		const u32 patience = (u32)WAIT_SPINS + (r0_uniform(roleSh[0]) / max(general, 1u) + 1u) * 4096u;
		u32 spins = 0;
		while (__hip_atomic_load(t.slowDone, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) < general) {
			__builtin_amdgcn_s_sleep(4);
			if (++spins > patience) { atomicOr(p.G.giveUp, 1u); break; } // (the host fails the run)
		}
	}
	__syncthreads();
	// (records and counts written by the general workgroups of this very launch are read past the caches; when nothing was handed
	// on, everything the list pass reads comes from earlier launches)
	list_write_pass(p, plan, t.levels, pub, role - general, t.listWgs, handedOn);
	if (t.next.header) {
		const u32 lanes = t.listWgs * WG;
		for (u32 i = (role - general) * WG + threadIdx.x; i * 4u < t.next.start[MAX_LEVELS] || i < t.next.listWgs; i += lanes) {
			reset_words(t.next, i);
			if (i < t.next.listWgs) t.next.listCounts[i] = 0;
		}
	}
}

// ------------------------------------------------------------------------------------------------------
// Incremental runs as three launches (TransVoxelRun::Execute with a Modification, src/TransVoxelImpl.cpp:385-466, :468-538):
//   k_dirty_head   one workgroup per level-0 block of the dirty box (:429-465: the touched blocks plus one ring): its bitmap
//                  from the sign masks of its 17 x 17 sample rows, slot (a new one when the block turns active: then its
//                  ancestors are activated as k_hierarchy does), accumulated consistency bits (:757: only ever set).  The
//                  workgroup that finishes last puts the run's counters into their start state (pool cursors behind what the
//                  pools hold) and lists, per level, the active slots among the box's blocks (Globals::workItems).
//   k_main<true>   the two queues over those lists (vx_main.inl)
//   k_dirty_tail   the general passes over what k_main handed on, then the rebuilt blocks' records and the header straight
//                  into page-locked host memory: the host's one wait is the kernel's completion.
// The box travels as launch arguments; nothing is uploaded, nothing is copied back behind the kernels.
// ------------------------------------------------------------------------------------------------------
struct DirtyPlan {
	u32 lo[MAX_LEVELS][3], hi[MAX_LEVELS][3]; // per level: the dirty box in block coordinates (internal axes x, y, z), [lo, hi)
	u32 start[MAX_LEVELS + 1];                // per level: its segment of the work array (start[l + 1] - start[l] = blocks in the box)
	u32 levels;
	u32* work;                                // [start[levels]] out: per level the active slots among the box's blocks
	u32* info;                                // [start[1]] scratch, per dirty level-0 block: bit 0 = not skipped by the emptiness rule, bit 1 = beyond the first capacity class
	u32* ticket;                              // finished workgroups of this kernel, counted over all its launches (never reset)
	u32 ticketTarget;                         // ... the value the last workgroup of this launch brings it to
	u32* header;                              // the run's counters: words [resetFrom, resetTo) start at zero,
	u32 resetFrom, resetTo;
	u32 poolVerts, poolIdx;                   // ... the pool cursors behind what the pools hold
};

__global__ __launch_bounds__(WG) void k_dirty_head(ExecParamsDev p, DirtyPlan d)
{
	__shared__ u32 rowMask[292];
	__shared__ u32 sh[8];
	const LevelDesc& L = p.levels[0];
	const GridView& g = p.G.grid;
	const int tid = (int)threadIdx.x, n = g.n;
	{
		const u32 k = blockIdx.x;
		const u32 dx = d.hi[0][0] - d.lo[0][0], dy = d.hi[0][1] - d.lo[0][1];
		const u32 bx = d.lo[0][0] + k % dx, by = d.lo[0][1] + (k / dx) % dy, bz = d.lo[0][2] + k / (dx * dy);
		const u32 id = block_coord_id(bx, by, bz, L.cnt);
		// sign masks of the 17 x 17 sample rows the block's cells read (clamped at the grid's far side like every fetch, :1194-1201)
		uint4 lo[2];
		i8 far[2];
#pragma unroll
		for (int q = 0; q < 2; ++q) {
			const int r = min(tid + q * WG, 288), kk = r / 17, j = r - kk * 17;
			const int y = min((int)(by * 16) + j, n - 1), z = min((int)(bz * 16) + kk, n - 1);
			lo[q] = *(const uint4*)(g.bDist + brick_offset(g, (int)(bx * 16), y, z));
			far[q] = g.bDist[brick_offset(g, min((int)(bx * 16) + 16, n - 1), y, z)];
		}
		// the emptiness rule (:1511-1527): the 27 flags around the block
		u32 flag = 1u;
		if (tid < 27) {
			const u32 cx = (u32)clampi((int)bx + (tid % 3) - 1, 0, (int)L.cnt - 1), cy = (u32)clampi((int)by + ((tid / 3) % 3) - 1, 0, (int)L.cnt - 1), cz = (u32)clampi((int)bz + (tid / 9) - 1, 0, (int)L.cnt - 1);
			flag = p.G.emptyFlags[block_coord_id(cx, cy, cz, L.cnt)] ? 1u : 0u;
		}
		int slot = L.slotOf[id]; // (level 0's entries are only ever written by the block's own workgroup)
#pragma unroll
		for (int q = 0; q < 2; ++q) {
			const int r = tid + q * WG;
			if (r < 289) rowMask[r] = sign_nibble(lo[q].x) | (sign_nibble(lo[q].y) << 4) | (sign_nibble(lo[q].z) << 8) | (sign_nibble(lo[q].w) << 12) | ((((u32)far[q] >> 7) & 1u) << 16);
		}
		if (tid < 8) sh[tid] = 0;
		__syncthreads();
		const bool skipped = __syncthreads_and(flag != 0u) != 0;
		const int y = tid & 15, z = tid >> 4;
		const u32 a = rowMask[z * 17 + y], b2 = rowMask[z * 17 + y + 1], c = rowMask[(z + 1) * 17 + y], e = rowMask[(z + 1) * 17 + y + 1];
		const u32 A = a & b2 & c & e, O = a | b2 | c | e;
		const u32 nt = ((O | (O >> 1)) & ~(A & (A >> 1))) & 0xFFFFu;
		u32 cnt = (u32)__popc(nt);
#pragma unroll
		for (int o = 32; o > 0; o >>= 1) cnt += (u32)__shfl_xor((int)cnt, o, 64);
		if ((tid & 63) == 0 && cnt) atomicAdd(&sh[0], cnt);
		__syncthreads();
		const u32 cells = sh[0];
		bool fresh = false;
		if (slot < 0 && cells) {
			if (tid == 0) {
				const u32 s = atomicAdd(L.nActive, 1u);
				L.slotOf[id] = (int)s;
				L.slotCoord[s] = id;
				sh[1] = s;
				// a block that turns active: its ancestors become active (k_hierarchy)
				for (u32 l = 1; l < d.levels; ++l) {
					const LevelDesc& U = p.levels[l];
					const u32 px = bx >> l, py = by >> l, pz = bz >> l;
					if (px >= U.cnt || py >= U.cnt || pz >= U.cnt) break;
					const u32 uid = block_coord_id(px, py, pz, U.cnt);
					if (atomicCAS(&U.slotOf[uid], -1, -2) != -1) break; // somebody else owns this ancestor chain
					const u32 us = atomicAdd(U.nActive, 1u);
					U.slotCoord[us] = uid;
					__hip_atomic_store(&U.slotOf[uid], (int)us, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
				}
			}
			__syncthreads();
			slot = (int)sh[1];
			fresh = true;
		}
		if (slot >= 0) {
			// (publish_level0_block of tv_block.h, one cell row per lane)
			if (tid == 0) { L.skip[slot] = skipped ? 1 : 0; L.ntCount[slot] = (u16)cells; }
			((u16*)(L.ntBits + (size_t)slot * 128))[tid] = (u16)nt;
			u16* cons = (u16*)(L.consBits + (size_t)slot * 128) + tid;
			const u32 add = skipped ? 0u : nt;
			*cons = (u16)(fresh ? add : ((u32)*cons | add));
		}
		if (tid == 0) {
			p.G.blockClass[id] = 0; // whatever a full run knew about the block without reading it no longer holds
			d.info[k] = (skipped ? 0u : 1u) | ((slot >= 0 && cells > (u32)LARGE_THRESHOLD) ? 2u : 0u);
		}
	}
	// ---- the workgroup that finishes last: counters, work lists --------------------------------------------------------------
	__threadfence();
	__syncthreads();
	if (tid == 0) sh[2] = atomicAdd(d.ticket, 1u) + 1u == d.ticketTarget ? 1u : 0u;
	__syncthreads();
	if (!sh[2]) return;
	__threadfence();
	for (u32 i = d.resetFrom + (u32)tid; i < d.resetTo; i += WG) d.header[i] = 0;
	__syncthreads();
	u32 notSkipped = 0, large = 0;
	for (u32 i = (u32)tid; i < d.start[1]; i += WG) { const u32 v = TV_LOAD_THROUGH(d.info + i); notSkipped += v & 1u; large += (v >> 1) & 1u; }
	{
		// (two sums in words of their own: a dirty box of a 1024^3 grid can hold more than 65 535 blocks)
#pragma unroll
		for (int o = 32; o > 0; o >>= 1) { notSkipped += (u32)__shfl_xor((int)notSkipped, o, 64); large += (u32)__shfl_xor((int)large, o, 64); }
		if ((tid & 63) == 0) { atomicAdd(&sh[3], notSkipped); atomicAdd(&sh[4], large); }
	}
	__syncthreads();
	if (tid == 0) {
		p.P.cursors[CUR_V] = d.poolVerts; p.P.cursors[CUR_I] = d.poolIdx;
		p.G.stats[2] = sh[3];
		*p.G.largeBlocks = sh[4];     // (the material blocks of k_main<true> add the levels above)
		p.G.largeBlocks[2] = sh[4];   // level 0 alone: the host announces the two kinds apart (run_dirty_fused)
	}
	// the work lists: a wave per level (levels beyond the fourth: a second turn), 64 coordinates of the level's box per step, four
	// steps' slots requested together; the list keeps the box's coordinate order
	{
		const u32 lane = (u32)tid & 63u;
		for (u32 l = (u32)tid >> 6; l < d.levels; l += (u32)(WG / 64)) {
			const LevelDesc& U = p.levels[l];
			const u32 V = d.start[l + 1] - d.start[l], dx = d.hi[l][0] - d.lo[l][0], dy = d.hi[l][1] - d.lo[l][1];
			u32 count = 0;
			for (u32 base = 0; base < V; base += 256u) {
				int slot[4];
#pragma unroll
				for (u32 q = 0; q < 4; ++q) {
					const u32 i = min(base + q * 64u + lane, V - 1u);
					slot[q] = TV_LOAD_THROUGH(&U.slotOf[block_coord_id(d.lo[l][0] + i % dx, d.lo[l][1] + (i / dx) % dy, d.lo[l][2] + i / (dx * dy), U.cnt)]);
				}
#pragma unroll
				for (u32 q = 0; q < 4; ++q) {
					const bool have = base + q * 64u + lane < V && slot[q] >= 0;
					const unsigned long long m = __ballot(have);
					if (have) d.work[d.start[l] + count + (u32)__popcll(m & ((1ull << lane) - 1ull))] = (u32)slot[q];
					count += (u32)__popcll(m);
				}
			}
			if (lane == 0) p.G.workCount[l] = count;
		}
	}
}

struct DirtyTailPlan {
	u32 wgs0, wgs1, gatherWgs; // workgroups: general pass of level 0 | of the levels >= 1 | record gather
	u32 levels;
	u32* roleTicket;           // header words (zero at launch): roles are handed out in the order the workgroups arrive,
	u32* slowDone;             //   finished general workgroups
	u32 start[MAX_LEVELS + 1];
	const u32* work;
	BlockRecord* hostRecs;     // page-locked: [start[levels]] the records of the rebuilt blocks, per level in work-list order
	u32* hostHeader;           // page-locked: the run's header
	const u32* devHeader;
	u32 headerWords, publishedWord;
};

// Roles by ticket, not by blockIdx: the gather workgroups wait for the general ones, and a wait may only depend on workgroups that
// are already running - whoever arrives first takes the general roles, whatever its place in the grid.
__global__ __launch_bounds__(WG) __attribute__((amdgpu_waves_per_eu(3))) void k_dirty_tail(ExecParamsDev p, DirtyTailPlan t)
{
	__shared__ u32 role;
	const u32 general = t.wgs0 + t.wgs1;
	if (threadIdx.x == 0) role = atomicAdd(t.roleTicket, 1u);
	__syncthreads();
	const u32 ticket = r0_uniform(role);
	if (ticket < general) {
		const bool wrote = ticket < t.wgs0 ? regular0_pass<REG_CAP_SMALL, 2>(p, 0u, ticket, t.wgs0)
		                                   : regular_pass<REG_CAP_SMALL, 2>(p, 1u, t.levels, 0u, ticket - t.wgs0, t.wgs1);
		if (wrote) __threadfence();
		__syncthreads();
		if (threadIdx.x == 0) atomicAdd(t.slowDone, 1u);
		return;
	}
	const u32 gw = ticket - general;
	if (threadIdx.x == 0) {
		const u32 handed = TV_LOAD_THROUGH(p.G.slowCount) + TV_LOAD_THROUGH(p.G.slowCount + 1);
		if (handed) {
			const u32 patience = (u32)WAIT_SPINS + (handed / max(general, 1u) + 1u) * 4096u; // (in proportion to the work handed on, see k_tail)
			u32 spins = 0;
			while (__hip_atomic_load(t.slowDone, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) < general) {
				__builtin_amdgcn_s_sleep(4);
				if (++spins > patience) { atomicOr(p.G.giveUp, 1u); break; } // (the host fails the run)
			}
		}
	}
	__syncthreads();
	// one record = 32 dwords: eight records per trip and workgroup
	static_assert(sizeof(BlockRecord) == 128, "a record is 32 dwords");
	const u32 total = t.start[t.levels];
	for (u32 pos = gw * 8u + (threadIdx.x >> 5); pos < total; pos += t.gatherWgs * 8u) {
		u32 l = 0;
		for (u32 q = 1; q < t.levels; ++q) if (pos >= t.start[q]) l = q;
		if (pos - t.start[l] >= TV_LOAD_THROUGH(p.G.workCount + l)) continue;
		const u32 slot = t.work[pos];
		((u32*)(t.hostRecs + pos))[threadIdx.x & 31u] = TV_LOAD_THROUGH((const u32*)(p.levels[l].records + slot) + (threadIdx.x & 31u));
	}
	if (gw == 0) {
		for (u32 i = threadIdx.x; i < t.headerWords; i += WG) if (i != t.publishedWord) t.hostHeader[i] = TV_LOAD_THROUGH(t.devHeader + i);
		__syncthreads();
		if (threadIdx.x == 0) { __threadfence_system(); t.hostHeader[t.publishedWord] = 1u; }
	}
}

// ------------------------------------------------------------------------------------------------------
// k_halo_move: the pieces of the halo messages of one side pair (below / above) between their fields and contiguous
// staging buffers (HaloMove, tv_block.h): blockIdx.y picks the message, one workgroup per row (n bytes of a voxel field,
// cnt bytes of the flag array).  Unpacking also writes the rows into the brick mirrors (tv_core.h GridView) and the lattice
// copies when the view carries them, so an exchange is two launches around the RCCL batch: pack, unpack - and one mirror
// pass over the halo block layers behind the unpack (refresh_halo_layers, vx_host.inl), which brings their sign summaries up
// to date: the summary of the plane that is resident is exact, and k_run_head relies on it.
// ------------------------------------------------------------------------------------------------------
struct HaloPair { HaloMove m[2]; };

__global__ __launch_bounds__(WG) void k_halo_move(HaloPair pair, GridView g, MirrorState X, int alongY)
{
	const HaloMove& mv = pair.m[blockIdx.y];
	u32 row = blockIdx.x;
	u32 pi = 0;
	if (!mv.count) return;
	while (pi + 1 < mv.count && row >= (u32)mv.piece[pi].layers * mv.piece[pi].rows) { row -= (u32)mv.piece[pi].layers * mv.piece[pi].rows; ++pi; }
	const HaloPiece& p = mv.piece[pi];
	if (row >= (u32)p.layers * p.rows) return;
	const int l = (int)(row / p.rows);
	const u32 a = row - (u32)l * p.rows;
	u8* f = p.field + halo_field_offset(p, p.firstLayer + l, a);
	u8* s = mv.staging + p.stagingOffset + (size_t)row * p.rowBytes;
	u8* dst = mv.unpack ? f : s;
	const u8* src = mv.unpack ? s : f;
	// the mirror of this piece's field, if it is a voxel field of the view (the flag array has none)
	u8* brick = nullptr;
	if (mv.unpack && g.bDist && p.rowBytes == (u32)g.n) {
		if (p.field == (const u8*)g.dist) brick = (u8*)const_cast<i8*>(g.bDist);
		else if (p.field == g.mat) brick = const_cast<u8*>(g.bMat);
		else if (p.field == g.blend) brick = const_cast<u8*>(g.bBlend);
	}
	const int layer = p.firstLayer + l;
	const int y = alongY ? layer : (int)a, z = alongY ? (int)a : layer;
	if ((p.rowBytes & 15u) == 0 && (((size_t)dst | (size_t)src) & 15u) == 0) {
		for (u32 i = threadIdx.x; i < p.rowBytes / 16; i += WG) {
			const uint4 v = ((const uint4*)src)[i];
			((uint4*)dst)[i] = v;
			if (brick) {
				*(uint4*)(brick + brick_offset(g, (int)(i * 16), y, z)) = v;
				if (brick == (u8*)g.bDist) lattice_rows_of(X, g.n, (int)(i * 16), y, z, v); // the lattice copies follow as well
			}
		}
	} else {
		for (u32 i = threadIdx.x; i < p.rowBytes; i += WG) {
			dst[i] = src[i];
			if (brick) brick[brick_offset(g, (int)i, y, z)] = src[i];
		}
	}
}

// ------------------------------------------------------------------------------------------------------
// k_selftest: the device forms of the exactness-critical arithmetic, checked exhaustively on the hardware they run on
// (vx_selftest; tests/test_gpu_parity.py).  One lane per gradient (dx, dy, dz) in [0, 255]^3 — central differences
// of int8 samples, up to sign — and, for the first 65536 lanes, per sample pair (v0, v1).
//   out[0]  crossed sample pairs where edge_t_crossing differs from the truncated integer quotient
//   out[1]  gradients whose normalised components differ between g and 0.5 g (tv_fast0.h drops the factor)
//   out[2]  gradients where normalize_fix_zero differs from fp32 sqrtf + IEEE division as hipcc compiles them
//   out[3..10] candidates for cheaper forms (see below): gradients with a component that differs from normalize_fix_zero
//   out[11] gradients where normalize_gradient (the form the fast passes use for end-point normals) differs from it
// ------------------------------------------------------------------------------------------------------
__device__ __forceinline__ float selftest_sqrt_newton(float x)
{
	// rsq + one Newton step with exact residual
	const float y = __builtin_amdgcn_rsqf(x), s = x * y, h = 0.5f * y;
	return __builtin_fmaf(__builtin_fmaf(-s, s, x), h, s);
}

__global__ __launch_bounds__(WG) void k_selftest(u32* out)
{
	const u32 i = blockIdx.x * WG + threadIdx.x;
	if (i < 65536u) {
		const int v0 = (int)(i8)(i & 0xFFu), v1 = (int)(i8)(i >> 8);
		if (v0 * v1 <= 0 && v0 != v1 && edge_t_crossing(v0, v1) != (v1 * 256) / (v1 - v0)) atomicAdd(&out[0], 1u);
	}
	const float g[3] = { (float)(i & 255u), (float)((i >> 8) & 255u), (float)(i >> 16) };
	float a[3] = { g[0], g[1], g[2] }, b[3] = { g[0] * 0.5f, g[1] * 0.5f, g[2] * 0.5f };
	normalize_fix_zero(a);
	normalize_fix_zero(b);
	if (__float_as_uint(a[0]) != __float_as_uint(b[0]) || __float_as_uint(a[1]) != __float_as_uint(b[1]) || __float_as_uint(a[2]) != __float_as_uint(b[2])) atomicAdd(&out[1], 1u);
	const float len2 = (g[0] * g[0] + g[1] * g[1]) + g[2] * g[2];
	{
		const float len = sqrtf(len2);
		float r[3] = { 0.f, 0.f, 0.f };
		if (!(len <= 1.1920929e-07f)) { r[0] = g[0] / len; r[1] = g[1] / len; r[2] = g[2] / len; }
		if (__float_as_uint(a[0]) != __float_as_uint(r[0]) || __float_as_uint(a[1]) != __float_as_uint(r[1]) || __float_as_uint(a[2]) != __float_as_uint(r[2])) atomicAdd(&out[2], 1u);
	}
	if (len2 > 0.f) {
		// exact length as normalize_fix_zero finds it
		float len = __builtin_amdgcn_sqrtf(len2);
		{
			const float down = __builtin_bit_cast(float, __builtin_bit_cast(int, len) - 1), up = __builtin_bit_cast(float, __builtin_bit_cast(int, len) + 1);
			const float rDown = __builtin_fmaf(-down, len, len2), rUp = __builtin_fmaf(-up, len, len2);
			len = (rDown <= 0.f) ? down : len;
			len = (rUp > 0.f) ? up : len;
		}
		const float y0 = __builtin_amdgcn_rcpf(len), y = __builtin_fmaf(__builtin_fmaf(-len, y0, 1.0f), y0, y0);
		u32 bad3 = 0, bad4 = 0, bad5 = 0, bad6 = 0, bad7 = 0;
		const float lenN = selftest_sqrt_newton(len2);
		if (__float_as_uint(lenN) != __float_as_uint(len)) bad4 = 1;                              // [4] length by rsq + Newton
		if (__float_as_uint(__builtin_amdgcn_sqrtf(len2)) != __float_as_uint(len)) bad7 = 1;      // [7] raw v_sqrt_f32
		const float yN0 = __builtin_amdgcn_rcpf(lenN), yN = __builtin_fmaf(__builtin_fmaf(-lenN, yN0, 1.0f), yN0, yN0);
#pragma unroll
		for (int c = 0; c < 3; ++c) {
			const float n = g[c];
			const float q0 = n * y, q1 = __builtin_fmaf(__builtin_fmaf(-len, q0, n), y, q0);
			if (__float_as_uint(q1) != __float_as_uint(a[c])) bad3 = 1;                            // [3] one correction of the quotient instead of two
			const float p0 = n * y0, p1 = __builtin_fmaf(__builtin_fmaf(-len, p0, n), y0, p0), p2 = __builtin_fmaf(__builtin_fmaf(-len, p1, n), y0, p1);
			if (__float_as_uint(p2) != __float_as_uint(a[c])) bad5 = 1;                            // [5] unrefined reciprocal, two corrections
			const float w0 = n * yN, w1 = __builtin_fmaf(__builtin_fmaf(-lenN, w0, n), yN, w0);
			if (__float_as_uint(w1) != __float_as_uint(a[c])) bad6 = 1;                            // [6] Newton length + one correction
		}
		{
			// [8] Newton length, unrefined reciprocal, one correction; [9] the same with the reciprocal taken from rsq(len2)
			const float yR = __builtin_amdgcn_rsqf(len2);
			u32 bad8 = 0, bad9 = 0, bad10 = 0;
#pragma unroll
			for (int c = 0; c < 3; ++c) {
				const float n = g[c];
				const float w0 = n * yN0, w1 = __builtin_fmaf(__builtin_fmaf(-lenN, w0, n), yN0, w0);
				if (__float_as_uint(w1) != __float_as_uint(a[c])) bad8 = 1;
				const float r0 = n * yR, r1 = __builtin_fmaf(__builtin_fmaf(-lenN, r0, n), yR, r0);
				if (__float_as_uint(r1) != __float_as_uint(a[c])) bad9 = 1;
				const float r2 = __builtin_fmaf(__builtin_fmaf(-lenN, r1, n), yR, r1);
				if (__float_as_uint(r2) != __float_as_uint(a[c])) bad10 = 1;                          // [10] [9] with two corrections
			}
			if (bad8) atomicAdd(&out[8], 1u);
			if (bad9) atomicAdd(&out[9], 1u);
			if (bad10) atomicAdd(&out[10], 1u);
			// [11] what the fast passes use: normalize_gradient
			float q[3] = { g[0], g[1], g[2] };
			normalize_gradient(q);
			if (__float_as_uint(q[0]) != __float_as_uint(a[0]) || __float_as_uint(q[1]) != __float_as_uint(a[1]) || __float_as_uint(q[2]) != __float_as_uint(a[2])) atomicAdd(&out[11], 1u);
		}
		if (bad3) atomicAdd(&out[3], 1u);
		if (bad4) atomicAdd(&out[4], 1u);
		if (bad5) atomicAdd(&out[5], 1u);
		if (bad6) atomicAdd(&out[6], 1u);
		if (bad7) atomicAdd(&out[7], 1u);
	}
}

// RCCL through its C API, bound at run time: a process that never shards a grid does not load the library
struct Rccl {
	typedef struct { char internal[128]; } UniqueId;
	void* lib = nullptr;
	int (*GetUniqueId)(UniqueId*) = nullptr;
	int (*CommInitRank)(void**, int, UniqueId, int) = nullptr;
	int (*CommDestroy)(void*) = nullptr;
	int (*GroupStart)() = nullptr;
	int (*GroupEnd)() = nullptr;
	int (*Send)(const void*, size_t, int, int, void*, hipStream_t) = nullptr;
	int (*Recv)(void*, size_t, int, int, void*, hipStream_t) = nullptr;
	const char* (*GetErrorString)(int) = nullptr;
	bool load(std::string& err)
	{
		if (lib) return true;
		lib = dlopen("librccl.so.1", RTLD_NOW | RTLD_LOCAL);
		if (!lib) lib = dlopen("librccl.so", RTLD_NOW | RTLD_LOCAL);
		if (!lib) { err = std::string("librccl.so not found: ") + dlerror(); return false; }
		bool ok = true;
		auto sym = [&](const char* name) { void* p = dlsym(lib, name); if (!p) { ok = false; err = std::string("missing RCCL symbol ") + name; } return p; };
		GetUniqueId = (int (*)(UniqueId*))sym("ncclGetUniqueId");
		CommInitRank = (int (*)(void**, int, UniqueId, int))sym("ncclCommInitRank");
		CommDestroy = (int (*)(void*))sym("ncclCommDestroy");
		GroupStart = (int (*)())sym("ncclGroupStart");
		GroupEnd = (int (*)())sym("ncclGroupEnd");
		Send = (int (*)(const void*, size_t, int, int, void*, hipStream_t))sym("ncclSend");
		Recv = (int (*)(void*, size_t, int, int, void*, hipStream_t))sym("ncclRecv");
		GetErrorString = (const char* (*)(int))sym("ncclGetErrorString");
		if (!ok) { dlclose(lib); lib = nullptr; }
		return ok;
	}
};
Rccl& rccl() { static Rccl r; return r; }

struct DirtyRanges { u32 start[MAX_LEVELS + 1]; };

__global__ __launch_bounds__(WG) void k_build_worklist(ExecParamsDev p, const u32* coords, DirtyRanges r, u32 levels, u32* work)
{
	const u32 k = blockIdx.x * WG + threadIdx.x;
	if (k >= r.start[levels]) return;
	u32 l = 0;
	for (u32 i = 1; i < levels; ++i) if (k >= r.start[i]) l = i;
	const int slot = p.levels[l].slotOf[coords[k]];
	if (slot >= 0) work[r.start[l] + atomicAdd(&p.G.workCount[l], 1u)] = (u32)slot;
}

__global__ __launch_bounds__(WG) void k_gather_records(ExecParamsDev p, DirtyRanges r, u32 levels, BlockRecord* out)
{
	const u32 k = blockIdx.x * WG + threadIdx.x;
	if (k >= r.start[levels]) return;
	u32 l = 0;
	for (u32 i = 1; i < levels; ++i) if (k >= r.start[i]) l = i;
	const u32 i = k - r.start[l];
	if (i < p.G.workCount[l]) out[k] = p.levels[l].records[p.G.workItems[l][i]];
}

// ------------------------------------------------------------------------------------------------------
// HIP backend
// ------------------------------------------------------------------------------------------------------
struct Backend {
	hipStream_t ownStream = nullptr, stream = nullptr;
	hipStream_t sideA = nullptr, sideB = nullptr;      // level-0 regular pass / transition pass run beside the material chain
	hipStream_t sideC = nullptr;                       // dense surfaces: the capacity classes above the first of level 0 (few workgroups per CU, long-running: beside the first class, not behind it); those of the levels >= 1 follow the transition pass on side stream B (a fifth stream would share a hardware queue with this one and wait behind it)
	hipEvent_t evMidC = nullptr, evMidD = nullptr, evSideC = nullptr, evSideD = nullptr;
	bool spreadC = false, spreadD = false;           // this run queued work on side stream C / D
	bool overlappedTail = false;                     // inside run_overlapped_tail
	hipEvent_t evClassified = nullptr, evMaterial = nullptr, evSideA = nullptr, evSideB = nullptr, evMain = nullptr;
	hipStream_t mainKeep = nullptr; // set while the tail of an overlapped run is queued on side stream A
	hipStream_t copyStream[4] = { nullptr, nullptr, nullptr, nullptr }; // d2h_bulk
	hipEvent_t evCopy = nullptr;
	hipEvent_t ev0 = nullptr, ev1 = nullptr;
	hipEvent_t evPrevEnd = nullptr; // the end of the previous timed run (idle_before_ms: diagnostics, VX_HOST_TIMING)
	bool havePrevEnd = false;
	hipEvent_t stageEv[9] = { nullptr, nullptr, nullptr, nullptr, nullptr, nullptr, nullptr, nullptr, nullptr }; // [8]: between the level-0 and the level >= 1 regular pass
	bool stageOn = false, stageValid = false;
	std::string lastError;
	int cus = 256;
	int device = 0;
	bool ok = true;
	// launch geometry knobs, read from the environment once when the context is created (tuning aids)
	// Runtime knobs (read once per context).  Seven in all with VX_POOL_SLACK and VX_HOST_TIMING (vx_host.inl); each selects a path
	// that production runs reach through their data - dense surfaces, grids beyond 1024^3, blocks with zero samples - so that the
	// tests can drive those paths on small fixtures (tests/test_gpu_parity.py::test_hip_runtime_knobs_select_equivalent_paths).
	struct Tuning {
		u32 fast = 3;        // VX_FAST: bit 0 = table-driven pass on level 0, bit 1 = on the levels >= 1 (0: every block through the general passes)
		u32 forceWide = 0;   // VX_FORCE_WIDE=1: the 64-bit-offset kernel variants (grids beyond 1024^3) on small grids too
		u32 upper = 1;       // VX_UPPER=0: the chain of launches (what dense surfaces run) instead of k_main
		u32 selfHead = 1;    // VX_SELF_HEAD=0: a classification pass (k_classify, k_hierarchy) instead of k_run_head handing out the slots
		u32 dirtyFused = 1;  // VX_DIRTY_FUSED=0: incremental runs as the chain of launches with work lists
		// fixed since round 6 (were environment variables while they were being measured; profiles/HISTORY.md has the sweeps)
		static constexpr u32 classifyRowGroup = 4, regWgsPerCu = 20, f1WgsPerCu = 20, foldBlocks = 65536, upWgsPerCu = 5, mainWgsPerCu = 4, mainBatch = 2, mainUpperNum = 1, mainUpperDen = 4;
		bool fast0() const { return (fast & 1u) != 0; }
		bool fast1() const { return (fast & 2u) != 0; }
	} tune;
	static u32 env_u32(const char* name, u32 fallback) { const char* v = getenv(name); return v ? (u32)atoi(v) : fallback; }

	bool check(hipError_t e, const char* what)
	{
		if (e == hipSuccess) return true;
		lastError = std::string(what) + ": " + hipGetErrorString(e);
		ok = false;
		return false;
	}

	static int device_count()
	{
		int count = 0;
		return hipGetDeviceCount(&count) == hipSuccess ? count : 0;
	}
	bool init(int device, std::string& err)
	{
		int count = 0;
		if (hipGetDeviceCount(&count) != hipSuccess || count <= 0) { err = "no HIP device"; return false; }
		if (!check(hipSetDevice(device), "hipSetDevice")) { err = lastError; return false; }
		this->device = device;
		tune.fast = env_u32("VX_FAST", 3);
		tune.forceWide = env_u32("VX_FORCE_WIDE", 0);
		tune.upper = env_u32("VX_UPPER", 1);
		tune.selfHead = env_u32("VX_SELF_HEAD", 1);
		tune.dirtyFused = env_u32("VX_DIRTY_FUSED", 1);
		hipDeviceProp_t prop;
		if (hipGetDeviceProperties(&prop, device) == hipSuccess) cus = prop.multiProcessorCount;
		if (!check(hipStreamCreateWithFlags(&ownStream, hipStreamNonBlocking), "hipStreamCreate")) { err = lastError; return false; }
		stream = ownStream;
		if (!check(hipStreamCreateWithFlags(&sideA, hipStreamNonBlocking), "hipStreamCreate(side A)")
		    || !check(hipStreamCreateWithFlags(&sideB, hipStreamNonBlocking), "hipStreamCreate(side B)")
		    || !check(hipStreamCreateWithFlags(&sideC, hipStreamNonBlocking), "hipStreamCreate(side C)")
		    || !check(hipEventCreateWithFlags(&evMidC, hipEventDisableTiming), "hipEventCreate")
		    || !check(hipEventCreateWithFlags(&evMidD, hipEventDisableTiming), "hipEventCreate")
		    || !check(hipEventCreateWithFlags(&evSideC, hipEventDisableTiming), "hipEventCreate")
		    || !check(hipEventCreateWithFlags(&evSideD, hipEventDisableTiming), "hipEventCreate")
		    || !check(hipEventCreateWithFlags(&evClassified, hipEventDisableTiming), "hipEventCreate")
		    || !check(hipEventCreateWithFlags(&evMaterial, hipEventDisableTiming), "hipEventCreate")
		    || !check(hipEventCreateWithFlags(&evSideA, hipEventDisableTiming), "hipEventCreate")
		    || !check(hipEventCreateWithFlags(&evSideB, hipEventDisableTiming), "hipEventCreate")
		    || !check(hipEventCreateWithFlags(&evMain, hipEventDisableTiming), "hipEventCreate")
		    || !check(hipEventCreate(&ev0), "hipEventCreate") || !check(hipEventCreate(&ev1), "hipEventCreate")) {
			err = lastError;
			return false;
		}
		const int regSmall = (int)(REG_TAB_LDS + sizeof(RegStateT<REG_CAP_SMALL>)), regLarge = (int)(REG_TAB_LDS + sizeof(RegStateT<4096>));
		const int trLds = (int)(TR_TAB_LDS + sizeof(TrState));
		const int r0Small = (int)(R0_TAB_LDS + sizeof(Reg0State<REG_CAP_SMALL>)), r0Large = (int)(R0_TAB_LDS + sizeof(Reg0State<4096>));
		const int f0Small = (int)(F0_TAB_LDS + sizeof(Fast0State<REG_CAP_SMALL>));
		if (!check(hipFuncSetAttribute((const void*)k_regular0<REG_CAP_SMALL, 0>, hipFuncAttributeMaxDynamicSharedMemorySize, r0Small), "hipFuncSetAttribute(k_regular0 small)")
		    || !check(hipFuncSetAttribute((const void*)k_regular0<4096, 0>, hipFuncAttributeMaxDynamicSharedMemorySize, r0Large), "hipFuncSetAttribute(k_regular0 large)")
		    || !check(hipFuncSetAttribute((const void*)k_regular0<REG_CAP_SMALL, 1>, hipFuncAttributeMaxDynamicSharedMemorySize, r0Small), "hipFuncSetAttribute(k_regular0 small, incremental)")
		    || !check(hipFuncSetAttribute((const void*)k_regular0<4096, 1>, hipFuncAttributeMaxDynamicSharedMemorySize, r0Large), "hipFuncSetAttribute(k_regular0 large, incremental)")
		    || !check(hipFuncSetAttribute((const void*)k_regular0<REG_CAP_SMALL, 2>, hipFuncAttributeMaxDynamicSharedMemorySize, r0Small), "hipFuncSetAttribute(k_regular0 small, handed on)")
		    || !check(hipFuncSetAttribute((const void*)k_regular0_fast<REG_CAP_SMALL>, hipFuncAttributeMaxDynamicSharedMemorySize, f0Small), "hipFuncSetAttribute(k_regular0_fast)")
		    || !check(hipFuncSetAttribute((const void*)k_regular0_fast<REG_CAP_MID>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)(F0_TAB_LDS + sizeof(Fast0State<REG_CAP_MID>))), "hipFuncSetAttribute(k_regular0_fast mid)")
		    || !check(hipFuncSetAttribute((const void*)k_regular0_fast<REG_CAP_BIG>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)(F0_TAB_LDS + sizeof(Fast0State<REG_CAP_BIG>))), "hipFuncSetAttribute(k_regular0_fast big)")
		    || !check(hipFuncSetAttribute((const void*)k_regular0<4096, 2>, hipFuncAttributeMaxDynamicSharedMemorySize, r0Large), "hipFuncSetAttribute(k_regular0 large, handed on)")) {
			err = lastError;
			return false;
		}
		if (!check(hipFuncSetAttribute((const void*)k_regular<REG_CAP_SMALL, 0>, hipFuncAttributeMaxDynamicSharedMemorySize, regSmall), "hipFuncSetAttribute(k_regular small)")
		    || !check(hipFuncSetAttribute((const void*)k_regular<REG_CAP_SMALL, 2>, hipFuncAttributeMaxDynamicSharedMemorySize, regSmall), "hipFuncSetAttribute(k_regular small, handed on)")
		    || !check(hipFuncSetAttribute((const void*)k_regular1_fast<REG_CAP_SMALL>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)(F0_TAB_LDS + sizeof(Fast1State<REG_CAP_SMALL>))), "hipFuncSetAttribute(k_regular1_fast)")
		    || !check(hipFuncSetAttribute((const void*)k_regular1_fast<REG_CAP_MID>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)(F0_TAB_LDS + sizeof(Fast1State<REG_CAP_MID>))), "hipFuncSetAttribute(k_regular1_fast mid)")
		    || !check(hipFuncSetAttribute((const void*)k_regular1_fast<REG_CAP_BIG>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)(F0_TAB_LDS + sizeof(Fast1State<REG_CAP_BIG>))), "hipFuncSetAttribute(k_regular1_fast big)")
		    || !check(hipFuncSetAttribute((const void*)k_dirty_regular0_fast<REG_CAP_BIG>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)(F0_TAB_LDS + sizeof(Fast0State<REG_CAP_BIG>))), "hipFuncSetAttribute(k_dirty_regular0_fast big)")
		    || !check(hipFuncSetAttribute((const void*)k_dirty_regular1_fast<REG_CAP_BIG>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)(F0_TAB_LDS + sizeof(Fast1State<REG_CAP_BIG>))), "hipFuncSetAttribute(k_dirty_regular1_fast big)")
		    || !check(hipFuncSetAttribute((const void*)k_regular<4096, 2>, hipFuncAttributeMaxDynamicSharedMemorySize, regLarge), "hipFuncSetAttribute(k_regular large, handed on)")
		    || !check(hipFuncSetAttribute((const void*)k_regular<4096, 0>, hipFuncAttributeMaxDynamicSharedMemorySize, regLarge), "hipFuncSetAttribute(k_regular large)")
		    || !check(hipFuncSetAttribute((const void*)k_transition<false>, hipFuncAttributeMaxDynamicSharedMemorySize, trLds), "hipFuncSetAttribute(k_transition)")
		    || !check(hipFuncSetAttribute((const void*)k_transition<true>, hipFuncAttributeMaxDynamicSharedMemorySize, trLds), "hipFuncSetAttribute(k_transition, wide)")
		    || !check(hipFuncSetAttribute((const void*)k_main<false>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)(UP_TAB_LDS + MAIN_STATE_LDS)), "hipFuncSetAttribute(k_main)")
		    || !check(hipFuncSetAttribute((const void*)k_main<true>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)(UP_TAB_LDS + MAIN_STATE_LDS)), "hipFuncSetAttribute(k_main, incremental)")
		    || !check(hipFuncSetAttribute((const void*)k_main<false, true>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)(UP_TAB_LDS + MAIN_STATE_LDS)), "hipFuncSetAttribute(k_main, partial)")) {
			err = lastError;
			return false;
		}
		return true;
	}
	void shutdown()
	{
		if (ev0) (void)hipEventDestroy(ev0);
		if (ev1) (void)hipEventDestroy(ev1);
		if (evPrevEnd) (void)hipEventDestroy(evPrevEnd);
		for (int i = 0; i < 9; ++i) if (stageEv[i]) (void)hipEventDestroy(stageEv[i]);
		for (hipStream_t& cs : copyStream) if (cs) { (void)hipStreamDestroy(cs); cs = nullptr; }
		if (evCopy) (void)hipEventDestroy(evCopy);
		if (sideA) (void)hipStreamDestroy(sideA);
		if (sideB) (void)hipStreamDestroy(sideB);
		if (sideC) (void)hipStreamDestroy(sideC);
		for (hipEvent_t e : { evMidC, evMidD, evSideC, evSideD }) if (e) (void)hipEventDestroy(e);
		for (hipEvent_t e : { evClassified, evMaterial, evSideA, evSideB, evMain }) if (e) (void)hipEventDestroy(e);
		if (ownStream) (void)hipStreamDestroy(ownStream);
	}
	bool wants_pyramid() const { return true; }
	bool wants_bricks() const { return true; }
	// dense fields -> mirrors: the blocks of the box { yb0, yb1, zb0, zb1 } (all x), or the listed blocks
	void run_rebrick(const GridView& g, const int dr[4], const int mr[4], const MirrorState& ms, const int box[4], const u32* ids, u32 count)
	{
		const RebrickRanges r = { dr[0], dr[1], dr[2], dr[3], mr[0], mr[1], mr[2], mr[3] };
		if (ids) {
			hipLaunchKernelGGL(k_rebrick, dim3(count), dim3(WG), 0, stream, g, r, ms, 0, 1, 0, ids);
		} else {
			const int nb = g.n >> 4, groups = (nb + 7) >> 3;
			const int yc = box[1] - box[0], zc = box[3] - box[2];
			if (yc <= 0 || zc <= 0) return;
			hipLaunchKernelGGL(k_rebrick, dim3((u32)(groups * yc * zc)), dim3(WG), 0, stream, g, r, ms, box[0], yc, box[2], (const u32*)nullptr);
		}
		check(hipGetLastError(), "k_rebrick launch");
	}
	void make_current() { (void)hipSetDevice(device); }
	void set_stream(void* s) { stream = s ? (hipStream_t)s : ownStream; }
	std::string error() const { return lastError; }
	void* alloc(size_t bytes)
	{
		void* p = nullptr;
		if (!check(hipMalloc(&p, bytes ? bytes : 16), "hipMalloc")) return nullptr;
		return p;
	}
	void free(void* p) { if (p) (void)hipFree(p); }
	// inter-process handle of a device allocation (its base pointer): what a renderer in another process opens with hipIpcOpenMemHandle
	bool ipc_handle(void* p, unsigned char out[64])
	{
		static_assert(sizeof(hipIpcMemHandle_t) == 64, "vx_ipc_meshes carries hipIpcMemHandle_t bytewise");
		hipIpcMemHandle_t h;
		if (!p || !check(hipIpcGetMemHandle(&h, p), "hipIpcGetMemHandle")) return false;
		memcpy(out, &h, 64);
		return true;
	}
	bool fill(void* p, int v, size_t bytes) { return check(hipMemsetAsync(p, v, bytes, stream), "hipMemsetAsync"); }
	bool h2d(void* d, const void* s, size_t bytes)
	{
		return check(hipMemcpyAsync(d, s, bytes, hipMemcpyHostToDevice, stream), "hipMemcpyAsync(H2D)")
		    && check(hipStreamSynchronize(stream), "hipStreamSynchronize");
	}
	// `height` pieces of `width` bytes, host -> device, with their own pitches (the rows of a slab out of a whole host grid)
	bool h2d_2d(void* d, size_t dpitch, const void* s, size_t spitch, size_t width, size_t height)
	{
		return check(hipMemcpy2DAsync(d, dpitch, s, spitch, width, height, hipMemcpyHostToDevice, stream), "hipMemcpy2DAsync(H2D)")
		    && check(hipStreamSynchronize(stream), "hipStreamSynchronize");
	}
	bool d2h(void* d, const void* s, size_t bytes)
	{
		return check(hipMemcpyAsync(d, s, bytes, hipMemcpyDeviceToHost, stream), "hipMemcpyAsync(D2H)")
		    && check(hipStreamSynchronize(stream), "hipStreamSynchronize");
	}
	void sync() { (void)hipStreamSynchronize(stream); }
	// (Polling the stream with hipStreamQuery instead was tried against the sporadic 80-95 ms stalls of DESIGN section 9: they also
	// hit kernel launches and event records, with and without polling, and polling costs 5-10 us per call.)
	bool sync_ok() { return check(hipStreamSynchronize(stream), "hipStreamSynchronize"); }
	bool d2h_async(void* d, const void* s, size_t bytes) { return check(hipMemcpyAsync(d, s, bytes, hipMemcpyDeviceToHost, stream), "hipMemcpyAsync(D2H)"); }
	bool h2d_async(void* d, const void* s, size_t bytes) { return check(hipMemcpyAsync(d, s, bytes, hipMemcpyHostToDevice, stream), "hipMemcpyAsync(H2D)"); } // (page-locked source)
	// a copy that does not queue behind the context's streams (diagnostics while a run is in flight)
	bool d2h_side(void* d, const void* s, size_t bytes)
	{
		(void)hipSetDevice(device);
		if (!copyStream[0] && hipStreamCreateWithFlags(&copyStream[0], hipStreamNonBlocking) != hipSuccess) return false;
		return hipMemcpyAsync(d, s, bytes, hipMemcpyDeviceToHost, copyStream[0]) == hipSuccess && hipStreamSynchronize(copyStream[0]) == hipSuccess;
	}
	void* alloc_pinned(size_t bytes)
	{
		void* p = nullptr;
		if (!check(hipHostMalloc(&p, bytes ? bytes : 16, hipHostMallocDefault), "hipHostMalloc")) return nullptr;
		return p;
	}
	void free_pinned(void* p) { if (p) (void)hipHostFree(p); }
	static void release_pinned(void* p) { if (p) (void)hipHostFree(p); } // arenas outlive their context
	// Large device -> page-locked host copies, after everything queued on the run's stream: cut into pieces that
	// can alternate between copy streams (VX_D2H_STREAMS > 1).  Measured on MI355X (tools/d2h_time.py, 499 MB): one stream
	// 57 GB/s, 2-4 streams 53-56 GB/s - the link is the limit, so one stream is the default.
	bool d2h_bulk(void* const* dst, const void* const* src, const size_t* bytes, int count)
	{
		const u32 lanes = 1; // (more copy streams were measured and bought nothing: one engine saturates the link, profiles/r03_d2h_time.txt)
		const size_t piece = (size_t)32 << 20;
		size_t total = 0;
		for (int i = 0; i < count; ++i) total += bytes[i];
		if (!total) return true;
		if (lanes == 1 || total <= piece) {
			for (int i = 0; i < count; ++i)
				if (bytes[i] && !check(hipMemcpyAsync(dst[i], src[i], bytes[i], hipMemcpyDeviceToHost, stream), "hipMemcpyAsync(D2H)")) return false;
			return check(hipStreamSynchronize(stream), "hipStreamSynchronize");
		}
		for (u32 l = 0; l < lanes; ++l)
			if (!copyStream[l] && !check(hipStreamCreateWithFlags(&copyStream[l], hipStreamNonBlocking), "hipStreamCreate(copy)")) return false;
		if (!evCopy && !check(hipEventCreateWithFlags(&evCopy, hipEventDisableTiming), "hipEventCreate(copy)")) return false;
		if (!check(hipEventRecord(evCopy, stream), "hipEventRecord(copy)")) return false;
		for (u32 l = 0; l < lanes; ++l) if (!check(hipStreamWaitEvent(copyStream[l], evCopy, 0), "hipStreamWaitEvent(copy)")) return false;
		u32 next = 0;
		bool ok = true;
		for (int i = 0; i < count && ok; ++i)
			for (size_t off = 0; off < bytes[i] && ok; off += piece) {
				const size_t len = std::min(piece, bytes[i] - off);
				ok = check(hipMemcpyAsync((char*)dst[i] + off, (const char*)src[i] + off, len, hipMemcpyDeviceToHost, copyStream[next]), "hipMemcpyAsync(D2H)");
				next = (next + 1) % lanes;
			}
		for (u32 l = 0; l < lanes; ++l) ok = check(hipStreamSynchronize(copyStream[l]), "hipStreamSynchronize(copy)") && ok;
		return ok;
	}
	void stage_enable(bool on)
	{
		stageOn = on;
		stageValid = false;
		if (on) for (int i = 0; i < 9; ++i) if (!stageEv[i] && !check(hipEventCreate(&stageEv[i]), "hipEventCreate(stage)")) { stageOn = false; return; }
	}
	void stage_mark(int i)
	{
		if (!stageOn) return;
		(void)hipEventRecord(stageEv[i], stream);
		if (i == 7) stageValid = true;
	}
	bool stage_ms(float* ms)
	{
		if (!stageOn || !stageValid) return false;
		if (hipEventSynchronize(stageEv[7]) != hipSuccess) return false;
		// reset + block classes, classify, hierarchy, material | regular level 0, regular levels >= 1 | transition, block lists
		static const int from[8] = { 0, 1, 2, 3, 4, 8, 5, 6 }, to[8] = { 1, 2, 3, 4, 8, 5, 6, 7 };
		for (int i = 0; i < 8; ++i) { ms[i] = 0.f; (void)hipEventElapsedTime(&ms[i], stageEv[from[i]], stageEv[to[i]]); }
		return true;
	}
	void begin_timing() { (void)hipEventRecord(ev0, stream); }
	float end_timing_ms()
	{
		(void)hipEventRecord(ev1, stream);
		if (!sync_ok()) return -1.f; // (ev1 is the last entry of the stream)
		float ms = 0.f;
		(void)hipEventElapsedTime(&ms, ev0, ev1);
		return ms;
	}
	void run_decode_grid(const u8* blob, const uint64_t* where, u32 n, i8* dist, u8* mat, u8* blend, u8* flags)
	{
		const u32 nb = n / 16;
		hipLaunchKernelGGL(k_decode_grid, dim3(((nb + 7) / 8) * nb * nb), dim3(WG), 0, stream, blob, (const unsigned long long*)where, n, dist, mat, blend, flags);
		check(hipGetLastError(), "k_decode_grid launch");
	}
	bool d2d(void* d, const void* s, size_t bytes) { return check(hipMemcpyAsync(d, s, bytes, hipMemcpyDeviceToDevice, stream), "hipMemcpyAsync(D2D)"); }
	void run_encode_grid(const GridView& g, u32* meta, const uint64_t* where, u8* blob)
	{
		const u32 nb = (u32)g.n / 16;
		hipLaunchKernelGGL(k_encode_grid, dim3(((nb + 7u) / 8u) * nb * nb), dim3(WG), 0, stream, g, meta, (const unsigned long long*)where, blob);
		check(hipGetLastError(), "k_encode_grid launch");
	}
	void run_heightmap(const GridView& g, const i8* map, u8* flags)
	{
		const u32 n = (u32)g.n, nb = n / 16;
		const size_t segs = (size_t)n * n * (n / 16);
		hipLaunchKernelGGL(k_heightmap, dim3((u32)((segs + WG - 1) / WG)), dim3(WG), 0, stream, g, map);
		hipLaunchKernelGGL(k_edit_flags, dim3(nb * nb * nb), dim3(WG), 0, stream, g, flags, (const u32*)nullptr, nb * nb * nb);
		check(hipGetLastError(), "k_heightmap launch");
	}
	// the synthetic terrain into the resident fields (dist over dr, material + blend over mr), BF_Empty of the listed blocks
	void run_terrain(const GridView& g, u32 seed, float* height, const int dr[4], const int mr[4], u8* flags, const u32* ids, u32 count, u32 style)
	{
		const u32 n = (u32)g.n;
		TerrainRange d = { dr[0], dr[1], dr[2], dr[3] }, m = { mr[0], mr[1], mr[2], mr[3] };
		hipLaunchKernelGGL(k_terrain_height, dim3((n * n + WG - 1) / WG), dim3(WG), 0, stream, n, seed, height);
		const size_t segs = (size_t)(n / 16) * (size_t)(d.y1 - d.y0) * (size_t)(d.z1 - d.z0);
		if (segs) hipLaunchKernelGGL(k_terrain_fill, dim3((u32)((segs + WG - 1) / WG)), dim3(WG), 0, stream, g, seed, (const float*)height, d, m, style);
		if (count) hipLaunchKernelGGL(k_edit_flags, dim3(count), dim3(WG), 0, stream, g, flags, ids, count);
		check(hipGetLastError(), "k_terrain launch");
	}
	void run_copy_segments(const u32* seg, u32 count, const void* src, void* dst, u32 elemBytes)
	{
		if (!count) return;
		hipLaunchKernelGGL(k_copy_segments, dim3(count), dim3(WG), 0, stream, seg, (const u32*)src, (u32*)dst, elemBytes / 4);
		check(hipGetLastError(), "k_copy_segments launch");
	}
	void run_scatter_blocks(const u32* ids, u32 count, u32 n, const u8* sd, const u8* sm, const u8* sb, u8* dist, u8* mat, u8* blend)
	{
		hipLaunchKernelGGL(k_scatter_blocks, dim3(count), dim3(WG), 0, stream, ids, n, sd, sm, sb, dist, mat, blend);
		check(hipGetLastError(), "k_scatter_blocks launch");
	}
	void run_box_ids(u32* out, const u32 first[3], const u32 count[3], u32 nb)
	{
		const u32 total = count[0] * count[1] * count[2];
		hipLaunchKernelGGL(k_box_ids, dim3((total + WG - 1) / WG), dim3(WG), 0, stream, out, first[0], first[1], first[2], count[0], count[1], total, nb);
		check(hipGetLastError(), "k_box_ids launch");
	}
	void run_edit(const GridView& g, u8* flags, const u32* ids, u32 count, const EditParams& e)
	{
		hipLaunchKernelGGL(k_edit, dim3(count), dim3(WG), 0, stream, g, ids, e);
		if (e.kind == EDIT_BALL) hipLaunchKernelGGL(k_edit_flags, dim3(count), dim3(WG), 0, stream, g, flags, ids, count);
		check(hipGetLastError(), "k_edit launch");
	}
	void end_timing_record() { (void)hipEventRecord(ev1, stream); }
	// Diagnostics (VX_HOST_TIMING): how long the stream sat idle on the DEVICE's clock between the end of the previous timed run
	// and the start of this one - to be called after this run was waited for; then marks this run's end for the next call.  A
	// host wait that is much longer than the kernels took is either a late wake-up of the host (idle time as always) or a late
	// start on the device (the idle time holds the delay).  -1: no previous run.
	float idle_before_ms()
	{
		float ms = -1.f;
		if (!evPrevEnd && hipEventCreate(&evPrevEnd) != hipSuccess) return ms;
		if (havePrevEnd && hipEventElapsedTime(&ms, evPrevEnd, ev0) != hipSuccess) ms = -1.f;
		havePrevEnd = hipEventRecord(evPrevEnd, stream) == hipSuccess && hipStreamSynchronize(stream) == hipSuccess;
		return ms;
	}
	float elapsed_ms() // after the stream was synchronised
	{
		float ms = 0.f;
		if (hipEventElapsedTime(&ms, ev0, ev1) != hipSuccess) return -1.f;
		return ms;
	}

	// header words = 0 and every level's block -> slot map = -1: done by the first launch of the run (k_run_head)
	ResetRanges pendingReset = {};
	bool pendingClean = false;   // the run's counters and the maps of the levels >= 1 are in their start state already (k_tail of the run before)
	ResetRanges nextReset = {};  // what k_tail puts into its start state for the next run (header == nullptr: nothing)
	bool tailCleaned = false;    // the last k_tail did
	void set_next_reset(u32* header, u32 headerWords, u32* listCounts, u32 listWgs, int* const* maps, const u32* ids)
	{
		ResetRanges r;
		u32 run = headerWords;
		r.header = header; r.headerWords = headerWords; r.partials = nullptr; r.partialCount = 0;
		r.listCounts = listCounts; r.listWgs = listWgs;
		r.maps[0] = nullptr;
		r.start[0] = run; // (level 0's map is not part of a set: k_run_head writes all of it)
		for (u32 l = 1; l < MAX_LEVELS; ++l) { r.start[l] = run; r.maps[l] = maps[l]; if (maps[l]) run += ids[l]; }
		r.start[MAX_LEVELS] = run;
		nextReset = r;
	}
	u32 headWorkgroups = 0; // of the last k_run_head launch: that many block-class partial sums sit behind the header
	template <typename P>
	void run_reset(const P& p, u32 levels, u32* header, u32 headerWords, u32* listCounts, u32 listWgs, bool alreadyClean = false)
	{
		ResetRanges r;
		pendingClean = alreadyClean;
		for (u32 l = 0; l < MAX_LEVELS; ++l) r.maps[l] = p.levels[l].slotOf;
		u32 run = headerWords;
		r.header = header; r.headerWords = headerWords;
		r.partials = header + headerWords; r.partialCount = 0;
		r.listCounts = listCounts; r.listWgs = listWgs;
		for (u32 l = 0; l < MAX_LEVELS; ++l) {
			r.start[l] = run;
			if (l < levels) run += p.levels[l].cnt * p.levels[l].cnt * p.levels[l].cnt;
		}
		r.start[MAX_LEVELS] = run;
		pendingReset = r;
	}

	template <typename P>
	static ExecParamsDev dev(const P& p)
	{
		static_assert(sizeof(P) == sizeof(ExecParamsDev), "ExecParams layout");
		ExecParamsDev d;
		memcpy(&d, &p, sizeof(d));
		return d;
	}

	// full runs over few level-0 blocks: k_classify does k_hierarchy's work (see there)
	template <typename P>
	bool classify_activates_ancestors(const P& p) const
	{
		const LevelDesc& L = p.levels[0];
		return (size_t)L.cnt * (L.yb1 - L.yb0) * (L.zb1 - L.zb0) <= (size_t)tune.foldBlocks;
	}
	// the runs whose head hands out the slots and whose k_main classifies (see k_run_head): what k_classify's launch and
	// k_hierarchy's cost is most of a small run
	template <typename P>
	bool self_head(const P& p, u32 levels) const { return single_stream(p, levels); }
	template <typename P>
	bool ancestors_with_classification(const P& p, u32 levels) const { return self_head(p, levels) || classify_activates_ancestors(p); }
	// carryClassified: the classify launch carries the event that releases the level-0 regular pass on side stream A
	template <typename P>
	void run_classify(const P& p, bool carryClassified)
	{
		const LevelDesc& L = p.levels[0];
		const u32 tilesX = (L.cnt + TB - 1) / TB;
		const u32 rowsY = L.yb1 - L.yb0;
		const u32 grid = tilesX * rowsY * (L.zb1 - L.zb0);
		if (!grid) return;
		if (self_head(p, p.G.levels)) {
			// no classification pass: k_run_head hands out the slots, k_main's level-0 blocks form their own bitmaps
			ResetRanges r = pendingReset;
			pendingReset.header = nullptr;
			if (r.header && !pendingClean) {
				const u32 lanes = std::max<u32>((r.start[MAX_LEVELS] + 3) / 4, r.listWgs);
				hipLaunchKernelGGL(k_reset, dim3((lanes + WG - 1) / WG), dim3(WG), 0, stream, dev(p), r);
			}
			r.header = nullptr;
			headWorkgroups = 0; // (no partial sums behind the header: see k_run_head)
			// (workgroups are 8 x 8 x 4 boxes of blocks at aligned places that cover the block range)
			const u32 boxes = ((L.cnt + 7u) >> 3) * (((L.yb1 + 7u) >> 3) - (L.yb0 >> 3)) * (((L.zb1 + 3u) >> 2) - (L.zb0 >> 2));
			hipLaunchKernelGGL(k_run_head, dim3(boxes), dim3(WG), 0, stream, dev(p), r, 1u);
			check(hipGetLastError(), "k_run_head launch");
			stage_mark(1);
			return;
		}
		{
			const ResetRanges r = pendingReset;
			pendingReset.header = nullptr;
			const u32 lanes = std::max<u32>(L.cnt * rowsY * (L.zb1 - L.zb0), r.header ? std::max<u32>((r.start[MAX_LEVELS] + 3) / 4, r.listWgs) : 0u);
			headWorkgroups = (lanes + WG - 1) / WG;
			hipLaunchKernelGGL(k_run_head, dim3(headWorkgroups), dim3(WG), 0, stream, dev(p), r, 0u);
		}
		const u32 rows = rowsY * (L.zb1 - L.zb0);
		u32 rowGroup = 0; // 0 = no remap
		if ((rows & 7u) == 0) {
			rowGroup = tune.classifyRowGroup;
			while (rowGroup > 1 && rows % (8 * rowGroup)) rowGroup >>= 1;
			if (!rowGroup) rowGroup = 1;
		}
		stage_mark(1); // stage times: [0] = reset + block classes, [1] = k_classify alone
		if (carryClassified && (!single_stream(p, p.G.levels) || largeClass)) doneEvent = evClassified; // (a single-stream run forks only for the upper capacity classes of dense surfaces)
		launch_with_event(k_classify, dim3(grid), 0u, dev(p), rowGroup, classify_activates_ancestors(p) ? 1u : 0u);
		check(hipGetLastError(), "k_classify launch");
	}
	template <typename P>
	void run_classify_blocks(const P& p, const u32* coords, u32 count)
	{
		if (!count) return;
		hipLaunchKernelGGL(k_classify_blocks, dim3(std::min<u32>(count, (u32)cus * 4)), dim3(WG), 0, stream, dev(p), coords, count);
		check(hipGetLastError(), "k_classify_blocks launch");
	}
	template <typename P>
	void run_build_worklist(const P& p, const u32* coords, const u32* start, const u32*, u32 levels, u32* work)
	{
		DirtyRanges r;
		for (u32 l = 0; l <= MAX_LEVELS; ++l) r.start[l] = start[l < levels ? l : levels];
		if (!r.start[levels]) return;
		hipLaunchKernelGGL(k_build_worklist, dim3((r.start[levels] + WG - 1) / WG), dim3(WG), 0, stream, dev(p), coords, r, levels, work);
		check(hipGetLastError(), "k_build_worklist launch");
	}
	template <typename P>
	void run_gather_records(const P& p, u32 levels, const u32* start, BlockRecord* out)
	{
		DirtyRanges r;
		for (u32 l = 0; l <= MAX_LEVELS; ++l) r.start[l] = start[l < levels ? l : levels];
		if (!r.start[levels]) return;
		hipLaunchKernelGGL(k_gather_records, dim3((r.start[levels] + WG - 1) / WG), dim3(WG), 0, stream, dev(p), r, levels, out);
		check(hipGetLastError(), "k_gather_records launch");
	}
	// A hand-over event between streams is attached to the kernel it follows (hipExtLaunchKernelGGL: the dispatch's own
	// completion signal) instead of being recorded behind it: a recorded event is one more packet in the stream, and the
	// next kernel of that stream starts ~8 us later.
	hipEvent_t doneEvent = nullptr; // completion event of the next launch made through launch_with_event
	template <typename K, typename... Args>
	void launch_with_event(K kernel, dim3 grid, u32 lds, Args... args)
	{
		if (doneEvent) hipExtLaunchKernelGGL(kernel, grid, dim3(WG), lds, stream, nullptr, doneEvent, 0u, args...);
		else hipLaunchKernelGGL(kernel, grid, dim3(WG), lds, stream, args...);
		doneEvent = nullptr;
	}
	template <typename P>
	void run_hierarchy(const P& p, u32 levels, bool carryClassified = false)
	{
		if (carryClassified && single_stream(p, levels) && !largeClass) carryClassified = false;
		if (levels < 2) { if (carryClassified) (void)hipEventRecord(evClassified, stream); return; }
		const u32 grid = (p.levels[0].cap + WG - 1) / WG;
		if (carryClassified) doneEvent = evClassified;
		launch_with_event(k_hierarchy, dim3(grid), 0u, dev(p), levels);
		check(hipGetLastError(), "k_hierarchy launch");
	}
	template <typename P>
	void run_material(const P& p, u32 level)
	{
		const u32 cap = p.levels[level].cap;
		if (!cap) return;
		const u32 grid = std::min<u32>(cap, (u32)cus * 8);
		launch_with_event(k_material, dim3(grid), 0u, dev(p), level);
		check(hipGetLastError(), "k_material launch");
	}
	// Regular cells of the levels [levelBegin, levels): level 0 has its own kernel (vx_regular0.inl).  The 4096-cell
	// capacity class is only launched when blocks that large are expected (largeClass, set by the host from the
	// previous run's count; a run that meets an unexpected one is repeated with the class enabled).
	bool largeClass = true;
	template <typename P>
	void launch_regular(const P& p, u32 levelBegin, u32 levels, hipStream_t on)
	{
		if (levelBegin == 0 && p.levels[0].cap) {
			const u32 cap = p.levels[0].cap;
			const u32 gridS = std::min<u32>(cap, (u32)cus * tune.regWgsPerCu);
			const u32 ldsS = R0_TAB_LDS + sizeof(Reg0State<REG_CAP_SMALL>), ldsL = R0_TAB_LDS + sizeof(Reg0State<4096>), gridL = std::min<u32>(cap, (u32)cus);
			if (p.G.dirty) {
				hipLaunchKernelGGL((k_regular0<REG_CAP_SMALL, 1>), dim3(gridS), dim3(WG), ldsS, on, dev(p), 0u);
				if (largeClass) hipLaunchKernelGGL((k_regular0<4096, 1>), dim3(gridL), dim3(WG), ldsL, on, dev(p), (u32)REG_CAP_SMALL);
			} else if (tune.fast0()) {
				// blocks without a zero sample: the table-driven pass; the others are handed on through Globals::slowItems
				// (dense surfaces: a second table-driven class up to REG_CAP_MID cells; beyond that, and for what either class
				// hands on, the general pass in its two classes)
				// Overlapped runs put the upper classes on a stream of their own (side C, released like this one by the
				// classification): with three or one workgroup per CU they run long and leave room, so they belong beside the
				// first class, not behind it (second bench workload: 3.84 -> 3.69 ms).  What the two table-driven classes hand
				// on is only complete when both are done.
				const bool spread = largeClass && (on == sideA || (level0Done && overlappedTail));
				hipStream_t upper = spread ? sideC : on;
				if (spread) { (void)hipStreamWaitEvent(sideC, evClassified, 0); spreadC = true; }
				if (!level0Done) hipLaunchKernelGGL((k_regular0_fast<REG_CAP_SMALL>), dim3(gridS), dim3(WG), F0_TAB_LDS + sizeof(Fast0State<REG_CAP_SMALL>), on, dev(p), 0u);
				if (largeClass) hipLaunchKernelGGL((k_regular0_fast<REG_CAP_MID>), dim3(std::min<u32>(cap, (u32)cus * 12)), dim3(WG), F0_TAB_LDS + sizeof(Fast0State<REG_CAP_MID>), upper, dev(p), (u32)REG_CAP_SMALL);
				// the third class (blocks beyond REG_CAP_MID cells), table-driven as well: behind the second on its stream; what the
				// three classes hand on (zero samples) is complete when all of them are done
				const u32 ldsB = F0_TAB_LDS + sizeof(Fast0State<REG_CAP_BIG>), gridB = std::min<u32>(cap, (u32)cus * 4);
				if (largeClass) hipLaunchKernelGGL((k_regular0_fast<REG_CAP_BIG>), dim3(gridB), dim3(WG), ldsB, upper, dev(p), (u32)REG_CAP_MID);
				if (spread) {
					(void)hipEventRecord(evMidC, sideC);
					(void)hipEventRecord(evSideC, sideC);
					(void)hipStreamWaitEvent(on, evMidC, 0);
				}
				tailWgs[0] = slow_grid(std::min<u32>(gridS, (u32)cus * 4), 0);
				if (!tailPending) hipLaunchKernelGGL((k_regular0<REG_CAP_SMALL, 2>), dim3(tailWgs[0]), dim3(WG), ldsS, on, dev(p), 0u);
				if (largeClass) {
					hipLaunchKernelGGL((k_regular0<4096, 2>), dim3(gridL), dim3(WG), ldsL, on, dev(p), (u32)REG_CAP_SMALL);
				}
			} else {
				hipLaunchKernelGGL((k_regular0<REG_CAP_SMALL, 0>), dim3(gridS), dim3(WG), ldsS, on, dev(p), 0u);
				if (largeClass) hipLaunchKernelGGL((k_regular0<4096, 0>), dim3(gridL), dim3(WG), ldsL, on, dev(p), (u32)REG_CAP_SMALL);
			}
			levelBegin = 1;
			if (stageOn && on == stream) (void)hipEventRecord(stageEv[8], on);
		}
		u32 cap = 0;
		for (u32 l = levelBegin; l < levels; ++l) cap += p.levels[l].cap;
		if (cap) {
			const u32 ldsS = REG_TAB_LDS + sizeof(RegStateT<REG_CAP_SMALL>);
			u32 generalBegin = levelBegin;
			// levels with a lattice copy: blocks without a zero lattice sample take the table-driven pass; what it hands on
			// (Globals::slowItems[1]) and the coarser levels go through the general pass.  (32-bit voxel offsets: mirrors < 4 GiB.)
			const u32 fastEnd = std::min<u32>(levels, PYRAMID_LEVELS);
			const bool mirrorsSmall = (unsigned long long)p.G.grid.n * (unsigned long long)p.G.grid.n * (unsigned long long)p.G.grid.n < (1ull << 32);
			if (!p.G.dirty && tune.fast1() && levelBegin == 1 && fastEnd > 1 && mirrorsSmall && p.G.pyr[1].data) {
				u32 capFast = 0;
				for (u32 l = 1; l < fastEnd; ++l) capFast += p.levels[l].cap;
				// (as on level 0: the upper classes beside the first one when the run is overlapped, i.e. when this is the main
				// stream of run_overlapped_tail - on side stream B, behind the transition pass)
				const bool spread = largeClass && overlappedTail && on == stream && !upperDone;
				hipStream_t sideD = sideB;
				hipStream_t upper = spread ? sideD : on;
				const u32 ldsL = REG_TAB_LDS + sizeof(RegStateT<4096>);
				if (spread) { (void)hipStreamWaitEvent(sideD, evMaterial, 0); spreadD = true; }
				if (!upperDone) hipLaunchKernelGGL(k_regular1_fast<REG_CAP_SMALL>, dim3(std::min<u32>(capFast, (u32)cus * tune.f1WgsPerCu)), dim3(WG), F0_TAB_LDS + sizeof(Fast1State<REG_CAP_SMALL>), on, dev(p), fastEnd, 0u);
				if (largeClass) hipLaunchKernelGGL(k_regular1_fast<REG_CAP_MID>, dim3(std::min<u32>(capFast, (u32)cus * 12)), dim3(WG), F0_TAB_LDS + sizeof(Fast1State<REG_CAP_MID>), upper, dev(p), fastEnd, (u32)REG_CAP_SMALL);
				if (largeClass) hipLaunchKernelGGL(k_regular1_fast<REG_CAP_BIG>, dim3(std::min<u32>(capFast, (u32)cus * 4)), dim3(WG), F0_TAB_LDS + sizeof(Fast1State<REG_CAP_BIG>), upper, dev(p), fastEnd, (u32)REG_CAP_MID);
				if (spread) {
					(void)hipEventRecord(evMidD, sideD);
					(void)hipEventRecord(evSideD, sideD);
					(void)hipStreamWaitEvent(on, evMidD, 0);
				}
				tailWgs[1] = slow_grid(std::min<u32>(capFast, (u32)cus * 4), 1);
				if (!tailPending) hipLaunchKernelGGL((k_regular<REG_CAP_SMALL, 2>), dim3(tailWgs[1]), dim3(WG), ldsS, on, dev(p), 1u, levels, 0u);
				if (largeClass) {
					hipLaunchKernelGGL((k_regular<4096, 2>), dim3(std::min<u32>(capFast, (u32)cus)), dim3(WG), ldsL, on, dev(p), 1u, levels, (u32)REG_CAP_SMALL);
				}
				generalBegin = fastEnd;
			}
			u32 capGeneral = 0;
			for (u32 l = generalBegin; l < levels; ++l) capGeneral += p.levels[l].cap;
			if (capGeneral) hipLaunchKernelGGL((k_regular<REG_CAP_SMALL, 0>), dim3(std::min<u32>(capGeneral, (u32)cus * tune.regWgsPerCu)), dim3(WG), ldsS, on, dev(p), generalBegin, levels, 0u);
			if (largeClass && capGeneral) hipLaunchKernelGGL((k_regular<4096, 0>), dim3(std::min<u32>(capGeneral, (u32)cus)), dim3(WG), REG_TAB_LDS + sizeof(RegStateT<4096>), on, dev(p), generalBegin, levels, (u32)REG_CAP_SMALL);
		}
		check(hipGetLastError(), "k_regular launch");
	}
	template <typename P>
	void run_regular(const P& p, u32 levels) { launch_regular(p, 0, levels, stream); }

	// Everything behind the classification as ONE launch (vx_main.inl) - where the table-driven regular pass of the levels >= 1
	// and the 32-bit-offset transition pass apply (lattice copies resident, mirrors below 4 GiB); otherwise the chain of
	// launches it replaces.  With the level-0 queue inside (the default) a full run is a single stream without events.
	// What the table-driven passes handed on in the previous full run of this context (blocks with a zero sample; ~0u =
	// unknown): the general passes behind them are launched with about that many workgroups - they stride over whatever
	// they find, and a launch of a thousand workgroups that find nothing costs 4.5 us against 2.
	bool tailPending = false; // single-stream runs: the general passes behind k_main wait for the list pass and share its launch (k_tail)
	u32 tailWgs[2] = { 0, 0 };
	u32* tailDone = nullptr;  // header word for k_tail's counter (set by the host with the header)
	u32 slowHint[2] = { ~0u, ~0u };
	u32 slow_grid(u32 full, int which) const { return slowHint[which] == ~0u ? full : std::max<u32>(1u, std::min<u32>(full, slowHint[which] ? std::max<u32>(slowHint[which] + slowHint[which] / 4, 32u) : 2u)); }
	u32 upperItemsHint = 0; // upper-queue items of the previous full run of this context (0 = unknown)
	bool upperDone = false;  // inside run_overlapped_tail: k_main did the first capacity class of the levels 1 .. fastEnd - 1
	bool level0Done = false; // ... and the first capacity class of level 0
	template <typename P>
	bool main_applies(const P& p, u32 levels) const
	{
		const bool mirrorsSmall = (unsigned long long)p.G.grid.n * (unsigned long long)p.G.grid.n * (unsigned long long)p.G.grid.n < (1ull << 32);
		// (dense surfaces - blocks beyond the first capacity class are expected - keep the chain of launches on five streams: the
		// upper classes run at one to three workgroups per CU for hundreds of microseconds and belong BESIDE the rest; behind
		// k_main they made the second bench workload 3.9 ms per step against 3.5)
		return tune.upper && tune.fast1() && !tune.forceWide && !largeClass && !p.G.dirty && levels > 1 && mirrorsSmall && p.G.pyr[1].data != nullptr;
	}
	template <typename P>
	bool single_stream(const P& p, u32 levels) const { return main_applies(p, levels) && tune.fast0() && tune.selfHead; }
	// Partial runs (vx_polygonize_from): the levels below emitFrom get caches and bitmaps but no meshes.  Only the single-stream
	// form offers it (partial_applies); the host asks before it launches and tells its caller what the run really did.
	u32 emitFrom = 0;
	template <typename P>
	bool partial_applies(const P& p, u32 levels) const { return single_stream(p, levels) && !stageOn; }
	template <typename P>
	void run_main(const P& p, u32 levels, bool withLevel0)
	{
		MainPlan plan;
		memset(&plan, 0, sizeof(plan));
		plan.levels = levels;
		plan.fastEnd = std::min<u32>(levels, PYRAMID_LEVELS);
		plan.level0 = withLevel0 ? 1u : 0u;
		plan.batch = tune.mainBatch;
		plan.upperNum = withLevel0 ? tune.mainUpperNum : 1u; plan.upperDen = withLevel0 ? tune.mainUpperDen : 1u;
		unsigned long long items = 0; // at most: one material item per block, one regular, one transition
		for (u32 l = 1; l < levels; ++l) items += (unsigned long long)p.levels[l].cap * (1u + (l < plan.fastEnd ? 1u : 0u) + (p.levels[l].hasTransitions ? 1u : 0u));
		if (withLevel0) items += p.levels[0].cap;
		if (!items) return;
		// persistent workgroups; without the level-0 queue at most as many as the previous run had items (the host's hint; any
		// number is correct - a workgroup that finds the queues empty leaves - but every workgroup costs a dequeue)
		u32 grid = (u32)std::min<unsigned long long>(items, (unsigned long long)cus * (withLevel0 ? tune.mainWgsPerCu : tune.upWgsPerCu));
		if (!withLevel0 && upperItemsHint) grid = std::max<u32>(std::min<u32>(grid, upperItemsHint), std::min<u32>(grid, (u32)cus));
		if (emitFrom && withLevel0) {
			// (no level-0 queue: the persistent workgroups are the upper queue's)
			plan.level0 = 0u; plan.upperNum = plan.upperDen = 1u; plan.emitFrom = emitFrom;
			launch_with_event(k_main<false, true>, dim3(grid), UP_TAB_LDS + MAIN_STATE_LDS, dev(p), plan);
		} else
			launch_with_event(k_main<false>, dim3(grid), UP_TAB_LDS + (withLevel0 ? MAIN_STATE_LDS : UP_STATE_LDS), dev(p), plan);
		check(hipGetLastError(), "k_main launch");
	}

	// ---- incremental runs as three launches (k_dirty_head | k_main<true> | k_dirty_tail) -------------------------------------------
	// where the table-driven passes apply and no block beyond the first capacity class is expected (a run that meets one says so
	// in its header and is repeated as the chain of launches)
	template <typename P>
	bool dirty_fused_applies(const P& p, u32 levels, bool largeExpected) const
	{
		const bool mirrorsSmall = (unsigned long long)p.G.grid.n * (unsigned long long)p.G.grid.n * (unsigned long long)p.G.grid.n < (1ull << 32);
		(void)largeExpected; // (blocks beyond the first capacity class: two launches more, see run_dirty_fused)
		return tune.dirtyFused && tune.upper && tune.fast0() && tune.fast1() && !tune.forceWide && levels > 1 && mirrorsSmall && p.G.pyr[1].data != nullptr;
	}
	struct DirtyLaunch {
		u32 lo[MAX_LEVELS][3], hi[MAX_LEVELS][3], start[MAX_LEVELS + 1];
		u32* work; u32* info; u32* ticket; u32 ticketTarget;
		u32* header; u32 resetFrom, resetTo, poolVerts, poolIdx;
		u32* roleTicket; u32* slowDone;
		BlockRecord* hostRecs; u32* hostHeader; u32 headerWords, publishedWord;
	};
	// largeExpected: blocks with more non-trivial cells than the first capacity class holds are left alone by k_main<true>; the
	// 4096-cell class of the general pass takes them from the work lists, in two launches of its own in front of the tail
	// (launched only when the run before met such blocks; a run that meets one unannounced says so in its header and is repeated)
	template <typename P>
	void run_dirty_fused(const P& p, u32 levels, const DirtyLaunch& q, bool large0Expected, bool largeUpperExpected)
	{
		DirtyPlan d;
		memset(&d, 0, sizeof(d));
		memcpy(d.lo, q.lo, sizeof(d.lo)); memcpy(d.hi, q.hi, sizeof(d.hi)); memcpy(d.start, q.start, sizeof(d.start));
		d.levels = levels; d.work = q.work; d.info = q.info; d.ticket = q.ticket; d.ticketTarget = q.ticketTarget;
		d.header = q.header; d.resetFrom = q.resetFrom; d.resetTo = q.resetTo; d.poolVerts = q.poolVerts; d.poolIdx = q.poolIdx;
		hipLaunchKernelGGL(k_dirty_head, dim3(q.start[1]), dim3(WG), 0, stream, dev(p), d);
		check(hipGetLastError(), "k_dirty_head launch");
		MainPlan plan;
		memset(&plan, 0, sizeof(plan));
		plan.levels = levels;
		plan.fastEnd = std::min<u32>(levels, PYRAMID_LEVELS);
		plan.level0 = 1u;
		const u32 slots = (u32)cus * tune.mainWgsPerCu;
		plan.batch = q.start[1] > 2u * slots ? tune.mainBatch : 1u; // (few blocks: every one its own workgroup)
		plan.upperNum = tune.mainUpperNum; plan.upperDen = tune.mainUpperDen;
		memcpy(plan.boxLo, q.lo, sizeof(plan.boxLo)); memcpy(plan.boxHi, q.hi, sizeof(plan.boxHi));
		u32 items = q.start[1], upperVol = 0;
		for (u32 l = 1; l < levels; ++l) {
			const u32 v = q.start[l + 1] - q.start[l];
			items += v * (1u + (l < plan.fastEnd ? 1u : 0u) + (p.levels[l].hasTransitions ? 1u : 0u));
			if (l >= plan.fastEnd) upperVol += v;
		}
		hipLaunchKernelGGL(k_main<true>, dim3(std::max<u32>(1u, std::min<u32>(items, slots))), dim3(WG), UP_TAB_LDS + MAIN_STATE_LDS, stream, dev(p), plan);
		check(hipGetLastError(), "k_main (incremental) launch");
		if (large0Expected || largeUpperExpected) {
			// (level 0 and the levels above it are announced apart: the coarse levels of a terrain hold such blocks where level 0
			// has none, and each pair of launches costs ~10 us even when it finds nothing)
			const u32 ldsL0 = R0_TAB_LDS + sizeof(Reg0State<4096>), ldsL = REG_TAB_LDS + sizeof(RegStateT<4096>);
			const u32 upper = largeUpperExpected ? q.start[levels] - q.start[1] : 0u;
			const u32 grid0 = std::max<u32>(1u, std::min<u32>(q.start[1], (u32)cus));
			{
				// blocks beyond k_main's capacity class: table-driven like everywhere else (a general block of this size takes 45 us
				// where a table-driven one takes a third); the general pass only sees what those hand on
				if (large0Expected) {
					hipLaunchKernelGGL(k_dirty_regular0_fast<REG_CAP_BIG>, dim3(std::max<u32>(1u, std::min<u32>(q.start[1], (u32)cus * 2))), dim3(WG), F0_TAB_LDS + sizeof(Fast0State<REG_CAP_BIG>), stream, dev(p), (u32)REG_CAP_SMALL);
					hipLaunchKernelGGL((k_regular0<4096, 2>), dim3(grid0), dim3(WG), ldsL0, stream, dev(p), (u32)REG_CAP_SMALL);
				}
				if (upper) {
					hipLaunchKernelGGL(k_dirty_regular1_fast<REG_CAP_BIG>, dim3(std::min<u32>(upper, (u32)cus * 2)), dim3(WG), F0_TAB_LDS + sizeof(Fast1State<REG_CAP_BIG>), stream, dev(p), plan.fastEnd, (u32)REG_CAP_SMALL);
					// (also the blocks of a level without a lattice copy: k_main<true> hands every one of them on)
					hipLaunchKernelGGL((k_regular<4096, 2>), dim3(std::min<u32>(upper, (u32)cus)), dim3(WG), ldsL, stream, dev(p), 1u, levels, (u32)REG_CAP_SMALL);
				}
			}
			check(hipGetLastError(), "large-class launches (incremental)");
		}
		DirtyTailPlan t;
		memset(&t, 0, sizeof(t));
		// a general workgroup per block that could be handed on (zero samples on a carved surface are not rare: the ball brush puts
		// them wherever x^2 + y^2 + z^2 = r^2 has lattice solutions, and a general block takes 40-60 us): they all run side by side,
		// the ones that find nothing leave at once
		u32 upperAll = 0;
		for (u32 l = 1; l < levels; ++l) upperAll += q.start[l + 1] - q.start[l];
		t.wgs0 = std::max<u32>(8u, std::min<u32>(q.start[1], 192u)); t.wgs1 = std::max<u32>(8u, std::min<u32>(upperAll, 192u));
		(void)upperVol;
		t.gatherWgs = std::max<u32>(1u, std::min<u32>((q.start[levels] + 7u) / 8u, 64u));
		t.levels = levels; t.roleTicket = q.roleTicket; t.slowDone = q.slowDone;
		memcpy(t.start, q.start, sizeof(t.start));
		t.work = q.work; t.hostRecs = q.hostRecs; t.hostHeader = q.hostHeader; t.devHeader = q.header; t.headerWords = q.headerWords; t.publishedWord = q.publishedWord;
		const u32 lds = std::max<u32>(R0_TAB_LDS + sizeof(Reg0State<REG_CAP_SMALL>), REG_TAB_LDS + sizeof(RegStateT<REG_CAP_SMALL>));
		hipLaunchKernelGGL(k_dirty_tail, dim3(t.wgs0 + t.wgs1 + t.gatherWgs), dim3(WG), lds, stream, dev(p), t);
		check(hipGetLastError(), "k_dirty_tail launch");
	}

	// the same as the single-stream branch of run_overlapped_tail with stage marks in between (vx_set_stage_timing)
	template <typename P>
	void run_main_staged(const P& p, u32 levels)
	{
		spreadC = spreadD = false;
		run_main(p, levels, true);
		stage_mark(4);
		upperDone = level0Done = true;
		launch_regular(p, 0, levels, stream); // (records stage event 8 between its level-0 part and the rest)
		upperDone = level0Done = false;
		stage_mark(5);
		stage_mark(6);
	}

	// Overlapped tail of a full run (after classify + hierarchy on the main stream):
	//   side stream A : regular cells of level 0 (independent of the material caches)
	//   main stream   : material chain L1..Lmax, then regular cells of levels >= 1
	//   side stream B : transition cells (after the material chain)
	// The small, latency-bound material launches no longer leave the chip idle.
	template <typename P>
	void run_overlapped_tail(const P& p, u32 levels)
	{
		// (the classify launch carried the event that releases the level-0 regular pass on side stream A)
		overlappedTail = true;
		spreadC = spreadD = false;
		bool usedSideB = false;
		if (single_stream(p, levels)) {
			// one launch for the level-0 blocks and the levels >= 1 (vx_main.inl); what the table-driven passes leave - the upper
			// capacity classes, what they hand on, levels beyond the lattice copies - follows on the same stream, and so do the
			// block lists and the header read-back: no second stream, no event
			run_main(p, levels, true);
			upperDone = level0Done = true;
			tailWgs[0] = tailWgs[1] = 0;
			tailPending = tailDone && p.levels[0].listCounts;
			launch_regular(p, 0, levels, stream);
			upperDone = level0Done = false;
			overlappedTail = false;
			if (spreadC) (void)hipStreamWaitEvent(stream, evSideC, 0); // (dense surfaces: the upper capacity classes of level 0 ran beside k_main on a side stream)
			return;
		}
		(void)hipStreamWaitEvent(sideA, evClassified, 0);
		launch_regular(p, 0, 1, sideA);
		if (main_applies(p, levels)) {
			// the levels >= 1 as one launch: material blocks, regular blocks of the first capacity class and transition blocks
			// wait for each other through device-side flags (vx_main.inl); what is left for launch_regular below - the upper
			// capacity classes, what the table-driven pass hands on, levels beyond the lattice copies - follows on this stream
			run_main(p, levels, false);
			upperDone = true;
		} else {
			// the last material launch carries the event that releases the transition pass on side stream B
			u32 lastMat = 0;
			for (u32 L = 1; L < levels; ++L) if (p.levels[L].cap) lastMat = L;
			for (u32 L = 1; L < levels; ++L) { if (L == lastMat) doneEvent = evMaterial; run_material(p, L); }
			if (!lastMat) (void)hipEventRecord(evMaterial, stream);
			(void)hipStreamWaitEvent(sideB, evMaterial, 0);
			{
				hipStream_t keep = stream;
				stream = sideB;
				run_transition(p, levels);
				stream = keep;
			}
			(void)hipEventRecord(evSideB, sideB);
			usedSideB = true;
		}
		if (levels > 1) launch_regular(p, 1, levels, stream);
		upperDone = false;
		// What follows the three branches (block lists, header read-back) runs on side stream A: on large grids the level-0
		// pass there is the last to finish, and a stream that waits for events which have already fired loses nothing,
		// whereas the main stream would start ~15 us after the event it waits for (1024^3: 0.53 -> 0.51 ms).
		overlappedTail = false;
		(void)hipEventRecord(evMain, stream);
		(void)hipStreamWaitEvent(sideA, evMain, 0);
		if (usedSideB) (void)hipStreamWaitEvent(sideA, evSideB, 0);
		if (spreadC) (void)hipStreamWaitEvent(sideA, evSideC, 0);
		if (spreadD) (void)hipStreamWaitEvent(sideA, evSideD, 0);
		mainKeep = stream;
		stream = sideA;
	}
	// behind the host's wait for the run: the main stream is the current one again
	u32 head_partials() const { return headWorkgroups; }
	void end_overlapped()
	{
		if (mainKeep) { stream = mainKeep; mainKeep = nullptr; }
	}
	bool stage_timing_on() const { return stageOn; }
	bool run_selftest(u32* dOut)
	{
		if (!fill(dOut, 0, 16 * 4)) return false;
		hipLaunchKernelGGL(k_selftest, dim3((1u << 24) / WG), dim3(WG), 0, stream, dOut);
		return check(hipGetLastError(), "k_selftest launch");
	}

	// ---- halo messages of attached slabs (vx_halo_exchange*, vx_host.inl) -----------------------------------------------
	// both messages of a direction (pack: what goes below / above; unpack: what came from below / above) in one launch;
	// either may be absent.  `g` carries the brick mirrors an unpack keeps current (null pointers: none).
	void run_halo_moves(const HaloMove* lo, const HaloMove* hi, const GridView& g, const MirrorState& ms, bool alongY)
	{
		HaloPair pair;
		memset(&pair, 0, sizeof(pair));
		if (lo) pair.m[0] = *lo;
		if (hi) pair.m[1] = *hi;
		const HaloMove &a = pair.m[0], &b = pair.m[1];
		u32 rows[2] = { 0, 0 };
		for (u32 i = 0; i < a.count; ++i) rows[0] += (u32)a.piece[i].layers * a.piece[i].rows;
		for (u32 i = 0; i < b.count; ++i) rows[1] += (u32)b.piece[i].layers * b.piece[i].rows;
		const u32 most = std::max(rows[0], rows[1]);
		if (!most) return;
		hipLaunchKernelGGL(k_halo_move, dim3(most, 2), dim3(WG), 0, stream, pair, g, ms, alongY ? 1 : 0);
		check(hipGetLastError(), "k_halo_move launch");
	}
	void* comm = nullptr;
	static bool comm_unique_id(void* id)
	{
		std::string err;
		return rccl().load(err) && rccl().GetUniqueId((Rccl::UniqueId*)id) == 0;
	}
	bool rccl_ok(int rc, const char* what)
	{
		if (rc == 0) return true;
		lastError = std::string(what) + ": " + (rccl().GetErrorString ? rccl().GetErrorString(rc) : "RCCL error");
		return false;
	}
	bool comm_init(int nranks, int rank, const void* id)
	{
		if (!rccl().load(lastError)) return false;
		comm_destroy();
		Rccl::UniqueId uid;
		memcpy(&uid, id, sizeof(uid));
		return rccl_ok(rccl().CommInitRank(&comm, nranks, uid, rank), "ncclCommInitRank");
	}
	void comm_destroy()
	{
		if (comm) { (void)hipStreamSynchronize(stream); (void)rccl().CommDestroy(comm); comm = nullptr; }
	}
	// one grouped batch: everything a rank sends and receives in an exchange progresses together (xGMI links are
	// point to point: the two neighbours are two different links)
	bool comm_exchange(int peerLo, const void* sendLo, size_t sendLoBytes, void* recvLo, size_t recvLoBytes,
	                   int peerHi, const void* sendHi, size_t sendHiBytes, void* recvHi, size_t recvHiBytes)
	{
		if (!comm) { lastError = "no communicator"; return false; }
		enum { NCCL_UINT8 = 1 };
		bool ok = rccl_ok(rccl().GroupStart(), "ncclGroupStart");
		if (ok && peerLo >= 0) ok = rccl_ok(rccl().Send(sendLo, sendLoBytes, NCCL_UINT8, peerLo, comm, stream), "ncclSend") && rccl_ok(rccl().Recv(recvLo, recvLoBytes, NCCL_UINT8, peerLo, comm, stream), "ncclRecv");
		if (ok && peerHi >= 0) ok = rccl_ok(rccl().Send(sendHi, sendHiBytes, NCCL_UINT8, peerHi, comm, stream), "ncclSend") && rccl_ok(rccl().Recv(recvHi, recvHiBytes, NCCL_UINT8, peerHi, comm, stream), "ncclRecv");
		const bool ended = rccl_ok(rccl().GroupEnd(), "ncclGroupEnd");
		return ok && ended;
	}
	// in-process transport (several contexts driven by one process): a copy between two contexts' staging buffers
	bool copy_from_peer(void* dst, Backend& from, const void* src, size_t bytes)
	{
		if (from.device == device) return check(hipMemcpyAsync(dst, src, bytes, hipMemcpyDeviceToDevice, stream), "hipMemcpyAsync(D2D)");
		return check(hipMemcpyPeerAsync(dst, device, src, from.device, bytes, stream), "hipMemcpyPeerAsync");
	}

	// the result's block lists, written on the device behind the last kernel of a full run
	// the header's way to the host: with the list pass (see HeaderPublish).  Returns false when the caller has to copy.
	HeaderPublish publish = {};
	bool lists_publish_header(u32* hostDst, const u32* devHeader, u32 words, u32* doneCounter)
	{
		publish.host = hostDst; publish.dev = devHeader; publish.words = words; publish.done = doneCounter;
		return true;
	}
	template <typename P>
	bool run_block_lists(const P& p, const ListPlan& plan, u32 levels)
	{
		const u32 wgs = plan.wgStart[levels];
		const HeaderPublish pub = publish;
		publish = HeaderPublish();
		const bool tail = tailPending;
		tailPending = false;
		tailCleaned = false;
		if (tail) {
			// (with or without list workgroups: the general passes were left to this launch)
			TailPlan t;
			t.wgs0 = tailWgs[0]; t.wgs1 = tailWgs[1]; t.listWgs = wgs; t.levels = levels; t.slowDone = tailDone; t.roleTicket = tailDone + 1;
			t.next = nextReset;
			if (!wgs) t.next.header = nullptr;
			tailCleaned = t.next.header != nullptr;
			nextReset.header = nullptr;
			const u32 lds = std::max<u32>(R0_TAB_LDS + sizeof(Reg0State<REG_CAP_SMALL>), REG_TAB_LDS + sizeof(RegStateT<REG_CAP_SMALL>));
			if (t.wgs0 + t.wgs1 + wgs) hipLaunchKernelGGL(k_tail, dim3(t.wgs0 + t.wgs1 + wgs), dim3(WG), lds, stream, dev(p), plan, wgs ? pub : HeaderPublish(), t);
			check(hipGetLastError(), "k_tail launch");
			return wgs && pub.host != nullptr;
		}
		if (!wgs) return false;
		if (!p.levels[0].listCounts) hipLaunchKernelGGL(k_list_count, dim3(wgs), dim3(LIST_WG), 0, stream, dev(p), plan, levels);
		hipLaunchKernelGGL(k_list_write, dim3(wgs), dim3(LIST_WG), 0, stream, dev(p), plan, levels, pub);
		check(hipGetLastError(), "k_list launch");
		return pub.host != nullptr;
	}

	template <typename P>
	void run_transition(const P& p, u32 levels)
	{
		u32 cap = 0;
		for (u32 l = 1; l < levels; ++l) if (p.levels[l].hasTransitions) cap += p.levels[l].cap;
		if (!cap) return;
		// five resident workgroups per CU, each striding over its items (measured at 1024^3: 1280 workgroups 0.107 ms,
		// 1024: 0.117, 1536: 0.126, one workgroup per block: 0.109 - a workgroup's start costs about as much as its planes)
		const u32 grid = std::min<u32>(cap, (u32)cus * 5);
		const bool wide = (size_t)p.G.grid.n * p.G.grid.n * p.G.grid.n >= ((size_t)1 << 32) || tune.forceWide;
		if (wide) hipLaunchKernelGGL(k_transition<true>, dim3(grid), dim3(WG), TR_TAB_LDS + sizeof(TrState), stream, dev(p), levels);
		else hipLaunchKernelGGL(k_transition<false>, dim3(grid), dim3(WG), TR_TAB_LDS + sizeof(TrState), stream, dev(p), levels);
		check(hipGetLastError(), "k_transition launch");
	}
};

} // namespace

#include "vx_host.inl"

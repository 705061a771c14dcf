// vx_fast1.inl — k_regular1_fast: the regular-cell pass of the LOD levels 1..3 (the levels with a lattice copy) for
// blocks without a zero lattice sample (tv_fast1.h has the per-lane logic), written for gfx950.  Included by vx_hip.hip
// behind vx_fast0.inl, whose table staging, list building and scans it shares.
//
// f1_block is one block's work; k_regular1_fast walks the run's flat list of active blocks (Globals::flatItems: level, slot,
// coordinate and cell count in one load) with it - the capacity classes of dense surfaces - and k_main (vx_main.inl)
// calls it for the first class inside the one launch of the levels >= 1.  One workgroup per block at a time: 17^3 lattice samples (17 contiguous bytes per row of the level's lattice copy) + the
// material ids of the block's cache + bitmap into LDS (27 KB in all; nothing is prefetched: the workgroups beside it hide
// the latency of the voxel fetches around the vertices - and every KB held here is one the level-0 pass on the other
// stream cannot use) | compact cell list | table-driven cells | bases + reservations +
// descriptors | one lane = one vertex (LOD chain, then 17 fetches with per-axis address terms) and one lane = one
// triangle, both through streaming stores.  A block with a zero sample, or whose chains end on a voxel, goes to
// k_regular<.., 2> through Globals::slowItems[1]; blocks above the capacity class to the next one (`lo`).
namespace {

// voxel addresses in the brick mirrors as sums of one term per axis (tv_core.h brick_offset, split): 32-bit offsets,
// valid while a mirror is smaller than 4 GiB (grids up to 1024^3; the host launches the general pass beyond)
template <typename OFF>
struct BrickSamplerT {
	const i8* bDist;
	const u8* bMat;
	const u8* bBlend;
	int last;          // n - 1
	u32 nb, rowsY;     // blocks per row, resident block rows per block plane
	int yb0, zb0;      // first resident block row / plane
	typedef OFF Off;
	// (brick_local's bit fields as sums: x + 4080 * (x >> 4) = (x >> 4) << 12 | (x & 15); with y & 15 = 4a + b the tile bits are
	// 128 a + 16 b = 16 * ((y & 15) + (y & 12)); with z & 15 = 2a + b they are 512 a + 64 b = 64 * ((z & 15) + 3 * (z & 14)) - a multiply-add
	// where the shifts, masks and ORs of the literal form cost three to four instructions more per term, and a vertex of a
	// level >= 1 forms fifteen to twenty terms)
	__device__ __forceinline__ Off tx(int x) const { x = max(0, min(x, last)); return (Off)x + (Off)__umul24((u32)x >> 4, 4080u); }
	__device__ __forceinline__ Off ty(int y) const
	{
		y = max(0, min(y, last));
		const u32 l = (u32)y & 15u;
		return ((Off)__umul24((u32)((y >> 4) - yb0), nb) << 12) + ((l + (l & 12u)) << 4);
	}
	__device__ __forceinline__ Off tz(int z) const
	{
		z = max(0, min(z, last));
		const u32 l = (u32)z & 15u;
		return ((Off)__umul24(__umul24((u32)((z >> 4) - zb0), rowsY), nb) << 12) + ((l + 3u * (l & 14u)) << 6);
	}
	__device__ __forceinline__ int dist(Off o) const { return bDist[o]; }
	__device__ __forceinline__ u32 mat(Off o, int, int, int) const { return (u32)bMat[o] | ((u32)bBlend[o] << 8); }
};
typedef BrickSamplerT<u32> F1BrickSampler; // valid while a mirror is smaller than 4 GiB

// One block of a level 1..3 (CAP = LDS capacity class; `lo`: blocks with at most that many non-trivial cells belong to a lower
// class).  GATED (k_main): bitmap, cache block and cell count come from the material work of another workgroup of the same
// launch; the lattice samples, which do not, are requested before the wait.  Otherwise (k_regular1_fast) `ntc` and `coord`
// come from the run's flat list.
template <int CAP, bool GATED>
__device__ __forceinline__ void f1_block(const ExecParamsDev& p, const F0Tables& T, const F1BrickSampler& smp, Fast1State<CAP>& st, u32* wgStats, u32* zeroFlag, u32& parity,
                                         u32 level, u32 slot, u32 coord, u32 ntc, u32 lo, const int tid)
{
	typedef R0<CAP> K;
	const u32 lane = (u32)tid & 63u, wave = (u32)tid >> 6;
	const LevelDesc& L = p.levels[level];
	if (!GATED) {
		if (ntc > (u32)CAP || (lo && ntc <= lo)) return;    // another capacity class owns those (the first class, lo == 0, also owns the empty blocks)
		if (ntc == 0) {                                     // (uniform) a surface-bearing block without a non-trivial coarse cell
			if (tid == 0) reg_write_empty_record(L, slot);
			return;
		}
	}
	u32 bx, by, bz;
	block_coords(coord, L.cnt, bx, by, bz);
	__syncthreads(); // the previous block is done with the LDS state (and the tables are staged)
	if (tid == 0) st.suspect = 0;
	PyramidRow rows[2];
	{
		const PyramidLevel& P = p.G.pyr[level];
#pragma unroll
		for (int q = 0; q < 2; ++q) {
			const int r = min(tid + q * WG, 288), k = r / 17, j = r - k * 17;
			rows[q] = pyramid_row17(P, (int)(bx * 16), (int)(by * 16) + j, (int)(bz * 16) + k);
		}
	}
	// (bitmap and cache block: written by the material work - in k_main by another workgroup of the same launch, hence past the L1)
	const u16* csrc = L.cache + (size_t)slot * BLOCK_CELLS;
	u32 bitsWord = 0;
	uint4 c0, c1;
	if (GATED) {
		// the flag carries this run's tag or the wait goes on: no word of an older run is ever taken for the cell count
		if (tid == 0) st.zero = wait_done(L.matDone + slot, p.G.epoch, p.G.giveUp);
		acquire_and_meet(tid < 64);
		ntc = r0_uniform(st.zero);
		if (ntc > (u32)CAP || (lo && ntc <= lo)) return;
		if (ntc == 0) {
			if (tid == 0) reg_write_empty_record(L, slot);
			return;
		}
	}
	bitsWord = TV_LOAD_THROUGH(&L.ntBits[(size_t)slot * 128 + (tid & 127)]);
	c0 = load16_through(csrc, (u32)tid * 16u); c1 = load16_through(csrc, (u32)(tid + WG) * 16u);

	TRACE_MARK(0);
	// ---- stage: bitmap, material cache block, 17 x 17 rows of 17 lattice samples; any zero among them? -------------
	{
		u32 zero = 0;
		if (tid < 128) st.ntBits[tid] = bitsWord;
		if (tid < 16) st.classCount[tid] = 0;
		// (the ids = the low bytes of the 16-bit entries: lane tid holds the entries [8 tid, 8 tid + 8) and [8 (tid + 256), ...))
		((uint2*)st.cacheId)[tid] = make_uint2(__builtin_amdgcn_perm(c0.y, c0.x, 0x06040200u), __builtin_amdgcn_perm(c0.w, c0.z, 0x06040200u));
		((uint2*)st.cacheId)[tid + WG] = make_uint2(__builtin_amdgcn_perm(c1.y, c1.x, 0x06040200u), __builtin_amdgcn_perm(c1.w, c1.z, 0x06040200u));
#pragma unroll
		for (int q = 0; q < 2; ++q) {
			const int r = tid + q * WG;
			if (r < 289) {
				const int k = r / 17, j = r - k * 17;
				u32* dst = (u32*)(st.samp + k * F1_SPLANE + j * F1_SROW);
				dst[0] = rows[q].lo.x; dst[1] = rows[q].lo.y; dst[2] = rows[q].lo.z; dst[3] = rows[q].lo.w; dst[4] = rows[q].far;
				zero |= f0_has_zero_byte(rows[q].lo.x) | f0_has_zero_byte(rows[q].lo.y) | f0_has_zero_byte(rows[q].lo.z) | f0_has_zero_byte(rows[q].lo.w) | ((rows[q].far & 0xFFu) == 0u ? 1u : 0u);
			}
		}
		if (__ballot(zero != 0) && lane == 0) zeroFlag[parity] = 1;
		if (tid == 0) zeroFlag[parity ^ 1u] = 0; // last read behind the previous block's second barrier
	}
	__syncthreads();
	TRACE_MARK(1);
	const bool clean = r0_uniform(zeroFlag[parity]) == 0;
	parity ^= 1u;
	if (VX_ABL & 8192) return;
	if (!clean) {
		if (tid == 0) p.G.slowItems[1][atomicAdd(&p.G.slowCount[1], 1u)] = (level << 24) | slot;
		return;
	}

	// ---- popcount prefix of the bitmap (every wave computes all of it: no exchange), compact cell list ----------
	{
		const u32 w0 = st.ntBits[lane], w1 = st.ntBits[lane + 64];
		const u32 c0 = (u32)__popc(w0), c1 = (u32)__popc(w1);
		const u32 i0 = wave_inclusive_scan_dpp(c0);
		const u32 half = (u32)__shfl((int)i0, 63, 64);
		const u32 i1 = wave_inclusive_scan_dpp(c1) + half;
		const u32 e0 = i0 - c0, e1 = i1 - c1;
		if (wave == 0) {
			st.wordPrefix[lane] = (u16)e0; st.wordPrefix[lane + 64] = (u16)e1;
			if (lane == 63) st.wordPrefix[128] = (u16)i1;
		}
		const int src = (tid >> 1) & 63;
		const u32 wLo = (u32)__shfl((int)w0, src, 64), wHi = (u32)__shfl((int)w1, src, 64);
		const u32 eLo = (u32)__shfl((int)e0, src, 64), eHi = (u32)__shfl((int)e1, src, 64);
		const u32 word = (wave >= 2) ? wHi : wLo;
		u32 kk = (wave >= 2) ? eHi : eLo;
		u32 bits = word & 0xFFFFu;
		if (tid & 1) { kk += (u32)__popc(bits); bits = word >> 16; }
		while (bits) {
			const u32 x = (u32)__builtin_ctz(bits);
			bits &= bits - 1;
			st.cellAN[kk++][0] = (u32)(tid * 16) + x;
		}
	}
	__syncthreads();

	TRACE_MARK(2);
	// ---- cells: wave w owns the compact cells [w * Q, w * Q + Q), Q a multiple of 64; local scan per wave -----------
	const u32 nt = r0_uniform(st.wordPrefix[128]);
	const u32 Q = ((nt + WG - 1) / WG) * 64u;
	const u32 kBeg = r0_uniform(min(wave * Q, nt)), kEnd = r0_uniform(min(wave * Q + Q, nt));
	{
		u32 carry = 0;
		for (u32 k0 = kBeg; k0 < kEnd; k0 += 64u) {
			const u32 k = k0 + lane;
			u32 cnt = 0;
			if (k < kEnd) {
				cnt = fx_cell<F1Layout>(st, T, k, st.classCount);
#if defined(VX_CASE_DUMP)
				L.caseDump[(size_t)slot * BLOCK_CELLS + (st.cellAN[k][0] & 0xFFFu)] = (u8)((st.cellAN[k][0] >> 12) & 0xFFu);
#endif
			}
			const u32 incl = wave_inclusive_scan_dpp(cnt);
			if (k < kEnd) st.cellC[k] = carry + incl - cnt;
			carry += (u32)__shfl((int)incl, 63, 64);
		}
		if (lane == 0) st.waveTot[wave] = carry;
	}
	__syncthreads();
	TRACE_MARK(3);
	{
		u32 waveBase = 0, tot = 0;
#pragma unroll
		for (u32 w = 0; w < (u32)(WG / 64); ++w) {
			const u32 s = st.waveTot[w];
			if (w < wave) waveBase += s;
			tot += s;
		}
		if (tid == WG - 1) {
			const u32 vTotal = tot & 0xFFFFu, tTotal = tot >> 16;
			st.vTotal = vTotal; st.tTotal = tTotal;
			reserve_both(p.P.cursors, vTotal, tTotal * 3u, st.vOff, st.iOff);
		}
		for (u32 k0 = kBeg; k0 < kEnd; k0 += 64u) {
			const u32 k = k0 + lane;
			if (k < kEnd) {
				const u32 base = st.cellC[k] + waveBase;
				st.cellC[k] = base;
				f0_describe(st, T, k, base, 0u, 0u);
			}
		}
	}
	__syncthreads();

	TRACE_MARK(4);
	const u32 vTotalU = r0_uniform(st.vTotal), tTotalU = r0_uniform(st.tTotal);
	const bool room = r0_uniform(st.vOff) + vTotalU <= p.P.vertCap && r0_uniform(st.iOff) + tTotalU * 3u <= p.P.idxCap;
	const int ox = (int)(bx * 16 * L.mult), oy = (int)(by * 16 * L.mult), oz = (int)(bz * 16 * L.mult);
	if (room) {
		u32 notInterior = 0;
		for (u32 chunk = 0; chunk == 0 || chunk * F1_VDESC < vTotalU || chunk * F1_TDESC < tTotalU; ++chunk) {
			const u32 cv = chunk * F1_VDESC, ct = chunk * F1_TDESC;
			if (chunk) {
				__syncthreads();
				for (u32 k = (u32)tid; k < nt; k += WG) f0_describe(st, T, k, st.cellC[k], cv, ct);
				__syncthreads();
			}
			const u32 vEnd = cv < vTotalU ? min(vTotalU - cv, (u32)F1_VDESC) : 0u;
			const u32 tEnd = ct < tTotalU ? min(tTotalU - ct, (u32)F1_TDESC) : 0u;
			PolyVertex* vOut = p.P.verts + r0_uniform(st.vOff) + cv;
			u32* iOut = p.P.idx + r0_uniform(st.iOff) + ct * 3u;
			for (u32 base = 0; base < vEnd || base < tEnd; base += WG) {
				const u32 j = base + (u32)tid;
				if (j < vEnd && !(VX_ABL & 32)) {
					const u32 desc = st.vdesc[j];
					const unsigned long long lut = K::lut_row_waterfall(p.G.lut, (u32)st.cacheId[desc & 0xFFFu]);
					if (!f1_vertex(st, T, smp, L.cache + (size_t)slot * BLOCK_CELLS, desc, (int)level, ox, oy, oz, lut, vOut + j)) notInterior = 1;
				}
				if (j < tEnd && !(VX_ABL & 64)) {
					u32 ids[3];
					f0_triangle(st, T, j, ids);
					u32* o3 = iOut + j * 3u;
					TV_STREAM_STORE(&o3[0], ids[0]); TV_STREAM_STORE(&o3[1], ids[1]); TV_STREAM_STORE(&o3[2], ids[2]);
				}
			}
		}
		if (__ballot(notInterior != 0) && lane == 0) st.suspect = 1;
	}
	TRACE_MARK(5);
	__syncthreads();
	if (tid == 0) {
		if (st.suspect) {
			// a chain ended on a voxel: the general pass (degenerate-triangle filter) redoes the block; the ranges reserved
			// above stay unused
			st.suspect = 0;
			p.G.slowItems[1][atomicAdd(&p.G.slowCount[1], 1u)] = (level << 24) | slot;
			atomicAdd(&p.G.slowCount[2], st.vTotal); atomicAdd(&p.G.slowCount[3], st.tTotal * 3u); // dead pool ranges (reported, vx_exec_info)
		} else {
			BlockRecord& r = L.records[slot];
			r.coordId = coord;
			r.vOff = st.vOff; r.vCount = room ? st.vTotal : 0; r.iOff = st.iOff; r.iCount = room ? st.tTotal * 3u : 0;
			count_listed_block(L, r.coordId, r.vCount);
			if (!L.hasTransitions) for (int f = 0; f < 6; ++f) { r.tvOff[f] = 0; r.tvCount[f] = 0; r.tiOff[f] = 0; r.tiCount[f] = 0; }
			r.degenerate = 0;
			r.ntCells = nt;
			r.pad = 0;
			if (!room) atomicOr(&p.P.cursors[CUR_OVF], 1u);
			wgStats[0] += nt;
			for (int i = 0; i < 16; ++i) wgStats[4 + i] += st.classCount[i];
		}
	}
}

template <int CAP>
__global__ __launch_bounds__(WG) __attribute__((amdgpu_waves_per_eu(5))) void k_regular1_fast(ExecParamsDev p, u32 levelEnd, u32 lo)
{
	if (lo && *p.G.largeBlocks == 0) return; // nothing for the capacity classes above the first (uniform over the grid)
	typedef Fast1State<CAP> ST;
	u8* tab = smem;
	ST& st = *(ST*)(smem + F0_TAB_LDS);
	__shared__ u32 wgStats[20];
	__shared__ u32 zeroFlag[2];

	const int tid = (int)threadIdx.x;
	if (tid < 20) wgStats[tid] = 0;
	if (tid < 2) zeroFlag[tid] = 0;
	const F0Tables T = f0_stage_tables(tab, p.tables); // visible after the first barrier of the item loop
	// the work items are the first entries of the run's list of active blocks of the levels >= 1 (Globals::flatItems, in
	// level order): those of the levels below levelEnd.  Counts and the item's entry come in one round trip.
	u32 total = 0;
	for (u32 l = 1; l < levelEnd; ++l) total += p.G.slotCounts[l];
	total = r0_uniform(total);
	if (blockIdx.x >= ((total + 63u) & ~63u)) return; // the grid is sized before the block counts are known
	const GridView& g = p.G.grid;
	const F1BrickSampler smp = { g.bDist, g.bMat, g.bBlend, g.n - 1, (u32)g.n >> 4, (u32)g.bRowsY, g.bYb0, g.bZb0 };
	u32 parity = 0;

	for (u32 it = blockIdx.x; it < ((total + 63u) & ~63u); it += gridDim.x) {
		const u32 item = xcd_item(it);
		if (item >= total) continue;
		const FlatItem fi = p.G.flatItems[item];
		f1_block<CAP, false>(p, T, smp, st, wgStats, zeroFlag, parity, r0_uniform(fi.where >> 24), r0_uniform(fi.where & 0xFFFFFFu), r0_uniform(fi.coordId), r0_uniform(fi.ntCells), lo, tid);
	}
	__syncthreads();
	if (tid < 20 && wgStats[tid]) atomicAdd(&p.G.stats[tid], wgStats[tid]);
}

// The same blocks' work over the work lists of an incremental run (Globals::workItems, levels 1 .. levelEnd - 1): the capacity
// classes above the first behind k_main<true>, which leaves blocks beyond its class alone.  Bitmaps, cache blocks and cell
// counts are what k_main<true>'s material blocks wrote in the launch before.  What a block hands on (a zero lattice sample, a
// chain that ends on a voxel) goes to k_regular<4096, 2> through Globals::slowItems[1] like everywhere.
template <int CAP>
__global__ __launch_bounds__(WG) __attribute__((amdgpu_waves_per_eu(5))) void k_dirty_regular1_fast(ExecParamsDev p, u32 levelEnd, u32 lo)
{
	if (*p.G.largeBlocks == 0) return; // nothing beyond the first class (uniform over the grid)
	typedef Fast1State<CAP> ST;
	u8* tab = smem;
	ST& st = *(ST*)(smem + F0_TAB_LDS);
	__shared__ u32 wgStats[20];
	__shared__ u32 zeroFlag[2];
	__shared__ WorkList wl;

	const int tid = (int)threadIdx.x;
	if (tid < 20) wgStats[tid] = 0;
	if (tid < 2) zeroFlag[tid] = 0;
	if (tid == 0) {
		u32 run = 0;
		for (u32 l = 0; l <= MAX_LEVELS; ++l) { wl.start[l] = run; if (l >= 1 && l < levelEnd) run += p.G.workCount[l]; }
	}
	const F0Tables T = f0_stage_tables(tab, p.tables); // visible after the first barrier of the item loop
	__syncthreads();
	const u32 total = r0_uniform(wl.start[MAX_LEVELS]);
	const GridView& g = p.G.grid;
	const F1BrickSampler smp = { g.bDist, g.bMat, g.bBlend, g.n - 1, (u32)g.n >> 4, (u32)g.bRowsY, g.bYb0, g.bZb0 };
	u32 parity = 0;
	for (u32 it = blockIdx.x; it < total; it += gridDim.x) {
		u32 level, idx;
		decode_item(wl, levelEnd, it, level, idx);
		const u32 slot = p.G.workItems[level][idx];
		const LevelDesc& L = p.levels[level];
		f1_block<CAP, false>(p, T, smp, st, wgStats, zeroFlag, parity, r0_uniform(level), r0_uniform(slot), r0_uniform(L.slotCoord[slot]), r0_uniform((u32)L.ntCount[slot]), lo, tid);
	}
	__syncthreads();
	if (tid < 20 && wgStats[tid]) atomicAdd(&p.G.stats[tid], wgStats[tid]);
}

} // namespace

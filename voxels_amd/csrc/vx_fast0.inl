// vx_fast0.inl — k_regular0_fast: the regular-cell pass of level 0 for blocks without a zero sample (tv_fast0.h has the
// per-lane logic and the reason why it is exact), written for gfx950.  Included by vx_hip.hip behind vx_regular0.inl,
// whose work distribution it shares: persistent workgroups stride over the level-0 slots, the next block's 19^3
// distances, 17^3 materials + blends and bitmap are requested into registers while the current block emits.  A block in whose staged samples a lane finds a zero byte is appended to Globals::slowItems[0] and left to
// k_regular0<.., 2>, launched right behind on the same stream (or to the first workgroups of k_tail).  f0_walk is the body:
// k_regular0_fast strides over all slots, k_main walks batches of consecutive slots taken from its queue - and there,
// where no classification pass ran, a block forms its own non-trivial bitmap first (SELF: f0_self_bits, one barrier more).
//
// Per block: deposit (+ zero test) | popcount prefix + compact cell list | cells: table-driven, no loop over table
// vertices, wave-contiguous ranges, DPP scans | bases + pool reservations + vertex / triangle descriptors | one lane =
// one vertex and one lane = one triangle (three indices).  5 barriers.
namespace {

#if defined(VX_F0_PROFILE)
// tools builds: where a level-0 block's time goes inside f0_walk (cycles / 64 as thread 0 sees them, summed over all blocks)
__device__ unsigned long long g_f0prof[12];
#define F0_TICK(i) do { const unsigned long long now_ = __builtin_readcyclecounter(); if (tid == 0) atomicAdd(&g_f0prof[i], (now_ - f0Tick) >> 6); f0Tick = now_; } while (0)
#else
#define F0_TICK(i) do { } while (0)
#endif

constexpr u32 F0_TAB_FIXED = TAB_F0_BYTES - TAB_F0_CASE;   // case rows | triangle rows | edge infos | direction masks
constexpr u32 F0_TAB_LDS = F0_TAB_FIXED + 2048;            // + the per-case vertex rows widened to 8 bytes

__device__ __forceinline__ F0Tables f0_stage_tables(u8* lds, const u8* image, u32 tid)
{
	copy16(lds, image + TAB_F0_CASE, F0_TAB_FIXED, tid);
	for (u32 i = tid; i < 256; i += WG) {
		const u8* src = image + TAB_REG_VERT + i * 6;
		unsigned long long row = 0;
#pragma unroll
		for (int b = 0; b < 6; ++b) row |= (unsigned long long)src[b] << (8 * b);
		*(unsigned long long*)(lds + F0_TAB_FIXED + i * 8) = row;
	}
	return f0_tables_from_image(lds - TAB_F0_CASE, (const unsigned long long*)(lds + F0_TAB_FIXED));
}
__device__ __forceinline__ F0Tables f0_stage_tables(u8* lds, const u8* image) { return f0_stage_tables(lds, image, threadIdx.x); }

// inclusive scan over the wave with DPP row shifts and row broadcasts (gfx9): no LDS traffic, six dependent adds
__device__ __forceinline__ u32 wave_inclusive_scan_dpp(u32 v)
{
	int x = (int)v;
	x += __builtin_amdgcn_update_dpp(0, x, 0x111, 0xF, 0xF, false); // row_shr:1
	x += __builtin_amdgcn_update_dpp(0, x, 0x112, 0xF, 0xF, false); // row_shr:2
	x += __builtin_amdgcn_update_dpp(0, x, 0x114, 0xF, 0xF, false); // row_shr:4
	x += __builtin_amdgcn_update_dpp(0, x, 0x118, 0xF, 0xF, false); // row_shr:8
	x += __builtin_amdgcn_update_dpp(0, x, 0x142, 0xA, 0xF, false); // row_bcast:15 into rows 1 and 3
	x += __builtin_amdgcn_update_dpp(0, x, 0x143, 0xC, 0xF, false); // row_bcast:31 into rows 2 and 3
	return (u32)x;
}

__device__ __forceinline__ u32 f0_has_zero_byte(u32 v) { return (v - 0x01010101u) & ~v & 0x80808080u; }

// What a workgroup holds of the NEXT block while it finishes the current one.  Lanes are mapped so that most of a
// row's address is a per-lane constant of the block:
//   distances: lane = (jj = tid % 19, group = tid / 19), rows (jj, kk = group + 13 q), q = 0, 1 — the 19 samples x = -1..17
//              of row (y, z) = (by * 16 - 1 + jj, bz * 16 - 1 + kk): the row's 16 bytes in its own brick + the dwords left and
//              right of them in the x-neighbour bricks (bricks of x-neighbour blocks follow each other);
//   materials: lane = (j = tid % 17, group = tid / 17), rows (j, k = group + 15 q), q = 0, 1 — samples 0..16 of the
//              material AND the blend row (same offset in both mirrors).
struct F0Prefetch {
	uint4 d[2]; u32 dl[2], dr[2];
	uint4 m[2], b[2]; u32 mf[2], bf[2];
	u32 bits;         // one word of the non-trivial bitmap (lanes < 128)
};

__device__ __forceinline__ int f0_opaque_tid()
{
	int t = (int)threadIdx.x;
	asm volatile("" : "+v"(t)); // keeps the lane's task decomposition from being hoisted out of the block loop (registers)
	return t;
}

// byte offset, relative to the block's own brick, of the brick row that holds voxel row (y, z) given as block-relative
// coordinates ry, rz in [-1, 17] (clamped to the grid by the caller through lo / hi = the valid range of ry / rz)
__device__ __forceinline__ int f0_row_part(int r, int lo, int hi, int brickStride, bool isZ)
{
	r = max(lo, min(r, hi));
	const int q = r >> 4, l = r & 15; // q in {-1, 0, 1}
	return q * brickStride + (isZ ? (((l >> 1) << 9) | ((l & 1) << 6)) : (((l >> 2) << 7) | ((l & 3) << 4)));
}

template <bool SELF = false>
__device__ __forceinline__ void f0_request(const GridView& g, const LevelDesc& L, const R0Block& b, F0Prefetch& pf)
{
	const int tid = f0_opaque_tid();
	const int n = g.n, cnt = (int)L.cnt;
	if (VX_ABL & 4096) { pf.d[0] = pf.d[1] = pf.m[0] = pf.m[1] = pf.b[0] = pf.b[1] = make_uint4(0x01010101u, 0x01010101u, 0x01010101u, 0x81818181u); pf.dl[0] = pf.dl[1] = pf.dr[0] = pf.dr[1] = pf.mf[0] = pf.mf[1] = pf.bf[0] = pf.bf[1] = 0x01010101u; pf.bits = 0; return; }
	// every lane loads (indices clamped into range): a conditional load with a default value makes the compiler wait for
	// the data right behind the load, and these requests must stay in flight
	if (!SELF) pf.bits = L.ntBits[(size_t)b.slot * 128 + (tid & 127)];
	const int brickRow = (n >> 4) * BRICK_BYTES, brickPlane = g.bRowsY * brickRow; // next block along y / z
	const size_t own = brick_base(g, (int)b.bx, (int)b.by, (int)b.bz);
	const bool firstX = b.bx == 0, lastX = (int)b.bx + 1 == cnt;
	// block-relative coordinates that stay inside the grid (the reference clamps every fetch, :1194-1201)
	const int yLo = -min((int)b.by * 16, 1), yHi = min(n - 1 - (int)b.by * 16, 17), zLo = -min((int)b.bz * 16, 1), zHi = min(n - 1 - (int)b.bz * 16, 17);
	{
		const i8* base = g.bDist + own;
		const int jj = min(tid % 19, 18), group = tid / 19;
		const int yPart = f0_row_part(jj - 1, yLo, yHi, brickRow, false);
#pragma unroll
		for (int q = 0; q < 2; ++q) {
			const int kk = min(group + 13 * q, 18);
			const int row = yPart + f0_row_part(kk - 1, zLo, zHi, brickPlane, true);
			pf.d[q] = *(const uint4*)(base + row);
			pf.dl[q] = *(const u32*)(base + row + (firstX ? 0 : 12 - BRICK_BYTES));
			pf.dr[q] = *(const u32*)(base + row + (lastX ? 12 : BRICK_BYTES));
		}
	}
	{
		const u8* mbase = g.bMat + own;
		const u8* bbase = g.bBlend + own;
		const int j = min(tid % 17, 16), group = tid / 17;
		const int yPart = f0_row_part(j, 0, yHi, brickRow, false);
#pragma unroll
		for (int q = 0; q < 2; ++q) {
			const int k = min(group + 15 * q, 16);
			const int row = yPart + f0_row_part(k, 0, zHi, brickPlane, true);
			const int far = row + (lastX ? 12 : BRICK_BYTES); // sample 16 in byte 0; at the grid's edge the row's last dword (see deposit)
			pf.m[q] = *(const uint4*)(mbase + row);
			pf.b[q] = *(const uint4*)(bbase + row);
			pf.mf[q] = *(const u32*)(mbase + far);
			pf.bf[q] = *(const u32*)(bbase + far);
		}
	}
}

// registers -> LDS, and: does a staged distance sample equal zero?  (The rows y, z = -1 and 17 are looked at as well: a
// zero there sends a block to the general pass without need, which costs time, never correctness.)
// SELF: nobody classified the block; the sign masks of its 19 x 19 sample rows (bit i = sample x = i, 0..16) are left in LDS
// (on top of the vertex descriptors, which are written two barriers later) for f0_self_bits.
template <bool SELF = false, typename ST>
__device__ __forceinline__ u32 f0_deposit(ST& st, const LevelDesc& L, const R0Block& b, const F0Prefetch& pf)
{
	const int tid = f0_opaque_tid();
	const bool firstX = b.bx == 0, lastX = b.bx + 1 == L.cnt;
	u32 zero = 0;
	if (!SELF) { if (tid < 128) st.ntBits[tid] = pf.bits; }
	else if (tid == 0) st.zero = 0; // (the block's count of non-trivial cells is summed up here)
	{
		const int jj = tid % 19, group = tid / 19;
#pragma unroll
		for (int q = 0; q < 2; ++q) {
			const int kk = group + 13 * q;
			if (kk < 19 && tid < 19 * 13) {
				u32 left = pf.dl[q], right = pf.dr[q];
				if (firstX) left = pf.d[q].x << 24;                       // no left neighbour: sample -1 = sample 0
				if (lastX) right = (pf.d[q].w >> 24) * 0x01010101u;       // no right neighbour: samples 16, 17 = sample 15
				u32* dst = (u32*)(st.samp + kk * SPLANE + jj * SROW);       // bytes 0..3: x = -4..-1, 4..19: x = 0..15, 20..23: x = 16..19
				dst[0] = left; dst[1] = pf.d[q].x; dst[2] = pf.d[q].y; dst[3] = pf.d[q].z; dst[4] = pf.d[q].w; dst[5] = right;
				zero |= f0_has_zero_byte(pf.d[q].x) | f0_has_zero_byte(pf.d[q].y) | f0_has_zero_byte(pf.d[q].z) | f0_has_zero_byte(pf.d[q].w) | f0_has_zero_byte(right | 0xFFFFFF00u);
				if (SELF) ((u32*)st.vdesc)[kk * 19 + jj] = sign_nibble(pf.d[q].x) | (sign_nibble(pf.d[q].y) << 4) | (sign_nibble(pf.d[q].z) << 8) | (sign_nibble(pf.d[q].w) << 12) | (((right >> 7) & 1u) << 16);
			}
		}
	}
	{
		const int j = tid % 17, group = tid / 17;
#pragma unroll
		for (int q = 0; q < 2; ++q) {
			const int k = group + 15 * q;
			if (k < 17 && tid < 17 * 15) {
				u32* dm = (u32*)(st.matId + k * F0_MPLANE + j * F0_MROW);
				u32* db = (u32*)(st.blend + k * F0_MPLANE + j * F0_MROW);
				dm[0] = pf.m[q].x; dm[1] = pf.m[q].y; dm[2] = pf.m[q].z; dm[3] = pf.m[q].w; dm[4] = lastX ? (pf.mf[q] >> 24) : pf.mf[q];
				db[0] = pf.b[q].x; db[1] = pf.b[q].y; db[2] = pf.b[q].z; db[3] = pf.b[q].w; db[4] = lastX ? (pf.bf[q] >> 24) : pf.bf[q];
			}
		}
	}
	return zero;
}

// The bitmap of a block nobody classified (k_run_head gave it its slot because its samples are not of one sign): lane
// (y, z) = (tid & 15, tid >> 4) forms the 16 non-trivial bits of its cell row from the sign masks f0_deposit left, exactly
// as k_classify does (a cell is non-trivial unless its eight corners agree in sign), keeps them for this walk and for the
// passes behind it (the general pass, incremental runs, the parent's vote) and adds its count to st.zero.
template <typename ST>
__device__ __forceinline__ void f0_self_bits(ST& st, const LevelDesc& L, const R0Block& b, const int tid)
{
	static_assert(sizeof(st.vdesc) >= 361 * 4, "the row masks lie on top of the vertex descriptors");
	const u32* rows = (const u32*)st.vdesc;
	const int y = tid & 15, z = tid >> 4;
	const int r00 = (z + 1) * 19 + (y + 1);
	const u32 a = rows[r00], b2 = rows[r00 + 1], c = rows[r00 + 19], d = rows[r00 + 20];
	const u32 A = a & b2 & c & d, O = a | b2 | c | d;
	const u32 nt = ((O | (O >> 1)) & ~(A & (A >> 1))) & 0xFFFFu;
	((u16*)st.ntBits)[tid] = (u16)nt;
	((u16*)(L.ntBits + (size_t)b.slot * 128))[tid] = (u16)nt;
	((u16*)(L.consBits + (size_t)b.slot * 128))[tid] = (u16)nt; // (a block the emptiness rule skips never gets here: f0_accept)
	u32 cnt = (u32)__popc(nt);
#pragma unroll
	for (int o = 32; o > 0; o >>= 1) cnt += (u32)__shfl_xor((int)cnt, o, 64);
	if ((tid & 63) == 0 && cnt) atomicAdd(&st.zero, cnt);
}

// The work list of a walk over level-0 slots: items first, first + stride, ... below `limit`.  REMAP (k_regular0_fast: static
// striding over all slots): item numbers go through xcd_item and `limit` is the padded slot count; otherwise (k_main: a batch
// of consecutive slots taken from the queue) an item is its slot.
template <bool REMAP>
__device__ __forceinline__ R0Candidate f0_peek(const ExecParamsDev& p, const LevelDesc& L, u32 total, u32 limit, u32 it)
{
	R0Candidate c;
	u32 item = REMAP ? xcd_item(it) : it;
	c.valid = (it < limit && item < total) ? 1u : 0u;
	if (!c.valid) item = 0;
	c.slot = item;
	if (!REMAP && p.G.dirty) c.slot = p.G.workItems[0][item]; // (uniform; k_main of an incremental run: the queue hands out entries of the level's work list)
	c.ntc = L.ntCount[c.slot];
	c.skip = L.skip[c.slot];
	c.coord = L.slotCoord[c.slot];
	return c;
}

template <int CAP, bool REMAP, bool SELF = false>
__device__ __forceinline__ bool f0_next(const ExecParamsDev& p, const LevelDesc& L, u32 total, u32 lo, u32 stride, u32 limit, u32& it, R0Candidate c, R0Block& b)
{
	for (;;) {
		if (SELF && r0_uniform(c.valid) && r0_uniform(c.skip)) {
			// a block the emptiness rule skips (its empty record follows in r0_accept): no cells, nothing for its parent's vote
			const u32 slot = r0_uniform(c.slot);
			((u16*)(L.ntBits + (size_t)slot * 128))[threadIdx.x] = 0;
			((u16*)(L.consBits + (size_t)slot * 128))[threadIdx.x] = 0;
			if (threadIdx.x == 0) L.ntCount[slot] = 0;
		}
		if (r0_accept<CAP>(L, lo, c, b)) return true;
		it += stride;
		if (it >= limit) return false;
		c = f0_peek<REMAP>(p, L, total, limit, it);
	}
}

// The blocks of one walk, software-pipelined: `cur` has its inputs requested, `nxt` is accepted and gets them requested while
// `cur` writes its output.  Leaves the LDS state free behind a barrier-less tail (callers meet before they reuse it).
template <int CAP, bool REMAP, bool SELF = false>
__device__ __forceinline__ void f0_walk(const ExecParamsDev& pIn, const F0Tables& T, Fast0State<CAP>& st, u32* wgStats, u32* zeroFlag, u32& parity,
                                        u32 total, u32 lo, u32 first, u32 stride, u32 limit, const int tid)
{
	typedef R0<CAP> K;
#define F0_PARAMS() (void)pIn; const ExecParamsDev& p = kernarg_params(); const LevelDesc& L = p.levels[0]; const GridView& g = p.G.grid; (void)L; (void)g
	F0_PARAMS();
	u32 it = first;
	R0Block cur, nxt;
	F0Prefetch pf;
	// as in k_regular0: `cur` has its inputs requested, `nxt` is accepted and gets them requested while `cur` writes its output
	bool have = f0_next<CAP, REMAP, SELF>(p, L, total, lo, stride, limit, it, f0_peek<REMAP>(p, L, total, limit, it), cur);
	if (have) f0_request<SELF>(g, L, cur, pf);
	it += stride;
	bool haveNext = have && f0_next<CAP, REMAP, SELF>(p, L, total, lo, stride, limit, it, f0_peek<REMAP>(p, L, total, limit, it), nxt);
#if defined(VX_F0_PROFILE)
	unsigned long long f0Tick = __builtin_readcyclecounter();
#endif
	while (have) {
		F0_PARAMS();
		const u32 candIt = it + stride;
		R0Candidate cand;
		F0_TICK(0); // (between blocks: next-item bookkeeping)
		__syncthreads(); // the previous block is done with the LDS state (and the tables are staged)
		F0_TICK(1);
		{
			const u32 z = f0_deposit<SELF>(st, L, cur, pf);
			if (__ballot(z != 0) && (tid & 63) == 0) zeroFlag[parity] = 1;
			if (tid == 0) zeroFlag[parity ^ 1u] = 0; // last read behind the previous block's second barrier
		}
		__syncthreads();
		F0_TICK(2); // deposit (waits for the prefetched loads) + barrier
		const bool clean = r0_uniform(zeroFlag[parity]) == 0;
		parity ^= 1u;
		bool requested = false;
		bool mine = !(VX_ABL & 65536); // (SELF: the block turns out to have geometry, and no more cells than this capacity class holds; tools: 65536 = stop behind the deposit)
		if (SELF && mine) {
			f0_self_bits(st, L, cur, tid);
			__syncthreads();
			const u32 cells = r0_uniform(st.zero);
			if (tid == 0) {
				L.ntCount[cur.slot] = (u16)cells;
				if (cells == 0) reg_write_empty_record(L, cur.slot);
				if (cells > (u32)LARGE_THRESHOLD) atomicAdd(p.G.largeBlocks, 1u); // (the host repeats the run with the upper classes)
			}
			mine = cells != 0 && cells <= (u32)CAP && !(VX_ABL & 131072); // (tools: 131072 = stop behind the own bitmap)
			F0_TICK(3); // own bitmap + barrier
		}
		if (mine && clean) {
			// ---- popcount prefix of the bitmap (every wave computes all of it: no exchange), compact cell list ----------
			{
				const int lane = tid & 63, wave = tid >> 6;
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
				// row tid = cells [tid * 16, tid * 16 + 16): half of word tid >> 1, held by lane (tid >> 1) & 63 of either set
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
			F0_TICK(4); // prefix + compact list + barrier
			if (!(VX_ABL & 262144)) { // (tools: stop behind the compact list)

			// ---- cells: wave w owns the compact cells [w * Q, w * Q + Q), Q a multiple of 64; local scan per wave -----------
			const u32 nt = r0_uniform(st.wordPrefix[128]);
			const u32 Q = ((nt + WG - 1) / WG) * 64u;
			const u32 lane = (u32)tid & 63u, wave = (u32)tid >> 6;
			const u32 kBeg = r0_uniform(min(wave * Q, nt)), kEnd = r0_uniform(min(wave * Q + Q, nt));
			{
				u32 carry = 0;
				for (u32 k0 = kBeg; k0 < kEnd; k0 += 64u) {
					const u32 k = k0 + lane;
					u32 cnt = 0;
					if (k < kEnd) {
						cnt = f0_cell(st, T, k, wgStats + 4);
#if defined(VX_CASE_DUMP)
						L.caseDump[(size_t)cur.slot * BLOCK_CELLS + (st.cellAN[k][0] & 0xFFFu)] = (u8)((st.cellAN[k][0] >> 12) & 0xFFu);
#endif
					}
					const u32 incl = wave_inclusive_scan_dpp(cnt);
					if (k < kEnd) st.cellC[k] = carry + incl - cnt;
					carry += (u32)__shfl((int)incl, 63, 64);
				}
				if (lane == 0) st.waveTot[wave] = carry;
			}
			__syncthreads();
			F0_TICK(5); // cells + barrier
			if (!(VX_ABL & 524288)) { // (tools: stop behind the cells)
			{
				u32 waveBase = 0, tot = 0;
#pragma unroll
				for (u32 w = 0; w < (u32)(WG / 64); ++w) {
					const u32 s = st.waveTot[w];
					if (w < wave) waveBase += s;
					tot += s;
				}
				const u32 vTotal = tot & 0xFFFFu, tTotal = tot >> 16;
				// both pool reservations are requested now; their results are first needed after the descriptors are written.  The
				// last lane makes them: its wave owns the tail of the compact list and is the first to run out of cells.
				if (tid == WG - 1) {
					st.vTotal = vTotal; st.tTotal = tTotal;
					if (VX_ABL & 32768) { st.vOff = (cur.slot * 701u) % (p.P.vertCap - 4096u); st.iOff = (cur.slot * 4001u) % (p.P.idxCap - 16384u); } // (tools: what the returning atomic costs)
					else reserve_both(p.P.cursors, vTotal, tTotal * 3u, st.vOff, st.iOff);
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
			F0_TICK(6); // bases + reservation + descriptors + barrier

			const u32 vTotalU = r0_uniform(st.vTotal), tTotalU = r0_uniform(st.tTotal);
			const int ox = (int)(cur.bx * 16), oy = (int)(cur.by * 16), oz = (int)(cur.bz * 16);
			const bool room = r0_uniform(st.vOff) + vTotalU <= p.P.vertCap && r0_uniform(st.iOff) + tTotalU * 3u <= p.P.idxCap;
			// Vertices and indices leave in ONE loop, and the next block's inputs are requested inside its first trip (a loop
			// with stores that is entered while loads are in flight makes the compiler drain the memory queue in front of it).
			if (room) {
				for (u32 chunk = 0; chunk == 0 || chunk * F0_VDESC < vTotalU || chunk * F0_TDESC < tTotalU; ++chunk) {
					const u32 cv = chunk * F0_VDESC, ct = chunk * F0_TDESC;
					if (chunk) {
						__syncthreads();
						for (u32 k = (u32)tid; k < nt; k += WG) f0_describe(st, T, k, st.cellC[k], cv, ct);
						__syncthreads();
					}
					const u32 vEnd = cv < vTotalU ? min(vTotalU - cv, (u32)F0_VDESC) : 0u;
					const u32 tEnd = ct < tTotalU ? min(tTotalU - ct, (u32)F0_TDESC) : 0u;
					PolyVertex* vOut = p.P.verts + r0_uniform(st.vOff) + cv;
					u32* iOut = p.P.idx + r0_uniform(st.iOff) + ct * 3u;
					for (u32 base = 0; base < vEnd || base < tEnd; base += WG) {
						if (!requested) { if (haveNext) f0_request<SELF>(g, L, nxt, pf); cand = f0_peek<REMAP>(p, L, total, limit, candIt); requested = true; }
						const u32 j = base + (u32)tid;
						if (j < vEnd && !(VX_ABL & 1)) {
							const u32 desc = st.vdesc[j], c = desc & 0xFFFu;
							const u32 cellId = st.matId[(c >> 8) * F0_MPLANE + ((c >> 4) & 15u) * F0_MROW + (c & 15u)];
							f0_vertex(st, T, desc, ox, oy, oz, K::lut_row_waterfall(p.G.lut, cellId), vOut + j);
						}
						if (j < tEnd && !(VX_ABL & 2)) {
							u32 ids[3];
							f0_triangle(st, T, j, ids);
							u32* o3 = iOut + j * 3u; // 12 bytes per lane, consecutive lanes consecutive triangles
							TV_STREAM_STORE(&o3[0], ids[0]); TV_STREAM_STORE(&o3[1], ids[1]); TV_STREAM_STORE(&o3[2], ids[2]);
						}
					}
				}
			}
			F0_TICK(7); // vertices + triangles (+ the next block's requests)
			if (tid == 0) {
				BlockRecord& r = L.records[cur.slot];
				r.coordId = cur.coord;
				r.vOff = st.vOff; r.vCount = room ? st.vTotal : 0; r.iOff = st.iOff; r.iCount = room ? st.tTotal * 3u : 0;
				count_listed_block(L, r.coordId, r.vCount);
				if (!L.hasTransitions) for (int f = 0; f < 6; ++f) { r.tvOff[f] = 0; r.tvCount[f] = 0; r.tiOff[f] = 0; r.tiCount[f] = 0; }
				r.degenerate = 0;
				r.ntCells = nt;
				r.pad = 0;
				if (!room) atomicOr(&p.P.cursors[CUR_OVF], 1u);
				wgStats[0] += nt;
			}
			F0_TICK(8); // record
			} }
		} else if (mine && tid == 0) {
			// a zero sample: the general pass takes the block
			p.G.slowItems[0][atomicAdd(&p.G.slowCount[0], 1u)] = cur.slot;
		}
		if (!requested) { if (haveNext) f0_request<SELF>(g, L, nxt, pf); cand = f0_peek<REMAP>(p, L, total, limit, candIt); }
		cur = nxt;
		have = haveNext;
		it = candIt;
		haveNext = have && f0_next<CAP, REMAP, SELF>(p, L, total, lo, stride, limit, it, cand, nxt);
	}
}

template <int CAP>
__global__ __launch_bounds__(WG) __attribute__((amdgpu_waves_per_eu(4))) void k_regular0_fast(ExecParamsDev p, u32 lo)
{
	if (lo && *p.G.largeBlocks == 0) return; // nothing for the capacity classes above the first (uniform over the grid)
	static_assert(F0_MROW == R0_MROW, "the prefetch of vx_regular0.inl stages 20-byte material rows");
	typedef Fast0State<CAP> ST;
	u8* tab = smem;
	ST& st = *(ST*)(smem + F0_TAB_LDS);
	__shared__ u32 wgStats[20];  // statistics of every block this workgroup handles, flushed once at the end
	__shared__ u32 zeroFlag[2];  // "a lane met a zero sample", by parity of the workgroup's block counter

	const LevelDesc& L = p.levels[0];
	const u32 total = r0_uniform(*L.nActive);
	if (blockIdx.x >= ((total + 63u) & ~63u)) return; // the grid is sized before the block counts are known
	const int tid = (int)threadIdx.x;
	if (tid < 20) wgStats[tid] = 0;
	if (tid < 2) zeroFlag[tid] = 0;
	const F0Tables T = f0_stage_tables(tab, p.tables);

	u32 parity = 0;
	f0_walk<CAP, true>(p, T, st, wgStats, zeroFlag, parity, total, lo, blockIdx.x, gridDim.x, (total + 63u) & ~63u, tid);
	__syncthreads();
	if (tid < 20 && wgStats[tid]) atomicAdd(&p.G.stats[tid], wgStats[tid]);
}

// The capacity classes above the first over the level-0 work list of an incremental run (Globals::workItems[0]), behind
// k_main<true>, which leaves blocks beyond its class alone: bitmaps and cell counts are k_dirty_head's.  A block with a zero
// sample goes to k_regular0<4096, 2> through Globals::slowItems[0].
template <int CAP>
__global__ __launch_bounds__(WG) __attribute__((amdgpu_waves_per_eu(4))) void k_dirty_regular0_fast(ExecParamsDev p, u32 lo)
{
	if (*p.G.largeBlocks == 0) return; // nothing beyond the first class (uniform over the grid)
	typedef Fast0State<CAP> ST;
	u8* tab = smem;
	ST& st = *(ST*)(smem + F0_TAB_LDS);
	__shared__ u32 wgStats[20];
	__shared__ u32 zeroFlag[2];

	const LevelDesc& L = p.levels[0];
	(void)L;
	const u32 total = r0_uniform(p.G.workCount[0]);
	if (blockIdx.x >= total) return;
	const int tid = (int)threadIdx.x;
	if (tid < 20) wgStats[tid] = 0;
	if (tid < 2) zeroFlag[tid] = 0;
	const F0Tables T = f0_stage_tables(tab, p.tables);

	u32 parity = 0;
	f0_walk<CAP, false>(p, T, st, wgStats, zeroFlag, parity, total, lo, blockIdx.x, gridDim.x, total, tid);
	__syncthreads();
	if (tid < 20 && wgStats[tid]) atomicAdd(&p.G.stats[tid], wgStats[tid]);
}

} // namespace

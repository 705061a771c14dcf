// vx_regular0.inl — the regular-cell pass of level 0 (PolygonizeBlock, src/TransVoxelImpl.cpp:1529-1750, for the
// surface-bearing level-0 blocks), written for gfx950.  Included by vx_hip.hip.
//
// Same closed form and the same per-cell arithmetic as the portable phases of tv_block.h (which the CPU emulation runs
// and which levels >= 1 still use on the GPU); what differs is how a workgroup gets through a block:
//   * the next block's inputs (19^3 distance neighbourhood, 17^3 material + blend samples, non-trivial bitmap) are
//     requested into registers while the current block emits its index list, so no phase waits for HBM;
//   * materials and blends of the block live in LDS: no phase after staging touches global memory except the LUT row
//     of a vertex and the output stores;
//   * cell classification, reuse resolution (TransVoxelImpl.cpp:1579-1718), degenerate filter (:1300-1321) and the
//     vertex / index counts are one pass over the compact cell list; cells whose samples are all non-zero — every cell
//     of an ordinary surface — take a short path: a crossed edge with non-zero end samples is seen identically by the
//     neighbour that owns it, so "the neighbour's reuse slot is valid" needs no lookup, only the material test does;
//   * one packed scan gives vertex and index offsets, both pool reservations are in flight together, vertices and
//     indices are emitted in the same phase.
// About 7 workgroup barriers per block instead of 23.
namespace {

constexpr int R0_MROW = 20;                  // bytes per staged material row: samples 0..16 of the row + padding
constexpr int R0_MPLANE = 17 * R0_MROW;
constexpr int R0_MBYTES = 17 * R0_MPLANE;
constexpr int R0_VDESC = 1024;               // vertex descriptors per chunk
constexpr int R0_IDESC = 2048;               // index descriptors per chunk
constexpr u32 R0_TAB_LDS = 512 + 64 + 2048 + 256; // regClass + regCell | edge words | regVert rows (8 bytes each) | regOwn

template <int CAP>
struct Reg0State {
	i8 samp[SAMP_BYTES + 8];                                  // tv_block.h layout: 19 x 19 rows of 24 bytes
	__attribute__((aligned(4))) u8 matId[R0_MBYTES + 4];      // material id of voxel (i,j,k), 0..16 from the block origin
	__attribute__((aligned(4))) u8 blend[R0_MBYTES + 4];
	u32 ntBits[128];
	u16 wordPrefix[132];
	u32 cellA[CAP];        // compact: cell id | case code << 12 | (corner sample == 0) mask << 20
	u32 cellB[CAP];        // compact: created vertices (12) | of those, placed at corner v0 (12) << 12 | kept triangles (5) << 24 | any INVALID vertex << 31
	u32 cellC[CAP];        // compact: created vertex count | kept index count << 16; after the scan: vertex base | index base << 16
	u16 ords[CAP];         // compact: ordinal (4 bits) of the vertex stored in each of the 4 reuse slots
	u16 invalidMask[CAP];  // compact: table vertices that resolve to INVALID_INDEX
	u16 vdesc[R0_VDESC];   // new vertices of the chunk: compact cell | table vertex << 12
	u16 idesc[R0_IDESC];   // indices of the chunk: compact cell | (triangle * 3 + corner) << 12
	u32 waveTot[4];
	u32 vOff, iOff, vTotal, iTotal, degenerate;
};

__device__ __forceinline__ int r0_samp_off(int corner) { return (corner & 1) + ((corner >> 1) & 1) * SROW + (corner >> 2) * SPLANE; }
__device__ __forceinline__ int r0_mat_off(int corner) { return (corner & 1) + ((corner >> 1) & 1) * R0_MROW + (corner >> 2) * R0_MPLANE; }

// tables in LDS: like stage_regular_tables, the per-case vertex rows widened to 8 bytes (one ds_read_b64 per case)
struct R0Tables {
	const u8* cls;        // 256
	const u8* cell;       // 16 x 16
	const u16* edge;      // 16 edge words
	const unsigned long long* vert; // 256 rows: 12 edge nibbles
	const u8* own;        // 256
	__device__ __forceinline__ u32 nibble(unsigned long long row, u32 vi) const { return (u32)(row >> (vi * 4u)) & 15u; }
};

__device__ __forceinline__ R0Tables r0_stage_tables(u8* lds, const u8* image)
{
	copy16(lds, image + TAB_REG_CLASS, 512);
	copy16(lds + 512, image + TAB_REG_EDGE, 64);
	for (u32 i = threadIdx.x; i < 256; i += WG) {
		const u8* src = image + TAB_REG_VERT + i * 6;
		unsigned long long row = 0;
#pragma unroll
		for (int b = 0; b < 6; ++b) row |= (unsigned long long)src[b] << (8 * b);
		*(unsigned long long*)(lds + 576 + i * 8) = row;
	}
	copy16(lds + 2624, image + TAB_REG_OWN, 256);
	R0Tables T;
	T.cls = lds; T.cell = lds + 256; T.edge = (const u16*)(lds + 512); T.vert = (const unsigned long long*)(lds + 576); T.own = lds + 2624;
	return T;
}

// tv_core.h view of the same LDS image, for the rare paths that run the portable per-cell code
__device__ __forceinline__ Tables r0_portable_tables(const u8* lds, const u8* packedRows)
{
	Tables T;
	T.regClassP = lds; T.regCellP = lds + 256; T.regEdgeP = (const u16*)(lds + 512); T.regVertP = packedRows; T.regOwnP = lds + 2624;
	T.trClassP = nullptr; T.trCornerP = nullptr; T.trCellP = nullptr; T.trVertP = nullptr; T.trEdgeP = nullptr; T.trOwnP = nullptr;
	return T;
}

// what a workgroup holds of the NEXT block while it finishes the current one
struct R0Prefetch {
	uint2 dOwn[3];    // distance half rows: the 8 bytes that lie in the row's own brick ...
	u32 dNb[3];       // ... and the 4 bytes from the x-neighbour brick (left of half 0, right of half 1)
	u32 bits;         // one word of the non-trivial bitmap (lanes < 128)
	uint4 m[3];       // material / blend rows, samples 0..15
	u32 mf[3];        // ... sample 16 (in byte 0)
};

struct R0Block {
	u32 slot, coord, bx, by, bz, ntc;
};

template <int CAP>
struct R0 {
	typedef Reg0State<CAP> ST;

	// ---- requests for block b (nothing is waited for here) --------------------------------------------------------
	// The lane's task decomposition (row numbers etc.) depends on the thread id alone; left alone the compiler computes it
	// once per kernel and keeps ~20 registers alive across the whole block loop (they end up in scratch, and every reload
	// drains the memory queue).  An opaque copy of the id makes it recompute the few values where they are used.
	static __device__ __forceinline__ int opaque_tid()
	{
		int t = (int)threadIdx.x;
		asm volatile("" : "+v"(t));
		return t;
	}

	static __device__ __forceinline__ void request(const GridView& g, const LevelDesc& L, const R0Block& b, R0Prefetch& pf)
	{
		const int tid = opaque_tid();
		const int n = g.n, cnt = (int)L.cnt;
		// every lane loads (indices clamped into range): a conditional load with a default value makes the compiler wait for
		// the data right behind the load, and these requests must stay in flight
		pf.bits = L.ntBits[(size_t)b.slot * 128 + (tid & 127)];
		// Everything comes from the brick mirrors (tv_core.h): a voxel row of a block is 16 contiguous bytes, 8 rows share a
		// 128-byte line.  Distance rows r = kk * 19 + jj: voxels [bx*16 - 4, bx*16 + 20) of row (y,z) = (by*16 - 1 + jj,
		// bz*16 - 1 + kk), clamped, as two halves of 12 bytes: 8 bytes of the row's own brick + 4 of the x-neighbour brick
		// (bricks of x-neighbour blocks follow each other).  At the grid's first / last block the missing neighbour is
		// replaced by the row itself and patched in deposit().
		const int brickRow = (n >> 4) * BRICK_BYTES, brickPlane = g.bRowsY * brickRow; // next block along y / z
		const size_t own = brick_base(g, (int)b.bx, (int)b.by, (int)b.bz);
		const bool firstX = b.bx == 0, lastX = (int)b.bx + 1 == cnt;
		{
			const i8* base = g.bDist + own;
#pragma unroll
			for (int q = 0; q < 3; ++q) {
				const int h = min(tid + q * WG, 721);
				const int r = h >> 1, half = h & 1;
				const int kk = r / 19, jj = r - kk * 19;
				const int y = clampi((int)b.by * 16 - 1 + jj, 0, n - 1), z = clampi((int)b.bz * 16 - 1 + kk, 0, n - 1);
				const int row = ((z >> 4) - (int)b.bz) * brickPlane + ((y >> 4) - (int)b.by) * brickRow + (int)brick_local(0u, (u32)y & 15u, (u32)z & 15u);
				pf.dOwn[q] = *(const uint2*)(base + row + half * 8);
				pf.dNb[q] = *(const u32*)(base + row + (half ? (lastX ? 12 : BRICK_BYTES) : (firstX ? 0 : 12 - BRICK_BYTES)));
			}
		}
		// material / blend rows: samples 0..16 of row (j,k), j,k = 0..16, clamped at the far side of the grid
		{
			const u8* mbase = g.bMat + own;
			const u8* bbase = g.bBlend + own;
#pragma unroll
			for (int q = 0; q < 3; ++q) {
				const int t = min(tid + q * WG, 577);
				const int arr = t >= 289 ? 1 : 0, r = t - arr * 289;
				const int k = r / 17, j = r - k * 17;
				const int y = min((int)b.by * 16 + j, n - 1), z = min((int)b.bz * 16 + k, n - 1);
				const int row = ((z >> 4) - (int)b.bz) * brickPlane + ((y >> 4) - (int)b.by) * brickRow + (int)brick_local(0u, (u32)y & 15u, (u32)z & 15u);
				const u8* src = (arr ? bbase : mbase) + row;
				pf.m[q] = *(const uint4*)src;
				pf.mf[q] = *(const u32*)(src + (lastX ? 12 : BRICK_BYTES)); // sample 16 in byte 0; at the grid's edge the row's last dword (see deposit)
			}
		}
	}

	// ---- registers -> LDS ---------------------------------------------------------------------------------------------
	static __device__ __forceinline__ void deposit(ST& st, const GridView& g, const LevelDesc& L, const R0Block& b, const R0Prefetch& pf)
	{
		const int tid = opaque_tid();
		const bool firstX = b.bx == 0, lastX = b.bx + 1 == L.cnt;
		if (tid < 128) st.ntBits[tid] = pf.bits;
#pragma unroll
		for (int q = 0; q < 3; ++q) {
			const int h = tid + q * WG;
			if (h < 722) {
				const int r = h >> 1, half = h & 1;
				u32 a = half ? pf.dOwn[q].x : pf.dNb[q], bb = half ? pf.dOwn[q].y : pf.dOwn[q].x, c = half ? pf.dNb[q] : pf.dOwn[q].y;
				if (!half && firstX) a = bb << 24;                          // no left neighbour: sample -1 = sample 0
				if (half && lastX) c = (bb >> 24) * 0x01010101u;            // no right neighbour: samples 16, 17 = sample 15
				u32* dst = (u32*)(st.samp + r * SROW + half * 12);
				dst[0] = a; dst[1] = bb; dst[2] = c;
			}
		}
#pragma unroll
		for (int q = 0; q < 3; ++q) {
			const int t = tid + q * WG;
			if (t < 578) {
				const int arr = t >= 289 ? 1 : 0, r = t - arr * 289;
				const int k = r / 17, j = r - k * 17;
				u32* dst = (u32*)((arr ? st.blend : st.matId) + k * R0_MPLANE + j * R0_MROW);
				dst[0] = pf.m[q].x; dst[1] = pf.m[q].y; dst[2] = pf.m[q].z; dst[3] = pf.m[q].w;
				dst[4] = lastX ? (pf.mf[q] >> 24) : pf.mf[q];
			}
		}
	}

	// level-0 central differences at a staged sample (tv_core.h normal_at over the LDS neighbourhood)
	static __device__ __forceinline__ void normal_lds(const i8* p, float out[3])
	{
		out[0] = (float)((int)p[1] - (int)p[-1]) * 0.5f;
		out[1] = (float)((int)p[SPLANE] - (int)p[-SPLANE]) * 0.5f;
		out[2] = (float)((int)p[SROW] - (int)p[-SROW]) * 0.5f;
		normalize_fix_zero(out);
	}

	// reuse slot of a neighbour cell, recomputed from its samples (only cells with a zero sample ask)
	struct Neighbour {
		const ST* st;
		const Tables* T;
		int cx, cy, cz;
		__device__ __forceinline__ void operator()(int dx, int dy, int dz, u32 slot, bool& valid, u32& mat) const
		{
			const int x = cx - dx, y = cy - dy, z = cz - dz;
			const u32 c = (u32)((z << 8) | (y << 4) | x);
			valid = false; mat = 0;
			if (!bit_get(st->ntBits, c)) return;
			i8 V[8];
			reg_cell_values(st->samp, x, y, z, V);
			valid = ((reg_slot_valid(*T, reg_zero_mask(V), reg_case_code(V)) >> slot) & 1u) != 0;
			mat = st->matId[z * R0_MPLANE + y * R0_MROW + x];
		}
	};

	struct LdsMaterials {
		const ST* st;
		int ox, oy, oz;
		__device__ __forceinline__ u32 operator()(int, const int P[3]) const
		{
			const int o = (P[2] - oz) * R0_MPLANE + (P[1] - oy) * R0_MROW + (P[0] - ox);
			return (u32)st->matId[o] | ((u32)st->blend[o] << 8);
		}
	};

	// ---- one compact cell: case, reuse resolution, degenerate filter, counts ---------------------------------------
	static __device__ __forceinline__ void cell(ST& st, const R0Tables& RT, const Tables& T, const LevelDesc& L, const R0Block& b, u32 k, u32* wgStats)
	{
		const u32 c = st.cellA[k] & 0xFFFu;
		const int cx = (int)(c & 15), cy = (int)((c >> 4) & 15), cz = (int)(c >> 8);
		const i8* sp = st.samp + samp_index(cx, cy, cz);
		i8 V[8];
		V[0] = sp[0]; V[1] = sp[1]; V[2] = sp[SROW]; V[3] = sp[SROW + 1];
		V[4] = sp[SPLANE]; V[5] = sp[SPLANE + 1]; V[6] = sp[SPLANE + SROW]; V[7] = sp[SPLANE + SROW + 1];
		const u32 code = reg_case_code(V), zeroMask = reg_zero_mask(V);
#if defined(VX_CASE_DUMP)
		L.caseDump[(size_t)b.slot * BLOCK_CELLS + c] = (u8)code;
#endif
		const int mo = cz * R0_MPLANE + cy * R0_MROW + cx;
		const u32 myMat = st.matId[mo];
		const u32 cls = RT.cls[code];
		const u32 geom = RT.cell[cls * 16];
		const u32 nv = geom >> 4, ntri = geom & 15u;
		atomicAdd(&wgStats[4 + cls], 1u);
		// reuseValidityMask from the bitmap (reg_mask3 of tv_block.h, with k = rank of this cell)
		const u32 row = (u32)((cz << 4) | cy);
		const u32 word = st.ntBits[row >> 1];
		const u32 rowBits = (word >> ((row & 1u) * 16u)) & 0xFFFFu;
		const u32 below = rowBits & ((1u << cx) - 1u);
		const u32 sliceBase = st.wordPrefix[cz * 8];
		u32 mask3 = below ? 1u : 0u;
		if (k - (u32)__popc(below) - sliceBase) mask3 |= 2u;
		if (sliceBase) mask3 |= 4u;
		const unsigned long long vrow = RT.vert[code];
		u32 count = 0, ords = 0, newMask = 0, atV0 = 0, invalid = 0;
		u32 keepMask = (1u << ntri) - 1u, kept = ntri;
		if (!zeroMask) {
			// every vertex lies strictly inside its edge: owned edges (direction 8) are created and stored; any other edge is
			// reused iff the mask allows its direction and the owner cell has this cell's material (its slot is valid
			// because it sees the same two non-zero samples), else created without being stored.
			for (u32 vi = 0; vi < nv; ++vi) {
				const u32 w = RT.edge[RT.nibble(vrow, vi)];
				const u32 dir = w >> 12, slot = (w >> 8) & 15u;
				bool create = true;
				if (dir == 8u) ords = (ords & ~(0xFu << (slot * 4u))) | (count << (slot * 4u));
				else if ((dir & mask3) == dir) {
					const u32 nm = st.matId[mo - (int)(dir & 1u) - (int)((dir >> 1) & 1u) * R0_MROW - (int)(dir >> 2) * R0_MPLANE];
					create = nm != myMat;
				}
				if (create) { newMask |= 1u << vi; ++count; }
			}
		} else {
			const Neighbour nb{ &st, &T, cx, cy, cz };
			for (u32 vi = 0; vi < nv; ++vi) {
				const u32 w = RT.edge[RT.nibble(vrow, vi)];
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
			// a vertex on a corner can make a triangle degenerate (PushBlocksToResult's filter); cells without a zero sample
			// keep all their triangles: three vertices strictly inside three distinct edges of a unit cell
			CellGeom geo;
			geo.mult = 1; geo.level = 0;
			geo.local[0] = cx; geo.local[1] = cy; geo.local[2] = cz;
			geo.base[0] = (int)(b.bx * 16) + cx; geo.base[1] = (int)(b.by * 16) + cy; geo.base[2] = (int)(b.bz * 16) + cz;
			const LocalDist d{ st.samp, (int)(b.bx * 16), (int)(b.by * 16), (int)(b.bz * 16) };
			const u8* cd = T.regCell(cls);
			keepMask = 0; kept = 0;
			for (u32 tr = 0; tr < ntri; ++tr) {
				const u32 a = cd[1 + tr * 3], bb = cd[2 + tr * 3], cc = cd[3 + tr * 3];
				bool keepIt = true;
				if (!(((invalid >> a) | (invalid >> bb) | (invalid >> cc)) & 1u)) {
					float pa[3], pb[3], pc[3];
					reg_vertex_position(d, geo, V, T.regVert(code, a), ((atV0 >> a) & 1u) != 0, pa);
					reg_vertex_position(d, geo, V, T.regVert(code, bb), ((atV0 >> bb) & 1u) != 0, pb);
					reg_vertex_position(d, geo, V, T.regVert(code, cc), ((atV0 >> cc) & 1u) != 0, pc);
					keepIt = !triangle_degenerate(pa, pb, pc);
				}
				if (keepIt) { keepMask |= 1u << tr; ++kept; }
			}
			if (kept != ntri) atomicAdd(&st.degenerate, ntri - kept);
		}
		st.cellA[k] = c | (code << 12) | (zeroMask << 20);
		st.cellB[k] = newMask | (atV0 << 12) | (keepMask << 24) | (invalid ? 0x80000000u : 0u);
		st.cellC[k] = count | ((kept * 3u) << 16);
		st.ords[k] = (u16)ords;
		st.invalidMask[k] = (u16)invalid;
	}

	// descriptors of cell k's new vertices and kept triangle corners that fall into the given chunks
	static __device__ __forceinline__ void describe(ST& st, u32 k, u32 base, u32 chunkV, u32 chunkI)
	{
		const u32 bWord = st.cellB[k];
		u32 m = bWord & 0xFFFu;
		u32 j = base & 0xFFFFu;
		if (j < chunkV + R0_VDESC && j + 12 > chunkV) {
			while (m) {
				const u32 vi = (u32)__builtin_ctz(m);
				m &= m - 1;
				if (j >= chunkV && j < chunkV + R0_VDESC) st.vdesc[j - chunkV] = (u16)(k | (vi << 12));
				++j;
			}
		}
		u32 keepMask = (bWord >> 24) & 31u;
		u32 pos = base >> 16;
		if (pos < chunkI + R0_IDESC && pos + 15 > chunkI) {
			while (keepMask) {
				const u32 tr = (u32)__builtin_ctz(keepMask);
				keepMask &= keepMask - 1;
#pragma unroll
				for (u32 e = 0; e < 3; ++e, ++pos)
					if (pos >= chunkI && pos < chunkI + R0_IDESC) st.idesc[pos - chunkI] = (u16)(k | ((tr * 3 + e) << 12));
			}
		}
	}

	// LUT row of a material id through the scalar cache.  The memory counter of the vector loads retires in order, so a
	// vector load issued (and consumed) behind the prefetch requests would make its wave wait for all of them; scalar loads
	// have their own counter.  A block has one or two material ids: the loop over distinct ids runs once or twice.
	static __device__ __forceinline__ unsigned long long lut_row_waterfall(const u8* lut, u32 id)
	{
		unsigned long long row = 0;
		bool pending = true;
		while (pending) {
			const u32 u = (u32)__builtin_amdgcn_readfirstlane((int)id);
			unsigned long long r;
			asm volatile("s_load_dwordx2 %0, %1, %2\n\ts_waitcnt lgkmcnt(0)" : "=s"(r) : "s"(lut), "s"(u * 8u) : "memory");
			if (id == u) { row = r; pending = false; }
		}
		return row;
	}

	// ---- one lane = one new vertex of the chunk --------------------------------------------------------------------
	static __device__ __forceinline__ void emit_vertex(const ST& st, const R0Tables& RT, const Globals& G, const R0Block& b, u32 j, PolyVertex* out)
	{
		const int ox = (int)(b.bx * 16), oy = (int)(b.by * 16), oz = (int)(b.bz * 16);
		{
			const u32 desc = st.vdesc[j];
			const u32 k = desc & 0xFFFu, vi = desc >> 12;
			const u32 a = st.cellA[k];
			const u32 c = a & 0xFFFu, code = (a >> 12) & 0xFFu;
			const int cx = (int)(c & 15), cy = (int)((c >> 4) & 15), cz = (int)(c >> 8);
			const u32 w = RT.edge[RT.nibble(RT.vert[code], vi)];
			const int v0 = (int)((w >> 4) & 15u), v1 = (int)(w & 15u);
			const i8* sp = st.samp + samp_index(cx, cy, cz);
			const int mo = cz * R0_MPLANE + cy * R0_MROW + cx;
			const u32 cellMat = (u32)st.matId[mo] | ((u32)st.blend[mo] << 8);
			const unsigned long long lut = lut_row_waterfall(G.lut, cellMat & 0xFFu);
			const i8* s0 = sp + r0_samp_off(v0);
			const i8* s1 = sp + r0_samp_off(v1);
			const int val0 = *s0, val1 = *s1;
			RawVertex rv;
			if (val0 != 0 && val1 != 0) {
				// strictly inside the edge (reg_edge_vertex of tv_core.h at level 0: no LOD chain, no boundary flags)
				const int t = edge_t(val0, val1), u = 256 - t;
				const float ft = (float)t, fu = (float)u;
				const int p0x = ox + cx + (v0 & 1), p0y = oy + cy + ((v0 >> 1) & 1), p0z = oz + cz + (v0 >> 2);
				const int p1x = ox + cx + (v1 & 1), p1y = oy + cy + ((v1 >> 1) & 1), p1z = oz + cz + (v1 >> 2);
				rv.p[0] = ft * (float)p0x + fu * (float)p1x;
				rv.p[1] = ft * (float)p0y + fu * (float)p1y;
				rv.p[2] = ft * (float)p0z + fu * (float)p1z;
				float N0[3], N1[3];
				normal_lds(s0, N0);
				normal_lds(s1, N1);
				const int m0 = mo + r0_mat_off(v0), m1 = mo + r0_mat_off(v1);
				const u32 M0 = (u32)st.matId[m0] | ((u32)st.blend[m0] << 8), M1 = (u32)st.matId[m1] | ((u32)st.blend[m1] << 8);
				if ((M0 & 0xFF) == (M1 & 0xFF) && (M0 & 0xFF) == (cellMat & 0xFF)) rv.mat = (M0 & 0xFF) | (lerp_blend(t, u, M0 >> 8, M1 >> 8) << 8);
				else rv.mat = cellMat;
				const float wt = ft / 256.f, wu = fu / 256.f;
				rv.n[0] = N0[0] * wt + N1[0] * wu; rv.n[1] = N0[1] * wt + N1[1] * wu; rv.n[2] = N0[2] * wt + N1[2] * wu;
				normalize_fix_zero(rv.n);
				rv.flags = 0;
				rv.s[0] = rv.p[0]; rv.s[1] = rv.p[1]; rv.s[2] = rv.p[2];
			} else {
				// on a corner (GenerateVertexFromPoint, :1450-1467)
				const int e = edge_end(val0, val1);
				const u32 atV0 = (st.cellB[k] >> 12) & 0xFFFu;
				const int corner = ((atV0 >> vi) & 1u) ? v0 : ((e == 0) ? v1 : v0);
				CellGeom geo;
				geo.mult = 1; geo.level = 0;
				geo.local[0] = cx; geo.local[1] = cy; geo.local[2] = cz;
				geo.base[0] = ox + cx; geo.base[1] = oy + cy; geo.base[2] = oz + cz;
				const LocalDist d{ st.samp, ox, oy, oz };
				reg_corner_vertex(d, LdsMaterials{ &st, ox, oy, oz }, geo, corner, cellMat, rv);
			}
			pack_vertex_row(rv, lut, out + j);
		}
	}

	// ---- one lane = one index of the chunk ----------------------------------------------------------------------------
	static __device__ __forceinline__ void flush_index(const ST& st, const R0Tables& RT, u32 j, u32* out)
	{
		{
			const u32 desc = st.idesc[j];
			const u32 k = desc & 0xFFFu, corner = desc >> 12;
			const u32 a = st.cellA[k], bWord = st.cellB[k];
			const u32 c = a & 0xFFFu, code = (a >> 12) & 0xFFu, zeroMask = a >> 20;
			const u32 vi = RT.cell[RT.cls[code] * 16 + 1 + corner];
			const u32 newMask = bWord & 0xFFFu;
			u32 id;
			if ((newMask >> vi) & 1u) {
				id = (st.cellC[k] & 0xFFFFu) + (u32)__popc(newMask & ((1u << vi) - 1u));
			} else if ((bWord >> 31) && ((st.invalidMask[k] >> vi) & 1u)) {
				id = INVALID_INDEX;
			} else {
				u32 dir, slot;
				reg_reuse_source(zeroMask, RT.edge[RT.nibble(RT.vert[code], vi)], dir, slot);
				const u32 c2 = c - ((dir & 1u) + (((dir >> 1) & 1u) << 4) + (((dir >> 2) & 1u) << 8));
				const u32 k2 = bit_rank(st.ntBits, st.wordPrefix, c2);
				id = (st.cellC[k2] & 0xFFFFu) + (((u32)st.ords[k2] >> (slot * 4u)) & 0xFu);
			}
			TV_STREAM_STORE(&out[j], id);
		}
	}
};

__device__ __forceinline__ u32 r0_uniform(u32 v) { return (u32)__builtin_amdgcn_readfirstlane((int)v); }

// what the work list says about item `it` (loads only: requested early, looked at late)
struct R0Candidate {
	u32 valid, slot, ntc, skip, coord;
};

// MODE: which level-0 slots a launch works on — 0: all of them (full run), 1: the work list of an incremental run,
// 2: the blocks the fast pass (vx_fast0.inl) handed on
template <int MODE>
__device__ __forceinline__ R0Candidate r0_peek(const ExecParamsDev& p, const LevelDesc& L, u32 total, u32 it)
{
	// loads without conditions (see R0::request): an item beyond the list reads entry 0 and is marked invalid
	R0Candidate c;
	u32 item = xcd_item(it);
	c.valid = (it < ((total + 63u) & ~63u) && item < total) ? 1u : 0u;
	if (!c.valid) item = 0;
	c.slot = item;
	if (MODE == 1) c.slot = p.G.workItems[0][item]; // incremental runs list the slots to rebuild (a dependent load: compiled in only there)
	if (MODE == 2) c.slot = p.G.slowItems[0][item];
	c.ntc = L.ntCount[c.slot];
	c.skip = L.skip[c.slot];
	c.coord = L.slotCoord[c.slot];
	return c;
}

// Does the candidate belong to this workgroup's capacity class and carry geometry?  Blocks without geometry get their
// empty record here (the first class, lo == 0, owns them).  Uniform over the workgroup.
template <int CAP>
__device__ __forceinline__ bool r0_accept(const LevelDesc& L, u32 lo, const R0Candidate& c, R0Block& b)
{
	// every field is taken out of its load register here, on every path: a load that is still formally pending when its
	// register is reused later costs a full drain of the memory queue at that point
	const u32 valid = r0_uniform(c.valid), slot = r0_uniform(c.slot), ntc = r0_uniform(c.ntc), skip = r0_uniform(c.skip), coord = r0_uniform(c.coord);
	if (!valid) return false;
	if ((lo && ntc <= lo) || ntc > (u32)CAP) return false;
	if (ntc == 0 || skip) {
		if (threadIdx.x == 0) reg_write_empty_record(L, slot);
		return false;
	}
	b.slot = slot; b.ntc = ntc; b.coord = coord;
	block_coords(b.coord, L.cnt, b.bx, b.by, b.bz);
	return true;
}

// next accepted item at or after `it` (stride: the workgroups of the pass), starting with an already requested candidate for `it`
template <int CAP, int MODE>
__device__ __forceinline__ bool r0_next_item(const ExecParamsDev& p, const LevelDesc& L, u32 total, u32 lo, u32 stride, u32& it, R0Candidate c, R0Block& b)
{
	const u32 padded = (total + 63u) & ~63u;
	for (;;) {
		if (r0_accept<CAP>(L, lo, c, b)) return true;
		it += stride;
		if (it >= padded) return false;
		c = r0_peek<MODE>(p, L, total, it);
	}
}

// Optional in-kernel phase profile (build with -DVX_R0_PROFILE, tools only): cycles between the marks, summed over the
// blocks a workgroup handles as seen by its thread 0, land in the header words behind the large-block counter.
#if defined(VX_R0_PROFILE)
#define R0_TICK(i) do { const unsigned long long now_ = __builtin_readcyclecounter(); prof[i] += (u32)(now_ - tick); tick = now_; } while (0)
#else
#define R0_TICK(i) do { } while (0)
#endif

// The pass as workgroup `first` of `stride` (a launch of its own: k_regular0 below; the slots the table-driven pass handed
// on are also walked by the first workgroups of k_tail).  Returns whether the workgroup wrote anything.
template <int CAP, int MODE>
__device__ __forceinline__ bool regular0_pass(const ExecParamsDev& p, u32 lo, const u32 first, const u32 stride)
{
#if defined(VX_R0_PROFILE)
	u32 prof[10] = { 0, 0, 0, 0, 0, 0, 0, 0, 0, 0 };
	unsigned long long tick = __builtin_readcyclecounter();
#endif
	typedef Reg0State<CAP> ST;
	typedef R0<CAP> K;
	if (lo && *p.G.largeBlocks == 0) return false; // nothing for the 4096-cell class (uniform over the grid)
	u8* tab = smem;
	ST& st = *(ST*)(smem + R0_TAB_LDS);
	__shared__ u32 wgStats[20]; // statistics of every block this workgroup handles, flushed once at the end

	const LevelDesc& L = p.levels[0];
	const u32 total = r0_uniform(MODE == 1 ? p.G.workCount[0] : (MODE == 2 ? *p.G.slowCount : *L.nActive));
	if (first >= ((total + 63u) & ~63u)) return false; // the grid is sized before the block counts are known
	const int tid = (int)threadIdx.x;
	if (tid < 20) wgStats[tid] = 0;
	const R0Tables RT = r0_stage_tables(tab, p.tables);
	const Tables T = r0_portable_tables(tab, p.tables + TAB_REG_VERT); // the 6-byte rows stay in global memory (rare paths)
	const GridView& g = p.G.grid;

	u32 it = first;
	R0Block cur, nxt;
	R0Prefetch pf;
	// Two blocks are known ahead: `cur` (inputs requested, being processed) and `nxt` (accepted; inputs requested while
	// `cur` writes its output).  The work list entry behind `nxt` is requested together with nxt's inputs and looked at
	// when the iteration ends — all requests of an iteration sit in ONE place (see the output loop below).
	bool have = r0_next_item<CAP, MODE>(p, L, total, lo, stride, it, r0_peek<MODE>(p, L, total, it), cur);
	const bool wrote = true; // (records of blocks without geometry count as well)
	if (have) K::request(g, L, cur, pf);
	it += stride;
	bool haveNext = have && r0_next_item<CAP, MODE>(p, L, total, lo, stride, it, r0_peek<MODE>(p, L, total, it), nxt);
	while (have) {
		const u32 candIt = it + stride;
		R0Candidate cand = { 0u, 0u, 0u, 0u, 0u };
		R0_TICK(9);
		__syncthreads(); // the previous block is done with the LDS state (and the tables are staged)
		R0_TICK(0);
		K::deposit(st, g, L, cur, pf);
		if (tid == 0) st.degenerate = 0;
		__syncthreads();
		R0_TICK(1);

		// ---- popcount prefix of the bitmap (every wave computes all of it: no exchange), compact cell list ----------
		{
			const int lane = tid & 63, wave = tid >> 6;
			const u32 w0 = st.ntBits[lane], w1 = st.ntBits[lane + 64];
			const u32 c0 = (u32)__popc(w0), c1 = (u32)__popc(w1);
			const u32 i0 = wave_inclusive_scan(c0);
			const u32 half = (u32)__shfl((int)i0, 63, 64);
			const u32 i1 = wave_inclusive_scan(c1) + half;
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
				st.cellA[kk++] = (u32)(tid * 16) + x;
			}
		}
		__syncthreads();
		R0_TICK(2);

		// ---- cells: lane owns `per` consecutive compact cells -----------------------------------------------------
		const u32 nt = r0_uniform(st.wordPrefix[128]);
		const u32 per = (nt + WG - 1) / WG;
		const u32 kBeg = min((u32)tid * per, nt), kEnd = min(kBeg + per, nt);
		u32 sum = 0;
		for (u32 k = kBeg; k < kEnd; ++k) {
			K::cell(st, RT, T, L, cur, k, wgStats);
			sum += st.cellC[k];
		}
		R0_TICK(3);
		{
			const u32 incl = wave_inclusive_scan(sum);
			if ((tid & 63) == 63) st.waveTot[tid >> 6] = incl;
			__syncthreads();
			R0_TICK(4);
			u32 waveBase = 0, tot = 0;
#pragma unroll
			for (int w = 0; w < WG / 64; ++w) {
				const u32 s = st.waveTot[w];
				if (w < (tid >> 6)) waveBase += s;
				tot += s;
			}
			const u32 vTotal = tot & 0xFFFFu, iTotal = tot >> 16;
			// both pool reservations are requested now; their results are first needed after the descriptors are written
			// the pool reservations are made by the last lane: it owns the tail of the compact list, its wave is the first to
			// run out of cells and can afford to wait for the two atomics while the others write their descriptors
			if (tid == WG - 1) {
				st.vTotal = vTotal; st.iTotal = iTotal;
				reserve_both(p.P.cursors, vTotal, iTotal, st.vOff, st.iOff);
			}
			u32 run = waveBase + incl - sum;
			for (u32 k = kBeg; k < kEnd; ++k) {
				const u32 v = st.cellC[k];
				st.cellC[k] = run;
				K::describe(st, k, run, 0, 0);
				run += v;
			}
		}
		R0_TICK(5);
		__syncthreads();
		R0_TICK(6);

		const u32 vTotalU = r0_uniform(st.vTotal), iTotalU = r0_uniform(st.iTotal);
		const bool room = r0_uniform(st.vOff) + vTotalU <= p.P.vertCap && r0_uniform(st.iOff) + iTotalU <= p.P.idxCap;
		// Vertices and indices leave in ONE loop (a vertex and an index per lane and trip: two independent dependency
		// chains for the scheduler), and the next block's inputs are requested inside its first trip: a loop with stores
		// that is entered while loads are in flight makes the compiler drain the memory queue in front of it.
		bool requested = false;
		if (room) {
			for (u32 chunk = 0; chunk == 0 || chunk * R0_VDESC < vTotalU || chunk * R0_IDESC < iTotalU; ++chunk) {
				if (chunk) {
					__syncthreads();
					for (u32 k = (u32)tid; k < nt; k += WG) K::describe(st, k, st.cellC[k], chunk * R0_VDESC, chunk * R0_IDESC);
					__syncthreads();
				}
				const u32 cv = chunk * R0_VDESC, ci = chunk * R0_IDESC;
				const u32 vEnd = cv < vTotalU ? min(vTotalU - cv, (u32)R0_VDESC) : 0u, iEnd = ci < iTotalU ? min(iTotalU - ci, (u32)R0_IDESC) : 0u;
				PolyVertex* vOut = p.P.verts + r0_uniform(st.vOff) + cv;
				u32* iOut = p.P.idx + r0_uniform(st.iOff) + ci;
				for (u32 base = 0; base < vEnd || base < iEnd; base += WG) {
					if (!requested) { if (haveNext) K::request(g, L, nxt, pf); cand = r0_peek<MODE>(p, L, total, candIt); requested = true; }
					const u32 j = base + (u32)tid;
					if (j < vEnd) K::emit_vertex(st, RT, p.G, cur, j, vOut);
					if (j < iEnd) K::flush_index(st, RT, j, iOut);
				}
			}
		}
		if (!requested) { if (haveNext) K::request(g, L, nxt, pf); cand = r0_peek<MODE>(p, L, total, candIt); }
		R0_TICK(7);
		if (tid == 0) {
			BlockRecord& r = L.records[cur.slot];
			r.coordId = cur.coord;
			r.vOff = st.vOff; r.vCount = room ? st.vTotal : 0; r.iOff = st.iOff; r.iCount = room ? st.iTotal : 0;
			count_listed_block(L, r.coordId, r.vCount);
			if (!L.hasTransitions) for (int f = 0; f < 6; ++f) { r.tvOff[f] = 0; r.tvCount[f] = 0; r.tiOff[f] = 0; r.tiCount[f] = 0; }
			r.degenerate = st.degenerate;
			r.ntCells = nt;
			r.pad = 0;
			if (!room) atomicOr(&p.P.cursors[CUR_OVF], 1u);
			wgStats[0] += nt;
			if (st.vTotal) wgStats[1] += st.degenerate;
		}
		cur = nxt;
		have = haveNext;
		it = candIt;
		R0_TICK(8);
		haveNext = have && r0_next_item<CAP, MODE>(p, L, total, lo, stride, it, cand, nxt);
	}
	__syncthreads();
	if (tid < 20 && wgStats[tid]) atomicAdd(&p.G.stats[tid], wgStats[tid]);
#if defined(VX_R0_PROFILE)
	if (tid == 0) for (int i = 0; i < 10; ++i) atomicAdd(&p.G.largeBlocks[4 + i], prof[i] >> 10); // units of 1024 cycles
#endif
	return wrote;
}

template <int CAP, int MODE>
__global__ __launch_bounds__(WG) __attribute__((amdgpu_waves_per_eu(4))) void k_regular0(ExecParamsDev p, u32 lo)
{
	(void)regular0_pass<CAP, MODE>(p, lo, blockIdx.x, gridDim.x);
}

} // namespace

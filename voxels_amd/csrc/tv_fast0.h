// tv_fast0.h — the regular-cell pass of level 0 for "clean" blocks: blocks whose 17^3 cell-corner samples hold no
// exact zero.  (PolygonizeBlock, src/TransVoxelImpl.cpp:1529-1750, restricted to the inputs where none of its
// endpoint rules can fire.)
//
// In such a block every vertex lies strictly inside its edge (t = 1..255), so
//   * no vertex sits on a corner: no GenerateVertexFromPoint (:1450-1467), no "fresh vertex at v0" (:1635-1639),
//     no INVALID_INDEX (:1628-1631), slot 0 is never used;
//   * no triangle can be degenerate (three vertices strictly inside three distinct edges of a unit cell), so
//     PushBlocksToResult's filter (:1300-1321) keeps everything;
//   * whether a vertex is reused is a function of (case, reuseValidityMask, "neighbour cell has my material") alone:
//     an owned edge (direction 8) always creates and stores; any other edge is reused iff the mask allows its direction
//     and the owner cell — which sees the same two non-zero samples, hence owns a valid slot — has this cell's
//     material (:1611-1627).
// That turns the per-vertex loop of the general pass into table look-ups: per case, the table vertices grouped by
// reuse direction (six 12-bit masks) and the list positions of the three owned slots.  Blocks with a zero sample go
// to the general pass (vx_regular0.inl) through a list; results are identical either way.
//
// The per-lane functions are __host__ __device__: k_regular0_fast (vx_fast0.inl) runs them out of LDS, and
// tests/emu runs f0_block_serial() on the CPU so that the formulation is checked against the oracle without a GPU.
#pragma once

#include "tv_block.h"

namespace tv {

enum { F0_MROW = 20, F0_MPLANE = 17 * F0_MROW, F0_MBYTES = 17 * F0_MPLANE }; // staged material / blend rows: samples 0..16 + padding
enum { F0_VDESC = 1024, F0_TDESC = 1024 };                                    // vertices / triangles described per chunk

// table image offsets (appended to the image of tv_core.h)
enum : u32 {
	TAB_F0_CASE = 9232,      // 256 x 3 dwords, see f0_build_tables
	TAB_F0_TRI = 12304,      // 16 x u64: the 15 triangle-corner nibbles of a class
	TAB_F0_EDGE = 12432,     // 16 x 16 bytes: per edge index, see F0Edge
	TAB_F0_DIRS = 12688,     // 8 bytes: reuse directions 1..6 (bit d-1) a reuseValidityMask allows, + padding to 16
	TAB_F0_BYTES = 12704
};

// what a vertex / an index needs to know about cell edge `e` (index into the regular edge-word table)
struct alignas(16) F0Edge {
	u32 x; // offset of corner v0 from the cell's base sample in the staged distances of level 0 | step to corner v1 << 16
	u32 y; // the same in the staged materials: offset | step << 10; v0 << 20 | axis << 23 | reuse direction << 25 | reuse slot << 29
	u32 z; // like x for the 17^3 staging of the levels >= 1 (tv_fast1.h: rows of F1_SROW bytes)
	u32 w; // unused
};
enum { F1_SROW = 20, F1_SPLANE = 17 * F1_SROW, F1_SBYTES = 17 * F1_SPLANE }; // levels >= 1: sample (i,j,k) at k * F1_SPLANE + j * F1_SROW + i

struct F0Tables {
	const u32* caseRow;                 // [256][3]
	const unsigned long long* vrow;     // [256]: edge index (nibble) of table vertex vi
	const unsigned long long* tri;      // [16]
	const F0Edge* edge;                 // [16]
	const u8* dirs;                     // [8]
};

// Host side: fills the TAB_F0_* part of the table image from Lengyel's tables (regClass[256], regCell[16][16],
// regVert[256][12] words (dir << 12 | slot << 8 | v0 << 4 | v1), edgeWords = the 16-entry word table of the image).
inline void f0_build_tables(u8* img, const unsigned char* regClass, const unsigned char* regCell, const unsigned short* regVert, const u16* edgeWords)
{
	for (u32 code = 0; code < 256; ++code) {
		const u32 cls = regClass[code], geom = regCell[cls * 16], nv = geom >> 4, ntri = geom & 15u;
		u32 dv[8] = { 0, 0, 0, 0, 0, 0, 0, 0 }, own[4] = { 15, 15, 15, 15 };
		for (u32 vi = 0; vi < nv; ++vi) {
			const u32 w = regVert[code * 12 + vi], dir = w >> 12, slot = (w >> 8) & 15u;
			if (dir == 8u) own[slot & 3u] = vi; else dv[dir & 7u] |= 1u << vi;
		}
		const u32 row[3] = { dv[1] | (nv << 12) | (dv[2] << 16) | (ntri << 28),
		                     dv[3] | (cls << 12) | (dv[4] << 16) | (own[1] << 28),
		                     dv[5] | (own[2] << 12) | (dv[6] << 16) | (own[3] << 28) };
		memcpy(img + TAB_F0_CASE + code * 12, row, 12);
	}
	for (u32 cls = 0; cls < 16; ++cls) {
		unsigned long long r = 0;
		for (u32 i = 0; i < 15; ++i) r |= (unsigned long long)(regCell[cls * 16 + 1 + i] & 15u) << (4 * i);
		memcpy(img + TAB_F0_TRI + cls * 8, &r, 8);
	}
	for (u32 e = 0; e < 16; ++e) {
		const u32 w = edgeWords[e], v0 = (w >> 4) & 15u, v1 = w & 15u, dir = w >> 12, slot = (w >> 8) & 15u;
		F0Edge info = { 0, 0, 0, 0 };
		if (w) {
			const u32 d = v1 - v0, axis = d == 1u ? 0u : (d == 2u ? 1u : 2u);
			const u32 sOff = (v0 & 1u) + ((v0 >> 1) & 1u) * SROW + (v0 >> 2) * SPLANE, sStep = axis == 0 ? 1u : (axis == 1 ? (u32)SROW : (u32)SPLANE);
			const u32 mOff = (v0 & 1u) + ((v0 >> 1) & 1u) * F0_MROW + (v0 >> 2) * F0_MPLANE, mStep = axis == 0 ? 1u : (axis == 1 ? (u32)F0_MROW : (u32)F0_MPLANE);
			info.x = sOff | (sStep << 16);
			info.y = mOff | (mStep << 10) | (v0 << 20) | (axis << 23) | ((dir & 15u) << 25) | ((slot & 3u) << 29);
			info.z = ((v0 & 1u) + ((v0 >> 1) & 1u) * F1_SROW + (v0 >> 2) * F1_SPLANE) | ((axis == 0 ? 1u : (axis == 1 ? (u32)F1_SROW : (u32)F1_SPLANE)) << 16);
		}
		memcpy(img + TAB_F0_EDGE + e * 16, &info, 16);
	}
	for (u32 m = 0; m < 8; ++m) {
		u32 a = 0;
		for (u32 d = 1; d <= 6; ++d) if ((d & m) == d) a |= 1u << (d - 1);
		img[TAB_F0_DIRS + m] = (u8)a;
	}
}

TV_HD F0Tables f0_tables_from_image(const u8* base, const unsigned long long* vrow)
{
	F0Tables T;
	T.caseRow = (const u32*)(base + TAB_F0_CASE); T.tri = (const unsigned long long*)(base + TAB_F0_TRI);
	T.edge = (const F0Edge*)(base + TAB_F0_EDGE); T.dirs = base + TAB_F0_DIRS; T.vrow = vrow;
	return T;
}

#if defined(__HIP_DEVICE_COMPILE__)
#define F0_CTZ(x) __builtin_ctz(x)
#else
#define F0_CTZ(x) __builtin_ctz(x)
#endif

template <int CAP>
struct Fast0State {
	i8 samp[SAMP_BYTES + 8];                                   // tv_block.h layout: 19 x 19 rows of 24 bytes
	__attribute__((aligned(4))) u8 matId[F0_MBYTES + 4];       // material id of voxel (i,j,k), 0..16 from the block origin
	__attribute__((aligned(4))) u8 blend[F0_MBYTES + 4];
	u32 ntBits[128];
	u16 wordPrefix[132];
	__attribute__((aligned(8))) u32 cellAN[CAP][2]; // [0]: cell id | case << 12 | class << 20 | triangles << 24   [1]: created vertices (12) | ordinals of the vertices stored in slots 1..3 (4 bits each) << 12
	u32 cellC[CAP];        // created vertices | triangles << 16; after the scan: vertex base | triangle base << 16
	u16 vdesc[F0_VDESC];   // new vertices of the chunk: cell id | edge index << 12
	u16 tdesc[F0_TDESC];   // triangles of the chunk: compact cell | triangle << 12
	u32 waveTot[8];
	u32 vOff, iOff, vTotal, tTotal, zero;
};

// Where a pass keeps the samples and the cell materials (level 0: 19^3 distances + 17^3 voxel materials, the material of
// a cell being the voxel at its base corner, :755-760; levels >= 1: 17^3 lattice samples + the block's material cache)
struct F0Layout {
	enum { SR = SROW, SP = SPLANE };
	template <typename ST> static TV_HD const i8* base_sample(const ST& st, int cx, int cy, int cz) { return st.samp + samp_index(cx, cy, cz); }
	// bit d-1: the cell in reuse direction d (x-1 | y-1 << 1 | z-1 << 2) has this cell's material id
	template <typename ST> static TV_HD u32 same_material(const ST& st, u32, int cx, int cy, int cz)
	{
		const u8* mp = st.matId + (cz * F0_MPLANE + cy * F0_MROW + cx);
		const u32 mine = mp[0];
		return (mp[-1] == mine ? 1u : 0u) | (mp[-F0_MROW] == mine ? 2u : 0u) | (mp[-F0_MROW - 1] == mine ? 4u : 0u)
		     | (mp[-F0_MPLANE] == mine ? 8u : 0u) | (mp[-F0_MPLANE - 1] == mine ? 16u : 0u) | (mp[-F0_MPLANE - F0_MROW] == mine ? 32u : 0u);
	}
};

// ---- one compact cell: case, reuse resolution, counts (returns created vertices | triangles << 16) ---------------------
template <typename LAY, typename ST>
TV_HD u32 fx_cell(ST& st, const F0Tables& T, u32 k, u32* classCount)
{
	const u32 c = st.cellAN[k][0] & 0xFFFu;
	const int cx = (int)(c & 15u), cy = (int)((c >> 4) & 15u), cz = (int)(c >> 8);
	const i8* sp = LAY::base_sample(st, cx, cy, cz);
	enum { SR = LAY::SR, SP = LAY::SP };
	// the sign of a sign-extended byte fills bits 7..31: bit 8 + i of corner i's sample is bit i of the case code
	const int v0 = sp[0], v1 = sp[1], v2 = sp[SR], v3 = sp[SR + 1], v4 = sp[SP], v5 = sp[SP + 1], v6 = sp[SP + SR], v7 = sp[SP + SR + 1];
	const u32 code = (((u32)v0 & 0x100u) | ((u32)v1 & 0x200u) | ((u32)v2 & 0x400u) | ((u32)v3 & 0x800u)
	                | ((u32)v4 & 0x1000u) | ((u32)v5 & 0x2000u) | ((u32)v6 & 0x4000u) | ((u32)v7 & 0x8000u)) >> 8;
	const u32* row = T.caseRow + code * 3u;
	const u32 w0 = row[0], w1 = row[1], w2 = row[2];
	const u32 nv = (w0 >> 12) & 15u, ntri = w0 >> 28, cls = (w1 >> 12) & 15u;
	// materials of this cell and of the six cells its non-owned edges come from
	const u32 eq = LAY::same_material(st, c, cx, cy, cz);
	// reuseValidityMask from the bitmap (TransVoxelImpl.cpp:1543-1548, :1742-1748): a non-trivial cell earlier in the
	// row / in an earlier row of the slice / in an earlier slice
	const u32 rowId = (u32)((cz << 4) | cy);
	const u32 rowBits = (st.ntBits[rowId >> 1] >> ((rowId & 1u) * 16u)) & 0xFFFFu;
	const u32 below = rowBits & ((1u << cx) - 1u);
	const u32 sliceBase = st.wordPrefix[cz * 8];
	u32 mask3 = below ? 1u : 0u;
	if (k - (u32)TV_POPC(below) - sliceBase) mask3 |= 2u;
	if (sliceBase) mask3 |= 4u;
	const u32 allow = (u32)T.dirs[mask3] & eq; // bit d-1: vertices with reuse direction d are reused
	u32 reused = 0;
	reused |= (0u - (allow & 1u)) & w0;
	reused |= (0u - ((allow >> 1) & 1u)) & (w0 >> 16);
	reused |= (0u - ((allow >> 2) & 1u)) & w1;
	reused |= (0u - ((allow >> 3) & 1u)) & (w1 >> 16);
	reused |= (0u - ((allow >> 4) & 1u)) & w2;
	reused |= (0u - ((allow >> 5) & 1u)) & (w2 >> 16);
	const u32 newMask = ((1u << nv) - 1u) & ~reused;
	const u32 p1 = w1 >> 28, p2 = (w2 >> 12) & 15u, p3 = w2 >> 28; // list positions of the vertices stored in slots 1..3 (15: none)
	const u32 o1 = (u32)TV_POPC(newMask & ((1u << p1) - 1u)), o2 = (u32)TV_POPC(newMask & ((1u << p2) - 1u)), o3 = (u32)TV_POPC(newMask & ((1u << p3) - 1u));
	st.cellAN[k][0] = c | (code << 12) | (cls << 20) | (ntri << 24);
	st.cellAN[k][1] = newMask | (o1 << 12) | (o2 << 16) | (o3 << 20);
	TV_ATOMIC_ADD(&classCount[cls], 1u);
	return (u32)TV_POPC(newMask) | (ntri << 16);
}

template <typename ST>
TV_HD u32 f0_cell(ST& st, const F0Tables& T, u32 k, u32* classCount) { return fx_cell<F0Layout>(st, T, k, classCount); }

// descriptors of cell k's new vertices and triangles that fall into the given chunks; base = vertex base | triangle base << 16
template <typename ST>
TV_HD void f0_describe(ST& st, const F0Tables& T, u32 k, u32 base, u32 chunkV, u32 chunkT)
{
	const u32 F0_VDESC = (u32)(sizeof(st.vdesc) / sizeof(st.vdesc[0])), F0_TDESC = (u32)(sizeof(st.tdesc) / sizeof(st.tdesc[0])); // (shadow the level-0 capacities)
	const u32 a = st.cellAN[k][0];
	u32 m = st.cellAN[k][1] & 0xFFFu;
	u32 j = base & 0xFFFFu;
	if (m && j < chunkV + F0_VDESC && j + 12 > chunkV) {
		const unsigned long long vrow = T.vrow[(a >> 12) & 0xFFu];
		const u32 c = a & 0xFFFu;
		while (m) {
			const u32 vi = (u32)F0_CTZ(m);
			m &= m - 1;
			if (j >= chunkV && j < chunkV + F0_VDESC) st.vdesc[j - chunkV] = (u16)(c | (((u32)(vrow >> (vi * 4u)) & 15u) << 12));
			++j;
		}
	}
	const u32 ntri = (a >> 24) & 7u;
	u32 t = base >> 16;
	if (t < chunkT + F0_TDESC && t + 5 > chunkT) {
		for (u32 tr = 0; tr < ntri; ++tr, ++t)
			if (t >= chunkT && t < chunkT + F0_TDESC) st.tdesc[t - chunkT] = (u16)(k | (tr << 12));
	}
}

// ---- one lane = one new vertex (reg_edge_vertex of tv_core.h at level 0 for a vertex strictly inside its edge) ---------
// desc = cell id | edge index << 12; (ox,oy,oz) = the block's origin in voxels
template <typename ST, typename SINK>
TV_HD void f0_vertex(const ST& st, const F0Tables& T, u32 desc, int ox, int oy, int oz, unsigned long long lutRow, const SINK& sink)
{
	const u32 c = desc & 0xFFFu;
	const F0Edge e = T.edge[desc >> 12];
	const int cx = (int)(c & 15u), cy = (int)((c >> 4) & 15u), cz = (int)(c >> 8);
	const i8* s0 = st.samp + samp_index(cx, cy, cz) + (e.x & 0xFFFFu);
	const i8* s1 = s0 + (e.x >> 16);
	const int val0 = *s0, val1 = *s1;
	// central differences at both end points (CalcNormal, :1239-1246: components ordered x, z, y).  The factor 0.5 of the
	// reference is dropped: scaling a vector by a power of two changes neither its normalised components nor any rounding
	// on the way (no overflow / underflow here), and a zero vector stays zero.
	float N0[3], N1[3];
	N0[0] = (float)((int)s0[1] - (int)s0[-1]); N0[1] = (float)((int)s0[SPLANE] - (int)s0[-SPLANE]); N0[2] = (float)((int)s0[SROW] - (int)s0[-SROW]);
	N1[0] = (float)((int)s1[1] - (int)s1[-1]); N1[1] = (float)((int)s1[SPLANE] - (int)s1[-SPLANE]); N1[2] = (float)((int)s1[SROW] - (int)s1[-SROW]);
	const int mo = cz * F0_MPLANE + cy * F0_MROW + cx;
	const int m0 = mo + (int)(e.y & 0x3FFu), m1 = m0 + (int)((e.y >> 10) & 0x3FFu);
	const u32 cellId = st.matId[mo], cellBlend = st.blend[mo];
	const u32 id0 = st.matId[m0], id1 = st.matId[m1], b0 = st.blend[m0], b1 = st.blend[m1];
	// t = (v1 << 8) / (v1 - v0), truncated (:1591); the samples are non-zero and of opposite sign
	const int t = edge_t_crossing(val0, val1), u = 256 - t;
	// position x256 = t * P0 + u * P1 with P1 = P0 + unit(axis): integers below 2^24, so the reference's fp32 expression
	// is exact and equals 256 * P0 + u along the edge's axis
	const u32 uu = (u32)u & 0x1FFu, axis = (e.y >> 23) & 3u; // (masked: 24-bit multiplies below)
	const int px = ((ox + cx + (int)((e.y >> 20) & 1u)) << 8) + (int)(axis == 0u ? uu : 0u);
	const int py = ((oy + cy + (int)((e.y >> 21) & 1u)) << 8) + (int)(axis == 1u ? uu : 0u);
	const int pz = ((oz + cz + (int)((e.y >> 22) & 1u)) << 8) + (int)(axis == 2u ? uu : 0u);
	RawVertex rv;
	rv.p[0] = (float)px; rv.p[1] = (float)py; rv.p[2] = (float)pz;
	rv.s[0] = rv.p[0]; rv.s[1] = rv.p[1]; rv.s[2] = rv.p[2];
	rv.flags = 0;
	normalize_gradient(N0);
	normalize_gradient(N1);
	// blend (t * b0 + u * b1) / 256 truncated (:1699): the fp32 expression is exact integer arithmetic (sum <= 256 * 255)
	if (id0 == id1 && id0 == cellId) rv.mat = id0 | (((((u32)t & 0x1FFu) * b0 + uu * b1) >> 8) << 8);
	else rv.mat = cellId | (cellBlend << 8);
	const float wt = (float)t / 256.f, wu = (float)u / 256.f;
	rv.n[0] = N0[0] * wt + N1[0] * wu; rv.n[1] = N0[1] * wt + N1[1] * wu; rv.n[2] = N0[2] * wt + N1[2] * wu;
	normalize_fix_zero(rv.n);
	sink(rv, lutRow);
}
template <typename ST>
TV_HD void f0_vertex(const ST& st, const F0Tables& T, u32 desc, int ox, int oy, int oz, unsigned long long lutRow, PolyVertex* out)
{
	f0_vertex(st, T, desc, ox, oy, oz, lutRow, VertexToMemory{ out });
}

// ---- one lane = one triangle of the chunk: its three indices --------------------------------------------------------
// A created vertex is the cell's vertex base plus its rank among the cell's created vertices; a reused one comes from the
// stored slot ordinal of the owner cell (direction and slot are properties of the edge).  Both forms are evaluated for
// every corner and one is picked: a wave nearly always holds both kinds, and without branches the look-ups of the three
// corners are in flight together.  (For a created vertex the "owner" resolves to a cell of the block or to the cell
// itself, so every read stays inside the staged state.)
template <typename ST>
TV_HD void f0_triangle(const ST& st, const F0Tables& T, u32 t, u32 out[3])
{
	const u32 d = st.tdesc[t];
	const u32 k = d & 0xFFFu, tr = d >> 12;
	const u32 a = st.cellAN[k][0], nm = st.cellAN[k][1];
	const u32 corners = (u32)(T.tri[(a >> 20) & 15u] >> (tr * 12u));
	const unsigned long long vrow = T.vrow[(a >> 12) & 0xFFu];
	const u32 vbase = st.cellC[k] & 0xFFFFu, c = a & 0xFFFu;
#pragma unroll
	for (u32 q = 0; q < 3; ++q) {
		const u32 vi = (corners >> (4u * q)) & 15u;
		const u32 created = vbase + (u32)TV_POPC(nm & ((1u << vi) - 1u) & 0xFFFu);
		const u32 e = (u32)(vrow >> (vi * 4u)) & 15u;
		const u32 info = T.edge[e].y, dir = (info >> 25) & 7u, slot = info >> 29;
		const u32 c2 = (c - ((dir & 1u) + ((dir & 2u) << 3) + ((dir & 4u) << 6))) & 0xFFFu;
		const u32 k2 = bit_rank(st.ntBits, st.wordPrefix, c2);
		const u32 k2c = k2 < (u32)(sizeof(st.cellC) / sizeof(st.cellC[0])) ? k2 : 0u; // unused results may come from anywhere, reads may not
		const u32 reused = (st.cellC[k2c] & 0xFFFFu) + ((st.cellAN[k2c][1] >> (8u + 4u * slot)) & 15u);
		out[q] = ((nm >> vi) & 1u) ? created : reused;
	}
}

#if !defined(__HIPCC__)
// ---- CPU emulation of one block (tests/emu): the same per-lane functions, serial scans ---------------------------------
// false: the block holds a zero sample and belongs to the general pass
template <int CAP>
inline bool f0_block_serial(Fast0State<CAP>& st, const F0Tables& T, const Globals& G, const LevelDesc& L, const Pools& P, u32 slot, u32 bx, u32 by, u32 bz, u32* stats)
{
	const GridView& g = G.grid;
	const int ox = (int)bx * 16, oy = (int)by * 16, oz = (int)bz * 16;
	bool zero = false;
	for (int k = -1; k <= 17; ++k) for (int j = -1; j <= 17; ++j) for (int i = -1; i <= 17; ++i) {
		const int v = dist_at(g, ox + i, oy + j, oz + k);
		st.samp[samp_index(i, j, k)] = (i8)v;
		if (v == 0 && i >= 0 && j >= 0 && k >= 0 && i <= 16 && j <= 16 && k <= 16) zero = true;
	}
	if (zero) return false;
	for (int k = 0; k <= 16; ++k) for (int j = 0; j <= 16; ++j) for (int i = 0; i <= 16; ++i) {
		const u32 m = mat_at(g, ox + i, oy + j, oz + k);
		st.matId[k * F0_MPLANE + j * F0_MROW + i] = (u8)m; st.blend[k * F0_MPLANE + j * F0_MROW + i] = (u8)(m >> 8);
	}
	u32 nt = 0;
	for (int w = 0; w < 128; ++w) { st.ntBits[w] = L.ntBits[(size_t)slot * 128 + w]; st.wordPrefix[w] = (u16)nt; nt += (u32)TV_POPC(st.ntBits[w]); }
	st.wordPrefix[128] = (u16)nt;
	for (u32 c = 0, k = 0; c < BLOCK_CELLS; ++c) if (bit_get(st.ntBits, c)) st.cellAN[k++][0] = c;
	u32 classCount[16] = { 0 }, run = 0;
	for (u32 k = 0; k < nt; ++k) { const u32 cnt = f0_cell(st, T, k, classCount); st.cellC[k] = run; run += cnt; }
	const u32 vTotal = run & 0xFFFFu, tTotal = run >> 16;
	const u32 vOff = TV_ATOMIC_ADD(&P.cursors[CUR_V], vTotal), iOff = TV_ATOMIC_ADD(&P.cursors[CUR_I], tTotal * 3u);
	const bool room = vOff + vTotal <= P.vertCap && iOff + tTotal * 3u <= P.idxCap;
	for (u32 chunk = 0; room && (chunk * F0_VDESC < vTotal || chunk * F0_TDESC < tTotal); ++chunk) {
		const u32 cv = chunk * F0_VDESC, ct = chunk * F0_TDESC;
		for (u32 k = 0; k < nt; ++k) f0_describe(st, T, k, st.cellC[k], cv, ct);
		const u32 vEnd = cv < vTotal ? (vTotal - cv < (u32)F0_VDESC ? vTotal - cv : (u32)F0_VDESC) : 0u;
		const u32 tEnd = ct < tTotal ? (tTotal - ct < (u32)F0_TDESC ? tTotal - ct : (u32)F0_TDESC) : 0u;
		for (u32 j = 0; j < vEnd; ++j) {
			const u32 desc = st.vdesc[j], c = desc & 0xFFFu;
			const u32 cellId = st.matId[(c >> 8) * F0_MPLANE + ((c >> 4) & 15u) * F0_MROW + (c & 15u)];
			f0_vertex(st, T, desc, ox, oy, oz, lut_row(G.lut, cellId), P.verts + vOff + cv + j);
		}
		for (u32 t = 0; t < tEnd; ++t) f0_triangle(st, T, t, P.idx + iOff + (ct + t) * 3u);
	}
	BlockRecord& r = L.records[slot];
	r.coordId = L.slotCoord[slot];
	r.vOff = vOff; r.vCount = room ? vTotal : 0; r.iOff = iOff; r.iCount = room ? tTotal * 3u : 0;
	if (!L.hasTransitions) for (int f = 0; f < 6; ++f) { r.tvOff[f] = 0; r.tvCount[f] = 0; r.tiOff[f] = 0; r.tiCount[f] = 0; }
	r.degenerate = 0; r.ntCells = nt; r.pad = 0;
	if (!room) TV_ATOMIC_OR(&P.cursors[CUR_OVF], 1u);
	stats[0] += nt;
	for (int i = 0; i < 16; ++i) stats[4 + i] += classCount[i];
	return true;
}
#endif

} // namespace tv

// tv_fast1.h — the regular-cell pass of the LOD levels >= 1 for blocks whose 17^3 lattice samples hold no exact zero
// (PolygonizeBlock, src/TransVoxelImpl.cpp:1529-1750, at cell sizes 2, 4, 8).
//
// As on level 0 (tv_fast0.h) the absence of zero corner samples makes the reuse decisions table look-ups: the cell pass,
// the descriptors and the triangle pass are the very functions of tv_fast0.h, with the cell materials coming from the
// block's material cache (CalculateMaterialForCellCache, :753-838, written by the material pass).  What level >= 1 adds:
//   * every vertex walks the LOD chain (FindBestVertexInLODChain, :1484-1509) down to the level-0 edge that holds the
//     crossing, then reads its normals and materials around that edge: 16 + L voxel fetches from the grid.  A voxel's
//     address in the brick mirrors is a SUM of one term per axis (brick_local's bit fields are disjoint per axis and
//     the brick index is linear in the block coordinates), so the terms are computed once per coordinate value and a
//     fetch costs two additions — `SMP` below hides whether that is the mirror (device) or the dense field (CPU emulation);
//   * boundary flags and the secondary position (:593-645, :1473-1482, :1729-1738);
//   * a chain can end ON a voxel (a zero inside the coarse edge): then triangles around that vertex may be degenerate and
//     the block needs PushBlocksToResult's filter (:1300-1321).  Such a block ("suspect") is handed to the general pass,
//     like a block with a zero lattice sample; for every other block all triangles are kept.
#pragma once

#include "tv_fast0.h"

namespace tv {

enum { F1_VDESC = 1024, F1_TDESC = 1024 }; // (one chunk for an ordinary block of a level >= 1 - ~700 vertices, as many triangles; the state has the room: the transition state is larger)

template <int CAP>
struct Fast1State {
	i8 samp[F1_SBYTES + 12];   // 17 x 17 rows of F1_SROW bytes (+ padding: the cache block below is copied in 16-byte pieces)
	__attribute__((aligned(16))) u8 cacheId[BLOCK_CELLS]; // material ids of the block's LevelMaterialCache entries (the cell pass compares ids; a vertex reads its cell's whole entry, id | blend << 8, from the cache in HBM with its voxel fetches: 4 KB of LDS less per workgroup, which matters to the level-0 pass running beside this one)
	u32 ntBits[128];
	u16 wordPrefix[132];
	__attribute__((aligned(8))) u32 cellAN[CAP][2];
	u32 cellC[CAP];
	u16 vdesc[F1_VDESC];
	u16 tdesc[F1_TDESC];
	u32 waveTot[8];
	u32 classCount[16];        // per-class cell counts of THIS block (added to the run's statistics once the block is final)
	u32 vOff, iOff, vTotal, tTotal, zero, suspect;
};

struct F1Layout {
	enum { SR = F1_SROW, SP = F1_SPLANE };
	template <typename ST> static TV_HD const i8* base_sample(const ST& st, int cx, int cy, int cz) { return st.samp + (cz * F1_SPLANE + cy * F1_SROW + cx); }
	template <typename ST> static TV_HD u32 same_material(const ST& st, u32 c, int, int, int)
	{
		const u8* mp = st.cacheId + c; // neighbours outside the block are never asked for (their mask bit is 0): any value will do
		const u32 mine = mp[0];
		return (mp[-1] == mine ? 1u : 0u) | (mp[-16] == mine ? 2u : 0u) | (mp[-17] == mine ? 4u : 0u)
		     | (mp[-256] == mine ? 8u : 0u) | (mp[-257] == mine ? 16u : 0u) | (mp[-272] == mine ? 32u : 0u);
	}
};

// CPU emulation: voxel addresses in the dense fields (also sums of one term per axis); materials through mat_at
struct F1HostSampler {
	const GridView* g;
	typedef size_t Off;
	TV_HD Off tx(int x) const { return (size_t)clampi(x, 0, g->n - 1); }
	TV_HD Off ty(int y) const { return (size_t)(clampi(y, 0, g->n - 1) - g->yOrigin) * (size_t)g->n; }
	TV_HD Off tz(int z) const { return (size_t)(clampi(z, 0, g->n - 1) - g->zOrigin) * (size_t)g->pitchY * (size_t)g->n; }
	TV_HD int dist(Off o) const { return g->dist[o]; }
	TV_HD u32 mat(Off, int x, int y, int z) const { return mat_at(*g, x, y, z); }
};

// ---- one lane = one new vertex (reg_edge_vertex of tv_core.h, for a cell without zero corner samples) ---------------
// desc = cell id | edge index << 12; (ox,oy,oz) = the block's origin in voxels; returns "strictly inside its level-0 edge"
template <typename ST, typename SMP, typename SINK>
TV_HD bool f1_vertex(const ST& st, const F0Tables& T, const SMP& smp, const u16* blockCache, u32 desc, int level, int ox, int oy, int oz, unsigned long long lutRow, const SINK& sink)
{
	typedef typename SMP::Off Off;
	const u32 c = desc & 0xFFFu;
	const F0Edge e = T.edge[desc >> 12];
	const int cx = (int)(c & 15u), cy = (int)((c >> 4) & 15u), cz = (int)(c >> 8);
	const int mult = 1 << level;
	const i8* s0 = st.samp + (cz * F1_SPLANE + cy * F1_SROW + cx) + (e.z & 0xFFFFu);
	int p0 = s0[0], p1 = s0[e.z >> 16];
	const u32 v0 = (e.y >> 20) & 7u, axis = (e.y >> 23) & 3u;
	const int ax = axis == 0u ? 1 : 0, ay = axis == 1u ? 1 : 0, az = axis == 2u ? 1 : 0;
	// end points of the coarse edge; the chain moves one of them to the middle, level times (:1495-1507)
	int x0 = ox + (cx + (int)(v0 & 1u)) * mult, y0 = oy + (cy + (int)((v0 >> 1) & 1u)) * mult, z0 = oz + (cz + (int)(v0 >> 2)) * mult;
	int len = mult; // P1 = P0 + len along the axis
	for (int lev = level; lev > 0; --lev) {
		const int h = len >> 1;
		const int mx = x0 + ax * h, my = y0 + ay * h, mz = z0 + az * h;
		const int midV = smp.dist(smp.tx(mx) + smp.ty(my) + smp.tz(mz));
		if (p0 * midV <= 0) p1 = midV;
		else { x0 = mx; y0 = my; z0 = mz; p0 = midV; }
		len = h;
	}
	// the level-0 edge P0 - P1 = P0 + unit(axis): one address term per coordinate value in reach of the two stencils
	const Off tx0 = smp.tx(x0), txm = smp.tx(x0 - 1), txp = smp.tx(x0 + 1), txq = smp.tx(x0 + 1 + ax);
	const Off ty0 = smp.ty(y0), tym = smp.ty(y0 - 1), typ = smp.ty(y0 + 1), tyq = smp.ty(y0 + 1 + ay);
	const Off tz0 = smp.tz(z0), tzm = smp.tz(z0 - 1), tzp = smp.tz(z0 + 1), tzq = smp.tz(z0 + 1 + az);
	// P1's centre / minus / plus terms per axis
	const Off tx1 = ax ? txp : tx0, tx1m = ax ? tx0 : txm, tx1p = ax ? txq : txp;
	const Off ty1 = ay ? typ : ty0, ty1m = ay ? ty0 : tym, ty1p = ay ? tyq : typ;
	const Off tz1 = az ? tzp : tz0, tz1m = az ? tz0 : tzm, tz1p = az ? tzq : tzp;
	const Off yz0 = ty0 + tz0, xz0 = tx0 + tz0, xy0 = tx0 + ty0, yz1 = ty1 + tz1, xz1 = tx1 + tz1, xy1 = tx1 + ty1;
	// everything a vertex reads around its two end points is requested before anything is computed with it
	const int a0 = smp.dist(txp + yz0), a1 = smp.dist(txm + yz0), a2 = smp.dist(xy0 + tzp), a3 = smp.dist(xy0 + tzm), a4 = smp.dist(xz0 + typ), a5 = smp.dist(xz0 + tym);
	const int b0 = smp.dist(tx1p + yz1), b1 = smp.dist(tx1m + yz1), b2 = smp.dist(xy1 + tz1p), b3 = smp.dist(xy1 + tz1m), b4 = smp.dist(xz1 + ty1p), b5 = smp.dist(xz1 + ty1m);
	const u32 M0 = smp.mat(tx0 + yz0, x0, y0, z0), M1 = smp.mat(tx1 + yz1, x0 + ax, y0 + ay, z0 + az);
	const u32 cellMat = TV_LOAD_THROUGH(&blockCache[c]); // (requested with the fetches above; past the L1: another workgroup of the launch wrote it)
	const bool interior = p0 * p1 < 0; // samples of strictly opposite sign: 0 < t < 256, vertex strictly inside its edge
	const int t = (p0 != p1) ? edge_t_crossing(p0, p1) : 0, u = 256 - t; // (:1671-1678; the chain keeps p0 * p1 <= 0)
	const u32 uu = (u32)u & 0x1FFu;
	RawVertex rv;
	// x256 position t * P0 + u * P1 = 256 * P0 + u * unit(axis): integers below 2^24, the reference's fp32 expression is exact
	rv.p[0] = (float)((x0 << 8) + (int)(ax ? uu : 0u)); rv.p[1] = (float)((y0 << 8) + (int)(ay ? uu : 0u)); rv.p[2] = (float)((z0 << 8) + (int)(az ? uu : 0u));
	// central differences, components ordered x, z, y (CalcNormal, :1239-1246; the factor 0.5 does not survive normalising)
	float N0[3] = { (float)(a0 - a1), (float)(a2 - a3), (float)(a4 - a5) }, N1[3] = { (float)(b0 - b1), (float)(b2 - b3), (float)(b4 - b5) };
	normalize_gradient(N0);
	normalize_gradient(N1);
	const int v1 = (int)v0 | (1 << axis);
	rv.flags = boundary_mask(cx, cy, cz, mult, (int)v0, v1);
	if ((M0 & 0xFFu) == (M1 & 0xFFu) && (M0 & 0xFFu) == (cellMat & 0xFFu)) rv.mat = (M0 & 0xFFu) | (((((u32)t & 0x1FFu) * (M0 >> 8) + uu * (M1 >> 8)) >> 8) << 8);
	else rv.mat = cellMat;
	const float wt = (float)t / 256.f, wu = (float)u / 256.f;
	rv.n[0] = N0[0] * wt + N1[0] * wu; rv.n[1] = N0[1] * wt + N1[1] * wu; rv.n[2] = N0[2] * wt + N1[2] * wu;
	normalize_fix_zero(rv.n);
	finish_secondary(rv, mult);
	sink(rv, lutRow);
	return interior;
}
template <typename ST, typename SMP>
TV_HD bool f1_vertex(const ST& st, const F0Tables& T, const SMP& smp, const u16* blockCache, u32 desc, int level, int ox, int oy, int oz, unsigned long long lutRow, PolyVertex* out)
{
	return f1_vertex(st, T, smp, blockCache, desc, level, ox, oy, oz, lutRow, VertexToMemory{ out });
}

#if !defined(__HIPCC__)
// ---- CPU emulation of one block (tests/emu).  false: the block belongs to the general pass (a zero lattice sample, or
//      a chain ended on a voxel); nothing of the result was written then except the pool cursors, which only grow.
template <int CAP>
inline bool f1_block_serial(Fast1State<CAP>& st, const F0Tables& T, const Globals& G, const LevelDesc& L, const Pools& P, u32 level, u32 slot, u32 bx, u32 by, u32 bz, u32* stats)
{
	const GridView& g = G.grid;
	const int mult = (int)L.mult, ox = (int)bx * 16 * mult, oy = (int)by * 16 * mult, oz = (int)bz * 16 * mult;
	for (int k = 0; k <= 16; ++k) for (int j = 0; j <= 16; ++j) for (int i = 0; i <= 16; ++i) {
		const int v = dist_at(g, ox + i * mult, oy + j * mult, oz + k * mult);
		if (v == 0) return false;
		st.samp[k * F1_SPLANE + j * F1_SROW + i] = (i8)v;
	}
	const u16* blockCache = L.cache + (size_t)slot * BLOCK_CELLS;
	for (u32 c = 0; c < (u32)BLOCK_CELLS; ++c) st.cacheId[c] = (u8)(blockCache[c] & 0xFFu);
	u32 nt = 0;
	for (int w = 0; w < 128; ++w) { st.ntBits[w] = L.ntBits[(size_t)slot * 128 + w]; st.wordPrefix[w] = (u16)nt; nt += (u32)TV_POPC(st.ntBits[w]); }
	st.wordPrefix[128] = (u16)nt;
	for (u32 c = 0, k = 0; c < BLOCK_CELLS; ++c) if (bit_get(st.ntBits, c)) st.cellAN[k++][0] = c;
	u32 classCount[16] = { 0 }, run = 0;
	for (u32 k = 0; k < nt; ++k) { const u32 cnt = fx_cell<F1Layout>(st, T, k, classCount); st.cellC[k] = run; run += cnt; }
	const u32 vTotal = run & 0xFFFFu, tTotal = run >> 16;
	const u32 vOff = TV_ATOMIC_ADD(&P.cursors[CUR_V], vTotal), iOff = TV_ATOMIC_ADD(&P.cursors[CUR_I], tTotal * 3u);
	const bool room = vOff + vTotal <= P.vertCap && iOff + tTotal * 3u <= P.idxCap;
	const F1HostSampler smp{ &g };
	bool suspect = false;
	for (u32 chunk = 0; room && (chunk * F1_VDESC < vTotal || chunk * F1_TDESC < tTotal); ++chunk) {
		const u32 cv = chunk * F1_VDESC, ct = chunk * F1_TDESC;
		for (u32 k = 0; k < nt; ++k) f0_describe(st, T, k, st.cellC[k], cv, ct);
		const u32 vEnd = cv < vTotal ? (vTotal - cv < (u32)F1_VDESC ? vTotal - cv : (u32)F1_VDESC) : 0u;
		const u32 tEnd = ct < tTotal ? (tTotal - ct < (u32)F1_TDESC ? tTotal - ct : (u32)F1_TDESC) : 0u;
		for (u32 j = 0; j < vEnd; ++j) {
			const u32 desc = st.vdesc[j];
			if (!f1_vertex(st, T, smp, blockCache, desc, (int)level, ox, oy, oz, lut_row(G.lut, st.cacheId[desc & 0xFFFu]), P.verts + vOff + cv + j)) suspect = true;
		}
		for (u32 t = 0; t < tEnd; ++t) f0_triangle(st, T, t, P.idx + iOff + (ct + t) * 3u);
	}
	if (suspect) return false;
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

// oracle/port.cpp — CPU restatement of the reference's polygonization path (and of the small part of
// its grid that feeds it), written from scratch for this repo.  TEST INFRASTRUCTURE ONLY: it is the
// checker the HIP path is compared with; nothing in the product may include, link or call it.
//
// Parity status: PINNED.  tests/test_oracle.py compares every output array of this file with the
// unmodified reference (oracle/_ref, built from /root/reference/src by oracle/Makefile) on seeded
// inputs, and with the committed fixtures in tests/golden/ that were generated from that reference.
//
// The algorithm follows the reference's *sequential* formulation on purpose (cell by cell, vertex by
// vertex, with explicit reuse slots) so that it is an independent check of the closed-form, parallel
// formulation used by the HIP kernels (voxels_amd/csrc/tv_core.h).  All citations are file:line into
// /root/reference.
//
// Deliberate, documented deviations (all are undefined behaviour in the reference):
//   * a cell whose material vote finds no child keeps Material = (255, 0) here; the reference leaves
//     the field uninitialised (TransVoxelImpl.cpp:1056-1085 never sets it, :829 skips the store);
//   * an INVALID reuse index (TransVoxelImpl.cpp:1627-1631, assert only) is pushed as 0xFFFFFFFF and
//     its secondary-position/degenerate-triangle reads are skipped instead of indexing out of range.
#include "vxo_api.h"

#include <algorithm>
#include <cfloat>
#include <cmath>
#include <cstdint>
#include <cstdlib>
#include <cstring>
#include <vector>
#include <omp.h>

namespace {

typedef uint8_t u8;
typedef uint32_t u32;

const u32 BE = 16;               // block extent (TransVoxelImpl.cpp:58, VoxelGrid.h:63)
const u32 BCELLS = 4096;
const u32 INVALID = 0xFFFFFFFFu; // TransVoxelImpl.cpp:544
const u8 EMPTY_MAT = 255;        // VoxelGrid.h:16

#include "../voxels_amd/csrc/tv_tables.inc"

// ---------------------------------------------------------------------------------------------
// Grid: dense bytes + the per-block run-length codec state the reference keeps (VoxelGrid.cpp).
// ---------------------------------------------------------------------------------------------
enum { BF_Empty = 1, BF_DistU = 2, BF_MatU = 4, BF_BlendU = 8 }; // VoxelGrid.h:70-79

// VoxelGrid.cpp:610-672.  Returns true when the run-length form is kept.
template <typename T>
bool RleEncode(const T* data, std::vector<T>& out, bool* isEmpty)
{
	out.clear();
	out.push_back(0);
	size_t ctr = 0;
	unsigned counter = 0;
	bool effective = true;
	if (isEmpty) *isEmpty = true;
	const int initial = data[0];
	T last = data[0];
	for (u32 i = 0; i < BCELLS; ++i) {
		const T cur = data[i];
		if (last == cur && counter < 0xFF) {
			++counter;
		} else {
			out[ctr] = (T)(unsigned char)counter;
			out.push_back(last);
			out.push_back(0);
			ctr = out.size() - 1;
			counter = 1;
			last = cur;
			const int sign = initial * (int)last;
			if (sign <= 0 && isEmpty) *isEmpty = false;
			if (out.size() > BCELLS) { effective = false; break; }
		}
	}
	if (effective) {
		out[ctr] = (T)(unsigned char)counter;
		out.push_back(last);
		return true;
	}
	if (isEmpty) *isEmpty = false;
	out.assign(data, data + BCELLS);
	return false;
}

// VoxelGrid.cpp:674-694
template <typename T>
void RleDecode(const T* data, size_t sz, bool raw, T* out)
{
	if (raw) { memcpy(out, data, sz); return; }
	for (size_t i = 0; i + 1 < sz; i += 2) {
		const unsigned len = (unsigned char)data[i];
		const T v = data[i + 1];
		for (unsigned k = 0; k < len; ++k) *out++ = v;
	}
}

struct BlockMeta { u32 flags; u32 szDist, szMat, szBlend; };

struct PGrid {
	u32 n, nb;
	std::vector<int8_t> dist;
	std::vector<u8> mat, blend;
	std::vector<BlockMeta> meta;

	size_t Idx(u32 x, u32 y, u32 z) const { return (size_t(z) * n + y) * n + x; }

	void GatherBlock(const u8* src, u32 bx, u32 by, u32 bz, u8* out) const
	{
		for (u32 z = 0; z < BE; ++z)
		for (u32 y = 0; y < BE; ++y)
			memcpy(out + z * 256 + y * 16, src + Idx(bx * BE, by * BE + y, bz * BE + z), 16);
	}
	void ScatterBlock(u8* dst, u32 bx, u32 by, u32 bz, const u8* in)
	{
		for (u32 z = 0; z < BE; ++z)
		for (u32 y = 0; y < BE; ++y)
			memcpy(dst + Idx(bx * BE, by * BE + y, bz * BE + z), in + z * 256 + y * 16, 16);
	}
	u32 BlockId(u32 bx, u32 by, u32 bz) const { return bx + by * nb + bz * nb * nb; } // VoxelGrid.h:139-144

	// Re-derives flags/sizes of one block from the dense bytes (what PushBlock / Modify*Data do,
	// VoxelGrid.cpp:52-77, :696-741).
	void RefreshBlock(u32 bx, u32 by, u32 bz, bool distance, bool material)
	{
		BlockMeta& m = meta[BlockId(bx, by, bz)];
		u8 tmp[BCELLS];
		if (distance) {
			GatherBlock((const u8*)dist.data(), bx, by, bz, tmp);
			std::vector<char> enc;
			bool empty = false;
			const bool ok = RleEncode<char>((const char*)tmp, enc, &empty);
			m.flags = ok ? (m.flags & ~BF_DistU) : (m.flags | BF_DistU);
			m.flags = empty ? (m.flags | BF_Empty) : (m.flags & ~BF_Empty);
			m.szDist = (u32)enc.size();
		}
		if (material) {
			std::vector<u8> enc;
			GatherBlock(mat.data(), bx, by, bz, tmp);
			bool ok = RleEncode<u8>(tmp, enc, nullptr);
			m.flags = ok ? (m.flags & ~BF_MatU) : (m.flags | BF_MatU);
			m.szMat = (u32)enc.size();
			GatherBlock(blend.data(), bx, by, bz, tmp);
			ok = RleEncode<u8>(tmp, enc, nullptr);
			m.flags = ok ? (m.flags & ~BF_BlendU) : (m.flags | BF_BlendU);
			m.szBlend = (u32)enc.size();
		}
	}
};

PGrid* NewGrid(u32 n)
{
	PGrid* g = new PGrid;
	g->n = n;
	g->nb = n / BE;
	const size_t tot = size_t(n) * n * n;
	g->dist.assign(tot, 0);
	g->mat.assign(tot, 0);
	g->blend.assign(tot, 0);
	g->meta.assign(size_t(g->nb) * g->nb * g->nb, BlockMeta{ 0, 0, 0, 0 });
	return g;
}

// VoxelGrid.cpp:37-40: sign * ceil(|v|), upper bound 127, then C conversion to char.
inline int8_t RoundDistance(float value)
{
	float a = std::ceil(std::fabs(value));
	if (a < -128.f) a = -128.f;
	float b = a * (float)(value > 0 ? 1 : -1);
	if (b > 127.f) b = 127.f;
	return (int8_t)(int)b; // out-of-range wraps exactly like the x86 build of the reference
}

// VoxelGrid.cpp:42-50
inline int8_t ClampGridDistance(int8_t v) { return v > 4 ? 4 : (v < -4 ? -4 : v); }

// ---------------------------------------------------------------------------------------------
// Polygon result (mirrors PolygonMap / PolygonBlock, TransVoxelImpl.h:33-133)
// ---------------------------------------------------------------------------------------------
struct MatInfo { u8 id, blend; };

struct OutBlock {
	u32 id;
	float minc[3], maxc[3];
	std::vector<vxo_vertex> verts;
	std::vector<u32> idx;
	std::vector<vxo_vertex> tverts[6];
	std::vector<u32> tidx[6];
};

struct PSurface {
	float extents[3];
	std::vector<std::vector<OutBlock> > levels;
	std::vector<std::vector<u8> > consistency;               // [level-0 block][cell]
	std::vector<std::vector<std::vector<MatInfo> > > lcache; // [level-1][block][cell]
	u32 stats[20];
	u32 nextId;
};

struct V4 { float x, y, z; u32 w; };
struct V3 { float x, y, z; };

// working data of one block while it is polygonized (TransVoxelImpl.cpp:922-967)
struct WBlock {
	u32 id, coordId, level, mult;
	u32 bx, by, bz;
	std::vector<V4> verts, sec;
	std::vector<V3> normals;
	std::vector<MatInfo> mats;
	std::vector<u32> idx;
	std::vector<V4> tverts[6], tsec[6];
	std::vector<V3> tnormals[6];
	std::vector<MatInfo> tmats[6];
	std::vector<u32> tidx[6];
	u32 trivial, nontrivial, perCase[16];
};

struct Cell {
	int bx, by, bz;     // Base (global voxel coords)
	int lx, ly, lz;     // LocalBase
	u32 blockCoordId, localId;
	int8_t V[8];
	u32 mult, level;
	bool onBoundary;
	MatInfo material;
};

struct Run {
	const PGrid& g;
	const u8* lut;
	const u8* valid;
	PSurface* res;
	int n;
	u32 levelsCount;

	Run(const PGrid& grid, const u8* l, const u8* v, PSurface* r) : g(grid), lut(l), valid(v), res(r), n((int)grid.n) {}

	static int Clampi(int v, int lo, int hi) { return v < lo ? lo : (v > hi ? hi : v); }

	// TransVoxelImpl.cpp:1140-1151, :1194-1201 : every fetch clamps to [0, N-1]
	int8_t D(int x, int y, int z) const { return g.dist[g.Idx(Clampi(x, 0, n - 1), Clampi(y, 0, n - 1), Clampi(z, 0, n - 1))]; }
	MatInfo M(int x, int y, int z) const
	{
		const size_t i = g.Idx(Clampi(x, 0, n - 1), Clampi(y, 0, n - 1), Clampi(z, 0, n - 1));
		return MatInfo{ g.mat[i], g.blend[i] };
	}

	// TransVoxelImpl.cpp:93-103
	static V3 NormalizeFixZero(V3 v)
	{
		const float len = std::sqrt((v.x * v.x + v.y * v.y) + v.z * v.z);
		if (len <= FLT_EPSILON) return V3{ 0.f, 0.f, 0.f };
		return V3{ v.x / len, v.y / len, v.z / len };
	}

	// TransVoxelImpl.cpp:1239-1246 : level-0 central differences, components ordered (x, z, y)
	V3 CalcNormal(int x, int y, int z) const
	{
		V3 v;
		v.x = (float)(D(x + 1, y, z) - D(x - 1, y, z)) * 0.5f;
		v.y = (float)(D(x, y, z + 1) - D(x, y, z - 1)) * 0.5f;
		v.z = (float)(D(x, y + 1, z) - D(x, y - 1, z)) * 0.5f;
		return NormalizeFixZero(v);
	}

	static void CornerOffset(int c, int mult, int& ox, int& oy, int& oz)
	{
		ox = (c & 1) ? mult : 0; oy = (c & 2) ? mult : 0; oz = (c & 4) ? mult : 0; // :710-735
	}

	// TransVoxelImpl.cpp:741-750
	static u32 CaseCode(const int8_t V[8])
	{
		u32 c = 0;
		for (int i = 0; i < 8; ++i) c |= (u32)((V[i] >> 7) & 1) << i;
		return c;
	}

	// TransVoxelImpl.cpp:570-591 : face bit order ZPos,YPos,XPos,ZNeg,YNeg,XNeg
	static bool CornerOnFace(int c, int face)
	{
		switch (face) {
		case 0: return (c & 4) != 0;
		case 1: return (c & 2) != 0;
		case 2: return (c & 1) != 0;
		case 3: return (c & 4) == 0;
		case 4: return (c & 2) == 0;
		default: return (c & 1) == 0;
		}
	}

	// TransVoxelImpl.cpp:593-645 (corner form when c0 == c1)
	static int OnBlockBoundary(const Cell& cell, int c0, int c1)
	{
		if (!cell.onBoundary || cell.mult == 1) return 0;
		int r = 0;
		if (cell.lx == 0 && CornerOnFace(c0, 5) && CornerOnFace(c1, 5)) r |= 1 << 5;
		if (cell.lx == 15 && CornerOnFace(c0, 2) && CornerOnFace(c1, 2)) r |= 1 << 2;
		if (cell.ly == 0 && CornerOnFace(c0, 4) && CornerOnFace(c1, 4)) r |= 1 << 4;
		if (cell.ly == 15 && CornerOnFace(c0, 1) && CornerOnFace(c1, 1)) r |= 1 << 1;
		if (cell.lz == 0 && CornerOnFace(c0, 3) && CornerOnFace(c1, 3)) r |= 1 << 3;
		if (cell.lz == 15 && CornerOnFace(c0, 0) && CornerOnFace(c1, 0)) r |= 1 << 0;
		return r;
	}

	// TransVoxelImpl.cpp:683-708 : inward direction of a face, length = cell size
	static void FaceInward(int face, int mult, float d[3])
	{
		d[0] = d[1] = d[2] = 0.f;
		switch (face) {
		case 0: d[2] = -(float)mult; break;
		case 1: d[1] = -(float)mult; break;
		case 2: d[0] = -(float)mult; break;
		case 3: d[2] = (float)mult; break;
		case 4: d[1] = (float)mult; break;
		default: d[0] = (float)mult; break;
		}
	}

	// TransVoxelImpl.cpp:1473-1482
	static void TransitionDelta(int faces, int mult, float out[3])
	{
		out[0] = out[1] = out[2] = 0.f;
		for (int i = 0; i < 6; ++i) {
			if (faces & (1 << i)) {
				float d[3];
				FaceInward(i, mult, d);
				out[0] += d[0] * 0.25f; out[1] += d[1] * 0.25f; out[2] += d[2] * 0.25f;
			}
		}
	}

	u32 BlocksPerAxis(u32 mult) const { return (g.n >> 4) / mult; } // :379-383

	// TransVoxelImpl.cpp:1052-1086
	Cell MakeCell(const WBlock& b, int lx, int ly, int lz) const
	{
		Cell c;
		c.mult = b.mult; c.level = b.level;
		c.bx = (lx + (int)b.bx * 16) * (int)b.mult;
		c.by = (ly + (int)b.by * 16) * (int)b.mult;
		c.bz = (lz + (int)b.bz * 16) * (int)b.mult;
		c.lx = lx; c.ly = ly; c.lz = lz;
		c.onBoundary = lx == 0 || lx == 15 || ly == 0 || ly == 15 || lz == 0 || lz == 15;
		c.blockCoordId = b.coordId;
		c.localId = (u32)(lz * 256 + ly * 16 + lx);
		c.material = MatInfo{ EMPTY_MAT, 0 };
		for (int i = 0; i < 8; ++i) {
			int ox, oy, oz;
			CornerOffset(i, (int)b.mult, ox, oy, oz);
			c.V[i] = D(c.bx + ox, c.by + oy, c.bz + oz);
		}
		return c;
	}

	// TransVoxelImpl.cpp:753-838
	void CellMaterial(Cell& cell)
	{
		if (cell.mult == 1) {
			res->consistency[cell.blockCoordId][cell.localId] = 1;
			cell.material = M(cell.bx, cell.by, cell.bz);
			return;
		}
		const u32 childMult = cell.mult >> 1;
		const u32 childExt = BE * childMult;
		const u32 cnt = BlocksPerAxis(childMult);
		u8 materials[8];
		int counters[8] = { 0 };
		unsigned blends[8] = { 0 };
		unsigned count = 0;
		for (u32 z = 0; z < 2; ++z)
		for (u32 y = 0; y < 2; ++y)
		for (u32 x = 0; x < 2; ++x) {
			const u32 nx = cell.bx + x * childMult, ny = cell.by + y * childMult, nz = cell.bz + z * childMult;
			const u32 blockId = (nz / childExt) * cnt * cnt + (ny / childExt) * cnt + (nx / childExt);
			const u32 localId = ((nz % childExt) / childMult) * 256 + ((ny % childExt) / childMult) * 16 + (nx % childExt) / childMult;
			MatInfo child{ EMPTY_MAT, 0 };
			if (cell.mult == 2) {
				if (res->consistency[blockId][localId]) child = M((int)nx, (int)ny, (int)nz);
			} else {
				child = res->lcache[cell.level - 2][blockId][localId];
			}
			bool found = false;
			for (unsigned k = 0; k < count; ++k) {
				if (materials[k] == child.id) { ++counters[k]; blends[k] += child.blend; found = true; break; }
			}
			if (!found && child.id != EMPTY_MAT) {
				materials[count] = child.id; counters[count] = 1; blends[count] = child.blend; ++count;
			}
		}
		if (count) {
			unsigned best = 0;
			for (unsigned k = 1; k < count; ++k) if (counters[k] > counters[best]) best = k; // first maximum
			cell.material.id = materials[best];
			cell.material.blend = (u8)(blends[best] / (unsigned)counters[best]);
			res->lcache[cell.level - 1][cell.blockCoordId][cell.localId] = cell.material;
		}
	}

	// TransVoxelImpl.cpp:1484-1509
	void LodChain(int level, int P0[3], int P1[3]) const
	{
		for (int lev = level; lev > 0; --lev) {
			int mid[3];
			for (int k = 0; k < 3; ++k) mid[k] = P0[k] + (P1[k] - P0[k]) / 2;
			const int midV = D(mid[0], mid[1], mid[2]);
			const int p0V = D(P0[0], P0[1], P0[2]);
			if (p0V * midV <= 0) { P1[0] = mid[0]; P1[1] = mid[1]; P1[2] = mid[2]; }
			else { P0[0] = mid[0]; P0[1] = mid[1]; P0[2] = mid[2]; }
		}
	}

	static u8 LerpBlend(long t, long u, u8 b0, u8 b1)
	{
		const float v = ((float)t * (float)b0 + (float)u * (float)b1) / 256.f; // :1699, :2085
		return (u8)(int)v;
	}

	// TransVoxelImpl.cpp:1450-1467
	u32 VertexFromCorner(WBlock& b, const Cell& cell, int corner)
	{
		int ox, oy, oz;
		CornerOffset(corner, (int)cell.mult, ox, oy, oz);
		const int x = cell.bx + ox, y = cell.by + oy, z = cell.bz + oz;
		b.normals.push_back(CalcNormal(x, y, z));
		const MatInfo mine = M(x, y, z);
		b.mats.push_back(cell.material.id != mine.id ? cell.material : mine);
		b.verts.push_back(V4{ (float)x * 256.f, (float)y * 256.f, (float)z * 256.f, (u32)OnBlockBoundary(cell, corner, corner) });
		return (u32)b.verts.size() - 1;
	}

	// TransVoxelImpl.cpp:1529-1750
	void PolygonizeBlock(WBlock& b)
	{
		std::vector<u32> slots(size_t(BCELLS) * 4, INVALID);
		u32 vidx[16];
		unsigned mask = 0;
		for (int cz = 0; cz < 16; ++cz) {
			mask &= 0xD;
			for (int cy = 0; cy < 16; ++cy) {
				mask &= 0xE;
				for (int cx = 0; cx < 16; ++cx) {
					Cell cell = MakeCell(b, cx, cy, cz);
					u32* mySlots = &slots[size_t(cell.localId) * 4];
					const u32 code = CaseCode(cell.V);
					if (code == 0 || code == 255) { ++b.trivial; continue; }
					CellMaterial(cell);
					++b.nontrivial;
					const u32 cls = TVT_REG_CLASS[code];
					++b.perCase[cls];
					const unsigned char* cd = &TVT_REG_CELL[cls * 16];
					const unsigned short* vd = &TVT_REG_VERT[code * 12];
					const int nVerts = cd[0] >> 4, nTris = cd[0] & 15;
					for (int vi = 0; vi < nVerts; ++vi) {
						const unsigned w = vd[vi];
						int direction = (int)(w >> 12);
						int slot = (int)((w >> 8) & 15);
						const int v0 = (int)((w >> 4) & 15), v1 = (int)(w & 15);
						bool checkReuse = true, create = true;
						long t = ((long)cell.V[v1] * 256) / ((long)cell.V[v1] - (long)cell.V[v0]);
						const bool endpoint = (t & 0xFF) == 0;
						if (endpoint) {
							if (t == 0 && v1 == 7) checkReuse = false;
							if (checkReuse) direction = ((t == 0) ? v1 : v0) ^ 7;
							slot = 0;
						}
						if (((unsigned)direction & mask) == (unsigned)direction && checkReuse) {
							const int rx = cx - (direction & 1), ry = cy - ((direction >> 1) & 1), rz = cz - ((direction >> 2) & 1);
							const u32 reuse = slots[size_t(rz * 256 + ry * 16 + rx) * 4 + slot];
							bool same = true;
							if (reuse != INVALID) same = b.mats[reuse].id == cell.material.id;
							if (same) {
								vidx[vi] = reuse;
								if (endpoint && reuse == INVALID) vidx[vi] = VertexFromCorner(b, cell, v0); // :1635-1639
								create = false;
							}
						}
						if (create) {
							if (endpoint) {
								const u32 index = VertexFromCorner(b, cell, (t == 0) ? v1 : v0);
								if (t == 0 && v1 == 7) mySlots[slot] = index;
								vidx[vi] = index;
							} else {
								int P0[3], P1[3], o[3];
								CornerOffset(v0, (int)cell.mult, o[0], o[1], o[2]);
								P0[0] = cell.bx + o[0]; P0[1] = cell.by + o[1]; P0[2] = cell.bz + o[2];
								CornerOffset(v1, (int)cell.mult, o[0], o[1], o[2]);
								P1[0] = cell.bx + o[0]; P1[1] = cell.by + o[1]; P1[2] = cell.bz + o[2];
								if (b.level) {
									LodChain((int)b.level, P0, P1);
									const long p0 = D(P0[0], P0[1], P0[2]), p1 = D(P1[0], P1[1], P1[2]);
									t = (p0 != p1) ? (p1 * 256) / (p1 - p0) : 0;
								}
								const V3 N0 = CalcNormal(P0[0], P0[1], P0[2]), N1 = CalcNormal(P1[0], P1[1], P1[2]);
								MatInfo M0 = M(P0[0], P0[1], P0[2]);
								const MatInfo M1 = M(P1[0], P1[1], P1[2]);
								const long u = 256 - t;
								const float ft = (float)t, fu = (float)u;
								V4 Q;
								Q.x = ft * (float)P0[0] + fu * (float)P1[0];
								Q.y = ft * (float)P0[1] + fu * (float)P1[1];
								Q.z = ft * (float)P0[2] + fu * (float)P1[2];
								Q.w = (u32)OnBlockBoundary(cell, v0, v1);
								b.verts.push_back(Q);
								if (M0.id == M1.id && M0.id == cell.material.id) {
									M0.blend = LerpBlend(t, u, M0.blend, M1.blend);
									b.mats.push_back(M0);
								} else {
									b.mats.push_back(cell.material);
								}
								const float wt = ft / 256.f, wu = fu / 256.f;
								b.normals.push_back(NormalizeFixZero(V3{ N0.x * wt + N1.x * wu, N0.y * wt + N1.y * wu, N0.z * wt + N1.z * wu }));
								const u32 index = (u32)b.verts.size() - 1;
								if (direction == 8) mySlots[slot] = index;
								vidx[vi] = index;
							}
						}
					}
					b.sec.resize(b.verts.size(), V4{ 0.f, 0.f, 0.f, 0u });
					for (int tr = 0; tr < nTris * 3; ++tr) {
						const u32 vi = vidx[cd[1 + tr]];
						b.idx.push_back(vi);
						if (vi == INVALID) continue;
						const V4 v = b.verts[vi];
						if ((int)v.w > 0) {
							float d[3];
							TransitionDelta((int)v.w, (int)cell.mult, d);
							b.sec[vi] = V4{ v.x + d[0] * 256.f, v.y + d[1] * 256.f, v.z + d[2] * 256.f, v.w };
						} else {
							b.sec[vi] = v;
						}
					}
					mask |= 1;
				}
				if (mask & 1) mask |= 2;
			}
			if (mask & 2) mask |= 4;
		}
	}

	// TransVoxelImpl.cpp:1754-2131
	void TransitionCells(WBlock& b)
	{
		static const int coeffs[9] = { 0x01, 0x02, 0x04, 0x80, 0x100, 0x08, 0x40, 0x20, 0x10 };
		static const int lowFace[6] = { 3, 4, 5, 0, 1, 2 };   // ZNeg,YNeg,XNeg,ZPos,YPos,XPos (:1795-1803)
		static const int reverseWinding[6] = { 0, 1, 0, 1, 0, 1 };
		static const int faceCorners[6][4] = { { 4, 5, 6, 7 }, { 2, 3, 6, 7 }, { 1, 3, 5, 7 }, { 0, 1, 2, 3 }, { 0, 1, 4, 5 }, { 0, 2, 4, 6 } };
		const int mult = (int)b.mult, half = mult >> 1;
		const int cnt = (int)BlocksPerAxis(b.mult);
		for (int f = 0; f < 6; ++f) {
			const int axis = (f % 3 == 0) ? 2 : ((f % 3 == 1) ? 1 : 0); // face normal axis: z, y, x
			const bool positive = f >= 3;
			int nbc[3] = { (int)b.bx, (int)b.by, (int)b.bz };
			// :1829-1835 (a delta of -0.5 fails the >= 0 test only for block coordinate 0)
			if (positive) { if (nbc[axis] + 1 >= cnt) continue; }
			else { if (nbc[axis] == 0) continue; }

			std::vector<u32> cur(16 * 10, INVALID), prev(16 * 10, INVALID);
			unsigned mask = 0;
			// in-plane axes: u = column axis, v = row axis
			const int ua = (axis == 0) ? 1 : 0;
			const int va = (axis == 2) ? 1 : 2;
			for (int row = 0; row < 16; ++row) {
				mask &= 2;
				for (int col = 0; col < 16; ++col) {
					int lc[3];
					lc[ua] = col; lc[va] = row; lc[axis] = positive ? 15 : 0;
					Cell low = MakeCell(b, lc[0], lc[1], lc[2]);
					CellMaterial(low);

					// the 13 samples: 3x3 full-resolution samples on the boundary plane + the 4 low-res corners
					int base[3] = { low.bx, low.by, low.bz };
					const int plane = base[axis] + (positive ? mult : 0);
					int8_t values[13];
					int coords[13][3];
					for (int j = 0; j < 3; ++j)
					for (int i = 0; i < 3; ++i) {
						int p[3];
						p[ua] = base[ua] + i * half; p[va] = base[va] + j * half; p[axis] = plane;
						const int k = i + 3 * j;
						coords[k][0] = p[0]; coords[k][1] = p[1]; coords[k][2] = p[2];
						values[k] = D(p[0], p[1], p[2]);
					}
					const int* lowIds = faceCorners[lowFace[f]];
					for (int k = 0; k < 4; ++k) {
						int o[3];
						CornerOffset(lowIds[k], mult, o[0], o[1], o[2]);
						coords[9 + k][0] = low.bx + o[0]; coords[9 + k][1] = low.by + o[1]; coords[9 + k][2] = low.bz + o[2];
						values[9 + k] = low.V[lowIds[k]];
					}
					// :1906 sample 8 is taken from the low-res cell (same point as hi-res (2,2))
					values[8] = values[12];
					coords[8][0] = coords[12][0]; coords[8][1] = coords[12][1]; coords[8][2] = coords[12][2];

					int caseCode = 0;
					for (int ci = 0; ci < 9; ++ci) caseCode += ((values[ci] >> 7) & 1) * coeffs[ci];
					if (caseCode == 0 || caseCode == 511) continue;

					const unsigned cls = TVT_TR_CLASS[caseCode];
					const int invert = (int)(cls >> 7);
					const unsigned char* cd = &TVT_TR_CELL[(cls & 0x7F) * 40];
					const unsigned short* vd = &TVT_TR_VERT[caseCode * 12];
					const int nVerts = cd[0] >> 4, nTris = cd[0] & 15;
					float moveDir[3];
					FaceInward(lowFace[f], mult, moveDir);

					u32 cellIdx[12];
					for (int vi = 0; vi < nVerts; ++vi) {
						const unsigned w = vd[vi];
						const int v0 = (int)((w >> 4) & 15), v1 = (int)(w & 15);
						int dir = (int)(w >> 12), slot = (int)((w >> 8) & 15);
						long t = ((long)values[v1] * 256) / ((long)values[v1] - (long)values[v0]);
						bool reused = false, addForReuse = true;
						const int corner = (t == 0) ? v1 : v0;
						const bool endpoint = (t & 0xFF) == 0;
						if (endpoint) { dir = TVT_TR_CORNER[corner] >> 4; slot = TVT_TR_CORNER[corner] & 15; }
						if (((unsigned)dir & mask) == (unsigned)dir) {
							addForReuse = false;
							const std::vector<u32>& rowSlots = (dir & 2) ? prev : cur;
							const int rc = col - (dir & 1);
							const u32 found = rowSlots[size_t(rc) * 10 + slot];
							if (found != INVALID && b.tmats[f][found].id == low.material.id) {
								reused = true;
								cellIdx[vi] = found;
							}
						}
						if (reused) continue;

						float P0[3] = { (float)coords[v0][0], (float)coords[v0][1], (float)coords[v0][2] };
						float P1[3] = { (float)coords[v1][0], (float)coords[v1][1], (float)coords[v1][2] };
						int I0[3] = { coords[v0][0], coords[v0][1], coords[v0][2] };
						int I1[3] = { coords[v1][0], coords[v1][1], coords[v1][2] };
						V3 N0{ 0, 0, 0 }, N1{ 0, 0, 0 };
						long u = 0;
						int adjacency = 0;
						if (endpoint) {
							if (t == 0) {
								u = 256;
								N1 = CalcNormal(I1[0], I1[1], I1[2]);
								if (v1 >= 9) adjacency = OnBlockBoundary(low, lowIds[v1 - 9], lowIds[v1 - 9]);
							} else {
								u = 0; t = 256;
								N0 = CalcNormal(I0[0], I0[1], I0[2]);
								if (v0 >= 9) adjacency = OnBlockBoundary(low, lowIds[v0 - 9], lowIds[v0 - 9]);
							}
						} else {
							const int lodOfEdge = (v0 >= 9) ? (int)b.level : (int)b.level - 1;
							if (lodOfEdge > 0) {
								LodChain(lodOfEdge, I0, I1);
								const long p0 = D(I0[0], I0[1], I0[2]), p1 = D(I1[0], I1[1], I1[2]);
								t = (p0 != p1) ? (p1 * 256) / (p1 - p0) : 0;
								for (int k = 0; k < 3; ++k) { P0[k] = (float)I0[k]; P1[k] = (float)I1[k]; }
							}
							u = 256 - t;
							N0 = CalcNormal(I0[0], I0[1], I0[2]);
							N1 = CalcNormal(I1[0], I1[1], I1[2]);
							if (v0 >= 9 && v1 >= 9) adjacency = OnBlockBoundary(low, lowIds[v0 - 9], lowIds[v1 - 9]);
						}
						MatInfo M0 = M(I0[0], I0[1], I0[2]);
						const MatInfo M1 = M(I1[0], I1[1], I1[2]);

						float S0[3] = { P0[0], P0[1], P0[2] }, S1[3] = { P1[0], P1[1], P1[2] };
						if (v0 >= 9 || v1 >= 9) {
							float delta[3];
							TransitionDelta(adjacency, mult, delta);
							const bool simple = adjacency == (1 << lowFace[f]);
							if (v0 >= 9) {
								for (int k = 0; k < 3; ++k) S0[k] += delta[k];
								if (simple) for (int k = 0; k < 3; ++k) P0[k] += 0.25f * moveDir[k];
							}
							if (v1 >= 9) {
								for (int k = 0; k < 3; ++k) S1[k] += delta[k];
								if (simple) for (int k = 0; k < 3; ++k) P1[k] += 0.25f * moveDir[k];
							}
						}
						const float ft = (float)t, fu = (float)u;
						b.tverts[f].push_back(V4{ ft * P0[0] + fu * P1[0], ft * P0[1] + fu * P1[1], ft * P0[2] + fu * P1[2], 0u });
						b.tsec[f].push_back(V4{ ft * S0[0] + fu * S1[0], ft * S0[1] + fu * S1[1], ft * S0[2] + fu * S1[2], (u32)adjacency });
						const float wt = ft / 256.f, wu = fu / 256.f;
						b.tnormals[f].push_back(NormalizeFixZero(V3{ N0.x * wt + N1.x * wu, N0.y * wt + N1.y * wu, N0.z * wt + N1.z * wu }));
						if (M0.id == M1.id && M0.id == low.material.id) {
							M0.blend = LerpBlend(t, u, M0.blend, M1.blend);
							b.tmats[f].push_back(M0);
						} else {
							b.tmats[f].push_back(low.material);
						}
						const u32 index = (u32)b.tverts[f].size() - 1;
						cellIdx[vi] = index;
						if (addForReuse && dir == 8) cur[size_t(col) * 10 + slot] = index;
					}
					for (int tr = 0; tr < nTris; ++tr) {
						u32 a = cellIdx[cd[1 + tr * 3]], bb = cellIdx[cd[2 + tr * 3]], c = cellIdx[cd[3 + tr * 3]];
						if (invert ^ reverseWinding[f]) std::swap(bb, c);
						b.tidx[f].push_back(a); b.tidx[f].push_back(bb); b.tidx[f].push_back(c);
					}
					mask |= 1;
				}
				prev.swap(cur);
				std::fill(cur.begin(), cur.end(), INVALID);
				mask |= 2;
			}
		}
	}

	// TransVoxelImpl.cpp:1511-1527
	bool BlockAndNeighboursEmpty(const WBlock& b) const
	{
		const int nb = (int)g.nb;
		for (int z = -1; z < 2; ++z)
		for (int y = -1; y < 2; ++y)
		for (int x = -1; x < 2; ++x) {
			const int cx = Clampi((int)b.bx + x, 0, nb - 1), cy = Clampi((int)b.by + y, 0, nb - 1), cz = Clampi((int)b.bz + z, 0, nb - 1);
			if (!(g.meta[g.BlockId(cx, cy, cz)].flags & BF_Empty)) return false;
		}
		return true;
	}

	// TransVoxelImpl.cpp:1248-1264, :1330-1369
	vxo_vertex Finalize(const V4& p, const V4& s, const V3& nrm, MatInfo m) const
	{
		vxo_vertex o;
		memset(&o, 0, sizeof(o));
		const float k = 1.f / 256.f;
		o.pos[0] = p.x * k; o.pos[1] = p.z * k; o.pos[2] = p.y * k;
		o.nrm[0] = nrm.x; o.nrm[1] = nrm.y; o.nrm[2] = nrm.z;
		u32 flags = s.w;
		if (flags) flags = (flags >> 3) | ((flags & 7) << 3);
		o.sec[0] = s.x * k; o.sec[1] = s.z * k; o.sec[2] = s.y * k;
		memcpy(&o.sec[3], &flags, 4);
		if (valid[m.id]) {
			const u8* e = lut + size_t(m.id) * 6;
			o.tex[3] = e[1]; o.tex[7] = e[0]; o.tex[6] = e[2]; // Txz, Tpy, Tny
			o.tex[2] = e[4]; o.tex[5] = e[3]; o.tex[4] = e[5]; // Uxz, Upy, Uny
			o.tex[1] = m.blend;
		}
		return o;
	}

	// TransVoxelImpl.cpp:1266-1428
	void PushBlock(const WBlock& b)
	{
		if (b.verts.empty()) return;
		std::vector<OutBlock>& out = res->levels[b.level];
		out.push_back(OutBlock());
		OutBlock& ob = out.back();
		ob.id = b.id;
		const float ext = (float)(b.mult * BE);
		const float c[3] = { (float)b.bx * ext, (float)b.by * ext, (float)b.bz * ext };
		ob.minc[0] = c[0]; ob.minc[1] = c[2]; ob.minc[2] = c[1];
		ob.maxc[0] = c[0] + ext; ob.maxc[1] = c[2] + ext; ob.maxc[2] = c[1] + ext;
		for (size_t i = 0; i + 2 < b.idx.size(); i += 3) {
			const u32 i0 = b.idx[i], i1 = b.idx[i + 1], i2 = b.idx[i + 2];
			bool keep = true;
			if (i0 != INVALID && i1 != INVALID && i2 != INVALID) {
				const V4 &v0 = b.verts[i0], &v1 = b.verts[i1], &v2 = b.verts[i2];
				const float ax = v1.x - v0.x, ay = v1.y - v0.y, az = v1.z - v0.z;
				const float bx = v2.x - v0.x, by = v2.y - v0.y, bz = v2.z - v0.z;
				const float cx = ay * bz - by * az, cy = az * bx - bz * ax, cz = ax * by - bx * ay;
				const float len2 = (cx * cx + cy * cy) + cz * cz;
				keep = len2 >= FLT_EPSILON;
			}
			if (keep) { ob.idx.push_back(i0); ob.idx.push_back(i1); ob.idx.push_back(i2); }
			else ++res->stats[3];
		}
		ob.verts.reserve(b.verts.size());
		for (size_t i = 0; i < b.verts.size(); ++i) ob.verts.push_back(Finalize(b.verts[i], b.sec[i], b.normals[i], b.mats[i]));
		for (int f = 0; f < 6; ++f) {
			ob.tverts[f].reserve(b.tverts[f].size());
			for (size_t i = 0; i < b.tverts[f].size(); ++i)
				ob.tverts[f].push_back(Finalize(b.tverts[f][i], b.tsec[f][i], b.tnormals[f][i], b.tmats[f][i]));
			ob.tidx[f] = b.tidx[f];
		}
	}

	// TransVoxelImpl.cpp:385-538
	void Execute(bool modify, const float* dirtyMin, const float* dirtyMax, std::vector<u32>* modifiedIds)
	{
		if (!modify) {
			res->extents[0] = (float)g.n; res->extents[1] = (float)g.n; res->extents[2] = (float)g.n;
		} else {
			memset(res->stats, 0, sizeof(res->stats));
		}
		levelsCount = 0;
		for (u32 v = g.n >> 4; v >>= 1;) ++levelsCount;
		++levelsCount;

		for (u32 level = 0; level < levelsCount; ++level) {
			if (res->levels.size() <= level) res->levels.push_back(std::vector<OutBlock>());
			const u32 mult = 1u << level;
			const u32 cnt = BlocksPerAxis(mult);
			std::vector<WBlock> blocks;
			if (!modify) {
				blocks.reserve(size_t(cnt) * cnt * cnt);
				for (u32 z = 0; z < cnt; ++z)
				for (u32 y = 0; y < cnt; ++y)
				for (u32 x = 0; x < cnt; ++x) {
					WBlock b;
					b.id = res->nextId++; b.coordId = z * cnt * cnt + y * cnt + x; b.level = level; b.mult = mult;
					b.bx = x; b.by = y; b.bz = z;
					b.trivial = b.nontrivial = 0; memset(b.perCase, 0, sizeof(b.perCase));
					blocks.push_back(std::move(b));
				}
				const size_t tot = size_t(cnt) * cnt * cnt;
				if (level == 0) res->consistency.assign(tot, std::vector<u8>(BCELLS, 0));
				else res->lcache.push_back(std::vector<std::vector<MatInfo> >(tot, std::vector<MatInfo>(BCELLS, MatInfo{ EMPTY_MAT, 0 })));
			} else {
				// :429-465 — everything in output (Y-up) coordinates
				const float bm = (float)(mult * BE);
				float lo[3], hi[3];
				for (int k = 0; k < 3; ++k) {
					lo[k] = std::floor(dirtyMin[k] / bm - 1.0f) * bm;
					hi[k] = std::floor(dirtyMax[k] / bm + 2.0f) * bm;
					lo[k] = std::min(std::max(lo[k], 0.f), res->extents[k]);
					hi[k] = std::min(std::max(hi[k], 0.f), res->extents[k]);
				}
				std::vector<OutBlock>& old = res->levels[level];
				old.erase(std::remove_if(old.begin(), old.end(), [&](const OutBlock& ob) {
					return ob.minc[0] >= lo[0] && ob.minc[1] >= lo[1] && ob.minc[2] >= lo[2]
						&& ob.minc[0] < hi[0] && ob.minc[1] < hi[1] && ob.minc[2] < hi[2];
				}), old.end());
				const float minB[3] = { lo[0] / bm, lo[1] / bm, lo[2] / bm }, maxB[3] = { hi[0] / bm, hi[1] / bm, hi[2] / bm };
				for (u32 z = (u32)minB[1]; z < (u32)maxB[1]; ++z)
				for (u32 y = (u32)minB[2]; y < (u32)maxB[2]; ++y)
				for (u32 x = (u32)minB[0]; x < (u32)maxB[0]; ++x) {
					WBlock b;
					b.id = res->nextId++; b.coordId = z * cnt * cnt + y * cnt + x; b.level = level; b.mult = mult;
					b.bx = x; b.by = y; b.bz = z;
					b.trivial = b.nontrivial = 0; memset(b.perCase, 0, sizeof(b.perCase));
					modifiedIds->push_back(b.id);
					blocks.push_back(std::move(b));
				}
			}

			const int nBlocks = (int)blocks.size();
			#pragma omp parallel for schedule(dynamic, 1)
			for (int i = 0; i < nBlocks; ++i) {
				WBlock& b = blocks[i];
				const bool empty = b.level == 0 && BlockAndNeighboursEmpty(b);
				if (!empty) {
					PolygonizeBlock(b);
					if (b.level && b.level != levelsCount - 1) TransitionCells(b);
				}
			}
			res->stats[0] += (u32)blocks.size();
			for (const WBlock& b : blocks) {
				res->stats[1] += b.trivial;
				res->stats[2] += b.nontrivial;
				for (int k = 0; k < 16; ++k) res->stats[4 + k] += b.perCase[k];
			}
			for (const WBlock& b : blocks) PushBlock(b);
		}
	}
};

} // namespace

struct vxo_grid { PGrid* g; };
struct vxo_surface { PSurface* s; };

extern "C" {

const char* vxo_kind(void) { return "port"; }

vxo_grid* vxo_grid_from_dense(uint32_t n, const int8_t* dist, const uint8_t* mat, const uint8_t* blend)
{
	PGrid* g = NewGrid(n);
	const size_t tot = size_t(n) * n * n;
	memcpy(g->dist.data(), dist, tot);
	if (mat) memcpy(g->mat.data(), mat, tot);
	if (blend) memcpy(g->blend.data(), blend, tot);
	for (u32 z = 0; z < g->nb; ++z) for (u32 y = 0; y < g->nb; ++y) for (u32 x = 0; x < g->nb; ++x) g->RefreshBlock(x, y, z, true, true);
	return new vxo_grid{ g };
}

vxo_grid* vxo_grid_from_float(uint32_t n, const float* values, const uint8_t* mat, const uint8_t* blend)
{
	PGrid* g = NewGrid(n);
	const size_t tot = size_t(n) * n * n;
	for (size_t i = 0; i < tot; ++i) g->dist[i] = ClampGridDistance(RoundDistance(values[i])); // VoxelGrid.cpp:125
	if (mat) memcpy(g->mat.data(), mat, tot);
	if (blend) memcpy(g->blend.data(), blend, tot);
	for (u32 z = 0; z < g->nb; ++z) for (u32 y = 0; y < g->nb; ++y) for (u32 x = 0; x < g->nb; ++x) g->RefreshBlock(x, y, z, true, true);
	return new vxo_grid{ g };
}

// VoxelGrid(unsigned w, const char* heightmap), src/VoxelGrid.cpp:159-213: distance = clamp(slice - 127 - height(row, column),
// -127, 127) squeezed to the grid's +-4 range (toGridDistValue), materials and blends zero
vxo_grid* vxo_grid_from_heightmap(uint32_t n, const char* heightmap)
{
	PGrid* g = NewGrid(n);
	for (u32 z = 0; z < n; ++z)
	for (u32 y = 0; y < n; ++y)
	for (u32 x = 0; x < n; ++x) {
		int h = ((int)z - 127) - (int)heightmap[(size_t)y * n + x];
		h = h < -127 ? -127 : (h > 127 ? 127 : h);
		g->dist[(size_t(z) * n + y) * n + x] = ClampGridDistance((int8_t)h);
	}
	for (u32 z = 0; z < g->nb; ++z) for (u32 y = 0; y < g->nb; ++y) for (u32 x = 0; x < g->nb; ++x) g->RefreshBlock(x, y, z, true, true);
	return new vxo_grid{ g };
}

void vxo_grid_destroy(vxo_grid* g) { if (g) { delete g->g; delete g; } }
uint32_t vxo_grid_size(const vxo_grid* g) { return g->g->n; }

void vxo_grid_read_dense(const vxo_grid* g, int8_t* dist, uint8_t* mat, uint8_t* blend)
{
	const size_t tot = g->g->dist.size();
	if (dist) memcpy(dist, g->g->dist.data(), tot);
	if (mat) memcpy(mat, g->g->mat.data(), tot);
	if (blend) memcpy(blend, g->g->blend.data(), tot);
}

void vxo_grid_block_flags(const vxo_grid* g, uint8_t* out)
{
	for (size_t i = 0; i < g->g->meta.size(); ++i) out[i] = (g->g->meta[i].flags & BF_Empty) ? 1 : 0;
}

uint32_t vxo_grid_memory_size(vxo_grid* g)
{
	size_t t = 0;
	for (const BlockMeta& m : g->g->meta) t += m.szDist + m.szMat + m.szBlend;
	return (uint32_t)t;
}

// VoxelGrid.cpp:331-366
static void TouchedBlocks(const PGrid& g, const float pos[3], const float ext[3], std::vector<u32>& out)
{
	const float cmin[3] = { pos[0] - ext[0], pos[1] - ext[1], pos[2] - ext[2] };
	const float cmax[3] = { pos[0] + ext[0], pos[1] + ext[1], pos[2] + ext[2] };
	for (u32 z = 0; z < g.nb; ++z) for (u32 y = 0; y < g.nb; ++y) for (u32 x = 0; x < g.nb; ++x) {
		const float bmin[3] = { (float)(x * 16), (float)(y * 16), (float)(z * 16) };
		bool hit = true;
		for (int k = 0; k < 3; ++k) {
			const float bmax = (bmin[k] + 8.f) + 8.f;
			if (cmin[k] > bmax || bmin[k] > cmax[k]) hit = false;
		}
		if (hit) out.push_back(g.BlockId(x, y, z));
	}
}

static float Clampf(float v, float lo, float hi) { return v < lo ? lo : (v > hi ? hi : v); }

// VoxelGrid.cpp:368-386
static void TouchedSection(const float pos[3], const float ext[3], const float bmin[3], float start[3], float end[3])
{
	for (int k = 0; k < 3; ++k) {
		const float p = pos[k] - ext[k] / 2;
		start[k] = Clampf(p, bmin[k], bmin[k] + 16.f) - bmin[k];
		end[k] = Clampf(p + ext[k], bmin[k], bmin[k] + 16.f) - bmin[k];
	}
}

static void ModifiedBox(const PGrid& g, const float pos[3], const float ext[3], float out_min[3], float out_max[3])
{
	// VoxelGrid.cpp:477-487 : returned in output (Y-up) order
	const float p[3] = { pos[0] - ext[0] / 2.0f, pos[1] - ext[1] / 2.0f, pos[2] - ext[2] / 2.0f };
	out_min[0] = std::max(0.f, p[0]); out_min[1] = std::max(0.f, p[2]); out_min[2] = std::max(0.f, p[1]);
	out_max[0] = std::min((float)g.n, out_min[0] + ext[0]);
	out_max[1] = std::min((float)g.n, out_min[1] + ext[2]);
	out_max[2] = std::min((float)g.n, out_min[2] + ext[1]);
}

// VoxelGrid.cpp:388-488 with the harness' ball brush (oracle/ref_harness.cpp BallSurface)
void vxo_grid_inject_ball(vxo_grid* gh, const float pos[3], const float ext[3], float radius, int type,
	float out_min[3], float out_max[3])
{
	PGrid& g = *gh->g;
	std::vector<u32> touched;
	TouchedBlocks(g, pos, ext, touched);
	for (u32 id : touched) {
		const u32 bx = id % g.nb, by = (id / g.nb) % g.nb, bz = id / (g.nb * g.nb);
		const float bmin[3] = { (float)(bx * 16), (float)(by * 16), (float)(bz * 16) };
		float bs[3], be[3];
		TouchedSection(pos, ext, bmin, bs, be);
		const float s0[3] = { bmin[0] + bs[0] - pos[0], bmin[1] + bs[1] - pos[1], bmin[2] + bs[2] - pos[2] };
		const float s1[3] = { bmin[0] + be[0] - pos[0], bmin[1] + be[1] - pos[1], bmin[2] + be[2] - pos[2] };
		std::vector<float> vals;
		for (float z = s0[2]; z < s1[2]; z += 1.f)
		for (float y = s0[1]; y < s1[1]; y += 1.f)
		for (float x = s0[0]; x < s1[0]; x += 1.f)
			vals.push_back(sqrtf(x * x + y * y + z * z) - radius);
		size_t vi = 0;
		for (float z = bs[2]; z < be[2]; ++z)
		for (float y = bs[1]; y < be[1]; ++y)
		for (float x = bs[0]; x < be[0]; ++x) {
			const size_t i = g.Idx(bx * 16 + (unsigned)x, by * 16 + (unsigned)y, bz * 16 + (unsigned)z);
			const float value = (float)g.dist[i];
			const float sv = vals[vi++];
			float r;
			if (type == 0) r = std::min(value, sv);
			else if (type == 1) r = std::max(value, sv);
			else r = std::max(-sv, value);
			g.dist[i] = RoundDistance(r);
		}
		g.RefreshBlock(bx, by, bz, true, false);
	}
	ModifiedBox(g, pos, ext, out_min, out_max);
}

// VoxelGrid.cpp:490-584
void vxo_grid_inject_material(vxo_grid* gh, const float pos[3], const float ext[3], uint8_t material,
	int add, float out_min[3], float out_max[3])
{
	PGrid& g = *gh->g;
	std::vector<u32> touched;
	TouchedBlocks(g, pos, ext, touched);
	const float coeff = (ext[0] / 2.0f) * 0.75f;
	for (u32 id : touched) {
		const u32 bx = id % g.nb, by = (id / g.nb) % g.nb, bz = id / (g.nb * g.nb);
		const float bmin[3] = { (float)(bx * 16), (float)(by * 16), (float)(bz * 16) };
		float bs[3], be[3];
		TouchedSection(pos, ext, bmin, bs, be);
		for (float z = bs[2]; z < be[2]; ++z)
		for (float y = bs[1]; y < be[1]; ++y)
		for (float x = bs[0]; x < be[0]; ++x) {
			const float cx = x + bmin[0] - pos[0], cy = y + bmin[1] - pos[1], cz = z + bmin[2] - pos[2];
			const float dist = std::sqrt((cx * cx + cy * cy) + cz * cz) / coeff;
			const u8 outBlend = (u8)(std::min(1.f, std::max(0.f, (1 - dist))) * 255.f);
			const size_t i = g.Idx(bx * 16 + (unsigned)x, by * 16 + (unsigned)y, bz * 16 + (unsigned)z);
			if (g.mat[i] == material) {
				g.blend[i] = (u8)std::max(0, std::min(255, (add ? 1 : -1) * (int)outBlend + (int)g.blend[i]));
			} else {
				g.mat[i] = material;
				g.blend[i] = outBlend;
			}
		}
		g.RefreshBlock(bx, by, bz, false, true);
	}
	ModifiedBox(g, pos, ext, out_min, out_max);
}

// VoxelGrid.cpp:269-315
size_t vxo_grid_pack(const vxo_grid* gh, char* out, size_t cap)
{
	const PGrid& g = *gh->g;
	std::vector<char> data;
	auto put32 = [&data](u32 v) { const char* p = (const char*)&v; data.insert(data.end(), p, p + 4); };
	put32(1); put32(g.n); put32(g.n); put32(g.n);
	for (const BlockMeta& m : g.meta) { put32(m.szDist); put32(m.szMat); put32(m.szBlend); }
	u8 tmp[BCELLS];
	for (u32 z = 0; z < g.nb; ++z) for (u32 y = 0; y < g.nb; ++y) for (u32 x = 0; x < g.nb; ++x) {
		const BlockMeta& m = g.meta[g.BlockId(x, y, z)];
		put32(m.flags);
		std::vector<char> ed;
		g.GatherBlock((const u8*)g.dist.data(), x, y, z, tmp);
		RleEncode<char>((const char*)tmp, ed, nullptr);
		data.insert(data.end(), ed.begin(), ed.end());
		std::vector<u8> eu;
		g.GatherBlock(g.mat.data(), x, y, z, tmp);
		RleEncode<u8>(tmp, eu, nullptr);
		data.insert(data.end(), (const char*)eu.data(), (const char*)eu.data() + eu.size());
		g.GatherBlock(g.blend.data(), x, y, z, tmp);
		RleEncode<u8>(tmp, eu, nullptr);
		data.insert(data.end(), (const char*)eu.data(), (const char*)eu.data() + eu.size());
	}
	if (out) memcpy(out, data.data(), std::min(cap, data.size()));
	return data.size();
}

// VoxelGrid.cpp:215-267
vxo_grid* vxo_grid_load(const char* blob, size_t size)
{
	(void)size;
	const char* p = blob;
	auto get32 = [&p]() { u32 v; memcpy(&v, p, 4); p += 4; return v; };
	if (get32() != 1) return nullptr;
	const u32 w = get32(); get32(); get32();
	PGrid* g = NewGrid(w);
	for (BlockMeta& m : g->meta) { m.szDist = get32(); m.szMat = get32(); m.szBlend = get32(); }
	u8 tmp[BCELLS];
	for (u32 z = 0; z < g->nb; ++z) for (u32 y = 0; y < g->nb; ++y) for (u32 x = 0; x < g->nb; ++x) {
		BlockMeta& m = g->meta[g->BlockId(x, y, z)];
		m.flags = get32();
		RleDecode<char>(p, m.szDist, (m.flags & BF_DistU) != 0, (char*)tmp); p += m.szDist;
		g->ScatterBlock((u8*)g->dist.data(), x, y, z, tmp);
		RleDecode<u8>((const u8*)p, m.szMat, (m.flags & BF_MatU) != 0, tmp); p += m.szMat;
		g->ScatterBlock(g->mat.data(), x, y, z, tmp);
		RleDecode<u8>((const u8*)p, m.szBlend, (m.flags & BF_BlendU) != 0, tmp); p += m.szBlend;
		g->ScatterBlock(g->blend.data(), x, y, z, tmp);
	}
	return new vxo_grid{ g };
}

vxo_surface* vxo_execute(const vxo_grid* g, const uint8_t* lut, const uint8_t* valid, int threads)
{
	if (threads > 0) omp_set_num_threads(threads);
	PSurface* s = new PSurface;
	memset(s->stats, 0, sizeof(s->stats));
	s->nextId = 0;
	Run run(*g->g, lut, valid, s);
	run.Execute(false, nullptr, nullptr, nullptr);
	return new vxo_surface{ s };
}

uint32_t vxo_execute_modify(const vxo_grid* g, const uint8_t* lut, const uint8_t* valid, int threads,
	vxo_surface* prev, const float min_corner[3], const float max_corner[3], uint32_t* modified_ids, uint32_t cap)
{
	if (threads > 0) omp_set_num_threads(threads);
	std::vector<u32> ids;
	Run run(*g->g, lut, valid, prev->s);
	run.Execute(true, min_corner, max_corner, &ids);
	for (size_t i = 0; i < ids.size() && i < cap; ++i) modified_ids[i] = ids[i];
	return (uint32_t)ids.size();
}

void vxo_surface_destroy(vxo_surface* s) { if (s) { delete s->s; delete s; } }
uint32_t vxo_surface_levels(const vxo_surface* s) { return (uint32_t)s->s->levels.size(); }
void vxo_surface_extents(const vxo_surface* s, float out[3]) { memcpy(out, s->s->extents, 12); }
uint32_t vxo_surface_blocks(const vxo_surface* s, uint32_t level) { return (uint32_t)s->s->levels[level].size(); }

void vxo_surface_level_totals(const vxo_surface* s, uint32_t level, uint64_t totals[4])
{
	totals[0] = totals[1] = totals[2] = totals[3] = 0;
	for (const OutBlock& b : s->s->levels[level]) {
		totals[0] += b.verts.size(); totals[1] += b.idx.size();
		for (int f = 0; f < 6; ++f) { totals[2] += b.tverts[f].size(); totals[3] += b.tidx[f].size(); }
	}
}

void vxo_surface_dump_level(const vxo_surface* s, uint32_t level, vxo_block_info* infos,
	vxo_vertex* verts, uint32_t* idx, vxo_vertex* tverts, uint32_t* tidx)
{
	size_t ov = 0, oi = 0, otv = 0, oti = 0, k = 0;
	for (const OutBlock& b : s->s->levels[level]) {
		vxo_block_info& info = infos[k++];
		info.id = b.id;
		info.n_verts = (u32)b.verts.size();
		info.n_idx = (u32)b.idx.size();
		if (verts && !b.verts.empty()) memcpy(verts + ov, b.verts.data(), b.verts.size() * 48);
		ov += b.verts.size();
		if (idx && !b.idx.empty()) memcpy(idx + oi, b.idx.data(), b.idx.size() * 4);
		oi += b.idx.size();
		for (int f = 0; f < 6; ++f) {
			info.n_tverts[f] = (u32)b.tverts[f].size();
			info.n_tidx[f] = (u32)b.tidx[f].size();
			if (tverts && !b.tverts[f].empty()) memcpy(tverts + otv, b.tverts[f].data(), b.tverts[f].size() * 48);
			otv += b.tverts[f].size();
			if (tidx && !b.tidx[f].empty()) memcpy(tidx + oti, b.tidx[f].data(), b.tidx[f].size() * 4);
			oti += b.tidx[f].size();
		}
		memcpy(info.min_corner, b.minc, 12);
		memcpy(info.max_corner, b.maxc, 12);
	}
}

void vxo_surface_stats(const vxo_surface* s, uint32_t stats[20]) { memcpy(stats, s->s->stats, 80); }

// TransVoxelImpl.cpp:196-220
uint32_t vxo_surface_cache_bytes(const vxo_surface* s)
{
	size_t total = 0;
	for (const auto& c : s->s->consistency) total += c.size();
	total >>= 3;
	for (const auto& lvl : s->s->lcache) for (const auto& blk : lvl) total += blk.size() * 2;
	return (uint32_t)total;
}

// TransVoxelImpl.cpp:222-235 (counts the six transition *vector objects*, 24 bytes each here, not their contents)
uint32_t vxo_surface_polygon_bytes(const vxo_surface* s)
{
	size_t r = 0;
	for (const auto& lvl : s->s->levels) for (const OutBlock& b : lvl) r += b.verts.size() * 48 + b.idx.size() * 4 + 6 * 24 + 6 * 24;
	return (uint32_t)r;
}

uint32_t vxo_log_errors(void) { return 0; }

} // extern "C"

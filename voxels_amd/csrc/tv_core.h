// tv_core.h — per-cell TransVoxel logic of the MI355X polygonizer, in closed (order-free) form.
//
// Everything here is a pure function of grid samples, Lengyel's tables and per-cell facts of the SAME block,
// so thousands of cells can be evaluated concurrently and still reproduce the reference's strictly sequential
// vertex numbering.  The functions are __host__ __device__: the HIP kernels (vx_hip.hip) call them from
// LDS-resident state, and tests/emu compiles the very same code with g++ to check the formulation on a CPU.
//
// Reference behaviour restated (all file:line into /root/reference/src/TransVoxelImpl.cpp):
//   sampling / clamping            :1140-1151, :1194-1201
//   normals                        :93-103, :1239-1246
//   LOD surface-shift correction   :1484-1509, :1666-1678
//   regular cell vertex rules      :1579-1718 (reuse, endpoint handling, material test)
//   boundary flags + secondary pos :593-645, :683-708, :1473-1482, :1729-1738
//   transition cells               :1754-2131
//   material vote                  :753-838
//   result packing                 :1248-1264, :1300-1369
//
// Why the closed form is exact (derivation in DESIGN.md §3):
//   * a reuse slot of a cell is only ever written by a vertex that cell creates unconditionally (owned edge,
//     direction 8, or the corner-7 endpoint), so "slot valid" and "ordinal of the slot's vertex" are functions
//     of that cell alone;
//   * every vertex a cell creates carries that cell's material id, so the reuse-by-material test compares the
//     two cells' material ids;
//   * the reference's data-dependent reuseValidityMask equals OR-scans of the block's non-trivial bitmap.
#pragma once

#include <stdint.h>
#include <stddef.h>
#include <string.h>

#if defined(__HIPCC__)
#include <hip/hip_runtime.h>
#define TV_HD __host__ __device__ __forceinline__
#else
#include <math.h>
#define TV_HD inline
#endif

// keeps the compiler's scheduler from moving instructions across (device code; nothing on the host)
#if defined(__HIP_DEVICE_COMPILE__)
#define TV_SCHED_FENCE() __builtin_amdgcn_sched_barrier(0)
#else
#define TV_SCHED_FENCE() do { } while (0)
#endif
// a store of data the run does not read again (mesh vertices, indices): streaming on the device, plain on the host
#if defined(__HIP_DEVICE_COMPILE__)
#define TV_STREAM_STORE(ptr, value) __builtin_nontemporal_store((value), (ptr))
#else
#define TV_STREAM_STORE(ptr, value) (*(ptr) = (value))
#endif

namespace tv {

typedef uint8_t u8;
typedef uint16_t u16;
typedef uint32_t u32;
typedef int8_t i8;

enum : u32 { INVALID_INDEX = 0xFFFFFFFFu };
enum : u32 { EMPTY_MATERIAL = 255u };   // VoxelGrid.h:16 — material id 255 is reserved by the reference
enum : u16 { EMPTY_MATINFO = 0x00FFu }; // id = 255, blend = 0 (TransVoxelImpl.cpp:424)

// 48-byte output vertex, bit-compatible with Voxels::PolygonVertex (include/Polygonizer.h:14-48)
struct alignas(16) PolyVertex {
	float pos[3];
	float sec[3];
	u32 secW;     // adjacency bit mask, raw integer bits (never touched by float ops: values 1..63 are denormals)
	float nrm[3];
	u8 tex[8];    // Reserved, Blend, Uxz, Txz, Uny, Upy, Tny, Tpy
};

// Offsets of the table image (tv_tables.inc) when staged as one byte array (LDS on the GPU)
enum : u32 {
	TAB_REG_CLASS = 0,       // 256 B
	TAB_REG_CELL = 256,      // 16 x 16 B
	TAB_TR_CLASS = 512,      // 512 B
	TAB_TR_CORNER = 1024,    // 16 B
	TAB_TR_CELL = 1040,      // 56 x 40 B = 2240 -> 3280
	// The vertex data of a case is a function of the edge alone (12 distinct words in the regular table, 16 in the
	// transition table), so the per-case rows hold 4-bit indices into two 16-entry word tables: 4.5 KB instead of 18 KB.
	TAB_REG_EDGE = 3280,     // 16 x u16: (reuseDir<<12)|(reuseSlot<<8)|(corner0<<4)|corner1 of each regular edge
	TAB_TR_EDGE = 3312,      // 16 x u16: the same for the transition cell edges
	TAB_REG_VERT = 3344,     // 256 x 12 nibbles = 1536 -> 4880
	TAB_TR_VERT = 4880,      // 512 x 12 nibbles = 3072 -> 7952
	TAB_REG_OWN = 7952,      // 256 B: reuse slots (bits 1..3) a regular case stores when none of its samples is 0
	TAB_TR_OWN = 8208,       // 512 x u16: the same for the transition cases (slots 0..9)
	TAB_BYTES = 9232
};

struct Tables {
	const u8* regClassP;   // 256
	const u8* regCellP;    // 16 x 16
	const u8* regVertP;    // 256 x 6: twelve edge indices per case
	const u16* regEdgeP;   // 16
	const u8* regOwnP;     // 256
	const u8* trClassP;    // 512
	const u8* trCornerP;   // 16
	const u8* trCellP;     // 56 x 40
	const u8* trVertP;     // 512 x 6
	const u16* trEdgeP;    // 16
	const u16* trOwnP;     // 512
	TV_HD u32 regClass(u32 code) const { return regClassP[code]; }
	TV_HD const u8* regCell(u32 cls) const { return regCellP + cls * 16; }
	TV_HD u32 regOwn(u32 code) const { return regOwnP[code]; }
	TV_HD u32 regVert(u32 code, u32 i) const { return regEdgeP[(regVertP[code * 6 + (i >> 1)] >> ((i & 1u) * 4u)) & 15u]; }
	TV_HD u32 trClass(u32 code) const { return trClassP[code]; }
	TV_HD const u8* trCell(u32 cls) const { return trCellP + cls * 40; }
	TV_HD u32 trCorner(u32 c) const { return trCornerP[c]; }
	TV_HD u32 trOwn(u32 code) const { return trOwnP[code]; }
	TV_HD u32 trVert(u32 code, u32 i) const { return trEdgeP[(trVertP[code * 6 + (i >> 1)] >> ((i & 1u) * 4u)) & 15u]; }
};

// all tables from one TAB_BYTES image
TV_HD Tables tables_from_image(const u8* base)
{
	Tables T;
	T.regClassP = base + TAB_REG_CLASS; T.regCellP = base + TAB_REG_CELL; T.regVertP = base + TAB_REG_VERT;
	T.regEdgeP = (const u16*)(base + TAB_REG_EDGE); T.regOwnP = base + TAB_REG_OWN;
	T.trClassP = base + TAB_TR_CLASS; T.trCornerP = base + TAB_TR_CORNER; T.trCellP = base + TAB_TR_CELL;
	T.trVertP = base + TAB_TR_VERT; T.trEdgeP = (const u16*)(base + TAB_TR_EDGE); T.trOwnP = (const u16*)(base + TAB_TR_OWN);
	return T;
}

// Dense voxel field resident in HBM: x fastest, then y, then z.  A rank of a multi-GPU run holds a slab of the global
// grid plus halo: the z-planes [zOrigin, ...) and, within every plane, the rows [yOrigin, yOrigin + pitchY) (a whole
// grid has zOrigin = yOrigin = 0 and pitchY = n; slabs are cut along z OR along y).  Coordinates are always global
// and are clamped to the GLOBAL extent [0, n-1] exactly like the reference clamps every fetch.
//
// Brick mirrors (GPU backend): the same three fields a second time, block-major — the 4096 bytes of a 16^3 block are
// contiguous, inside it 128-byte tiles of 16 x 4 x 2 voxels (brick_local).  The dense fields are the interchange layout
// (uploads, attached slabs, the grid file codec, edits, halo messages, the streaming classify pass); everything that
// gathers around the surface — block neighbourhoods, boundary planes, the samples around a vertex — reads the bricks:
// a dense row puts 16 useful bytes into every 128-byte line it touches and a plane x = const one, a tile 128 / 8.  The
// mirrors are brought up to date where the grid changes (vx_host.inl: rebrick), never by a polygonization.
struct GridView {
	const i8* dist;
	const u8* mat;
	const u8* blend;
	int n;          // global grid edge
	int zOrigin;    // global z of plane 0 of dist[]
	int zOriginMat; // global z of plane 0 of mat[] / blend[]
	int yOrigin, yOriginMat; // global y of row 0 of every plane
	int pitchY, pitchYMat;   // rows per plane
	const i8* bDist;         // brick mirrors (nullptr: none — the CPU emulation reads the dense fields)
	const u8* bMat;
	const u8* bBlend;
	int bYb0, bZb0;          // first resident block row / block plane of the mirrors (halo layers included)
	int bRowsY;              // resident block rows per block plane
};

TV_HD int clampi(int v, int lo, int hi) { return v < lo ? lo : (v > hi ? hi : v); }

// element offsets of global voxel (x,y,z), already clamped to the grid
TV_HD size_t dist_offset(const GridView& g, int x, int y, int z) { return ((size_t)(z - g.zOrigin) * g.pitchY + (y - g.yOrigin)) * g.n + x; }
TV_HD size_t mat_offset(const GridView& g, int x, int y, int z) { return ((size_t)(z - g.zOriginMat) * g.pitchYMat + (y - g.yOriginMat)) * g.n + x; }

// brick mirrors: byte offset of voxel (x,y,z) of a block (0..15 each) inside its brick: tile (y >> 2, z >> 1) of 16 x 4 x 2
// voxels = one 128-byte line, tiles ordered y fastest; a voxel row (16 bytes) stays contiguous
enum { BRICK_BYTES = 4096 };
TV_HD u32 brick_local(u32 x, u32 y, u32 z) { return ((z >> 1) << 9) | ((y >> 2) << 7) | ((z & 1u) << 6) | ((y & 3u) << 4) | x; }
TV_HD size_t brick_base(const GridView& g, int bx, int by, int bz) { return (((size_t)(bz - g.bZb0) * (size_t)g.bRowsY + (size_t)(by - g.bYb0)) * (size_t)(g.n >> 4) + (size_t)bx) * BRICK_BYTES; }
TV_HD size_t brick_offset(const GridView& g, int x, int y, int z)
{
#if defined(__HIP_DEVICE_COMPILE__)
	// the brick's index with 24-bit multiplies (block planes and rows per plane <= 129, blocks per row <= 128: every
	// product stays below 2^22), widened once: the 64-bit multiply-adds of brick_base run at a quarter of the rate
	const u32 brick = __umul24(__umul24((u32)((z >> 4) - g.bZb0), (u32)g.bRowsY) + (u32)((y >> 4) - g.bYb0), (u32)g.n >> 4) + (u32)(x >> 4);
	return ((size_t)brick << 12) | brick_local((u32)x & 15u, (u32)y & 15u, (u32)z & 15u);
#else
	return brick_base(g, x >> 4, y >> 4, z >> 4) + brick_local((u32)x & 15u, (u32)y & 15u, (u32)z & 15u);
#endif
}

TV_HD int dist_at(const GridView& g, int x, int y, int z)
{
	x = clampi(x, 0, g.n - 1); y = clampi(y, 0, g.n - 1); z = clampi(z, 0, g.n - 1);
#if defined(__HIP_DEVICE_COMPILE__)
	return g.bDist[brick_offset(g, x, y, z)];
#else
	return g.dist[dist_offset(g, x, y, z)];
#endif
}

// material info packed as id | blend << 8
TV_HD u32 mat_at(const GridView& g, int x, int y, int z)
{
	x = clampi(x, 0, g.n - 1); y = clampi(y, 0, g.n - 1); z = clampi(z, 0, g.n - 1);
#if defined(__HIP_DEVICE_COMPILE__)
	const size_t i = brick_offset(g, x, y, z);
	return (u32)g.bMat[i] | ((u32)g.bBlend[i] << 8);
#else
	const size_t i = mat_offset(g, x, y, z);
	return (u32)g.mat[i] | ((u32)g.blend[i] << 8);
#endif
}

TV_HD void normalize_fix_zero(float v[3])
{
	const float len2 = (v[0] * v[0] + v[1] * v[1]) + v[2] * v[2];
#if defined(__HIP_DEVICE_COMPILE__)
	// Correctly rounded square root = the hardware's 1-ulp v_sqrt_f32 nudged by the sign of the two residuals, which is
	// what the compiler emits for sqrtf minus its input scaling and class test: len2 is 0 or in [1e-16, 2e5] here,
	// never a denormal, infinity or NaN (0 passes through: the neighbours of 0 give NaN residuals, both tests fail).
	float len = __builtin_amdgcn_sqrtf(len2);
	{
		const float down = __builtin_bit_cast(float, __builtin_bit_cast(int, len) - 1), up = __builtin_bit_cast(float, __builtin_bit_cast(int, len) + 1);
		const float rDown = __builtin_fmaf(-down, len, len2), rUp = __builtin_fmaf(-up, len, len2);
		len = (rDown <= 0.f) ? down : len;
		len = (rUp > 0.f) ? up : len;
	}
#else
	const float len = sqrtf(len2);
#endif
	if (len <= 1.1920929e-07f) { v[0] = v[1] = v[2] = 0.f; return; }
#if defined(__HIP_DEVICE_COMPILE__)
	// Three IEEE divisions by the same denominator: the compiler's fp32 division (reciprocal, two Newton steps on the
	// quotient, final fused correction) with the reciprocal refined once instead of three times.  The operands are far
	// from the ranges where v_div_scale / v_div_fixup would intervene (len in [1e-7, 256], |v| <= 128 and never a
	// denormal), so every quotient is the correctly rounded one, bit for bit what v / len gives on the host.
	const float y0 = __builtin_amdgcn_rcpf(len);
	const float y = __builtin_fmaf(__builtin_fmaf(-len, y0, 1.0f), y0, y0);
#pragma unroll
	for (int i = 0; i < 3; ++i) {
		const float n = v[i];
		const float q0 = n * y;
		const float q1 = __builtin_fmaf(__builtin_fmaf(-len, q0, n), y, q0);
		v[i] = __builtin_fmaf(__builtin_fmaf(-len, q1, n), y, q1);
	}
#else
	v[0] = v[0] / len; v[1] = v[1] / len; v[2] = v[2] / len;
#endif
}

// normalize_fix_zero for a vector of INTEGER components in [-255, 255] (a central difference of int8 samples, scaled or
// not by a power of two): same bits, fewer operations.  On the device the length comes from rsq + one Newton step and
// every quotient gets one correction instead of two — for these 2^24 inputs (up to signs) both are exact, checked
// exhaustively on the hardware itself (vx_selftest, results[4], [6], [11]); arbitrary vectors (an interpolated normal)
// keep normalize_fix_zero.
TV_HD void normalize_gradient(float v[3])
{
#if defined(__HIP_DEVICE_COMPILE__)
	const float len2 = (v[0] * v[0] + v[1] * v[1]) + v[2] * v[2];
	if (!(len2 > 0.f)) { v[0] = v[1] = v[2] = 0.f; return; } // integer components: the length is 0 or at least 1
	const float r = __builtin_amdgcn_rsqf(len2), s = len2 * r, h = 0.5f * r;
	const float len = __builtin_fmaf(__builtin_fmaf(-s, s, len2), h, s);
	const float y0 = __builtin_amdgcn_rcpf(len);
	const float y = __builtin_fmaf(__builtin_fmaf(-len, y0, 1.0f), y0, y0);
#pragma unroll
	for (int i = 0; i < 3; ++i) {
		const float n = v[i], q0 = n * y;
		v[i] = __builtin_fmaf(__builtin_fmaf(-len, q0, n), y, q0);
	}
#else
	normalize_fix_zero(v);
#endif
}

// distance sampler reading the dense field in HBM (global coordinates, clamped like every reference fetch)
struct GlobalDist {
	const GridView* g;
	TV_HD int operator()(int x, int y, int z) const { return dist_at(*g, x, y, z); }
};

// level-0 central differences; components come out ordered (x, z, y) — the output is Y-up
template <typename D>
TV_HD void normal_at(const D& d, int x, int y, int z, float out[3])
{
	out[0] = (float)(d(x + 1, y, z) - d(x - 1, y, z)) * 0.5f;
	out[1] = (float)(d(x, y, z + 1) - d(x, y, z - 1)) * 0.5f;
	out[2] = (float)(d(x, y + 1, z) - d(x, y - 1, z)) * 0.5f;
	normalize_fix_zero(out);
}

// `level` bisection steps toward the level-0 edge that holds the crossing.  val0/val1 are the samples at P0/P1
// (the reference re-fetches them every step, TransVoxelImpl.cpp:1495-1497; they are carried here, so the chain
// costs one fetch per level).
template <typename D>
TV_HD void lod_chain(const D& d, int level, int P0[3], int P1[3], int& val0, int& val1)
{
	for (int lev = level; lev > 0; --lev) {
		const int mx = P0[0] + (P1[0] - P0[0]) / 2, my = P0[1] + (P1[1] - P0[1]) / 2, mz = P0[2] + (P1[2] - P0[2]) / 2;
		const int midV = d(mx, my, mz);
		if (val0 * midV <= 0) { P1[0] = mx; P1[1] = my; P1[2] = mz; val1 = midV; }
		else { P0[0] = mx; P0[1] = my; P0[2] = mz; val0 = midV; }
	}
}

// Everything a vertex reads around its two end points — the stencils of both central differences and both materials —
// requested together, before anything is computed with them: as separate reads followed by their arithmetic (normal_at
// twice, then the materials) the device code waits for every pair of samples in turn, nine memory round trips per
// vertex instead of one.
struct EndpointSamples {
	int a[6], b[6]; // x+1, x-1, z+1, z-1, y+1, y-1 around P0 / P1 (normal_at's order)
	u32 M0, M1;
};

template <typename D, typename MF>
TV_HD void gather_endpoints(const D& d, const MF& mats, const int P0[3], const int P1[3], EndpointSamples& e)
{
	e.a[0] = d(P0[0] + 1, P0[1], P0[2]); e.a[1] = d(P0[0] - 1, P0[1], P0[2]);
	e.a[2] = d(P0[0], P0[1], P0[2] + 1); e.a[3] = d(P0[0], P0[1], P0[2] - 1);
	e.a[4] = d(P0[0], P0[1] + 1, P0[2]); e.a[5] = d(P0[0], P0[1] - 1, P0[2]);
	e.b[0] = d(P1[0] + 1, P1[1], P1[2]); e.b[1] = d(P1[0] - 1, P1[1], P1[2]);
	e.b[2] = d(P1[0], P1[1], P1[2] + 1); e.b[3] = d(P1[0], P1[1], P1[2] - 1);
	e.b[4] = d(P1[0], P1[1] + 1, P1[2]); e.b[5] = d(P1[0], P1[1] - 1, P1[2]);
	e.M0 = mats(0, P0); e.M1 = mats(1, P1);
}

// normal_at() from samples already fetched
TV_HD void normal_from(const int s[6], float out[3])
{
	out[0] = (float)(s[0] - s[1]) * 0.5f;
	out[1] = (float)(s[2] - s[3]) * 0.5f;
	out[2] = (float)(s[4] - s[5]) * 0.5f;
	normalize_fix_zero(out);
}

// the same for the transition vertices: the halved differences of int8 samples are exact in fp32 and normalize_gradient
// gives normalize_fix_zero's bits for them (its comment; vx_selftest)
TV_HD void gradient_from(const int s[6], float out[3])
{
	out[0] = (float)(s[0] - s[1]) * 0.5f;
	out[1] = (float)(s[2] - s[3]) * 0.5f;
	out[2] = (float)(s[4] - s[5]) * 0.5f;
	normalize_gradient(out);
}

// (v1 << 8) / (v1 - v0) with C truncation.  Evaluated as a correctly rounded fp32 division: for |v| <= 128 the
// quotient is either an integer or at least 1/255 away from one while fp32 resolves 1/1024 there, so truncating
// the rounded quotient is exact (checked exhaustively over all 65280 int8 pairs in tests/test_core_math.py).
TV_HD int edge_t(int v0, int v1) { return (int)((float)(v1 * 256) / (float)(v1 - v0)); }

// edge_t for a crossed edge: v0 * v1 <= 0, v0 != v1 (the quotient lies in [0, 256]).  The device form multiplies by the
// hardware reciprocal (1 ulp) instead of dividing: the product is off by less than 5e-5, an exact quotient is an integer
// or at least 1/255 away from one, and a bias of 2^-10 puts every case on the right side of the truncation
// (checked for all pairs on the device by tests/test_gpu_parity.py::test_hip_edge_t_crossing_exhaustive).
TV_HD int edge_t_crossing(int v0, int v1)
{
#if defined(__HIP_DEVICE_COMPILE__)
	return (int)__builtin_fmaf((float)(v1 * 256), __builtin_amdgcn_rcpf((float)(v1 - v0)), 0.0009765625f);
#else
	return edge_t(v0, v1);
#endif
}

// Where edge_t lands without dividing, for the decisions that only ask "is the vertex on a corner, and which":
// 0 <=> edge_t == 0 (v1 == 0: |v1 * 256| >= |v1 - v0| otherwise), 256 <=> edge_t == 256 (v0 == 0, the samples
// of a crossed edge never share a strict sign), 1 = strictly inside the edge.  Same exhaustive test as edge_t.
TV_HD int edge_end(int v0, int v1) { return v1 == 0 ? 0 : (v0 == 0 ? 256 : 1); }

// the same from a cell's "corner sample == 0" mask
TV_HD int edge_end_bits(u32 zeroMask, int c0, int c1) { return ((zeroMask >> c1) & 1u) ? 0 : (((zeroMask >> c0) & 1u) ? 256 : 1); }

TV_HD u32 reg_zero_mask(const i8 V[8])
{
	u32 m = 0;
#pragma unroll
	for (int i = 0; i < 8; ++i) m |= (V[i] == 0 ? 1u : 0u) << i;
	return m;
}

TV_HD u32 lerp_blend(int t, int u, u32 b0, u32 b1)
{
	const float v = ((float)t * (float)b0 + (float)u * (float)b1) / 256.f;
	return (u32)(int)v & 0xFFu;
}

// 6-bit boundary mask (bit order ZPos,YPos,XPos,ZNeg,YNeg,XNeg) of the edge c0-c1 (corner when c0 == c1) of the
// cell at local coordinates (lx,ly,lz); zero at level 0 and for interior cells.
TV_HD u32 boundary_mask(int lx, int ly, int lz, int mult, int c0, int c1)
{
	if (mult == 1) return 0;
	const int both = c0 & c1, none = ~(c0 | c1);
	u32 r = 0;
	if (lx == 0 && (none & 1)) r |= 1u << 5;
	if (lx == 15 && (both & 1)) r |= 1u << 2;
	if (ly == 0 && (none & 2)) r |= 1u << 4;
	if (ly == 15 && (both & 2)) r |= 1u << 1;
	if (lz == 0 && (none & 4)) r |= 1u << 3;
	if (lz == 15 && (both & 4)) r |= 1u << 0;
	return r;
}

// sum over flagged faces of 0.25 * cell size along the face's inward normal (internal Z-up axes)
TV_HD void transition_delta(u32 faces, int mult, float out[3])
{
	const float q = (float)mult * 0.25f;
	out[0] = out[1] = out[2] = 0.f;
	if (faces & 1u) out[2] += -q;
	if (faces & 2u) out[1] += -q;
	if (faces & 4u) out[0] += -q;
	if (faces & 8u) out[2] += q;
	if (faces & 16u) out[1] += q;
	if (faces & 32u) out[0] += q;
}

// Vertex before packing: coordinates are x256 (internal Z-up axes)
struct RawVertex {
	float p[3];
	float s[3];
	u32 flags;
	float n[3];
	u32 mat; // id | blend << 8
};

// PushBlocksToResult packing: x(1/256), y<->z swap, flag swizzle, texture ids from the material LUT
// lut = 256 x 8 bytes: {Ids0[0..2], Ids1[0..2], valid, pad}
TV_HD unsigned long long lut_row(const u8* lut, u32 materialId)
{
	// one aligned 8-byte fetch of the LUT row {Ids0[3], Ids1[3], valid, pad}
	unsigned long long row;
	memcpy(&row, lut + (materialId & 0xFFu) * 8, 8);
	return row;
}

#if defined(__HIPCC__)
// a packed vertex in registers: the twelve dwords of a PolyVertex (three 16-byte pieces); device code only
struct VertexRegs { u32 w[12]; };
__device__ __forceinline__ void pack_vertex_regs(const RawVertex& r, unsigned long long row, VertexRegs& o)
{
#if defined(__HIP_DEVICE_COMPILE__)
	const float k = 1.f / 256.f;
	u32 f = r.flags;
	if (f) f = (f >> 3) | ((f & 7u) << 3);
	// texture bytes {0, blend, Ids1[1], Ids0[1] | Ids1[2], Ids1[0], Ids0[2], Ids0[0]} picked out of the row with two
	// byte permutes (selector 0..3 = low dword, 4..7 = high dword, 0x0C = constant 0)
	const u32 lo = (u32)row, hi = (u32)(row >> 32);
	const bool ok = ((hi >> 16) & 0xFFu) != 0;
	u32 t0 = __builtin_amdgcn_perm(hi, lo, 0x01040C0Cu) | (((r.mat >> 8) & 0xFFu) << 8);
	u32 t1 = __builtin_amdgcn_perm(hi, lo, 0x00020305u);
	if (!ok) { t0 = 0; t1 = 0; }
	o.w[0] = __float_as_uint(r.p[0] * k); o.w[1] = __float_as_uint(r.p[2] * k); o.w[2] = __float_as_uint(r.p[1] * k); o.w[3] = __float_as_uint(r.s[0] * k);
	o.w[4] = __float_as_uint(r.s[2] * k); o.w[5] = __float_as_uint(r.s[1] * k); o.w[6] = f; o.w[7] = __float_as_uint(r.n[0]);
	o.w[8] = __float_as_uint(r.n[1]); o.w[9] = __float_as_uint(r.n[2]); o.w[10] = t0; o.w[11] = t1;
#endif
}
#endif

TV_HD void pack_vertex_row(const RawVertex& r, unsigned long long row, PolyVertex* out)
{
#if defined(__HIP_DEVICE_COMPILE__)
	// the vertex leaves as three aligned 16-byte stores (a member-wise copy is split at the member boundaries: 12 + 16 + 12 + 8 bytes)
	VertexRegs o;
	pack_vertex_regs(r, row, o);
	typedef u32 __attribute__((ext_vector_type(4))) v4u;
	v4u* dst = (v4u*)out;
	const v4u a = { o.w[0], o.w[1], o.w[2], o.w[3] };
	const v4u b = { o.w[4], o.w[5], o.w[6], o.w[7] };
	const v4u c = { o.w[8], o.w[9], o.w[10], o.w[11] };
	// The meshes are written once and not read again by the run: streaming (non-temporal) stores keep them from
	// displacing the voxel lines the neighbouring blocks are about to read.  Measured at 1024^3: level-0 pass 0.196 ->
	// 0.181 ms, levels >= 1 0.128 -> 0.115, the whole step 0.498 -> 0.463.
	TV_STREAM_STORE(&dst[0], a);
	TV_STREAM_STORE(&dst[1], b);
	TV_STREAM_STORE(&dst[2], c);
#else
	const float k = 1.f / 256.f;
	u32 f = r.flags;
	if (f) f = (f >> 3) | ((f & 7u) << 3);
	PolyVertex o;
	o.pos[0] = r.p[0] * k; o.pos[1] = r.p[2] * k; o.pos[2] = r.p[1] * k;
	o.sec[0] = r.s[0] * k; o.sec[1] = r.s[2] * k; o.sec[2] = r.s[1] * k;
	o.secW = f;
	o.nrm[0] = r.n[0]; o.nrm[1] = r.n[1]; o.nrm[2] = r.n[2];
	u8 e[8];
#pragma unroll
	for (int i = 0; i < 8; ++i) e[i] = (u8)(row >> (8 * i));
	const bool ok = e[6] != 0;
	o.tex[0] = 0;
	o.tex[1] = ok ? (u8)(r.mat >> 8) : 0;
	o.tex[2] = ok ? e[4] : 0; // Uxz = Ids1[1]
	o.tex[3] = ok ? e[1] : 0; // Txz = Ids0[1]
	o.tex[4] = ok ? e[5] : 0; // Uny = Ids1[2]
	o.tex[5] = ok ? e[3] : 0; // Upy = Ids1[0]
	o.tex[6] = ok ? e[2] : 0; // Tny = Ids0[2]
	o.tex[7] = ok ? e[0] : 0; // Tpy = Ids0[0]
	*out = o;
#endif
}

// Where a packed vertex goes: to its record in memory (three 16-byte stores of the lane)
struct VertexToMemory {
	PolyVertex* out;
	TV_HD void operator()(const RawVertex& r, unsigned long long row) const { pack_vertex_row(r, row, out); }
};

TV_HD void pack_vertex(const RawVertex& r, const u8* lut, PolyVertex* out) { pack_vertex_row(r, lut_row(lut, r.mat), out); }

// degenerate-triangle test of PushBlocksToResult on x256 positions, plain fp32 (no fused multiply-add)
TV_HD bool triangle_degenerate(const float* v0, const float* v1, const float* v2)
{
	const float ax = v1[0] - v0[0], ay = v1[1] - v0[1], az = v1[2] - v0[2];
	const float bx = v2[0] - v0[0], by = v2[1] - v0[1], bz = v2[2] - v0[2];
	const float cx = ay * bz - by * az, cy = az * bx - bz * ax, cz = ax * by - bx * ay;
	const float len2 = (cx * cx + cy * cy) + cz * cz;
	return !(len2 >= 1.1920929e-07f);
}

// ---------------------------------------------------------------------------------------------------------
// Regular cells
// ---------------------------------------------------------------------------------------------------------
TV_HD u32 reg_case_code(const i8 V[8])
{
	u32 c = 0;
#pragma unroll
	for (int i = 0; i < 8; ++i) c |= ((u32)(V[i] >> 7) & 1u) << i;
	return c;
}

// bit s set <=> this cell itself creates and stores a vertex in reuse slot s (s = 0: the corner-7 vertex).
// Without a zero sample no vertex sits on a corner and the mask is a property of the case (table regOwn).
TV_HD u32 reg_slot_valid(const Tables& T, u32 zeroMask, u32 code)
{
	if (!zeroMask) return T.regOwn(code);
	const u32 nv = T.regCell(T.regClass(code))[0] >> 4;
	u32 m = 0;
	for (u32 vi = 0; vi < nv; ++vi) {
		const u32 w = T.regVert(code, vi);
		const int v0 = (w >> 4) & 15, v1 = w & 15;
		const int t = edge_end_bits(zeroMask, v0, v1);
		if ((t & 0xFF) == 0) { if (t == 0 && v1 == 7) m |= 1u; }
		else if ((w >> 12) == 8u) m |= 1u << ((w >> 8) & 15);
	}
	return m;
}

enum { RK_NEW_EDGE = 0, RK_NEW_CORNER = 1, RK_REUSE = 2, RK_INVALID = 3 };
enum { NO_SLOT = 0xFF };

struct Resolution {
	u8 kind;
	u8 a;      // NEW_EDGE: v0 | NEW_CORNER: corner | REUSE: direction
	u8 b;      // NEW_EDGE: v1 |                     | REUSE: slot
	u8 store;  // slot this new vertex is stored in, or NO_SLOT
	int t;     // edge_end() of the cell's own corner values
};

// How vertex `w` (table word) of the non-trivial cell (cx,cy,cz) gets its index.
//   mask3  : bit0 = a non-trivial cell exists earlier in this row, bit1 = in an earlier row of this slice,
//            bit2 = in an earlier slice (the reference's reuseValidityMask)
//   nb(dx,dy,dz, slot, &valid, &mat): reuse slot of the neighbour cell at (cx-dx, cy-dy, cz-dz)
//   zeroMask: bit i = corner sample i is exactly 0 (all the resolution needs to know about the values)
template <typename NB>
TV_HD Resolution reg_resolve(u32 zeroMask, u32 w, u32 mask3, u32 myMatId, const NB& nb)
{
	Resolution r;
	const int v0 = (w >> 4) & 15, v1 = w & 15;
	u32 dir = w >> 12, slot = (w >> 8) & 15;
	const int t = edge_end_bits(zeroMask, v0, v1);
	const bool endpoint = (t & 0xFF) == 0;
	bool check = true;
	if (endpoint) {
		if (t == 0 && v1 == 7) check = false;
		else dir = (u32)((t == 0) ? v1 : v0) ^ 7u;
		slot = 0;
	}
	r.t = t;
	if (check && (dir & mask3) == dir) {
		bool valid; u32 nmat;
		nb((int)(dir & 1), (int)((dir >> 1) & 1), (int)((dir >> 2) & 1), slot, valid, nmat);
		if (!valid) {
			// empty slot counts as "same material": endpoint -> a fresh, unstored vertex at v0 (sic), else INVALID
			r.kind = endpoint ? RK_NEW_CORNER : RK_INVALID; r.a = (u8)v0; r.b = 0; r.store = NO_SLOT;
			return r;
		}
		if (nmat == myMatId) { r.kind = RK_REUSE; r.a = (u8)dir; r.b = (u8)slot; r.store = NO_SLOT; return r; }
	}
	if (endpoint) {
		r.kind = RK_NEW_CORNER; r.a = (u8)((t == 0) ? v1 : v0); r.b = 0;
		r.store = (t == 0 && v1 == 7) ? 0 : NO_SLOT;
	} else {
		r.kind = RK_NEW_EDGE; r.a = (u8)v0; r.b = (u8)v1;
		r.store = ((w >> 12) == 8u) ? (u8)slot : (u8)NO_SLOT;
	}
	return r;
}

// (direction, slot) a reused vertex `w` comes from, as reg_resolve derives them
TV_HD void reg_reuse_source(u32 zeroMask, u32 w, u32& dir, u32& slot)
{
	const int v0 = (w >> 4) & 15, v1 = w & 15;
	const int t = edge_end_bits(zeroMask, v0, v1);
	dir = w >> 12; slot = (w >> 8) & 15;
	if ((t & 0xFF) == 0) { dir = (u32)((t == 0) ? v1 : v0) ^ 7u; slot = 0; }
}

struct CellGeom {
	int base[3];  // global voxel coordinates of corner 0
	int local[3]; // cell coordinates inside the block
	int mult;     // cell size in voxels = 1 << level
	int level;
};

TV_HD void corner_pos(const CellGeom& c, int corner, int P[3])
{
	P[0] = c.base[0] + ((corner & 1) ? c.mult : 0);
	P[1] = c.base[1] + ((corner & 2) ? c.mult : 0);
	P[2] = c.base[2] + ((corner & 4) ? c.mult : 0);
}

TV_HD void finish_secondary(RawVertex& o, int mult)
{
	if ((int)o.flags > 0) {
		float d[3];
		transition_delta(o.flags, mult, d);
		o.s[0] = o.p[0] + d[0] * 256.f; o.s[1] = o.p[1] + d[1] * 256.f; o.s[2] = o.p[2] + d[2] * 256.f;
	} else {
		o.s[0] = o.p[0]; o.s[1] = o.p[1]; o.s[2] = o.p[2];
	}
}

// x256 position of an edge vertex (after the LOD chain); also returns the final t and endpoints
template <typename D>
TV_HD bool reg_edge_position(const D& d, const CellGeom& c, int v0, int v1, int t0, int val0, int val1,
                             int P0[3], int P1[3], int& t, float pos[3])
{
	corner_pos(c, v0, P0);
	corner_pos(c, v1, P1);
	t = t0;
	int p0 = val0, p1 = val1;
	if (c.level > 0) {
		lod_chain(d, c.level, P0, P1, p0, p1);
		t = (p0 != p1) ? edge_t(p0, p1) : 0;
	}
	const bool interior = p0 * p1 < 0; // samples of strictly opposite sign: 0 < t < 256, vertex strictly inside its edge
	const float ft = (float)t, fu = (float)(256 - t);
	pos[0] = ft * (float)P0[0] + fu * (float)P1[0];
	pos[1] = ft * (float)P0[1] + fu * (float)P1[1];
	pos[2] = ft * (float)P0[2] + fu * (float)P1[2];
	return interior;
}

// material source reading the grid in HBM at the (possibly LOD-shifted) end points
struct GridMaterials {
	const GridView* g;
	TV_HD u32 operator()(int, const int P[3]) const { return mat_at(*g, P[0], P[1], P[2]); }
};

// An edge vertex once its level-0 edge P0-P1 with samples p0 / p1 is known (the cell's own edge at level 0, the end of
// the LOD chain above): position, normals, material, adjacency.  `mats(which, P)` = id | blend << 8 at end point `which`
// (0/1) located at P: the grid, or values fetched ahead.
template <typename D, typename MF>
TV_HD bool reg_edge_finish(const D& d, const MF& mats, const CellGeom& c, int v0, int v1, const int P0[3], const int P1[3], int p0, int p1, int t, u32 cellMat, RawVertex& o)
{
	EndpointSamples e;
	gather_endpoints(d, mats, P0, P1, e);
	const bool interior = p0 * p1 < 0; // samples of strictly opposite sign: 0 < t < 256, vertex strictly inside its edge
	const int u = 256 - t;
	const float ft = (float)t, fu = (float)u;
	o.p[0] = ft * (float)P0[0] + fu * (float)P1[0];
	o.p[1] = ft * (float)P0[1] + fu * (float)P1[1];
	o.p[2] = ft * (float)P0[2] + fu * (float)P1[2];
	float N0[3], N1[3];
	normal_from(e.a, N0);
	normal_from(e.b, N1);
	const u32 M0 = e.M0, M1 = e.M1;
	o.flags = boundary_mask(c.local[0], c.local[1], c.local[2], c.mult, v0, v1);
	if ((M0 & 0xFF) == (M1 & 0xFF) && (M0 & 0xFF) == (cellMat & 0xFF)) o.mat = (M0 & 0xFF) | (lerp_blend(t, u, M0 >> 8, M1 >> 8) << 8);
	else o.mat = cellMat;
	const float wt = ft / 256.f, wu = fu / 256.f;
	o.n[0] = N0[0] * wt + N1[0] * wu; o.n[1] = N0[1] * wt + N1[1] * wu; o.n[2] = N0[2] * wt + N1[2] * wu;
	normalize_fix_zero(o.n);
	finish_secondary(o, c.mult);
	return interior;
}

template <typename D, typename MF>
TV_HD bool reg_edge_vertex(const D& d, const MF& mats, const CellGeom& c, int v0, int v1, int t0, int val0, int val1, u32 cellMat, RawVertex& o)
{
	int P0[3], P1[3];
	corner_pos(c, v0, P0);
	corner_pos(c, v1, P1);
	int t = t0, p0 = val0, p1 = val1;
	if (c.level > 0) {
		lod_chain(d, c.level, P0, P1, p0, p1);
		t = (p0 != p1) ? edge_t(p0, p1) : 0;
	}
	return reg_edge_finish(d, mats, c, v0, v1, P0, P1, p0, p1, t, cellMat, o);
}

template <typename D>
TV_HD void reg_vertex_position(const D& d, const CellGeom& c, const i8 V[8], u32 w, bool atV0, float pos[3]);

TV_HD void reg_corner_position(const CellGeom& c, int corner, float pos[3])
{
	int P[3];
	corner_pos(c, corner, P);
	pos[0] = (float)P[0] * 256.f; pos[1] = (float)P[1] * 256.f; pos[2] = (float)P[2] * 256.f;
}

// x256 position of the vertex described by table word w, as the cell that references it sees it: `atV0` marks the
// one case where the reference places a fresh vertex at corner v0 whatever the endpoint (TransVoxelImpl.cpp:1637)
template <typename D>
TV_HD void reg_vertex_position(const D& d, const CellGeom& c, const i8 V[8], u32 w, bool atV0, float pos[3])
{
	const int v0 = (w >> 4) & 15, v1 = w & 15;
	const int t = edge_t(V[v0], V[v1]);
	if (atV0) reg_corner_position(c, v0, pos);
	else if ((t & 0xFF) == 0) reg_corner_position(c, (t == 0) ? v1 : v0, pos);
	else { int P0[3], P1[3], tt; reg_edge_position(d, c, v0, v1, t, V[v0], V[v1], P0, P1, tt, pos); }
}

template <typename D, typename MF>
TV_HD void reg_corner_vertex(const D& d, const MF& mats, const CellGeom& c, int corner, u32 cellMat, RawVertex& o)
{
	int P[3];
	corner_pos(c, corner, P);
	o.p[0] = (float)P[0] * 256.f; o.p[1] = (float)P[1] * 256.f; o.p[2] = (float)P[2] * 256.f;
	int s[6];
	s[0] = d(P[0] + 1, P[1], P[2]); s[1] = d(P[0] - 1, P[1], P[2]);
	s[2] = d(P[0], P[1], P[2] + 1); s[3] = d(P[0], P[1], P[2] - 1);
	s[4] = d(P[0], P[1] + 1, P[2]); s[5] = d(P[0], P[1] - 1, P[2]);
	const u32 mine = mats(0, P);
	normal_from(s, o.n);
	o.mat = ((cellMat & 0xFF) != (mine & 0xFF)) ? cellMat : mine;
	o.flags = boundary_mask(c.local[0], c.local[1], c.local[2], c.mult, corner, corner);
	finish_secondary(o, c.mult);
}

// ---------------------------------------------------------------------------------------------------------
// Material vote of a level >= 1 cell over its 8 children (child(i) -> id | blend << 8, id 255 = no entry)
// ---------------------------------------------------------------------------------------------------------
TV_HD u32 vote_entries(const u32 e[8])
{
	u32 ids[8], cnt[8], bl[8];
	u32 count = 0;
	for (u32 i = 0; i < 8; ++i) { // children in x-fastest order
		const u32 c = e[i];
		const u32 id = c & 0xFF;
		if (id == EMPTY_MATERIAL) continue;
		bool found = false;
		for (u32 k = 0; k < count; ++k) {
			if (ids[k] == id) { ++cnt[k]; bl[k] += c >> 8; found = true; break; }
		}
		if (!found) { ids[count] = id; cnt[count] = 1; bl[count] = c >> 8; ++count; }
	}
	if (!count) return EMPTY_MATINFO;
	u32 best = 0;
	for (u32 k = 1; k < count; ++k) if (cnt[k] > cnt[best]) best = k; // first maximum wins
	return ids[best] | (((bl[best] / cnt[best]) & 0xFF) << 8);
}

// all eight child entries are fetched before the vote so that the loads are in flight together
template <typename ChildFn>
TV_HD u32 vote_material(const ChildFn& child)
{
	u32 e[8];
#pragma unroll
	for (u32 i = 0; i < 8; ++i) e[i] = child(i);
	return vote_entries(e);
}

// ---------------------------------------------------------------------------------------------------------
// Transition cells.  Face f = 0..5: ZNeg, YNeg, XNeg, ZPos, YPos, XPos (internal axes) = output faces
// YNeg, ZNeg, XNeg, YPos, ZPos, XPos.  In-plane axes (u = column, v = row): z-faces (x,y), y-faces (x,z),
// x-faces (y,z).  Sample k = i + 3j is the full-resolution sample at (u,v) = (i,j) * mult/2 on the block's
// boundary plane; samples 9..12 are the low-res corners (0,0) (2,0) (0,2) (2,2).
// ---------------------------------------------------------------------------------------------------------
struct FaceGeom {
	int axis, ua, va; // normal axis and in-plane axes (0 = x, 1 = y, 2 = z)
	bool positive;
	u32 lowFaceBit;   // 1 << faceId of the low-res cell's face lying on the boundary
	int lowFace;      // that face id (bit order ZPos,YPos,XPos,ZNeg,YNeg,XNeg)
};

TV_HD FaceGeom face_geom(int f)
{
	FaceGeom g;
	const int m = f % 3;
	g.axis = (m == 0) ? 2 : ((m == 1) ? 1 : 0);
	g.ua = (g.axis == 0) ? 1 : 0;
	g.va = (g.axis == 2) ? 1 : 2;
	g.positive = f >= 3;
	g.lowFace = g.positive ? m : m + 3; // f0 -> ZNeg(3), f1 -> YNeg(4), f2 -> XNeg(5), f3 -> ZPos(0), ...
	g.lowFaceBit = 1u << g.lowFace;
	return g;
}

// (u, v, a) = coordinates along the face's in-plane axes and its normal axis -> (x, y, z).  Written with comparisons,
// not as P[fg.ua] = u ...: an array indexed by a run-time value lives in scratch memory on the device.
TV_HD void face_scatter(const FaceGeom& fg, int u, int v, int a, int P[3])
{
	P[0] = fg.axis == 0 ? a : u;
	P[1] = fg.axis == 2 ? v : (fg.axis == 1 ? a : u);
	P[2] = fg.axis == 2 ? a : v;
}

// corner ids (0..7) of the low-res cell for samples 9..12
TV_HD int low_corner_id(const FaceGeom& fg, int k)
{
	const int i = k & 1, j = k >> 1;
	int c = (i << fg.ua) | (j << fg.va);
	if (fg.positive) c |= 1 << fg.axis;
	return c;
}

TV_HD u32 tr_case_code(const i8 v[9])
{
	// weights 1,2,4,0x80,0x100,8,0x40,0x20,0x10 for samples 0..8
	return ((u32)(v[0] >> 7) & 1u) | (((u32)(v[1] >> 7) & 1u) << 1) | (((u32)(v[2] >> 7) & 1u) << 2)
	     | (((u32)(v[3] >> 7) & 1u) << 7) | (((u32)(v[4] >> 7) & 1u) << 8) | (((u32)(v[5] >> 7) & 1u) << 3)
	     | (((u32)(v[6] >> 7) & 1u) << 6) | (((u32)(v[7] >> 7) & 1u) << 5) | (((u32)(v[8] >> 7) & 1u) << 4);
}

// 13 sample values from the 9 plane samples
TV_HD void tr_expand_values(const i8 v9[9], i8 v13[13])
{
#pragma unroll
	for (int i = 0; i < 9; ++i) v13[i] = v9[i];
	v13[9] = v9[0]; v13[10] = v9[2]; v13[11] = v9[6]; v13[12] = v9[8];
}

// zeroMask: bit i = transition sample i (0..12, expanded) is exactly 0 — all these decisions need of the values
TV_HD void tr_vertex_dir_slot_z(const Tables& T, u32 zeroMask, u32 w, int& t, u32& dir, u32& slot, bool& endpoint, int& corner)
{
	const int v0 = (w >> 4) & 15, v1 = w & 15;
	dir = w >> 12; slot = (w >> 8) & 15;
	t = edge_end_bits(zeroMask, v0, v1); // 0 / 256 / 1 = inside
	corner = (t == 0) ? v1 : v0;
	endpoint = (t & 0xFF) == 0;
	if (endpoint) { const u32 cd = T.trCorner(corner); dir = cd >> 4; slot = cd & 15; }
}

TV_HD u32 tr_zero_mask(const i8 v[13])
{
	u32 m = 0;
#pragma unroll
	for (int i = 0; i < 13; ++i) m |= (v[i] == 0 ? 1u : 0u) << i;
	return m;
}

TV_HD void tr_vertex_dir_slot(const Tables& T, const i8 v[13], u32 w, int& t, u32& dir, u32& slot, bool& endpoint, int& corner)
{
	tr_vertex_dir_slot_z(T, tr_zero_mask(v), w, t, dir, slot, endpoint, corner);
}

// bit s set <=> the cell creates and stores a vertex in reuse slot s (0..9)
TV_HD u32 tr_slot_valid(const Tables& T, const i8 v[13], u32 code)
{
	bool anyZero = false;
#pragma unroll
	for (int i = 0; i < 9; ++i) anyZero = anyZero || v[i] == 0;
	if (!anyZero) return T.trOwn(code); // no vertex on a sample point: the mask is a property of the case
	const u32 nv = (u32)(T.trCell(T.trClass(code) & 0x7F)[0]) >> 4;
	u32 m = 0;
	for (u32 vi = 0; vi < nv; ++vi) {
		int t, corner; u32 dir, slot; bool endpoint;
		tr_vertex_dir_slot(T, v, T.trVert(code, vi), t, dir, slot, endpoint, corner);
		if (dir == 8u) m |= 1u << slot;
	}
	return m;
}

struct TrResolution {
	u8 kind;   // RK_REUSE or RK_NEW_EDGE (all new transition vertices use the edge form)
	u8 dir, slot;
	u8 store;  // slot or NO_SLOT
	u8 endpoint;
	int t;     // edge_end() of the cell's own sample values
};

//   mask2 : bit0 = a non-trivial transition cell exists earlier in this row, bit1 = row > 0
//   nb(dcol, drow, slot, &valid, &mat): reuse slot of the cell at (col - dcol, row - drow) of this face
template <typename NB>
TV_HD TrResolution tr_resolve(const Tables& T, u32 zeroMask, u32 w, u32 mask2, u32 myMatId, const NB& nb)
{
	TrResolution r;
	int t, corner; u32 dir, slot; bool endpoint;
	tr_vertex_dir_slot_z(T, zeroMask, w, t, dir, slot, endpoint, corner);
	r.t = t; r.dir = (u8)dir; r.slot = (u8)slot; r.endpoint = endpoint ? 1 : 0;
	bool addForReuse = true;
	if ((dir & mask2) == dir) {
		addForReuse = false;
		bool valid; u32 nmat;
		nb((int)(dir & 1), (int)((dir >> 1) & 1), slot, valid, nmat);
		if (valid && nmat == myMatId) { r.kind = RK_REUSE; r.store = NO_SLOT; return r; }
	}
	r.kind = RK_NEW_EDGE;
	r.store = (addForReuse && dir == 8u) ? (u8)slot : (u8)NO_SLOT;
	return r;
}

struct TrCellGeom {
	int lowBase[3];  // global coordinates of the low-res cell's corner 0
	int local[3];    // low-res cell coordinates in the block
	int mult, level;
};

// global coordinates of transition sample k (0..12)
TV_HD void tr_sample_pos(const FaceGeom& fg, const TrCellGeom& c, int k, int P[3])
{
	int i, j;
	if (k < 9) { i = k % 3; j = k / 3; }
	else { i = ((k - 9) & 1) * 2; j = ((k - 9) >> 1) * 2; }
	const int half = c.mult >> 1;
	int off[3];
	face_scatter(fg, i * half, j * half, fg.positive ? c.mult : 0, off);
	P[0] = c.lowBase[0] + off[0]; P[1] = c.lowBase[1] + off[1]; P[2] = c.lowBase[2] + off[2];
}

// `smp` addresses voxels as sums of one term per axis (tv_fast1.h: F1HostSampler for the dense fields, the kernels'
// F1BrickSampler for the brick mirrors): every edge of a transition cell runs along one axis, so the LOD chain and the two
// stencils need a handful of terms instead of a full address computation per fetch.
// p0, p1: the samples at the vertex's end points v0 = (w >> 4) & 15, v1 = w & 15 (expanded sample numbering 0..12; the caller
// reads the two out of the staged plane: of the cell's 13 samples a new vertex needs no other)
template <typename SMP>
TV_HD void tr_new_vertex(const SMP& smp, const FaceGeom& fg, const TrCellGeom& c, int p0, int p1, u32 w,
                         const TrResolution& r, u32 lowMat, RawVertex& o)
{
	typedef typename SMP::Off Off;
	const int v0 = (w >> 4) & 15, v1 = w & 15;
	int I0[3], I1[3];
	tr_sample_pos(fg, c, v0, I0);
	tr_sample_pos(fg, c, v1, I1);
	float N0[3] = { 0.f, 0.f, 0.f }, N1[3] = { 0.f, 0.f, 0.f };
	int t = r.t, u = 0;
	u32 adjacency = 0;
	if (!r.endpoint) {
		// FindBestVertexInLODChain (:1484-1509): the end points differ along one axis by a power of two; every step
		// halves the (signed) distance, keeping the half that holds the sign change (one fetch per step)
		const int lodOfEdge = (v0 >= 9) ? c.level : c.level - 1;
		for (int lev = lodOfEdge; lev > 0; --lev) {
			const int mx = I0[0] + (I1[0] - I0[0]) / 2, my = I0[1] + (I1[1] - I0[1]) / 2, mz = I0[2] + (I1[2] - I0[2]) / 2;
			const int midV = smp.dist(smp.tx(mx) + smp.ty(my) + smp.tz(mz));
			if (p0 * midV <= 0) { I1[0] = mx; I1[1] = my; I1[2] = mz; p1 = midV; }
			else { I0[0] = mx; I0[1] = my; I0[2] = mz; p0 = midV; }
		}
	}
	// both stencils and both materials in one round trip (an end-point vertex uses one of the two normals; the other
	// stencil is read all the same: no second dependent trip, no branch around loads)
	int a[6], bb[6];
	u32 M0, M1;
	{
		// x+1, x-1, z+1, z-1, y+1, y-1 around P0 / P1 (CalcNormal's order, :1239-1246); P0's requests are issued before
		// P1's address terms are computed (fewer values alive at once: the kernel runs at its register limit)
		{
			const Off x0 = smp.tx(I0[0]), x0m = smp.tx(I0[0] - 1), x0p = smp.tx(I0[0] + 1);
			const Off y0 = smp.ty(I0[1]), y0m = smp.ty(I0[1] - 1), y0p = smp.ty(I0[1] + 1);
			const Off z0 = smp.tz(I0[2]), z0m = smp.tz(I0[2] - 1), z0p = smp.tz(I0[2] + 1);
			const Off yz0 = y0 + z0, xy0 = x0 + y0, xz0 = x0 + z0;
			a[0] = smp.dist(x0p + yz0); a[1] = smp.dist(x0m + yz0); a[2] = smp.dist(xy0 + z0p); a[3] = smp.dist(xy0 + z0m); a[4] = smp.dist(xz0 + y0p); a[5] = smp.dist(xz0 + y0m);
			M0 = smp.mat(x0 + yz0, I0[0], I0[1], I0[2]);
		}
		TV_SCHED_FENCE();
		{
			const Off x1 = smp.tx(I1[0]), x1m = smp.tx(I1[0] - 1), x1p = smp.tx(I1[0] + 1);
			const Off y1 = smp.ty(I1[1]), y1m = smp.ty(I1[1] - 1), y1p = smp.ty(I1[1] + 1);
			const Off z1 = smp.tz(I1[2]), z1m = smp.tz(I1[2] - 1), z1p = smp.tz(I1[2] + 1);
			const Off yz1 = y1 + z1, xy1 = x1 + y1, xz1 = x1 + z1;
			bb[0] = smp.dist(x1p + yz1); bb[1] = smp.dist(x1m + yz1); bb[2] = smp.dist(xy1 + z1p); bb[3] = smp.dist(xy1 + z1m); bb[4] = smp.dist(xz1 + y1p); bb[5] = smp.dist(xz1 + y1m);
			M1 = smp.mat(x1 + yz1, I1[0], I1[1], I1[2]);
		}
	}
	if (r.endpoint) {
		if (t == 0) {
			u = 256;
			gradient_from(bb, N1);
			if (v1 >= 9) { const int cid = low_corner_id(fg, v1 - 9); adjacency = boundary_mask(c.local[0], c.local[1], c.local[2], c.mult, cid, cid); }
		} else {
			u = 0; t = 256;
			gradient_from(a, N0);
			if (v0 >= 9) { const int cid = low_corner_id(fg, v0 - 9); adjacency = boundary_mask(c.local[0], c.local[1], c.local[2], c.mult, cid, cid); }
		}
	} else {
		t = (p0 != p1) ? edge_t_crossing(p0, p1) : 0; // (the chain keeps p0 * p1 <= 0)
		u = 256 - t;
		gradient_from(a, N0);
		gradient_from(bb, N1);
		if (v0 >= 9 && v1 >= 9) adjacency = boundary_mask(c.local[0], c.local[1], c.local[2], c.mult, low_corner_id(fg, v0 - 9), low_corner_id(fg, v1 - 9));
	}
	float P0[3] = { (float)I0[0], (float)I0[1], (float)I0[2] }, P1[3] = { (float)I1[0], (float)I1[1], (float)I1[2] };
	float S0[3] = { P0[0], P0[1], P0[2] }, S1[3] = { P1[0], P1[1], P1[2] };
	if (v0 >= 9 || v1 >= 9) {
		float delta[3], move[3];
		transition_delta(adjacency, c.mult, delta);
		transition_delta(fg.lowFaceBit, c.mult, move); // 0.25 * inward direction of the low-res face
		const bool simple = adjacency == fg.lowFaceBit;
		if (v0 >= 9) {
			S0[0] += delta[0]; S0[1] += delta[1]; S0[2] += delta[2];
			if (simple) { P0[0] += move[0]; P0[1] += move[1]; P0[2] += move[2]; }
		}
		if (v1 >= 9) {
			S1[0] += delta[0]; S1[1] += delta[1]; S1[2] += delta[2];
			if (simple) { P1[0] += move[0]; P1[1] += move[1]; P1[2] += move[2]; }
		}
	}
	const float ft = (float)t, fu = (float)u;
	o.p[0] = ft * P0[0] + fu * P1[0]; o.p[1] = ft * P0[1] + fu * P1[1]; o.p[2] = ft * P0[2] + fu * P1[2];
	o.s[0] = ft * S0[0] + fu * S1[0]; o.s[1] = ft * S0[1] + fu * S1[1]; o.s[2] = ft * S0[2] + fu * S1[2];
	o.flags = adjacency;
	const float wt = ft / 256.f, wu = fu / 256.f;
	o.n[0] = N0[0] * wt + N1[0] * wu; o.n[1] = N0[1] * wt + N1[1] * wu; o.n[2] = N0[2] * wt + N1[2] * wu;
	normalize_fix_zero(o.n);
	if ((M0 & 0xFF) == (M1 & 0xFF) && (M0 & 0xFF) == (lowMat & 0xFF)) o.mat = (M0 & 0xFF) | (lerp_blend(t, u, M0 >> 8, M1 >> 8) << 8);
	else o.mat = lowMat;
}

} // namespace tv

// vx_vertices.inl — the vertex pass: one lane = one output vertex, for every block of every level of a run.
//
// The per-block kernels (k_regular0, ...) decide WHICH vertices exist and where they go (reuse resolution, offsets,
// index lists: the part of PolygonizeBlock that is about topology) and leave an 8-byte descriptor in each new vertex's
// 48-byte slot of the vertex pool.  This kernel does the numeric part of src/TransVoxelImpl.cpp:1579-1718 /
// :1450-1509 / :1330-1369 for all of them at once: end-point samples, LOD chain, central-difference normals,
// material blend, packing — a plain streaming kernel (no LDS, no barriers, no cross-lane traffic) that runs at
// full occupancy; consecutive lanes are neighbouring cells of one block, so their sample gathers hit the same lines.
//
// A descriptor is recognisable in place: its second word is a negative quiet NaN (0xFFFxxxxx), which no vertex
// position can be, so slots that already hold a finished vertex are left alone.
namespace {

enum { VK_REGULAR = 0, VK_TRANSITION = 1 };
constexpr u32 VDESC_TAG = 0xFFF00000u;

__device__ __forceinline__ u32 vdesc_word0(u32 slot, u32 level, u32 kind) { return slot | (level << 22) | (kind << 25); }

__global__ __launch_bounds__(WG) void k_vertices(ExecParamsDev p, u32 first)
{
	const u32 j = first + blockIdx.x * WG + threadIdx.x;
	if (p.P.cursors[CUR_OVF]) return;               // the run is going to be repeated with larger pools
	if (j >= p.P.cursors[CUR_V] || j >= p.P.vertCap) return;
	PolyVertex* out = p.P.verts + j;
	const uint2 d = *(const uint2*)out;
	if ((d.y & 0xFFF00000u) != VDESC_TAG) return;
	const u32 slot = d.x & 0x3FFFFFu, level = (d.x >> 22) & 7u;
	if (level >= p.G.levels || slot >= p.levels[level].cap) return; // not a descriptor of this run
	const LevelDesc& L = p.levels[level];
	const GridView& g = p.G.grid;
	u32 bx, by, bz;
	block_coords(L.slotCoord[slot], L.cnt, bx, by, bz);
	const u32 c = d.y & 0xFFFu;
	const int v0 = (int)((d.y >> 12) & 7u), v1 = (int)((d.y >> 15) & 7u);
	const bool atV0 = ((d.y >> 18) & 1u) != 0;
	CellGeom geo;
	geo.mult = (int)L.mult; geo.level = (int)level;
	geo.local[0] = (int)(c & 15); geo.local[1] = (int)((c >> 4) & 15); geo.local[2] = (int)(c >> 8);
	geo.base[0] = (int)((bx * 16 + (c & 15)) * L.mult); geo.base[1] = (int)((by * 16 + ((c >> 4) & 15)) * L.mult); geo.base[2] = (int)((bz * 16 + (c >> 8)) * L.mult);
	const GlobalDist dd{ &g };
	int P0[3], P1[3];
	corner_pos(geo, v0, P0);
	corner_pos(geo, v1, P1);
	const int val0 = dd(P0[0], P0[1], P0[2]), val1 = dd(P1[0], P1[1], P1[2]);
	const u32 cellMat = level == 0 ? mat_at(g, geo.base[0], geo.base[1], geo.base[2]) : (u32)L.cache[(size_t)slot * BLOCK_CELLS + c];
	const unsigned long long lut = lut_row(p.G.lut, cellMat);
	const int e = edge_end(val0, val1);
	RawVertex rv;
	if (e != 1) {
		const int corner = atV0 ? v0 : ((e == 0) ? v1 : v0);
		reg_corner_vertex(dd, GridMaterials{ &g }, geo, corner, cellMat, rv);
	} else {
		reg_edge_vertex(dd, GridMaterials{ &g }, geo, v0, v1, edge_t(val0, val1), val0, val1, cellMat, rv);
	}
	pack_vertex_row(rv, lut, out);
}

} // namespace

/* oracle/vxo_api.h — C interface shared by the two CPU checkers:
 *   oracle/_ref/libvoxels_ref.so  : the UNMODIFIED reference sources (/root/reference/src/*.cpp)
 *                                   compiled where they lie, behind oracle/ref_harness.cpp;
 *   oracle/libvoxels_port.so      : oracle/port.cpp, this repo's CPU restatement of the same path.
 * TEST INFRASTRUCTURE ONLY: nothing under voxels_amd/ or the product libraries may include, link
 * or load anything declared here (only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline
 * leg do).  Coordinates follow the reference: grids are Z-up, x fastest, dense index
 * (z*N + y)*N + x; polygon output is Y-up (include/Polygonizer.h of the reference).
 */
#ifndef VXO_API_H
#define VXO_API_H
#include <stdint.h>
#include <stddef.h>
#ifdef __cplusplus
extern "C" {
#endif

typedef struct vxo_grid vxo_grid;
typedef struct vxo_surface vxo_surface;

/* 48-byte vertex, bit-compatible with Voxels::PolygonVertex (include/Polygonizer.h:14-48). */
typedef struct vxo_vertex {
	float pos[3];
	float sec[4];      /* sec[3] holds the adjacency bit mask as raw integer bits */
	float nrm[3];
	uint8_t tex[8];    /* Reserved, Blend, Uxz, Txz, Uny, Upy, Tny, Tpy */
} vxo_vertex;

typedef struct vxo_block_info {
	uint32_t id;
	uint32_t n_verts;
	uint32_t n_idx;
	uint32_t n_tverts[6];
	uint32_t n_tidx[6];
	float min_corner[3];
	float max_corner[3];
} vxo_block_info;

const char* vxo_kind(void); /* "reference" or "port" */

/* Grid built from already-quantised bytes (Grid::Create(w,d,h) + Modify*Data per block:
 * no +-4 clamp, BF_Empty computed by the codec). */
vxo_grid* vxo_grid_from_dense(uint32_t n, const int8_t* dist, const uint8_t* mat, const uint8_t* blend);
/* Grid built through Grid::Create(w,d,h,0,0,0,1,surface): float distances are quantised by the
 * reference rule (VoxelGrid.cpp:37-50). */
vxo_grid* vxo_grid_from_float(uint32_t n, const float* values, const uint8_t* mat, const uint8_t* blend);
/* Grid::Create(w, heightmap) (src/VoxelGrid.cpp:159-213): heightmap = w*w signed bytes, row-major (row = y) */
vxo_grid* vxo_grid_from_heightmap(uint32_t n, const char* heightmap);
void vxo_grid_destroy(vxo_grid* g);
uint32_t vxo_grid_size(const vxo_grid* g);
void vxo_grid_read_dense(const vxo_grid* g, int8_t* dist, uint8_t* mat, uint8_t* blend);
void vxo_grid_block_flags(const vxo_grid* g, uint8_t* empty_flags /* (n/16)^3, block id order */);
uint32_t vxo_grid_memory_size(vxo_grid* g);
/* Grid::InjectSurface with a ball of radius r centred on `pos` (d = |p| - r in brush coords). */
void vxo_grid_inject_ball(vxo_grid* g, const float pos[3], const float ext[3], float radius, int type,
                          float out_min[3], float out_max[3]);
void vxo_grid_inject_material(vxo_grid* g, const float pos[3], const float ext[3], uint8_t material,
                              int add, float out_min[3], float out_max[3]);
/* PackForSave / Load (file format v1). pack returns the size; copies min(size, cap) bytes. */
size_t vxo_grid_pack(const vxo_grid* g, char* out, size_t cap);
vxo_grid* vxo_grid_load(const char* blob, size_t size);

/* Polygonizer::Execute. lut = 256 x {Ids0[3], Ids1[3]}, valid[id]==0 -> GetMaterial returns null.
 * threads <= 0 keeps the OpenMP default. */
vxo_surface* vxo_execute(const vxo_grid* g, const uint8_t* lut, const uint8_t* valid, int threads);
/* Execute with a Modification over `prev` (mutated in place and returned). Dirty corners are in
 * OUTPUT (Y-up) coordinates as Grid::InjectSurface returns them. Returns the number of modified
 * block ids; copies min(count, cap) of them. */
uint32_t vxo_execute_modify(const vxo_grid* g, const uint8_t* lut, const uint8_t* valid, int threads,
                            vxo_surface* prev, const float min_corner[3], const float max_corner[3],
                            uint32_t* modified_ids, uint32_t cap);
void vxo_surface_destroy(vxo_surface* s);
uint32_t vxo_surface_levels(const vxo_surface* s);
void vxo_surface_extents(const vxo_surface* s, float out[3]);
uint32_t vxo_surface_blocks(const vxo_surface* s, uint32_t level);
/* totals[0..3] = verts, idx, tverts, tidx summed over the level's blocks */
void vxo_surface_level_totals(const vxo_surface* s, uint32_t level, uint64_t totals[4]);
/* Concatenated dump of one level in block order: regular then per-face transition data. */
void vxo_surface_dump_level(const vxo_surface* s, uint32_t level, vxo_block_info* infos,
                            vxo_vertex* verts, uint32_t* idx, vxo_vertex* tverts, uint32_t* tidx);
/* stats[0..3] = BlocksCalculated, TrivialCells, NonTrivialCells, DegenerateTrianglesRemoved;
 * stats[4..19] = PerCaseCellsCount */
void vxo_surface_stats(const vxo_surface* s, uint32_t stats[20]);
uint32_t vxo_surface_cache_bytes(const vxo_surface* s);
uint32_t vxo_surface_polygon_bytes(const vxo_surface* s);
uint32_t vxo_log_errors(void); /* number of LS_Error messages seen since load */

#ifdef __cplusplus
}
#endif
#endif

/* voxels_hip.h — C ABI of libvoxels_hip.so, the MI355X (gfx950) TransVoxel polygonizer.
 *
 * This is the drop-in boundary for the reference's polygonization path.  The reference has no C interface
 * below Polygonizer::Execute (include/Polygonizer.h:230-232 -> src/TransVoxelImpl.cpp:74-79, :2153-2169,
 * TransVoxelRun::Execute :468-538); the entry points below are what a host binding of that call needs:
 * the C++ host layer of this repo (include/Voxels.h, voxels_amd/csrc/vx_api_cpp.cpp) implements
 * Voxels::Polygonizer::Execute on top of them, and INTEGRATION.md shows the equivalent patch to the
 * reference's own TransVoxelImpl::Execute.
 *
 * Conventions: plain pointers and sizes only; every function returns VX_OK (0) or a negative VX_ERR_* code and
 * never throws; vx_last_error() gives a message.  Grids are cubes of edge n (multiple of 16), Z-up, dense,
 * x fastest: index (z*n + y)*n + x (reference: src/VoxelGrid.h:31-35).  Output is Y-up, exactly the bytes of
 * Voxels::PolygonVertex / BlockPolygons (include/Polygonizer.h:14-106).  One polygonization at a time per
 * context (the reference has the same restriction, src/TransVoxelImpl.cpp:2144).
 */
#ifndef VOXELS_HIP_H
#define VOXELS_HIP_H

#include <stdint.h>
#include <stddef.h>

#ifdef __cplusplus
extern "C" {
#endif

#define VX_OK 0
#define VX_ERR_INVALID (-1)   /* bad argument or call order */
#define VX_ERR_DEVICE (-2)    /* HIP runtime error (no GPU, out of memory, launch failure) */
#define VX_ERR_OVERFLOW (-3)  /* output pools could not be grown */

typedef struct vx_ctx vx_ctx;

/* 48 bytes, bit-identical to Voxels::PolygonVertex (include/Polygonizer.h:14-48) */
typedef struct vx_vertex {
	float pos[3];
	float sec[4];     /* sec[3]: transition-face adjacency mask as raw integer bits */
	float nrm[3];
	uint8_t tex[8];   /* Reserved, Blend, Uxz, Txz, Uny, Upy, Tny, Tpy */
} vx_vertex;

/* One emitted block = one Voxels::BlockPolygons (include/Polygonizer.h:52-106) */
typedef struct vx_block_info {
	uint32_t id;            /* BlockPolygons::GetId */
	uint32_t n_verts;       /* GetVertices count */
	uint32_t n_idx;         /* GetIndices count */
	uint32_t n_tverts[6];   /* GetTransitionVertices count per TransitionFaceId */
	uint32_t n_tidx[6];     /* GetTransitionIndices count per TransitionFaceId */
	float min_corner[3];    /* GetMinimalCorner */
	float max_corner[3];    /* GetMaximalCorner */
} vx_block_info;

typedef struct vx_exec_info {
	uint32_t levels;            /* LOD levels produced (PolygonSurface::GetLevelsCount) */
	uint32_t retries;           /* re-runs after growing the output pools */
	float device_ms;            /* device time of the last run, HIP events on the context's stream */
	uint64_t total_verts;       /* the vertex pool's cursor: vertices of all meshes (regular + transition, incl. blocks dropped as
	                             * empty) PLUS the ranges a table-driven block of a level >= 1 had reserved when it turned out to
	                             * hold a zero sample and was handed to the general pass (a handful of blocks per run; the
	                             * general pass reserves again).  Byte figures derived from meshes - bench.py's roofline,
	                             * mesh_bytes - sum the blocks' own counts (vx_download_level / vx_level_counts), not this. */
	uint64_t total_indices;     /* the index pool's cursor, likewise */
	uint32_t active_blocks[8];  /* surface-bearing blocks per level */
	uint64_t algorithmic_bytes; /* n^3 + 2*4096*surface blocks + 48*V + 4*I */
	uint32_t blocks_read;       /* level-0 blocks whose distance samples the run had to read (the others are proven
	                               surface-free by the BF_Empty flags of their 27-neighbourhood); 0 for incremental runs */
	float mirror_ms;            /* device time this call spent bringing the library's mirrors of the grid up to date (brick
	                               order, lattice copies, sign summaries: the one place where all n^3 samples are read) —
	                               0 when the grid did not change since the last run; not part of device_ms */
	uint32_t first_meshed_level; /* 0 for an ordinary run; vx_polygonize_from: the first level whose meshes the run produced */
} vx_exec_info;

/* ---- context ------------------------------------------------------------------------------------------- */
/* Number of HIP devices this process sees (0 and VX_ERR_DEVICE when there is none). */
int vx_device_count(int* count);
int vx_ctx_create(int device_index, vx_ctx** out);
void vx_ctx_destroy(vx_ctx* ctx);
const char* vx_last_error(const vx_ctx* ctx);
/* Run on a caller-provided hipStream_t (e.g. PyTorch's current stream); NULL = the context's own stream. */
int vx_set_stream(vx_ctx* ctx, void* hip_stream);

/* ---- grid residency (what TransVoxelRun reads through VoxelGrid::GetBlockData / GetMaterialBlockData /
 *      IsBlockEmpty, src/VoxelGrid.cpp:586-608) ------------------------------------------------------------ */
/* Copy a whole host grid to the device. empty_flags[(n/16)^3] = BF_Empty of every block in block-id order
 * (src/VoxelGrid.h:139-144); mat/blend may be NULL (all zero). */
int vx_grid_upload(vx_ctx* ctx, uint32_t n, const int8_t* dist, const uint8_t* mat, const uint8_t* blend,
                   const uint8_t* empty_flags);
/* Grid::Create(w, heightmap) (src/VoxelGrid.cpp:159-213) evaluated on the device: heightmap = w*w signed bytes, row = y;
 * distance(x,y,z) = clamp((z - 127) - heightmap[y][x], -127, 127) squeezed to the grid's +-4 range, materials 0; BF_Empty of
 * every block by the codec's rule.  Only the w*w bytes cross PCIe. */
int vx_grid_create_heightmap(vx_ctx* ctx, uint32_t w, const int8_t* heightmap);
/* The same, from the Grid file format v1 (what Grid::PackForSave writes and Grid::Load reads, src/VoxelGrid.cpp:215-315):
 * header {1, w, d, h}, 3 stream sizes per block, then per block in id order {flags, distance stream, material stream,
 * blend stream}; a stream is RLE pairs (u8 run length, value) or 4096 raw bytes when its BF_*Uncompressed flag is set
 * (CompressBlock / DecompressBlock, :610-694).  The blob goes to the device as it is and is expanded there; BF_Empty
 * comes from the per-block flags.  Replaces Grid::Load + vx_grid_upload (no host decode, no 3 bytes/voxel transfer). */
int vx_grid_upload_packed(vx_ctx* ctx, const void* blob, uint64_t size);
/* The resident grid as a Grid file (Grid::PackForSave, src/VoxelGrid.cpp:269-315): every block is run-length encoded on
 * the device by the codec's rules (runs of at most 255; a stream whose code would exceed 4096 bytes is stored raw and
 * flagged), the file is assembled in `out`.  Byte-identical to what the reference writes for the same grid.
 * *size receives the file size; with out == NULL (or capacity too small: VX_ERR_INVALID) nothing is written. */
int vx_grid_pack(vx_ctx* ctx, void* out, uint64_t capacity, uint64_t* size);
/* One 16^3 block of the resident grid back to the host, x fastest (Grid::GetBlockDistanceData / GetBlockMaterialData,
 * src/VoxelGrid.cpp:586-608); any output may be NULL.  empty_flag receives BF_Empty. */
int vx_grid_read_block(vx_ctx* ctx, uint32_t block_id, int8_t* dist, uint8_t* mat, uint8_t* blend, uint8_t* empty_flag);
/* Use caller-owned DEVICE memory (multi-GPU slabs, PyTorch tensors).  This rank polygonizes the z-range
 * [z_begin, z_end) of the global n^3 grid.  d_dist holds the z-planes [dist_z0, ...) and must cover
 * [z_begin-1, z_end+1] clamped to the grid; d_mat/d_blend hold planes [mat_z0, ...) covering [z_begin, z_end]
 * clamped.  d_empty_flags is the FULL (n/16)^3 flag array (neighbour layers of other ranks included). */
/* The library keeps brick-ordered mirrors of the fields for its gathers and refreshes them where IT changes
 * the grid (vx_grid_fill_terrain, vx_halo_exchange*).  A caller that rewrites attached memory itself after a
 * polygonization has run must call vx_grid_invalidate (below) — or attach again — before the next one; otherwise that run
 * silently polygonizes the OLD contents (nothing detects the staleness). */
int vx_grid_attach(vx_ctx* ctx, uint32_t n, uint32_t z_begin, uint32_t z_end,
                   const void* d_dist, int32_t dist_z0, const void* d_mat, const void* d_blend, int32_t mat_z0,
                   const void* d_empty_flags);
/* The same for a slab cut along y (the usual choice for terrains: a height field puts nearly all of its surface into a
 * few z-layers, so z-slabs leave most ranks idle): this rank polygonizes the rows [y_begin, y_end) of every z-plane.
 * d_dist is [n planes][dist_rows rows][n] with row 0 = global y dist_y0 and must cover [y_begin-1, y_end+1] clamped;
 * d_mat / d_blend are [n][mat_rows][n] with row 0 = global y mat_y0 covering [y_begin, y_end] clamped. */
int vx_grid_attach_y(vx_ctx* ctx, uint32_t n, uint32_t y_begin, uint32_t y_end,
                     const void* d_dist, int32_t dist_y0, uint32_t dist_rows,
                     const void* d_mat, const void* d_blend, int32_t mat_y0, uint32_t mat_rows, const void* d_empty_flags);
/* A slab of rows of a HOST grid onto the device, halo included: the context polygonizes the rows [y_begin, y_end) of every
 * z-plane of the whole n^3 host grid `dist` / `mat` / `blend` (layout of vx_grid_upload; mat / blend may be NULL) and copies
 * what that takes - distance rows [y_begin - 1, y_end + 2), material and blend rows [y_begin, y_end + 1), clamped to the grid,
 * and all BF_Empty flags - into memory of its own (strided copies straight from the host arrays: a host that owns the whole
 * Voxels::Grid needs no halo exchange between its devices).  Afterwards the context is in the state vx_grid_attach_y leaves
 * it in.  What a multi-device Polygonizer::Execute calls once per device (voxels_amd/csrc/vx_api_cpp.cpp). */
int vx_grid_upload_slab_y(vx_ctx* ctx, uint32_t n, uint32_t y_begin, uint32_t y_end, const int8_t* dist, const uint8_t* mat,
                          const uint8_t* blend, const uint8_t* empty_flags);
/* Tell the library that the caller rewrote the resident (attached) fields in place — the zero-copy use of
 * vx_grid_attach*, where the application edits its own device tensors (the reference's Grid::Modify*BlockData on
 * memory the library does not own).  The mirrors are rebuilt by the next polygonization; the emptiness flags stay the
 * caller's business, as with vx_grid_attach. */
int vx_grid_invalidate(vx_ctx* ctx);
/* Forget what earlier runs of this context learned about ITS surfaces - which capacity classes to launch, how many blocks the
 * general passes take over, how many upper-queue items a run has: the next run starts from the conservative defaults of a new
 * context (all capacity classes launched).  For a context that is handed to another owner with another grid (libVoxels.so:
 * the context InitializeVoxels warmed up on a toy terrain, adopted by the application's first Polygonizer).  No reference
 * counterpart: the reference keeps no state between Execute calls beyond the PolygonSurface itself. */
int vx_ctx_forget_hints(vx_ctx* ctx);
/* ---- generation on the device ------------------------------------------------------------------------------------------
 * Grid::Create(w, h, d, ..., VoxelSurface*) samples an application callback on the host (src/VoxelGrid.cpp:79-132) and
 * quantises the samples (:37-50).  For the benchmark's synthetic surface (include/voxels_synth.h, vxs_terrain) the same
 * step runs where the grid lives: the fields are byte for byte what vxs_terrain + vxs_block_empty_flags produce on the
 * host, without 3 n^3 bytes crossing PCIe.
 * vx_grid_create_terrain: a whole n^3 grid owned by the context (like vx_grid_upload).
 * vx_grid_fill_terrain: the slab attached with vx_grid_attach / vx_grid_attach_y — every resident layer that lies
 * inside the grid (the halo included) and the BF_Empty flags of the rank's own blocks (the neighbours' flag layers come
 * with vx_halo_exchange). */
int vx_grid_create_terrain(vx_ctx* ctx, uint32_t n, uint32_t seed);
/* style as in vxs_terrain_ex (include/voxels_synth.h): 0 = the terrain, 1 = "caves" (surface in a large share of all blocks) */
int vx_grid_create_terrain_ex(vx_ctx* ctx, uint32_t n, uint32_t seed, uint32_t style);
int vx_grid_fill_terrain(vx_ctx* ctx, uint32_t seed);

/* ---- multi-GPU: halo exchange of attached slabs -------------------------------------------
 * The reference has one address space and an OpenMP block loop (src/TransVoxelImpl.cpp:500-503); here the grid is cut
 * into slabs (vx_grid_attach / vx_grid_attach_y), one per GPU, rank r owning the r-th slab along the cut axis.  What a
 * rank reads beyond its own layers (a layer = a z-plane, or the y-row of every plane) comes from its neighbours:
 *   from the slab above: 2 distance layers, 1 material layer, 1 blend layer, the BF_Empty flags of its first block layer
 *   from the slab below: 1 distance layer, the BF_Empty flags of its last block layer
 * The attached buffers must have exactly that halo (dist_z0 = z_begin - 1 with 3 layers more than the slab, mat_z0 =
 * z_begin with 1 more; likewise along y), as voxels_amd/slab.py lays them out.
 *
 * vx_halo_exchange: one process per GPU.  vx_comm_init joins an RCCL communicator (the id comes from
 * vx_comm_unique_id on rank 0 and travels by whatever means the launcher has); the exchange is one grouped
 * ncclSend/ncclRecv batch per call on the context's stream with pack / unpack kernels around it — no host wait.
 * vx_halo_exchange_group: one process driving several contexts (one per GPU, or several slabs on one GPU): the same
 * packing with peer copies as transport; contexts in slab order. */
#define VX_COMM_ID_BYTES 128
int vx_comm_unique_id(void* id /* VX_COMM_ID_BYTES bytes out */);
int vx_comm_init(vx_ctx* ctx, int nranks, int rank, const void* id /* VX_COMM_ID_BYTES bytes */);
int vx_comm_destroy(vx_ctx* ctx);
int vx_halo_exchange(vx_ctx* ctx);
int vx_halo_exchange_group(vx_ctx* const* ctxs, int count);
/* Re-upload `count` edited 16^3 blocks (block ids, x-fastest 4096-byte blocks) + the full flag array. */
int vx_grid_update_blocks(vx_ctx* ctx, uint32_t count, const uint32_t* block_ids, const int8_t* dist,
                          const uint8_t* mat, const uint8_t* blend, const uint8_t* empty_flags);
/* ---- edits on the device (Grid::InjectSurface / Grid::InjectMaterial, src/VoxelGrid.cpp:388-584) -------------------
 * For a grid that lives on the device only (vx_grid_upload / vx_grid_upload_packed): the same arithmetic per voxel, the
 * same per-block sections and the same "touched block" rule as the reference, BF_Empty of every touched block
 * recomputed by the codec's rule (CompressBlock, :610-672).  out_min / out_max receive the modified box exactly as
 * Grid::Inject* returns it (output, Y-up, order) — feed it to vx_polygonize_dirty.
 * vx_grid_inject_ball: InjectSurface with the analytic VoxelSurface  f(x,y,z) = sqrt(x^2 + y^2 + z^2) - radius  sampled
 * relative to `position` (the sphere brush of doc_source/Modification.md); type = InjectionType (0 IT_Add,
 * 1 IT_SubtractAddInner, 2 IT_Subtract). */
int vx_grid_inject_ball(vx_ctx* ctx, const float position[3], const float extents[3], float radius, int type,
                        float out_min[3], float out_max[3]);
int vx_grid_inject_material(vx_ctx* ctx, const float position[3], const float extents[3], uint8_t material,
                            int add_subtract_blend, float out_min[3], float out_max[3]);

/* MaterialMap::GetMaterial resolved on the host (include/MaterialMap.h:19-30): lut[id] = {DiffuseIds0[3],
 * DiffuseIds1[3]}, valid[id] == 0 means GetMaterial returned NULL (texture bytes stay 0). */
int vx_material_lut(vx_ctx* ctx, const uint8_t* lut /*256*6*/, const uint8_t* valid /*256*/);

/* ---- polygonization = TransVoxelRun::Execute (src/TransVoxelImpl.cpp:468-538) ------------------------ */
/* num_levels = 0: all log2(n/16)+1 levels like the reference; otherwise only levels 0..num_levels-1 (the
 * "last level has no transitions" rule still uses the reference's level count). */
int vx_polygonize(vx_ctx* ctx, uint32_t num_levels, vx_exec_info* info);
/* The same run without the meshes of the levels below first_meshed_level - for a caller that gets those from other devices
 * (libVoxels.so with VOXELS_DEVICES = N: helper contexts polygonize the finer levels slab by slab) but needs everything a later
 * Modification continues from in THIS context: the slot maps, non-trivial / consistency bitmaps and material caches of every level
 * (the reference keeps them in the PolygonMap, src/TransVoxelImpl.h:81-133; a Modification reads the caches of blocks it does not
 * rebuild, :753-838).  Levels below first_meshed_level list no blocks and count nothing into the statistics; info->first_meshed_level
 * says what the run really did (0 where the partial form is not available - dense surfaces, stage timing: every level was meshed). */
int vx_polygonize_from(vx_ctx* ctx, uint32_t num_levels, uint32_t first_meshed_level, vx_exec_info* info);
/* Incremental re-polygonization of a dirty box (src/TransVoxelImpl.cpp:429-465); corners in OUTPUT (Y-up)
 * coordinates as Grid::InjectSurface returns them.  Returns the new block ids. */
int vx_polygonize_dirty(vx_ctx* ctx, const float min_corner[3], const float max_corner[3], vx_exec_info* info,
                        uint32_t* modified_ids, uint32_t cap, uint32_t* count);

/* Incremental runs append rebuilt blocks to the output pools and leave the replaced blocks' ranges behind.  This packs
 * the live meshes to the front of fresh pools on the device (offsets reported by vx_level_ranges change, contents and
 * order of everything downloaded do not).  vx_polygonize_dirty calls it by itself once more than half of the pools is
 * dead; applications holding device pointers may call it at a time of their choosing. */
int vx_compact_pools(vx_ctx* ctx);

/* ---- results (PolygonSurface accessors, include/Polygonizer.h:136-178) ------------------------------- */
int vx_level_counts(vx_ctx* ctx, uint32_t level, uint32_t* n_blocks, uint64_t totals[4] /* verts, idx, tverts, tidx */);
/* Blocks of one level in GetBlockForLevel order, concatenated: regular vertices/indices, then per block the
 * transition vertices/indices of faces 0..5.  Any output pointer may be NULL. */
int vx_download_level(vx_ctx* ctx, uint32_t level, vx_block_info* infos, vx_vertex* verts, uint32_t* idx,
                      vx_vertex* tverts, uint32_t* tidx);
/* Device-resident hand-off (renderer interop): the meshes stay in two device pools; a block's meshes are contiguous
 * ranges of them.  A full run rewrites the pools; an incremental run appends the rebuilt blocks behind what is there
 * (ranges of kept blocks stay valid, ranges of replaced blocks become garbage until the next full run).  d_verts /
 * d_indices are valid until the next run on this context (an incremental run may move the pools when it has to grow
 * them); indices are relative to the start of their own mesh, exactly as downloaded. */
typedef struct vx_block_ranges {
	uint32_t v_off, i_off;        /* first vertex / first index of the regular mesh in the pools */
	uint32_t tv_off[6], ti_off[6]; /* the same for the six transition meshes */
} vx_block_ranges;
int vx_device_meshes(vx_ctx* ctx, const vx_vertex** d_verts, const uint32_t** d_indices, uint64_t* n_verts, uint64_t* n_indices);
int vx_level_ranges(vx_ctx* ctx, uint32_t level, vx_block_ranges* ranges /* one per block, vx_download_level order */);
/* The same two pools for a renderer in ANOTHER process (or behind another API's external-memory import): inter-process
 * handles of the two device allocations (hipIpcGetMemHandle; on this platform dmabuf-backed, HSA_ENABLE_IPC_MODE_LEGACY=0).
 * The importer maps them with hipIpcOpenMemHandle and reads n_verts vertices / n_indices indices from the mapping's start;
 * vx_level_ranges / vx_device_block_table (copied over by the application) say which ranges are which block's.  The reference
 * hands out per-block host arrays (include/Polygonizer.h:72-99); this is their device-resident, cross-process form.
 * `generation` changes whenever the pools were rewritten or replaced (every full run; an incremental run that had to pack
 * or grow them): an importer holding a mapping of an older generation must drop it and ask again.  Appending incremental
 * runs keep the generation and only raise the counts.  Exporter and importer must run on the same HIP runtime (a handle of
 * ROCm 7.0's runtime is not accepted by 7.2's hipIpcOpenMemHandle: tests/ipc_reader.py). */
typedef struct vx_ipc_meshes {
	uint8_t verts_handle[64], indices_handle[64]; /* hipIpcMemHandle_t, bytewise */
	uint64_t n_verts, n_indices;
	uint64_t verts_capacity, indices_capacity;    /* elements the allocations hold (what an appending run may still fill) */
	uint64_t generation;
} vx_ipc_meshes;
int vx_export_meshes(vx_ctx* ctx, vx_ipc_meshes* out);
/* Host copy of both pools in ONE step (what BlockPolygons::GetVertices / GetIndices, include/Voxels.h:210-231, hand out —
 * per-block arrays — as views: block k of a level owns verts[ranges[k].v_off ..] etc., with vx_level_ranges /
 * vx_download_level(infos only) giving offsets and counts).  The copy lands in page-locked memory at DMA speed and is
 * not copied again: the caller owns `arena` (and with it verts / indices) until vx_host_meshes_release, independent of
 * later runs and of the context's lifetime.  Released arenas are recycled by the next acquire (page-locking memory is
 * slow; a steady caller never pays it twice).
 * in/out: meshes->arena == NULL asks for a fresh copy; an arena returned by an earlier acquire on the same context is
 * brought up to date instead (after incremental runs only the appended part of the pools travels; it may be exchanged
 * for a larger one - always use the returned pointers). */
typedef struct vx_host_meshes {
	const vx_vertex* verts;
	const uint32_t* indices;
	uint64_t n_verts, n_indices;
	void* arena;
} vx_host_meshes;
int vx_host_meshes_acquire(vx_ctx* ctx, vx_host_meshes* meshes);
void vx_host_meshes_release(void* arena);
/* frees the recycled arenas of this process (optional; e.g. before unloading the library) */
void vx_host_meshes_trim(void);
/* Page-locks an arena for at least n_verts vertices and n_indices indices ahead of time and puts it into the process's
 * recycling list: the first vx_host_meshes_acquire of that size then finds it instead of page-locking inside the call
 * (locking 0.5 GB takes ~130 ms on the bench host, copying into it 9 ms).  What InitializeVoxels does when
 * VOXELS_PREWARM_MB is set (reference src/Voxels.cpp:35-60 is where an application pays one-time costs). */
int vx_host_meshes_reserve(vx_ctx* ctx, uint64_t n_verts, uint64_t n_indices);
/* The same lists as a device-resident table: what PushBlocksToResult (src/TransVoxelImpl.cpp:1266-1293) assembles per
 * block — ranges of its 1 + 6 meshes in the pools, id (:149-152), Y-up corners (:1283-1293) — for the blocks of one level
 * in GetBlockForLevel order (:395-401; blocks without a regular vertex are left out, :1274).  A full run writes the
 * tables on the device as its last step, so a renderer can consume a run without any host round trip beyond the
 * per-level counts; after an incremental run the (host-maintained) lists are uploaded on the first call.
 * The table is valid until the next run on this context. */
typedef struct vx_listed_block {
	uint32_t coord_id;                     /* (bz * cnt + by) * cnt + bx, internal axes (Z up) */
	uint32_t v_off, v_count, i_off, i_count;
	uint32_t tv_off[6], tv_count[6], ti_off[6], ti_count[6];
	uint32_t degenerate, nt_cells, reserved;
	uint32_t id;
	float min_corner[3], max_corner[3];
} vx_listed_block;
int vx_device_block_table(vx_ctx* ctx, uint32_t level, const vx_listed_block** d_table, uint32_t* n_blocks);
/* stats[0..3] = BlocksCalculated, TrivialCells, NonTrivialCells, DegenerateTrianglesRemoved; stats[4..19] =
 * PerCaseCellsCount (include/Polygonizer.h:110-132) */
int vx_stats(vx_ctx* ctx, uint32_t stats[20]);

/* The device forms of the exactness-critical arithmetic checked exhaustively on the GPU they run on (the reference does
 * this arithmetic on the host: t = (v1 << 8) / (v1 - v0), src/TransVoxelImpl.cpp:1591; normalizeFixZero, :93-103):
 * results[0] = crossed int8 sample pairs whose t differs from the truncated quotient, results[1] = gradients whose normal
 * changes when the gradient is scaled by 0.5, results[2] = gradients whose normal differs from fp32 sqrt + division;
 * results[11] = gradients whose normal differs between normalize_gradient (the cheaper form the fast passes use for the
 * end-point normals, valid for integer gradients only) and normalizeFixZero; all four must be 0.  results[3..10] count
 * mismatches of other candidate forms (informational). */
int vx_selftest(vx_ctx* ctx, uint32_t results[16]);

/* Optional per-stage device timing (HIP events between the kernels of vx_polygonize; adds a few event records).
 * ms[0..7] = reset + block classes, classify, hierarchy, material (all levels), regular cells of level 0, of the levels >= 1,
 * transition cells, block lists of the LAST run. */
int vx_set_stage_timing(vx_ctx* ctx, int enable);
/* What the eight slots of vx_stage_times meant in the last run with stage timing: 0 = the chain of launches (reset + block
 * classes, classify, hierarchy, material, regular level 0, regular levels >= 1, transition, block lists); 1 = the single-stream
 * form (k_reset + k_run_head, nothing, nothing, k_main, what follows k_main for level 0, ... for the levels >= 1, nothing,
 * block lists - with stage timing the parts of k_tail are launches of their own, so that they can be timed). */
int vx_stage_layout(vx_ctx* ctx, int* layout);
/* Diagnostics: the first `count` 32-bit words of the run's device header (queue heads, counters), copied on a stream of
 * its own so that it also works - from another host thread - while a run is in flight. */
int vx_debug_header(vx_ctx* ctx, uint32_t* out, uint32_t count);
int vx_stage_times(vx_ctx* ctx, float ms[8]);

/* name of the code object actually running the kernels ("hip:gfx950") — lets callers assert the native path */
const char* vx_backend(void);

#ifdef __cplusplus
}
#endif
#endif

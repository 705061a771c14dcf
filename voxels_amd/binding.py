"""ctypes binding of include/voxels_hip.h (libvoxels_hip.so)."""
import ctypes as C
import os
import sys

import numpy as np

LISTED_BLOCK_DTYPE = np.dtype([
    ("coord_id", "<u4"), ("v_off", "<u4"), ("v_count", "<u4"), ("i_off", "<u4"), ("i_count", "<u4"),
    ("tv_off", "<u4", 6), ("tv_count", "<u4", 6), ("ti_off", "<u4", 6), ("ti_count", "<u4", 6),
    ("degenerate", "<u4"), ("nt_cells", "<u4"), ("reserved", "<u4"), ("id", "<u4"),
    ("min_corner", "<f4", 3), ("max_corner", "<f4", 3)])
VERTEX_DTYPE = np.dtype([("pos", "<f4", 3), ("sec", "<f4", 4), ("nrm", "<f4", 3), ("tex", "u1", 8)])
BLOCK_INFO_DTYPE = np.dtype([
    ("id", "<u4"), ("n_verts", "<u4"), ("n_idx", "<u4"),
    ("n_tverts", "<u4", 6), ("n_tidx", "<u4", 6),
    ("min_corner", "<f4", 3), ("max_corner", "<f4", 3)])
assert VERTEX_DTYPE.itemsize == 48 and BLOCK_INFO_DTYPE.itemsize == 84


class VoxelsHipError(RuntimeError):
    pass


class _HostMeshesStruct(C.Structure):
    _fields_ = [("verts", C.c_void_p), ("indices", C.c_void_p), ("n_verts", C.c_uint64), ("n_indices", C.c_uint64),
                ("arena", C.c_void_p)]


class HostMeshes:
    """Owner of one arena (include/voxels_hip.h vx_host_meshes): .verts / .indices are views into page-locked memory,
    valid until release() (or garbage collection of this object)."""

    def __init__(self, lib, m):
        self._lib, self._arena = lib, m.arena
        nv, ni = int(m.n_verts), int(m.n_indices)
        self.verts = (np.ctypeslib.as_array(C.cast(m.verts, C.POINTER(C.c_uint8)), (nv * VERTEX_DTYPE.itemsize,)).view(VERTEX_DTYPE)
                      if nv else np.zeros(0, VERTEX_DTYPE))
        self.indices = (np.ctypeslib.as_array(C.cast(m.indices, C.POINTER(C.c_uint32)), (ni,)) if ni else np.zeros(0, np.uint32))

    def _take(self):
        a, self._arena = self._arena, None
        self.verts = self.indices = None
        return a

    def release(self):
        a = self._take()
        if a:
            self._lib.vx_host_meshes_release(a)

    def __del__(self):
        try:
            self.release()
        except Exception:
            pass


class ExecInfo(C.Structure):
    _fields_ = [("levels", C.c_uint32), ("retries", C.c_uint32), ("device_ms", C.c_float),
                ("total_verts", C.c_uint64), ("total_indices", C.c_uint64),
                ("active_blocks", C.c_uint32 * 8), ("algorithmic_bytes", C.c_uint64),
                ("blocks_read", C.c_uint32), ("mirror_ms", C.c_float), ("first_meshed_level", C.c_uint32)]


def hip_library_path():
    """The in-tree HIP build.  VOXELS_HIP_LIBRARY names another build of the SAME library (A/B measurements of kernel
    variants compiled from the same sources with different -D switches, tools/ab_build.py) — never a different backend."""
    return os.environ.get("VOXELS_HIP_LIBRARY") or os.path.join(os.path.dirname(os.path.abspath(__file__)), "csrc", "libvoxels_hip.so")


def _ptr(a):
    return None if a is None else a.ctypes.data_as(C.c_void_p)


class HipLibrary:
    """Loads the C-ABI library.  The default is the in-tree HIP build; a missing library is an error."""

    def __init__(self, path=None):
        path = path or hip_library_path()
        if not os.path.exists(path):
            raise VoxelsHipError(
                "%s not found: build it with `python -c 'import __graft_entry__ as g; g.build()'` "
                "(hipcc --offload-arch=gfx950). There is no CPU fallback." % path)
        # torch wheels bundle their own HIP runtime; when torch is in the process it has to initialise before the
        # system runtime this library links (the other order leaves torch without a visible GPU)
        torch = sys.modules.get("torch")
        if torch is not None and torch.cuda.is_available():
            torch.cuda.init()
        lib = C.CDLL(path)
        vp, u32, i32 = C.c_void_p, C.c_uint32, C.c_int32
        lib.vx_backend.restype = C.c_char_p
        lib.vx_ctx_create.argtypes = [C.c_int, C.POINTER(vp)]
        lib.vx_ctx_destroy.argtypes = [vp]
        lib.vx_last_error.restype = C.c_char_p
        lib.vx_last_error.argtypes = [vp]
        lib.vx_set_stream.argtypes = [vp, vp]
        lib.vx_grid_upload.argtypes = [vp, u32, vp, vp, vp, vp]
        lib.vx_grid_upload_packed.argtypes = [vp, vp, C.c_uint64]
        lib.vx_grid_attach_y.argtypes = [vp, u32, u32, u32, vp, C.c_int32, u32, vp, vp, C.c_int32, u32, vp]
        lib.vx_device_meshes.argtypes = [vp, vp, vp, vp, vp]
        lib.vx_export_meshes.argtypes = [vp, vp]
        lib.vx_compact_pools.argtypes = [vp]
        lib.vx_grid_pack.argtypes = [vp, vp, C.c_uint64, vp]
        lib.vx_grid_create_heightmap.argtypes = [vp, u32, vp]
        lib.vx_grid_create_terrain.argtypes = [vp, u32, u32]
        lib.vx_grid_create_terrain_ex.argtypes = [vp, u32, u32, u32]
        lib.vx_grid_fill_terrain.argtypes = [vp, u32]
        lib.vx_grid_inject_ball.argtypes = [vp, vp, vp, C.c_float, C.c_int, vp, vp]
        lib.vx_grid_inject_material.argtypes = [vp, vp, vp, C.c_uint8, C.c_int, vp, vp]
        lib.vx_level_ranges.argtypes = [vp, u32, vp]
        lib.vx_device_block_table.argtypes = [vp, u32, vp, vp]
        lib.vx_comm_unique_id.argtypes = [vp]
        lib.vx_comm_init.argtypes = [vp, C.c_int, C.c_int, vp]
        lib.vx_comm_destroy.argtypes = [vp]
        lib.vx_halo_exchange.argtypes = [vp]
        lib.vx_halo_exchange_group.argtypes = [vp, C.c_int]
        lib.vx_grid_read_block.argtypes = [vp, u32, vp, vp, vp, vp]
        lib.vx_grid_attach.argtypes = [vp, u32, u32, u32, vp, i32, vp, vp, i32, vp]
        lib.vx_grid_update_blocks.argtypes = [vp, u32, vp, vp, vp, vp, vp]
        lib.vx_grid_invalidate.argtypes = [vp]
        lib.vx_ctx_forget_hints.argtypes = [vp]
        lib.vx_material_lut.argtypes = [vp, vp, vp]
        lib.vx_polygonize.argtypes = [vp, u32, C.POINTER(ExecInfo)]
        lib.vx_polygonize_from.argtypes = [vp, u32, u32, C.POINTER(ExecInfo)]
        lib.vx_polygonize_dirty.argtypes = [vp, vp, vp, C.POINTER(ExecInfo), vp, u32, C.POINTER(u32)]
        lib.vx_level_counts.argtypes = [vp, u32, C.POINTER(u32), vp]
        lib.vx_download_level.argtypes = [vp, u32, vp, vp, vp, vp, vp]
        lib.vx_host_meshes_acquire.argtypes = [vp, vp]
        lib.vx_host_meshes_release.argtypes = [vp]
        lib.vx_host_meshes_release.restype = None
        lib.vx_host_meshes_trim.argtypes = []
        lib.vx_host_meshes_trim.restype = None
        lib.vx_host_meshes_reserve.argtypes = [vp, C.c_uint64, C.c_uint64]
        lib.vx_stats.argtypes = [vp, vp]
        lib.vx_selftest.argtypes = [vp, vp]
        lib.vx_stage_layout.argtypes = [vp, C.POINTER(C.c_int)]
        lib.vx_set_stage_timing.argtypes = [vp, C.c_int]
        lib.vx_stage_times.argtypes = [vp, vp]
        self.lib = lib
        self.path = path
        self.backend = lib.vx_backend().decode()


class Level:
    """One LOD level: blocks in PolygonSurface::GetBlockForLevel order, arrays concatenated."""

    def __init__(self, infos, verts, idx, tverts, tidx):
        self.infos, self.verts, self.idx, self.tverts, self.tidx = infos, verts, idx, tverts, tidx

    def totals(self):
        return (len(self.infos), len(self.verts), len(self.idx), len(self.tverts), len(self.tidx))


class Polygonizer:
    """Host-side mirror of Voxels::Polygonizer for the device path.

    upload(dist, mat, blend, empty_flags)  ~ the Grid the reference's Execute reads
    execute(num_levels=0)                   ~ Polygonizer::Execute (full run); returns exec info
    level(l) / stats()                      ~ PolygonSurface accessors
    """

    def __init__(self, device=0, library=None):
        self._L = library or HipLibrary()
        self._lib = self._L.lib
        h = C.c_void_p()
        rc = self._lib.vx_ctx_create(int(device), C.byref(h))
        if rc != 0:
            raise VoxelsHipError("vx_ctx_create failed (%d): no usable HIP device?" % rc)
        self._h = h
        self.n = 0

    def close(self):
        if getattr(self, "_h", None):
            self._lib.vx_ctx_destroy(self._h)
            self._h = None

    def __del__(self):
        self.close()

    def _check(self, rc, what):
        if rc != 0:
            raise VoxelsHipError("%s failed (%d): %s" % (what, rc, self._lib.vx_last_error(self._h).decode()))

    @property
    def backend(self):
        return self._L.backend

    def set_stream(self, stream_handle):
        self._check(self._lib.vx_set_stream(self._h, C.c_void_p(stream_handle)), "vx_set_stream")

    def upload(self, dist, mat, blend, empty_flags):
        n = dist.shape[0]
        assert dist.shape == (n, n, n) and dist.dtype == np.int8 and dist.flags.c_contiguous
        for a in (mat, blend):
            assert a is None or (a.shape == (n, n, n) and a.dtype == np.uint8 and a.flags.c_contiguous)
        empty_flags = np.ascontiguousarray(empty_flags, np.uint8)
        assert empty_flags.size == (n // 16) ** 3
        self._keep = (dist, mat, blend, empty_flags)
        self._check(self._lib.vx_grid_upload(self._h, n, _ptr(dist), _ptr(mat), _ptr(blend), _ptr(empty_flags)), "vx_grid_upload")
        self.n = n

    def upload_packed(self, blob):
        """Grid file format v1 (Grid::PackForSave) straight to the device; expanded there."""
        blob = np.ascontiguousarray(np.frombuffer(blob, np.uint8) if not isinstance(blob, np.ndarray) else blob.view(np.uint8))
        self._check(self._lib.vx_grid_upload_packed(self._h, _ptr(blob), blob.size), "vx_grid_upload_packed")
        self.n = int(np.frombuffer(blob[4:8].tobytes(), np.uint32)[0])

    def create_heightmap(self, heightmap):
        """Grid::Create(w, heightmap) evaluated on the device; heightmap int8 [w, w] (row = y)."""
        hm = np.ascontiguousarray(heightmap, np.int8)
        assert hm.ndim == 2 and hm.shape[0] == hm.shape[1]
        self._check(self._lib.vx_grid_create_heightmap(self._h, hm.shape[0], _ptr(hm)), "vx_grid_create_heightmap")
        self.n = hm.shape[0]

    def pack(self):
        """Grid::PackForSave of the resident grid (encoded on the device) -> uint8 array."""
        size = C.c_uint64()
        self._check(self._lib.vx_grid_pack(self._h, None, 0, C.byref(size)), "vx_grid_pack")
        out = np.zeros(size.value, np.uint8)
        self._check(self._lib.vx_grid_pack(self._h, _ptr(out), out.size, C.byref(size)), "vx_grid_pack")
        return out

    def read_block(self, block_id):
        """(dist int8[16,16,16] (z,y,x), mat, blend, BF_Empty) of one resident block."""
        d = np.zeros((16, 16, 16), np.int8)
        m = np.zeros((16, 16, 16), np.uint8)
        b = np.zeros((16, 16, 16), np.uint8)
        f = np.zeros(1, np.uint8)
        self._check(self._lib.vx_grid_read_block(self._h, int(block_id), _ptr(d), _ptr(m), _ptr(b), _ptr(f)), "vx_grid_read_block")
        return d, m, b, int(f[0])

    def column(self, n, x, y):
        """The n distance samples of the voxel column (x, y) of the resident grid (all z), read block by block: what tools use
        to find the surface without generating the grid on the host a second time."""
        nb = n // 16
        return np.concatenate([self.read_block((bz * nb + y // 16) * nb + x // 16)[0][:, y % 16, x % 16] for bz in range(nb)])

    def inject_ball(self, pos, ext, radius, inj_type):
        """Grid::InjectSurface with the analytic ball brush, on the device; returns the modified box (output order)."""
        pos = np.ascontiguousarray(pos, np.float32); ext = np.ascontiguousarray(ext, np.float32)
        mn, mx = np.zeros(3, np.float32), np.zeros(3, np.float32)
        self._check(self._lib.vx_grid_inject_ball(self._h, _ptr(pos), _ptr(ext), C.c_float(radius), int(inj_type), _ptr(mn), _ptr(mx)), "vx_grid_inject_ball")
        return mn, mx

    def inject_material(self, pos, ext, material, add):
        pos = np.ascontiguousarray(pos, np.float32); ext = np.ascontiguousarray(ext, np.float32)
        mn, mx = np.zeros(3, np.float32), np.zeros(3, np.float32)
        self._check(self._lib.vx_grid_inject_material(self._h, _ptr(pos), _ptr(ext), int(material), int(bool(add)), _ptr(mn), _ptr(mx)), "vx_grid_inject_material")
        return mn, mx

    def compact_pools(self):
        self._check(self._lib.vx_compact_pools(self._h), "vx_compact_pools")

    def device_meshes(self):
        """(device pointer of the vertex pool, of the index pool, vertices, indices) of the last full run."""
        dv, di = C.c_void_p(), C.c_void_p()
        nv, ni = C.c_uint64(), C.c_uint64()
        self._check(self._lib.vx_device_meshes(self._h, C.byref(dv), C.byref(di), C.byref(nv), C.byref(ni)), "vx_device_meshes")
        return dv.value, di.value, nv.value, ni.value

    def export_meshes(self):
        """Inter-process handles of the two pools (vx_export_meshes): dict with the two 64-byte handles, the counts, the
        capacities and the pools' generation."""
        class IpcMeshes(C.Structure):
            _fields_ = [("verts_handle", C.c_uint8 * 64), ("indices_handle", C.c_uint8 * 64), ("n_verts", C.c_uint64), ("n_indices", C.c_uint64),
                        ("verts_capacity", C.c_uint64), ("indices_capacity", C.c_uint64), ("generation", C.c_uint64)]
        m = IpcMeshes()
        self._check(self._lib.vx_export_meshes(self._h, C.byref(m)), "vx_export_meshes")
        return {"verts_handle": bytes(m.verts_handle), "indices_handle": bytes(m.indices_handle), "n_verts": int(m.n_verts), "n_indices": int(m.n_indices),
                "verts_capacity": int(m.verts_capacity), "indices_capacity": int(m.indices_capacity), "generation": int(m.generation)}

    def create_terrain(self, n, seed=1337, style=0):
        """The synthetic noise terrain (voxels_amd.synth.terrain) generated on the device into a grid the context owns."""
        self._check(self._lib.vx_grid_create_terrain_ex(self._h, int(n), int(seed), int(style)), "vx_grid_create_terrain_ex")
        self.n = n

    def fill_terrain(self, seed=1337):
        """The same into the attached slab (own layers + halo) with the BF_Empty flags of the rank's own blocks."""
        self._check(self._lib.vx_grid_fill_terrain(self._h, int(seed)), "vx_grid_fill_terrain")

    def comm_unique_id(self):
        """128-byte RCCL id (rank 0 creates it, every rank passes it to comm_init)."""
        buf = np.zeros(128, np.uint8)
        rc = self._lib.vx_comm_unique_id(_ptr(buf))
        if rc != 0:
            raise VoxelsHipError("vx_comm_unique_id failed (%d): RCCL not available?" % rc)
        return buf

    def comm_init(self, nranks, rank, unique_id):
        uid = np.ascontiguousarray(unique_id, np.uint8)
        assert uid.size == 128
        self._check(self._lib.vx_comm_init(self._h, int(nranks), int(rank), _ptr(uid)), "vx_comm_init")

    def halo_exchange(self):
        """Halo of the attached slab over RCCL (queued on the context's stream, no host wait)."""
        self._check(self._lib.vx_halo_exchange(self._h), "vx_halo_exchange")

    @staticmethod
    def halo_exchange_group(polys):
        """The same between several contexts of this process (slab order)."""
        arr = (C.c_void_p * len(polys))(*[p._h for p in polys])
        rc = polys[0]._lib.vx_halo_exchange_group(arr, len(polys))
        if rc != 0:
            bad = next((p for p in polys if p._lib.vx_last_error(p._h)), polys[0])
            raise VoxelsHipError("vx_halo_exchange_group failed (%d): %s" % (rc, bad._lib.vx_last_error(bad._h).decode()))

    def device_block_table(self, lvl):
        """(device pointer, count) of the level's block table (vx_listed_block records, GetBlockForLevel order)."""
        tab, nb = C.c_void_p(), C.c_uint32()
        self._check(self._lib.vx_device_block_table(self._h, int(lvl), C.byref(tab), C.byref(nb)), "vx_device_block_table")
        return tab.value, nb.value

    def level_ranges(self, lvl):
        """Per block (download order): offsets of its meshes in the device pools."""
        nb = self.level(lvl, with_data=False).infos.size
        dt = np.dtype([("v_off", np.uint32), ("i_off", np.uint32), ("tv_off", np.uint32, 6), ("ti_off", np.uint32, 6)])
        r = np.zeros(nb, dt)
        self._check(self._lib.vx_level_ranges(self._h, int(lvl), _ptr(r)), "vx_level_ranges")
        return r

    def attach(self, n, z_begin, z_end, d_dist, dist_z0, d_mat, d_blend, mat_z0, d_flags):
        """Device pointers (ints), e.g. torch tensors' data_ptr().  The library mirrors the fields for its gathers: after
        rewriting the tensors in place call invalidate() (or attach again), or the next execute() sees the old contents."""
        self._check(self._lib.vx_grid_attach(self._h, n, z_begin, z_end, C.c_void_p(d_dist), dist_z0,
                                             C.c_void_p(d_mat), C.c_void_p(d_blend), mat_z0, C.c_void_p(d_flags)),
                    "vx_grid_attach")
        self.n = n

    def attach_y(self, n, y_begin, y_end, d_dist, dist_y0, dist_rows, d_mat, d_blend, mat_y0, mat_rows, d_flags):
        """Slab cut along y: device arrays [n][rows][n] (see vx_grid_attach_y)."""
        self._check(self._lib.vx_grid_attach_y(self._h, n, y_begin, y_end, C.c_void_p(d_dist), dist_y0, dist_rows,
                                               C.c_void_p(d_mat), C.c_void_p(d_blend), mat_y0, mat_rows, C.c_void_p(d_flags)),
                    "vx_grid_attach_y")
        self.n = n

    def invalidate(self):
        """The attached tensors were rewritten in place by the caller: the library's mirrors of them are rebuilt by the
        next execute().  Without this call (or a new attach) a run after an in-place edit polygonizes the OLD contents."""
        self._check(self._lib.vx_grid_invalidate(self._h), "vx_grid_invalidate")

    def forget_hints(self):
        """What earlier runs taught this context about its surfaces (capacity classes, launch sizes) is forgotten: the next run
        starts from a new context's conservative defaults (vx_ctx_forget_hints)."""
        self._check(self._lib.vx_ctx_forget_hints(self._h), "vx_ctx_forget_hints")

    def update_blocks(self, block_ids, dist, mat, blend, empty_flags):
        block_ids = np.ascontiguousarray(block_ids, np.uint32)
        self._check(self._lib.vx_grid_update_blocks(self._h, block_ids.size, _ptr(block_ids), _ptr(dist), _ptr(mat),
                                                    _ptr(blend), _ptr(np.ascontiguousarray(empty_flags, np.uint8))),
                    "vx_grid_update_blocks")

    def set_materials(self, lut, valid=None):
        lut = np.ascontiguousarray(lut, np.uint8)
        assert lut.shape == (256, 6)
        valid = None if valid is None else np.ascontiguousarray(valid, np.uint8)
        self._check(self._lib.vx_material_lut(self._h, _ptr(lut), _ptr(valid)), "vx_material_lut")

    def execute(self, num_levels=0):
        info = ExecInfo()
        self._check(self._lib.vx_polygonize(self._h, int(num_levels), C.byref(info)), "vx_polygonize")
        self.info = info
        return info

    def execute_from(self, num_levels, first_meshed_level):
        """vx_polygonize_from: caches and bitmaps of every level, meshes only from first_meshed_level up (info.first_meshed_level
        tells what the run really did)."""
        info = ExecInfo()
        self._check(self._lib.vx_polygonize_from(self._h, int(num_levels), int(first_meshed_level), C.byref(info)), "vx_polygonize_from")
        self.info = info
        return info

    def execute_dirty(self, min_corner, max_corner):
        """Polygonizer::Execute with a Modification: re-polygonize the dirty box (output, Y-up, coordinates).
        Returns the ids of the rebuilt blocks (Modification::GetModifiedBlocks)."""
        mn = np.ascontiguousarray(min_corner, np.float32)
        mx = np.ascontiguousarray(max_corner, np.float32)
        info = ExecInfo()
        cap = 1 << 20
        ids = getattr(self, "_dirty_ids", None)  # (kept: allocating and zeroing 4 MB per call cost more than a small incremental run)
        if ids is None:
            ids = self._dirty_ids = np.zeros(cap, np.uint32)
        cnt = C.c_uint32()
        self._check(self._lib.vx_polygonize_dirty(self._h, _ptr(mn), _ptr(mx), C.byref(info), _ptr(ids), cap, C.byref(cnt)),
                    "vx_polygonize_dirty")
        self.info = info
        return ids[:cnt.value].copy()

    def debug_header(self, count=352):
        """vx_debug_header: the device copy of the run's header words (diagnostics)"""
        out = np.zeros(count, np.uint32)
        self._check(self._lib.vx_debug_header(self._h, _ptr(out), int(count)), "vx_debug_header")
        return out

    def level(self, lvl, with_data=True):
        nb = C.c_uint32()
        tot = np.zeros(4, np.uint64)
        self._check(self._lib.vx_level_counts(self._h, lvl, C.byref(nb), _ptr(tot)), "vx_level_counts")
        infos = np.zeros(nb.value, BLOCK_INFO_DTYPE)
        if not with_data:
            self._check(self._lib.vx_download_level(self._h, lvl, _ptr(infos), None, None, None, None), "vx_download_level")
            return Level(infos, np.zeros(0, VERTEX_DTYPE), np.zeros(0, np.uint32), np.zeros(0, VERTEX_DTYPE), np.zeros(0, np.uint32))
        verts = np.zeros(int(tot[0]), VERTEX_DTYPE)
        idx = np.zeros(int(tot[1]), np.uint32)
        tverts = np.zeros(int(tot[2]), VERTEX_DTYPE)
        tidx = np.zeros(int(tot[3]), np.uint32)
        self._check(self._lib.vx_download_level(self._h, lvl, _ptr(infos), _ptr(verts), _ptr(idx), _ptr(tverts), _ptr(tidx)),
                    "vx_download_level")
        return Level(infos, verts, idx, tverts, tidx)

    def reserve_host_meshes(self, n_verts, n_indices):
        """vx_host_meshes_reserve: page-lock an arena of that size ahead of time (it waits in the recycling list)."""
        self._check(self._lib.vx_host_meshes_reserve(self._h, int(n_verts), int(n_indices)), "vx_host_meshes_reserve")

    def host_meshes(self, previous=None):
        """vx_host_meshes_acquire: both pools on the host (page-locked, one DMA), as numpy views.  Block k of level l owns
        verts[level_ranges(l)[k]['v_off'] : ... + level(l, False).infos[k]['n_verts']] etc.  `previous`: a HostMeshes of an
        earlier acquire on this context, brought up to date instead (incremental runs); do not use it afterwards."""
        m = _HostMeshesStruct()
        if previous is not None:
            m.arena = previous._take()
        rc = self._lib.vx_host_meshes_acquire(self._h, C.byref(m))
        if rc != 0 and m.arena:
            # the C side always writes the arena back (possibly a different one): nothing owns it on this path, so it goes
            # back to the library's recycling list instead of leaking page-locked memory (`previous` is spent either way)
            self._lib.vx_host_meshes_release(m.arena)
            m.arena = None
        self._check(rc, "vx_host_meshes_acquire")
        return HostMeshes(self._lib, m)

    def all_levels(self):
        return [self.level(l) for l in range(self.info.levels)]

    def set_stage_timing(self, enable):
        self._check(self._lib.vx_set_stage_timing(self._h, int(bool(enable))), "vx_set_stage_timing")


    def stage_layout(self):
        """vx_stage_layout: 1 if the last run with stage timing used the single-stream form (k_main), else 0"""
        v = C.c_int(0)
        self._check(self._lib.vx_stage_layout(self._h, C.byref(v)), "vx_stage_layout")
        return v.value

    def stage_times(self):
        """ms of (reset, classify, hierarchy, material, regular level 0, regular levels >= 1, transition, block lists) of the last run."""
        out = np.zeros(8, np.float32)
        self._check(self._lib.vx_stage_times(self._h, _ptr(out)), "vx_stage_times")
        return out

    def selftest(self):
        """vx_selftest: mismatch counts of the device arithmetic against its definition, exhaustively ([0..2], [11] must be 0)."""
        r = np.zeros(16, np.uint32)
        self._check(self._lib.vx_selftest(self._h, _ptr(r)), "vx_selftest")
        return r

    def stats(self):
        out = np.zeros(20, np.uint32)
        self._check(self._lib.vx_stats(self._h, _ptr(out)), "vx_stats")
        return out

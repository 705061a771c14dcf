#!/usr/bin/env python3
"""Generate tests/golden/*.npz from the UNMODIFIED reference (oracle/_ref/libvoxels_ref.so, built from
/root/reference/src by `make -C oracle ref`).  Run in the build container only; the outputs are committed.

Each fixture stores the exact input bytes (int8 distances, u8 material, u8 blend) and, per LOD level, the
reference's block infos, vertices, indices, transition vertices/indices and the statistics, so the tests
never need /root/reference at run time."""
import hashlib
import os
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.dirname(HERE))
import fields  # noqa: E402
import vxo  # noqa: E402


def dump(path, ref, grid, extra=None, surface=None):
    d, m, b = grid.read_dense()
    s = surface if surface is not None else ref.execute(grid)
    out = {"dist": d, "mat": m, "blend": b, "flags": grid.block_flags(), "stats": s.stats(),
           "levels": np.array([s.levels_count], np.uint32),
           "cache_bytes": np.array([s.cache_bytes()], np.uint64), "memory_size": np.array([grid.memory_size()], np.uint64)}
    for li, lv in enumerate(s.all_levels()):
        out["L%d_infos" % li] = lv.infos
        out["L%d_verts" % li] = lv.verts
        out["L%d_idx" % li] = lv.idx
        out["L%d_tverts" % li] = lv.tverts
        out["L%d_tidx" % li] = lv.tidx
    if extra:
        out.update(extra)
    np.savez_compressed(path, **out)
    print("wrote", path, os.path.getsize(path), "bytes")
    return s


def main():
    ref = vxo.load_ref()
    assert ref is not None, "build oracle/_ref first (make -C oracle ref)"
    # 1. the survey's known-answer case: 64^3 sphere through Grid::Create (float quantisation path)
    f = vxo.sphere_field(64)
    g = ref.grid_from_float(f)
    s = dump(os.path.join(HERE, "sphere64.npz"), ref, g)
    assert "%016x" % vxo.index_hash(s.all_levels()) == "473e8b8c4d4f3c9d"
    # 2. 32^3 terrain with materials (2 levels: no transitions since level 1 is the last)
    f = fields.terrain_field(32, 11)
    m, b = fields.materials_for(32, 11)
    dump(os.path.join(HERE, "terrain32_mat.npz"), ref, ref.grid_from_float(f, m, b))
    # 3. 64^3 full-range noise + materials (3 levels, transitions on level 1, wide t range)
    q = fields.quantize_full_range(fields.smooth_noise(64, 7, scale=16, amp=3.0))
    m, b = fields.materials_for(64, 7)
    dump(os.path.join(HERE, "noise64_fullrange_mat.npz"), ref, ref.grid_from_dense(q, m, b))
    # 4. 64^3 terrain + sphere carve + incremental re-polygonization (Modification)
    f = fields.terrain_field(64, 3)
    m, b = fields.materials_for(64, 3)
    g = ref.grid_from_float(f, m, b)
    s = ref.execute(g)
    d0, m0, b0 = g.read_dense()
    mn, mx = g.inject_ball((30.0, 33.5, 31.25), (20, 20, 20), 7.0, 2)
    ids = ref.execute_modify(g, s, mn, mx)
    dump(os.path.join(HERE, "terrain64_carve_modify.npz"), ref, g,
         extra={"pre_dist": d0, "pre_mat": m0, "pre_blend": b0, "box_min": mn, "box_max": mx, "modified_ids": ids,
                "inject_pos": np.array([30.0, 33.5, 31.25], np.float32), "inject_ext": np.array([20, 20, 20], np.float32),
                "inject_radius": np.array([7.0], np.float32)}, surface=s)
    # 5. float quantisation vectors (VoxelGrid.cpp:37-50): values -> bytes stored by Grid::Create
    rng = np.random.RandomState(5)
    v = np.concatenate([rng.uniform(-9, 9, 4096 - 32).astype(np.float32),
                        np.array([0, -0.0, 0.5, -0.5, 1, -1, 3.999, 4, 4.0001, -4, -4.0001, 126.5, 127, 127.5, 200,
                                  -127, -127.5, -128, 1e-8, -1e-8, 2.5, -2.5, 3, -3, 99.9, -99.9, 100, -100, 5, -5, 7.25, -7.25],
                                 np.float32)])
    fld = np.zeros((16, 16, 16), np.float32)
    fld.reshape(-1)[:] = v
    g = ref.grid_from_float(fld)
    np.savez_compressed(os.path.join(HERE, "quantize16.npz"), values=fld, dist=g.read_dense()[0])
    print("wrote quantize16.npz")


if __name__ == "__main__":
    main()

"""CPU tests of the product's core logic: the closed-form per-cell formulation (voxels_amd/csrc/tv_core.h), the
block phases (tv_block.h) and the host orchestration (vx_host.inl) — compiled with the CPU emulation backend
(tests/emu) — against the reference fixtures and the oracle port.  The HIP kernels run the same phase code; their
GPU-only parts (LDS staging, wave scans, atomics, bit-parallel classify) are covered by tests/test_gpu_parity.py."""
import numpy as np
import pytest

import fields
import vxo
from emu_lib import emu_library
from golden_io import Golden


@pytest.fixture(scope="module")
def emu():
    return emu_library()


@pytest.fixture(scope="module")
def port():
    import subprocess, os
    if not os.path.exists(vxo.PORT_SO):
        subprocess.check_call(["make", "-C", os.path.join(vxo.ROOT, "oracle"), "port"])
    return vxo.load_port()


def make_poly(emu):
    from voxels_amd.binding import Polygonizer
    p = Polygonizer(library=emu)
    p.set_materials(vxo.default_lut())
    return p


@pytest.mark.parametrize("name", ["sphere64", "terrain32_mat", "noise64_fullrange_mat"])
def test_emu_matches_reference_fixture(emu, name):
    gold = Golden(name)
    p = make_poly(emu)
    p.upload(gold.dist, gold.mat, gold.blend, gold.flags)
    p.execute()
    ok, msg = fields.surface_equal(p.all_levels(), gold.levels)
    assert ok, msg
    assert np.array_equal(p.stats(), gold.stats)


def check(emu, port, d, m, b, label):
    g = port.grid_from_dense(d, m, b)
    s = port.execute(g)
    p = make_poly(emu)
    p.upload(d, m, b, g.block_flags())
    p.execute()
    ok, msg = fields.surface_equal(p.all_levels(), s.all_levels())
    assert ok, label + ": " + msg
    assert np.array_equal(p.stats(), s.stats()), label


def test_emu_zero_heavy_and_white_noise(emu, port):
    rng = np.random.RandomState(0)
    for seed in range(3):
        n = 32
        d = np.clip(np.round(fields.smooth_noise(n, 100 + seed, scale=8, amp=2.0) * 1.5), -4, 4).astype(np.int8)
        m = rng.randint(0, 4, (n, n, n)).astype(np.uint8)
        b = rng.randint(0, 256, (n, n, n)).astype(np.uint8)
        check(emu, port, d, m, b, "zeros %d" % seed)
    d = rng.randint(-128, 128, (32, 32, 32)).astype(np.int8)
    check(emu, port, d, rng.randint(0, 3, (32, 32, 32)).astype(np.uint8), rng.randint(0, 256, (32, 32, 32)).astype(np.uint8), "white noise")


def test_emu_degenerate_filter_at_coarse_levels(emu, port):
    """Zero-heavy 128^3 field: degenerate triangles at cell sizes 1..8 (the suspect-cell shortcut must agree with the
    reference's test everywhere)."""
    n = 128
    d = np.clip(np.round(fields.smooth_noise(n, 77, scale=16, amp=2.5) * 1.2), -4, 4).astype(np.int8)
    zero = np.zeros((n, n, n), np.uint8)
    check(emu, port, d, zero, zero, "zeros 128")


@pytest.mark.parametrize("h", [15.5, 16.0, 31.5])
def test_emu_planes_and_emptiness_skip(emu, port, h):
    n = 64
    z = np.arange(n).reshape(n, 1, 1) * np.ones((n, n, n))
    d = np.ascontiguousarray(np.clip(np.sign(z - h) * np.ceil(np.abs(z - h)), -4, 4).astype(np.int8))
    zero = np.zeros((n, n, n), np.uint8)
    check(emu, port, d, zero, zero, "plane %g" % h)


def test_emu_synth_terrain_vs_port(emu, port):
    from voxels_amd import synth
    d, m, b = synth.terrain(64)
    g = port.grid_from_dense(d, m, b)
    assert np.array_equal(synth.block_empty_flags(d), g.block_flags())
    check(emu, port, d, m, b, "synth terrain 64")


def test_emu_coarse_block_without_cells(emu, port):
    """A bubble smaller than a level-1 cell makes its level-1 (and level-2) block surface-bearing (it contains
    level-0 cells) although no coarse cell is non-trivial: such blocks must yield empty records, not stale data."""
    n = 64
    d = np.full((n, n, n), 4, np.int8)
    d[21, 21, 21] = -3            # odd coordinates: invisible to every coarser sampling
    zero = np.zeros((n, n, n), np.uint8)
    check(emu, port, d, zero, zero, "bubble")


def test_emu_level_limit(emu):
    gold = Golden("noise64_fullrange_mat")
    p = make_poly(emu)
    p.upload(gold.dist, gold.mat, gold.blend, gold.flags)
    p.execute(2)
    lv = p.all_levels()
    assert len(lv) == 2
    ok, msg = fields.surface_equal(lv, gold.levels[:2])
    assert ok, msg


def test_emu_pool_overflow_retry(emu):
    """Output pools start from a size guess; a denser surface must trigger the transparent grow-and-rerun."""
    rng = np.random.RandomState(3)
    n = 32
    d = rng.randint(-128, 128, (n, n, n)).astype(np.int8)
    zero = np.zeros((n, n, n), np.uint8)
    p = make_poly(emu)
    port = vxo.load_port()
    g = port.grid_from_dense(d, zero, zero)
    p.upload(d, zero, zero, g.block_flags())
    info = p.execute()
    assert info.retries >= 1
    ok, msg = fields.surface_equal(p.all_levels(), port.execute(g).all_levels())
    assert ok, msg


def test_emu_z_slabs_union_equals_whole(emu):
    """Two contexts, each owning half the grid in z with the halo planes the path needs (1 below, 2 above for
    distances; 1 above for material/blend): the union of their blocks is the single-context result."""
    from voxels_amd import synth
    from voxels_amd.slab import merge_rank_levels
    n, levels = 128, 3           # coarsest block = 64 voxels -> two slabs of 64
    d, m, b = synth.terrain(n, seed=5)
    flags = synth.block_empty_flags(d)
    whole = make_poly(emu)
    whole.upload(d, m, b, flags)
    whole.execute(levels)
    ref_levels, ref_stats = whole.all_levels(), whole.stats()
    parts, stats = [], np.zeros(20, np.uint64)
    keep = []
    for r in range(2):
        z0, z1 = r * 64, (r + 1) * 64
        lo, hi = max(z0 - 1, 0), min(z1 + 2, n)
        dd = np.ascontiguousarray(d[lo:hi])
        mm = np.ascontiguousarray(m[z0:min(z1 + 1, n)])
        bb = np.ascontiguousarray(b[z0:min(z1 + 1, n)])
        keep.append((dd, mm, bb))
        p = make_poly(emu)
        p.attach(n, z0, z1, dd.ctypes.data, lo, mm.ctypes.data, bb.ctypes.data, z0, flags.ctypes.data)
        p.execute(levels)
        parts.append(p.all_levels())
        stats += p.stats()
    ok, msg = fields.surface_equal(merge_rank_levels(parts), ref_levels)
    assert ok, msg
    assert np.array_equal(stats.astype(np.uint32), ref_stats)


def test_emu_y_slabs_union_equals_whole(emu):
    """The same with slabs cut along y (rows of every z-plane): what bench.py uses for terrains, whose surface sits in a few
    z-layers.  Arrays are [n][rows][n]; the ranks' blocks interleave in block-id order."""
    from voxels_amd import synth
    from voxels_amd.slab import merge_rank_levels
    n, levels = 128, 3
    d, m, b = synth.terrain(n, seed=6)
    flags = synth.block_empty_flags(d)
    whole = make_poly(emu)
    whole.upload(d, m, b, flags)
    whole.execute(levels)
    ref_levels, ref_stats = whole.all_levels(), whole.stats()
    parts, stats, keep = [], np.zeros(20, np.uint64), []
    for r in range(2):
        y0, y1 = r * 64, (r + 1) * 64
        lo, hi = max(y0 - 1, 0), min(y1 + 2, n)
        hm = min(y1 + 1, n)
        dd = np.ascontiguousarray(d[:, lo:hi])
        mm = np.ascontiguousarray(m[:, y0:hm])
        bb = np.ascontiguousarray(b[:, y0:hm])
        keep.append((dd, mm, bb))
        p = make_poly(emu)
        p.attach_y(n, y0, y1, dd.ctypes.data, lo, hi - lo, mm.ctypes.data, bb.ctypes.data, y0, hm - y0, flags.ctypes.data)
        p.execute(levels)
        parts.append(p.all_levels())
        stats += p.stats()
    ok, msg = fields.surface_equal(merge_rank_levels(parts), ref_levels)
    assert ok, msg
    assert np.array_equal(stats.astype(np.uint32), ref_stats)


def test_emu_carve_modify_matches_reference_fixture(emu):
    """Config 5 in small: full run, sphere carve (Grid::InjectSurface result taken from the fixture), incremental
    re-polygonization of the dirty box — against the reference's own Modification run."""
    gold = Golden("terrain64_carve_modify")
    p = make_poly(emu)
    port = vxo.load_port()
    pre = (gold["pre_dist"], gold["pre_mat"], gold["pre_blend"])
    p.upload(*pre, port.grid_from_dense(*pre).block_flags())
    p.execute()
    ids, (d, m, b) = fields.edited_blocks(pre, (gold.dist, gold.mat, gold.blend))
    p.update_blocks(ids, d.view(np.int8), m, b, gold.flags)
    mod = p.execute_dirty(gold["box_min"], gold["box_max"])
    assert np.array_equal(mod, gold["modified_ids"])
    ok, msg = fields.surface_equal(p.all_levels(), gold.levels)
    assert ok, msg
    assert np.array_equal(p.stats(), gold.stats)


def test_emu_repeated_edits_vs_port(emu, port):
    """Several edits in a row (carve, add, carve at a grid corner): caches persist across incremental runs."""
    n = 64
    f = fields.terrain_field(n, 9)
    m, b = fields.materials_for(n, 9)
    g = port.grid_from_float(f, m, b)
    s = port.execute(g)
    p = make_poly(emu)
    pre = g.read_dense()
    p.upload(*pre, g.block_flags())
    p.execute()
    hm = p.host_meshes()
    fields.check_host_meshes(p, hm)
    for t, pos, ext, r in ((2, (30.0, 33.5, 31.25), (20, 20, 20), 7.0), (0, (40, 20, 25), (16, 16, 16), 6.0),
                           (2, (3.0, 60.0, 30.0), (12, 12, 12), 5.0), (2, (31.0, 33.0, 31.0), (10, 10, 10), 4.0)):
        mn, mx = g.inject_ball(pos, ext, r, t)
        ref_ids = port.execute_modify(g, s, mn, mx)
        post = g.read_dense()
        ids, (dd, mm, bb) = fields.edited_blocks(pre, post)
        p.update_blocks(ids, dd.view(np.int8), mm, bb, g.block_flags())
        pre = post
        got = p.execute_dirty(mn, mx)
        assert np.array_equal(got, ref_ids)
        ok, msg = fields.surface_equal(p.all_levels(), s.all_levels())
        assert ok, msg
        assert np.array_equal(p.stats(), s.stats())
        hm = p.host_meshes(previous=hm)  # only what the run appended travels
        fields.check_host_meshes(p, hm)
    # compaction rewrites the pools: an arena of the old layout is refilled as a whole, not patched
    p.compact_pools()
    hm = p.host_meshes(previous=hm)
    fields.check_host_meshes(p, hm)
    ok, msg = fields.surface_equal(p.all_levels(), s.all_levels())
    assert ok, msg
    # an arena outlives the surface it was filled from (and the context); a released one is handed out again
    keep = hm.verts.copy()
    p.execute()
    assert np.array_equal(hm.verts, keep)
    p._lib.vx_host_meshes_trim()  # (earlier tests left arenas in the recycling list)
    fresh = p.host_meshes()
    fields.check_host_meshes(p, fresh)
    addr = fresh.verts.ctypes.data
    fresh.release()
    again = [p.host_meshes(), p.host_meshes()]  # (the first may be the copy vx_download_level made for check_host_meshes)
    assert addr in [a.verts.ctypes.data for a in again]
    for a in again:
        fields.check_host_meshes(p, a)
        a.release()
    # an arena reserved ahead of time (InitializeVoxels with VOXELS_PREWARM_MB does this) is the one the next acquire of a
    # fitting size is handed: no page-locking inside the call
    p._lib.vx_host_meshes_trim()
    p.reserve_host_meshes(int(p.info.total_verts) + 1000, int(p.info.total_indices) + 1000)
    first = p.host_meshes()
    assert first.verts.size == int(p.info.total_verts) and first.indices.size == int(p.info.total_indices)
    fields.check_host_meshes(p, first)
    first.release()


def test_emu_polygonize_from_falls_back_to_a_full_run(emu, port):
    """vx_polygonize_from on a backend without the partial form (the emulation; on the GPU: dense surfaces, stage timing):
    every level is meshed and the call says so - callers (libVoxels.so's multi-device Execute) rely on that answer."""
    gold = Golden("terrain32_mat")
    p = make_poly(emu)
    p.upload(gold.dist, gold.mat, gold.blend, gold.flags)
    info = p.execute_from(0, 1)
    assert info.first_meshed_level == 0
    ok, msg = fields.surface_equal(p.all_levels(), gold.levels)
    assert ok, msg
    assert np.array_equal(p.stats(), gold.stats)


def test_emu_transition_face_batches(emu, port):
    """White noise at 64^3: every level-1 block has three neighbour faces whose 3 x 256 transition cells are all
    non-trivial — more than the 512-cell transition state holds, so the faces go through in several batches."""
    rng = np.random.RandomState(7)
    n = 64
    d = rng.randint(-128, 128, (n, n, n)).astype(np.int8)
    check(emu, port, d, rng.randint(0, 3, (n, n, n)).astype(np.uint8), rng.randint(0, 256, (n, n, n)).astype(np.uint8), "white noise 64")


def _packed_case(n, seed, noisy):
    rng = np.random.RandomState(seed)
    if noisy:
        d = rng.randint(-128, 128, (n, n, n)).astype(np.int8)          # incompressible: raw distance streams
        d[: n // 2] = np.clip(d[: n // 2], 3, 9)                          # ... next to compressible ones
    else:
        f = fields.terrain_field(n, seed)
        d = fields.quantize_full_range(f, scale=2.0)
    m, b = fields.materials_for(n, seed)
    return np.ascontiguousarray(d), m, b


def check_packed(poly_factory, port, d, m, b, label):
    """Grid file format v1 written by the reference (PackForSave), expanded on the device (or its CPU emulation):
    every block must come back exactly, and the surface must be the reference's."""
    n = d.shape[0]
    g = port.grid_from_dense(d, m, b)
    blob = g.pack()
    p = poly_factory()
    p.upload_packed(blob)
    flags = g.block_flags()
    nb = n // 16
    ids = np.unique(np.concatenate([np.arange(min(nb ** 3, 40)), np.random.RandomState(1).randint(0, nb ** 3, 60)]))
    for bid in ids:
        bx, by, bz = bid % nb, (bid // nb) % nb, bid // (nb * nb)
        sl = (slice(bz * 16, bz * 16 + 16), slice(by * 16, by * 16 + 16), slice(bx * 16, bx * 16 + 16))
        bd, bm, bb, fl = p.read_block(bid)
        assert np.array_equal(bd, d[sl]) and np.array_equal(bm, m[sl]) and np.array_equal(bb, b[sl]), "%s: block %d differs" % (label, bid)
        assert fl == flags[bid], "%s: BF_Empty of block %d" % (label, bid)
    p.execute()
    s = port.execute(g)
    ok, msg = fields.surface_equal(p.all_levels(), s.all_levels())
    assert ok, label + ": " + msg
    assert np.array_equal(p.stats(), s.stats()), label
    return len(blob)


@pytest.mark.parametrize("noisy", [False, True])
def test_emu_upload_packed_grid(emu, port, noisy):
    d, m, b = _packed_case(64, 21, noisy)
    check_packed(lambda: make_poly(emu), port, d, m, b, "packed noisy=%s" % noisy)


def test_emu_upload_packed_rejects_garbage(emu):
    p = make_poly(emu)
    with pytest.raises(Exception):
        p.upload_packed(np.zeros(64, np.uint8))
    hdr = np.array([1, 32, 32, 32], np.uint32).view(np.uint8)
    with pytest.raises(Exception):
        p.upload_packed(np.concatenate([hdr, np.zeros(10, np.uint8)]))  # size table cut short


def test_emu_device_resident_meshes(emu):
    """vx_device_meshes + vx_level_ranges describe exactly what vx_download_level copies (with the emulation backend
    the "device" pools are host memory, so they can be read in place)."""
    import ctypes as C
    gold = Golden("noise64_fullrange_mat")
    p = make_poly(emu)
    p.upload(gold.dist, gold.mat, gold.blend, gold.flags)
    p.execute()
    dv, di, nv, ni = p.device_meshes()
    vdt = p.level(0).verts.dtype
    verts = np.frombuffer((C.c_char * (nv * 48)).from_address(dv), vdt)
    idx = np.frombuffer((C.c_char * (ni * 4)).from_address(di), np.uint32)
    for l in range(3):
        lv, rg = p.level(l), p.level_ranges(l)
        ov = oi = otv = oti = 0
        for k, info in enumerate(lv.infos):
            assert np.array_equal(verts[rg["v_off"][k]:rg["v_off"][k] + info["n_verts"]], lv.verts[ov:ov + info["n_verts"]])
            assert np.array_equal(idx[rg["i_off"][k]:rg["i_off"][k] + info["n_idx"]], lv.idx[oi:oi + info["n_idx"]])
            ov += info["n_verts"]; oi += info["n_idx"]
            for f in range(6):
                a, c = info["n_tverts"][f], info["n_tidx"][f]
                assert np.array_equal(verts[rg["tv_off"][k][f]:rg["tv_off"][k][f] + a], lv.tverts[otv:otv + a])
                assert np.array_equal(idx[rg["ti_off"][k][f]:rg["ti_off"][k][f] + c], lv.tidx[oti:oti + c])
                otv += a; oti += c


def check_device_edits(p, port, n, seed, surface_tol=0.0):
    """§8(f) row 2: a chain of ball / material edits applied to the resident grid by the library itself, compared after
    every edit with the reference doing the same Grid::InjectSurface / InjectMaterial + incremental Execute."""
    from voxels_amd import synth
    d0, m0, b0 = synth.terrain(n, seed=seed)
    g = port.grid_from_dense(d0, m0, b0)
    s = port.execute(g)
    p.upload_packed(g.pack())
    p.execute()
    c = n / 2.0
    nb = n // 16
    edits = (("ball", 2, (c - 2.0, c + 1.5, c - 0.75), (20, 20, 20), 7.0), ("ball", 0, (c + 8, c - 12, c - 7), (16, 16, 16), 6.0),
             ("mat", 3, (c - 3.0, c + 2.0, c - 2.0), (14, 14, 14), 1), ("ball", 2, (3.0, n - 4.0, c - 2), (12, 12, 12), 5.0),
             ("ball", 1, (c + 0.5, c + 0.25, c - 4.5), (9, 9, 9), 3.5), ("mat", 3, (c - 1.0, c + 1.0, c - 1.0), (10, 10, 10), 0),
             # non-cubic extents at a fractional position: the brush's row loop makes 6 trips, the grid's 5 (float rounding),
             # so the reference pairs voxels with shifted samples (found by tools/fuzz_parity.py)
             ("ball", 0, (37.4, 7.1, 51.3), (11.7, 5.0, 20.8), 2.4210112751143384))
    for kind, a, pos, ext, r in edits:
        pre = g.read_dense()
        if kind == "ball":
            mn, mx = g.inject_ball(pos, ext, r, a)
            mn2, mx2 = p.inject_ball(pos, ext, r, a)
        else:
            mn, mx = g.inject_material(pos, ext, a, bool(r))
            mn2, mx2 = p.inject_material(pos, ext, a, bool(r))
        assert np.array_equal(mn, mn2) and np.array_equal(mx, mx2)
        post = g.read_dense()
        ids, _ = fields.edited_blocks(pre, post)
        assert ids.size or (pos[0] == 37.4), "the edit must change something"  # the fixed-position regression case may miss the surface of a big grid
        flags = g.block_flags()
        # every block around the edit: data and BF_Empty
        lo = np.maximum(np.floor((np.array(pos) - np.array(ext)) / 16).astype(int) - 1, 0)
        hi = np.minimum(np.ceil((np.array(pos) + np.array(ext)) / 16).astype(int) + 1, nb)
        for bz in range(lo[2], hi[2]):
            for by in range(lo[1], hi[1]):
                for bx in range(lo[0], hi[0]):
                    bid = (bz * nb + by) * nb + bx
                    sl = (slice(bz * 16, bz * 16 + 16), slice(by * 16, by * 16 + 16), slice(bx * 16, bx * 16 + 16))
                    bd, bm, bb, fl = p.read_block(bid)
                    assert np.array_equal(bd, post[0][sl]), "distances of block %d after %s" % (bid, kind)
                    assert np.array_equal(bm, post[1][sl]) and np.array_equal(bb, post[2][sl]), "materials of block %d after %s" % (bid, kind)
                    assert fl == flags[bid], "BF_Empty of block %d after %s" % (bid, kind)
        ref_ids = port.execute_modify(g, s, mn, mx)
        got = p.execute_dirty(mn2, mx2)
        assert np.array_equal(got, ref_ids)
        ok, msg = fields.surface_equal(p.all_levels(), s.all_levels(), nrm_tol=surface_tol)
        assert ok, msg
        assert np.array_equal(p.stats(), s.stats())


def test_emu_edit_sequence_with_tight_pools(emu, port, monkeypatch):
    """VX_POOL_SLACK=64: every pool rule meets its limit within a few edits - the first full run grows its pools, the first
    incremental run makes room, later ones do not fit and pack the pools (a third dead) or grow them, and a call that finds more
    than half dead packs first.  Every step: the oracle's ids, statistics and surface; the host copy follows each new layout."""
    monkeypatch.setenv("VX_POOL_SLACK", "64")
    n = 64
    f = fields.terrain_field(n, 21)
    m, b = fields.materials_for(n, 21)
    g = port.grid_from_float(f, m, b)
    s = port.execute(g)
    p = make_poly(emu)
    p.upload_packed(g.pack())
    info = p.execute()
    assert info.retries >= 1  # (64 vertices of pool for a surface of thousands)
    ok, msg = fields.surface_equal(p.all_levels(), s.all_levels())
    assert ok, msg
    rng = np.random.RandomState(5)
    sizes = []
    for k in range(24):
        pos = tuple(float(x) for x in rng.uniform(14, n - 14, 3).round(2))
        args = (pos, (16.0, 16.0, 16.0), float(rng.uniform(3, 7)), 2 if k % 3 else 0)
        mn, mx = g.inject_ball(*args)
        mn2, mx2 = p.inject_ball(*args)
        assert np.array_equal(mn, mn2) and np.array_equal(mx, mx2)
        ref_ids = port.execute_modify(g, s, mn, mx)
        got = p.execute_dirty(mn, mx)
        assert np.array_equal(got, ref_ids), k
        ok, msg = fields.surface_equal(p.all_levels(), s.all_levels())
        assert ok, "edit %d: %s" % (k, msg)
        assert np.array_equal(p.stats(), s.stats())
        sizes.append(int(p.info.total_verts))
    assert any(b < a for a, b in zip(sizes, sizes[1:])), "the pools were never packed: %s" % sizes
    assert np.array_equal(p.pack(), g.pack())


def test_emu_device_edits(emu, port):
    check_device_edits(make_poly(emu), port, 64, 23)


def check_brushes_anywhere(p, port):
    """vx_grid_inject_ball / vx_grid_inject_material find the touched blocks as a box (three per-axis scans instead of the
    reference's walk over every block, src/VoxelGrid.cpp:388-584; the ids are written by k_box_ids): brushes inside, across the
    grid's sides, outside it, of zero extent, on block boundaries - the grid afterwards is the reference's, byte for byte, and
    so is the box handed back."""
    n = 48
    rng = np.random.RandomState(77)
    f = fields.terrain_field(n, 3)
    m, b = fields.materials_for(n, 3)
    g = port.grid_from_float(f, m, b)
    p.upload_packed(g.pack())
    cases = [((16.0, 16.0, 16.0), (16.0, 16.0, 16.0), 5.0), ((0.0, 0.0, 0.0), (10.0, 10.0, 10.0), 6.0), ((float(n), float(n), float(n)), (8.0, 8.0, 8.0), 5.0),
             ((24.0, 24.0, 24.0), (0.0, 0.0, 0.0), 3.0), ((-30.0, 20.0, 20.0), (10.0, 10.0, 10.0), 4.0), ((20.0, 200.0, 20.0), (12.0, 12.0, 12.0), 4.0),
             ((31.999, 32.0, 32.001), (0.001, 16.0, 15.999), 7.0), ((47.5, 0.5, 23.0), (3.0, 3.0, 40.0), 9.0)]
    for _ in range(40):
        pos = tuple(float(x) for x in rng.uniform(-10, n + 10, 3).round(3))
        ext = tuple(float(x) for x in rng.choice([0.5, 3.0, 8.0, 17.0, 40.0], 3))
        cases.append((pos, ext, float(rng.uniform(1, 12))))
    for k, (pos, ext, r) in enumerate(cases):
        t = (2, 0, 1)[k % 3]
        mn, mx = g.inject_ball(pos, ext, r, t)
        mn2, mx2 = p.inject_ball(pos, ext, r, t)
        assert np.array_equal(mn, mn2) and np.array_equal(mx, mx2), (k, pos, ext)
        if k % 4 == 3:
            a = g.inject_material(pos, ext, 7 + k % 5, k % 8 < 4)
            c = p.inject_material(pos, ext, 7 + k % 5, k % 8 < 4)
            assert np.array_equal(a[0], c[0]) and np.array_equal(a[1], c[1]), (k, pos, ext)
        assert np.array_equal(p.pack(), g.pack()), "brush %d at %s, extent %s" % (k, pos, ext)


def test_emu_brushes_anywhere_touch_the_reference_s_blocks(emu, port):
    check_brushes_anywhere(make_poly(emu), port)


def check_compaction(p, port, n):
    """Pools after a chain of edits hold dead ranges; vx_compact_pools packs the live meshes into fresh pools without
    changing anything a caller can download, and further incremental runs keep working."""
    from voxels_amd import synth
    d0, m0, b0 = synth.terrain(n, seed=31)
    g = port.grid_from_dense(d0, m0, b0)
    s = port.execute(g)
    p.upload_packed(g.pack())
    p.execute()
    c = n / 2.0
    for i, (pos, r) in enumerate((((c, c, c), 6.0), ((c + 5, c - 3, c + 2), 5.0), ((c - 6, c + 4, c - 1), 7.0))):
        mn, mx = g.inject_ball(pos, (2 * r + 4,) * 3, r, 2)
        p.inject_ball(pos, (2 * r + 4,) * 3, r, 2)
        port.execute_modify(g, s, mn, mx)
        p.execute_dirty(mn, mx)
        before = p.device_meshes()
        p.compact_pools()
        after = p.device_meshes()
        assert after[2] <= before[2] and after[3] <= before[3]
        live_v = sum(int(p.level(l, with_data=False).infos["n_verts"].sum() + p.level(l, with_data=False).infos["n_tverts"].sum()) for l in range(len(s.all_levels())))
        assert after[2] == live_v, "compaction leaves exactly the live vertices"
        ok, msg = fields.surface_equal(p.all_levels(), s.all_levels(), nrm_tol=1e-5)
        assert ok, "after compaction %d: %s" % (i, msg)


def test_emu_pool_compaction(emu, port):
    check_compaction(make_poly(emu), port, 64)


def check_pack(p, port, n, seed, noisy):
    """vx_grid_pack (device RLE encode) writes byte for byte the file the reference's PackForSave writes — for a
    grid uploaded dense, after a round trip through the packed upload, and after edits made on the device."""
    d, m, b = _packed_case(n, seed, noisy)
    g = port.grid_from_dense(d, m, b)
    want = g.pack()
    p.upload(d, m, b, g.block_flags())
    got = p.pack()
    assert got.size == want.size and np.array_equal(got, want), "dense upload -> pack"
    p.upload_packed(want)
    assert np.array_equal(p.pack(), want), "packed upload -> pack"
    c = n / 2.0
    g.inject_ball((c, c - 1.5, c + 2.0), (18, 18, 18), 6.0, 2)
    p.inject_ball((c, c - 1.5, c + 2.0), (18, 18, 18), 6.0, 2)
    g.inject_material((c + 1, c, c - 2.0), (12, 12, 12), 2, True)
    p.inject_material((c + 1, c, c - 2.0), (12, 12, 12), 2, True)
    assert np.array_equal(p.pack(), g.pack()), "after device edits"


@pytest.mark.parametrize("noisy", [False, True])
def test_emu_grid_pack(emu, port, noisy):
    check_pack(make_poly(emu), port, 64, 41, noisy)


def check_heightmap(p, port, n, seed, nrm_tol=0.0):
    """§8(f) row 3: Grid::Create(w, heightmap) evaluated on the device — file bytes (= voxel data + codec state) and the
    polygonized surface equal the reference's for the same height map."""
    rng = np.random.RandomState(seed)
    base = fields.smooth_noise(n, seed, scale=max(4, n // 4), amp=0.2 * n, octaves=3)[0]
    hm = np.clip(np.round(base + rng.uniform(-1, 1, (n, n))) + (n // 2 - 127), -128, 127).astype(np.int8)
    g = port.grid_from_heightmap(n, hm)
    p.create_heightmap(hm)
    assert np.array_equal(p.pack(), g.pack()), "the generated grid written as a file"
    flags = g.block_flags()
    for bid in (0, 1, (n // 16) ** 3 // 2, (n // 16) ** 3 - 1):
        assert p.read_block(bid)[3] == flags[bid]
    p.execute()
    s = port.execute(g)
    ok, msg = fields.surface_equal(p.all_levels(), s.all_levels(), nrm_tol=nrm_tol)
    assert ok, msg
    assert np.array_equal(p.stats(), s.stats())


def test_emu_create_heightmap(emu, port):
    check_heightmap(make_poly(emu), port, 64, 3)


def test_emu_rejects_misaligned_slabs_and_oversized_grids(emu):
    """a slab whose ORIGIN is not a multiple of the coarsest block would make the coarser levels read planes
    outside slab + halo; grids beyond 2048 would overrun the level tables."""
    from voxels_amd.binding import VoxelsHipError
    n = 64
    d = np.zeros((n, n, n), np.int8)
    m = np.zeros((n, n, n), np.uint8)
    flags = np.zeros((n // 16) ** 3, np.uint8)
    p = make_poly(emu)
    p.attach(n, 16, 48, d.ctypes.data, 0, m.ctypes.data, m.ctypes.data, 0, flags.ctypes.data)  # 32 thick, origin 16
    with pytest.raises(VoxelsHipError):
        p.execute(2)                                                                          # coarsest block = 32
    p.execute(1)                                                                              # level 0 alone is fine
    q = make_poly(emu)
    q.attach(n, 32, 64, d.ctypes.data, 0, m.ctypes.data, m.ctypes.data, 0, flags.ctypes.data)
    q.execute(2)
    with pytest.raises(VoxelsHipError):
        make_poly(emu).attach(4096, 0, 4096, d.ctypes.data, 0, m.ctypes.data, m.ctypes.data, 0, flags.ctypes.data)


@pytest.mark.parametrize("world,axis", [(2, "z"), (2, "y"), (4, "y")])
def test_emu_halo_exchange_group(emu, world, axis):
    """vx_halo_exchange_group (the C-ABI halo exchange with in-process transport): region logic + piece descriptors on the CPU."""
    import torch
    from voxels_amd import synth
    n, levels = 128, 3 if world == 2 else 2
    d, m, b = synth.terrain(n, 0, n, 7)
    whole = make_poly(emu)
    whole.upload(d, m, b, synth.block_empty_flags(d))
    whole.execute(levels)

    def mk():
        p = make_poly(emu)
        p.set_materials(vxo.default_lut())
        return p
    whole.set_materials(vxo.default_lut())
    whole.execute(levels)
    fields.check_halo_exchange_group(mk, torch, torch.device("cpu"), n, levels, world, axis, whole.all_levels())


def test_emu_device_terrain(emu):
    import torch
    fields.check_device_terrain(lambda: make_poly(emu), torch, torch.device("cpu"), 64, seed=11)


def test_emu_caves_style_matches_host_generator_and_oracle(emu):
    """The "caves" style of the synthetic generator (bench.py's second workload): device path of the emulation = host
    generator byte for byte, surface = the oracle's."""
    from voxels_amd import synth
    n = 64
    d, m, b = synth.terrain(n, style=1)
    dev = make_poly(emu)
    dev.create_terrain(n, 1337, 1)
    host = make_poly(emu)
    host.upload(d, m, b, synth.block_empty_flags(d))
    assert np.array_equal(dev.pack(), host.pack())
    # (Polygonizer.column: a voxel column of the resident grid, what bench.py and the tools find the surface height with)
    assert np.array_equal(dev.column(n, 37, 21), d[:, 21, 37]) and np.array_equal(dev.column(n, 0, 63), d[:, 63, 0])
    port = vxo.load_port()
    s = port.execute(port.grid_from_dense(d, m, b))
    dev.execute()
    ok, msg = fields.surface_equal(dev.all_levels(), s.all_levels())
    assert ok, msg



def test_emu_empty_surface_host_meshes(emu):
    """A grid without any surface: no blocks, empty pools, and the one-step host copy hands out empty arrays."""
    n = 32
    d = np.full((n, n, n), 50, np.int8)
    z = np.zeros((n, n, n), np.uint8)
    p = make_poly(emu)
    p.upload(d, z, z, np.ones((n // 16) ** 3, np.uint8))
    info = p.execute()
    hm = p.host_meshes()
    assert hm.verts.size == 0 and hm.indices.size == 0
    assert all(p.level(l).totals() == (0, 0, 0, 0, 0) for l in range(info.levels))
    hm.release()

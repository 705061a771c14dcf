"""GPU parity tests (run with -m gpu on an MI355X): the HIP path, through the C ABI (include/voxels_hip.h), against
  * the committed fixtures generated from the unmodified reference (tests/golden),
  * the oracle port (oracle/libvoxels_port.so) and, when present, the unmodified reference (oracle/_ref) on the
    same seeded inputs,
  * size-independent properties at sizes the CPU checkers do not finish quickly.
Bar: indices, block infos, positions, secondary positions, texture bytes and statistics bit-exact; normals within
1e-5 (fp32)."""
import os
import sys

import numpy as np
import pytest

import fields
import vxo
from golden_io import Golden

pytestmark = pytest.mark.gpu

NRM_TOL = 1e-5


@pytest.fixture(scope="module")
def poly():
    # torch's bundled HIP runtime must initialise before libvoxels_hip.so pulls in the system one (same order as
    # bench.py), or torch cannot see the GPU later in this process (the slab tests use torch device tensors)
    import torch
    torch.cuda.init()
    from voxels_amd import Polygonizer
    p = Polygonizer(device=0)
    assert p.backend == "hip:gfx950", "the native HIP library must be the one running"
    p.set_materials(vxo.default_lut())
    yield p
    p.close()


@pytest.fixture(scope="module")
def port():
    o = vxo.load_port()
    assert o is not None, "oracle/libvoxels_port.so missing (run __graft_entry__.build())"
    return o


def run_hip(poly, d, m, b, flags, levels=0):
    poly.upload(d, m, b, flags)
    poly.execute(levels)
    return poly.all_levels(), poly.stats()


def check_against(poly, oracle, d, m, b, label):
    g = oracle.grid_from_dense(d, m, b)
    s = oracle.execute(g)
    lv, st = run_hip(poly, d, m, b, g.block_flags())
    ok, msg = fields.surface_equal(lv, s.all_levels(), nrm_tol=NRM_TOL)
    assert ok, "%s vs %s: %s" % (label, oracle.kind, msg)
    assert np.array_equal(st, s.stats()), "%s stats %s vs %s" % (label, st, s.stats())
    return lv


@pytest.mark.parametrize("name", ["sphere64", "terrain32_mat", "noise64_fullrange_mat"])
def test_hip_matches_reference_fixture(poly, name):
    gold = Golden(name)
    lv, st = run_hip(poly, gold.dist, gold.mat, gold.blend, gold.flags)
    ok, msg = fields.surface_equal(lv, gold.levels, nrm_tol=NRM_TOL)
    assert ok, msg
    assert np.array_equal(st, gold.stats)


@pytest.mark.parametrize("knob,value", [("VX_FAST", "0"), ("VX_FAST", "1"), ("VX_FAST", "2"), ("VX_UPPER", "0"), ("VX_SELF_HEAD", "0"),
                                         ("VX_FORCE_WIDE", "1"), ("VX_DIRTY_FUSED", "0"), ("VX_HOST_TIMING", "1"), ("VX_POOL_SLACK", "64")])
def test_hip_runtime_knobs_select_equivalent_paths(knob, value):
    """Every runtime knob the library still reads (INTEGRATION.md lists them; the numeric tuning knobs of the earlier rounds are
    constants now) selects a path production runs reach through their data.  Each one, on fixtures with materials and on a
    full run followed by an incremental one, gives the reference's bytes."""
    from voxels_amd import Polygonizer
    os.environ[knob] = value
    try:
        p = Polygonizer(device=0)
        p.set_materials(vxo.default_lut())
        for name in ("noise64_fullrange_mat", "terrain32_mat", "sphere64"):
            gold = Golden(name)
            lv, st = run_hip(p, gold.dist, gold.mat, gold.blend, gold.flags)
            ok, msg = fields.surface_equal(lv, gold.levels, nrm_tol=NRM_TOL)
            assert ok, "%s=%s, %s: %s" % (knob, value, name, msg)
            assert np.array_equal(st, gold.stats)
    finally:
        del os.environ[knob]
    # ... and a full run with an edit and an incremental run behind it, against a context with the defaults
    from voxels_amd import digest
    q = Polygonizer(device=0)
    q.set_materials(vxo.default_lut())
    got = []
    try:
        for c in (p, q):
            c.create_terrain(128, 21)
            c.execute(0)
            mn, mx = c.inject_ball((60.0, 64.0, 70.0), (24.0, 24.0, 24.0), 11.0, 2)
            c.execute_dirty(mn, mx)
            got.append(digest.surface_digest(c.all_levels()))
        assert digest.digests_equal(got[0], got[1]), "%s=%s: the incremental run differs from the default configuration's" % (knob, value)
    finally:
        p.close(); q.close()


def test_hip_wide_offset_variants_match_reference_fixture():
    """The kernels' 64-bit-offset variants (grids beyond 1024^3, whose mirrors exceed 4 GiB) forced onto a fixture
    (VX_FORCE_WIDE is read when the context is created)."""
    from voxels_amd import Polygonizer
    os.environ["VX_FORCE_WIDE"] = "1"
    try:
        p = Polygonizer(device=0)
    finally:
        del os.environ["VX_FORCE_WIDE"]
    p.set_materials(vxo.default_lut())
    for name in ("noise64_fullrange_mat", "terrain32_mat"):
        gold = Golden(name)
        lv, st = run_hip(p, gold.dist, gold.mat, gold.blend, gold.flags)
        ok, msg = fields.surface_equal(lv, gold.levels, nrm_tol=NRM_TOL)
        assert ok, name + ": " + msg
        assert np.array_equal(st, gold.stats)


def test_hip_known_answer_hash(poly):
    gold = Golden("sphere64")
    lv, _ = run_hip(poly, gold.dist, gold.mat, gold.blend, gold.flags)
    assert "%016x" % vxo.index_hash(lv) == "473e8b8c4d4f3c9d"


def test_normals_are_bit_identical_too(poly, port):
    """north_star allows 1e-5 on normals; they are in fact bit-identical (IEEE sqrt / division on both sides, the
    division spelled out with a shared reciprocal on the device).  ~0.5 M vertices, regular and transition."""
    from voxels_amd import synth
    d, m, b = synth.terrain(256, 0, 256, 77)
    g = port.grid_from_dense(d, m, b)
    ref = port.execute(g).all_levels()
    lv, _ = run_hip(poly, d, m, b, g.block_flags())
    ok, msg = fields.surface_equal(lv, ref, nrm_tol=0.0)
    assert ok, msg
    gold = Golden("noise64_fullrange_mat")
    lv, _ = run_hip(poly, gold.dist, gold.mat, gold.blend, gold.flags)
    ok, msg = fields.surface_equal(lv, gold.levels, nrm_tol=0.0)
    assert ok, msg


@pytest.mark.parametrize("seed,n", [(41, 32), (42, 64), (43, 128)])
def test_hip_vs_oracle_terrain(poly, port, seed, n):
    f = fields.terrain_field(n, seed)
    m, b = fields.materials_for(n, seed)
    g = port.grid_from_float(f, m, b)
    d = g.read_dense()[0]
    check_against(poly, port, d, m, b, "terrain n=%d" % n)
    ref = vxo.load_ref()
    if ref is not None and n <= 64:
        check_against(poly, ref, d, m, b, "terrain n=%d" % n)


@pytest.mark.parametrize("seed,n", [(51, 32), (52, 64), (53, 96)])
def test_hip_vs_oracle_fullrange_noise(poly, port, seed, n):
    q = fields.quantize_full_range(fields.smooth_noise(n, seed, scale=8, amp=3.0))
    m, b = fields.materials_for(n, seed)
    check_against(poly, port, q, m, b, "noise n=%d" % n)


def test_hip_zero_heavy_fields(poly, port):
    """Coarsely quantised fields: many exact zeros -> endpoint vertices, corner reuse, degenerate triangles."""
    rng = np.random.RandomState(0)
    for seed in range(4):
        n = 32 if seed % 2 == 0 else 64
        f = fields.smooth_noise(n, 100 + seed, scale=8, amp=2.0)
        d = np.clip(np.round(f * 1.5), -4, 4).astype(np.int8)
        m = rng.randint(0, 4, (n, n, n)).astype(np.uint8)
        b = rng.randint(0, 256, (n, n, n)).astype(np.uint8)
        check_against(poly, port, d, m, b, "zeros seed=%d" % seed)


def test_hip_degenerate_filter_at_coarse_levels(poly, port):
    """Zero-heavy 256^3 field (cell sizes 1..16): thousands of degenerate triangles on every level."""
    n = 256
    d = np.clip(np.round(fields.smooth_noise(n, 78, scale=32, amp=2.5) * 1.2), -4, 4).astype(np.int8)
    zero = np.zeros((n, n, n), np.uint8)
    check_against(poly, port, d, zero, zero, "zeros 256")


def test_hip_coarse_block_without_cells(poly, port):
    """Surface-bearing coarse blocks without any non-trivial coarse cell (regression: their record was left stale)."""
    n = 64
    d = np.full((n, n, n), 4, np.int8)
    d[21, 21, 21] = -3
    d[40:43, 9, 50] = -2
    zero = np.zeros((n, n, n), np.uint8)
    for _ in range(2):
        check_against(poly, port, d, zero, zero, "bubble")


def test_hip_white_noise_worst_case(poly, port):
    """Every cell non-trivial: 4096 non-trivial cells per block (maximum per-block state)."""
    rng = np.random.RandomState(7)
    d = rng.randint(-128, 128, (32, 32, 32)).astype(np.int8)
    m = rng.randint(0, 3, (32, 32, 32)).astype(np.uint8)
    b = rng.randint(0, 256, (32, 32, 32)).astype(np.uint8)
    check_against(poly, port, d, m, b, "white noise")


@pytest.mark.parametrize("h", [15.5, 16.0, 20.0, 31.5, 47.5])
def test_hip_axis_aligned_planes(poly, port, h):
    """Planes on / between block boundaries: exercises the level-0 emptiness skip (TransVoxelImpl.cpp:1511-1527)."""
    n = 64
    z = np.arange(n).reshape(n, 1, 1) * np.ones((n, n, n))
    d = np.ascontiguousarray(np.clip(np.sign(z - h) * np.ceil(np.abs(z - h)), -4, 4).astype(np.int8))
    zero = np.zeros((n, n, n), np.uint8)
    check_against(poly, port, d, zero, zero, "plane z=%g" % h)


def test_hip_empty_and_full_grids(poly, port):
    for v in (4, -4, 0):
        d = np.full((32, 32, 32), v, np.int8)
        zero = np.zeros((32, 32, 32), np.uint8)
        lv = check_against(poly, port, d, zero, zero, "constant %d" % v)
        assert all(l.totals() == (0, 0, 0, 0, 0) for l in lv)


def test_hip_invalid_material_entries(poly, port):
    gold = Golden("terrain32_mat")
    valid = np.ones(256, np.uint8)
    valid[1] = 0
    poly.set_materials(vxo.default_lut(), valid)
    try:
        lv, _ = run_hip(poly, gold.dist, gold.mat, gold.blend, gold.flags)
        s = port.execute(port.grid_from_dense(gold.dist, gold.mat, gold.blend), valid=valid)
        ok, msg = fields.surface_equal(lv, s.all_levels(), nrm_tol=NRM_TOL)
        assert ok, msg
    finally:
        poly.set_materials(vxo.default_lut())


def test_hip_level_limit(poly, port):
    """num_levels restricts the output to levels 0..k-1 without changing them."""
    gold = Golden("noise64_fullrange_mat")
    lv2, _ = run_hip(poly, gold.dist, gold.mat, gold.blend, gold.flags, levels=2)
    assert len(lv2) == 2
    ok, msg = fields.surface_equal(lv2, gold.levels[:2], nrm_tol=NRM_TOL)
    assert ok, msg
    lv1, _ = run_hip(poly, gold.dist, gold.mat, gold.blend, gold.flags, levels=1)
    ok, msg = fields.surface_equal(lv1, gold.levels[:1], nrm_tol=NRM_TOL)
    assert ok, msg


def test_hip_full_runs_follow_each_other_without_a_reset(poly, port):
    """A full run that follows a full run starts without k_reset: the last kernel of a run (k_tail) leaves the OTHER set of
    run counters, upper-level block maps and list counts in its start state.  Level limits that change between the runs,
    an incremental run in between and the classification-pass variant of the pipeline must all leave the next run a
    clean set."""
    from voxels_amd import Polygonizer, synth
    n = 128
    d, m, b = synth.terrain(n, 0, n, 29)
    g = port.grid_from_dense(d, m, b)
    ref = port.execute(g).all_levels()
    poly.upload(d, m, b, g.block_flags())
    for levels in (4, 2, 4, 1, 3, 4, 4):
        poly.execute(levels)
        ok, msg = fields.surface_equal(poly.all_levels(), ref[:levels], nrm_tol=NRM_TOL)
        assert ok, "levels=%d: %s" % (levels, msg)
    # an incremental run in between (it works on the current set and re-lists the blocks it rebuilt at the end of the lists,
    # like the reference: its result is compared elsewhere), then full runs again
    assert len(poly.execute_dirty((40, 40, 40), (72, 72, 72))) > 0
    for levels in (3, 4):
        poly.execute(levels)
        ok, msg = fields.surface_equal(poly.all_levels(), ref[:levels], nrm_tol=NRM_TOL)
        assert ok, "levels=%d behind the incremental run: %s" % (levels, msg)
    # the same through the pipeline with a classification pass (the form dense surfaces and incremental runs use)
    os.environ["VX_SELF_HEAD"] = "0"
    try:
        q = Polygonizer(device=0)
    finally:
        del os.environ["VX_SELF_HEAD"]
    q.set_materials(vxo.default_lut())
    q.upload(d, m, b, g.block_flags())
    for levels in (4, 2, 4):
        q.execute(levels)
        ok, msg = fields.surface_equal(q.all_levels(), ref[:levels], nrm_tol=NRM_TOL)
        assert ok, "classification pass, levels=%d: %s" % (levels, msg)
    q.close()


def test_hip_rerun_is_deterministic_per_block(poly):
    """Pool offsets may differ between runs (atomic reservation) but every block's bytes must not."""
    gold = Golden("noise64_fullrange_mat")
    a, _ = run_hip(poly, gold.dist, gold.mat, gold.blend, gold.flags)
    b, _ = run_hip(poly, gold.dist, gold.mat, gold.blend, gold.flags)
    ok, msg = fields.surface_equal(a, b)
    assert ok, msg


def test_hip_256_properties(poly, port):
    """256^3 terrain (config 2 size): full comparison with the port on level 0 counts + structural properties."""
    n = 256
    f = fields.terrain_field(n, 61)
    m, b = fields.materials_for(n, 61)
    g = port.grid_from_float(f, m, b)
    d = g.read_dense()[0]
    lv = check_against(poly, port, d, m, b, "terrain 256")
    for l in lv:
        # every index addresses a vertex of its own block; triangle lists
        ov = oi = 0
        for info in l.infos:
            nv, ni = int(info["n_verts"]), int(info["n_idx"])
            assert ni % 3 == 0
            if ni:
                assert l.idx[oi:oi + ni].max() < nv
            ov += nv
            oi += ni
        # unit (or zero) normals
        if len(l.verts):
            ln = np.linalg.norm(l.verts["nrm"], axis=1)
            assert np.all((np.abs(ln - 1) < 1e-4) | (ln == 0))


_BENCH_ORACLE = {}


def bench_oracle(port, n):
    """The oracle's surface of the bench terrain (levels 0..3), computed once per size for the tests that compare with it."""
    if n not in _BENCH_ORACLE:
        from voxels_amd import synth
        d, m, b = synth.terrain(n)
        g = port.grid_from_dense(d, m, b)
        _BENCH_ORACLE[n] = (port.execute(g).all_levels(), np.ascontiguousarray(g.block_flags(), np.uint8))
    return _BENCH_ORACLE[n][0][:4], _BENCH_ORACLE[n][1]


def bench_oracle_all_levels(port, n):
    """... and every level the reference produces (log2(n / 16) + 1)."""
    bench_oracle(port, n)
    return _BENCH_ORACLE[n][0]


@pytest.mark.parametrize("n", [512, 1024])
def test_hip_bench_terrain_full_parity(poly, port, n):
    """The bench workload itself (synthetic terrain, LOD levels 0..3) against the oracle port, every byte, twice
    (pool offsets differ between runs, block contents must not).  The grid is generated on the device, as bench.py does."""
    ref, flags = bench_oracle(port, n)
    poly.create_terrain(n)
    for _ in range(2):
        poly.execute(4)
        ok, msg = fields.surface_equal(poly.all_levels(), ref, nrm_tol=NRM_TOL)
        assert ok, msg


@pytest.mark.parametrize("axis", ["y", "z"])
def test_hip_config4_1024_eight_slabs(port, axis):
    """BASELINE config 4 on one GPU: the 1024^3 terrain, LOD levels 0..3, cut into 8 slabs (128 voxels = one level-3 block
    layer each) along y and along z.  Eight contexts hold one slab each — generated on the device, own layers only — the
    halo comes from vx_halo_exchange_group, every context polygonizes its slab, and the merged result must be the
    oracle's surface of the whole grid, every byte."""
    import torch
    from voxels_amd import Polygonizer
    from voxels_amd.slab import SlabBuffers, merge_rank_levels
    n, levels, world = 1024, 4, 8
    ref, _ = bench_oracle(port, n)
    dev = torch.device("cuda", 0)
    polys, slabs = [], []
    for r in range(world):
        slab = SlabBuffers(torch, n, r, world, dev, axis=axis)
        p = Polygonizer(device=0)
        p.set_materials(vxo.default_lut())
        slab.attach(p)
        p.fill_terrain(1337)
        # drop everything the rank does not own: halo layers and the other ranks' flags must come from the exchange
        if axis == "z":
            slab.dist[0].zero_(); slab.dist[-2:].zero_(); slab.mat[-1].zero_(); slab.blend[-1].zero_()
        else:
            slab.dist[:, 0].zero_(); slab.dist[:, -2:].zero_(); slab.mat[:, -1].zero_(); slab.blend[:, -1].zero_()
        polys.append(p); slabs.append(slab)
    torch.cuda.synchronize()
    Polygonizer.halo_exchange_group(polys)
    parts = []
    for p in polys:
        p.execute(levels)
        parts.append(p.all_levels())
    # the levels whose block is larger than a slab (4, 5, 6: VERDICT r5 item 6) from the slabs' joined fields on one context
    from voxels_amd.slab import CoarseLevels

    def make():
        q = Polygonizer(device=0)
        q.set_materials(vxo.default_lut())
        return q
    torch.cuda.synchronize()
    coarse = CoarseLevels.from_slabs(slabs, make)
    for p in polys:
        p.close()
    del slabs
    assert coarse.first == 4 and coarse.levels == 7
    upper = coarse.execute()
    coarse.close()
    ok, msg = fields.surface_equal(merge_rank_levels(parts), ref, nrm_tol=NRM_TOL)
    assert ok, msg
    ok, msg = fields.surface_equal(merge_rank_levels(parts) + upper, bench_oracle_all_levels(port, n), nrm_tol=NRM_TOL)
    assert ok, "all 7 levels of the sharded run: " + msg
    # the correctness bit `bench.py --gpus 8` prints (single-GPU emulation of it): the ranks' digests, packed and summed as
    # its all_reduce does, equal the digest of the whole surface - here the oracle's
    from voxels_amd import digest
    summed = digest.unpack(sum(digest.pack(digest.surface_digest(part)) for part in parts), levels)
    assert digest.digests_equal(summed, digest.surface_digest(ref)), "summed rank digests vs oracle digest"


@pytest.mark.parametrize("seed", [300, 304, 317])
def test_hip_slab_fuzz_seeds(seed):
    """Three fixed configurations of tools/fuzz_slabs.py (random field kind, size, level limit, world size, axis): the
    ranks' slabs polygonized one after the other merge into the whole-grid result of the same library."""
    import sys
    sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tools"))
    import fuzz_slabs
    assert fuzz_slabs.check_seed(seed) is not None


def test_hip_carve_modify_matches_reference_fixture(poly, port):
    """Config 5 in small (reference fixture): full run, sphere carve, incremental re-polygonization of the dirty box."""
    gold = Golden("terrain64_carve_modify")
    pre = (gold["pre_dist"], gold["pre_mat"], gold["pre_blend"])
    poly.upload(*pre, port.grid_from_dense(*pre).block_flags())
    poly.execute()
    ids, (d, m, b) = fields.edited_blocks(pre, (gold.dist, gold.mat, gold.blend))
    poly.update_blocks(ids, d.view(np.int8), m, b, gold.flags)
    mod = poly.execute_dirty(gold["box_min"], gold["box_max"])
    assert np.array_equal(mod, gold["modified_ids"])
    ok, msg = fields.surface_equal(poly.all_levels(), gold.levels, nrm_tol=NRM_TOL)
    assert ok, msg
    assert np.array_equal(poly.stats(), gold.stats)


def test_hip_fast_flag_protocol_equals_the_conservative_one(port):
    """k_main's workgroups hand material caches, bitmaps and cell counts to each other through write-through stores, relaxed
    flags and write-through loads (no fences).  libvoxels_hip_conservative.so is the same source with release / acquire
    fences around every flag (-DVX_CONSERVATIVE_SYNC): both libraries on the same grids - full runs with every level, the
    bench's 4 levels, an incremental run - must produce the same bytes, and one of them is checked against the oracle."""
    from voxels_amd import Polygonizer, digest, synth
    from voxels_amd.binding import HipLibrary
    path = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "voxels_amd", "csrc", "libvoxels_hip_conservative.so")
    assert os.path.exists(path), "libvoxels_hip_conservative.so missing (run __graft_entry__.build())"
    slow = Polygonizer(device=0, library=HipLibrary(path))
    fast = Polygonizer(device=0)
    for q in (slow, fast):
        q.set_materials(vxo.default_lut())
    try:
        for n, seed, levels in ((256, 5, 0), (512, 1337, 4), (128, 9, 0)):
            got = []
            for q in (slow, fast):
                q.create_terrain(n, seed)
                for _ in range(3):  # (repeated: the flags of one run must not be taken for the next one's)
                    q.execute(levels)
                got.append(digest.surface_digest(q.all_levels()))
            assert digest.digests_equal(got[0], got[1]), "n=%d: fast and conservative flag protocols disagree" % n
            if n == 256:
                d, m, b = synth.terrain(n, seed=seed)
                ref = port.execute(port.grid_from_dense(d, m, b)).all_levels()
                ok, msg = fields.surface_equal(slow.all_levels(), ref, nrm_tol=NRM_TOL)
                assert ok, msg
        # an incremental run on both (k_main<true>: the same flags, a subset of the blocks)
        got = []
        for q in (slow, fast):
            mn, mx = q.inject_ball((60.0, 64.0, 70.0), (24.0, 24.0, 24.0), 11.0, 2)
            q.execute_dirty(mn, mx)
            got.append(digest.surface_digest(q.all_levels()))
        assert digest.digests_equal(got[0], got[1]), "incremental run: fast and conservative flag protocols disagree"
    finally:
        slow.close(); fast.close()


def test_hip_partial_run_keeps_every_cache(port):
    """vx_polygonize_from (what libVoxels.so's primary context runs when helper devices mesh the finer levels): no meshes below
    the first meshed level, the levels from it up byte for byte those of a full run, statistics that add up with those of the
    levels left out - and every cache a Modification continues from: the same edit after a partial and after a full run
    rebuilds the same blocks with the same bytes, against the oracle."""
    from voxels_amd import Polygonizer, synth
    n, first = 256, 2
    full, part = Polygonizer(device=0), Polygonizer(device=0)
    try:
        for q in (full, part):
            q.set_materials(vxo.default_lut())
            q.create_terrain(n, 11)
        full.execute()
        info = part.execute_from(0, first)
        assert info.first_meshed_level == first and info.levels == full.info.levels
        a, b = full.all_levels(), part.all_levels()
        for l in range(first):
            assert len(b[l].infos) == 0, "level %d lists blocks" % l
        ok, msg = fields.surface_equal(b[first:], a[first:], nrm_tol=0.0)
        assert ok, msg
        low = Polygonizer(device=0)
        low.set_materials(vxo.default_lut()); low.create_terrain(n, 11); low.execute(first)
        assert np.array_equal(part.stats() + low.stats(), full.stats())
        low.close()
        # the same carve on both, and on the oracle
        d, m, bl = synth.terrain(n, seed=11)
        g = port.grid_from_dense(d, m, bl)
        s = port.execute(g)
        col = d[:, n // 2, n // 2]
        zs = float(np.argmax(col >= 0)) if (col >= 0).any() else n / 2.0
        for pos, ext, r in (((n / 2.0, n / 2.0, zs), (30.0, 30.0, 30.0), 13.0), ((n / 2.0 + 21.5, n / 2.0 - 9.25, zs + 2.0), (24.0, 24.0, 24.0), 9.0)):
            mn, mx = g.inject_ball(pos, ext, r, 2)
            ref_ids = port.execute_modify(g, s, mn, mx)
            got = []
            for q in (full, part):
                a0, b0 = q.inject_ball(pos, ext, r, 2)
                got.append(q.execute_dirty(a0, b0))
                assert np.array_equal(q.stats(), s.stats())
            assert np.array_equal(got[0], ref_ids) and np.array_equal(got[1], ref_ids)
            ok, msg = fields.surface_equal(full.all_levels(), s.all_levels(), nrm_tol=NRM_TOL)
            assert ok, msg
            # the partial context lists, below its first meshed level, exactly the rebuilt blocks - with the bytes the full one has for them
            fa, pa = full.all_levels(), part.all_levels()
            ok, msg = fields.surface_equal(pa[first:], fa[first:], nrm_tol=0.0)
            assert ok, msg
            for l in range(first):
                assert len(pa[l].infos) > 0
                ok, msg = fields.listed_blocks_equal_by_id(pa[l], fa[l])
                assert ok, "level %d: %s" % (l, msg)
    finally:
        full.close(); part.close()


def test_hip_incremental_runs_as_chain_of_launches(port):
    """Incremental runs are three launches by default (k_dirty_head | k_main<true> | k_dirty_tail); the chain of launches
    with work lists remains for surfaces with blocks beyond the first capacity class.  The same chain of edits through a
    context that is told to use the chain (VX_DIRTY_FUSED=0), against the oracle."""
    from voxels_amd import Polygonizer, synth
    n = 128
    os.environ["VX_DIRTY_FUSED"] = "0"
    try:
        q = Polygonizer(device=0)
    finally:
        del os.environ["VX_DIRTY_FUSED"]
    q.set_materials(vxo.default_lut())
    d0, m0, b0 = synth.terrain(n, seed=33)
    g = port.grid_from_dense(d0, m0, b0)
    s = port.execute(g)
    q.upload(*g.read_dense(), g.block_flags())
    q.execute()
    c = n / 2.0
    col = d0[:, n // 2, n // 2]
    zs = float(np.argmax(col >= 0)) if (col >= 0).any() else c
    for t, pos, ext, r in ((2, (c, c, zs), (30, 30, 30), 14.0), (0, (c + 9, c - 11, zs + 3), (16, 16, 16), 6.0), (2, (c - 20, c + 4, zs - 2), (12, 12, 12), 5.0)):
        mn, mx = g.inject_ball(pos, ext, r, t)
        ref_ids = port.execute_modify(g, s, mn, mx)
        a, bq = q.inject_ball(pos, ext, r, t)
        assert np.array_equal(a, mn) and np.array_equal(bq, mx)
        got = q.execute_dirty(mn, mx)
        assert np.array_equal(got, ref_ids)
        ok, msg = fields.surface_equal(q.all_levels(), s.all_levels(), nrm_tol=NRM_TOL)
        assert ok, msg
        assert np.array_equal(q.stats(), s.stats())
    q.close()


@pytest.mark.parametrize("fused", ["1", "0"])
def test_hip_edit_sequence_with_tight_pools(port, fused):
    """VX_POOL_SLACK=64 (tests/test_emu.py has the same sequence on the emulation): full runs that grow their pools, incremental
    runs that do not fit - repeated behind packing or growing the pools, with the slots and cache blocks the failed attempt
    already claimed - and packing in front of a run; both incremental paths.  Every step against the oracle."""
    from voxels_amd import Polygonizer, synth
    n = 64
    os.environ["VX_POOL_SLACK"] = "64"
    os.environ["VX_DIRTY_FUSED"] = fused
    try:
        q = Polygonizer(device=0)
    finally:
        del os.environ["VX_POOL_SLACK"], os.environ["VX_DIRTY_FUSED"]
    try:
        q.set_materials(vxo.default_lut())
        d0, m0, b0 = synth.terrain(n, seed=21)
        g = port.grid_from_dense(d0, m0, b0)
        s = port.execute(g)
        q.upload_packed(g.pack())
        assert q.execute().retries >= 1
        rng = np.random.RandomState(5)
        sizes = []
        for k in range(24):
            pos = tuple(float(x) for x in rng.uniform(14, n - 14, 3).round(2))
            args = (pos, (16.0, 16.0, 16.0), float(rng.uniform(3, 7)), 2 if k % 3 else 0)
            mn, mx = g.inject_ball(*args)
            q.inject_ball(*args)
            ref_ids = port.execute_modify(g, s, mn, mx)
            got = q.execute_dirty(mn, mx)
            assert np.array_equal(got, ref_ids), k
            ok, msg = fields.surface_equal(q.all_levels(), s.all_levels(), nrm_tol=NRM_TOL)
            assert ok, "edit %d: %s" % (k, msg)
            assert np.array_equal(q.stats(), s.stats())
            sizes.append(int(q.info.total_verts))
        assert any(b < a for a, b in zip(sizes, sizes[1:])), "the pools were never packed: %s" % sizes
        assert np.array_equal(q.pack(), g.pack())
    finally:
        q.close()


@pytest.mark.parametrize("n", [64, 256])
def test_hip_repeated_edits_vs_port(poly, port, n):
    """Chained edits with incremental runs: device-resident caches persist like the reference's PolygonMap caches."""
    from voxels_amd import synth
    d0, m0, b0 = synth.terrain(n, seed=21)
    g = port.grid_from_dense(d0, m0, b0)
    s = port.execute(g)
    pre = g.read_dense()
    poly.upload(*pre, g.block_flags())
    poly.execute()
    hm = poly.host_meshes()  # the one-DMA host copy of the pools, kept up to date across the incremental runs
    fields.check_host_meshes(poly, hm)
    c = n / 2.0
    edits = ((2, (c - 2.0, c + 1.5, c - 0.75), (20, 20, 20), 7.0), (0, (c + 8, c - 12, c - 7), (16, 16, 16), 6.0),
             (2, (3.0, n - 4.0, c - 2), (12, 12, 12), 5.0), (2, (c - 1.0, c + 1.0, c - 1.0), (10, 10, 10), 4.0))
    for t, pos, ext, r in edits:
        mn, mx = g.inject_ball(pos, ext, r, t)
        ref_ids = port.execute_modify(g, s, mn, mx)
        post = g.read_dense()
        ids, (dd, mm, bb) = fields.edited_blocks(pre, post)
        poly.update_blocks(ids, dd.view(np.int8), mm, bb, g.block_flags())
        pre = post
        got = poly.execute_dirty(mn, mx)
        assert np.array_equal(got, ref_ids)
        ok, msg = fields.surface_equal(poly.all_levels(), s.all_levels(), nrm_tol=NRM_TOL)
        assert ok, msg
        assert np.array_equal(poly.stats(), s.stats())
        hm = poly.host_meshes(previous=hm)
        fields.check_host_meshes(poly, hm)
    poly.compact_pools()  # the pools are rewritten: an arena of the old layout is refilled as a whole
    hm = poly.host_meshes(previous=hm)
    fields.check_host_meshes(poly, hm)
    ok, msg = fields.surface_equal(poly.all_levels(), s.all_levels(), nrm_tol=NRM_TOL)
    assert ok, msg
    other = poly.host_meshes()  # a second arena filled from scratch holds the same bytes
    assert np.array_equal(other.verts, hm.verts) and np.array_equal(other.indices, hm.indices)


def test_hip_config2_256_lod0_only(poly, port):
    """BASELINE config 2: 256^3 noise terrain, LOD 0 regular cells only."""
    from voxels_amd import synth
    d, m, b = synth.terrain(256)
    g = port.grid_from_dense(d, m, b)
    ref = port.execute(g).all_levels()[:1]
    poly.upload(d, m, b, g.block_flags())
    info = poly.execute(1)
    assert info.levels == 1
    ok, msg = fields.surface_equal(poly.all_levels(), ref, nrm_tol=NRM_TOL)
    assert ok, msg


def test_hip_device_arithmetic_exhaustive(poly):
    """edge_t_crossing for all crossed int8 sample pairs, normalize_fix_zero for all 2^24 non-negative int8 gradients
    (scaled and unscaled; against sqrtf + IEEE division), on the GPU (vx_selftest)."""
    r = poly.selftest()
    print("selftest:", r.tolist())
    assert r[0] == 0 and r[1] == 0 and r[2] == 0 and r[11] == 0, r.tolist()


def widen_near_surface(d):
    """Full-range distances for a clamped (+-4) field: every voxel nearer than the clamp gets |d| * 16 - r, r = a
    position hash in [0, 15] (sign kept, zeros stay zero), so that t = (v1 << 8) / (v1 - v0) takes hundreds of values;
    what Grid::ModifyBlockDistanceData lets an application write (no +-4 clamp, include/Grid.h:132)."""
    n = d.shape[0]
    ax = np.arange(n, dtype=np.int32)
    r = ((ax.reshape(n, 1, 1) * 283 + ax.reshape(1, n, 1) * 179 + ax.reshape(1, 1, n) * 73) & 15).astype(np.int16)
    a = np.abs(d.astype(np.int16))
    wide = np.where((a > 0) & (a < 4), np.sign(d).astype(np.int16) * (a * 16 - r), d.astype(np.int16))
    return wide.astype(np.int8)


def test_hip_config3_512_three_levels(poly, port):
    """BASELINE config 3: 512^3 terrain with materials, LOD levels 0..2 — level 2 is not the reference's last level
    (TransVoxelImpl.cpp:523-525), so all of levels 1 and 2 carry transition cells — against the oracle; then once more
    with full-range distances pushed through vx_grid_update_blocks (Grid::ModifyBlockDistanceData)."""
    from voxels_amd import synth
    n = 512
    ref, flags = bench_oracle(port, n)
    poly.create_terrain(n)
    info = poly.execute(3)
    assert info.levels == 3
    got = poly.all_levels()
    assert int(got[1].infos["n_tverts"].sum()) > 0 and int(got[2].infos["n_tverts"].sum()) > 0
    ok, msg = fields.surface_equal(got, ref[:3], nrm_tol=NRM_TOL)
    assert ok, msg
    d, m, b = synth.terrain(n)
    wide = widen_near_surface(d)
    g = port.grid_from_dense(wide, m, b)
    ref_wide = port.execute(g).all_levels()[:3]
    ids, (dd, mm, bb) = fields.edited_blocks((d, m, b), (wide, m, b))
    poly.update_blocks(ids, dd.view(np.int8), mm, bb, g.block_flags())
    poly.execute(3)
    ok, msg = fields.surface_equal(poly.all_levels(), ref_wide, nrm_tol=NRM_TOL)
    assert ok, "full-range distances: " + msg
    t_values = np.unique(np.abs(wide[np.abs(wide) < 64]))
    assert len(t_values) > 40


def test_hip_caves_workload_256(poly, port):
    """The second bench workload (bench.py config.extra): the "caves" style of the generator — surface in most blocks, many
    blocks near the 640-cell class boundary — generated on the device (bytes = the host generator's) and polygonized,
    LOD levels 0..3, against the oracle."""
    from voxels_amd import Polygonizer, synth
    n = 256
    d, m, b = synth.terrain(n, style=1)
    g = port.grid_from_dense(d, m, b)
    ref = port.execute(g).all_levels()[:4]
    poly.create_terrain(n, 1337, 1)
    host = Polygonizer(device=0)
    host.upload(d, m, b, g.block_flags())
    assert np.array_equal(poly.pack(), host.pack())
    host.close()
    info = poly.execute(4)
    assert info.active_blocks[0] > 0.5 * (n // 16) ** 3
    ok, msg = fields.surface_equal(poly.all_levels(), ref, nrm_tol=NRM_TOL)
    assert ok, msg


def test_hip_config5_512_carve_incremental(poly, port):
    """BASELINE config 5: 512^3 terrain, sphere carve (IT_Subtract, r = 20) at the surface, incremental re-polygonization
    of the dirty blocks; parity with the oracle doing the same two calls."""
    from voxels_amd import synth
    n = 512
    d, m, b = synth.terrain(n)
    g = port.grid_from_dense(d, m, b)
    s = port.execute(g)
    pre = g.read_dense()
    poly.upload(*pre, g.block_flags())
    poly.execute()
    # surface height under (256, 256): first non-negative distance going up
    col = pre[0][:, 256, 256]
    h = float(np.argmax(col >= 0))
    mn, mx = g.inject_ball((256.0, 256.0, h), (48, 48, 48), 20.0, 2)
    ref_ids = port.execute_modify(g, s, mn, mx)
    post = g.read_dense()
    ids, (dd, mm, bb) = fields.edited_blocks(pre, post)
    poly.update_blocks(ids, dd.view(np.int8), mm, bb, g.block_flags())
    got = poly.execute_dirty(mn, mx)
    print("edit: %d grid blocks uploaded, %d polygon blocks rebuilt, device %.3f ms" % (len(ids), len(got), poly.info.device_ms))
    assert np.array_equal(got, ref_ids)
    ok, msg = fields.surface_equal(poly.all_levels(), s.all_levels(), nrm_tol=NRM_TOL)
    assert ok, msg
    assert np.array_equal(poly.stats(), s.stats())


@pytest.mark.gpu
@pytest.mark.parametrize("world,axis", [(2, "z"), (4, "z"), (2, "y"), (4, "y")])
def test_hip_slab_attach_matches_full_run(poly, port, world, axis):
    """The multi-GPU data path on one GPU: every rank's slab (own planes + halo planes, attached device tensors) is
    polygonized on its own and the concatenation must equal the reference's surface of the whole grid — including the
    quiet-block shortcut, which samples the halo planes for the neighbouring slabs' signs."""
    import torch
    from voxels_amd import synth
    from voxels_amd.slab import SlabBuffers, merge_rank_levels
    n, levels, seed = 256, 3, 11
    d, m, b = synth.terrain(n, 0, n, seed)
    g = port.grid_from_dense(d, m, b)
    ref = port.execute(g)
    flags = np.ascontiguousarray(g.block_flags(), np.uint8)
    assert np.array_equal(flags, synth.block_empty_flags(d))
    dev = torch.device("cuda", 0)
    per_rank, stats = [], np.zeros(20, np.int64)
    for r in range(world):
        slab = SlabBuffers(torch, n, r, world, dev, axis=axis)
        slab.fill_from_full(d, m, b, flags)
        torch.cuda.synchronize()
        slab.attach(poly)
        poly.execute(levels)
        per_rank.append(poly.all_levels())
        stats += poly.stats().astype(np.int64)
    merged = merge_rank_levels(per_rank)
    want = ref.all_levels()[:levels]
    # only the first `levels` levels are compared; the reference ran all of them
    ok, msg = fields.surface_equal(merged, want, nrm_tol=NRM_TOL)
    assert ok, msg


@pytest.mark.gpu
@pytest.mark.parametrize("world,axis", [(2, "z"), (4, "z"), (2, "y"), (4, "y")])
def test_hip_halo_exchange_group(port, world, axis):
    """Row b8 on one GPU: `world` contexts hold one slab each with NOTHING but their own layers; vx_halo_exchange_group
    (k_halo_move pack -> peer copy -> unpack, the flags' boundary layers included) must make the union of their runs the
    reference's surface.  The RCCL variant (vx_halo_exchange) shares everything but the transport."""
    import torch
    from voxels_amd import Polygonizer, synth
    n, levels, seed = 256, 3, 7
    d, m, b = synth.terrain(n, 0, n, seed)
    ref = port.execute(port.grid_from_dense(d, m, b))

    def mk():
        p = Polygonizer(device=0)
        p.set_materials(vxo.default_lut())
        return p
    fields.check_halo_exchange_group(mk, torch, torch.device("cuda", 0), n, levels, world, axis, ref.all_levels(), seed=seed, nrm_tol=NRM_TOL)


@pytest.mark.gpu
@pytest.mark.parametrize("axis", ["y", "z"])
def test_hip_halo_exchange_after_a_neighbour_changed(port, axis):
    """A rank whose own rows did NOT change keeps its mirrors; when its neighbour's rows change, the exchange brings new halo
    rows into those mirrors - and the sign summaries of the halo block layer, which decide what the last owned block layer
    polygonizes at all, have to follow them (refresh_halo_layers).  Two slabs of one terrain run once; then the upper rank
    replaces its rows by those of another terrain, the ranks exchange, and the union must be the surface of the mixed grid."""
    import torch
    from voxels_amd import Polygonizer, synth
    from voxels_amd.slab import SlabBuffers, merge_rank_levels
    n, levels, world = 128, 3, 2
    dev = torch.device("cuda", 0)
    d, m, b = synth.terrain(n, 0, n, 3)
    d2, m2, b2 = synth.terrain(n, 0, n, 19)
    half = n // 2
    mixed = [np.ascontiguousarray(a.copy()) for a in (d, m, b)]
    for a, a2 in zip(mixed, (d2, m2, b2)):
        if axis == "z":
            a[half:] = a2[half:]
        else:
            a[:, half:] = a2[:, half:]
    ref = port.execute(port.grid_from_dense(*mixed)).all_levels()
    flags0, flags1 = synth.block_empty_flags(d), synth.block_empty_flags(mixed[0])
    polys, slabs = [], []
    for r in range(world):
        slab = SlabBuffers(torch, n, r, world, dev, axis=axis)
        slab.fill_from_full(d, m, b, flags0)
        p = Polygonizer(device=0)
        p.set_materials(vxo.default_lut())
        slab.attach(p)
        p.execute(levels)   # mirrors and sign summaries of the first terrain are current on both ranks
        polys.append(p)
        slabs.append(slab)
    # the upper rank's own rows change (it tells the library so); the lower rank is untouched and keeps its mirrors
    up = slabs[1]
    sl = slice(up.z0, up.z1)
    nb = n // 16
    if axis == "z":
        per = flags1.size // world
        up.fill_own(np.ascontiguousarray(mixed[0][sl]), np.ascontiguousarray(mixed[1][sl]), np.ascontiguousarray(mixed[2][sl]), flags1[per:])
    else:
        own = np.zeros_like(flags1).reshape(nb, nb, nb)
        own[:, up.z0 // 16:up.z1 // 16] = flags1.reshape(nb, nb, nb)[:, up.z0 // 16:up.z1 // 16]
        up.fill_own(np.ascontiguousarray(mixed[0][:, sl]), np.ascontiguousarray(mixed[1][:, sl]), np.ascontiguousarray(mixed[2][:, sl]), own.reshape(-1))
    torch.cuda.synchronize()
    polys[1].invalidate()
    Polygonizer.halo_exchange_group(polys)
    parts = []
    for p in polys:
        p.execute(levels)
        parts.append(p.all_levels())
    ok, msg = fields.surface_equal(merge_rank_levels(parts), ref[:levels], nrm_tol=NRM_TOL)
    assert ok, msg


@pytest.mark.gpu
@pytest.mark.parametrize("n", [256])
def test_hip_device_terrain(n):
    """§8(f) row 3: k_terrain_height / k_terrain_fill + k_edit_flags reproduce vxs_terrain + vxs_block_empty_flags byte for
    byte (whole grid through the grid file; 2 slabs per axis with their halo layers)."""
    import torch
    from voxels_amd import Polygonizer
    fields.check_device_terrain(lambda: Polygonizer(device=0), torch, torch.device("cuda", 0), n)


@pytest.mark.gpu
def test_hip_rccl_communicator_single_rank(port):
    """vx_comm_unique_id / vx_comm_init / vx_halo_exchange with one rank: the RCCL library is found, the communicator
    comes up on this GPU, and an exchange without neighbours is an empty group that leaves the slab intact."""
    import torch
    from voxels_amd import Polygonizer, synth
    from voxels_amd.slab import SlabBuffers
    n, levels = 64, 2
    d, m, b = synth.terrain(n, 0, n, 3)
    flags = synth.block_empty_flags(d)
    dev = torch.device("cuda", 0)
    slab = SlabBuffers(torch, n, 0, 1, dev, axis="y")
    slab.fill_own(d, m, b, flags)
    torch.cuda.synchronize()
    p = Polygonizer(device=0)
    p.set_materials(vxo.default_lut())
    slab.attach(p)
    p.comm_init(1, 0, p.comm_unique_id())
    p.halo_exchange()
    p.execute(levels)
    ref = port.execute(port.grid_from_dense(d, m, b))
    ok, msg = fields.surface_equal(p.all_levels(), ref.all_levels()[:levels], nrm_tol=NRM_TOL)
    assert ok, msg


@pytest.mark.gpu
@pytest.mark.parametrize("n,noisy", [(128, False), (64, True)])
def test_hip_upload_packed_grid(poly, port, n, noisy):
    """§8(f) row 1: the reference's grid file (PackForSave) expanded on the device — blocks byte for byte, flags, and
    the surface polygonized from it."""
    from test_emu import _packed_case, check_packed
    d, m, b = _packed_case(n, 33, noisy)
    check_packed_hip(poly, port, d, m, b, "packed n=%d noisy=%s" % (n, noisy))


def check_packed_hip(poly, port, d, m, b, label):
    n = d.shape[0]
    g = port.grid_from_dense(d, m, b)
    blob = g.pack()
    poly.upload_packed(blob)
    flags = g.block_flags()
    nb = n // 16
    for bid in np.unique(np.random.RandomState(2).randint(0, nb ** 3, 80)):
        bx, by, bz = bid % nb, (bid // nb) % nb, bid // (nb * nb)
        sl = (slice(bz * 16, bz * 16 + 16), slice(by * 16, by * 16 + 16), slice(bx * 16, bx * 16 + 16))
        bd, bm, bb, fl = poly.read_block(bid)
        assert np.array_equal(bd, d[sl]) and np.array_equal(bm, m[sl]) and np.array_equal(bb, b[sl]), "%s: block %d differs" % (label, bid)
        assert fl == flags[bid]
    poly.execute()
    s = port.execute(g)
    ok, msg = fields.surface_equal(poly.all_levels(), s.all_levels(), nrm_tol=NRM_TOL)
    assert ok, label + ": " + msg
    assert np.array_equal(poly.stats(), s.stats()), label


@pytest.mark.gpu
def test_hip_meshes_exported_to_another_process(poly):
    """§8(f) row 4 (interop): vx_export_meshes hands out inter-process handles of the two pools; a process that knows nothing
    of the library (tests/ipc_reader.py: hipIpcOpenMemHandle + hipMemcpy) reads exactly the bytes vx_device_meshes points at.
    An appending incremental run keeps the generation, a full run changes it."""
    import ctypes as C
    import hashlib
    import subprocess
    gold = Golden("noise64_fullrange_mat")
    poly.upload(gold.dist, gold.mat, gold.blend, gold.flags)
    poly.execute()
    dv, di, nv, ni = poly.device_meshes()
    ex = poly.export_meshes()
    assert ex["n_verts"] == nv and ex["n_indices"] == ni and ex["verts_capacity"] >= nv and ex["indices_capacity"] >= ni
    hip = C.CDLL("libamdhip64.so")
    hip.hipMemcpy.argtypes = [C.c_void_p, C.c_void_p, C.c_size_t, C.c_int]
    verts = np.zeros(nv * 48, np.uint8)
    idx = np.zeros(ni, np.uint32)
    assert hip.hipMemcpy(verts.ctypes.data_as(C.c_void_p), C.c_void_p(dv), nv * 48, 2) == 0
    assert hip.hipMemcpy(idx.ctypes.data_as(C.c_void_p), C.c_void_p(di), ni * 4, 2) == 0
    want = "ipc digest %s %s" % (hashlib.sha256(verts.tobytes()).hexdigest(), hashlib.sha256(idx.tobytes()).hexdigest())
    env = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY="0")
    r = subprocess.run([sys.executable, os.path.join(os.path.dirname(os.path.abspath(__file__)), "ipc_reader.py"), ex["verts_handle"].hex(), ex["indices_handle"].hex(), str(nv), str(ni)],
                       capture_output=True, text=True, timeout=300, env=env)
    assert r.returncode == 0, r.stderr[-2000:]
    assert want in r.stdout, (want, r.stdout[-500:])
    # a full run rewrites the pools: another generation
    poly.execute()
    assert poly.export_meshes()["generation"] != ex["generation"]


@pytest.mark.gpu
def test_hip_device_resident_meshes(poly):
    """§8(f) row 4 (first half): the pools stay on the device; vx_device_meshes + vx_level_ranges must describe exactly
    what vx_download_level copies.  The pools are read back here with a plain hipMemcpy on the raw pointers."""
    import ctypes as C
    gold = Golden("noise64_fullrange_mat")
    poly.upload(gold.dist, gold.mat, gold.blend, gold.flags)
    poly.execute()
    dv, di, nv, ni = poly.device_meshes()
    hip = C.CDLL("libamdhip64.so")
    hip.hipMemcpy.argtypes = [C.c_void_p, C.c_void_p, C.c_size_t, C.c_int]
    vdt = poly.level(0).verts.dtype
    verts = np.zeros(nv, vdt)
    idx = np.zeros(ni, np.uint32)
    assert hip.hipMemcpy(verts.ctypes.data_as(C.c_void_p), C.c_void_p(dv), nv * 48, 2) == 0
    assert hip.hipMemcpy(idx.ctypes.data_as(C.c_void_p), C.c_void_p(di), ni * 4, 2) == 0
    from voxels_amd.binding import LISTED_BLOCK_DTYPE
    for l in range(3):
        # the device-built block table (written by the run itself) read back raw: same blocks, same order, same ranges
        tab, nb = poly.device_block_table(l)
        table = np.zeros(nb, LISTED_BLOCK_DTYPE)
        assert LISTED_BLOCK_DTYPE.itemsize == 156
        assert nb == 0 or hip.hipMemcpy(table.ctypes.data_as(C.c_void_p), C.c_void_p(tab), nb * 156, 2) == 0
        lv, rg = poly.level(l), poly.level_ranges(l)
        assert nb == lv.infos.size
        for name_t, name_i in (("id", "id"), ("v_count", "n_verts"), ("i_count", "n_idx"), ("tv_count", "n_tverts"), ("ti_count", "n_tidx"),
                               ("min_corner", "min_corner"), ("max_corner", "max_corner")):
            assert np.array_equal(table[name_t], lv.infos[name_i]), name_t
        for name in ("v_off", "i_off", "tv_off", "ti_off"):
            assert np.array_equal(table[name], rg[name]), name
        assert np.all(np.diff(table["coord_id"].astype(np.int64)) > 0)
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


@pytest.mark.gpu
@pytest.mark.parametrize("n", [64, 256])
def test_hip_device_edits(poly, port, n):
    """§8(f) row 2: Grid::InjectSurface (ball brush) / InjectMaterial executed on the resident grid by k_edit, BF_Empty by
    k_edit_flags — voxel data, flags, modified box, rebuilt block ids and the surface after every edit of a chain."""
    from test_emu import check_device_edits
    check_device_edits(poly, port, n, 23, surface_tol=NRM_TOL)


@pytest.mark.gpu
def test_hip_brushes_anywhere_touch_the_reference_s_blocks(poly, port):
    """Brushes inside, across and outside the grid (tests/test_emu.py has the same list on the emulation): the touched blocks'
    ids are written on the device (k_box_ids), the grid afterwards is the reference's byte for byte."""
    from test_emu import check_brushes_anywhere
    check_brushes_anywhere(poly, port)


@pytest.mark.gpu
def test_hip_pool_compaction(poly, port):
    """vx_compact_pools after incremental runs: live meshes packed on the device, downloads unchanged."""
    from test_emu import check_compaction
    check_compaction(poly, port, 128)


@pytest.mark.gpu
@pytest.mark.parametrize("n,noisy", [(128, False), (64, True)])
def test_hip_grid_pack(poly, port, n, noisy):
    """§8(f) row 1, encode half: k_encode_grid writes the reference's file byte for byte."""
    from test_emu import check_pack
    check_pack(poly, port, n, 43, noisy)


@pytest.mark.gpu
@pytest.mark.parametrize("n", [64, 256])
def test_hip_create_heightmap(poly, port, n):
    """§8(f) row 3: the reference's height-map constructor evaluated by k_heightmap (+ BF_Empty by k_edit_flags)."""
    from test_emu import check_heightmap
    check_heightmap(poly, port, n, 9, nrm_tol=NRM_TOL)


@pytest.mark.gpu
def test_hip_zero_sample_grid_behind_a_clean_one(port):
    """A height map's surface blocks all hold exact zeros: the table-driven passes hand every one of them to the general
    pass.  Behind a run that handed on nothing (which sizes the general workgroups of the next run at two per pass) the list
    workgroups of k_tail once gave up waiting for them ("a dependency wait ... timed out"): their patience is in proportion
    to the work handed on now, and a changed grid launches the general passes at full width."""
    from voxels_amd import Polygonizer
    q = Polygonizer(device=0)
    q.set_materials(vxo.default_lut())
    try:
        q.create_terrain(256, 7)
        for _ in range(2):
            q.execute()   # nothing handed on: the hint for the next run is zero
        n = 256
        base = fields.smooth_noise(n, 5, scale=n // 4, amp=0.2 * n, octaves=3)[0]
        hm = np.clip(np.round(base) + (n // 2 - 127), -128, 127).astype(np.int8)
        g = port.grid_from_heightmap(n, hm)
        q.create_heightmap(hm)
        for _ in range(2):
            q.execute()
            ok, msg = fields.surface_equal(q.all_levels(), port.execute(g).all_levels(), nrm_tol=NRM_TOL)
            assert ok, msg
    finally:
        q.close()


@pytest.mark.gpu
def test_hip_device_only_pipeline_512(poly, port):
    """Everything the widened path offers, chained on a grid that never exists on the host in dense form: height-map
    constructor -> polygonize -> write file -> ball edit -> incremental polygonize -> compact -> write file; against the
    oracle doing the same with Grid::Create / Execute / PackForSave / InjectSurface."""
    n = 512
    base = fields.smooth_noise(n, 77, scale=n // 4, amp=0.2 * n, octaves=3)[0]
    hm = np.clip(np.round(base) + (n // 2 - 127), -128, 127).astype(np.int8)
    g = port.grid_from_heightmap(n, hm)
    poly.create_heightmap(hm)
    poly.execute()
    s = port.execute(g)
    ok, msg = fields.surface_equal(poly.all_levels(), s.all_levels(), nrm_tol=NRM_TOL)
    assert ok, msg
    assert np.array_equal(poly.pack(), g.pack())
    zs = int(hm[n // 2, n // 2]) + 127
    pos, ext, r = (n / 2.0, n / 2.0, float(zs)), (44.0, 44.0, 44.0), 20.0
    mn, mx = g.inject_ball(pos, ext, r, 2)
    mn2, mx2 = poly.inject_ball(pos, ext, r, 2)
    assert np.array_equal(mn, mn2) and np.array_equal(mx, mx2)
    ref_ids = port.execute_modify(g, s, mn, mx)
    got = poly.execute_dirty(mn2, mx2)
    assert np.array_equal(got, ref_ids)
    poly.compact_pools()
    ok, msg = fields.surface_equal(poly.all_levels(), s.all_levels(), nrm_tol=NRM_TOL)
    assert ok, msg
    assert np.array_equal(poly.stats(), s.stats())
    assert np.array_equal(poly.pack(), g.pack())


@pytest.mark.gpu
def test_hip_single_block_grid(poly, port):
    """16^3: one block, one level, no neighbours anywhere (every tile, slab and class array at its minimum size)."""
    rng = np.random.RandomState(16)
    d = np.clip(np.round(fields.smooth_noise(16, 3, scale=6, amp=3.0)), -4, 4).astype(np.int8)
    m = rng.randint(0, 3, (16, 16, 16)).astype(np.uint8)
    b = rng.randint(0, 256, (16, 16, 16)).astype(np.uint8)
    check_against(poly, port, d, m, b, "16^3")
    g = port.grid_from_dense(d, m, b)
    poly.upload_packed(g.pack())
    assert np.array_equal(poly.pack(), g.pack())


@pytest.mark.gpu
def test_hip_brick_mirrors_follow_the_grid(port):
    """The gathers of a run read brick-ordered mirrors of the fields; the mirrors have to follow every way a
    grid changes between two runs of ONE context: a second upload, host-edited blocks (vx_grid_update_blocks) followed by a
    FULL run, and attached memory that the caller rewrote and attached again."""
    import torch
    from voxels_amd import Polygonizer, synth
    from voxels_amd.slab import SlabBuffers
    n, levels = 128, 3
    p = Polygonizer(device=0)
    p.set_materials(vxo.default_lut())

    def expect(d, m, b):
        return port.execute(port.grid_from_dense(d, m, b)).all_levels()[:levels]

    def check(tag, want):
        p.execute(levels)
        ok, msg = fields.surface_equal(p.all_levels(), want, nrm_tol=NRM_TOL)
        assert ok, tag + ": " + msg

    a = synth.terrain(n, 0, n, 3)
    b2 = synth.terrain(n, 0, n, 4)
    p.upload(*a, synth.block_empty_flags(a[0]))
    check("first upload", expect(*a))
    p.upload(*b2, synth.block_empty_flags(b2[0]))
    check("second upload into the same context", expect(*b2))
    # host-edited blocks: a cube of blocks takes the other terrain's voxels
    d, m, bl = (x.copy() for x in b2)
    nb = n // 16
    ids = [(z * nb + y) * nb + x for z in range(2, 5) for y in range(1, 4) for x in range(3, 6)]
    blocks = [np.zeros((len(ids), 16, 16, 16), t) for t in (np.int8, np.uint8, np.uint8)]
    for i, bid in enumerate(ids):
        bx, by, bz = bid % nb, (bid // nb) % nb, bid // (nb * nb)
        sl = (slice(bz * 16, bz * 16 + 16), slice(by * 16, by * 16 + 16), slice(bx * 16, bx * 16 + 16))
        for dst, src, blk in zip((d, m, bl), a, blocks):
            dst[sl] = src[sl]
            blk[i] = src[sl]
    p.update_blocks(np.array(ids, np.uint32), blocks[0], blocks[1], blocks[2], synth.block_empty_flags(d))
    check("vx_grid_update_blocks + full run", expect(d, m, bl))
    # attached memory rewritten by the caller, attached again
    dev = torch.device("cuda", 0)
    slab = SlabBuffers(torch, n, 0, 1, dev, axis="z")
    slab.fill_from_full(*a, synth.block_empty_flags(a[0]))
    torch.cuda.synchronize()
    slab.attach(p)
    check("attached", expect(*a))
    slab.fill_from_full(*b2, synth.block_empty_flags(b2[0]))
    torch.cuda.synchronize()
    slab.attach(p)
    check("rewritten and attached again", expect(*b2))
    p.close()

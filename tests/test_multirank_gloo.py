"""world_size-2 CPU test (gloo) of the multi-GPU path: z-slab sharding + halo exchange (voxels_amd/slab.py) +
per-rank polygonization through vx_grid_attach, merged and compared with the ORACLE's surface of the whole grid (the
restatement pinned against the unmodified reference, oracle/port.cpp) — and, for the statistics of a level-limited run,
which the reference cannot produce, with a single-rank run of the same library."""
import os
import subprocess
import sys

import numpy as np

import fields
import vxo
from emu_lib import emu_library

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


import pytest


@pytest.mark.parametrize("axis,port", [("z", "29611"), ("y", "29613")])
def test_two_ranks_equal_one(tmp_path, axis, port):
    from voxels_amd import synth
    from voxels_amd.binding import Level, Polygonizer
    from voxels_amd.slab import merge_rank_levels
    n, levels, world = 128, 3, 2
    env = dict(os.environ, MASTER_ADDR="127.0.0.1", MASTER_PORT=port, WORLD_SIZE=str(world), OMP_NUM_THREADS="2")
    procs = []
    for r in range(world):
        e = dict(env, RANK=str(r), LOCAL_RANK=str(r))
        procs.append(subprocess.Popen([sys.executable, os.path.join(ROOT, "tests", "mr_worker.py"), str(tmp_path), str(n), str(levels), axis], env=e))
    for p in procs:
        assert p.wait(timeout=600) == 0
    parts, stats, digest_sums, coarse, digest_all = [], np.zeros(20, np.uint64), [], [], None
    for r in range(world):
        z = np.load(os.path.join(str(tmp_path), "rank%d.npz" % r))
        digest_sums.append(z["digest_sum"])
        parts.append([Level(z["L%d_infos" % l], z["L%d_verts" % l], z["L%d_idx" % l], z["L%d_tverts" % l], z["L%d_tidx" % l]) for l in range(levels)])
        stats += z["stats"]
        if int(z["coarse_count"]):
            assert r == 0 and int(z["coarse_first"]) == levels
            coarse = [Level(z["L%d_infos" % l], z["L%d_verts" % l], z["L%d_idx" % l], z["L%d_tverts" % l], z["L%d_tidx" % l]) for l in range(levels, levels + int(z["coarse_count"]))]
            digest_all = z["digest_all"]
    d, m, b = synth.terrain(n, seed=5)
    whole = Polygonizer(library=emu_library())
    whole.set_materials(vxo.default_lut())
    whole.upload(d, m, b, synth.block_empty_flags(d))
    whole.execute(levels)
    merged = merge_rank_levels(parts)
    oracle = vxo.load_port()
    assert oracle is not None, "oracle/libvoxels_port.so missing (run __graft_entry__.build())"
    ref = oracle.execute(oracle.grid_from_dense(d, m, b)).all_levels()[:levels]
    ok, msg = fields.surface_equal(merged, ref)
    assert ok, "2 ranks vs oracle: " + msg
    ok, msg = fields.surface_equal(merged, whole.all_levels())
    assert ok, msg
    # the correctness bit of bench.py --gpus N: the all-reduced digest equals the digest of the whole surface
    from voxels_amd import digest
    for ds in digest_sums:
        assert digest.digests_equal(digest.unpack(ds, levels), digest.surface_digest(ref)), "all-reduced digest vs oracle"
    assert np.array_equal(stats.astype(np.uint32), whole.stats())
    # EVERY level the reference produces: the slabs' levels 0..2 and, from rank 0's CoarseLevels (the
    # gathered fields, vx_polygonize_from), level 3 - whose one block spans both slabs - against the oracle's whole surface
    ref_all = oracle.execute(oracle.grid_from_dense(d, m, b)).all_levels()
    assert len(ref_all) == levels + 1 and len(coarse) == 1
    ok, msg = fields.surface_equal(merged + coarse, ref_all)
    assert ok, "2 ranks + coarse levels vs oracle: " + msg
    assert digest.digests_equal(digest.unpack(digest_all, levels + 1), digest.surface_digest(ref_all)), "digest of all levels vs oracle"

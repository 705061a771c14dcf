"""Two ranks on two GPUs through the C ABI (row b8): vx_grid_fill_terrain + vx_comm_init + vx_halo_exchange over RCCL, one
process per GPU, merged and compared with the ORACLE's surface of the whole grid (oracle/port.cpp, pinned against the
unmodified reference) - and the all-reduced digest that bench.py --gpus N prints as its correctness bit with the oracle's.  Needs two GPUs (skipped otherwise: RCCL refuses
two ranks on one device); the same path with in-process transport runs on one GPU in
tests/test_gpu_parity.py::test_hip_halo_exchange_group."""
import os
import subprocess
import sys

import numpy as np
import pytest

import fields

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.mark.gpu
@pytest.mark.parametrize("axis,port", [("y", "29621"), ("z", "29623")])
def test_two_gpus_rccl_equal_one(tmp_path, axis, port):
    import torch
    if torch.cuda.device_count() < 2:
        pytest.skip("needs two GPUs")
    from voxels_amd import Polygonizer, synth
    from voxels_amd.binding import Level
    from voxels_amd.slab import merge_rank_levels
    n, levels, world = 256, 3, 2
    env = dict(os.environ, MASTER_ADDR="127.0.0.1", MASTER_PORT=port, WORLD_SIZE=str(world), HSA_ENABLE_IPC_MODE_LEGACY="0")
    procs = []
    for r in range(world):
        e = dict(env, RANK=str(r), LOCAL_RANK=str(r))
        procs.append(subprocess.Popen([sys.executable, os.path.join(ROOT, "tests", "mr_worker.py"), str(tmp_path), str(n), str(levels), axis, "rccl"], env=e))
    for p in procs:
        assert p.wait(timeout=900) == 0
    parts = []
    for r in range(world):
        z = np.load(os.path.join(str(tmp_path), "rank%d.npz" % r))
        parts.append([Level(z["L%d_infos" % l], z["L%d_verts" % l], z["L%d_idx" % l], z["L%d_tverts" % l], z["L%d_tidx" % l]) for l in range(levels)])
    import vxo
    port = vxo.load_port()
    assert port is not None, "oracle/libvoxels_port.so missing (run __graft_entry__.build())"
    d, m, b = synth.terrain(n, seed=5)
    ref = port.execute(port.grid_from_dense(d, m, b)).all_levels()[:levels]
    merged = merge_rank_levels(parts)
    ok, msg = fields.surface_equal(merged, ref, nrm_tol=1e-5)
    assert ok, "2 GPUs over RCCL vs oracle: " + msg
    from voxels_amd import digest
    for r in range(world):
        ds = np.load(os.path.join(str(tmp_path), "rank%d.npz" % r))["digest_sum"]
        assert digest.digests_equal(digest.unpack(ds, levels), digest.surface_digest(ref)), "all-reduced digest vs oracle"

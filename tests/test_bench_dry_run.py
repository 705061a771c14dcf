"""bench.py's own control flow for N > 1 on CPU: `bench.py --gpus 2 --dry-run-cpu` under gloo with the
CPU emulation of the device phases named through VOXELS_HIP_LIBRARY (the script itself never refers to it) - rank
bookkeeping, the transport negotiation (the emulation has no RCCL: the torch.distributed fallback and what the line says
about it), the halo exchange, the timed loop with its barriers and MAX over ranks, the digest all-reduce, the coarse
levels on rank 0 (CoarseLevels) and the JSON line - so that the first launch on a real 8-GPU node cannot die in Python
nobody ran.  The numbers of such a line mean nothing; its structure and its self-check do."""
import json
import os
import subprocess
import sys

import pytest

from emu_lib import emu_library

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.mark.parametrize("axis,port", [("y", "29621"), ("z", "29623")])
def test_bench_two_ranks_on_cpu(axis, port):
    from voxels_amd import build
    emu_library()  # (built; the path goes to the ranks through the environment)
    path = build.build_emu()
    env = dict(os.environ, MASTER_ADDR="127.0.0.1", MASTER_PORT=port, WORLD_SIZE="2", OMP_NUM_THREADS="2", VOXELS_HIP_LIBRARY=path)
    cmd = [sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "2", "--warmup", "1", "--n", "128", "--levels", "3",
           "--slab-axis", axis, "--dry-run-cpu", "--no-cpu-baseline", "--no-extra", "--no-isolated"]
    procs = [subprocess.Popen(cmd, env=dict(env, RANK=str(r), LOCAL_RANK=str(r)), stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True) for r in range(2)]
    outs = [p.communicate(timeout=900) for p in procs]
    for r, (p, (so, se)) in enumerate(zip(procs, outs)):
        assert p.returncode == 0, "rank %d failed:\n%s" % (r, se[-3000:])
    lines = [l for l in outs[0][0].splitlines() if l.startswith("{")]
    assert len(lines) == 1 and not [l for l in outs[1][0].splitlines() if l.startswith("{")], "rank 0 prints ONE JSON line, rank 1 none"
    line = json.loads(lines[0])
    assert line["n_gpus"] == 2 and line["steps"] == 2 and line["warmup"] == 1 and line["scaling"] == "strong"
    cfg = line["config"]
    assert cfg["multi_gpu_check"] == "equal"
    assert cfg["halo_transport"] == "torch-distributed" and cfg["c_abi_rccl_error"], "the emulation has no RCCL: the fallback transport, and the line says so"
    detail = cfg["multi_gpu_check_detail"]
    # 128^3 on two ranks: the slabs run the levels 0..2, level 3 (one block across both slabs) is rank 0's
    assert detail["coarse_levels"]["levels"] == [3, 4] and len(detail["totals_per_level_blocks_verts_indices_tverts_tindices"]) == 4
    assert line["dry_run"] is True

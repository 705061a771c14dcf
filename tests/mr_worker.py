"""Worker of tests/test_multirank_gloo.py: one rank of a world_size-N run of the slab path on CPU (gloo + the
emulation backend), mirroring what bench.py does with RCCL on GPUs."""
import os
import sys

import numpy as np
import torch
import torch.distributed as dist

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))


def main():
    out_dir, n, levels = sys.argv[1], int(sys.argv[2]), int(sys.argv[3])
    axis = sys.argv[4] if len(sys.argv) > 4 else "z"
    rank, world = int(os.environ["RANK"]), int(os.environ["WORLD_SIZE"])
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from emu_lib import emu_library
    from voxels_amd import synth
    from voxels_amd.binding import Polygonizer
    from voxels_amd.slab import SlabBuffers
    import vxo

    slab = SlabBuffers(torch, n, rank, world, torch.device("cpu"), axis=axis)
    if axis == "z":
        d, m, b = synth.terrain(n, slab.z0, slab.z1, seed=5)
        slab.fill_own(d, m, b, synth.block_empty_flags(d))
    else:  # own rows of every plane; the flags of all blocks are derived locally (own blocks interleave in id order)
        d, m, b = synth.terrain(n, seed=5)
        sl = slice(slab.z0, slab.z1)
        slab.fill_own(np.ascontiguousarray(d[:, sl]), np.ascontiguousarray(m[:, sl]), np.ascontiguousarray(b[:, sl]), synth.block_empty_flags(d))
    slab.gather_flags(dist)
    slab.halo_exchange(dist)
    p = Polygonizer(library=emu_library())
    p.set_materials(vxo.default_lut())
    slab.attach(p)
    p.execute(levels)
    out = {"stats": p.stats()}
    for li, lv in enumerate(p.all_levels()):
        out["L%d_infos" % li] = lv.infos
        out["L%d_verts" % li] = lv.verts
        out["L%d_idx" % li] = lv.idx
        out["L%d_tverts" % li] = lv.tverts
        out["L%d_tidx" % li] = lv.tidx
    np.savez(os.path.join(out_dir, "rank%d.npz" % rank), **out)
    dist.barrier()
    dist.destroy_process_group()


if __name__ == "__main__":
    main()

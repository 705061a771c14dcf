"""Worker of tests/test_multirank_gloo.py and tests/test_multirank_rccl.py: one rank of a world_size-N run of the slab path —
on CPU (gloo + the emulation backend, halo exchange by voxels_amd/slab.py) or, with a fifth argument "rccl", on GPU
`LOCAL_RANK` through the C ABI exactly like bench.py: vx_grid_fill_terrain, vx_comm_init, vx_halo_exchange."""
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
    if len(sys.argv) > 5 and sys.argv[5] == "rccl":
        return main_rccl(out_dir, n, levels, axis, rank, world)
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
    # the levels whose block is larger than a slab: on rank 0, from the gathered fields (voxels_amd/slab.py CoarseLevels)
    from voxels_amd.slab import CoarseLevels

    def make():
        q = Polygonizer(library=emu_library())
        q.set_materials(vxo.default_lut())
        return q
    coarse = CoarseLevels(slab, dist, make)
    save(out_dir, rank, p, coarse.execute(), coarse.first)
    coarse.close()
    dist.barrier()
    dist.destroy_process_group()


def save(out_dir, rank, p, coarse=None, coarse_first=0):
    """the rank's levels + the digest of the whole sharded run as bench.py --gpus N forms it: every rank's digest
    (voxels_amd/digest.py), packed and summed with all_reduce"""
    from voxels_amd import digest
    all_levels = p.all_levels()
    packed = torch.from_numpy(digest.pack(digest.surface_digest(all_levels)))
    dist.all_reduce(packed, op=dist.ReduceOp.SUM)
    out = {"stats": p.stats(), "digest_sum": packed.numpy()}
    named = list(enumerate(all_levels)) + [(coarse_first + i, lv) for i, lv in enumerate(coarse or [])]
    out["coarse_first"], out["coarse_count"] = coarse_first, len(coarse or [])
    if coarse:
        # what bench.py --gpus N compares: the ranks' summed digest joined with rank 0's digest of the coarse levels
        out["digest_all"] = digest.pack(digest.join(digest.unpack(packed.numpy(), len(all_levels)), digest.surface_digest(coarse, first_level=coarse_first)))
    for li, lv in named:
        out["L%d_infos" % li] = lv.infos
        out["L%d_verts" % li] = lv.verts
        out["L%d_idx" % li] = lv.idx
        out["L%d_tverts" % li] = lv.tverts
        out["L%d_tidx" % li] = lv.tidx
    np.savez(os.path.join(out_dir, "rank%d.npz" % rank), **out)


def main_rccl(out_dir, n, levels, axis, rank, world):
    from voxels_amd import Polygonizer, synth
    from voxels_amd.slab import SlabBuffers
    local = int(os.environ.get("LOCAL_RANK", rank))
    torch.cuda.set_device(local)
    dev = torch.device("cuda", local)
    slab = SlabBuffers(torch, n, rank, world, dev, axis=axis)
    p = Polygonizer(device=local)
    p.set_materials(synth.default_lut())
    slab.attach(p)
    p.fill_terrain(5)                      # own layers + halo + own flags; the neighbours' flag layers are still missing
    uid = torch.from_numpy(p.comm_unique_id() if rank == 0 else np.zeros(128, np.uint8))
    dist.broadcast(uid, 0)                 # over gloo: the id only has to reach every rank somehow
    p.comm_init(world, rank, uid.numpy())
    # wipe the halo so that the exchange has to deliver it
    if axis == "z":
        slab.dist[0].zero_(); slab.dist[-2:].zero_(); slab.mat[-1].zero_(); slab.blend[-1].zero_()
    else:
        slab.dist[:, 0].zero_(); slab.dist[:, -2:].zero_(); slab.mat[:, -1].zero_(); slab.blend[:, -1].zero_()
    torch.cuda.synchronize()
    p.invalidate()                         # the attached tensors were rewritten behind the library's back
    p.halo_exchange()
    p.execute(levels)
    from voxels_amd.slab import CoarseLevels

    def make():
        q = Polygonizer(device=local)
        q.set_materials(synth.default_lut())
        return q
    coarse = CoarseLevels(slab, dist, make)  # (the gather moves device tensors over gloo: staged through the host by torch)
    save(out_dir, rank, p, coarse.execute(), coarse.first)
    coarse.close()
    dist.barrier()
    dist.destroy_process_group()


if __name__ == "__main__":
    main()

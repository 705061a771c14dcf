"""Z-slab sharding of the voxel grid over the ranks of one node and the halo exchange the path needs.

Blocks of every LOD level are owned by exactly one rank when the slab thickness is a multiple of the coarsest
block size (16 << (levels-1)).  Beyond its own planes [z0, z1) a rank reads:
  * distance plane z0-1            (level-0 central-difference normals of vertices on plane z0)
  * distance planes z1 and z1+1    (far corners of the last cell layer; normals of vertices on plane z1)
  * material and blend plane z1    (materials of vertices on plane z1)
Transition faces only use samples ON the slab boundary plane, so they add nothing.  The exchange is one grouped
send/recv pair per neighbour (RCCL over xGMI on the GPUs, gloo in the CPU tests)."""
import numpy as np


class SlabBuffers:
    """Device (or host) tensors holding a rank's slab plus halo planes.

    dist  [planes + 3, n, n]  plane 0 <-> global z0 - 1
    mat   [planes + 1, n, n]  plane 0 <-> global z0
    blend [planes + 1, n, n]
    """

    def __init__(self, torch, n, rank, world, device):
        assert n % world == 0
        self.torch, self.n, self.rank, self.world = torch, n, rank, world
        self.planes = n // world
        self.z0, self.z1 = rank * self.planes, (rank + 1) * self.planes
        self.dist = torch.zeros((self.planes + 3, n, n), dtype=torch.int8, device=device)
        self.mat = torch.zeros((self.planes + 1, n, n), dtype=torch.uint8, device=device)
        self.blend = torch.zeros((self.planes + 1, n, n), dtype=torch.uint8, device=device)
        self.flags = torch.zeros(((n // 16) ** 3,), dtype=torch.uint8, device=device)

    def fill_own(self, d, m, b, flags_own):
        t = self.torch
        self.dist[1:self.planes + 1].copy_(t.from_numpy(d))
        self.mat[:self.planes].copy_(t.from_numpy(m))
        self.blend[:self.planes].copy_(t.from_numpy(b))
        per = flags_own.size
        self.flags[self.rank * per:(self.rank + 1) * per].copy_(t.from_numpy(flags_own))

    def gather_flags(self, dist_pkg):
        """Every rank needs the BF_Empty flags of the neighbouring slabs' boundary block layers; they are tiny, so
        all ranks simply gather the whole array."""
        if self.world == 1:
            return
        per = self.flags.numel() // self.world
        mine = self.flags[self.rank * per:(self.rank + 1) * per].clone()
        chunks = [self.torch.empty_like(mine) for _ in range(self.world)]
        dist_pkg.all_gather(chunks, mine)
        self.flags.copy_(self.torch.cat(chunks))

    def halo_exchange(self, dist_pkg):
        if self.world == 1:
            return
        r, w, p = self.rank, self.world, self.planes
        ops = []
        if r > 0:
            ops.append(dist_pkg.P2POp(dist_pkg.isend, self.dist[1:3], r - 1))
            ops.append(dist_pkg.P2POp(dist_pkg.isend, self.mat[0:1], r - 1))
            ops.append(dist_pkg.P2POp(dist_pkg.isend, self.blend[0:1], r - 1))
            ops.append(dist_pkg.P2POp(dist_pkg.irecv, self.dist[0:1], r - 1))
        if r < w - 1:
            ops.append(dist_pkg.P2POp(dist_pkg.irecv, self.dist[p + 1:p + 3], r + 1))
            ops.append(dist_pkg.P2POp(dist_pkg.irecv, self.mat[p:p + 1], r + 1))
            ops.append(dist_pkg.P2POp(dist_pkg.irecv, self.blend[p:p + 1], r + 1))
            ops.append(dist_pkg.P2POp(dist_pkg.isend, self.dist[p:p + 1], r + 1))
        for work in dist_pkg.batch_isend_irecv(ops):
            work.wait()

    def attach(self, poly):
        poly.attach(self.n, self.z0, self.z1, self.dist.data_ptr(), self.z0 - 1, self.mat.data_ptr(),
                    self.blend.data_ptr(), self.z0, self.flags.data_ptr())


def merge_rank_levels(per_rank_levels):
    """Concatenate per-rank Level lists (rank order = block-id order because slabs are z-major)."""
    from .binding import Level
    out = []
    for l in range(len(per_rank_levels[0])):
        parts = [r[l] for r in per_rank_levels]
        out.append(Level(np.concatenate([p.infos for p in parts]), np.concatenate([p.verts for p in parts]),
                         np.concatenate([p.idx for p in parts]), np.concatenate([p.tverts for p in parts]),
                         np.concatenate([p.tidx for p in parts])))
    return out

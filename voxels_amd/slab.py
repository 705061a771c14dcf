"""Slab sharding of the voxel grid over the ranks of one node, the halo exchange the path needs, and the coarse levels a
sharded run leaves to one rank (CoarseLevels).

Blocks of every LOD level are owned by exactly one rank when the slab thickness is a multiple of the coarsest
block size (16 << (levels-1)).  Beyond its own planes [z0, z1) a rank reads:
  * distance plane z0-1            (level-0 central-difference normals of vertices on plane z0)
  * distance planes z1 and z1+1    (far corners of the last cell layer; normals of vertices on plane z1)
  * material and blend plane z1    (materials of vertices on plane z1)
Transition faces only use samples ON the slab boundary plane, so they add nothing.  The exchange is one grouped
send/recv pair per neighbour (RCCL over xGMI on the GPUs, gloo in the CPU tests)."""
import numpy as np


class SlabBuffers:
    """Device (or host) tensors holding a rank's slab plus halo.

    axis "z" (slabs of whole z-planes):              axis "y" (slabs of rows of every z-plane):
      dist  [planes + 3, n, n]  plane 0 <-> z0 - 1      dist  [n, planes + 3, n]  row 0 <-> y0 - 1
      mat   [planes + 1, n, n]  plane 0 <-> z0          mat   [n, planes + 1, n]  row 0 <-> y0
      blend like mat                                    blend like mat
    A height-field terrain keeps nearly all of its surface in a few z-layers, so z-slabs give most ranks nothing to do;
    y-slabs cut across the surface and balance.  The halo requirement is the same along either axis.
    """

    def __init__(self, torch, n, rank, world, device, axis="z"):
        assert n % world == 0 and axis in ("z", "y")
        self.torch, self.n, self.rank, self.world, self.axis = torch, n, rank, world, axis
        self.planes = n // world
        self.z0, self.z1 = rank * self.planes, (rank + 1) * self.planes  # owned range along the slab axis
        p = self.planes
        dshape = (p + 3, n, n) if axis == "z" else (n, p + 3, n)
        mshape = (p + 1, n, n) if axis == "z" else (n, p + 1, n)
        self.dist = torch.zeros(dshape, dtype=torch.int8, device=device)
        self.mat = torch.zeros(mshape, dtype=torch.uint8, device=device)
        self.blend = torch.zeros(mshape, dtype=torch.uint8, device=device)
        self.flags = torch.zeros(((n // 16) ** 3,), dtype=torch.uint8, device=device)

    def _d(self, a, b):
        """slice [a, b) of the distance tensor along the slab axis"""
        return self.dist[a:b] if self.axis == "z" else self.dist[:, a:b]

    def _m(self, t, a, b):
        return t[a:b] if self.axis == "z" else t[:, a:b]

    def fill_own(self, d, m, b, flags_own):
        """d, m, b: this rank's own part ([planes, n, n] for z-slabs, [n, planes, n] for y-slabs); flags_own: the flags of
        the rank's own blocks in block-id order (z-slabs) or the full flag array (y-slabs, where own blocks interleave)."""
        t = self.torch
        self._d(1, self.planes + 1).copy_(t.from_numpy(d))
        self._m(self.mat, 0, self.planes).copy_(t.from_numpy(m))
        self._m(self.blend, 0, self.planes).copy_(t.from_numpy(b))
        if self.axis == "z":
            per = flags_own.size
            self.flags[self.rank * per:(self.rank + 1) * per].copy_(t.from_numpy(flags_own))
        else:
            self.flags.copy_(t.from_numpy(np.ascontiguousarray(flags_own)))

    def fill_from_full(self, d, m, b, flags):
        """everything (own part + halo + all flags) from whole-grid host arrays — no exchange needed afterwards"""
        t, n, p = self.torch, self.n, self.planes
        lo, hi = max(self.z0 - 1, 0), min(self.z1 + 2, n)
        hm = min(self.z1 + 1, n)
        if self.axis == "z":
            self.dist[lo - (self.z0 - 1):hi - (self.z0 - 1)].copy_(t.from_numpy(d[lo:hi]))
            self.mat[:hm - self.z0].copy_(t.from_numpy(m[self.z0:hm]))
            self.blend[:hm - self.z0].copy_(t.from_numpy(b[self.z0:hm]))
        else:
            self.dist[:, lo - (self.z0 - 1):hi - (self.z0 - 1)].copy_(t.from_numpy(np.ascontiguousarray(d[:, lo:hi])))
            self.mat[:, :hm - self.z0].copy_(t.from_numpy(np.ascontiguousarray(m[:, self.z0:hm])))
            self.blend[:, :hm - self.z0].copy_(t.from_numpy(np.ascontiguousarray(b[:, self.z0:hm])))
        self.flags.copy_(t.from_numpy(np.ascontiguousarray(flags, np.uint8)))

    def gather_flags(self, dist_pkg):
        """Every rank needs the BF_Empty flags of the neighbouring slabs' boundary block layers; they are tiny, so
        all ranks simply gather the whole array (z-slabs; with y-slabs fill_own already takes the full array)."""
        if self.world == 1 or self.axis != "z":
            return
        per = self.flags.numel() // self.world
        mine = self.flags[self.rank * per:(self.rank + 1) * per].clone()
        chunks = [self.torch.empty_like(mine) for _ in range(self.world)]
        dist_pkg.all_gather(chunks, mine)
        self.flags.copy_(self.torch.cat(chunks))

    def halo_exchange(self, dist_pkg):
        """1 distance layer from the slab below; 2 distance layers + 1 material + 1 blend layer from the slab above (a layer
        = a z-plane or a y-row of every plane).  One grouped send/recv batch; strided row slices travel through
        contiguous staging tensors.  (torch.distributed transport, used by the CPU tests; the product path is
        Polygonizer.halo_exchange = vx_halo_exchange.  It rewrites the attached tensors behind the library's back: call
        attach() again before the next polygonization.)"""
        if self.world == 1:
            return
        r, w, p = self.rank, self.world, self.planes
        contiguous = self.axis == "z"
        ops, landings = [], []

        def send(x, peer):
            ops.append(dist_pkg.P2POp(dist_pkg.isend, x if contiguous else x.contiguous(), peer))

        def recv(x, peer):
            if contiguous:
                ops.append(dist_pkg.P2POp(dist_pkg.irecv, x, peer))
            else:
                tmp = self.torch.empty(x.shape, dtype=x.dtype, device=x.device)
                landings.append((x, tmp))
                ops.append(dist_pkg.P2POp(dist_pkg.irecv, tmp, peer))

        if r > 0:
            send(self._d(1, 3), r - 1)
            send(self._m(self.mat, 0, 1), r - 1)
            send(self._m(self.blend, 0, 1), r - 1)
            recv(self._d(0, 1), r - 1)
        if r < w - 1:
            recv(self._d(p + 1, p + 3), r + 1)
            recv(self._m(self.mat, p, p + 1), r + 1)
            recv(self._m(self.blend, p, p + 1), r + 1)
            send(self._d(p, p + 1), r + 1)
        for work in dist_pkg.batch_isend_irecv(ops):
            work.wait()
        for dst, tmp in landings:
            dst.copy_(tmp)

    def attach(self, poly):
        if self.axis == "z":
            poly.attach(self.n, self.z0, self.z1, self.dist.data_ptr(), self.z0 - 1, self.mat.data_ptr(),
                        self.blend.data_ptr(), self.z0, self.flags.data_ptr())
        else:
            poly.attach_y(self.n, self.z0, self.z1, self.dist.data_ptr(), self.z0 - 1, self.planes + 3, self.mat.data_ptr(),
                          self.blend.data_ptr(), self.z0, self.planes + 1, self.flags.data_ptr())


def sharded_levels(n, world):
    """How many levels a sharded run of `world` slabs produces itself: those whose block (16 << level voxels) fits the slab,
    so that every block has one owner.  The reference always produces log2(n / 16) + 1 (src/TransVoxelImpl.cpp:490-492); the
    rest are coarse_levels()'s."""
    planes, k = n // world, 0
    while (16 << k) <= planes and (16 << k) <= n:
        k += 1
    return k


def all_levels_count(n):
    k = 1
    while (16 << k) <= n:
        k += 1
    return k


def gather_whole_fields(slab, dist_pkg, dst=0):
    """Every rank's OWN layers of the three fields, joined on rank `dst` into whole-grid tensors [n, n, n] (None on the other
    ranks), and the BF_Empty flags of all blocks (every rank contributes the flags it owns; the others are zero until a
    halo exchange fills the neighbours' in).  One gather per field, no host copy."""
    t, p, w, r = slab.torch, slab.planes, slab.world, slab.rank
    dim = 0 if slab.axis == "z" else 1
    out = []
    for own in (slab._d(1, p + 1), slab._m(slab.mat, 0, p), slab._m(slab.blend, 0, p)):
        own = own.contiguous()
        chunks = [t.empty_like(own) for _ in range(w)] if r == dst else None
        dist_pkg.gather(own, chunks, dst=dst)
        out.append(t.cat(chunks, dim=dim).contiguous() if r == dst else None)
        del chunks
    flags = slab.flags.clone()
    dist_pkg.all_reduce(flags, op=dist_pkg.ReduceOp.MAX)
    return out[0], out[1], out[2], flags


class CoarseLevels:
    """The levels a sharded run cannot produce slab by slab - those whose block is larger than a slab (1024^3 on 8 ranks:
    levels 4..6) - on ONE rank, so that a sharded run yields every level the reference does (last sentence).

    What such a level reads is not a coarse lattice alone: every vertex walks its LOD chain down to the level-0 edge that
    holds the crossing (FindBestVertexInLODChain, src/TransVoxelImpl.cpp:1484-1509) and takes its normals and materials
    from the level-0 voxels around that edge (:1239-1246, :1698-1703), its cell materials are votes over the caches of
    the level below (:753-838), and the transition cells of level L read the level L-1 lattice - so the 64^3-strided
    sample set SURVEY names is not enough; the crossings can lie anywhere.  The rank therefore gathers the owned layers of
    the three fields once per grid change (3 B per voxel over the links: 384 MB per sending rank at 1024^3 / 8, a few
    milliseconds over xGMI; never part of a timed step of an unchanged grid), attaches them to a context of its own and
    runs vx_polygonize_from(first level = sharded_levels): slot maps, bitmaps and material caches of every level, meshes of
    the coarse ones only (0.15 ms at 1024^3).  Costs that rank 6 B per voxel of device memory (fields + mirrors)."""

    def __init__(self, slab, dist_pkg, make_polygonizer, dst=0):
        self.slab, self.dst = slab, dst
        self.first = sharded_levels(slab.n, slab.world)
        self.levels = all_levels_count(slab.n)
        self.poly = None
        self.fields = None
        if self.first >= self.levels:
            return  # (a single slab, or slabs as thick as the grid's coarsest block: nothing is missing)
        d, m, b, flags = gather_whole_fields(slab, dist_pkg, dst)
        if slab.rank == dst:
            self.fields = (d, m, b, flags)  # (attached memory: must outlive the context)
            self.poly = make_polygonizer()
            self.poly.attach(slab.n, 0, slab.n, d.data_ptr(), 0, m.data_ptr(), b.data_ptr(), 0, flags.data_ptr())

    @classmethod
    def from_slabs(cls, slabs, make_polygonizer):
        """The same for all slabs of a grid held by ONE process (several contexts on one device: tests, tools): the whole-grid
        tensors are joined from the slabs' own layers directly."""
        t, first = slabs[0].torch, slabs[0]
        self = cls.__new__(cls)
        self.slab, self.dst = first, first.rank
        self.first, self.levels = sharded_levels(first.n, first.world), all_levels_count(first.n)
        self.poly = self.fields = None
        if self.first >= self.levels:
            return self
        p, dim = first.planes, 0 if first.axis == "z" else 1
        d = t.cat([x._d(1, p + 1) for x in slabs], dim=dim).contiguous()
        m = t.cat([x._m(x.mat, 0, p) for x in slabs], dim=dim).contiguous()
        b = t.cat([x._m(x.blend, 0, p) for x in slabs], dim=dim).contiguous()
        flags = slabs[0].flags.clone()
        for x in slabs[1:]:
            flags = t.maximum(flags, x.flags)
        self.fields = (d, m, b, flags)
        self.poly = make_polygonizer()
        self.poly.attach(first.n, 0, first.n, d.data_ptr(), 0, m.data_ptr(), b.data_ptr(), 0, flags.data_ptr())
        return self

    def execute(self):
        """-> the Level list of the coarse levels [first, levels) on rank dst (empty list where nothing is missing), None elsewhere"""
        if self.first >= self.levels:
            return [] if self.slab.rank == self.dst else None
        if self.poly is None:
            return None
        self.info = self.poly.execute_from(0, self.first)
        return self.poly.all_levels()[self.first:]

    def close(self):
        if self.poly is not None:
            self.poly.close()
        self.poly = self.fields = None


def merge_rank_levels(per_rank_levels):
    """One Level list from per-rank Level lists, blocks in block-id order (with z-slabs that is rank order; with
    y-slabs the ranks' blocks interleave)."""
    from .binding import Level
    out = []
    for l in range(len(per_rank_levels[0])):
        blocks = []  # (id, info, verts, idx, tverts, tidx)
        for r in per_rank_levels:
            lv = r[l]
            ov = oi = otv = oti = 0
            for info in lv.infos:
                nv, ni, ntv, nti = int(info["n_verts"]), int(info["n_idx"]), int(info["n_tverts"].sum()), int(info["n_tidx"].sum())
                blocks.append((int(info["id"]), info, lv.verts[ov:ov + nv], lv.idx[oi:oi + ni], lv.tverts[otv:otv + ntv], lv.tidx[oti:oti + nti]))
                ov += nv; oi += ni; otv += ntv; oti += nti
        blocks.sort(key=lambda t: t[0])
        first = per_rank_levels[0][l]
        cat = lambda k, proto: np.concatenate([b[k] for b in blocks]) if blocks else proto[:0]
        infos = np.array([b[1] for b in blocks], dtype=first.infos.dtype) if blocks else first.infos[:0]
        out.append(Level(infos, cat(2, first.verts), cat(3, first.idx), cat(4, first.tverts), cat(5, first.tidx)))
    return out

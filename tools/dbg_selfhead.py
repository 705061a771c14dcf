"""debug: per-block differences between the run with and without a classification pass (y-slabs, attached tensors)"""
import os, sys
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from voxels_amd import synth, binding
from voxels_amd.slab import SlabBuffers

n, levels, seed, world, axis = 256, 3, 11, 2, "y"
d, m, b = synth.terrain(n, 0, n, seed)
flags = synth.block_empty_flags(d)
dev = torch.device("cuda", 0)
res = {}
for mode in ("1", "0"):
    os.environ["VX_SELF_HEAD"] = mode
    poly = binding.Polygonizer()
    out = []
    for r in range(world):
        slab = SlabBuffers(torch, n, r, world, dev, axis=axis)
        slab.fill_from_full(d, m, b, flags)
        torch.cuda.synchronize()
        slab.attach(poly)
        poly.execute(levels)
        lv = poly.level(0, with_data=False)
        out.append(lv.infos.copy())
    res[mode] = out
for r in range(world):
    a, c = res["1"][r], res["0"][r]
    print("rank", r, "blocks", len(a), len(c), a.dtype.names)
    ka = {int(x[a.dtype.names[0]]): x for x in a}
    kc = {int(x[c.dtype.names[0]]): x for x in c}
    bad = 0
    for k in sorted(set(ka) | set(kc)):
        xa, xc = ka.get(k), kc.get(k)
        if xa is None or xc is None or (int(xa['n_verts']), int(xa['n_idx'])) != (int(xc['n_verts']), int(xc['n_idx'])):
            nb = n // 16
            if bad < 12: print("  id", k, "coord", k % nb, (k // nb) % nb, k // (nb * nb), "self", None if xa is None else (int(xa["n_verts"]), int(xa["n_idx"])), "classic", None if xc is None else (int(xc["n_verts"]), int(xc["n_idx"])))
            bad += 1
    print("  differing blocks:", bad)

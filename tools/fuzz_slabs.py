#!/usr/bin/env python3
"""Randomised check of the slab path on ONE GPU: random fields (tools/fuzz_parity.py kinds), sizes, level limits, world sizes
and slab axes; the ranks' slabs (own part + halo, attached device tensors) polygonized one after the other must merge
into the whole-grid result of the same library.  Usage (GPU box): python tools/fuzz_slabs.py [seconds] [first_seed]"""
import os
import sys
import time

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
sys.path.insert(0, os.path.join(ROOT, "tools"))
if __name__ == "__main__":
    torch.cuda.init()
import fields  # noqa: E402
import fuzz_parity as fz  # noqa: E402
import vxo  # noqa: E402
from voxels_amd import Polygonizer, synth  # noqa: E402
from voxels_amd.slab import SlabBuffers, merge_rank_levels  # noqa: E402


_CTX = {}


def contexts():
    if not _CTX:
        _CTX["whole"] = Polygonizer(device=0)
        _CTX["whole"].set_materials(vxo.default_lut())
        _CTX["part"] = Polygonizer(device=0)
        _CTX["part"].set_materials(vxo.default_lut())
    return _CTX["whole"], _CTX["part"]


def check_seed(seed):
    """One configuration derived from `seed`; returns its description, or None when the draw is not a valid slab layout
    (the caller moves on to the next seed); raises AssertionError on a mismatch."""
    dev = torch.device("cuda", 0)
    whole, part = contexts()
    rng = np.random.RandomState(seed)
    n, levels, world, axis = int(rng.choice([64, 128, 256])), int(rng.choice([1, 2, 3])), int(rng.choice([2, 4])), str(rng.choice(["z", "y"]))
    if n % ((16 << (levels - 1)) * world):
        return None
    d, m, b = fz.make_field(int(rng.randint(0, 5)), n, seed + 1)
    flags = synth.block_empty_flags(d)
    whole.upload(d, m, b, flags)
    whole.execute(levels)
    want, want_stats = whole.all_levels(), whole.stats()
    parts, stats = [], np.zeros(20, np.uint64)
    for r in range(world):
        slab = SlabBuffers(torch, n, r, world, dev, axis=axis)
        slab.fill_from_full(d, m, b, flags)
        torch.cuda.synchronize()
        slab.attach(part)
        part.execute(levels)
        parts.append(part.all_levels())
        stats += part.stats()
    ok, msg = fields.surface_equal(merge_rank_levels(parts), want, nrm_tol=0.0)
    desc = "seed %d n %d levels %d world %d axis %s" % (seed, n, levels, world, axis)
    assert ok and np.array_equal(stats.astype(np.uint32), want_stats), "MISMATCH %s: %s" % (desc, msg)
    return desc


def main():
    budget = float(sys.argv[1]) if len(sys.argv) > 1 else 60.0
    seed = int(sys.argv[2]) if len(sys.argv) > 2 else 300
    t0, runs = time.time(), 0
    while time.time() - t0 < budget:
        if check_seed(seed) is not None:
            runs += 1
        seed += 1
    print("slab fuzz ok: %d configurations" % runs)


if __name__ == "__main__":
    main()

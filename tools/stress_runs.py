#!/usr/bin/env python3
"""Many full runs in a row on resident grids: every run's counts (vertices, indices, active blocks, statistics) must equal
the first run's, and the mesh digest of the last run the first run's - a race in k_main's flag protocol, a set that was
not left clean by k_tail or a lost hand-over would show up as a difference or as a failed run.
Usage (GPU box): python tools/stress_runs.py [runs_small=3000] [runs_large=600]"""
import os
import sys
import time

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from voxels_amd import Polygonizer, digest, synth  # noqa: E402
from voxels_amd.slab import SlabBuffers  # noqa: E402


def signature(p, info):
    return (int(info.total_verts), int(info.total_indices), tuple(int(x) for x in info.active_blocks[:info.levels]), tuple(int(x) for x in p.stats()))


def stress(name, p, levels, runs):
    info = p.execute(levels)
    first = signature(p, info)
    d0 = digest.surface_digest(p.all_levels())
    t = time.perf_counter()
    for i in range(runs):
        info = p.execute(levels)
        s = signature(p, info)
        if s != first:
            print("%s: run %d differs: %s vs %s" % (name, i, s, first))
            return False
    dt = time.perf_counter() - t
    ok = digest.digests_equal(digest.surface_digest(p.all_levels()), d0)
    print("%s: %d runs, %.4f ms per run (counts and statistics read back every run), last digest %s the first" % (name, runs, dt / runs * 1e3, "equals" if ok else "DIFFERS FROM"))
    return ok


def main():
    small = int(sys.argv[1]) if len(sys.argv) > 1 else 3000
    large = int(sys.argv[2]) if len(sys.argv) > 2 else 600
    ok = True
    p = Polygonizer(device=0)
    p.set_materials(synth.default_lut())
    for n, runs in ((128, small), (256, small // 2), (1024, large)):
        p.create_terrain(n, 1337)
        ok = stress("%d^3 terrain" % n, p, 4, runs) and ok
    dev = torch.device("cuda", 0)
    slab = SlabBuffers(torch, 1024, 3, 8, dev, axis="y")
    slab.attach(p)
    p.fill_terrain(1337)
    ok = stress("slab 3/8 of 1024^3", p, 4, small) and ok
    print("stress", "ok" if ok else "FAILED")
    sys.exit(0 if ok else 1)


if __name__ == "__main__":
    main()

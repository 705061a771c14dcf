#!/usr/bin/env python3
"""Per-rank step time of the slab path, all "ranks" of a world size run one after the other on ONE GPU (no exchange):
shows how evenly an axis splits the bench terrain.  Usage (GPU box): python tools/slab_time.py [axis] [n]"""
import os
import sys
import time

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
torch.cuda.init()
import vxo  # noqa: E402
from voxels_amd import Polygonizer, synth  # noqa: E402
from voxels_amd.slab import SlabBuffers  # noqa: E402


def main():
    axis = sys.argv[1] if len(sys.argv) > 1 else "y"
    n = int(sys.argv[2]) if len(sys.argv) > 2 else 1024
    levels = 4
    dev = torch.device("cuda", 0)
    d, m, b = synth.terrain(n, 0, n, 1337)
    flags = synth.block_empty_flags(d)
    for world in (2, 4, 8):
        res = []
        for r in range(world):
            slab = SlabBuffers(torch, n, r, world, dev, axis=axis)
            slab.fill_from_full(d, m, b, flags)
            torch.cuda.synchronize()
            p = Polygonizer(device=0)
            p.set_materials(vxo.default_lut())
            slab.attach(p)
            for _ in range(5):
                p.execute(levels)
            dt = None
            for _ in range(3):  # best of three batches (a single slow call would otherwise decide the maximum over ranks)
                t = time.perf_counter()
                for _ in range(20):
                    info = p.execute(levels)
                batch = (time.perf_counter() - t) / 20
                dt = batch if dt is None else min(dt, batch)
            res.append((dt * 1e3, info.device_ms, int(info.active_blocks[0])))
            p.close()
            del slab
        print("axis %s world %d: max %.3f ms | " % (axis, world, max(a for a, _, _ in res)) + " ".join("r%d %.3f (%d blk)" % (i, a, c) for i, (a, _, c) in enumerate(res)))


if __name__ == "__main__":
    main()

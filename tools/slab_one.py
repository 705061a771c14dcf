#!/usr/bin/env python3
"""One rank's slab of the bench terrain polygonized a few times (for kernel traces of the multi-GPU per-rank step on one GPU).
Usage (GPU box): python tools/slab_one.py [world] [rank] [axis] [n] [reps]"""
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
torch.cuda.init()
from voxels_amd import Polygonizer, synth  # noqa: E402
from voxels_amd.slab import SlabBuffers  # noqa: E402


def main():
    world = int(sys.argv[1]) if len(sys.argv) > 1 else 8
    rank = int(sys.argv[2]) if len(sys.argv) > 2 else 3
    axis = sys.argv[3] if len(sys.argv) > 3 else "y"
    n = int(sys.argv[4]) if len(sys.argv) > 4 else 1024
    reps = int(sys.argv[5]) if len(sys.argv) > 5 else 6
    dev = torch.device("cuda", 0)
    slab = SlabBuffers(torch, n, rank, world, dev, axis=axis)
    p = Polygonizer(device=0)
    p.set_materials(synth.default_lut())
    slab.attach(p)
    p.fill_terrain(1337)
    for _ in range(reps):
        info = p.execute(4)
    print("world %d rank %d axis %s: device %.3f ms, %d level-0 surface blocks" % (world, rank, axis, info.device_ms, int(info.active_blocks[0])))


if __name__ == "__main__":
    main()

#!/usr/bin/env python3
"""Time of the mirrors' rebuild (k_rebrick over the whole resident grid) at 1024^3: the grid is declared changed, the next run
reports what bringing the mirrors up to date took (vx_exec_info.mirror_ms, HIP events around the launch).
Usage (GPU box): python tools/rebrick_time.py [n=1024] [reps=6]"""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch  # noqa: E402
torch.cuda.init()
from voxels_amd import Polygonizer, synth  # noqa: E402

n = int(sys.argv[1]) if len(sys.argv) > 1 else 1024
reps = int(sys.argv[2]) if len(sys.argv) > 2 else 6
p = Polygonizer(device=0)
p.set_materials(synth.default_lut())
p.create_terrain(n, 1337)
p.execute(4)
ms = []
for _ in range(reps):
    p.invalidate()
    ms.append(float(p.execute(4).mirror_ms))
vox = n ** 3
total = 6 * vox + sum(vox >> (3 * l) for l in range(1, 4))
best = min(ms)
print("%d^3 mirrors: best %.4f ms, median %.4f ms = %.0f GB/s (%.3f of 8 TB/s) %s" % (n, best, sorted(ms)[len(ms) // 2], total / best / 1e6, total / best / 1e6 / 8000, os.environ.get("VOXELS_HIP_LIBRARY", "")))

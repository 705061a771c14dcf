#!/usr/bin/env python3
"""A chain of sphere carves with incremental runs on a resident terrain (BASELINE config 5 shape), for kernel traces and
timings: python tools/edit_run.py [n=512] [levels=0] [edits=8] [frac=0.37].  Prints per-call wall and device time."""
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch  # noqa: E402,F401  (its HIP runtime first, as in bench.py)
torch.cuda.init()
from voxels_amd import Polygonizer, synth  # noqa: E402


def main():
    n = int(sys.argv[1]) if len(sys.argv) > 1 else 512
    levels = int(sys.argv[2]) if len(sys.argv) > 2 else 0
    edits = int(sys.argv[3]) if len(sys.argv) > 3 else 8
    frac = float(sys.argv[4]) if len(sys.argv) > 4 else 0.37  # ball centres off the lattice (0 = lattice points: exact zero samples on the ball)
    p = Polygonizer(device=0)
    p.set_materials(synth.default_lut())
    p.create_terrain(n, 1337)
    p.execute(levels)
    p.level(0, with_data=False)  # (the host copy of the block lists: fetched once after a full run)
    col = p.column(n, n // 2, n // 2)
    zs = float(np.argmax(col >= 0)) if (col >= 0).any() else n * 0.5  # a point on the surface
    calls, devs, blocks = [], [], []
    for k in range(edits):
        pos = (n / 2.0 + 23.0 * (k % 4) - 30.0 + frac, n / 2.0 + 19.0 * (k // 4) - 20.0 + frac * 1.65, zs + 2.0 * (k % 3) + frac * 0.62)
        mn, mx = p.inject_ball(pos, (44.0, 44.0, 44.0), 20.0, 2)
        t = time.perf_counter()
        got = p.execute_dirty(mn, mx)
        calls.append((time.perf_counter() - t) * 1e3); devs.append(p.info.device_ms); blocks.append(got.size)
    print("n %d levels %d: per call ms %s" % (n, p.info.levels, " ".join("%.3f" % c for c in calls)))
    print("device ms %s" % " ".join("%.3f" % c for c in devs))
    print("blocks %s; steady mean %.4f ms per call, device %.4f ms" % (blocks, float(np.mean(calls[2:])), float(np.mean(devs[2:]))))


if __name__ == "__main__":
    main()

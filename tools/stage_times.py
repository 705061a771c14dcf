#!/usr/bin/env python3
"""Serialised per-stage device times of one polygonization of the bench terrain (HIP events between the stages).
Usage (GPU box): python tools/stage_times.py [n] [levels] [reps]"""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from voxels_amd import Polygonizer, synth  # noqa: E402


def main():
    n = int(sys.argv[1]) if len(sys.argv) > 1 else 1024
    levels = int(sys.argv[2]) if len(sys.argv) > 2 else 4
    reps = int(sys.argv[3]) if len(sys.argv) > 3 else 10
    d, m, b = synth.terrain(n)
    from voxels_amd.binding import HipLibrary
    p = Polygonizer(library=HipLibrary(os.environ["VX_LIB"])) if os.environ.get("VX_LIB") else Polygonizer()
    p.upload(d, m, b, synth.block_empty_flags(d))
    p.set_stage_timing(True)
    for _ in range(3):
        p.execute(levels)
    acc = np.zeros(8)
    for _ in range(reps):
        p.execute(levels)
        acc += p.stage_times()
    acc /= reps
    print("n=%d levels=%d  reset %.4f classify %.4f hierarchy %.4f material %.4f regular0 %.4f regularN %.4f transition %.4f lists %.4f  sum %.4f ms" % ((n, levels) + tuple(acc) + (acc.sum(),)))


if __name__ == "__main__":
    main()

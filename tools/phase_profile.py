#!/usr/bin/env python3
"""Attribute k_regular time to its phases by running the pipeline with vx_debug_phase_limit = 1..7 (profiling aid).
Usage (on the GPU box): python tools/phase_profile.py [n] [levels]"""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
from voxels_amd import Polygonizer, synth  # noqa: E402


def main():
    n = int(sys.argv[1]) if len(sys.argv) > 1 else 1024
    levels = int(sys.argv[2]) if len(sys.argv) > 2 else 4
    d, m, b = synth.terrain(n)
    fl = synth.block_empty_flags(d)
    p = Polygonizer()
    p.upload(d, m, b, fl)
    p.set_stage_timing(True)
    names = ["1 load bits+samples", "2 prefix", "3 list+cells", "4 count", "5 scan+alloc v", "6 describe+emit verts", "7 keep masks", "8 scan+alloc i", "0 full"]
    prev = 0.0
    for lim in [int(x, 0) for x in os.environ.get('VX_LIMS', '1,2,3,4,5,6,7,8,0').split(',')]:
        p.debug_phase_limit(lim)
        acc = np.zeros(6)
        for _ in range(6):
            p.execute(levels)
            acc += p.stage_times()
        acc /= 6
        print("limit %-22s k_regular %.4f ms (+%.4f)   [classify %.4f material %.4f transition %.4f]" % (names[(lim & 0xFF) - 1 if (lim & 0xFF) else 8] + " flags %x" % (lim >> 8), acc[4], acc[4] - prev, acc[1], acc[3], acc[5]))
        prev = acc[4]


if __name__ == "__main__":
    main()

#!/usr/bin/env python3
"""The second ('caves') bench workload alone, for profilers: the 1024^3 caves field generated on the device, a few
polygonizations.  Usage (GPU box): python tools/caves_run.py [n=1024] [levels=4] [runs=3]"""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from voxels_amd import Polygonizer, synth  # noqa: E402


def main():
    n = int(sys.argv[1]) if len(sys.argv) > 1 else 1024
    levels = int(sys.argv[2]) if len(sys.argv) > 2 else 4
    runs = int(sys.argv[3]) if len(sys.argv) > 3 else 3
    p = Polygonizer(device=0)
    p.set_materials(synth.default_lut())
    p.create_terrain(n, 1337, 1)
    for _ in range(1 + runs):
        info = p.execute(levels)
    print("caves n=%d levels=%d device_ms %.4f active %s" % (n, levels, info.device_ms, [int(x) for x in info.active_blocks[:levels]]))


if __name__ == "__main__":
    main()

#!/usr/bin/env python3
"""One warm full run of the bench terrain under a profile build (VOXELS_HIP_LIBRARY=tools/ab/<x>prof.so): the build prints
its in-kernel phase profile to stderr after every run; the last run's is what counts.  Usage: python tools/prof_once.py [n=1024] [levels=4]"""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch  # noqa: E402
torch.cuda.init()
from voxels_amd import Polygonizer, synth  # noqa: E402

n = int(sys.argv[1]) if len(sys.argv) > 1 else 1024
levels = int(sys.argv[2]) if len(sys.argv) > 2 else 4
p = Polygonizer(device=0)
p.set_materials(synth.default_lut())
p.create_terrain(n, 1337)
for _ in range(3):
    p.execute(levels)
sys.stderr.write("==== last run ====\n")
info = p.execute(levels)
sys.stderr.write("device ms %.4f\n" % info.device_ms)

#!/usr/bin/env python3
"""k_decode_grid alone, for profilers: the bench terrain packed on the device, uploaded as a packed file a few times.
Usage (GPU box): python tools/decode_only.py [n=1024] [runs=3]"""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from voxels_amd import Polygonizer, synth  # noqa: E402

n = int(sys.argv[1]) if len(sys.argv) > 1 else 1024
runs = int(sys.argv[2]) if len(sys.argv) > 2 else 3
p = Polygonizer(device=0)
p.set_materials(synth.default_lut())
p.create_terrain(n, 1337)
blob = p.pack()
for _ in range(runs):
    p.upload_packed(blob)
print("file bytes", blob.size)

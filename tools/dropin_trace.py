#!/usr/bin/env python3
"""Where a cold Polygonizer::Execute through libVoxels.so spends its time: the bench terrain written as a grid file,
tools/dropin_bench run on it with VOXELS_TRACE and VX_HOST_TIMING.  Usage (GPU box): python tools/dropin_trace.py [n] [runs] [ENV=VALUE ...]"""
import os
import subprocess
import sys
import tempfile

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from voxels_amd import Polygonizer, synth  # noqa: E402

args = [a for a in sys.argv[1:] if "=" not in a]
envs = dict(a.split("=", 1) for a in sys.argv[1:] if "=" in a)
n = int(args[0]) if args else 1024
runs = args[1] if len(args) > 1 else "3"
p = Polygonizer(device=0)
p.set_materials(synth.default_lut())
p.create_terrain(n, 1337)
blob = p.pack()
p.close()
with tempfile.NamedTemporaryFile(suffix=".vxgrid", delete=False) as f:
    f.write(blob.tobytes())
    path = f.name
r = subprocess.run([os.path.join(ROOT, "tools", "dropin_bench"), path, runs], capture_output=True, text=True, env=dict(os.environ, VOXELS_TRACE="1", VX_HOST_TIMING="1", **envs))
os.unlink(path)
print("\n".join(r.stderr.splitlines()[-24:]))
print(r.stdout.strip().splitlines()[-1] if r.stdout.strip() else "no output")

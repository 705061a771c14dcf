"""Time of the one-step mesh download (vx_host_meshes_acquire: one copy stream, 32 MB pieces - the cut is fixed since round 6,
profiles/r03_d2h_time.txt has the sweep that settled it).
usage: python tools/d2h_time.py [n=1024] [levels=4]"""
import os, sys, time
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import numpy as np
from voxels_amd.binding import Polygonizer

n = int(sys.argv[1]) if len(sys.argv) > 1 else 1024
levels = int(sys.argv[2]) if len(sys.argv) > 2 else 4
p = Polygonizer()
p.create_terrain(n, 1337)
lut = np.zeros((256, 6), np.uint8)
for i in range(256):
    lut[i] = [(i * 6 + k) % 251 for k in range(6)]
p.set_materials(lut)
p.execute(levels)
mb = (p.info.total_verts * 48 + p.info.total_indices * 4) / 1e6
print("pools: %.1f MB" % mb)
for lanes, piece in ((1, 32),):
    best = 1e9
    for r in range(4):
        p.execute(levels)
        t = time.perf_counter()
        hm = p.host_meshes()
        dt = (time.perf_counter() - t) * 1e3
        hm.release()
        best = min(best, dt) if r else best  # the first call may have to page-lock a new arena
        if r == 0:
            first = dt
    print("streams %d piece %3d MB: first %.2f ms, best %.2f ms = %.1f GB/s" % (lanes, piece, first, best, mb / best))

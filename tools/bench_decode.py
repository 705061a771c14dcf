#!/usr/bin/env python3
"""§8(f) row 1 measurement: Grid file format v1 -> dense fields resident in HBM.
  vx_grid_upload_packed (H2D of the packed file + k_decode_grid)   vs   vx_grid_upload of the dense arrays (3 B/voxel)
  vs the reference's own Grid::Load (oracle/_ref, host cores).     Usage (GPU box): python tools/bench_decode.py [n]"""
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import vxo  # noqa: E402
from voxels_amd import Polygonizer, synth  # noqa: E402


def main():
    n = int(sys.argv[1]) if len(sys.argv) > 1 else 512
    d, m, b = synth.terrain(n, 0, n, 1337)
    oracle = vxo.load_ref() or vxo.load_port()
    g = oracle.grid_from_dense(d, m, b)
    blob = g.pack()
    p = Polygonizer()
    p.set_materials(vxo.default_lut())
    best_p = best_d = best_c = 1e9
    for _ in range(3):
        t = time.perf_counter(); p.upload_packed(blob); best_p = min(best_p, time.perf_counter() - t)
    info = p.execute(0)
    packed_totals = [p.level(l, with_data=False).infos["n_idx"].sum() for l in range(info.levels)]
    fl = g.block_flags()
    for _ in range(3):
        t = time.perf_counter(); p.upload(d, m, b, fl); best_d = min(best_d, time.perf_counter() - t)
    info = p.execute(0)
    dense_totals = [p.level(l, with_data=False).infos["n_idx"].sum() for l in range(info.levels)]
    assert packed_totals == dense_totals
    for _ in range(2):
        t = time.perf_counter(); g2 = oracle.grid_load(blob); best_c = min(best_c, time.perf_counter() - t)
    vox = float(n) ** 3
    print("grid %d^3: file %.1f MB (%.1fx smaller than 3 B/voxel)" % (n, blob.size / 1e6, 3 * vox / blob.size))
    print("  vx_grid_upload_packed : %8.2f ms  -> %7.1f GB/s of dense field (%.0f Mvoxel/s)" % (best_p * 1e3, 3 * vox / best_p / 1e9, vox / best_p / 1e6))
    print("  vx_grid_upload (dense): %8.2f ms  -> %7.1f GB/s" % (best_d * 1e3, 3 * vox / best_d / 1e9))
    print("  reference Grid::Load (%s, host): %8.2f ms -> %7.1f GB/s" % (oracle.kind, best_c * 1e3, 3 * vox / best_c / 1e9))


if __name__ == "__main__":
    main()

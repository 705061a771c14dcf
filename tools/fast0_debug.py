#!/usr/bin/env python3
"""A/B of the level-0 regular pass on the GPU: the table-driven pass for zero-free blocks (VX_FAST=3, default) against
the general pass (VX_FAST=0) on the same grids, block by block and field by field — where do they differ?
Usage: python tools/fast0_debug.py [n ...]"""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import vxo  # noqa: E402
from voxels_amd import Polygonizer, synth  # noqa: E402


def run(fast, d, m, b, flags, levels):
    os.environ["VX_FAST"] = "3" if fast else "0"
    p = Polygonizer()
    p.set_materials(vxo.default_lut())
    p.upload(d, m, b, flags)
    info = p.execute(levels)
    lv = p.all_levels()
    st = p.stats()
    p.close()
    return lv, st, info


def main():
    sizes = [int(a) for a in sys.argv[1:]] or [64, 256]
    bad = 0
    for n in sizes:
        d, m, b = synth.terrain(n, 0, n, 7)
        flags = synth.block_empty_flags(d)
        a, sa, ia = run(True, d, m, b, flags, 0)
        g, sg, ig = run(False, d, m, b, flags, 0)
        print("n=%d: device ms fast %.3f general %.3f" % (n, ia.device_ms, ig.device_ms))
        if not np.array_equal(sa, sg):
            print("  stats differ:", sa.tolist(), sg.tolist()); bad += 1
        for li, (A, G) in enumerate(zip(a, g)):
            print("  L%d: fast %s general %s" % (li, A.totals(), G.totals()))
            if A.infos.size != G.infos.size:
                print("  L%d block counts differ" % li); bad += 1; continue
            for name in A.infos.dtype.names:
                if name in ("v_off", "i_off", "tv_off", "ti_off"):
                    continue
                if not np.array_equal(A.infos[name], G.infos[name]):
                    w = np.flatnonzero((A.infos[name] != G.infos[name]).reshape(A.infos.size, -1).any(axis=1))
                    print("  L%d info.%s differs in %d blocks, first %d: %s vs %s" % (li, name, w.size, w[0], A.infos[name][w[0]], G.infos[name][w[0]])); bad += 1
            if A.totals() != G.totals():
                continue
            if not np.array_equal(A.idx, G.idx):
                w = np.flatnonzero(A.idx != G.idx)
                print("  L%d %d of %d indices differ, first at %d: %d vs %d" % (li, w.size, A.idx.size, w[0], A.idx[w[0]], G.idx[w[0]])); bad += 1
            for fld in ("pos", "sec", "nrm", "tex"):
                xa, xb = A.verts[fld], G.verts[fld]
                ne = (xa.view(np.uint32 if fld != "tex" else np.uint8) != xb.view(np.uint32 if fld != "tex" else np.uint8))
                if ne.any():
                    w = np.flatnonzero(ne.reshape(len(xa), -1).any(axis=1))
                    print("  L%d verts.%s differs in %d of %d vertices, first %d: %s vs %s" % (li, fld, w.size, len(xa), w[0], xa[w[0]], xb[w[0]])); bad += 1
    print("fast0 A/B:", "IDENTICAL" if not bad else "%d differences" % bad)
    return 1 if bad else 0


if __name__ == "__main__":
    sys.exit(main())

#!/usr/bin/env python3
"""HIP vs oracle port on the big synthetic terrains (512^3 / 1024^3, 4 LOD levels) + run-to-run determinism.
Usage (GPU box): python tools/big_parity.py 512 1024"""
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import fields  # noqa: E402
import vxo  # noqa: E402
from voxels_amd import Polygonizer, synth  # noqa: E402

for n in [int(a) for a in sys.argv[1:]] or [512]:
    levels = 4
    d, m, b = synth.terrain(n)
    port = vxo.load_port()
    t = time.time()
    g = port.grid_from_dense(d, m, b)
    s = port.execute(g)
    ref = s.all_levels()[:levels]
    print("n=%d port %.1fs" % (n, time.time() - t), [l.totals() for l in ref], flush=True)
    p = Polygonizer()
    p.set_materials(vxo.default_lut())
    p.upload(d, m, b, g.block_flags())
    prev = None
    for rep in range(3):
        p.execute(levels)
        lv = p.all_levels()
        tot = [l.totals() for l in lv]
        ok, msg = fields.surface_equal(lv, ref, nrm_tol=1e-5)
        print(" run %d totals %s parity %s %s" % (rep, "same" if tot == [l.totals() for l in ref] else tot, ok, msg), flush=True)
    p.close()

#!/usr/bin/env python3
"""§8(f) row 2 measurement (BASELINE config 5 shape): 512^3 terrain, sphere carve of radius 20 at the surface, then the
incremental re-polygonization.  Compares the edit done on the device (vx_grid_inject_ball) with the host path
(reference Grid::InjectSurface on the host + vx_grid_update_blocks).  Usage (GPU box): python tools/bench_edit.py [n]"""
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


def main():
    n = int(sys.argv[1]) if len(sys.argv) > 1 else 512
    d, m, b = synth.terrain(n, 0, n, 1337)
    oracle = vxo.load_ref() or vxo.load_port()
    g = oracle.grid_from_dense(d, m, b)
    p = Polygonizer()
    p.set_materials(vxo.default_lut())
    p.upload_packed(g.pack())
    p.execute(0)
    col = d[:, n // 2, n // 2]
    zs = int(np.argmax(col >= 0)) if (col >= 0).any() else n // 2  # a point on the surface
    pos, ext, r = (n / 2.0, n / 2.0, float(zs)), (44.0, 44.0, 44.0), 20.0
    t = time.perf_counter(); mn, mx = p.inject_ball(pos, ext, r, 2); t_dev = time.perf_counter() - t
    t = time.perf_counter(); ids = p.execute_dirty(mn, mx); t_poly = time.perf_counter() - t
    dev1 = p.info.device_ms
    # steady state: a second, smaller edit next to the first (block lists, scratch buffers and pools are in place now)
    pos2 = (pos[0] + 9.0, pos[1] - 7.0, pos[2] + 3.0)
    t = time.perf_counter(); mnb, mxb = p.inject_ball(pos2, (30.0, 30.0, 30.0), 12.0, 2); t_dev2 = time.perf_counter() - t
    t = time.perf_counter(); idsb = p.execute_dirty(mnb, mxb); t_poly2 = time.perf_counter() - t
    dev2 = p.info.device_ms
    # host path for the same edit on a second context
    q = Polygonizer()
    q.set_materials(vxo.default_lut())
    pre = g.read_dense()
    q.upload(*pre, g.block_flags())
    q.execute(0)
    t = time.perf_counter()
    mn2, mx2 = g.inject_ball(pos, ext, r, 2)
    t_host_edit = time.perf_counter() - t
    post = g.read_dense()
    bids, (dd, mm, bb) = fields.edited_blocks(pre, post)
    t = time.perf_counter(); q.update_blocks(bids, dd.view(np.int8), mm, bb, g.block_flags()); t_up = time.perf_counter() - t
    ids2 = q.execute_dirty(mn2, mx2)
    assert np.array_equal(ids, ids2) and np.array_equal(mn, mn2)
    g.inject_ball(pos2, (30.0, 30.0, 30.0), 12.0, 2)
    # steady state (BASELINE config 5): a chain of r = 20 carves along the surface, per call wall time and device time - on the
    # three-launch path (k_dirty_head | k_main<true> | k_dirty_tail) and, in a second context, on the chain of launches
    steady = {}
    # (ball centres: lattice points put exact zero samples wherever x^2 + y^2 + z^2 = r^2 has integer solutions - blocks with a zero
    # sample take the general pass, 40-60 us each, in the run's last kernel; an application's brush positions are arbitrary
    # floats, so both kinds are timed: "frac" = centres off the lattice, "int" = the lattice-point worst case)
    for label, env, frac in (("fused", None, (0.37, 0.61, 0.23)), ("fused, lattice-point centres", None, (0.0, 0.0, 0.0)), ("chain", "0", (0.37, 0.61, 0.23))):
        if env is not None:
            os.environ["VX_DIRTY_FUSED"] = env
        w = Polygonizer()
        os.environ.pop("VX_DIRTY_FUSED", None)
        w.set_materials(vxo.default_lut())
        w.upload(*pre, oracle.grid_from_dense(*pre).block_flags())
        w.execute(0)
        calls, devs, blocks = [], [], 0
        for k in range(16):
            pk = (pos[0] + 23.0 * (k % 4) - 30.0 + frac[0], pos[1] + 19.0 * (k // 4) - 20.0 + frac[1], pos[2] + 2.0 * (k % 3) + frac[2])
            a, bq = w.inject_ball(pk, ext, r, 2)
            t = time.perf_counter(); got = w.execute_dirty(a, bq); dt = time.perf_counter() - t
            if k >= 2:
                calls.append(dt * 1e3); devs.append(w.info.device_ms); blocks += got.size
        steady[label] = (float(np.mean(calls)), float(np.median(calls)), float(np.min(calls)), float(np.mean(devs)), blocks / len(calls), w, float(np.max(calls)))
    from voxels_amd import digest
    da, db = digest.surface_digest(steady["fused"][5].all_levels()), digest.surface_digest(steady["chain"][5].all_levels())
    assert digest.digests_equal(da, db), "the two incremental paths disagree"
    print("grid %d^3, IT_Subtract ball r=%g, extents %s: %d blocks changed, %d blocks rebuilt" % (n, r, ext, bids.size, ids.size))
    for label in ("fused", "fused, lattice-point centres", "chain"):
        m, med, lo, dv, nb, _, hi = steady[label]
        print("  steady state, %s: vx_polygonize_dirty %.4f ms per call (median %.4f, best %.4f, worst %.4f), device %.4f ms, %.0f blocks rebuilt per call" % (label, m, med, lo, hi, dv, nb))
    print("  surfaces after the 16 edits: equal on both paths (digest %016x)" % int(da[1]))
    print("  device edit (vx_grid_inject_ball, incl. BF_Empty refresh): %7.3f ms" % (t_dev * 1e3))
    print("  incremental polygonization (vx_polygonize_dirty)          : %7.3f ms (device %.3f ms)" % (t_poly * 1e3, dev1))
    print("  second edit (r=12): device edit %.3f ms, vx_polygonize_dirty %.3f ms (device %.3f ms), %d blocks rebuilt" % (t_dev2 * 1e3, t_poly2 * 1e3, dev2, idsb.size))
    print("  host path: reference Grid::InjectSurface %7.3f ms + vx_grid_update_blocks %7.3f ms" % (t_host_edit * 1e3, t_up * 1e3))


if __name__ == "__main__":
    main()

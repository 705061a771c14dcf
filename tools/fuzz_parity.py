#!/usr/bin/env python3
"""Randomised parity sweep of the HIP path against the oracle (not part of the test suite; run on the GPU box):
terrains, full-range noise, zero-heavy fields, sparse bubbles and flat slabs with random materials, sizes 32..128,
every level; with a third argument `edits`: random chains of device edits + incremental runs (VX_FUZZ_CHAIN = edits per
chain, default 5; VX_FUZZ_N = grid size).
Usage: python tools/fuzz_parity.py [seconds] [first_seed] [edits]"""
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import fields  # noqa: E402
import vxo  # noqa: E402
from voxels_amd import Polygonizer  # noqa: E402


def make_field(kind, n, seed):
    rng = np.random.RandomState(seed)
    if kind == 0:
        f = fields.terrain_field(n, seed)
        d = fields.quantize_full_range(f, scale=rng.choice([1.0, 3.0, 12.0]))
    elif kind == 1:
        d = fields.quantize_full_range(fields.smooth_noise(n, seed, scale=rng.choice([4, 8, 16]), amp=1.0), scale=rng.choice([20.0, 60.0, 120.0]))
    elif kind == 2:  # zero-heavy
        d = np.clip(np.round(fields.smooth_noise(n, seed, scale=8, amp=2.0) * 1.5), -4, 4).astype(np.int8)
    elif kind == 3:  # sparse bubbles in a saturated field: many quiet / skipped blocks
        d = np.full((n, n, n), 127 if seed & 1 else -127, np.int8)
        for _ in range(rng.randint(1, 6)):
            c = rng.randint(2, n - 2, 3)
            r = rng.randint(1, max(2, n // 6))
            z, y, x = np.ogrid[:n, :n, :n]
            dist = np.sqrt((z - c[0]) ** 2 + (y - c[1]) ** 2 + (x - c[2]) ** 2) - r
            d = np.where(dist < 3, np.clip(np.round(dist * (1 if seed & 1 else -1) * 20), -127, 127), d).astype(np.int8)
    else:  # flat slabs on / off block boundaries
        h = rng.choice([15.5, 16.0, 16.5, 31.0, 32.0, n / 2.0 + 0.25])
        axis = rng.randint(0, 3)
        coord = np.arange(n).reshape([n if a == axis else 1 for a in range(3)]) * np.ones((n, n, n))
        d = np.clip(np.sign(coord - h) * np.ceil(np.abs(coord - h)), -rng.choice([4, 60, 127]), rng.choice([4, 60, 127])).astype(np.int8)
    m = rng.randint(0, rng.choice([1, 3, 200]) + 1, (n, n, n)).astype(np.uint8)
    b = rng.randint(0, 256, (n, n, n)).astype(np.uint8)
    return np.ascontiguousarray(d), m, b


def fuzz_edits(oracle, p, budget, seed):
    """random chains of device edits + incremental runs: grid file bytes (voxels + codec state), rebuilt ids, surface"""
    t0, chains, edits = time.time(), 0, 0
    while time.time() - t0 < budget:
        rng = np.random.RandomState(seed)
        n = int(os.environ["VX_FUZZ_N"]) if os.environ.get("VX_FUZZ_N") else int(rng.choice([32, 64, 64]))
        d, m, b = make_field(int(rng.choice([0, 1, 2])), n, seed)
        g = oracle.grid_from_dense(d, m, b)
        s = oracle.execute(g)
        p.upload_packed(g.pack())
        p.execute()
        for _ in range(int(os.environ.get("VX_FUZZ_CHAIN", "5"))):  # (long chains: pool growth, compaction, the capacity classes coming and going)
            pos = tuple(float(x) for x in rng.uniform(-4, n + 4, 3).round(rng.choice([0, 1, 2])))
            ext = tuple(float(x) for x in rng.uniform(3, 26, 3).round(rng.choice([0, 1])))
            if rng.rand() < 0.7:
                args = (pos, ext, float(rng.uniform(1.5, 11)), int(rng.randint(0, 3)))
                mn, mx = g.inject_ball(*args)
                mn2, mx2 = p.inject_ball(*args)
            else:
                args = (pos, ext, int(rng.randint(0, 5)), bool(rng.randint(0, 2)))
                mn, mx = g.inject_material(*args)
                mn2, mx2 = p.inject_material(*args)
            edits += 1
            if not (np.array_equal(mn, mn2) and np.array_equal(mx, mx2) and np.array_equal(p.pack(), g.pack())):
                print("EDIT MISMATCH seed %d args %s" % (seed, args))
                sys.exit(1)
            ref_ids = oracle.execute_modify(g, s, mn, mx)
            got = p.execute_dirty(mn2, mx2)
            ok, msg = fields.surface_equal(p.all_levels(), s.all_levels(), nrm_tol=0.0)
            if not (np.array_equal(got, ref_ids) and ok and np.array_equal(p.stats(), s.stats())):
                print("INCREMENTAL MISMATCH seed %d args %s: %s" % (seed, args, msg))
                sys.exit(1)
        # a FULL run over the edited grid: the library's mirrors (bricks, lattice copies, sign summaries) followed the edits
        full = oracle.execute(g)
        p.execute()
        ok, msg = fields.surface_equal(p.all_levels(), full.all_levels(), nrm_tol=0.0)
        if not ok or not np.array_equal(p.stats(), full.stats()):
            print("FULL RUN AFTER EDITS MISMATCH seed %d: %s" % (seed, msg))
            sys.exit(1)
        chains += 1
        seed += 1
    print("edit fuzz ok: %d chains, %d edits" % (chains, edits))


def main():
    budget = float(sys.argv[1]) if len(sys.argv) > 1 else 60.0
    seed = int(sys.argv[2]) if len(sys.argv) > 2 else 1000
    oracle = vxo.load_port()
    if os.environ.get("VX_FUZZ_EMU"):  # CPU emulation of the device phases (tests/emu) instead of the GPU
        from emu_lib import emu_library
        from voxels_amd.binding import Polygonizer as EmuPolygonizer
        p = EmuPolygonizer(library=emu_library())
    else:
        p = Polygonizer()
    p.set_materials(vxo.default_lut())
    if len(sys.argv) > 3 and sys.argv[3] == "edits":
        return fuzz_edits(oracle, p, budget, seed)
    t0, runs, partial = time.time(), 0, 0
    while time.time() - t0 < budget:
        kind, n = seed % 5, [32, 64, 64, 128][(seed // 5) % 4]
        d, m, b = make_field(kind, n, seed)
        g = oracle.grid_from_dense(d, m, b)
        s = oracle.execute(g)
        p.upload(d, m, b, g.block_flags())
        p.execute()
        ok, msg = fields.surface_equal(p.all_levels(), s.all_levels(), nrm_tol=0.0)
        if not ok or not np.array_equal(p.stats(), s.stats()):
            print("MISMATCH seed %d kind %d n %d: %s" % (seed, kind, n, msg))
            sys.exit(1)
        p.upload_packed(g.pack())
        if not np.array_equal(p.pack(), g.pack()):
            print("PACK MISMATCH seed %d" % seed)
            sys.exit(1)
        if seed % 3 == 0 and len(s.all_levels()) > 1 and not os.environ.get("VX_FUZZ_EMU"):
            # a partial run (vx_polygonize_from): the levels from `first` up as in the full run, nothing below - and, behind an
            # edit, exactly the rebuilt blocks below, with the oracle's bytes (the caches the partial run left are complete)
            rng = np.random.RandomState(seed)
            ref = s.all_levels()
            first = int(rng.randint(1, len(ref) + 1))
            info = p.execute_from(0, first)
            if info.first_meshed_level:
                got = p.all_levels()
                ok, msg = fields.surface_equal(got[first:], ref[first:], nrm_tol=0.0)
                if not ok or any(len(got[l].infos) for l in range(first)):
                    print("PARTIAL RUN MISMATCH seed %d first %d: %s" % (seed, first, msg))
                    sys.exit(1)
                pos = tuple(float(x) for x in rng.uniform(4, n - 4, 3).round(1))
                args = (pos, (18.0, 18.0, 18.0), float(rng.uniform(3, 8)), int(rng.randint(0, 3)))
                mn, mx = g.inject_ball(*args)
                p.inject_ball(*args)
                ref_ids = oracle.execute_modify(g, s, mn, mx)
                ids = p.execute_dirty(mn, mx)
                got, ref = p.all_levels(), s.all_levels()
                ok, msg = fields.surface_equal(got[first:], ref[first:], nrm_tol=0.0)
                for l in range(first):
                    if ok:
                        ok, msg = fields.listed_blocks_equal_by_id(got[l], ref[l])
                if not (ok and np.array_equal(ids, ref_ids) and np.array_equal(p.stats(), s.stats())):
                    print("PARTIAL RUN + EDIT MISMATCH seed %d first %d: %s" % (seed, first, msg))
                    sys.exit(1)
                partial += 1
        runs += 1
        seed += 1
    print("fuzz ok: %d grids (%d of them also as a partial run + an edit), seeds up to %d, normals compared bitwise" % (runs, partial, seed - 1))


if __name__ == "__main__":
    main()

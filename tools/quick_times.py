#!/usr/bin/env python3
"""Step times of the fixed workloads a round is judged on, in one process: the 1024^3 bench terrain, a 128^3 grid (the
fixed cost of a run) and one rank's y-slab of an 8-rank job (1024^3 / 8), each with the environment variants given on
the command line (tuning knobs are read when a context is created).
Usage (GPU box): python tools/quick_times.py [VAR=VALUE[,VAR=VALUE..]] ...     ("-" = the defaults)"""
import os
import sys
import time

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
torch.cuda.init()
from voxels_amd import Polygonizer, synth  # noqa: E402
from voxels_amd.slab import SlabBuffers  # noqa: E402


def timed(p, levels, reps=20, batches=3):
    for _ in range(5):
        p.execute(levels)
    best = None
    for _ in range(batches):
        t = time.perf_counter()
        for _ in range(reps):
            info = p.execute(levels)
        dt = (time.perf_counter() - t) / reps * 1e3
        best = dt if best is None else min(best, dt)
    return best, info


def main():
    variants = sys.argv[1:] or ["-"]
    dev = torch.device("cuda", 0)
    which = os.environ.get("QT_WORKLOADS", "1024,128,slab").split(",")
    for spec in variants:
        saved = {}
        if spec != "-":
            for kv in spec.split(","):
                k, v = kv.split("=")
                saved[k] = os.environ.get(k)
                os.environ[k] = v
        out = []
        if "1024" in which:
            p = Polygonizer(device=0); p.set_materials(synth.default_lut()); p.create_terrain(1024, 1337)
            ms, info = timed(p, 4)
            out.append("1024^3 %.4f ms (dev %.4f, %d verts)" % (ms, info.device_ms, info.total_verts))
            p.close()
        if "512" in which:
            p = Polygonizer(device=0); p.set_materials(synth.default_lut()); p.create_terrain(512, 1337)
            ms, info = timed(p, 4)
            out.append("512^3 %.4f ms (dev %.4f)" % (ms, info.device_ms))
            p.close()
        if "128" in which:
            p = Polygonizer(device=0); p.set_materials(synth.default_lut()); p.create_terrain(128, 1337)
            ms, info = timed(p, 4, reps=50)
            out.append("128^3 %.4f ms (dev %.4f, %d verts)" % (ms, info.device_ms, info.total_verts))
            p.close()
        if "slab" in which:
            slab = SlabBuffers(torch, 1024, 3, 8, dev, axis="y")
            p = Polygonizer(device=0); p.set_materials(synth.default_lut()); slab.attach(p); p.fill_terrain(1337)
            ms, info = timed(p, 4, reps=50)
            out.append("slab 3/8 %.4f ms (dev %.4f, %d blk)" % (ms, info.device_ms, int(info.active_blocks[0])))
            p.close(); del slab
        if "caves" in which:
            p = Polygonizer(device=0); p.set_materials(synth.default_lut()); p.create_terrain(1024, 1337, 1)
            ms, info = timed(p, 4, reps=5, batches=2)
            out.append("caves %.4f ms (dev %.4f)" % (ms, info.device_ms))
            p.close()
        print("%-40s %s" % (spec, " | ".join(out)), flush=True)
        for k, v in saved.items():
            if v is None:
                del os.environ[k]
            else:
                os.environ[k] = v


if __name__ == "__main__":
    main()

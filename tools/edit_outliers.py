#!/usr/bin/env python3
"""Where the slow calls of an edit sequence come from: bench.py's config.edit chain (512^3, carves of r = 20) with
VX_HOST_TIMING=1 - per call its wall time, and on stderr the library's own split (enqueue / wait / kernels / pools packed or
grown) and the container's CPU throttling counters.  Usage (GPU box): VX_HOST_TIMING=1 [EDIT_HOST_GENERATOR=1]
[EDIT_SETTLE=seconds] python tools/edit_outliers.py [carves]"""
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from voxels_amd import Polygonizer, synth  # noqa: E402


def main():
    carves = int(sys.argv[1]) if len(sys.argv) > 1 else 14
    en, seed = 512, 1337
    ep = Polygonizer()
    ep.set_materials(synth.default_lut())
    ep.create_terrain(en, seed)
    ep.execute(0)
    ep.level(0, with_data=False)
    # EDIT_HOST_GENERATOR=1: find the surface by generating the grid on the host as well - a burst of one thread per CPU right in
    # front of the edits, which is what makes the container's CPU quota stall one of them (profiles/r05_edit_stalls.txt)
    if os.environ.get("EDIT_HOST_GENERATOR"):
        col = synth.terrain(en, 0, en, seed, materials=False)[0][:, en // 2, en // 2]
    else:
        col = ep.column(en, en // 2, en // 2)
    zs = float(np.argmax(col >= 0)) if (col >= 0).any() else en * 0.5
    def throttled():
        # (cgroup CPU bandwidth control: a burst of threads - the host generator above - can exhaust the container's quota and
        # put every thread to sleep until the 100 ms period ends, whatever it was doing)
        for path in ("/sys/fs/cgroup/cpu.stat", "/sys/fs/cgroup/cpu/cpu.stat"):
            try:
                return " ".join(l.strip() for l in open(path) if "throttled" in l)
            except OSError:
                pass
        return "cpu.stat not readable"
    settle = float(os.environ.get("EDIT_SETTLE", "0"))
    sys.stderr.write("before the edits: %s\n" % throttled())
    if settle:
        time.sleep(settle)
        sys.stderr.write("after %.1f s of sleep: %s\n" % (settle, throttled()))
    for k in range(carves):
        pos = (en / 2.0 + 23.0 * (k % 4) - 30.0 + 0.37, en / 2.0 + 19.0 * ((k // 4) % 4) - 20.0 + 0.61, zs + 2.0 * (k % 3) + 0.23)
        mn, mx = ep.inject_ball(pos, (44.0, 44.0, 44.0), 20.0, 2)
        sys.stderr.flush()
        t = time.perf_counter()
        ep.execute_dirty(mn, mx)
        dt = (time.perf_counter() - t) * 1e3
        sys.stderr.write("call %2d: %8.3f ms  (pool: %d verts, %d indices)\n" % (k, dt, ep.info.total_verts, ep.info.total_indices))
    sys.stderr.write("after the edits: %s\n" % throttled())
    ep.close()


if __name__ == "__main__":
    main()

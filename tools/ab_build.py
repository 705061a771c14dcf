#!/usr/bin/env python3
"""Variants of libvoxels_hip.so for A/B measurements on the GPU box: the same sources with extra -D switches, written to
tools/ab/<name>.so (git-ignored; they travel with the snapshot).  Usage: python tools/ab_build.py name=-DFLAG[,-DFLAG2] ..."""
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from voxels_amd import build as b  # noqa: E402


def main():
    out_dir = os.path.join(ROOT, "tools", "ab")
    os.makedirs(out_dir, exist_ok=True)
    procs = []
    for spec in sys.argv[1:]:
        name, _, flags = spec.partition("=")
        out = os.path.join(out_dir, name + ".so")
        cmd = ["/opt/rocm/bin/hipcc"] + b.HIP_FLAGS + [f for f in flags.split(",") if f] + ["-o", out, os.path.join(b.CSRC, "vx_hip.hip")]
        procs.append((name, subprocess.Popen(cmd, cwd=b.CSRC, stdout=subprocess.DEVNULL, stderr=subprocess.PIPE, text=True)))
    for name, p in procs:
        _, err = p.communicate()
        print(name, "ok" if p.returncode == 0 else "FAILED\n" + err[-2000:])


if __name__ == "__main__":
    main()

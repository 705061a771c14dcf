#!/usr/bin/env python3
"""Static instruction mix of one gfx950 kernel per source line (no GPU needed): compiles vx_hip.hip with line tables,
walks the kernel's assembly and attributes every instruction to the .loc in force.
Usage: python tools/asm_lines.py <kernel-name-substring> [min_count] [extra hipcc flags...]"""
import collections
import os
import re
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
SRC = os.path.join(ROOT, "voxels_amd", "csrc", "vx_hip.hip")


def main():
    want = sys.argv[1]
    min_count = int(sys.argv[2]) if len(sys.argv) > 2 else 8
    extra = sys.argv[3:]
    out = "/tmp/asm_lines.s"
    subprocess.check_call(["/opt/rocm/bin/hipcc", "--offload-arch=gfx950", "-O3", "-std=c++17", "-ffp-contract=off", "-fPIC", "-S", "--cuda-device-only",
                           "-gline-tables-only", "-o", out, SRC] + extra, stderr=subprocess.DEVNULL)
    files = {}
    per = collections.defaultdict(lambda: collections.Counter())
    cur = None
    on = False
    total = collections.Counter()
    for line in open(out):
        m = re.match(r"\s*\.file\s+(\d+)\s+\"([^\"]*)\"(?:\s+\"([^\"]*)\")?", line)
        if m:
            files[int(m.group(1))] = os.path.basename(m.group(3) or m.group(2))
            continue
        if re.match(r"^_Z\w*%s\w*:" % re.escape(want), line):
            on = True
            continue
        if on and line.startswith(".Lfunc_end"):
            break
        if not on:
            continue
        m = re.match(r"\s*\.loc\s+(\d+)\s+(\d+)", line)
        if m:
            cur = (files.get(int(m.group(1)), m.group(1)), int(m.group(2)))
            continue
        m = re.match(r"\s+(v_|s_|ds_|global_|buffer_|flat_|scratch_)(\w+)", line)
        if not m:
            continue
        op = m.group(1) + m.group(2)
        kind = {"v_": "valu", "s_": "salu", "ds_": "lds"}.get(m.group(1), "vmem")
        if op.startswith(("s_waitcnt", "s_nop", "s_barrier")):
            kind = "wait"
        if op.startswith(("v_readlane", "v_writelane", "v_readfirstlane")):
            kind = "lane"
        per[cur][kind] += 1
        total[kind] += 1
    print("total:", dict(total))
    rows = sorted(per.items(), key=lambda kv: (kv[0][0] if kv[0] else "", kv[0][1] if kv[0] else 0))
    for loc, c in rows:
        n = sum(c.values())
        if n >= min_count:
            print("%-22s %5d  valu %4d salu %4d lds %3d vmem %3d lane %3d wait %3d" % ("%s:%d" % loc if loc else "?", n, c["valu"], c["salu"], c["lds"], c["vmem"], c["lane"], c["wait"]))


if __name__ == "__main__":
    main()

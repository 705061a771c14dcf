#!/usr/bin/env python3
"""Run one polygonization per phase limit (1..8, 0) so that `rocprofv3 --pmc ... --kernel-trace` attributes
instruction counts to the phases of k_regular.  Usage (GPU box):
  rocprofv3 --pmc SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_WAVE_CYCLES --kernel-trace -d out -o p -- python tools/phase_pmc.py
  python tools/phase_pmc.py --report out/p_results.db"""
import os
import sqlite3
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
LIMS = [1, 2, 3, 4, 5, 6, 7, 8, 0]


def report(path):
    db = sqlite3.connect(path)
    tabs = [r[0] for r in db.execute("select name from sqlite_master where type='table'")]
    kd = [t for t in tabs if "kernel_dispatch" in t][0]
    ks = [t for t in tabs if "kernel_symbol" in t][0]
    pe = [t for t in tabs if "pmc_event" in t][0]
    pi = [t for t in tabs if "info_pmc" in t][0]
    rows = db.execute(f"select d.id, p.name, sum(e.value), d.end-d.start from {pe} e join {pi} p on e.pmc_id=p.id join {kd} d on e.event_id=d.event_id "
                      f"join {ks} s on d.kernel_id=s.id where s.kernel_name like '%k_regularILi640%' group by d.id, p.name order by d.id").fetchall()
    by = {}
    for did, name, val, dur in rows:
        by.setdefault(did, {})[name] = val
        by[did]["ns"] = dur
    ids = sorted(by)
    names = ["1 load", "2 prefix", "3 list+cells", "4 count", "5 scan v", "6 describe+emit verts", "7 keep", "8 scan i", "0 full(+indices)"]
    prev = {}
    print("%-24s %10s %12s %12s %12s %12s" % ("phase (cumulative->delta)", "us", "VALU", "SALU", "LDS", "VMEM_RD"))
    for k, did in enumerate(ids[-len(LIMS):]):
        cur = by[did]
        print("%-24s %10.0f" % (names[k], cur["ns"] / 1e3) + "".join(" %12.0f" % (cur.get(c, 0) - prev.get(c, 0)) for c in ("SQ_INSTS_VALU", "SQ_INSTS_SALU", "SQ_INSTS_LDS", "SQ_INSTS_VMEM_RD")))
        prev = cur


def main():
    if len(sys.argv) > 2 and sys.argv[1] == "--report":
        return report(sys.argv[2])
    from voxels_amd import Polygonizer, synth
    n = 1024
    d, m, b = synth.terrain(n)
    p = Polygonizer()
    p.upload(d, m, b, synth.block_empty_flags(d))
    p.set_stage_timing(True)  # serialised: one k_regular<640> dispatch per run
    p.execute(4)
    for lim in LIMS:
        p.debug_phase_limit(lim)
        p.execute(4)


if __name__ == "__main__":
    main()

#!/usr/bin/env python3
"""Timeline (start/end relative to the first kernel, microseconds) of the kernel dispatches of the LAST pipeline run in a
rocprofv3 rocpd sqlite output.  Usage: tools/rocpd_timeline.py DB [first_kernel_substring] [count]"""
import sqlite3
import sys


def main():
    db = sqlite3.connect(sys.argv[1])
    first = sys.argv[2] if len(sys.argv) > 2 else "k_run_head"
    tabs = [r[0] for r in db.execute("select name from sqlite_master where type='table'")]
    kd = [t for t in tabs if "kernel_dispatch" in t][0]
    ks = [t for t in tabs if "kernel_symbol" in t][0]
    cols = [r[1] for r in db.execute("pragma table_info(%s)" % kd)]
    q = "queue_id" if "queue_id" in cols else ("stream_id" if "stream_id" in cols else "0")
    rows = db.execute(f"select s.kernel_name, d.start, d.end, d.{q}, d.grid_size_x, d.workgroup_size_x from {kd} d join {ks} s on d.kernel_id = s.id order by d.start").fetchall()
    starts = [i for i, r in enumerate(rows) if first in r[0]]
    if not starts:
        print("no kernel matching", first)
        return
    which = int(sys.argv[3]) if len(sys.argv) > 3 else -2
    i0 = starts[which]
    i1 = starts[which + 1] if which + 1 < len(starts) and which + 1 != 0 else len(rows)
    t0 = rows[i0][1]
    for r in rows[i0:i1]:
        name = r[0].replace("_ZN12_GLOBAL__N_1", "")[:40]
        print("%-42s q%-3s start %9.1f  end %9.1f  dur %8.1f  grid %d" % (name, r[3], (r[1] - t0) / 1e3, (r[2] - t0) / 1e3, (r[2] - r[1]) / 1e3, r[4]))


if __name__ == "__main__":
    main()

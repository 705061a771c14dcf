#!/usr/bin/env python3
"""Per-kernel summary (calls, total/avg/min/max ns, % of GPU time, LDS/scratch/VGPR) of a rocprofv3 rocpd
sqlite output (`rocprofv3 --kernel-trace --stats -d DIR -o NAME -- cmd`), written as a text table.
Usage: tools/rocpd_summary.py gpurun_out/prof/NAME_results.db [out.txt] [--pmc]"""
import sqlite3
import sys


def main():
    path = sys.argv[1]
    out = sys.argv[2] if len(sys.argv) > 2 and not sys.argv[2].startswith("--") else None
    db = sqlite3.connect(path)
    tabs = [r[0] for r in db.execute("select name from sqlite_master where type='table'")]
    kd = [t for t in tabs if "kernel_dispatch" in t][0]
    ks = [t for t in tabs if "kernel_symbol" in t][0]
    rows = db.execute(
        f"select s.kernel_name, count(*), sum(d.end-d.start), avg(d.end-d.start), min(d.end-d.start), max(d.end-d.start), "
        f"max(d.group_segment_size), max(d.private_segment_size), max(s.arch_vgpr_count), max(s.sgpr_count), max(d.grid_size_x), max(d.workgroup_size_x) "
        f"from {kd} d join {ks} s on d.kernel_id = s.id group by s.kernel_name order by 3 desc").fetchall()
    total = sum(r[2] for r in rows) or 1
    lines = ["%-72s %7s %14s %12s %12s %12s %6s %8s %8s %5s %5s %10s" % ("kernel", "calls", "total_ns", "avg_ns", "min_ns", "max_ns", "pct", "lds_B", "scratch", "vgpr", "sgpr", "grid")]
    for r in rows:
        name = r[0]
        if len(name) > 72:
            name = name[:69] + "..."
        lines.append("%-72s %7d %14d %12.0f %12d %12d %6.2f %8d %8d %5d %5d %10d" % (name, r[1], r[2], r[3], r[4], r[5], 100.0 * r[2] / total, r[6], r[7], r[8], r[9], r[10]))
    if "--pmc" in sys.argv:
        pe = [t for t in tabs if "pmc_event" in t]
        pi = [t for t in tabs if "info_pmc" in t]
        if pe and pi:
            q = (f"select s.kernel_name, p.name, count(*), sum(e.value), avg(e.value) from {pe[0]} e "
                 f"join {pi[0]} p on e.pmc_id = p.id join {kd} d on e.event_id = d.event_id join {ks} s on d.kernel_id = s.id "
                 f"group by s.kernel_name, p.name order by 1, 2")
            lines.append("")
            lines.append("%-60s %-24s %8s %18s %16s" % ("kernel", "counter", "samples", "sum", "avg_per_dispatch"))
            for r in db.execute(q):
                lines.append("%-60s %-24s %8d %18.0f %16.1f" % (r[0][:60], r[1], r[2], r[3], r[4]))
    text = "\n".join(lines) + "\n"
    if out:
        open(out, "w").write(text)
    sys.stdout.write(text)


if __name__ == "__main__":
    main()

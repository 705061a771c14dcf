#!/bin/bash
# Kernel trace of one serialised polygonization of the bench terrain: every dispatch with its duration, in order.
# Usage (GPU box): bash tools/ktrace.sh <outdir> [n] [levels]
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
out=$1; mkdir -p "$out"
rocprofv3 --kernel-trace -d "$out/kt" -o k -- python tools/stage_times.py ${2:-1024} ${3:-4} 2 > "$out/kt.log" 2>&1
python - "$(find "$out/kt" -name '*.db' | head -1)" <<'PY'
import sqlite3, sys
db = sqlite3.connect(sys.argv[1])
tabs = [r[0] for r in db.execute("select name from sqlite_master where type='table'")]
kd = [t for t in tabs if "kernel_dispatch" in t][0]
ks = [t for t in tabs if "kernel_symbol" in t][0]
rows = db.execute(f"select s.kernel_name, d.start, d.end, d.grid_size_x from {kd} d join {ks} s on d.kernel_id = s.id order by d.start").fetchall()
rows = rows[-16:]
t0 = rows[0][1]
for name, a, b, g in rows:
    print("%-60s start %8.1f dur %8.1f us grid %d" % (name.replace("_ZN12_GLOBAL__N_1", "")[:60], (a - t0) / 1e3, (b - a) / 1e3, g))
PY
rm -rf "$out/kt"

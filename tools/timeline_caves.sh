#!/bin/bash
# Stream timeline of one step of the second ("caves") bench workload - the last overlapped one: three serialised runs for the
# stage times follow it in bench.py.  Usage (GPU box): bash tools/timeline_caves.sh <outdir>
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
out=$1; mkdir -p "$out"
rocprofv3 --kernel-trace -d "$out/ktc" -o k -- python bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-isolated > "$out/ktc.log" 2>&1
python tools/rocpd_timeline.py "$(find "$out/ktc" -name '*.db' | head -1)" k_run_head -4 > "$out/timeline_caves.txt" 2>&1
cat "$out/timeline_caves.txt" | head -30
rm -rf "$out/ktc"

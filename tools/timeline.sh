#!/bin/bash
# Stream timeline of one overlapped polygonization of the bench terrain (kernel trace of bench.py's normal mode).
# Usage (GPU box): bash tools/timeline.sh <outdir>
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
out=$1; mkdir -p "$out"
rocprofv3 --kernel-trace -d "$out/kt" -o k -- python bench.py --steps 6 --warmup 2 --no-cpu-baseline --no-extra --no-isolated > "$out/kt.log" 2>&1
python tools/rocpd_timeline.py "$(find "$out/kt" -name '*.db' | head -1)" > "$out/timeline_overlapped.txt" 2>&1
cat "$out/timeline_overlapped.txt"
rm -rf "$out/kt"

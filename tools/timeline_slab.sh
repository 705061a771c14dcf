#!/bin/bash
# Stream timeline of one rank's step of an N-rank run (one GPU): bash tools/timeline_slab.sh <outdir> [world=8] [rank=3]
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
out=$1; w=${2:-8}; r=${3:-3}; mkdir -p "$out"
rocprofv3 --kernel-trace -d "$out/kts" -o k -- python tools/slab_one.py $w $r y 1024 8 > "$out/kts.log" 2>&1
python tools/rocpd_timeline.py "$(find "$out/kts" -name '*.db' | head -1)" k_run_head -1 > "$out/timeline_slab_${w}_${r}.txt" 2>&1
cat "$out/timeline_slab_${w}_${r}.txt" | head -24; tail -2 "$out/kts.log"
rm -rf "$out/kts"

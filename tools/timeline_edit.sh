#!/bin/bash
# Stream timeline of one incremental run (BASELINE config 5 shape) on the three-launch path and on the chain of launches.
# Usage (GPU box): bash tools/timeline_edit.sh <outdir> [n=512] [levels=0]
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
out=$1; n=${2:-512}; lv=${3:-0}; mkdir -p "$out"
python tools/edit_run.py $n $lv 10 > "$out/edit_fused.txt" 2>&1
VX_DIRTY_FUSED=0 python tools/edit_run.py $n $lv 10 > "$out/edit_chain.txt" 2>&1
rocprofv3 --kernel-trace -d "$out/ktf" -o k -- python tools/edit_run.py $n $lv 6 > "$out/ktf.log" 2>&1
python tools/rocpd_timeline.py "$(find "$out/ktf" -name '*.db' | head -1)" k_dirty_head -1 > "$out/timeline_edit_fused.txt" 2>&1
VX_DIRTY_FUSED=0 rocprofv3 --kernel-trace -d "$out/ktc" -o k -- python tools/edit_run.py $n $lv 6 > "$out/ktc.log" 2>&1
python tools/rocpd_timeline.py "$(find "$out/ktc" -name '*.db' | head -1)" k_classify_blocks -1 > "$out/timeline_edit_chain.txt" 2>&1
rm -rf "$out/ktf" "$out/ktc"
grep -v amdgpu "$out/edit_fused.txt" | tail -3; grep -v amdgpu "$out/edit_chain.txt" | tail -3
cat "$out/timeline_edit_fused.txt"; cat "$out/timeline_edit_chain.txt"

#!/bin/bash
# Collect PMC counters for the bench kernels with rocprofv3 (counters in their own passes, kernel-trace only).
# Usage (on the GPU box): [PMC_CMD="python ..."] bash tools/pmc_run.sh <outdir> "<counters pass 1>" "<counters pass 2>" ...
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
out=$1; shift
mkdir -p "$out"
cmd=${PMC_CMD:-python bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-extra --no-isolated}
i=0
for c in "$@"; do
  i=$((i+1))
  rocprofv3 --pmc $c --kernel-trace -d "$out/pass$i" -o pmc -- $cmd > "$out/pass$i.log" 2>&1
  python tools/rocpd_summary.py "$(find "$out/pass$i" -name '*.db' | head -1)" "$out/pass$i.txt" --pmc > /dev/null 2>&1 || echo "summary failed for pass $i"
  rm -rf "$out/pass$i"
done

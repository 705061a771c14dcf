#!/bin/bash
# Per-kernel durations of the bench workload with the library's streams serialised (one kernel at a time), and
# optionally PMC counters of the same command (each counter set in its own pass, kernel-trace only).
# Usage (on the GPU box): bash tools/kstats.sh <outdir> ["<pmc counters pass 1>" "<pmc counters pass 2>" ...]
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
out=$1; shift
mkdir -p "$out"
rocprofv3 --kernel-trace -d "$out/kt" -o k -- python bench.py --steps 10 --warmup 2 --no-cpu-baseline --no-extra --no-isolated --serialize > "$out/kt.log" 2>&1
python tools/rocpd_summary.py "$(find "$out/kt" -name '*.db' | head -1)" "$out/kernel_stats_serialized.txt" | cut -c1-175 | head -14
rm -rf "$out/kt"
i=0
for c in "$@"; do
  i=$((i+1))
  rocprofv3 --pmc $c --kernel-trace -d "$out/pass$i" -o pmc -- python bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-extra --no-isolated --serialize > "$out/pass$i.log" 2>&1
  python tools/rocpd_summary.py "$(find "$out/pass$i" -name '*.db' | head -1)" "$out/pmc_pass$i.txt" --pmc | grep -E "k_main|k_tail|k_run_head|k_regular|k_transition|k_material|k_classify" | grep -v "^_ZN.*kd  " | cut -c1-130
  rm -rf "$out/pass$i"
done

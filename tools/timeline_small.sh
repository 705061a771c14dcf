#!/bin/bash
# Stream timeline of one overlapped polygonization of a SMALL grid (the fixed cost of a run: launches and their dependencies).
# Usage (GPU box): bash tools/timeline_small.sh <outdir> [n=128] [levels=4]
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
out=$1; n=${2:-128}; lv=${3:-4}; mkdir -p "$out"
python bench.py --n $n --levels $lv --steps 20 --warmup 3 --no-cpu-baseline --no-extra --no-isolated > "$out/bench_$n.json" 2> "$out/bench_$n.err"
python - "$out/bench_$n.json" <<'PY'
import json, sys
d = json.load(open(sys.argv[1]))
print("n", d["config"]["grid"], "ms_per_step", d["ms_per_step"], d["config"]["stage_ms_serialized"], "polygonize_ms", d["config"]["e2e_ms"]["polygonize_ms"])
PY
rocprofv3 --kernel-trace -d "$out/kt$n" -o k -- python bench.py --n $n --levels $lv --steps 6 --warmup 2 --no-cpu-baseline --no-extra --no-isolated > "$out/kt$n.log" 2>&1
python tools/rocpd_timeline.py "$(find "$out/kt$n" -name '*.db' | head -1)" > "$out/timeline_$n.txt" 2>&1
cat "$out/timeline_$n.txt" | head -24
rm -rf "$out/kt$n"

#!/bin/bash
cd "$GRAFT_REPO_ROOT"
out=gpurun_out/${1:-r03G}
mkdir -p $out
timeout 1200 python -m pytest tests -m gpu -q > $out/gputests.log 2>&1; echo "pytest rc $?" >> $out/gputests.log; tail -3 $out/gputests.log
timeout 600 python bench.py --no-cpu-baseline > $out/bench.json 2> $out/bench.err
python - $out/bench.json <<'PY'
import json, sys
d = json.load(open(sys.argv[1]))
print("step", d["ms_per_step"], "frac", d["roofline"]["frac"], d["config"]["stage_ms_serialized"])
print("extra", d["config"]["extra"]["ms_per_step"], d["config"]["extra"]["surface_block_share"])
PY
bash tools/timeline_caves.sh $out

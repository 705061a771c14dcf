#!/bin/bash
cd "$GRAFT_REPO_ROOT"
out=gpurun_out/${1:-r03v}
mkdir -p $out
timeout 1200 python -m pytest tests -m gpu -q > $out/gputests.log 2>&1; echo "pytest rc $?" >> $out/gputests.log; tail -5 $out/gputests.log
bash tools/timeline_small.sh $out 128 4
timeout 600 python bench.py --no-extra --no-cpu-baseline > $out/bench.json 2> $out/bench.err
python - $out/bench.json <<'PY'
import json, sys
d = json.load(open(sys.argv[1]))
print("step", d["ms_per_step"], "frac", d["roofline"]["frac"], d["config"]["stage_ms_serialized"])
PY

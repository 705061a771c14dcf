#!/bin/bash
cd "$GRAFT_REPO_ROOT"
out=gpurun_out/${1:-r03i}
mkdir -p $out
timeout 1200 python -m pytest tests -m gpu -q > $out/gputests.log 2>&1; echo "pytest rc $?" >> $out/gputests.log; tail -3 $out/gputests.log
timeout 900 python bench.py > $out/bench.json 2> $out/bench.err; tail -3 $out/bench.err
python - $out/bench.json <<'PY'
import json, sys
d = json.load(open(sys.argv[1]))
print("step", d["ms_per_step"], "value", d["value"], "frac", d["roofline"]["frac"])
print(d["config"]["stage_ms_serialized"])
print("cold", d["config"].get("cold"))
print("extra", d["config"].get("extra"))
print("e2e", d["config"]["e2e_ms"])
print("cpu", d.get("cpu_baseline"))
PY

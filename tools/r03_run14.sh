#!/bin/bash
cd "$GRAFT_REPO_ROOT"
out=gpurun_out/${1:-r03B}
mkdir -p $out
timeout 1200 python -m pytest tests -m gpu -q > $out/gputests.log 2>&1; echo "pytest rc $?" >> $out/gputests.log; tail -3 $out/gputests.log
timeout 600 python bench.py --no-extra --no-cpu-baseline > $out/bench.json 2> $out/bench.err
python - $out/bench.json <<'PY'
import json, sys
d = json.load(open(sys.argv[1]))
print("step", d["ms_per_step"], "frac", d["roofline"]["frac"], d["config"]["stage_ms_serialized"], d["config"]["e2e_ms"]["polygonize_ms"])
PY
timeout 300 bash tools/timeline.sh $out/tl > $out/tl.log 2>&1; head -16 $out/tl/timeline_overlapped.txt
timeout 300 bash tools/timeline_small.sh $out/tls 128 4 > $out/tls.log 2>&1; head -3 $out/tls.log
timeout 600 python tools/slab_time.py y > $out/slab_time_y.txt 2>&1; cat $out/slab_time_y.txt

#!/bin/bash
cd "$GRAFT_REPO_ROOT"
out=gpurun_out/${1:-r03k}
mkdir -p $out
timeout 300 python tools/fast0_debug.py 64 256 > $out/fast0_debug.txt 2>&1; tail -4 $out/fast0_debug.txt
timeout 1200 python -m pytest tests -m gpu -q -x > $out/gputests.log 2>&1; echo "pytest rc $?" >> $out/gputests.log; tail -3 $out/gputests.log
timeout 600 python bench.py --no-extra --no-cpu-baseline --serialize > $out/bench_ser.json 2> $out/bench_ser.err
VOXELS_HIP_LIBRARY=tools/ab/trprof.so timeout 600 python bench.py --steps 3 --warmup 1 --no-extra --no-cpu-baseline --serialize > $out/bench_trprof.json 2> $out/trprof.txt
grep "transition profile" $out/trprof.txt | tail -14
timeout 600 python bench.py --no-extra --no-cpu-baseline > $out/bench.json 2> $out/bench.err
python - $out/bench_ser.json $out/bench.json <<'PY'
import json, sys
for f in sys.argv[1:]:
    d = json.load(open(f))
    print("step", d["ms_per_step"], "frac", d["roofline"]["frac"], d["config"]["stage_ms_serialized"])
PY

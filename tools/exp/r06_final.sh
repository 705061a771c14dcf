#!/bin/bash
cd "$GRAFT_REPO_ROOT"
out=gpurun_out/final; mkdir -p $out
python -c "import __graft_entry__ as g; g.smoke()" > $out/smoke.txt 2>&1; echo "smoke rc $?" >> $out/smoke.txt
timeout 1500 python -m pytest tests -m gpu -q > $out/gputests.log 2>&1; echo "pytest rc $?" >> $out/gputests.log
timeout 900 python bench.py > $out/bench_default.json 2> $out/bench_default.err; echo "bench rc $?" >> $out/smoke.txt
tail -2 $out/smoke.txt; grep -E "passed|failed|rc " $out/gputests.log | tail -3; head -c 330 $out/bench_default.json; echo

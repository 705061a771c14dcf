#!/bin/bash
cd "$GRAFT_REPO_ROOT"
out=gpurun_out/t2; mkdir -p $out
timeout 1500 python -m pytest tests/test_gpu_parity.py -m gpu -x -q -k "config4" > $out/tests.log 2>&1; echo "pytest rc $?" >> $out/tests.log
tail -5 $out/tests.log
timeout 900 python bench.py --steps 20 --warmup 3 > $out/bench.json 2> $out/bench.err; echo "bench rc $?"; head -c 600 $out/bench.json; echo; tail -3 $out/bench.err

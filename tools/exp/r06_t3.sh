#!/bin/bash
cd "$GRAFT_REPO_ROOT"
out=gpurun_out/t3; mkdir -p $out
timeout 1800 python -m pytest tests -m gpu -x -q > $out/tests.log 2>&1; echo "pytest rc $?" >> $out/tests.log
grep -E "passed|failed|rc " $out/tests.log | tail -3
ABL_VARIANTS="a0" bash tools/exp/r06_ablate.sh 2>&1 | tail -2
QT_WORKLOADS=1024,128,slab timeout 300 python tools/quick_times.py - 2>&1 | grep -v amdgpu.ids

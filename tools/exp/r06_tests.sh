#!/bin/bash
cd "$GRAFT_REPO_ROOT"
out=gpurun_out/tests; mkdir -p $out
timeout 1800 python -m pytest tests -m gpu -x -q > $out/tests.log 2>&1; echo "pytest rc $?" >> $out/tests.log
tail -15 $out/tests.log
QT_WORKLOADS=1024,128,slab timeout 300 python tools/quick_times.py - 2>&1 | grep -v amdgpu.ids | tee $out/times.txt

#!/bin/bash
cd "$GRAFT_REPO_ROOT"
out=gpurun_out/exp3; mkdir -p $out
for i in 1 2; do
QT_WORKLOADS=1024,128,slab VX_MAIN_HEADS=1 timeout 300 python tools/quick_times.py - 2>&1 | grep -v amdgpu.ids
done > $out/times.txt 2>&1
cat $out/times.txt
timeout 1500 python -m pytest tests -m gpu -x -q > $out/tests.log 2>&1; echo "pytest rc $?" >> $out/tests.log
tail -5 $out/tests.log

#!/bin/bash
cd "$GRAFT_REPO_ROOT"
out=gpurun_out/exp10; mkdir -p $out
VOXELS_HIP_LIBRARY=tools/ab/trace.so python tools/prof_once.py 128 4 2> $out/t128.txt > /dev/null
awk '/==== last run ====/{on=1} on' $out/t128.txt | grep -v amdgpu > $out/trace128.txt
{
for i in 1 2; do
QT_WORKLOADS=1024,128,slab timeout 300 python tools/quick_times.py - 2>&1 | grep -v amdgpu.ids
VOXELS_HIP_LIBRARY=tools/ab/rbM.so QT_WORKLOADS=1024,128,slab timeout 300 python tools/quick_times.py - 2>&1 | grep -v amdgpu.ids
done
} > $out/times.txt 2>&1
cat $out/times.txt
timeout 1500 python -m pytest tests -m gpu -x -q > $out/tests.log 2>&1; echo "pytest rc $?" >> $out/tests.log
grep -E "passed|failed|rc " $out/tests.log | tail -3

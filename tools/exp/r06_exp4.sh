#!/bin/bash
# round 6, experiment 4: tickets drawn ahead in k_main; k_rebrick with batched loads and whole-line stores
cd "$GRAFT_REPO_ROOT"
out=gpurun_out/exp4; mkdir -p $out
{
python tools/rebrick_time.py 2>&1 | grep -v amdgpu.ids
VOXELS_HIP_LIBRARY=tools/ab/a0.so python tools/rebrick_time.py 2>&1 | grep -v amdgpu.ids
for i in 1 2; do
QT_WORKLOADS=1024,128,slab timeout 300 python tools/quick_times.py - 2>&1 | grep -v amdgpu.ids
VOXELS_HIP_LIBRARY=tools/ab/a0.so QT_WORKLOADS=1024,128,slab timeout 300 python tools/quick_times.py - 2>&1 | grep -v amdgpu.ids
done
} > $out/times.txt 2>&1
cat $out/times.txt
timeout 1500 python -m pytest tests -m gpu -x -q > $out/tests.log 2>&1; echo "pytest rc $?" >> $out/tests.log
tail -5 $out/tests.log

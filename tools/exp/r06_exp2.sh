#!/bin/bash
cd "$GRAFT_REPO_ROOT"
out=gpurun_out/exp2; mkdir -p $out
for i in 1 2; do
QT_WORKLOADS=1024,128 VX_MAIN_HEADS=1 timeout 300 python tools/quick_times.py - 2>&1 | grep -v amdgpu.ids
VOXELS_HIP_LIBRARY=tools/ab/abl_noself.so QT_WORKLOADS=1024,128 VX_MAIN_HEADS=1 timeout 300 python tools/quick_times.py - 2>&1 | grep -v amdgpu.ids
done > $out/times.txt 2>&1
cat $out/times.txt

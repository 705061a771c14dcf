#!/bin/bash
cd "$GRAFT_REPO_ROOT"
out=gpurun_out/exp5; mkdir -p $out
{
for i in 1 2; do
QT_WORKLOADS=1024 timeout 600 python tools/quick_times.py VX_MAIN_AHEAD=0 VX_MAIN_AHEAD=1 VX_MAIN_AHEAD=2 VX_MAIN_AHEAD=3 2>&1 | grep -v amdgpu.ids
VOXELS_HIP_LIBRARY=tools/ab/a0.so QT_WORKLOADS=1024 timeout 300 python tools/quick_times.py - 2>&1 | grep -v amdgpu.ids
done
} > $out/times.txt 2>&1
cat $out/times.txt

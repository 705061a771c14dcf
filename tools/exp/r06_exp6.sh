#!/bin/bash
cd "$GRAFT_REPO_ROOT"
out=gpurun_out/exp6; mkdir -p $out
{
for i in 1 2; do
for v in ${VARIANTS:-new0 a32768 a0}; do
echo -n "$v: "; VOXELS_HIP_LIBRARY=tools/ab/$v.so QT_WORKLOADS=${QT_WORKLOADS:-1024} timeout 300 python tools/quick_times.py - 2>&1 | grep -v amdgpu.ids
done
done
} > $out/times.txt 2>&1
cat $out/times.txt

#!/bin/bash
cd "$GRAFT_REPO_ROOT"
out=gpurun_out/exp8; mkdir -p $out
{
for v in ${VARIANTS:-rb4 rbA1 rbA2 rbA3 rbA7}; do VOXELS_HIP_LIBRARY=tools/ab/$v.so python tools/rebrick_time.py 2>&1 | grep -v amdgpu.ids; done
} > $out/times.txt 2>&1
cat $out/times.txt

#!/bin/bash
cd "$GRAFT_REPO_ROOT"
out=gpurun_out/trace; mkdir -p $out
VOXELS_HIP_LIBRARY=tools/ab/trace.so python tools/prof_once.py 128 4 2> $out/t128.txt > /dev/null
awk '/==== last run ====/{on=1} on' $out/t128.txt | grep -v amdgpu > $out/trace128.txt
wc -l $out/trace128.txt

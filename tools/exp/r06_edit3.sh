#!/bin/bash
cd "$GRAFT_REPO_ROOT"
out=gpurun_out/edit3; mkdir -p $out
for i in 1 2; do
for v in base.so ../../voxels_amd/csrc/libvoxels_hip.so; do
echo "== $v"
VOXELS_HIP_LIBRARY=tools/ab/$v VX_HOST_TIMING=1 timeout 600 python tools/bench_edit.py 512 2> $out/ht.txt | grep "steady state, fused:"
grep "vx host, dirty\] lists" $out/ht.txt | sed -n 11,13p
done
done

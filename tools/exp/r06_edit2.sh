#!/bin/bash
cd "$GRAFT_REPO_ROOT"
out=gpurun_out/edit2; mkdir -p $out
for v in base.so ../../voxels_amd/csrc/libvoxels_hip.so; do
echo "== $v"
VOXELS_HIP_LIBRARY=tools/ab/$v VX_HOST_TIMING=1 timeout 600 python tools/bench_edit.py 512 2> $out/ht.txt | grep "steady state, fused:"
grep "vx host, dirty\] lists" $out/ht.txt | sed -n 8,14p
done
timeout 1500 python -m pytest tests -m gpu -x -q 2>&1 | grep -E "passed|failed"
timeout 400 python tools/fuzz_parity.py 120 81000 edits 2>&1 | grep -v amdgpu.ids | tail -1
VX_FUZZ_CHAIN=40 VX_FUZZ_N=128 timeout 400 python tools/fuzz_parity.py 100 82000 edits 2>&1 | grep -v amdgpu.ids | tail -1

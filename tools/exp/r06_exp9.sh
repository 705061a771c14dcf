#!/bin/bash
cd "$GRAFT_REPO_ROOT"
out=gpurun_out/exp9; mkdir -p $out
{
for v in rbM rbL rbM rbL; do VOXELS_HIP_LIBRARY=tools/ab/$v.so python tools/rebrick_time.py 2>&1 | grep -v amdgpu.ids; done
} > $out/times.txt 2>&1
cat $out/times.txt
timeout 1500 python -m pytest tests -m gpu -x -q > $out/tests.log 2>&1; echo "pytest rc $?" >> $out/tests.log
grep -E "passed|failed|rc " $out/tests.log | tail -3

#!/bin/bash
cd "$GRAFT_REPO_ROOT"
out=gpurun_out/fuzz8; mkdir -p $out
{
echo "== round 6, final build, third sweep (tools/fuzz_parity.py 400 91000 | tools/fuzz_slabs.py 300 92000 | tools/fuzz_parity.py 200 93000 edits | tools/stress_runs.py 60000 6000)"
timeout 700 python tools/fuzz_parity.py 400 91000 2>&1 | grep -v amdgpu.ids | tail -1
timeout 600 python tools/fuzz_slabs.py 300 92000 2>&1 | grep -v amdgpu.ids | tail -1
timeout 500 python tools/fuzz_parity.py 200 93000 edits 2>&1 | grep -v amdgpu.ids | tail -1
timeout 600 python tools/stress_runs.py 60000 6000 2>&1 | grep -v amdgpu.ids | tail -5
} > $out/fuzz.txt 2>&1
cat $out/fuzz.txt

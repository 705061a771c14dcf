#!/bin/bash
cd "$GRAFT_REPO_ROOT"
out=gpurun_out/fuzz7; mkdir -p $out
{
echo "== round 6, final build, second sweep (tools/fuzz_parity.py 300 71000 | tools/fuzz_parity.py 200 72000 edits | tools/fuzz_slabs.py 250 73000 | VX_DIRTY_FUSED=0 VX_FUZZ_CHAIN=20 tools/fuzz_parity.py 100 74000 edits | VX_UPPER=0 tools/fuzz_parity.py 100 75000)"
timeout 600 python tools/fuzz_parity.py 300 71000 2>&1 | grep -v amdgpu.ids | tail -2
timeout 500 python tools/fuzz_parity.py 200 72000 edits 2>&1 | grep -v amdgpu.ids | tail -2
timeout 600 python tools/fuzz_slabs.py 250 73000 2>&1 | grep -v amdgpu.ids | tail -2
VX_DIRTY_FUSED=0 VX_FUZZ_CHAIN=20 timeout 400 python tools/fuzz_parity.py 100 74000 edits 2>&1 | grep -v amdgpu.ids | tail -2
VX_UPPER=0 timeout 400 python tools/fuzz_parity.py 100 75000 2>&1 | grep -v amdgpu.ids | tail -2
} > $out/fuzz.txt 2>&1
cat $out/fuzz.txt

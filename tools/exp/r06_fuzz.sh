#!/bin/bash
cd "$GRAFT_REPO_ROOT"
out=gpurun_out/fuzz6; mkdir -p $out
{
echo "== round 6, build of $(date -u +%F) (kernel parameters read afresh per item, transition vertices from cellBits, k_rebrick with whole-tile stores and LDS-collected lattice rows) (tools/fuzz_parity.py 150 61000 | tools/fuzz_parity.py 100 62000 edits | tools/fuzz_slabs.py 150 63000 | VX_FUZZ_CHAIN=40 VX_FUZZ_N=128 tools/fuzz_parity.py 100 64000 edits)"
timeout 400 python tools/fuzz_parity.py 150 61000 2>&1 | grep -v amdgpu.ids | tail -2
timeout 400 python tools/fuzz_parity.py 100 62000 edits 2>&1 | grep -v amdgpu.ids | tail -2
timeout 400 python tools/fuzz_slabs.py 150 63000 2>&1 | grep -v amdgpu.ids | tail -2
VX_FUZZ_CHAIN=40 VX_FUZZ_N=128 timeout 400 python tools/fuzz_parity.py 100 64000 edits 2>&1 | grep -v amdgpu.ids | tail -2
} > $out/fuzz.txt 2>&1
cat $out/fuzz.txt

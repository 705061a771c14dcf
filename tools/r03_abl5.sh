#!/bin/bash
cd "$GRAFT_REPO_ROOT"
source tools/r03_abl_fn.sh
out=gpurun_out/${1:-r03u}; mkdir -p $out
B=voxels_amd/csrc/libvoxels_hip.so
run base $B A=1
run f09 tools/ab/f09.so A=1
run f09_w4 tools/ab/f09.so VX_REG_WGS_PER_CU=4

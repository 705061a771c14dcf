#!/bin/bash
cd "$GRAFT_REPO_ROOT"
source tools/r03_abl_fn.sh
out=gpurun_out/${1:-r03x}; mkdir -p $out
B=voxels_amd/csrc/libvoxels_hip.so
run base $B A=1
run nb tools/ab/nb.so A=1
run nb_w25 tools/ab/nb.so VX_REG_WGS_PER_CU=25
run nb_w15 tools/ab/nb.so VX_REG_WGS_PER_CU=15

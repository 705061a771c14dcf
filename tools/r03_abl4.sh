#!/bin/bash
cd "$GRAFT_REPO_ROOT"
source tools/r03_abl_fn.sh
out=gpurun_out/${1:-r03r}; mkdir -p $out
B=voxels_amd/csrc/libvoxels_hip.so
run base $B A=1
run f1w5 $B VX_F1_WGS_PER_CU=5
run f1w6 $B VX_F1_WGS_PER_CU=6
run f1w10 $B VX_F1_WGS_PER_CU=10
run tr1280 $B VX_TR_GRID=1280
run tr1536 $B VX_TR_GRID=1536
run tr1024 $B VX_TR_GRID=1024
run r0w4 $B VX_REG_WGS_PER_CU=4
run r0w8 $B VX_REG_WGS_PER_CU=8

#!/bin/bash
cd "$GRAFT_REPO_ROOT"
source tools/r03_abl_fn.sh
out=gpurun_out/${1:-r03q}; mkdir -p $out
run tr5 tools/ab/tr5.so A=1
run tr7 tools/ab/tr7.so A=1
run tr8 tools/ab/tr8.so A=1
run tr9 tools/ab/tr9.so A=1

#!/bin/bash
cd "$GRAFT_REPO_ROOT"
out=gpurun_out/edit; mkdir -p $out
VX_HOST_TIMING=1 timeout 600 python tools/bench_edit.py 512 > $out/bench_edit.txt 2> $out/host_timing.txt
tail -25 $out/bench_edit.txt
grep "vx host, dirty" $out/host_timing.txt | tail -40

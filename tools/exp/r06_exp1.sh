#!/bin/bash
# round 6, experiment 1: per-XCD level-0 heads (VX_MAIN_HEADS=8, the default) against one head - times, counters, in-kernel profiles
cd "$GRAFT_REPO_ROOT"
out=gpurun_out/exp1; mkdir -p $out
timeout 600 python -m pytest tests/test_gpu_parity.py -m gpu -x -q -k "config2 or config3 or terrain or 1024 or sphere or conservative" > $out/tests.log 2>&1; echo "pytest rc $?" >> $out/tests.log
QT_WORKLOADS=1024 timeout 900 python tools/quick_times.py - VX_MAIN_HEADS=1 - VX_MAIN_HEADS=1 VX_MAIN_GRANULE=8 VX_MAIN_GRANULE=16 VX_MAIN_GRANULE=64 VX_MAIN_GRANULE=128 VX_MAIN_GRANULE=64,VX_MAIN_BATCH=4 VX_MAIN_UPPER_NUM=1,VX_MAIN_UPPER_DEN=3 2>&1 | grep -v amdgpu.ids > $out/quick_times.txt
QT_WORKLOADS=128,slab timeout 300 python tools/quick_times.py - VX_MAIN_HEADS=1 2>&1 | grep -v amdgpu.ids >> $out/quick_times.txt
timeout 600 bash tools/pmc_run.sh $out/pmc8 "FETCH_SIZE" "WRITE_SIZE" "TCC_HIT_sum TCC_MISS_sum" > $out/pmc8.log 2>&1
VX_MAIN_HEADS=1 timeout 600 bash tools/pmc_run.sh $out/pmc1 "FETCH_SIZE" "WRITE_SIZE" "TCC_HIT_sum TCC_MISS_sum" > $out/pmc1.log 2>&1
for v in mainprof f0prof trprof; do
  VOXELS_HIP_LIBRARY=tools/ab/$v.so QT_WORKLOADS=1024 timeout 300 python tools/quick_times.py - 2>&1 | grep -v amdgpu.ids | tail -14 > $out/$v.txt
done
cat $out/quick_times.txt; tail -3 $out/tests.log
grep -h k_main $out/pmc8/pass*.txt | cut -c1-160; grep -h k_main $out/pmc1/pass*.txt | cut -c1-160
cat $out/mainprof.txt $out/f0prof.txt $out/trprof.txt

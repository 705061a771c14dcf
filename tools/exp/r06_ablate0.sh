#!/bin/bash
# round 6: what the parts of k_main cost - variant libraries with one part switched off (tools/ab/a<bits>.so, -DVX_ABL=<bits>):
# step time (quick_times.py) and SQ_INSTS_VALU / SQ_INSTS_SALU / SQ_INSTS_LDS per launch (one counter pass each)
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
out=gpurun_out/ablate0; mkdir -p $out
for v in ${ABL_VARIANTS:-p0 p16 p17 p18 p19 p3}; do
  lib=tools/ab/$v.so
  t=$(VOXELS_HIP_LIBRARY=$lib QT_WORKLOADS=1024 timeout 200 python tools/quick_times.py - 2>&1 | grep -v amdgpu.ids | tail -1)
  VOXELS_HIP_LIBRARY=$lib rocprofv3 --pmc SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_ACTIVE_INST_VALU --kernel-trace -d $out/p_$v -o pmc -- python tools/prof_once.py 1024 4 > $out/p_$v.log 2>&1
  python tools/rocpd_summary.py "$(find $out/p_$v -name '*.db' | head -1)" $out/p_$v.txt --pmc > /dev/null 2>&1
  rm -rf $out/p_$v
  echo "$v | $t | $(grep -E 'k_mainILb0ELb0' $out/p_$v.txt | grep -E 'SQ_INSTS_VALU|SQ_INSTS_SALU|SQ_INSTS_LDS|SQ_WAIT_ANY|SQ_WAVE_CYCLES|SQ_ACTIVE_INST_VALU' | awk '{printf "%s %.1fM/launch; ", $2, $4/$3*32/1e6}')"
done | tee $out/summary.txt

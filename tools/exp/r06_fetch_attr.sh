#!/bin/bash
# round 6: where k_main's fetches come from - FETCH_SIZE / WRITE_SIZE / L2 hits of the ablation builds (tools/ab/a<bits>.so, see r06_ablate.sh for the bits),
# and the lane utilisation of the product build (SQ_THREAD_CYCLES_VALU against SQ_ACTIVE_INST_VALU)
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
out=gpurun_out/fetch; mkdir -p $out
pass() { # lib tag counters...
  lib=$1; tag=$2; shift 2
  VOXELS_HIP_LIBRARY=$lib rocprofv3 --pmc "$@" --kernel-trace -d $out/p_$tag -o pmc -- python tools/prof_once.py 1024 4 > $out/p_$tag.log 2>&1
  python tools/rocpd_summary.py "$(find $out/p_$tag -name '*.db' | head -1)" $out/p_$tag.txt --pmc > /dev/null 2>&1
  rm -rf $out/p_$tag
  echo "$tag | $(grep -E 'k_mainILb0ELb0' $out/p_$tag.txt | awk 'NF==5 {printf "%s %.1f; ", $2, $5}')"
}
for v in ${ABL_VARIANTS:-a0 a128 a4 a8 a16 a12}; do
  [ -f tools/ab/$v.so ] || continue
  pass tools/ab/$v.so ${v}_fetch FETCH_SIZE
  pass tools/ab/$v.so ${v}_tcc TCC_HIT_sum TCC_MISS_sum
done | tee $out/summary.txt
pass tools/ab/a0.so a0_lanes SQ_THREAD_CYCLES_VALU SQ_ACTIVE_INST_VALU SQ_INSTS_VALU | tee -a $out/summary.txt
pass tools/ab/a0.so a0_busy SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_WAIT_INST_ANY SQ_WAIT_ANY | tee -a $out/summary.txt
pass tools/ab/a0.so a0_lds SQ_INSTS_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_ACTIVE_INST_LDS | tee -a $out/summary.txt
pass tools/ab/a0.so a0_mem SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_SMEM SQ_INSTS_FLAT | tee -a $out/summary.txt
for v in mainprof f0prof trprof; do
  VOXELS_HIP_LIBRARY=tools/ab/$v.so QT_WORKLOADS=1024 timeout 300 python tools/quick_times.py - 2>&1 | grep -v amdgpu.ids | tail -14 > $out/$v.txt
done
cat $out/mainprof.txt $out/f0prof.txt $out/trprof.txt

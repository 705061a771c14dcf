#!/bin/bash
# round 6: instruction-cache behaviour of k_main (99 KB of code against a 64 KB instruction cache shared by two CUs)
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
out=gpurun_out/icache; mkdir -p $out
pass() { # tag counters...
  tag=$1; shift
  rocprofv3 --pmc "$@" --kernel-trace -d $out/p_$tag -o pmc -- python tools/prof_once.py 1024 4 > $out/p_$tag.log 2>&1
  python tools/rocpd_summary.py "$(find $out/p_$tag -name '*.db' | head -1)" $out/p_$tag.txt --pmc > /dev/null 2>&1
  rm -rf $out/p_$tag
  echo "$tag | $(grep -E 'k_mainILb0ELb0' $out/p_$tag.txt | awk 'NF==5 {printf "%s %.1f; ", $2, $5}')"
}
pass ic1 SQC_ICACHE_REQ SQC_ICACHE_HITS SQC_ICACHE_MISSES SQC_ICACHE_MISSES_DUPLICATE | tee $out/summary.txt
pass ic2 SQ_IFETCH SQ_IFETCH_LEVEL SQC_TC_INST_REQ SQC_ICACHE_BUSY_CYCLES | tee -a $out/summary.txt
pass ic3 SQ_WAIT_INST_ANY SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_SALU | tee -a $out/summary.txt
pass ic4 SQ_INST_CYCLES_VMEM SQ_INST_CYCLES_SALU SQ_INST_CYCLES_SMEM SQ_ACTIVE_INST_ANY | tee -a $out/summary.txt
pass ic5 SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_MISC | tee -a $out/summary.txt
pass ic6 SQ_INSTS_BRANCH SQ_INSTS_SENDMSG SQ_INSTS_VSKIPPED SQ_INSTS_EXP_GDS | tee -a $out/summary.txt
pass ic7 SQ_WAIT_INST_LDS SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_FLAT SQ_INST_LEVEL_LDS | tee -a $out/summary.txt
grep -E "k_rebrick" $out/p_ic1.txt | awk 'NF==5 {printf "rebrick %s %.1f; ", $2, $5}'; echo

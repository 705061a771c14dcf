#!/bin/bash
# Everything a round's profiles/ holds, from ONE GPU call: GPU test log, default bench line, serialised kernel stats + SQ
# counter passes, HBM traffic passes (FETCH_SIZE / WRITE_SIZE / TCC hit-miss) of the headline workload and of the second
# ('caves') workload, stream timelines (1024^3, 128^3, one rank's slab of an 8-rank job, caves), per-rank slab times, the
# small-run times, a stress of 100 000 consecutive runs, the grid-file decode kernel alone, device arithmetic self-test.  Usage (GPU box): bash tools/profiles_run.sh <outdir>
# Afterwards, here: bash tools/profiles_collect.sh <outdir> <label>   (copies the summaries into profiles/<label>_*)
cd "$GRAFT_REPO_ROOT"
out=${1:-gpurun_out/prof}
mkdir -p $out
python -c "from voxels_amd import Polygonizer; p = Polygonizer(); print('selftest', p.selftest().tolist())" > $out/selftest.txt 2>&1
timeout 1200 python -m pytest tests -m gpu -q > $out/gputests.log 2>&1; echo "pytest rc $?" >> $out/gputests.log
timeout 900 python bench.py > $out/bench_default.json 2> $out/bench_default.err
timeout 600 bash tools/kstats.sh $out/ks "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR" "SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_SCA SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_LDS_BANK_CONFLICT SQ_INSTS_SMEM" "SQ_THREAD_CYCLES_VALU SQ_ACTIVE_INST_VALU SQ_INSTS_VALU SQ_INSTS_SALU" > $out/ks.log 2>&1
timeout 600 bash tools/pmc_run.sh $out/pmc "FETCH_SIZE" "WRITE_SIZE" "TCC_HIT_sum TCC_MISS_sum" > $out/pmc.log 2>&1
PMC_CMD="python tools/caves_run.py 1024 4 3" timeout 600 bash tools/pmc_run.sh $out/pmc_caves "FETCH_SIZE" "WRITE_SIZE" > $out/pmc_caves.log 2>&1
timeout 300 bash tools/timeline.sh $out/tl > $out/tl.log 2>&1
timeout 300 bash tools/timeline_small.sh $out/tls 128 4 > $out/tls.log 2>&1
timeout 300 bash tools/timeline_slab.sh $out/tlslab 8 3 > $out/tlslab.log 2>&1
timeout 300 bash tools/timeline_caves.sh $out/tlc > $out/tlc.log 2>&1
timeout 400 bash tools/timeline_edit.sh $out/tle 512 0 > $out/tle.log 2>&1
timeout 400 python tools/bench_edit.py 512 2>&1 | grep -v amdgpu.ids > $out/bench_edit.txt
timeout 600 python tools/slab_time.py y > $out/slab_time_y.txt 2>&1
timeout 300 python tools/quick_times.py > $out/quick_times.txt 2>&1
timeout 300 python tools/rebrick_time.py 1024 8 2>&1 | grep -v amdgpu.ids > $out/rebrick_time.txt
[ -f tools/ab/trace.so ] && VOXELS_HIP_LIBRARY=tools/ab/trace.so timeout 300 python tools/prof_once.py 128 4 2>&1 > /dev/null | awk '/==== last run ====/{on=1} on' | grep -v amdgpu > $out/main_trace_128.txt
timeout 300 python tools/stress_runs.py 40000 4000 2>&1 | grep -v amdgpu.ids > $out/stress.txt
(cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT" && timeout 300 rocprofv3 --kernel-trace --stats -d $out/dec -o k -- python tools/decode_only.py > $out/dec.log 2>&1; python tools/rocpd_summary.py "$(find $out/dec -name '*.db' | head -1)" $out/decode_kernel_stats_full.txt > /dev/null 2>&1; grep -v "^W2026\|simple_timer" $out/decode_kernel_stats_full.txt | cut -c1-150 | head -8 > $out/decode_kernel_stats.txt; rm -rf $out/dec)
tail -3 $out/gputests.log; cat $out/selftest.txt; head -c 400 $out/bench_default.json; echo; head -12 $out/ks.log; tail -4 $out/slab_time_y.txt; tail -1 $out/quick_times.txt

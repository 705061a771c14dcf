#!/bin/bash
# Everything a round's profiles/ holds, from one GPU call: GPU test log, default bench line, serialised kernel stats + SQ
# counter passes, HBM traffic passes (FETCH_SIZE / WRITE_SIZE / TCC hit-miss), stream timelines (1024^3 and 128^3),
# per-rank slab times, device arithmetic self-test.  Usage (GPU box): bash tools/profiles_run.sh <outdir>
cd "$GRAFT_REPO_ROOT"
out=${1:-gpurun_out/prof}
mkdir -p $out
python -c "from voxels_amd import Polygonizer; p = Polygonizer(); print('selftest', p.selftest().tolist())" > $out/selftest.txt 2>&1
timeout 1200 python -m pytest tests -m gpu -q > $out/gputests.log 2>&1; echo "pytest rc $?" >> $out/gputests.log
timeout 900 python bench.py > $out/bench_default.json 2> $out/bench_default.err
timeout 900 bash tools/kstats.sh $out/ks "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR" "SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_SCA SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_LDS_BANK_CONFLICT SQ_INSTS_SMEM" > $out/ks.log 2>&1
timeout 900 bash tools/pmc_run.sh $out/pmc "FETCH_SIZE" "WRITE_SIZE" "TCC_HIT_sum TCC_MISS_sum" > $out/pmc.log 2>&1
timeout 300 bash tools/timeline.sh $out/tl > $out/tl.log 2>&1
timeout 300 bash tools/timeline_small.sh $out/tls 128 4 > $out/tls.log 2>&1
timeout 600 python tools/slab_time.py y > $out/slab_time_y.txt 2>&1
tail -3 $out/gputests.log; cat $out/selftest.txt; head -c 400 $out/bench_default.json; echo; head -12 $out/ks.log; cat $out/slab_time_y.txt

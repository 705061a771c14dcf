#!/bin/bash
# round 3, first GPU call: instruction issue rates, the new level-0 pass against the general one, the GPU suite, bench, profiles
cd "$GRAFT_REPO_ROOT"
out=gpurun_out/${1:-r03b}
mkdir -p $out
./tools/valu_rate > $out/valu_rate.txt 2>&1
timeout 300 python tools/fast0_debug.py 64 256 512 > $out/fast0_debug.txt 2>&1
python -c "from voxels_amd import Polygonizer; p = Polygonizer(); print('selftest', p.selftest().tolist())" > $out/selftest.txt 2>&1
timeout 1200 python -m pytest tests -m gpu -q > $out/gputests.log 2>&1
echo "pytest rc $?" >> $out/gputests.log
timeout 600 python bench.py > $out/bench.json 2> $out/bench.err
VX_FAST0=0 VX_FAST1=0 timeout 300 python bench.py --steps 30 --no-cpu-baseline > $out/bench_nofast.json 2> $out/bench_nofast.err
timeout 900 bash tools/kstats.sh $out/ks "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR" "SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_SCA SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_LDS_BANK_CONFLICT SQ_INSTS_SMEM" > $out/ks.log 2>&1
tail -3 $out/gputests.log; cat $out/selftest.txt; cat $out/fast0_debug.txt | tail -8; cat $out/valu_rate.txt | tail -9; head -c 600 $out/bench.json; echo; head -c 300 $out/bench_nofast.json; echo; tail -30 $out/ks.log

#!/bin/bash
# After `gpurun -- bash tools/profiles_run.sh gpurun_out/prof`: copy the summaries into profiles/<label>_* (tracked) and
# rebuild profiles/pmc_latest.json / pmc_caves_latest.json from the counter passes.  Usage: bash tools/profiles_collect.sh <dir> <label>
set -e
cd "$(dirname "$0")/.."
d=${1:-gpurun_out/prof}; l=${2:-r04}
cp $d/gputests.log profiles/${l}_gputests.log
cp $d/selftest.txt profiles/${l}_selftest.txt
cp $d/bench_default.json profiles/${l}_bench_default.json
cp $d/ks/kernel_stats_serialized.txt profiles/${l}_kernel_stats_serialized.txt
for i in 1 2 3; do [ -f $d/ks/pmc_pass$i.txt ] && cp $d/ks/pmc_pass$i.txt profiles/${l}_sq_pass$i.txt; done
for i in 1 2 3; do cp $d/pmc/pass$i.txt profiles/${l}_pmc_pass$i.txt; done
for i in 1 2; do cp $d/pmc_caves/pass$i.txt profiles/${l}_pmc_caves_pass$i.txt; done
cp $d/tl/timeline_overlapped.txt profiles/${l}_timeline_1024.txt
cp $d/tls/timeline_128.txt profiles/${l}_timeline_128.txt
grep -v "^W2026\|simple_timer" $d/tlslab.log > profiles/${l}_timeline_slab_8_3.txt
cp $d/tlc/timeline_caves.txt profiles/${l}_timeline_caves.txt
cp $d/slab_time_y.txt profiles/${l}_slab_time_y.txt
cp $d/quick_times.txt profiles/${l}_quick_times.txt
[ -f $d/rebrick_time.txt ] && cp $d/rebrick_time.txt profiles/${l}_rebrick_time.txt
[ -s $d/main_trace_128.txt ] && cp $d/main_trace_128.txt profiles/${l}_main_trace_128.txt
cp $d/stress.txt profiles/${l}_stress.txt
cp $d/decode_kernel_stats.txt profiles/${l}_decode_kernel_stats.txt
cp $d/tle/timeline_edit_fused.txt profiles/${l}_timeline_edit_fused.txt
head -20 $d/tle/timeline_edit_chain.txt > profiles/${l}_timeline_edit_chain.txt
(grep -v amdgpu $d/tle/edit_fused.txt | tail -3; echo "--- VX_DIRTY_FUSED=0"; grep -v amdgpu $d/tle/edit_chain.txt | tail -3) > profiles/${l}_edit_times.txt
cp $d/bench_edit.txt profiles/${l}_bench_edit.txt
python tools/pmc_json.py $d/pmc 1024 4 1 $l profiles/pmc_latest.json
python tools/pmc_json.py $d/pmc_caves 1024 4 1 ${l}_caves profiles/pmc_caves_latest.json "python tools/caves_run.py 1024 4 3"
ls profiles | grep "^${l}_"

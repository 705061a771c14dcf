cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
mkdir -p gpurun_out/c40
timeout 900 python bench.py > gpurun_out/c40/bench.json 2> gpurun_out/c40/bench.err; head -c 150 gpurun_out/c40/bench.json; echo
rocprofv3 --kernel-trace -d gpurun_out/c40/kt -o k -- python tools/partial_time.py > gpurun_out/c40/kt.log 2>&1
python tools/rocpd_timeline.py "$(find gpurun_out/c40/kt -name '*.db' | head -1)" k_run_head -1 | head -6
rm -rf gpurun_out/c40/kt

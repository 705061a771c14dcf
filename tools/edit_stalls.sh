# Sporadic 75-95 ms calls in edit sequences, explained: the container's CPU bandwidth quota (cpu.max, e.g. 16 CPUs' worth per
# 100 ms on a 256-CPU host).  A burst of host threads right in front of the edits - here the host generator used to find the
# surface height, EDIT_HOST_GENERATOR=1 - exhausts the period's quota and every thread of the process sleeps until the period
# ends, inside whatever it was doing (a kernel launch, the wait, an allocation); nr_throttled counts it.  Without the burst
# (the height is read from the resident grid) no call stalls.  Usage (GPU box): bash tools/edit_stalls.sh
mkdir -p gpurun_out/stalls
(
echo "cpu.max: $(cat /sys/fs/cgroup/cpu.max 2>/dev/null)   CPUs visible: $(nproc)"
for gen in 1 1 1 1 "" "" "" ""; do
  echo "== EDIT_HOST_GENERATOR=${gen:-0}: 150 edits"
  EDIT_HOST_GENERATOR=$gen VX_HOST_TIMING=1 timeout 200 python tools/edit_outliers.py 150 2>&1 | grep -E "^call|lists \+ box|throttled" | awk '/throttled/ { print } /lists/ { line = $0 } /^call/ { n++; s+=$3; if ($3+0 > 1.5) { slow++; print line; print } } END { printf("calls %d, mean %.3f ms, slower than 1.5 ms: %d\n", n, s/n, slow) }' | cut -c1-230
done
) > gpurun_out/stalls/o.txt 2>&1
cat gpurun_out/stalls/o.txt

#!/bin/bash
# FETCH_SIZE / WRITE_SIZE calibration on known access patterns (tools/pmc_calib.hip). Usage (GPU box): bash tools/pmc_calib.sh <outdir>
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
out=$1; mkdir -p "$out"
[ -x tools/pmc_calib ] || hipcc --offload-arch=gfx950 -O3 -o tools/pmc_calib tools/pmc_calib.hip || exit 1
for c in FETCH_SIZE WRITE_SIZE; do
  rocprofv3 --pmc $c --kernel-trace -d "$out/calib_$c" -o calib -- tools/pmc_calib > "$out/calib_expected.txt" 2> "$out/calib_$c.log"
  python tools/rocpd_summary.py "$(find "$out/calib_$c" -name '*.db' | head -1)" "$out/calib_$c.txt" --pmc > /dev/null
  rm -rf "$out/calib_$c"
done
python - "$out" <<'PY'
import re, sys
out = sys.argv[1]
exp = {}
for line in open(out + "/calib_expected.txt"):
    p = line.split()
    if len(p) == 5 and p[1] == "read":
        exp[p[0]] = (int(p[2]), int(p[4]))
def kname(text):
    m = re.search(r"calib_gather1ILi(\d+)E", text)
    if m: return "calib_gather1<%s>" % m.group(1)
    m = re.search(r"calib_(read16|read4|read1|rows24|write16|write48|write4)", text)
    return m.group(0) if m else None
def counters(path, counter):
    res, dur = {}, {}
    for line in open(path):
        p = line.split()
        if len(p) >= 5 and p[-4] == counter:
            k = kname(line)
            if k: res[k] = float(p[-2]) * 1024.0
        elif len(p) >= 12 and "calib_" in line and p[-11].isdigit() and p[-4] != counter:
            k = kname(line)
            if k: dur[k] = float(p[-9])
    return res, dur
f, dur = counters(out + "/calib_FETCH_SIZE.txt", "FETCH_SIZE")
w, _ = counters(out + "/calib_WRITE_SIZE.txt", "WRITE_SIZE")
lines = ["%-22s %14s %14s %8s %14s %14s %8s %10s %9s" % ("kernel", "must_read_B", "FETCH_SIZE_B", "ratio", "must_write_B", "WRITE_SIZE_B", "ratio", "avg_us", "GB/s")]
for k, (r, wr) in exp.items():
    fs, ws, d = f.get(k, 0.0), w.get(k, 0.0), dur.get(k, 0.0)
    lines.append("%-22s %14d %14.0f %8.3f %14d %14.0f %8.3f %10.1f %9.1f" % (k, r, fs, fs / r if r else 0.0, wr, ws, ws / wr if wr else 0.0, d / 1e3, (r + wr) / d if d else 0.0))
open(out + "/pmc_calibration.txt", "w").write("\n".join(lines) + "\n")
print("\n".join(lines))
PY

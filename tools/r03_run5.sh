#!/bin/bash
cd "$GRAFT_REPO_ROOT"
out=gpurun_out/${1:-r03e}
mkdir -p $out
run() { # name extra-env...
  name=$1; shift
  env "$@" timeout 300 python bench.py --steps 40 --no-cpu-baseline > $out/bench_$name.json 2> $out/bench_$name.err
  python - "$out/bench_$name.json" $name <<'PY'
import json, sys
try:
    d = json.load(open(sys.argv[1]))
    s = d["config"]["stage_ms_serialized"]
    print("%-14s step %.4f  " % (sys.argv[2], d["ms_per_step"]) + " ".join("%s %.4f" % (k.replace("k_", ""), v) for k, v in s.items()))
except Exception as e:
    print(sys.argv[2], "FAILED", e)
PY
}
VX_F0_THREADS=384 timeout 300 python tools/fast0_debug.py 64 256 > $out/fast0_debug_384.txt 2>&1; tail -1 $out/fast0_debug_384.txt
run t256 VX_F0_THREADS=256
run t384 VX_F0_THREADS=384
run t384_wgs8 VX_F0_THREADS=384 VX_REG_WGS_PER_CU=8
run t384_wgs12 VX_F0_THREADS=384 VX_REG_WGS_PER_CU=12
run t384_wgs32 VX_F0_THREADS=384 VX_REG_WGS_PER_CU=32

#!/bin/bash
cd "$GRAFT_REPO_ROOT"
out=gpurun_out/${1:-r03h}
mkdir -p $out
export VX_POOL_VERTS=16000000 VX_POOL_INDICES=48000000
run() { # name lib extra-env...
  name=$1; lib=$2; shift; shift
  env VOXELS_HIP_LIBRARY=$lib "$@" timeout 300 python bench.py --steps 40 --no-cpu-baseline --serialize > $out/bench_$name.json 2> $out/bench_$name.err
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
run default ""
for v in coalesced novstore; do run $v "$GRAFT_REPO_ROOT/tools/ab/$v.so"; done

#!/bin/bash
cd "$GRAFT_REPO_ROOT"
out=gpurun_out/${1:-r03p}
mkdir -p $out
run() { # name lib env...
  name=$1; lib=$2; shift 2
  env "$@" VOXELS_HIP_LIBRARY=$lib timeout 300 python bench.py --steps 8 --warmup 2 --no-extra --no-cpu-baseline --serialize > $out/bench_$name.json 2> $out/bench_$name.err
  python - $out/bench_$name.json $name <<'PY'
import json, sys
try:
    d = json.load(open(sys.argv[1]))
    s = d["config"]["stage_ms_serialized"]
    print(sys.argv[2], "regular(L>=1)", s["k_regular"], "transition", s["k_transition"], "regular0", s["k_regular0"], "material", s["k_material"])
except Exception as e:
    print(sys.argv[2], "failed", e)
PY
}
B=voxels_amd/csrc/libvoxels_hip.so
run base $B A=1
run tw5 tools/ab/tw5.so A=1
run tw5_g1280 tools/ab/tw5.so VX_TR_GRID=1280
run tw5_g2560 tools/ab/tw5.so VX_TR_GRID=2560
run base_g1024 $B VX_TR_GRID=1024
run base_g2048 $B VX_TR_GRID=2048

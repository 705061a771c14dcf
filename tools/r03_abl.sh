#!/bin/bash
cd "$GRAFT_REPO_ROOT"
out=gpurun_out/${1:-r03n}
mkdir -p $out
for v in ${VARIANTS:-base tr1 tr2 tr3 tr4 tr5 f11 f12 f13 f14}; do
  lib=tools/ab/$v.so; [ $v = base ] && lib=voxels_amd/csrc/libvoxels_hip.so
  VOXELS_HIP_LIBRARY=$lib timeout 300 python bench.py --steps 8 --warmup 2 --no-extra --no-cpu-baseline --serialize > $out/bench_$v.json 2> $out/bench_$v.err
  python - $out/bench_$v.json $v <<'PY'
import json, sys
try:
    d = json.load(open(sys.argv[1]))
    s = d["config"]["stage_ms_serialized"]
    print(sys.argv[2], "regular(L>=1)", s["k_regular"], "transition", s["k_transition"], "regular0", s["k_regular0"])
except Exception as e:
    print(sys.argv[2], "failed", e)
PY
done

#!/bin/bash
cd "$GRAFT_REPO_ROOT"
source tools/r03_abl_fn.sh
out=gpurun_out/${1:-r03D}; mkdir -p $out
B=voxels_amd/csrc/libvoxels_hip.so
for v in base cabl; do
  lib=tools/ab/$v.so; [ $v = base ] && lib=$B
  VOXELS_HIP_LIBRARY=$lib timeout 300 python bench.py --steps 8 --warmup 2 --no-extra --no-cpu-baseline --serialize > $out/bench_$v.json 2> $out/bench_$v.err
  python - $out/bench_$v.json $v <<'PY'
import json, sys
try:
    d = json.load(open(sys.argv[1])); print(sys.argv[2], d["config"]["stage_ms_serialized"])
except Exception as e:
    print(sys.argv[2], "failed", e, open(sys.argv[1].replace(".json", ".err")).read()[-300:])
PY
done

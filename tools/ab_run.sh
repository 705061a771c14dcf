#!/bin/bash
# A/B or ablation timing of variant libraries (tools/ab_build.py) on the bench workload, one kernel at a time.
# Usage (GPU box): bash tools/ab_run.sh <outdir> <name>[:ENV=VALUE[:ENV=VALUE..]] ...
#   name = "base" (the product library) or a variant built into tools/ab/<name>.so; the part behind the first ':' is
#   exported for that run (e.g. base:VX_UPPER=0).  Prints the serialised stage times of every run.
cd "$GRAFT_REPO_ROOT"
out=$1; shift; mkdir -p "$out"
for spec in "$@"; do
  name=${spec%%:*}; envs=""; [ "$spec" != "$name" ] && envs=$(echo "${spec#*:}" | tr ':' ' ')
  lib=tools/ab/$name.so; [ "$name" = base ] && lib=voxels_amd/csrc/libvoxels_hip.so
  tag=$(echo "$spec" | tr ':=' '__')
  env $envs VOXELS_HIP_LIBRARY=$lib timeout 300 python bench.py --steps 8 --warmup 2 --no-extra --no-cpu-baseline --serialize > "$out/bench_$tag.json" 2> "$out/bench_$tag.err"
  python - "$out/bench_$tag.json" "$spec" <<'PY'
import json, sys
try:
    print(sys.argv[2], json.load(open(sys.argv[1]))["config"]["stage_ms_serialized"])
except Exception as e:
    print(sys.argv[2], "failed:", e)
PY
done

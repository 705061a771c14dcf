#!/bin/bash
# round 3, third GPU call: instruction price list, self-test candidates, A/B of kernel variants (stage times from bench.py), transition profile
cd "$GRAFT_REPO_ROOT"
out=gpurun_out/${1:-r03c}
mkdir -p $out
./tools/valu_rate > $out/valu_rate.txt 2>&1
python -c "from voxels_amd import Polygonizer; p = Polygonizer(); print('selftest', p.selftest().tolist())" > $out/selftest.txt 2>&1
timeout 600 python -m pytest tests -m gpu -q -x > $out/gputests.log 2>&1
echo "pytest rc $?" >> $out/gputests.log
for v in default oldprefetch exactnormals sepreserve shflscan; do
  lib=""
  if [ $v != default ]; then lib="$GRAFT_REPO_ROOT/tools/ab/$v.so"; fi
  VOXELS_HIP_LIBRARY=$lib timeout 300 python bench.py --steps 40 --no-cpu-baseline > $out/bench_$v.json 2> $out/bench_$v.err
  python - "$out/bench_$v.json" $v <<'PY'
import json, sys
try:
    d = json.load(open(sys.argv[1]))
    s = d["config"]["stage_ms_serialized"]
    print("%-14s step %.4f  " % (sys.argv[2], d["ms_per_step"]) + " ".join("%s %.4f" % (k.replace("k_", ""), v) for k, v in s.items()))
except Exception as e:
    print(sys.argv[2], "FAILED", e)
PY
done
VX_LIB=$GRAFT_REPO_ROOT/tools/ab/trprof.so timeout 300 python tools/stage_times.py 1024 4 3 > $out/trprof.txt 2>&1
python - > $out/ntcells.txt 2>&1 <<'PY'
import numpy as np
from voxels_amd import Polygonizer, synth
p = Polygonizer(); p.set_materials(synth.default_lut()); p.create_terrain(1024); p.execute(4)
for l in range(4):
    t = p.level(l, with_data=False).infos
    v, i = t["n_verts"], t["n_idx"]
    print("L%d listed blocks %d | verts mean %.0f p50 %d p90 %d p99 %d max %d | tris mean %.0f p90 %d max %d" % (l, len(t), v.mean(), np.percentile(v, 50), np.percentile(v, 90), np.percentile(v, 99), v.max(), (i / 3).mean(), np.percentile(i / 3, 90), (i / 3).max()))
    print("   verts histogram (bins of 128):", np.bincount(np.minimum(v // 128, 16)).tolist())
PY
cat $out/selftest.txt; tail -2 $out/gputests.log; cat $out/ntcells.txt; tail -30 $out/trprof.txt; cat $out/valu_rate.txt

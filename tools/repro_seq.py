import sys, os
sys.path.insert(0, "."); sys.path.insert(0, "tests")
import numpy as np
import torch
torch.cuda.init()
import vxo, fields
import test_gpu_parity as T
from voxels_amd import Polygonizer
poly = Polygonizer(device=0); poly.set_materials(vxo.default_lut())
port = vxo.load_port()
seq = [("carve", lambda: T.test_hip_carve_modify_matches_reference_fixture(poly, port)),
       ("edits64", lambda: T.test_hip_repeated_edits_vs_port(poly, port, 64)),
       ("edits256", lambda: T.test_hip_repeated_edits_vs_port(poly, port, 256)),
       ("config5", lambda: T.test_hip_config5_512_carve_incremental(poly, port)),
       ("dev64", lambda: T.test_hip_device_edits(poly, port, 64)),
       ("dev256", lambda: T.test_hip_device_edits(poly, port, 256)),
       ("compaction", lambda: T.test_hip_pool_compaction(poly, port)),
       ("pipeline", lambda: T.test_hip_device_only_pipeline_512(poly, port))]
only = sys.argv[1].split(",") if len(sys.argv) > 1 else None
for name, fn in seq:
    if only and name not in only: continue
    try:
        fn(); print(name, "ok", flush=True)
    except Exception as e:
        print(name, "FAILED:", str(e)[:300], flush=True)
        h = poly.debug_header(352)
        print("slots", h[:8], "cursors", h[32:35], "slow", h[224:228], "upper", h[256], "giveup", h[288:293], "l0head", h[320], "large", h[176:178], flush=True)
        break

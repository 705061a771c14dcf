import sys
sys.path.insert(0, ".")
import torch; torch.cuda.init()
from voxels_amd import Polygonizer, synth
p = Polygonizer(device=0); p.set_materials(synth.default_lut()); p.create_terrain(1024, 1337)
for first in (0, 1, 3, 4):
    for _ in range(4):
        info = p.execute_from(0, first)
    import time
    t = time.perf_counter()
    for _ in range(20): info = p.execute_from(0, first)
    dt = (time.perf_counter() - t) / 20 * 1e3
    print("1024^3, all 7 levels, meshes from level %d (honoured: %d): %.4f ms per call, device %.4f ms" % (first, info.first_meshed_level, dt, info.device_ms))

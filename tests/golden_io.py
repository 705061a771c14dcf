"""Load tests/golden/*.npz (written by tests/golden/make_golden.py from the unmodified reference)."""
import os

import numpy as np

import vxo

GOLDEN_DIR = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")


class Golden:
    def __init__(self, name):
        self.z = np.load(os.path.join(GOLDEN_DIR, name + ".npz"))
        self.dist, self.mat, self.blend = self.z["dist"], self.z["mat"], self.z["blend"]
        self.flags, self.stats = self.z["flags"], self.z["stats"]
        n = int(self.z["levels"][0])
        self.levels = [vxo.Level(self.z["L%d_infos" % i], self.z["L%d_verts" % i], self.z["L%d_idx" % i],
                                 self.z["L%d_tverts" % i], self.z["L%d_tidx" % i]) for i in range(n)]

    def __getitem__(self, k):
        return self.z[k]

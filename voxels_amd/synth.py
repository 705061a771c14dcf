"""ctypes binding of include/voxels_synth.h (libvoxels_synth.so): synthetic inputs for tests and bench.py."""
import ctypes as C
import os

import numpy as np

_PATH = os.path.join(os.path.dirname(os.path.abspath(__file__)), "csrc", "libvoxels_synth.so")
_lib = None


def _load():
    global _lib
    if _lib is None:
        if not os.path.exists(_PATH):
            raise RuntimeError("%s not found: run __graft_entry__.build()" % _PATH)
        _lib = C.CDLL(_PATH)
        _lib.vxs_terrain.argtypes = [C.c_uint32] * 4 + [C.c_void_p] * 3
        _lib.vxs_terrain_ex.argtypes = [C.c_uint32] * 5 + [C.c_void_p] * 3
        _lib.vxs_sphere.argtypes = [C.c_uint32] * 3 + [C.c_float, C.c_void_p]
        _lib.vxs_block_empty_flags.argtypes = [C.c_uint32, C.c_uint32, C.c_void_p, C.c_void_p]
    return _lib


def terrain(n, z0=0, z1=None, seed=1337, materials=True, style=0):
    """(dist int8, mat u8, blend u8) of planes [z0, z1) of the n^3 noise terrain, each shaped [z1-z0, n, n].
    style 1 = "caves": surface in a large share of all blocks (include/voxels_synth.h)."""
    z1 = n if z1 is None else z1
    lib = _load()
    d = np.zeros((z1 - z0, n, n), np.int8)
    m = np.zeros((z1 - z0, n, n), np.uint8) if materials else None
    b = np.zeros((z1 - z0, n, n), np.uint8) if materials else None
    lib.vxs_terrain_ex(n, z0, z1, seed, style, d.ctypes.data, m.ctypes.data if materials else None, b.ctypes.data if materials else None)
    return d, m, b


def sphere(n, z0=0, z1=None, r_frac=0.35):
    z1 = n if z1 is None else z1
    d = np.zeros((z1 - z0, n, n), np.int8)
    _load().vxs_sphere(n, z0, z1, r_frac, d.ctypes.data)
    return d


def block_empty_flags(dist):
    planes, n, _ = dist.shape
    assert planes % 16 == 0 and n % 16 == 0 and dist.dtype == np.int8 and dist.flags.c_contiguous
    fl = np.zeros((n // 16) ** 2 * (planes // 16), np.uint8)
    _load().vxs_block_empty_flags(n, planes, dist.ctypes.data, fl.ctypes.data)
    return fl


def default_lut():
    """MaterialMap of the synthetic workloads and fixtures: material m -> Ids0 = (6m, 6m+1, 6m+2), Ids1 = (6m+3..6m+5)
    (mod 251), the six texture ids of Voxels::MaterialMap::Material (include/MaterialMap.h)."""
    return (np.arange(256 * 6, dtype=np.uint32) % 251).astype(np.uint8).reshape(256, 6)

"""Exactness of the float-evaluated integer division used for edge interpolation (tv_core.h edge_t): for every pair
of distinct int8 samples it must equal the reference's truncating integer division (v1 << 8) / (v1 - v0)
(src/TransVoxelImpl.cpp:1591, :1674, :1942, :2025)."""
import ctypes as C

import numpy as np

from emu_lib import emu_library


def test_edge_t_matches_integer_division_for_all_int8_pairs():
    lib = emu_library().lib
    lib.emu_edge_t.restype = C.c_int
    lib.emu_edge_t.argtypes = [C.c_int, C.c_int]
    bad = 0
    for v0 in range(-128, 128):
        for v1 in range(-128, 128):
            if v0 == v1:
                continue
            a, b = v1 * 256, v1 - v0
            q = abs(a) // abs(b)
            if (a < 0) != (b < 0):
                q = -q
            bad += lib.emu_edge_t(v0, v1) != q
    assert bad == 0


def test_float_division_is_exact_vectorised():
    v0, v1 = np.meshgrid(np.arange(-128, 128, dtype=np.int32), np.arange(-128, 128, dtype=np.int32), indexing="ij")
    m = v0 != v1
    a, b = (v1 * 256)[m], (v1 - v0)[m]
    qi = (np.abs(a) // np.abs(b)) * np.sign(a) * np.sign(b)
    qf = np.trunc(a.astype(np.float32) / b.astype(np.float32)).astype(np.int32)
    assert np.array_equal(qi, qf)


def test_edge_end_classifies_like_edge_t_on_crossed_edges():
    """edge_end (no division) must agree with edge_t about "vertex on corner v1 / on corner v0 / inside" for every
    pair of samples an edge vertex can have: one negative, one non-negative (the case code puts vertices only there)."""
    lib = emu_library().lib
    for f in (lib.emu_edge_t, lib.emu_edge_end):
        f.restype = C.c_int
        f.argtypes = [C.c_int, C.c_int]
    bad = 0
    for v0 in range(-128, 128):
        for v1 in range(-128, 128):
            if (v0 < 0) == (v1 < 0):
                continue
            t, e = lib.emu_edge_t(v0, v1), lib.emu_edge_end(v0, v1)
            bad += (t == 0) != (e == 0) or (t == 256) != (e == 256) or ((t & 0xFF) == 0) != ((e & 0xFF) == 0)
    assert bad == 0

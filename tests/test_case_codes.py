"""case codes are not part of the public result, so they are pinned here directly — a CPU restatement of
the reference's two rules over the dense field against what the kernels looked up (libvoxels_hip_casedump.so: the
product sources compiled with -DVX_CASE_DUMP, which makes the passes record every code they use):
  * regular cells: Cell::CalcCaseCode, src/TransVoxelImpl.cpp:741-750 — bit i = sign of corner sample i, corners at
    base + ((i&1), (i>>1&1), (i>>2&1)) * 2^L clamped to the grid (:710-739, :1194-1201); trivial cells (0 / 255, :1560)
    are not looked up;
  * transition cells: the 9-bit code with weights {1,2,4,0x80,0x100,8,0x40,0x20,0x10} over the 3 x 3 half-resolution
    samples of the boundary plane (:1819, :1857-1921); 0 / 511 are skipped (:1923); faces without a neighbour block are
    not visited (:1829-1835); levels 0 and last have no transitions.
The restatement below follows those lines, not the kernels' bit-parallel formulation."""
import ctypes as C
import os

import numpy as np
import pytest

from golden_io import Golden

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
DUMP_SO = os.path.join(ROOT, "voxels_amd", "csrc", "libvoxels_hip_casedump.so")
TR_WEIGHTS = (0x01, 0x02, 0x04, 0x80, 0x100, 0x08, 0x40, 0x20, 0x10)


def sample(d, x, y, z):
    n = d.shape[0]
    return d[np.minimum(z, n - 1), np.minimum(y, n - 1), np.minimum(x, n - 1)]


def regular_codes(d, level, bx, by, bz):
    """CalcCaseCode of the 4096 cells of a block (x fastest), 0 for trivial cells."""
    mult = 1 << level
    c = np.arange(16)
    cz, cy, cx = np.meshgrid(c, c, c, indexing="ij")
    code = np.zeros((16, 16, 16), np.int32)
    for i in range(8):
        v = sample(d, (bx * 16 + cx + (i & 1)) * mult, (by * 16 + cy + ((i >> 1) & 1)) * mult, (bz * 16 + cz + (i >> 2)) * mult)
        code |= (v < 0).astype(np.int32) << i
    code[(code == 0) | (code == 255)] = 0
    return code.reshape(-1).astype(np.uint8)


def transition_codes(d, level, cnt, bx, by, bz):
    """9-bit codes of the 6 x 16 x 16 transition cells of a block (cell id = face * 256 + row * 16 + column), 0 where the
    cell is trivial or the face has no neighbour block."""
    mult, half = 1 << level, (1 << level) >> 1
    out = np.zeros((6, 16, 16), np.int32)
    origin = (bx * 16 * mult, by * 16 * mult, bz * 16 * mult)
    bc = (bx, by, bz)
    row, col = np.meshgrid(np.arange(16), np.arange(16), indexing="ij")
    for t in range(6):
        na = (2, 1, 0)[t % 3]                      # normal axis: z, y, x
        ca, ra = ((0, 1), (0, 2), (1, 2))[t % 3]   # column axis, row axis
        if (t < 3 and bc[na] == 0) or (t >= 3 and bc[na] + 1 >= cnt):
            continue
        code = np.zeros((16, 16), np.int32)
        for k in range(9):
            i, j = k % 3, k // 3
            p = [None, None, None]
            p[ca] = origin[ca] + col * mult + i * half
            p[ra] = origin[ra] + row * mult + j * half
            p[na] = np.full((16, 16), origin[na] + (16 * mult if t >= 3 else 0))
            code += (sample(d, p[0], p[1], p[2]) < 0).astype(np.int32) * TR_WEIGHTS[k]
        code[(code == 0) | (code == 511)] = 0
        out[t] = code
    return out.reshape(-1).astype(np.uint16)


def skipped_by_emptiness(flags, cnt, bx, by, bz):
    f = flags.reshape(cnt, cnt, cnt)
    r = lambda v: slice(max(v - 1, 0), min(v + 2, cnt))
    return bool(f[r(bz), r(by), r(bx)].all())


@pytest.mark.gpu
@pytest.mark.parametrize("name", ["noise64_fullrange_mat", "terrain128"])
def test_case_codes_match_the_reference_rules(name):
    from voxels_amd import build, synth
    from voxels_amd.binding import HipLibrary, Polygonizer
    build.build_hip_casedump()
    lib = HipLibrary(DUMP_SO)
    lib.lib.vx_debug_case_dump.argtypes = [C.c_void_p, C.c_uint32, C.c_uint32, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p]
    if name == "terrain128":
        d, m, b = synth.terrain(128, 0, 128, 9)
        flags = synth.block_empty_flags(d)
    else:
        g = Golden(name)
        d, m, b, flags = g.dist, g.mat, g.blend, g.flags
    n = d.shape[0]
    p = Polygonizer(library=lib)
    p.upload(d, m, b, flags)
    info = p.execute()
    ref_levels = int(np.log2(n // 16)) + 1
    assert info.levels == ref_levels
    checked_regular = checked_transition = 0
    for level in range(info.levels):
        cnt = (n // 16) >> level
        cap = cnt ** 3
        coords = np.zeros(cap, np.uint32)
        cases = np.zeros((cap, 4096), np.uint8)
        tr = np.zeros((cap, 1536), np.uint16)
        count = C.c_uint32()
        assert lib.lib.vx_debug_case_dump(p._h, level, cap, coords.ctypes.data, cases.ctypes.data, tr.ctypes.data, C.byref(count)) == 0
        assert count.value == info.active_blocks[level]
        for s in range(count.value):
            bx, by, bz = int(coords[s]) % cnt, (int(coords[s]) // cnt) % cnt, int(coords[s]) // (cnt * cnt)
            want = regular_codes(d, level, bx, by, bz)
            if level == 0 and skipped_by_emptiness(flags, cnt, bx, by, bz):
                want[:] = 0                       # the reference does not polygonize such a block (:1511-1527)
            assert np.array_equal(cases[s], want), (level, bx, by, bz)
            checked_regular += int((want != 0).sum())
            want_tr = transition_codes(d, level, cnt, bx, by, bz) if 0 < level < ref_levels - 1 else np.zeros(1536, np.uint16)
            assert np.array_equal(tr[s], want_tr), (level, bx, by, bz)
            checked_transition += int((want_tr != 0).sum())
    assert checked_regular > 1000 and checked_transition > 100

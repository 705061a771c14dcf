"""Seeded synthetic voxel fields used by the parity tests (numpy; small sizes).  Big grids for the bench
come from the product's own generator (vx_synth_terrain, voxels_amd/csrc/vx_synth.cpp)."""
import numpy as np


def smooth_noise(n, seed, scale=8, amp=6.0, octaves=2):
    """Band-limited random field (trilinear upsampling of coarse random lattices), float32, Z-up [z,y,x]."""
    rng = np.random.RandomState(seed)
    out = np.zeros((n, n, n), np.float32)
    a = amp
    s = scale
    for _ in range(octaves):
        m = max(2, n // s + 2)
        lat = rng.uniform(-1, 1, (m, m, m)).astype(np.float32)
        c = np.arange(n, dtype=np.float32) / np.float32(s)
        i0 = np.floor(c).astype(np.int32)
        f = (c - i0).astype(np.float32)
        f = f * f * (3 - 2 * f)

        def lerp_axis(v, axis):
            lo = np.take(v, i0, axis=axis)
            hi = np.take(v, i0 + 1, axis=axis)
            shape = [1, 1, 1]
            shape[axis] = n
            w = f.reshape(shape)
            return lo + (hi - lo) * w

        v = lerp_axis(lerp_axis(lerp_axis(lat, 0), 1), 2)
        out += a * v
        a *= 0.5
        s = max(2, s // 2)
    return out.astype(np.float32)


def terrain_field(n, seed, caves=True):
    """Height-field terrain (+ optional 3-D noise for overhangs): d = z - h(x,y) - cave(x,y,z), clamped +-100."""
    rng = np.random.RandomState(seed)
    base = smooth_noise(n, seed, scale=max(4, n // 4), amp=0.25 * n, octaves=3)[0]  # 2-D slice as heightmap
    h = np.float32(n / 2) + base
    z = np.arange(n, dtype=np.float32).reshape(n, 1, 1)
    d = z - h.reshape(1, n, n)
    if caves:
        d = d - smooth_noise(n, seed + 17, scale=max(4, n // 8), amp=5.0, octaves=2)
    return np.clip(d, -100, 100).astype(np.float32)


def materials_for(n, seed, nmat=3):
    """Material ids by dithered height band + smooth blend; exercises M0 != M1 and reuse-by-material failures."""
    rng = np.random.RandomState(seed + 101)
    z = np.arange(n, dtype=np.float32).reshape(n, 1, 1)
    jitter = rng.uniform(-4, 4, (n, n, n)).astype(np.float32)
    band = np.clip(((z + jitter) * nmat / n), 0, nmat - 1e-3)
    mat = band.astype(np.uint8)
    blend = (255 * (band - np.floor(band))).astype(np.uint8)
    return np.ascontiguousarray(mat), np.ascontiguousarray(blend)


def quantize_full_range(f, scale=12.0):
    """int8 distances using the whole range (bypasses the +-4 clamp of Grid::Create), so t takes many values."""
    q = np.clip(np.round(f * scale), -127, 127)
    return q.astype(np.int8)


def surface_equal(a_levels, b_levels, nrm_tol=0.0):
    """Compare two lists of vxo.Level. Returns (ok, message)."""
    if len(a_levels) != len(b_levels):
        return False, "level count %d vs %d" % (len(a_levels), len(b_levels))
    for li, (a, b) in enumerate(zip(a_levels, b_levels)):
        if a.totals() != b.totals():
            return False, "L%d totals %s vs %s" % (li, a.totals(), b.totals())
        for name in a.infos.dtype.names:
            if not np.array_equal(a.infos[name], b.infos[name]):
                return False, "L%d block info field %s differs" % (li, name)
        if not np.array_equal(a.idx, b.idx):
            return False, "L%d indices differ" % li
        if not np.array_equal(a.tidx, b.tidx):
            return False, "L%d transition indices differ" % li
        for kind, va, vb in (("verts", a.verts, b.verts), ("tverts", a.tverts, b.tverts)):
            for fld in ("pos", "sec", "tex"):
                xa = va[fld].view(np.uint32) if fld != "tex" else va[fld]
                xb = vb[fld].view(np.uint32) if fld != "tex" else vb[fld]
                if not np.array_equal(xa, xb):
                    bad = np.argwhere(xa != xb)[0]
                    return False, "L%d %s.%s differs at %s: %s vs %s" % (li, kind, fld, bad, va[fld][bad[0]], vb[fld][bad[0]])
            if nrm_tol == 0.0:
                if not np.array_equal(va["nrm"].view(np.uint32), vb["nrm"].view(np.uint32)):
                    bad = np.argwhere(va["nrm"].view(np.uint32) != vb["nrm"].view(np.uint32))[0]
                    return False, "L%d %s.nrm differs bitwise at %s: %s vs %s" % (li, kind, bad, va["nrm"][bad[0]], vb["nrm"][bad[0]])
            elif len(va):
                err = np.abs(va["nrm"] - vb["nrm"]).max()
                if err > nrm_tol:
                    return False, "L%d %s.nrm max err %g" % (li, kind, err)
    return True, "ok"


def edited_blocks(pre, post):
    """ids + data of the 16^3 blocks that differ between two dense grids (what Grid edits hand to the device)."""
    n = pre[0].shape[0]
    nb = n // 16
    diff = np.zeros((n, n, n), bool)
    for a, b in zip(pre, post):
        diff |= a != b
    blk = diff.reshape(nb, 16, nb, 16, nb, 16).any(axis=(1, 3, 5))
    ids = np.flatnonzero(blk.ravel()).astype(np.uint32)
    out = []
    for arr in post:
        v = arr.reshape(nb, 16, nb, 16, nb, 16).transpose(0, 2, 4, 1, 3, 5).reshape(nb ** 3, 4096)
        out.append(np.ascontiguousarray(v[ids]))
    return ids, out




def check_halo_exchange_group(make_poly, torch, device, n, levels, world, axis, reference_levels, seed=7, nrm_tol=0.0):
    """`world` contexts of one process, each with ONLY its own slab filled (halo layers and the neighbours' flag layers
    zero): vx_halo_exchange_group must bring in exactly what the path reads beyond the slab, i.e. the union of the
    ranks' results equals the surface of the whole grid.  Shared by the CPU emulation and the GPU test."""
    from voxels_amd import synth
    from voxels_amd.binding import Polygonizer
    from voxels_amd.slab import SlabBuffers, merge_rank_levels
    d, m, b = synth.terrain(n, 0, n, seed)
    flags = synth.block_empty_flags(d)
    polys, slabs = [], []
    per = flags.size // world
    for r in range(world):
        slab = SlabBuffers(torch, n, r, world, device, axis=axis)
        sl = slice(slab.z0, slab.z1)
        if axis == "z":
            slab.fill_own(np.ascontiguousarray(d[sl]), np.ascontiguousarray(m[sl]), np.ascontiguousarray(b[sl]), flags[r * per:(r + 1) * per])
        else:
            own = np.zeros_like(flags).reshape(n // 16, n // 16, n // 16)  # [bz][by][bx]: only this rank's block rows
            own[:, slab.z0 // 16:slab.z1 // 16] = flags.reshape(own.shape)[:, slab.z0 // 16:slab.z1 // 16]
            slab.fill_own(np.ascontiguousarray(d[:, sl]), np.ascontiguousarray(m[:, sl]), np.ascontiguousarray(b[:, sl]), own.reshape(-1))
        p = make_poly()
        slab.attach(p)
        polys.append(p)
        slabs.append(slab)
    if hasattr(torch, "cuda") and device.type == "cuda":
        torch.cuda.synchronize()
    # twice: the second exchange finds the library's mirrors of the fields current and has to keep them so (the received
    # rows are written into them by the unpack kernel), as in a run that exchanges before every step
    for round_ in range(2):
        Polygonizer.halo_exchange_group(polys)
        parts = []
        for p in polys:
            p.execute(levels)
            parts.append(p.all_levels())
        ok, msg = surface_equal(merge_rank_levels(parts), reference_levels[:levels], nrm_tol=nrm_tol)
        assert ok, "round %d: %s" % (round_, msg)


def check_device_terrain(make_poly, torch, device, n, seed=1337, world=2):
    """§8(f) row 3: the synthetic terrain generated on the device equals the host generator byte for byte — the whole
    grid (compared through the grid file: every voxel of all three fields + the codec's flags) and slabs with halo."""
    from voxels_amd import synth
    from voxels_amd.slab import SlabBuffers
    d, m, b = synth.terrain(n, 0, n, seed)
    flags = synth.block_empty_flags(d)
    host = make_poly()
    host.upload(d, m, b, flags)
    dev = make_poly()
    dev.create_terrain(n, seed)
    assert np.array_equal(host.pack(), dev.pack())
    for axis in ("z", "y"):
        for r in range(world):
            want = SlabBuffers(torch, n, r, world, device, axis=axis)
            want.fill_from_full(d, m, b, flags)
            got = SlabBuffers(torch, n, r, world, device, axis=axis)
            p = make_poly()
            got.attach(p)
            p.fill_terrain(seed)
            assert torch.equal(got.dist, want.dist) and torch.equal(got.mat, want.mat) and torch.equal(got.blend, want.blend), (axis, r)
            fl_got, fl_want = got.flags.cpu().numpy().reshape(n // 16, n // 16, n // 16), flags.reshape(n // 16, n // 16, n // 16)
            own = slice(got.z0 // 16, got.z1 // 16)
            if axis == "z":
                assert np.array_equal(fl_got[own], fl_want[own]), (axis, r)
            else:
                assert np.array_equal(fl_got[:, own], fl_want[:, own]), (axis, r)


def check_host_meshes(poly, hm):
    """The one-step host copy (vx_host_meshes_acquire) holds exactly what vx_download_level copies out block by block."""
    for l, lev in enumerate(poly.all_levels()):
        r = poly.level_ranges(l)
        ov = oi = otv = oti = 0
        for k, info in enumerate(lev.infos):
            nv, ni = int(info["n_verts"]), int(info["n_idx"])
            assert np.array_equal(hm.verts[r["v_off"][k]:r["v_off"][k] + nv], lev.verts[ov:ov + nv]), (l, k)
            assert np.array_equal(hm.indices[r["i_off"][k]:r["i_off"][k] + ni], lev.idx[oi:oi + ni]), (l, k)
            ov += nv; oi += ni
            for f in range(6):
                nv, ni = int(info["n_tverts"][f]), int(info["n_tidx"][f])
                assert np.array_equal(hm.verts[r["tv_off"][k][f]:r["tv_off"][k][f] + nv], lev.tverts[otv:otv + nv]), (l, k, f)
                assert np.array_equal(hm.indices[r["ti_off"][k][f]:r["ti_off"][k][f] + ni], lev.tidx[oti:oti + ni]), (l, k, f)
                otv += nv; oti += ni


def listed_blocks_equal_by_id(part, full):
    """Every block `part` (a Level) lists is listed by `full` under the same id with the same bytes (regular mesh, transition
    meshes, corners, counts).  Returns (ok, message).  For surfaces that hold a subset of a level's blocks - the levels a
    partial run (vx_polygonize_from) did not mesh list only what later Modifications rebuilt."""
    def starts(counts):
        return np.concatenate([[0], np.cumsum(counts.astype(np.int64))]).astype(np.int64)
    where = {int(i): k for k, i in enumerate(full.infos["id"])}
    fv, fi = starts(full.infos["n_verts"]), starts(full.infos["n_idx"])
    ftv, fti = starts(full.infos["n_tverts"].sum(axis=1)), starts(full.infos["n_tidx"].sum(axis=1))
    pv, pi = starts(part.infos["n_verts"]), starts(part.infos["n_idx"])
    ptv, pti = starts(part.infos["n_tverts"].sum(axis=1)), starts(part.infos["n_tidx"].sum(axis=1))
    for k, bid in enumerate(part.infos["id"]):
        j = where.get(int(bid))
        if j is None:
            return False, "block id %d is not in the full surface" % bid
        if part.infos[k] != full.infos[j]:
            return False, "block %d: infos differ" % bid
        for name, a, ao, b, bo in (("verts", part.verts, pv, full.verts, fv), ("idx", part.idx, pi, full.idx, fi),
                                   ("tverts", part.tverts, ptv, full.tverts, ftv), ("tidx", part.tidx, pti, full.tidx, fti)):
            if not np.array_equal(a[ao[k]:ao[k + 1]], b[bo[j]:bo[j + 1]]):
                return False, "block %d: %s differ" % (bid, name)
    return True, "ok"


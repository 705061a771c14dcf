"""voxels_amd/digest.py (the correctness bit of `bench.py --gpus N`): the digest of a surface must not depend on how its
blocks are spread over ranks or ordered, and must change when any byte of any mesh, count or id changes.  CPU only: the
surface comes from the emulation backend."""
import numpy as np

import vxo
from emu_lib import emu_library
from voxels_amd import digest, synth
from voxels_amd.binding import Level, Polygonizer


def split_level(lv, pick):
    """the blocks of `lv` whose position satisfies pick(k), as a Level of their own"""
    ov = np.concatenate([[0], np.cumsum(lv.infos["n_verts"])]).astype(np.int64)
    oi = np.concatenate([[0], np.cumsum(lv.infos["n_idx"])]).astype(np.int64)
    otv = np.concatenate([[0], np.cumsum(lv.infos["n_tverts"].sum(axis=1))]).astype(np.int64)
    oti = np.concatenate([[0], np.cumsum(lv.infos["n_tidx"].sum(axis=1))]).astype(np.int64)
    ks = [k for k in range(len(lv.infos)) if pick(k)]
    cat = lambda arr, off: np.concatenate([arr[off[k]:off[k + 1]] for k in ks]) if ks else arr[:0]
    return Level(lv.infos[ks], cat(lv.verts, ov), cat(lv.idx, oi), cat(lv.tverts, otv), cat(lv.tidx, oti))


def surface(n=64, seed=5, levels=3):
    d, m, b = synth.terrain(n, seed=seed)
    p = Polygonizer(library=emu_library())
    p.set_materials(vxo.default_lut())
    p.upload(d, m, b, synth.block_empty_flags(d))
    p.execute(levels)
    lv = p.all_levels()
    p.close()
    return lv


def test_digest_adds_up_over_any_partition_and_order():
    whole = surface()
    ref = digest.surface_digest(whole)
    assert ref[0][:, 1].sum() > 0 and ref[0][1:, 3].sum() > 0, "the test surface needs regular and transition meshes"
    for parts in (2, 3):
        ranks = [[split_level(lv, lambda k, r=r: k % parts == r) for lv in whole] for r in range(parts)]
        ds = [digest.surface_digest(r) for r in ranks]
        assert digest.digests_equal(digest.combine(ds), ref)
        # the form bench.py all-reduces: element-wise sums of packed digests
        assert digest.digests_equal(digest.unpack(sum(digest.pack(d) for d in ds), len(whole)), ref)
    # block order inside a rank does not matter either
    rev = [split_level(lv, lambda k: True) for lv in whole]
    for lv in rev:
        order = list(range(len(lv.infos)))[::-1]
        parts = [split_level(lv, lambda k, q=q: k == q) for q in order]
        lv.infos = np.concatenate([p.infos for p in parts]) if parts else lv.infos
        lv.verts = np.concatenate([p.verts for p in parts]) if parts else lv.verts
        lv.idx = np.concatenate([p.idx for p in parts]) if parts else lv.idx
        lv.tverts = np.concatenate([p.tverts for p in parts]) if parts else lv.tverts
        lv.tidx = np.concatenate([p.tidx for p in parts]) if parts else lv.tidx
    assert digest.digests_equal(digest.surface_digest(rev), ref)


def test_digest_sees_every_kind_of_change():
    whole = surface()
    ref = digest.surface_digest(whole)

    def changed(mutate):
        lv = [Level(l.infos.copy(), l.verts.copy(), l.idx.copy(), l.tverts.copy(), l.tidx.copy()) for l in whole]
        mutate(lv)
        return not digest.digests_equal(digest.surface_digest(lv), ref)

    def swap_two_indices(lv):
        i = lv[0].idx
        j = np.flatnonzero(i[:-1] != i[1:])[0]
        i[j], i[j + 1] = i[j + 1], i[j]

    def nudge_position(lv):
        lv[1].verts["pos"][3, 1] = np.nextafter(lv[1].verts["pos"][3, 1], np.float32(1e9))

    def nudge_normal(lv):
        lv[0].verts["nrm"][7, 0] = np.nextafter(lv[0].verts["nrm"][7, 0], np.float32(2))

    def texture(lv):
        lv[0].verts["tex"][0, 5] ^= 1

    def transition_index(lv):
        lv[1].tidx[0] ^= 1

    def block_id(lv):
        lv[2].infos["id"][0] += 1

    def move_vertex_between_blocks(lv):  # same streams, different split
        lv[0].infos["n_verts"][0] -= 1
        lv[0].infos["n_verts"][1] += 1

    def face_split(lv):
        k = np.flatnonzero(lv[1].infos["n_tverts"].sum(axis=1) > 0)[0]
        f = np.flatnonzero(lv[1].infos["n_tverts"][k] > 0)[0]
        lv[1].infos["n_tverts"][k, f] -= 1
        lv[1].infos["n_tverts"][k, (f + 1) % 6] += 1

    def drop_level(lv):
        lv.pop()

    for m in (swap_two_indices, nudge_position, nudge_normal, texture, transition_index, block_id, move_vertex_between_blocks, face_split, drop_level):
        assert changed(m), m.__name__
    # without normals the digest ignores them (comparisons across implementations that only promise 1e-5 there)
    lv = [Level(l.infos.copy(), l.verts.copy(), l.idx.copy(), l.tverts.copy(), l.tidx.copy()) for l in whole]
    nudge_normal(lv)
    assert digest.digests_equal(digest.surface_digest(lv, with_normals=False), digest.surface_digest(whole, with_normals=False))

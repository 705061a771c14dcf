"""Order-independent digest of a polygonized surface: what `bench.py --gpus N` all-reduces over the ranks and compares with
a whole-grid run on rank 0, so that a multi-GPU bench line carries a correctness bit.

The result of Polygonizer::Execute is a set of blocks per LOD level (PolygonSurface::GetBlockForLevel,
include/Polygonizer.h:180-189); a sharded run produces the same set spread over the ranks.  Every block contributes one
64-bit value that depends on its level, its id (ids number the blocks of the whole grid, so they do not depend on who
owns the block), its corners, its counts and every byte of its 1 + 6 meshes in order; the digest is the sum of these
values modulo 2^64 plus the per-level totals, both of which add up over ranks."""
import numpy as np

_M1 = np.uint64(0x9E3779B97F4A7C15)
_M2 = np.uint64(0xBF58476D1CE4E5B9)
_M3 = np.uint64(0x94D049BB133111EB)


def _mix(x):
    """splitmix64 finaliser, vectorised (uint64 arithmetic wraps)."""
    x = (x ^ (x >> np.uint64(30))) * _M2
    x = (x ^ (x >> np.uint64(27))) * _M3
    return x ^ (x >> np.uint64(31))


def _stream_sums(words, counts, words_per_elem):
    """Per block: sum over its words of mix(word + position-in-block * M1).  `words`: uint64 array of all blocks'
    elements in block order; `counts`: elements per block."""
    nblocks = len(counts)
    out = np.zeros(nblocks, np.uint64)
    if words.size == 0:
        return out
    per_block = counts.astype(np.int64) * words_per_elem
    ends = np.cumsum(per_block)
    starts = ends - per_block
    pos = np.arange(words.size, dtype=np.uint64) - np.repeat(starts.astype(np.uint64), per_block)
    h = _mix(words + (pos + np.uint64(1)) * _M1)
    nz = per_block > 0
    # reduceat over the non-empty blocks (their starts are strictly increasing)
    sums = np.add.reduceat(h, starts[nz])
    out[nz] = sums
    return out


def _vertex_words(verts, with_normals):
    raw = np.ascontiguousarray(verts).view(np.uint64).reshape(-1, 6)
    if with_normals:
        return raw.reshape(-1), 6
    # bytes 0..27: position + secondary position, 28..39: normal, 40..47: texture ids
    b = np.ascontiguousarray(verts).view(np.uint8).reshape(-1, 48)
    keep = np.concatenate([b[:, :28], np.zeros((b.shape[0], 4), np.uint8), b[:, 40:]], axis=1)
    return np.ascontiguousarray(keep).view(np.uint64).reshape(-1), 5


def surface_digest(levels, with_normals=True, first_level=0):
    """levels: list of Level (voxels_amd.binding.Level or the oracle's), the first of which is LOD level `first_level` (a block's
    value depends on its level).  Returns (totals int64[len(levels), 5], hash uint64): totals = blocks, vertices, indices,
    transition vertices, transition indices per level."""
    totals = np.zeros((len(levels), 5), np.int64)
    total_hash = np.uint64(0)
    with np.errstate(over="ignore"):
        for li, lv in enumerate(levels):
            inf = lv.infos
            totals[li] = (len(inf), len(lv.verts), len(lv.idx), len(lv.tverts), len(lv.tidx))
            if len(inf) == 0:
                continue
            key = _mix(inf["id"].astype(np.uint64) * _M1 + np.uint64(first_level + li + 1))
            h = key.copy()
            vw, per = _vertex_words(lv.verts, with_normals)
            h += _mix(_stream_sums(vw, inf["n_verts"], per) + np.uint64(11)) * np.uint64(3)
            h += _mix(_stream_sums(lv.idx.astype(np.uint64), inf["n_idx"], 1) + np.uint64(13)) * np.uint64(5)
            tvw, per = _vertex_words(lv.tverts, with_normals)
            h += _mix(_stream_sums(tvw, inf["n_tverts"].sum(axis=1), per) + np.uint64(17)) * np.uint64(7)
            h += _mix(_stream_sums(lv.tidx.astype(np.uint64), inf["n_tidx"].sum(axis=1), 1) + np.uint64(19)) * np.uint64(9)
            # the per-face split of the transition meshes and the corners
            meta = np.concatenate([inf["n_tverts"].astype(np.uint64), inf["n_tidx"].astype(np.uint64),
                                   inf["min_corner"].view(np.uint32).astype(np.uint64), inf["max_corner"].view(np.uint32).astype(np.uint64)], axis=1)
            w = (np.arange(meta.shape[1], dtype=np.uint64) + np.uint64(1)) * _M1
            h += _mix((meta * w).sum(axis=1, dtype=np.uint64) + np.uint64(23)) * np.uint64(11)
            total_hash += (_mix(h ^ key) * key).sum(dtype=np.uint64)
    return totals, np.uint64(total_hash)


def combine(parts):
    """Sum of the digests of the ranks of a sharded run."""
    totals = sum(p[0] for p in parts)
    with np.errstate(over="ignore"):
        h = np.uint64(0)
        for p in parts:
            h = h + np.uint64(p[1])
    return totals, np.uint64(h)


def join(fine, coarse):
    """The digest of the levels [0, k) and the digest of the levels [k, ...) (surface_digest(..., first_level=k)) as the digest
    of all of them: what a sharded run's slabs and the rank that takes the coarse levels produce together."""
    with np.errstate(over="ignore"):
        return np.concatenate([fine[0], coarse[0]]), np.uint64(np.uint64(fine[1]) + np.uint64(coarse[1]))


def digests_equal(a, b):
    return np.array_equal(a[0], b[0]) and int(a[1]) == int(b[1])


def pack(d):
    """A digest as an int64 vector that can be summed element-wise over ranks (all_reduce SUM): the totals, then the hash
    in four 16-bit pieces (their sums stay far below 2^63 for any realistic world size)."""
    h = int(d[1])
    return np.concatenate([d[0].reshape(-1), np.array([(h >> (16 * k)) & 0xFFFF for k in range(4)], np.int64)]).astype(np.int64)


def unpack(vec, levels):
    """The inverse of pack() applied to an element-wise sum of packed digests."""
    vec = np.asarray(vec, np.int64)
    h = sum(int(vec[-4 + k]) << (16 * k) for k in range(4)) & 0xFFFFFFFFFFFFFFFF
    return vec[:-4].reshape(levels, 5).copy(), np.uint64(h)

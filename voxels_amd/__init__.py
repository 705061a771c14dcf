"""voxels_amd — MI355X-native TransVoxel polygonizer (drop-in for stoyannk/voxels' Polygonizer::Execute path).

The compute path is hand-written HIP for gfx950 in voxels_amd/csrc (libvoxels_hip.so, C ABI in
include/voxels_hip.h).  This package is the thin Python host side used by tests and bench.py: it mirrors
the reference's operator interface for the path (Grid -> Polygonizer.Execute -> PolygonSurface levels/blocks)
and fails loudly when the HIP library is missing — there is no CPU fallback.
"""
from .binding import (BLOCK_INFO_DTYPE, VERTEX_DTYPE, HipLibrary, Level, Polygonizer, VoxelsHipError,
                      hip_library_path)

__all__ = ["BLOCK_INFO_DTYPE", "VERTEX_DTYPE", "HipLibrary", "Level", "Polygonizer", "VoxelsHipError",
           "hip_library_path"]

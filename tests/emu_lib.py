"""Test-only access to the CPU emulation of the device pipeline (tests/emu/libvoxels_emu.so).  The product never
loads this library; tests pass it explicitly to the binding to exercise the shared core logic and host code
without a GPU."""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def emu_library():
    from voxels_amd import build
    from voxels_amd.binding import HipLibrary
    path = build.build_emu()
    lib = HipLibrary(path)
    assert lib.backend.startswith("emu:")
    return lib

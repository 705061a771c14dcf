"""The C-ABI library loads on a machine without a GPU and exports every symbol include/voxels_hip.h declares
(no compute calls here).  Also checks the struct sizes the header promises."""
import ctypes as C
import os
import re

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.fixture(scope="module")
def hip_so():
    from voxels_amd import build
    return build.build_hip()


def declared_functions(header):
    text = open(header).read()
    text = re.sub(r"/\*.*?\*/", "", text, flags=re.S)
    return sorted(set(re.findall(r"\b(vx[s]?_[a-z_0-9]+)\s*\(", text)))


def test_hip_library_exports_every_declared_symbol(hip_so):
    lib = C.CDLL(hip_so)
    names = declared_functions(os.path.join(ROOT, "include", "voxels_hip.h"))
    assert len(names) >= 15
    for n in names:
        assert hasattr(lib, n), "libvoxels_hip.so does not export " + n
    lib.vx_backend.restype = C.c_char_p
    assert lib.vx_backend() == b"hip:gfx950"


def test_synth_library_exports(hip_so):
    from voxels_amd import build
    lib = C.CDLL(build.build_synth())
    for n in declared_functions(os.path.join(ROOT, "include", "voxels_synth.h")):
        assert hasattr(lib, n), n


def test_context_creation_fails_loudly_without_gpu(hip_so):
    """No CPU fallback: without a HIP device vx_ctx_create must return an error, not a working context."""
    import torch
    if torch.cuda.is_available():
        pytest.skip("a GPU is present")
    lib = C.CDLL(hip_so)
    h = C.c_void_p()
    rc = lib.vx_ctx_create(0, C.byref(h))
    assert rc != 0 and not h.value


def test_binding_struct_sizes():
    from voxels_amd.binding import BLOCK_INFO_DTYPE, VERTEX_DTYPE
    assert VERTEX_DTYPE.itemsize == 48 and BLOCK_INFO_DTYPE.itemsize == 84


def test_product_package_never_references_the_oracle_or_emulation():
    """The product path must not import, link or load anything under oracle/ or tests/."""
    bad = []
    for dirpath, _, files in os.walk(os.path.join(ROOT, "voxels_amd")):
        for f in files:
            if f.endswith((".py", ".h", ".hip", ".cpp", ".inl")):
                text = open(os.path.join(dirpath, f), errors="ignore").read()
                for needle in ("libvoxels_port", "libvoxels_ref", "libvoxels_emu", "import vxo", "oracle/"):
                    # build.py may BUILD the checkers (oracle, emulation, reference drop-in binary); nothing may use them
                    if needle in text and f != "build.py":
                        bad.append((f, needle))
    assert not bad, bad

"""Drop-in test of the C++ API: tests/cpp/dropin_test.cpp, written only against the reference's public API, built
against (a) the reference's headers + the unmodified reference library and (b) this repo's include/ + libVoxels.so.
Both binaries must write byte-identical dumps (all levels, blocks, vertices, indices, statistics, the modification's
block ids, block accessors, the packed grid).  (a) is built where /root/reference exists and travels in oracle/_ref."""
import os
import subprocess

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
OURS = os.path.join(ROOT, "tests", "cpp", "dropin_ours")
REF = os.path.join(ROOT, "oracle", "_ref", "dropin_ref")


def test_public_headers_compile_standalone(tmp_path):
    """Every reference header name resolves and the API surface compiles (no GPU needed)."""
    src = tmp_path / "t.cpp"
    src.write_text('#include "Declarations.h"\n#include "Structs.h"\n#include "Version.h"\n#include "Library.h"\n'
                   '#include "VoxelSurface.h"\n#include "MaterialMap.h"\n#include "Grid.h"\n#include "Polygonizer.h"\n'
                   '#include <Voxels.h>\nstatic_assert(sizeof(Voxels::PolygonVertex) == 48, "");\nint main() { return VOXELS_VERSION == 0x00050001 ? 0 : 1; }\n')
    subprocess.check_call(["g++", "-std=c++14", "-fsyntax-only", "-I" + os.path.join(ROOT, "include"), str(src)])


@pytest.mark.gpu
@pytest.mark.parametrize("n,variant", [(64, ""), (128, ""), (64, "B")])
def test_same_application_same_bytes(tmp_path, n, variant):
    """variant "B": the Modification runs on a second Polygonizer after the first was reused on another grid - a surface
    carries its caches (src/TransVoxelImpl.h:81-95), whoever updates it."""
    from voxels_amd import build
    build.build_cpp_api()
    build.build_dropin_tests()
    if not os.path.exists(REF):
        pytest.skip("oracle/_ref/dropin_ref not built (needs /root/reference at build time)")
    a, b = str(tmp_path / "ours.bin"), str(tmp_path / "ref.bin")
    extra = [variant] if variant else []
    subprocess.check_call([OURS, str(n), a] + extra)
    subprocess.check_call([REF, str(n), b] + extra)
    da, db = open(a, "rb").read(), open(b, "rb").read()
    assert len(da) == len(db), (len(da), len(db))
    if da != db:
        first = next(i for i in range(len(da)) if da[i] != db[i])
        raise AssertionError("dumps differ at byte %d of %d" % (first, len(da)))


@pytest.mark.gpu
@pytest.mark.parametrize("n,devices,variant", [(64, 2, ""), (64, 4, "B"), (128, 2, ""), (128, 4, ""), (128, 8, "B")])
def test_same_application_same_bytes_on_several_devices(tmp_path, n, devices, variant):
    """The same application, unchanged, with VOXELS_DEVICES=N: Polygonizer::Execute cuts the grid into N slabs of rows, one
    helper context per slab (on a box with one GPU they share it), the finer levels come from the helpers, the levels coarser
    than a slab and the statistics from the primary context, the Modification continues on the primary.  The dump must still
    be the reference binary's, byte for byte."""
    from voxels_amd import build
    build.build_cpp_api()
    build.build_dropin_tests()
    if not os.path.exists(REF):
        pytest.skip("oracle/_ref/dropin_ref not built (needs /root/reference at build time)")
    a, b = str(tmp_path / "ours.bin"), str(tmp_path / "ref.bin")
    extra = [variant] if variant else []
    subprocess.check_call([OURS, str(n), a] + extra, env=dict(os.environ, VOXELS_DEVICES=str(devices)), timeout=300)
    subprocess.check_call([REF, str(n), b] + extra, timeout=300)
    da, db = open(a, "rb").read(), open(b, "rb").read()
    assert len(da) == len(db), (len(da), len(db))
    if da != db:
        first = next(i for i in range(len(da)) if da[i] != db[i])
        raise AssertionError("dumps differ at byte %d of %d" % (first, len(da)))


@pytest.mark.gpu
@pytest.mark.parametrize("n,devices,variant", [(128, 2, ""), (128, 4, "B"), (64, 2, "")])
def test_several_devices_split_the_levels(tmp_path, n, devices, variant):
    """A gentler surface (DROPIN_GENTLE: no block beyond the first capacity class), where the primary context of a multi-device
    Execute really leaves the finer levels to the helpers (vx_polygonize_from: VOXELS_TRACE says from which level it meshed):
    the surface is then helper-made blocks below that level and the primary's above, a Modification drops the helper-made
    blocks inside its box and appends the primary's - and the dump is still the reference binary's, byte for byte."""
    from voxels_amd import build
    build.build_cpp_api()
    build.build_dropin_tests()
    if not os.path.exists(REF):
        pytest.skip("oracle/_ref/dropin_ref not built (needs /root/reference at build time)")
    a, b = str(tmp_path / "ours.bin"), str(tmp_path / "ref.bin")
    extra = [variant] if variant else []
    env = dict(os.environ, DROPIN_GENTLE="1")
    out = subprocess.run([OURS, str(n), a] + extra, env=dict(env, VOXELS_DEVICES=str(devices), VOXELS_TRACE="1"), timeout=300, capture_output=True, text=True)
    assert out.returncode == 0, out.stderr[-500:]
    meshed = [int(l.split("meshed from level")[1].split()[0]) for l in out.stderr.splitlines() if "meshed from level" in l]
    assert meshed and min(meshed) >= 1, "the primary meshed every level itself: %s" % out.stderr[-800:]
    subprocess.check_call([REF, str(n), b] + extra, env=env, timeout=300)
    da, db = open(a, "rb").read(), open(b, "rb").read()
    assert len(da) == len(db), (len(da), len(db))
    if da != db:
        first = next(i for i in range(len(da)) if da[i] != db[i])
        raise AssertionError("dumps differ at byte %d of %d" % (first, len(da)))


@pytest.mark.gpu
@pytest.mark.parametrize("n,devices,gentle", [(64, 1, False), (128, 1, True), (128, 2, True), (128, 4, False)])
def test_a_surface_edited_again_and_again(tmp_path, n, devices, gentle):
    """DROPIN_EDITS=4: four more Modifications of the same surface behind the first - overlapping boxes, adding and carving in
    turn, the two Polygonizers in turn.  On several devices with the gentle surface the finer levels are helper-made blocks
    that shrink edit by edit while the primary's rebuilt blocks are dropped and appended again (the reference's vector order,
    src/TransVoxelImpl.cpp:443-464, :1274-1293)."""
    from voxels_amd import build
    build.build_cpp_api()
    build.build_dropin_tests()
    if not os.path.exists(REF):
        pytest.skip("oracle/_ref/dropin_ref not built (needs /root/reference at build time)")
    a, b = str(tmp_path / "ours.bin"), str(tmp_path / "ref.bin")
    env = dict(os.environ, DROPIN_EDITS="4")
    if gentle:
        env["DROPIN_GENTLE"] = "1"
    subprocess.check_call([OURS, str(n), a], env=dict(env, VOXELS_DEVICES=str(devices)), timeout=300)
    subprocess.check_call([REF, str(n), b], env=env, timeout=300)
    da, db = open(a, "rb").read(), open(b, "rb").read()
    assert len(da) == len(db), (len(da), len(db))
    if da != db:
        first = next(i for i in range(len(da)) if da[i] != db[i])
        raise AssertionError("dumps differ at byte %d of %d" % (first, len(da)))


@pytest.mark.gpu
def test_device_mirror_tracks_grids_and_edits():
    """a Polygonizer reused on a new grid at a recycled address must upload it; two Polygonizers on one grid
    must both see an edit (tests/cpp/mirror_test.cpp)."""
    from voxels_amd import build
    build.build_cpp_api()
    build.build_dropin_tests()
    out = subprocess.run([os.path.join(ROOT, "tests", "cpp", "mirror_ours")], capture_output=True, text=True, timeout=300)
    assert out.returncode == 0 and out.stdout.strip().endswith("OK"), out.stdout + out.stderr

"""Address arithmetic of the grid's mirrors (brick layout of the fields and of the lattice copies), checked
on the host: tests/cpp/brick_layout_test.cpp includes the product headers and verifies that the maps are bijections into
their allocations and keep rows / lines together."""
import os
import subprocess

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_brick_and_lattice_offsets(tmp_path):
    exe = str(tmp_path / "brick_layout_test")
    subprocess.check_call(["g++", "-std=c++14", "-O1", "-Wno-unknown-pragmas", "-o", exe, os.path.join(ROOT, "tests", "cpp", "brick_layout_test.cpp")])
    out = subprocess.run([exe], capture_output=True, text=True, timeout=120)
    assert out.returncode == 0 and out.stdout.strip() == "OK", out.stdout + out.stderr

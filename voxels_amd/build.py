"""Build helpers: compile the HIP library for gfx950 in-tree (the .so travels with the repo snapshot)."""
import os
import shutil
import subprocess

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(HERE)
CSRC = os.path.join(HERE, "csrc")

HIP_FLAGS = ["--offload-arch=gfx950", "-O3", "-std=c++17", "-ffp-contract=off", "-fPIC", "-shared", "-ldl"]


def _newer(target, sources):
    if not os.path.exists(target):
        return True
    t = os.path.getmtime(target)
    return any(os.path.getmtime(s) > t for s in sources)


def build_hip(force=False, verbose=True):
    """hipcc --offload-arch=gfx950 ... -o voxels_amd/csrc/libvoxels_hip.so (cross-compiles without a GPU)."""
    out = os.path.join(CSRC, "libvoxels_hip.so")
    srcs = [os.path.join(CSRC, f) for f in ("vx_hip.hip", "vx_regular0.inl", "vx_fast0.inl", "vx_fast1.inl", "vx_main.inl", "vx_host.inl", "tv_block.h", "tv_core.h", "tv_fast0.h", "tv_fast1.h", "tv_tables.inc")]
    srcs.append(os.path.join(ROOT, "include", "voxels_hip.h"))
    if not force and not _newer(out, srcs):
        return out
    hipcc = shutil.which("hipcc") or "/opt/rocm/bin/hipcc"
    cmd = [hipcc] + HIP_FLAGS + ["-Rpass-analysis=kernel-resource-usage", "-o", out, os.path.join(CSRC, "vx_hip.hip")]
    if verbose:
        print(" ".join(cmd))
    r = subprocess.run(cmd, cwd=CSRC, capture_output=True, text=True)
    table = kernel_resources(r.stderr)
    if r.returncode != 0:
        raise RuntimeError("hipcc failed:\n" + "\n".join(l for l in r.stderr.splitlines() if "remark:" not in l)[-4000:])
    # Policy: no kernel of the library spills or keeps arrays in scratch memory (a spilled variant of k_regular produced
    # wrong vertices now and then).  VX_ALLOW_SCRATCH=1 lifts the check for experiments.
    bad = [(k, v["ScratchSize"]) for k, v in table.items() if v.get("ScratchSize", 0)]
    # the gate must not pass because the remark format changed and nothing was parsed
    missing = [k for k in ("k_regular0", "k_regular", "k_transition", "k_classify", "k_material", "k_main", "k_tail", "k_run_head", "k_dirty_head", "k_dirty_tail")
               if not any(k in name and "ScratchSize" in v for name, v in table.items())]
    if missing:
        os.remove(out)
        raise RuntimeError("kernel resource remarks not found for %s: the no-scratch policy cannot be checked (hipcc remark format changed?)" % missing)
    with open(os.path.join(CSRC, "kernel_resources.txt"), "w") as f:
        f.write("%-72s %6s %8s %10s %10s\n" % ("kernel", "VGPRs", "scratch", "occupancy", "staticLDS"))
        for k, v in sorted(table.items()):
            f.write("%-72s %6d %8d %10d %10d\n" % (k[:72], v.get("VGPRs", 0), v.get("ScratchSize", 0), v.get("Occupancy", 0), v.get("LDS Size", 0)))
    if bad and not os.environ.get("VX_ALLOW_SCRATCH"):
        # A frame can be reserved without ever being touched (spill slots of scalar registers that were all turned into
        # vector-register lanes afterwards): what the policy forbids is scratch TRAFFIC, so a flagged kernel passes if its
        # code holds no instruction that can reach the private segment.  One device-only compile to assembly serves all
        # flagged kernels.
        touching = _kernels_touching_scratch(hipcc, [k for k, _ in bad])
        bad = [(k, n) for k, n in bad if k in touching]
    if bad and not os.environ.get("VX_ALLOW_SCRATCH"):
        os.remove(out)
        raise RuntimeError("kernels using scratch memory: %s" % bad)
    return out


def _kernels_touching_scratch(hipcc, kernels):
    """Which of `kernels` (mangled names) hold an instruction that can reach the private segment: any scratch_* instruction,
    or any buffer_* access whose resource is the private-segment descriptor (s[0:3] under the default calling convention -
    also the constant-offset forms without `offen`).  A kernel whose code cannot be found counts as touching."""
    import re
    import tempfile
    touching = set(kernels)
    with tempfile.TemporaryDirectory() as tmp:
        asm = os.path.join(tmp, "vx.s")
        flags = [f for f in HIP_FLAGS if f not in ("-shared", "-fPIC", "-ldl")]
        r = subprocess.run([hipcc] + flags + ["-S", "--cuda-device-only", "-o", asm, os.path.join(CSRC, "vx_hip.hip")], cwd=CSRC, capture_output=True, text=True)
        if r.returncode != 0 or not os.path.exists(asm):
            return touching
        current, dirty, seen = None, False, set()
        with open(asm) as f:
            for line in f:
                m = re.match(r"^(\S+):", line)
                if m and m.group(1) in touching | seen:
                    current, dirty = m.group(1), False
                    seen.add(current)
                    continue
                if current is None:
                    continue
                if line.startswith(".Lfunc_end"):
                    if not dirty:
                        touching.discard(current)
                    current = None
                    continue
                code = line.split(";")[0]
                if "scratch_" in code or (re.search(r"\bbuffer_(load|store|atomic)", code) and re.search(r"s\[0:3\]", code)):
                    dirty = True
    return touching


def kernel_resources(remarks):
    """-Rpass-analysis=kernel-resource-usage remarks -> {kernel: {VGPRs, SGPRs, ScratchSize, Occupancy, LDS Size}}"""
    import re
    table, cur = {}, None
    for line in remarks.splitlines():
        m = re.search(r"remark:\s+Function Name: (\S+)", line)
        if m:
            cur = table.setdefault(m.group(1), {})
            continue
        m = re.search(r"remark:\s+(VGPRs|SGPRs|ScratchSize \[bytes/lane\]|Occupancy \[waves/SIMD\]|LDS Size \[bytes/block\]): (\d+)", line)
        if m and cur is not None:
            cur[m.group(1).split(" [")[0]] = int(m.group(2))
    return table


def build_hip_casedump(force=False):
    """Test build of the HIP library that also records the case codes it looks up (tests/test_case_codes.py)."""
    out = os.path.join(CSRC, "libvoxels_hip_casedump.so")
    srcs = [os.path.join(CSRC, f) for f in ("vx_hip.hip", "vx_regular0.inl", "vx_fast0.inl", "vx_fast1.inl", "vx_main.inl", "vx_host.inl", "tv_block.h", "tv_core.h", "tv_fast0.h", "tv_fast1.h", "tv_tables.inc")]
    if not force and not _newer(out, srcs):
        return out
    hipcc = shutil.which("hipcc") or "/opt/rocm/bin/hipcc"
    subprocess.check_call([hipcc] + HIP_FLAGS + ["-DVX_CASE_DUMP", "-o", out, os.path.join(CSRC, "vx_hip.hip")], cwd=CSRC)
    return out


def build_hip_conservative(force=False):
    """Test build of the HIP library whose in-kernel dependency flags use release / acquire fences instead of write-through
    stores and loads (-DVX_CONSERVATIVE_SYNC, vx_hip.hip): tests/test_gpu_parity.py compares it with the product library."""
    out = os.path.join(CSRC, "libvoxels_hip_conservative.so")
    srcs = [os.path.join(CSRC, f) for f in ("vx_hip.hip", "vx_regular0.inl", "vx_fast0.inl", "vx_fast1.inl", "vx_main.inl", "vx_host.inl", "tv_block.h", "tv_core.h", "tv_fast0.h", "tv_fast1.h", "tv_tables.inc")]
    if not force and not _newer(out, srcs):
        return out
    hipcc = shutil.which("hipcc") or "/opt/rocm/bin/hipcc"
    subprocess.check_call([hipcc] + HIP_FLAGS + ["-DVX_CONSERVATIVE_SYNC", "-o", out, os.path.join(CSRC, "vx_hip.hip")], cwd=CSRC)
    return out


def build_synth(force=False):
    """Host-side synthetic input generator (include/voxels_synth.h)."""
    out = os.path.join(CSRC, "libvoxels_synth.so")
    srcs = [os.path.join(CSRC, "vx_synth.cpp"), os.path.join(CSRC, "vx_terrain_math.h"), os.path.join(ROOT, "include", "voxels_synth.h")]
    if not force and not _newer(out, srcs):
        return out
    subprocess.check_call(["g++", "-std=c++14", "-O2", "-fPIC", "-shared", "-fopenmp", "-ffp-contract=off", "-o", out,
                           os.path.join(CSRC, "vx_synth.cpp")], cwd=CSRC)
    return out


def build_cpp_api(force=False):
    """libVoxels.so: the reference's public C++ API (include/Voxels.h) on top of libvoxels_hip.so."""
    out = os.path.join(CSRC, "libVoxels.so")
    srcs = [os.path.join(CSRC, f) for f in ("vx_api_cpp.cpp", "vx_grid_host.cpp", "vx_grid_host.h")]
    srcs += [os.path.join(ROOT, "include", f) for f in ("Voxels.h", "voxels_hip.h")]
    if not force and not _newer(out, srcs + [os.path.join(CSRC, "libvoxels_hip.so")]):
        return out
    subprocess.check_call(["g++", "-std=c++14", "-O2", "-fPIC", "-shared", "-fvisibility=hidden", "-o", out,
                           os.path.join(CSRC, "vx_api_cpp.cpp"), os.path.join(CSRC, "vx_grid_host.cpp"),
                           "-L" + CSRC, "-lvoxels_hip", "-Wl,-rpath,$ORIGIN"], cwd=CSRC)
    return out


def build_dropin_tests(force=False):
    """tests/cpp/dropin_test.cpp compiled against OUR headers + libVoxels.so, and (where /root/reference exists)
    the very same source against the REFERENCE headers + the unmodified reference library (oracle/_ref)."""
    src = os.path.join(ROOT, "tests", "cpp", "dropin_test.cpp")
    out = os.path.join(ROOT, "tests", "cpp", "dropin_ours")
    if force or _newer(out, [src, os.path.join(CSRC, "libVoxels.so")]):
        subprocess.check_call(["g++", "-std=c++14", "-O2", "-o", out, src, "-I" + os.path.join(ROOT, "include"),
                               "-L" + CSRC, "-lVoxels", "-Wl,-rpath," + CSRC])
    msrc = os.path.join(ROOT, "tests", "cpp", "mirror_test.cpp")
    mout = os.path.join(ROOT, "tests", "cpp", "mirror_ours")
    if force or _newer(mout, [msrc, os.path.join(CSRC, "libVoxels.so")]):
        subprocess.check_call(["g++", "-std=c++14", "-O2", "-o", mout, msrc, "-I" + os.path.join(ROOT, "include"),
                               "-L" + CSRC, "-lVoxels", "-Wl,-rpath," + CSRC])
    ref_out = os.path.join(ROOT, "oracle", "_ref", "dropin_ref")
    ref_lib = os.path.join(ROOT, "oracle", "_ref", "libvoxels_ref.so")
    if os.path.isdir("/root/reference/include") and os.path.exists(ref_lib) and (force or _newer(ref_out, [src, ref_lib])):
        subprocess.check_call(["/opt/rocm/lib/llvm/bin/clang++", "-std=c++14", "-O2", "-w", "-fms-extensions", "-fdeclspec",
                               "-DVOXELS_API=", "-DVOXELS_CDECL=", "-o", ref_out, src, "-I/root/reference/include",
                               "-L" + os.path.dirname(ref_lib), "-lvoxels_ref", "-Wl,-rpath,$ORIGIN", "-Wl,-rpath,/opt/rocm/lib/llvm/lib"])
    return out, ref_out


def build_dropin_bench(force=False):
    """tools/dropin_bench.cpp: end-to-end Polygonizer::Execute through libVoxels.so (used by bench.py's e2e figure)."""
    src = os.path.join(ROOT, "tools", "dropin_bench.cpp")
    out = os.path.join(ROOT, "tools", "dropin_bench")
    if force or _newer(out, [src, os.path.join(CSRC, "libVoxels.so")]):
        subprocess.check_call(["g++", "-std=c++14", "-O2", "-o", out, src, "-I" + os.path.join(ROOT, "include"),
                               "-L" + CSRC, "-lVoxels", "-Wl,-rpath," + CSRC])
    return out


def build_emu(force=False):
    """CPU emulation of the device phases — tests only (tests/emu)."""
    d = os.path.join(ROOT, "tests", "emu")
    out = os.path.join(d, "libvoxels_emu.so")
    srcs = [os.path.join(d, "emu.cpp"), os.path.join(d, "emu_backend.inl")] + \
           [os.path.join(CSRC, f) for f in ("vx_host.inl", "tv_block.h", "tv_core.h", "tv_fast0.h", "tv_fast1.h", "tv_tables.inc")]
    if not force and not _newer(out, srcs):
        return out
    subprocess.check_call(["g++", "-std=c++14", "-O2", "-fPIC", "-shared", "-ffp-contract=off", "-Wno-unknown-pragmas",
                           "-Wno-subobject-linkage", "-o", out, os.path.join(d, "emu.cpp")], cwd=d)
    return out


def build_oracle(with_reference=True):
    """The CPU checkers (test infrastructure): the port always, the unmodified reference when /root/reference exists."""
    subprocess.check_call(["make", "-C", os.path.join(ROOT, "oracle"), "port"])
    if with_reference and os.path.isdir("/root/reference/src"):
        subprocess.check_call(["make", "-C", os.path.join(ROOT, "oracle"), "ref"])

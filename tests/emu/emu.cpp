// tests/emu/emu.cpp — CPU emulation of the device pipeline, TEST INFRASTRUCTURE ONLY.
//
// It compiles the same per-cell logic (voxels_amd/csrc/tv_core.h), the same block phases (tv_block.h) and the
// same host orchestration (vx_host.inl) as the product, but runs the phases sequentially (tid = 0, one
// "thread") with serial scans instead of HIP kernels.  tests/test_emu.py uses it to check the closed-form
// parallel formulation and the host logic against the oracle on machines without a GPU.  The product library
// (libvoxels_hip.so) is never built from this file and never loads it.
#include <cstdlib>
#include <cstring>
#include <string>
#include <vector>

#include "../../voxels_amd/csrc/tv_block.h"
#include "../../voxels_amd/csrc/tv_fast0.h"
#include "../../voxels_amd/csrc/tv_fast1.h"
#include "../../voxels_amd/csrc/vx_terrain_math.h"

#define VX_BACKEND_NAME "emu:cpu (tests only)"

namespace {
using namespace tv;

template <typename T>
u32 exclusive_scan(T* a, u32 n)
{
	u32 run = 0;
	for (u32 i = 0; i < n; ++i) { const u32 v = a[i]; a[i] = (T)run; run += v; }
	return run;
}
}

#include "emu_backend.inl"
#include "../../voxels_amd/csrc/vx_host.inl"

// test hook for tests/test_core_math.py
extern "C" int emu_edge_t(int v0, int v1) { return tv::edge_t(v0, v1); }
extern "C" int emu_edge_end(int v0, int v1) { return tv::edge_end(v0, v1); }

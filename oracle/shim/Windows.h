// oracle/shim/Windows.h — stand-in for <Windows.h> so the UNMODIFIED reference sources
// (/root/reference/src/*.cpp) compile on Linux.  Test infrastructure only; written for
// this repo (nothing here comes from the reference).  Needed by stdafx.h:8,
// TransVoxelImpl.cpp:131 (_countof), :508 (GetCurrentThreadId), Voxels.cpp:22-27
// (_aligned_malloc/_aligned_free).
#pragma once
#include <cstdlib>
#include <cstring>
#include <cstdio>
#include <cmath>
#include <cfloat>
#include <cassert>
#include <climits>
#include <limits>
#include <thread>
#include <functional>
#include <algorithm>
#include <memory>
#include <vector>
#include <malloc.h>

#ifndef _countof
#define _countof(a) (sizeof(a) / sizeof((a)[0]))
#endif

static inline unsigned long GetCurrentThreadId()
{
	return (unsigned long)std::hash<std::thread::id>()(std::this_thread::get_id());
}

static inline void* _aligned_malloc(size_t size, size_t alignment)
{
	void* p = nullptr;
	if (alignment < sizeof(void*)) alignment = sizeof(void*);
	if (posix_memalign(&p, alignment, size) != 0) return nullptr;
	return p;
}

static inline void _aligned_free(void* p) { free(p); }

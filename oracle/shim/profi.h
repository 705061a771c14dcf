// oracle/shim/profi.h — no-op stand-ins for the un-vendored "profi" profiler macros used at
// TransVoxelImpl.cpp:470,971,1054,1127,1177,1268,1531,1533,1756 and VoxelGrid.cpp:87,...
// Test infrastructure only.
#pragma once
#define PROFI_FUNC
#define PROFI_SCOPE(x)
#define PROFI_SCOPE_S1(x)
#define PROFI_SCOPE_S2(x)
#define PROFI_SCOPE_S3(x)

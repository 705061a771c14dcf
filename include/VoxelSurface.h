// VoxelSurface.h — kept so that '#include "VoxelSurface.h"' (reference include/VoxelSurface.h) keeps working; everything lives in Voxels.h.
#pragma once
#include "Voxels.h"

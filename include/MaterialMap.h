// MaterialMap.h — kept so that '#include "MaterialMap.h"' (reference include/MaterialMap.h) keeps working; everything lives in Voxels.h.
#pragma once
#include "Voxels.h"

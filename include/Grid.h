// Grid.h — kept so that '#include "Grid.h"' (reference include/Grid.h) keeps working; everything lives in Voxels.h.
#pragma once
#include "Voxels.h"

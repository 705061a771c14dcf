// Polygonizer.h — kept so that '#include "Polygonizer.h"' (reference include/Polygonizer.h) keeps working; everything lives in Voxels.h.
#pragma once
#include "Voxels.h"

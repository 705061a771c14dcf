// Structs.h — kept so that '#include "Structs.h"' (reference include/Structs.h) keeps working; everything lives in Voxels.h.
#pragma once
#include "Voxels.h"

// Declarations.h — kept so that '#include "Declarations.h"' (reference include/Declarations.h) keeps working; everything lives in Voxels.h.
#pragma once
#include "Voxels.h"

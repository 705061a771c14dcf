// Library.h — kept so that '#include "Library.h"' (reference include/Library.h) keeps working; everything lives in Voxels.h.
#pragma once
#include "Voxels.h"

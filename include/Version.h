// Version.h — kept so that '#include "Version.h"' (reference include/Version.h) keeps working; everything lives in Voxels.h.
#pragma once
#include "Voxels.h"

// Library identity + small shared entry points.
#include "dc_common.h"

extern "C" const char* dc_version(void) { return "depthcore 0.3.0 (round 3)"; }
extern "C" const char* dc_arch(void) { return "gfx950"; }

// Library identity + small shared entry points.
#include "dc_common.h"

extern "C" const char* dc_version(void) { return "depthcore 0.2.0 (round 2)"; }
extern "C" const char* dc_arch(void) { return "gfx950"; }

// Library identity + small shared entry points.
#include "dc_common.h"

extern "C" const char* dc_version(void) { return "depthcore 0.1.0 (round 1)"; }
extern "C" const char* dc_arch(void) { return "gfx950"; }

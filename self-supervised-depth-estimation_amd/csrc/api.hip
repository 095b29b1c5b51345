// Library identity + small shared entry points.
#include "dc_common.h"

extern "C" const char* dc_version(void) { return "depthcore 0.4.0 (round 4)"; }
extern "C" const char* dc_arch(void) { return "gfx950"; }

// Every entry point reports a failed launch by reading HIP's per-thread "last error" (DC_CHECK_LAUNCH).  An error raised by
// something else on the calling thread -- a hipGraph capture that was invalidated, a refused call of the framework's --
// stays there until somebody reads it, and the next dc_* launch would take the blame.  Returns (and clears) that code.
extern "C" int dc_clear_error(void) { return (int)hipGetLastError(); }

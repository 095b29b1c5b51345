// Library identity + small shared entry points.
#include "dc_common.h"

extern "C" const char* dc_version(void) { return "depthcore 0.6.0 (round 6)"; }
extern "C" const char* dc_arch(void) { return "gfx950"; }

// Every entry point reports a failed launch by reading HIP's per-thread "last error" (DC_CHECK_LAUNCH).  An error raised by
// something else on the calling thread -- a hipGraph capture that was invalidated, a refused call of the framework's --
// stays there until somebody reads it, and the next dc_* launch would take the blame.  Returns (and clears) that code.
extern "C" int dc_clear_error(void) { return (int)hipGetLastError(); }

// A hipGraph capture that was invalidated leaves its origin stream in capture mode until somebody ends the capture; every
// later launch on that stream (and on the streams that joined it) then fails.  Ends whatever capture `stream` is in, discards
// the graph, clears the error state.  Returns the capture status found (0 none, 1 active, 2 invalidated).
extern "C" int dc_abort_capture(void* stream) {
    hipStream_t st = (hipStream_t)stream;
    hipStreamCaptureStatus cs = hipStreamCaptureStatusNone;
    // (the query itself FAILS with hipErrorStreamCaptureInvalidated on an invalidated capture: that is the case to repair)
    const bool query_failed = hipStreamIsCapturing(st, &cs) != hipSuccess;
    (void)hipGetLastError();
    if (query_failed || cs != hipStreamCaptureStatusNone) {
        hipGraph_t g = nullptr;
        (void)hipStreamEndCapture(st, &g);
        if (g) (void)hipGraphDestroy(g);
        if (query_failed) cs = hipStreamCaptureStatusInvalidated;
    }
    (void)hipGetLastError();
    return (int)cs;
}

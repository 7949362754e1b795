// pf_host.h -- host-side plumbing shared by the C-ABI translation units: error channel, HIP checks.
#pragma once

#include <hip/hip_runtime.h>
#include <cstdarg>
#include <cstdio>
#include <cstdlib>
#include <cstring>

namespace pf {

// thread-local message behind pfx_last_error()
char* last_error_buf();
void set_error(const char* fmt, ...);
void clear_error();

// The reference API is void and cannot report failure; a GPU library can fail (no device, out of
// memory).  Failing silently -- or falling back to some CPU path -- would hide that the HIP path is
// not the one running, so the reference-layout entry points abort loudly instead.
[[noreturn]] void die(const char* what);
// The same report for callers that would rather survive it: after pfx_set_abort_on_error(0) the message stays in
// pfx_last_error(), pfx_error_count() goes up by one and fail() RETURNS -- the entry point then returns at once, the
// call having had no effect.  Default (and whenever the mode is 1): die().
void fail(const char* what);

// Philox key word 1 of the next filter initialised through the reference-layout API: a process-wide counter that
// particleFilterSeed_set puts back to 0, so that "set the seed, initialise, run" reproduces its sample path however
// many filters the process has initialised before (the reference reseeds its one global stream, pfRandom.c:51-60).
uint32_t next_filter_stream();

#define PF_HIP_OK(expr)                                                                         \
    do {                                                                                        \
        hipError_t e_ = (expr);                                                                 \
        if (e_ != hipSuccess) {                                                                 \
            pf::set_error("%s failed: %s (%s:%d)", #expr, hipGetErrorString(e_), __FILE__, __LINE__); \
            return -1;                                                                          \
        }                                                                                       \
    } while (0)

}  // namespace pf

// pf_host.h -- host-side plumbing shared by the C-ABI translation units: error channel, HIP checks.
#pragma once

#include <hip/hip_runtime.h>
#include <cstdarg>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <mutex>

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
// pfRandom.c:56-60: a filter initialised before any particleFilterSeed_set is seeded from the clock.  PF_HIP_SEED (an
// unsigned decimal) stands in for the clock, so that programs which never set a seed -- csvlocalize.c, csvslam.c, the JNI
// shim -- can be run reproducibly without a source change.
unsigned int unset_seed();

// The reference-layout entry points (particleFilter.h, pfStages.h) take this lock for the length of a call.  The
// reference is not thread-safe (one global rand_r state, pfRandom.c:51-52), yet one of its callers drives a filter from
// two threads without a lock of its own -- mqttlocalize.c: the MQTT callback thread deposits (:294,301) while main polls
// getTagLoc and reads lastX/Y/Z (:116,129-131).  There a race costs a torn pose; here a deposit and a pose query
// interleaving on one filter's host-side state machine (deferred resampling, pending launches) would cost the filter.
// One process-wide lock, recursive because entry points forward to each other; the pfx_* extension stays lock-free
// (one thread per filter or batch, as include/pfx.h says).
std::recursive_mutex& api_mutex();
#define PF_API_LOCK std::lock_guard<std::recursive_mutex> pf_api_lock_(pf::api_mutex())

#define PF_HIP_OK(expr)                                                                         \
    do {                                                                                        \
        hipError_t e_ = (expr);                                                                 \
        if (e_ != hipSuccess) {                                                                 \
            pf::set_error("%s failed: %s (%s:%d)", #expr, hipGetErrorString(e_), __FILE__, __LINE__); \
            return -1;                                                                          \
        }                                                                                       \
    } while (0)

}  // namespace pf

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

#define PF_HIP_OK(expr)                                                                         \
    do {                                                                                        \
        hipError_t e_ = (expr);                                                                 \
        if (e_ != hipSuccess) {                                                                 \
            pf::set_error("%s failed: %s (%s:%d)", #expr, hipGetErrorString(e_), __FILE__, __LINE__); \
            return -1;                                                                          \
        }                                                                                       \
    } while (0)

}  // namespace pf

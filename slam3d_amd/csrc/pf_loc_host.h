// pf_loc_host.h -- host-side pieces of a Loc deposit shared by the translation units that launch the Loc kernels
// (pf_loc.hip: one filter; pf_loc_batch.hip: many filters in one set of launches): the VIO accumulator
// (particleFilter.c:70-104, 389-404), the per-deposit scalars derived from it (pfMeasurement.c:31-32), the range gate's
// squared-distance bounds (pfMeasurement.c:118-123), device selection and launch checks.
#pragma once

#include "pf_loc_kernels.h"
#include "pf_host.h"

#include <cmath>
#include <cstring>
#include <limits>

namespace pf {

struct VioState {               // particleFilter.h:52-62 without the `initialized` byte
    double firstT = 0.0; float firstX = 0, firstY = 0, firstZ = 0, firstDist = 0;
    double lastT = 0.0;  float lastX = 0, lastY = 0, lastZ = 0, lastDist = 0;
};

inline void vio_deposit(VioState& v, double t, float x, float y, float z, float dist)
{   // particleFilter.c:70-104
    if (v.firstT == 0.0) {
        v.firstT = v.lastT = t;
        v.firstX = v.lastX = x; v.firstY = v.lastY = y; v.firstZ = v.lastZ = z;
        v.firstDist = v.lastDist = dist;
        return;
    }
    if (dist > v.lastDist) {
        v.lastDist = dist;
    } else {
        const float dx = x - v.lastX, dy = y - v.lastY, dz = z - v.lastZ;
        v.lastDist += sqrtf(dx * dx + dy * dy + dz * dz);
    }
    v.lastT = t; v.lastX = x; v.lastY = y; v.lastZ = z;
}

inline void vio_commit(VioState& v, float d[5])
{   // particleFilter.c:389-404
    d[0] = (float)(v.lastT - v.firstT);
    d[1] = v.lastX - v.firstX; d[2] = v.lastY - v.firstY; d[3] = v.lastZ - v.firstZ;
    d[4] = v.lastDist - v.firstDist;
    v.firstT = v.lastT; v.firstX = v.lastX; v.firstY = v.lastY; v.firstZ = v.lastZ; v.firstDist = v.lastDist;
}

struct DeviceGuard {
    int prev = -1;
    explicit DeviceGuard(int dev) { hipGetDevice(&prev); if (prev != dev) hipSetDevice(dev); else prev = -1; }
    ~DeviceGuard() { if (prev >= 0) hipSetDevice(prev); }
};

inline int check_launch(const char* what)
{
    hipError_t e = hipGetLastError();
    if (e != hipSuccess) { set_error("launch of %s failed: %s", what, hipGetErrorString(e)); return -1; }
    return 0;
}

inline Motion motion_of(const float d[5])
{   // pfMeasurement.c:31-32
    return Motion{ d[1], d[2], d[3], sqrtf(d[4]) * kVioStdXyz, sqrtf(d[0]) * kVioStdTheta };
}
inline bool is_zero_motion(const float d[5])
{   // exact zeros only: then x,y,z are unchanged bit for bit and theta only wraps (NaN compares false -> motion)
    return d[0] == 0.0f && d[1] == 0.0f && d[2] == 0.0f && d[3] == 0.0f && d[4] == 0.0f;
}
inline float float_of_bits(uint32_t b) { float f; memcpy(&f, &b, 4); return f; }

inline Gate gate_of(float bx, float by, float bz, float range, float stdRange)
{   // pfMeasurement.c:22,114,122
    const float inf = std::numeric_limits<float>::infinity();
    Gate g{ bx, by, bz, range, 3 * stdRange, (range < 3.0f) ? 0.1f : 0.5f, 0.0f, inf };    // bounds: everything kept
    const float r0 = g.range, t0 = g.thresh;
    if (std::isnan(r0) || std::isnan(t0) || t0 == inf) return g;            // fabsf(..) > thresh is never true
    if (t0 == -inf || r0 == -inf) { g.d2Lo = inf; g.d2Hi = -1.0f; return g; }   // always true: nothing kept (but NaN)
    if (r0 == inf) { g.d2Lo = inf; g.d2Hi = inf; return g; }               // only sqrtf(inf) - inf = NaN is kept
    // e(d2) = sqrtf(d2) - range (float) is non-decreasing in d2 >= 0; the gate keeps |e| <= thresh.  Non-negative floats
    // are ordered like their bit patterns (0 .. 0x7F800000 = +inf): bisect for the first d2 with e >= -thresh and the
    // first with e > thresh (both exist: e(+inf) = +inf).
    const volatile float r = r0, t = t0;
    auto e_of = [&](uint32_t bits) { const volatile float p = sqrtf(float_of_bits(bits)); const volatile float e = p - r; return (float)e; };
    uint32_t lo = 0, hi = 0x7F800000u;
    while (lo < hi) { const uint32_t mid = lo + (hi - lo) / 2; if (e_of(mid) >= -t) hi = mid; else lo = mid + 1; }
    const uint32_t first = lo;
    lo = 0; hi = 0x7F800000u;
    while (lo < hi) { const uint32_t mid = lo + (hi - lo) / 2; if (e_of(mid) > t) hi = mid; else lo = mid + 1; }
    const uint32_t past = lo;
    if (first >= past) { g.d2Lo = inf; g.d2Hi = -1.0f; }                    // nothing is kept
    else { g.d2Lo = float_of_bits(first); g.d2Hi = float_of_bits(past - 1); }
    return g;
}

// particleFilter.c:274-290 on the six pose sums {w, w x, w y, w z, w cos, w sin}: weighted mean, circular mean of the
// heading, plus the VIO delta that has not been committed yet, rotated by the estimated heading
inline void pose_from_sums_vio(const VioState& v, const double p[6], double* t, float* x, float* y, float* z, float* theta)
{
    const float s = (float)p[0], xs = (float)p[1], ys = (float)p[2], zs = (float)p[3], cs = (float)p[4], sn = (float)p[5];
    *t = v.lastT;
    *x = xs / s;
    *y = ys / s;
    *z = zs / s;
    *theta = atan2f(sn, cs);
    const float dx = v.lastX - v.firstX, dy = v.lastY - v.firstY, dz = v.lastZ - v.firstZ;
    const float co = cosf(*theta), si = sinf(*theta);
    *x += dx * co - dy * si;
    *y += dx * si + dy * co;
    *z += dz;
}

}  // namespace pf

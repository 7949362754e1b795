// pf_common.h -- device-side building blocks shared by the particle-filter kernels (gfx950).
//
// Arithmetic follows the reference's float32 expressions term by term; the translation units
// are compiled with -ffp-contract=off so hipcc never fuses a multiply-add (the reference is built
// for baseline x86-64, which has none).  Only the noise source differs: counter-based
// Philox4x32-10 instead of the serial rand_r stream (pfRandom.c:26-49).
#pragma once

#include <hip/hip_runtime.h>
#include <stdint.h>

namespace pf {

constexpr int kThreads = 256;               // 4 wavefronts of 64
constexpr int kPerThread = 4;               // one float4 per array per thread
constexpr int kTile = kThreads * kPerThread; // 1024 particles per workgroup tile

constexpr float kPi = 3.14159274101257324219f;      // (float)M_PI
constexpr float kTwoPi = 6.28318548202514648438f;   // 2 * (float)M_PI  (pfMeasurement.c:47)
constexpr float kInv2p31 = 4.656612873077392578125e-10f;  // 1 / 2^31, exact

// pfMeasurement.c:20-22, pfResample.c:22-26
constexpr float kVioStdXyz = 1e-3f;
constexpr float kVioStdTheta = 1e-6f;
constexpr float kResampleThresh = 0.5f;
constexpr float kRadiusSpawnThresh = 4.0f;
constexpr float kWeightSpawnThresh = 0.4f;
constexpr float kPctSpawn = 0.05f;
constexpr float kHxyz = 0.1f;

enum : uint32_t { P_VIO = 1, P_RSTART = 2, P_JITTER = 3, P_SPAWN = 4 };

// ---------------------------------------------------------------- weights as they enter a sum
// CDF rule (oracle/pf_oracle.c cdf_weight): a weight enters every weight total and every CDF rounded
// to the nearest multiple of 2^-20, and those are added in float64.  Multiples of 2^-20 that total less than 2^33
// add up EXACTLY in float64, whatever the order: the tile / chunk / shard / piece structure of the sums below, the
// deferred and the undeferred path, one GPU or sixteen all produce the same CDF bit for bit, and so does the
// oracle's plain sequential loop.  (The reference's serial float32 sum, pfResample.c:45,49, resolves 2^-10 at
// N = 10 000 and nothing at all past 2^24.)  Three full-rate float32 instructions per particle.
constexpr float kCdfScale = 1048576.0f;                     // 2^20
constexpr float kCdfInvScale = 9.5367431640625e-07f;        // 2^-20
constexpr double kCdfExactBelow = 8589934592.0;             // 2^33: totals below this are exact
// tiles whose weight total stays below this have their CDF offsets in 32-bit integers (units of 2^-20)
constexpr double kTileU32Below = 4095.0;

__host__ __device__ __forceinline__ float cdf_units(float w)
{   // the weight in units of 2^-20, an integer-valued float (round to nearest even: v_rndne_f32 / rintf)
    return __builtin_rintf(w * kCdfScale);
}
__host__ __device__ __forceinline__ float cdf_weight(float w) { return cdf_units(w) * kCdfInvScale; }

// ---------------------------------------------------------------- Philox4x32-10
struct u32x4 { uint32_t a, b, c, d; };

__host__ __device__ __forceinline__ u32x4 philox4x32_10(uint32_t c0, uint32_t c1, uint32_t c2, uint32_t c3,
                                                       uint32_t k0, uint32_t k1)
{
#pragma unroll
    for (int r = 0; r < 10; ++r) {
        const uint64_t p0 = (uint64_t)0xD2511F53u * c0;
        const uint64_t p1 = (uint64_t)0xCD9E8D57u * c2;
#if defined(__HIP_DEVICE_COMPILE__)
        // three-input xor in one instruction (v_bitop3_b32, truth table 0x96); the compiler does not form it itself
        const uint32_t n0 = __builtin_amdgcn_bitop3_b32((uint32_t)(p1 >> 32), c1, k0, 0x96);
        const uint32_t n2 = __builtin_amdgcn_bitop3_b32((uint32_t)(p0 >> 32), c3, k1, 0x96);
#else
        const uint32_t n0 = (uint32_t)(p1 >> 32) ^ c1 ^ k0;
        const uint32_t n2 = (uint32_t)(p0 >> 32) ^ c3 ^ k1;
#endif
        c1 = (uint32_t)p1; c3 = (uint32_t)p0; c0 = n0; c2 = n2;
        k0 += 0x9E3779B9u; k1 += 0xBB67AE85u;
    }
    return { c0, c1, c2, c3 };
}

// The sub-stream of (seed, stream; epoch, purpose, particle index): block b supplies words 4b..4b+3.
struct RngKey { uint32_t seed, stream, epoch; };

__host__ __device__ __forceinline__ u32x4 rng_block(const RngKey& k, uint32_t purpose, uint64_t index, uint32_t block)
{
#ifdef PF_EXPERIMENT_NO_PHILOX      // tools/dev timing experiments only: what the kernels cost without the generator
    return { (uint32_t)index * 2654435761u, k.epoch + purpose, (uint32_t)index + block, k.seed };
#endif
    return philox4x32_10((uint32_t)index, (uint32_t)(index >> 32), k.epoch, purpose + 16u * block, k.seed, k.stream);
}

// rand_r_func() stand-in: a 31-bit value from one Philox word
__host__ __device__ __forceinline__ uint32_t r31(uint32_t word) { return word >> 1; }

// pfRandom.c:62-65 -- (float)r / RAND_MAX, RAND_MAX rounding to 2^31 in float: [0,1] inclusive
__host__ __device__ __forceinline__ float uniform01(uint32_t word) { return (float)r31(word) * kInv2p31; }
// pfRandom.c:100-103 -- (r + 1) / 2^31 in (0,1]
__host__ __device__ __forceinline__ float uniform_nonzero(uint32_t word) { return (float)(r31(word) + 1u) * kInv2p31; }

// ---------------------------------------------------------------- float transcendentals, sized for this path
// The streaming and resampling kernels are bound by vector-ALU issue, not by HBM, as long as they call the
// general-purpose library routines (measured: 355 VALU instructions per particle in k_loc_stream, ~100 % VALU
// busy).  Every angle on this path is bounded (headings |theta| < 4 pi, Box-Muller angles in (0, 2 pi]), every
// logarithm argument lies in [2^-31, 1], so the routines below drop the large-argument, denormal and NaN paths.

// sin and cos of x, |x| <~ 100: three-step Cody-Waite reduction by pi/2 (fused multiply-adds on purpose: the
// reduction needs the exact product), then polynomials on [-pi/4, pi/4] (own least-squares fit,
// tools/fit_sincos.py).  Max error 7.1e-8 absolute against float64 over |x| <= 13; differs from glibc's
// sinf/cosf by at most 1 ulp (5.96e-8) on 4 % of arguments.
__device__ __forceinline__ void sincos_bounded(float x, float& s, float& c)
{
#ifdef PF_EXP_HW_SINCOS         // timing experiment: the hardware sine / cosine (argument in revolutions)
    const float rev = x * 0.15915494309189535f;
    s = __builtin_amdgcn_sinf(rev); c = __builtin_amdgcn_cosf(rev);
#else
    const float k = __builtin_rintf(x * 0.636619747f);              // nearest multiple of pi/2
    float r = __builtin_fmaf(k, -1.57079625129699707031e+00f, x);   // pi/2 = hi + mid + lo
    r = __builtin_fmaf(k, -7.54978941586159635335e-08f, r);
    r = __builtin_fmaf(k, -5.39030285815811905290e-15f, r);
    const float z = r * r;
    float ps = __builtin_fmaf(z, -1.9502914801705629e-04f, 8.332069963216782e-03f);
    ps = __builtin_fmaf(z, ps, -1.6666653752326965e-01f);
    ps = __builtin_fmaf(r, z * ps, r);
    float pc = __builtin_fmaf(z, 2.4421184207312763e-05f, -1.388721400871873e-03f);
    pc = __builtin_fmaf(z, pc, 4.166664183139801e-02f);
    pc = __builtin_fmaf(z, pc, -0.5f);
    pc = __builtin_fmaf(z, pc, 1.0f);
    const uint32_t q = (uint32_t)(int)k;
    const bool odd = (q & 1u) != 0u;
    const float so = odd ? pc : ps, co = odd ? ps : pc;
    s = __uint_as_float(__float_as_uint(so) ^ ((q & 2u) << 30));
    c = __uint_as_float(__float_as_uint(co) ^ (((q + 1u) & 2u) << 30));
#endif
}

// ln(u) for u in [2^-31, 1]: the hardware log2 (1 ulp) times ln 2 carried in two floats
__device__ __forceinline__ float log_unit(float u)
{
    const float y = __builtin_amdgcn_logf(u);
    const float h = y * 6.93147182464599609375e-01f;
    float l = __builtin_fmaf(y, 6.93147182464599609375e-01f, -h);
    l = __builtin_fmaf(y, -1.90465429995776804525e-09f, l);
    return h + l;
}

// pfRandom.c:67-74 -- Box-Muller pair from two words (u1 first):  f = sqrt(-2 ln u1),  (f cos 2 pi u2, f sin 2 pi u2).
// Everything here is a hardware transcendental: v_log_f32 (log2; measured on MI355X over u in (0,1]: 1.6e-7
// relative after the multiplication by ln 2), v_sqrt_f32 (1 ulp), and v_sin_f32 / v_cos_f32, which take their
// argument in REVOLUTIONS -- u2 itself, no multiplication by 2 pi and no range reduction; measured maximum absolute
// error 1.25e-7 over all 2^24 arguments k / 2^24 (tools/dev/hwsincos.hip).  The reference rounds g = u2 * 2 pi to
// float before taking sinf/cosf, which moves the angle by up to 3.7e-7 rad: the two ways agree to about 4e-7
// absolute in the unit normal, i.e. 4e-8 m after the 0.1 m jitter scale, 4e-10 m after the VIO noise scale.
__device__ __forceinline__ void normal2(uint32_t w0, uint32_t w1, float& a, float& b)
{
#ifdef PF_EXPERIMENT_NO_BOXMULLER   // tools/dev timing experiments only
    a = uniform_nonzero(w0) - 0.5f; b = uniform_nonzero(w1) - 0.5f; return;
#endif
    const float f = __builtin_amdgcn_sqrtf(__builtin_amdgcn_logf(uniform_nonzero(w0)) * -1.38629436111989061883f);   // -2 ln 2
    const float u2 = uniform_nonzero(w1);
    a = f * __builtin_amdgcn_cosf(u2);
    b = f * __builtin_amdgcn_sinf(u2);
}

// sqrtf(x), correctly rounded, for x = 0 or x in the normal range (squared distances in metres): the hardware square
// root (1 ulp) corrected by the two residual tests the compiler's own expansion uses, without that expansion's
// rescaling of denormal arguments and its class test -- nine instructions instead of seventeen.  Bit-identical to
// sqrtf for every argument that is not denormal, infinite or NaN.
__device__ __forceinline__ float sqrt_rn(float x)
{
    const float s = __builtin_amdgcn_sqrtf(x);
    const float sm = __uint_as_float(__float_as_uint(s) - 1u), sp = __uint_as_float(__float_as_uint(s) + 1u);
    const float rm = __builtin_fmaf(-sm, s, x);     // x - s_minus * s
    const float rp = __builtin_fmaf(-sp, s, x);     // x - s_plus * s
    float r = (rm <= 0.0f) ? sm : s;
    r = (rp > 0.0f) ? sp : r;
    return r;
}

// fmodf(x, 2*pi_f) with the result keeping x's sign (pfMeasurement.c:47, pfInit.c:78).  For
// |x| < 4*pi the quotient is 0 or +-1 and |x| - 2*pi is exact (Sterbenz), so no division is needed;
// anything larger (possible after a wide heading jitter) takes the library routine.
__device__ __forceinline__ float wrap_two_pi(float x)
{
    const float ax = fabsf(x);
    if (ax < kTwoPi) return x;
    if (ax < 2.0f * kTwoPi) return copysignf(ax - kTwoPi, x);
    return fmodf(x, kTwoPi);
}

// pfInit.c:56-66 + pfRandom.c:76-98 -- particle on the range shell around (cx,cy,cz), random heading.
// Words are consumed in the reference's draw order: radius tries (first non-negative of up to 10),
// elevation, azimuth, heading.
__device__ __forceinline__ void spawn_from_range(const RngKey& key, uint64_t index, float cx, float cy, float cz,
                                                 float range, float stdRange,
                                                 float& x, float& y, float& z, float& theta)
{
    u32x4 blk = rng_block(key, P_SPAWN, index, 0);
    uint32_t nblk = 1, pos = 0;
    auto next = [&]() -> uint32_t {
        if (pos == 4) { blk = rng_block(key, P_SPAWN, index, nblk++); pos = 0; }
        const uint32_t v = pos == 0 ? blk.a : pos == 1 ? blk.b : pos == 2 ? blk.c : blk.d;
        ++pos;
        return v;
    };
    float rad = 0.0f;
    for (int k = 0; k < 10; ++k) {
        const float t = range + 3.0f * stdRange * (uniform01(next()) * 2.0f - 1.0f);
        if (t < 0.0f) continue;
        rad = t;
        break;
    }
    const float elev = asinf(uniform01(next()) * 2.0f - 1.0f);
    const float azim = uniform01(next()) * 2.0f * kPi;
    float se, ce, sa, ca;
    sincos_bounded(elev, se, ce);
    sincos_bounded(azim, sa, ca);
    const float c = rad * ce;
    x = cx + c * ca;
    y = cy + c * sa;
    z = cz + rad * se;
    theta = uniform01(next()) * 2.0f * kPi;
}

// pfInit.c:68-79 -- jittered copy of an ancestor (w = 1 is written by the caller).  In two halves, the four unit
// normals (which depend on nothing but the counter) and their use, so that a kernel can have the first half done while
// it waits for the ancestor to arrive from memory.
__device__ __forceinline__ void jitter_normals(const RngKey& key, uint64_t index, float (&nj)[4])
{
    const u32x4 r = rng_block(key, P_JITTER, index, 0);
    normal2(r.a, r.b, nj[0], nj[1]);
    normal2(r.c, r.d, nj[2], nj[3]);
}
__device__ __forceinline__ void spawn_from_other_n(const float (&nj)[4], float ox, float oy, float oz, float oth,
                                                   float hXyz, float hTheta, float& x, float& y, float& z, float& theta)
{
    x = ox + nj[0] * hXyz;
    y = oy + nj[1] * hXyz;
    z = oz + nj[2] * hXyz;
    theta = wrap_two_pi(oth + nj[3] * hTheta);
}
__device__ __forceinline__ void spawn_from_other(const RngKey& key, uint64_t index, float ox, float oy, float oz, float oth,
                                                 float hXyz, float hTheta, float& x, float& y, float& z, float& theta)
{
    float nj[4];
    jitter_normals(key, index, nj);
    spawn_from_other_n(nj, ox, oy, oz, oth, hXyz, hTheta, x, y, z, theta);
}

// pfMeasurement.c:36-47 (Loc), :61-74 (SLAM tag), :90-104 (SLAM beacon) -- VIO motion of one particle: rotate the
// delta by the heading, add Gaussian noise, wrap the heading.  Returns sin/cos of the NEW heading (the weight
// sums need them).  The heading noise is stdTheta * n with |n| < 6.6: while that stays below 1e-3 rad
// (smallTurn, decided once per launch from stdTheta <= 1e-4) the sine and cosine of the heading BEFORE the noise
// follow from those after it by the angle-difference formulas to second order (error < 5e-11), which saves one
// sincos per particle in kernels bound by vector-ALU issue.  Also in two halves (the unit normals, their use).
struct Motion { float dx, dy, dz, stdXyz, stdTheta; };

__device__ __forceinline__ void vio_normals(const RngKey& key, uint64_t index, float (&nv)[4])
{
    const u32x4 r = rng_block(key, P_VIO, index, 0);
    normal2(r.a, r.b, nv[0], nv[1]);
    normal2(r.c, r.d, nv[2], nv[3]);
}
__device__ __forceinline__ void motion_particle_n(const float (&nv)[4], const Motion& mo, bool smallTurn,
                                                  float& x, float& y, float& z, float& th, float& sn, float& cs)
{
    const float d = mo.stdTheta * nv[3];
    float s0, c0;
    if (smallTurn) {
        th = wrap_two_pi(th + d);
        sincos_bounded(th, sn, cs);
        const float h = 0.5f * d * d;
        c0 = __builtin_fmaf(d, sn, __builtin_fmaf(-h, cs, cs));     // cos(theta - d)
        s0 = __builtin_fmaf(-d, cs, __builtin_fmaf(-h, sn, sn));    // sin(theta - d)
    } else {
        // (the empty volatile statement keeps this a branch: without it the compiler evaluates this side's extra sincos
        // for every particle and selects afterwards -- 25 instructions per particle in kernels bound by vector-ALU issue)
        asm volatile("; heading noise too large for the angle-difference form");
        sincos_bounded(th, s0, c0);
        th = wrap_two_pi(th + d);
        sincos_bounded(th, sn, cs);
    }
    const float pDx = mo.dx * c0 - mo.dy * s0;
    const float pDy = mo.dx * s0 + mo.dy * c0;
    x += pDx + mo.stdXyz * nv[0];
    y += pDy + mo.stdXyz * nv[1];
    z += mo.dz + mo.stdXyz * nv[2];
}
__device__ __forceinline__ void motion_particle(const RngKey& key, uint64_t index, const Motion& mo, bool smallTurn,
                                                float& x, float& y, float& z, float& th, float& sn, float& cs)
{
    float nv[4];
    vio_normals(key, index, nv);
    motion_particle_n(nv, mo, smallTurn, x, y, z, th, sn, cs);
}

// ---------------------------------------------------------------- streaming loads and stores
// NT = non-temporal form (global_load / global_store ... nt).  Every particle array is read once and written once per
// pass and is larger than every cache; which accesses gain from the hint is a measured choice per site (PF_NT_* below,
// pf_loc_kernels.h).
typedef float pf_v4f __attribute__((ext_vector_type(4)));
template <bool NT>
__device__ __forceinline__ float4 ld4(const float* p)
{
    if (NT) {
        const pf_v4f v = __builtin_nontemporal_load(reinterpret_cast<const pf_v4f*>(p));
        return make_float4(v.x, v.y, v.z, v.w);
    }
    return *reinterpret_cast<const float4*>(p);
}
template <bool NT>
__device__ __forceinline__ float ld1(const float* p)
{
    if (NT) return __builtin_nontemporal_load(p);
    return *p;
}
template <bool NT>
__device__ __forceinline__ void st4(float* p, const float4& v)
{
    if (NT) {
        pf_v4f t; t.x = v.x; t.y = v.y; t.z = v.z; t.w = v.w;
        __builtin_nontemporal_store(t, reinterpret_cast<pf_v4f*>(p));
    } else {
        *reinterpret_cast<float4*>(p) = v;
    }
}

// ---------------------------------------------------------------- wave / block reductions (wave = 64)
__device__ __forceinline__ double wave_sum(double v)
{
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) v += __shfl_xor(v, off, 64);
    return v;
}

// Sum K doubles per thread over a 256-thread block; result valid in thread 0.  `lds` holds 4*K doubles.
template <int K>
__device__ __forceinline__ void block_sum(double (&v)[K], double* lds)
{
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
#pragma unroll
    for (int k = 0; k < K; ++k) v[k] = wave_sum(v[k]);
    if (lane == 0) {
#pragma unroll
        for (int k = 0; k < K; ++k) lds[wave * K + k] = v[k];
    }
    __syncthreads();
    if (threadIdx.x == 0) {
#pragma unroll
        for (int k = 0; k < K; ++k) v[k] = (lds[k] + lds[K + k]) + (lds[2 * K + k] + lds[3 * K + k]);
    }
}

// Sum K doubles per thread over a 256-thread block and store the K totals to out[0..K) (global memory), going
// through LDS instead of cross-lane shuffles.  A wave-shuffle tree costs 12 ds_bpermute per double and wave
// (shuffles of 64-bit values are LDS-crossbar instructions on gfx950): with K = 7 that was 336 LDS instructions per
// workgroup and made the LDS pipeline, not HBM, the bound of the streaming kernels (measured: 0.078 -> 0.058 ms at
// N = 16 M without it).  Here: every thread parks its K values (K conflict-free ds_write_b64), K*32 threads add eight
// of them each, K threads add the 32 partials -- 2 barriers, ~70 LDS instructions per workgroup, fixed order.
// `lds` holds K*256 doubles.
template <int K>
__device__ __forceinline__ void block_sum_store(const double (&v)[K], double* lds, double* __restrict__ out)
{
    const int tid = threadIdx.x;
#pragma unroll
    for (int k = 0; k < K; ++k) lds[k * kThreads + tid] = v[k];
    __syncthreads();
    if (tid < K * 32) {
        const int k = tid >> 5, j = tid & 31;
        const double* p = lds + k * kThreads + j;
        double s = p[0];
#pragma unroll
        for (int i = 1; i < 8; ++i) s += p[32 * i];
        lds[k * kThreads + j] = s;
    }
    __syncthreads();
    if (tid < K) {
        const double* p = lds + tid * kThreads;
        double s = p[0];
#pragma unroll
        for (int j = 1; j < 32; ++j) s += p[j];
        out[tid] = s;
    }
}

// The same result with a quarter of the LDS, for kernels whose occupancy hangs on it: the four lanes of a quad are
// added first with two DPP quad permutes per double (vector-ALU moves, not LDS-crossbar shuffles), one lane per
// quad parks the sum, K*8 threads add eight of them each, K threads add the eight partials.  Must be reached by
// all 256 threads.  `lds` holds K*64 doubles.
template <int CTRL>
__device__ __forceinline__ double quad_perm_f64(double v)
{
    const int lo = __builtin_amdgcn_update_dpp(0, __double2loint(v), CTRL, 0xf, 0xf, true);
    const int hi = __builtin_amdgcn_update_dpp(0, __double2hiint(v), CTRL, 0xf, 0xf, true);
    return __hiloint2double(hi, lo);
}

template <int K>
__device__ __forceinline__ double block_sum_quads(const double (&v)[K], double* lds)
{   // returns total k in thread k (k < K), 0 elsewhere
    const int tid = threadIdx.x;
    __syncthreads();                        // a previous call's last reads of lds
#pragma unroll
    for (int k = 0; k < K; ++k) {
        double s = v[k];
        s += quad_perm_f64<0xB1>(s);        // lanes 1 0 3 2
        s += quad_perm_f64<0x4E>(s);        // lanes 2 3 0 1
        if ((tid & 3) == 0) lds[k * 64 + (tid >> 2)] = s;
    }
    __syncthreads();
    if (tid < K * 8) {
        const int k = tid >> 3, j = tid & 7;
        const double* p = lds + k * 64 + j;
        double s = p[0];
#pragma unroll
        for (int i = 1; i < 8; ++i) s += p[8 * i];
        lds[k * 64 + j] = s;
    }
    __syncthreads();
    double r = 0.0;
    if (tid < K) {
        const double* p = lds + tid * 64;
        r = p[0];
#pragma unroll
        for (int j = 1; j < 8; ++j) r += p[j];
    }
    return r;
}

// block_sum_quads for the LAST thing a workgroup does, on LDS nothing else has touched since the last barrier: the same
// additions in the same order, but one barrier instead of three -- wavefronts 1 to 3 are done once their partial sums are
// parked, wavefront 0 adds them up alone (a single wavefront runs in lockstep: its own LDS writes are ordered before its
// reads by the wait on the LDS counter, no barrier needed).
template <int K>
__device__ __forceinline__ double block_sum_quads_tail(const double (&v)[K], double* lds)
{
    const int tid = threadIdx.x;
#pragma unroll
    for (int k = 0; k < K; ++k) {
        double s = v[k];
        s += quad_perm_f64<0xB1>(s);
        s += quad_perm_f64<0x4E>(s);
        if ((tid & 3) == 0) lds[k * 64 + (tid >> 2)] = s;
    }
    __syncthreads();
    double r = 0.0;
    if (tid < 64) {
        if (tid < K * 8) {
            const int k = tid >> 3, j = tid & 7;
            const double* p = lds + k * 64 + j;
            double s = p[0];
#pragma unroll
            for (int i = 1; i < 8; ++i) s += p[8 * i];
            lds[k * 64 + j] = s;
        }
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
        __builtin_amdgcn_wave_barrier();
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
        if (tid < K) {
            const double* p = lds + tid * 64;
            r = p[0];
#pragma unroll
            for (int j = 1; j < 8; ++j) r += p[j];
        }
    }
    return r;
}

// Inclusive scan over the 64 lanes of a wavefront in six DPP steps, no LDS: row_shr 1, 2, 4, 8 inside each row of 16
// lanes (a lane without a source adds the identity), then the last lane of rows 0 and 2 broadcast into rows 1 and 3
// (row_bcast:15) and lane 31 into the upper half (row_bcast:31).  The sums it is used for are exact (integers, or
// float64 sums of multiples of 2^-20), so the order of the additions does not show.
template <int CTRL, int ROW_MASK>
__device__ __forceinline__ uint32_t dpp_or_zero(uint32_t v)
{
    return (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, CTRL, ROW_MASK, 0xf, false);
}
template <int CTRL, int ROW_MASK>
__device__ __forceinline__ double dpp_or_zero(double v)
{
    const int lo = __builtin_amdgcn_update_dpp(0, __double2loint(v), CTRL, ROW_MASK, 0xf, false);
    const int hi = __builtin_amdgcn_update_dpp(0, __double2hiint(v), CTRL, ROW_MASK, 0xf, false);
    return __hiloint2double(hi, lo);
}
template <typename T>
__device__ __forceinline__ T wave_inclusive_scan(T v)
{
    v += dpp_or_zero<0x111, 0xf>(v);        // row_shr:1
    v += dpp_or_zero<0x112, 0xf>(v);        // row_shr:2
    v += dpp_or_zero<0x114, 0xf>(v);        // row_shr:4
    v += dpp_or_zero<0x118, 0xf>(v);        // row_shr:8
    v += dpp_or_zero<0x142, 0xa>(v);        // row_bcast:15 -> rows 1 and 3
    v += dpp_or_zero<0x143, 0xc>(v);        // row_bcast:31 -> rows 2 and 3
    return v;
}

// Inclusive running maximum over the 64 lanes of a wavefront, same six steps, for values that are not negative (a lane
// without a source compares against zero).
__device__ __forceinline__ double wave_inclusive_max_nonneg(double v)
{
    double u;
    u = dpp_or_zero<0x111, 0xf>(v); v = u > v ? u : v;
    u = dpp_or_zero<0x112, 0xf>(v); v = u > v ? u : v;
    u = dpp_or_zero<0x114, 0xf>(v); v = u > v ? u : v;
    u = dpp_or_zero<0x118, 0xf>(v); v = u > v ? u : v;
    u = dpp_or_zero<0x142, 0xa>(v); v = u > v ? u : v;
    u = dpp_or_zero<0x143, 0xc>(v); v = u > v ? u : v;
    return v;
}

// Inclusive scan of one double per thread across a 256-thread block.  `lds` holds 4 doubles.
__device__ __forceinline__ double block_inclusive_scan(double v, double* lds, double& block_total)
{
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    v = wave_inclusive_scan(v);
    if (lane == 63) lds[wave] = v;
    __syncthreads();
    const double w0 = lds[0], w1 = lds[1], w2 = lds[2], w3 = lds[3];
    double pre = 0.0;
    if (wave > 0) pre = w0;
    if (wave > 1) pre += w1;
    if (wave > 2) pre += w2;
    block_total = ((w0 + w1) + w2) + w3;
    return v + pre;
}

// Inclusive running maximum of one double per thread across a 256-thread block (exact: max is
// associative).  Used to make a parallel prefix sum of weights, which are not negative, non-decreasing.  `lds` holds
// 4 doubles.
__device__ __forceinline__ double block_inclusive_max(double v, double* lds)
{
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    v = wave_inclusive_max_nonneg(v);
    if (lane == 63) lds[wave] = v;
    __syncthreads();
    for (int k = 0; k < wave; ++k) v = lds[k] > v ? lds[k] : v;
    return v;
}

// first j in [0, n) with u < cdf[j]; n-1 if there is none (pfResample.c:70-72 with the clamp)
template <typename T>
__device__ __forceinline__ int upper_bound_clamped(const T* cdf, int n, T u)
{
    int lo = 0, hi = n - 1;             // answer lies in [lo, hi]
    while (lo < hi) {
        const int mid = (lo + hi) >> 1;
        if (u < cdf[mid]) hi = mid; else lo = mid + 1;
    }
    return lo;
}

// The same selection over a table of exactly kTile entries in LDS whose entries from n on hold +inf:
// ten unrolled, branch-free halving steps (first j with u < cdf[j], clamped to n-1; NaNs end at n-1 as above).
__device__ __forceinline__ int upper_bound_tile(const double* cdf, int n, double u)
{
    int pos = 0;
#pragma unroll
    for (int step = kTile / 2; step >= 1; step >>= 1) pos += (u < cdf[pos + step - 1]) ? 0 : step;
    return pos < n ? pos : n - 1;
}

// ---------------------------------------------------------------- the same selection on 32-bit offsets
// Inside a tile whose weight total is below kTileU32Below the CDF is held as integer offsets from the tile's start
// value T, in units of 2^-20 (exact: every CDF value is a multiple of 2^-20).  For an offspring position u >= T,
// u - T is exact in float64 and  u < CDF(j)  <=>  floor((u - T) * 2^20) < offset(j),  so the integer search below
// returns exactly what upper_bound_tile returns on the float64 table -- at full-rate 32-bit compares and half the
// LDS bytes.
__device__ __forceinline__ uint32_t cvt_u32_sat(double v)
{   // v_cvt_u32_f64: truncates, saturates to [0, 2^32 - 1], NaN -> 0 (a C++ cast is undefined outside the range)
    uint32_t r;
    asm("v_cvt_u32_f64 %0, %1" : "=v"(r) : "v"(v));
    return r;
}
__device__ __forceinline__ uint32_t cvt_u32_sat(float v)
{
    uint32_t r;
    asm("v_cvt_u32_f32 %0, %1" : "=v"(r) : "v"(v));
    return r;
}

__device__ __forceinline__ int upper_bound_tile_u32(const uint32_t* tab, int n, uint32_t d)
{   // entries from n on hold 0xFFFFFFFF.  Five levels of three independent probes instead of ten dependent ones: the
    // search sits on the chain of waits in front of the ancestor loads, where the depth of dependent LDS reads counts
    // for more than their number (measured: -0.7 % on the fused launch).  The result is the number of entries <= d.
    int pos = 0;
#pragma unroll
    for (int step = kTile / 4; step >= 1; step >>= 2) {
        const uint32_t a = tab[pos + step - 1], b = tab[pos + 2 * step - 1], c = tab[pos + 3 * step - 1];
        pos += ((d < a) ? 0 : step) + ((d < b) ? 0 : step) + ((d < c) ? 0 : step);
    }
    return pos < n ? pos : n - 1;
}

// Inclusive scan of one 32-bit integer per thread across a 256-thread block (exact, any order).  `lds` holds 4 words.
__device__ __forceinline__ uint32_t block_inclusive_scan_u32(uint32_t v, uint32_t* lds)
{
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    v = wave_inclusive_scan(v);
    if (lane == 63) lds[wave] = v;
    __syncthreads();
    uint32_t pre = 0;
    if (wave > 0) pre = lds[0];
    if (wave > 1) pre += lds[1];
    if (wave > 2) pre += lds[2];
    return v + pre;
}

// One tile's CDF in LDS, in whichever of the two forms its total allows.  `tab` holds kTile doubles (the integer
// form uses the first half of it), `red` 4 doubles.  w: the thread's four weights (elements 4 * tid .. + 3 of the
// tile), tileStart / tileEnd: CDF value before the tile's first / after its last particle, nValid: particles of the
// tile that exist.  Returns true (uniformly over the workgroup) if the integer form was built.  Ends with a barrier.
__device__ __forceinline__ bool build_tile_cdf(const float (&w)[4], double tileStart, double tileEnd, double grandTotal,
                                               int nValid, double* tab, double* red)
{
    const int tid = threadIdx.x;
    const bool small = (tileEnd - tileStart < kTileU32Below) && (grandTotal < kCdfExactBelow);     // NaN: false
    if (small) {
        uint32_t* const tab32 = reinterpret_cast<uint32_t*>(tab);
        const uint32_t a0 = cvt_u32_sat(cdf_units(w[0])), a1 = a0 + cvt_u32_sat(cdf_units(w[1]));
        const uint32_t a2 = a1 + cvt_u32_sat(cdf_units(w[2])), a3 = a2 + cvt_u32_sat(cdf_units(w[3]));
        const uint32_t off = block_inclusive_scan_u32(a3, reinterpret_cast<uint32_t*>(red)) - a3;
        uint4 o;
        o.x = (tid * 4 + 0 < nValid) ? off + a0 : 0xFFFFFFFFu;
        o.y = (tid * 4 + 1 < nValid) ? off + a1 : 0xFFFFFFFFu;
        o.z = (tid * 4 + 2 < nValid) ? off + a2 : 0xFFFFFFFFu;
        o.w = (tid * 4 + 3 < nValid) ? off + a3 : 0xFFFFFFFFu;
        *reinterpret_cast<uint4*>(tab32 + tid * 4) = o;
    } else {
        // pfResample.c:45,49 in double: tile start + inclusive scan
        const double a0 = cdf_weight(w[0]), a1 = a0 + cdf_weight(w[1]), a2 = a1 + cdf_weight(w[2]), a3 = a2 + cdf_weight(w[3]);
        double total;
        const double incl = block_inclusive_scan(a3, red, total);
        const double off = tileStart + (incl - a3);
        const double inf = __builtin_huge_val();            // pad entries of the last tile: never selected
        tab[tid * 4 + 0] = (tid * 4 + 0 < nValid) ? off + a0 : inf;
        tab[tid * 4 + 1] = (tid * 4 + 1 < nValid) ? off + a1 : inf;
        tab[tid * 4 + 2] = (tid * 4 + 2 < nValid) ? off + a2 : inf;
        tab[tid * 4 + 3] = (tid * 4 + 3 < nValid) ? off + a3 : inf;
    }
    __syncthreads();
    return small;
}

// The same selection for an offspring whose predecessor (the offspring before it, same table) was given ancestor
// `prev`: positions do not decrease, so the answer is the first j >= prev with d < tab[j] -- three probes forward, and
// the halving search only if that was not enough (a run of weightless particles in between).  Same result as
// upper_bound_tile_u32 by construction.  The table needs four more entries of 0xFFFFFFFF behind its kTile.
__device__ __forceinline__ int upper_bound_tile_u32_from(const uint32_t* tab, int n, uint32_t d, int prev)
{
    int p = prev;
    if (!(d < tab[p])) {
        ++p;
        if (!(d < tab[p])) {
            ++p;
            if (!(d < tab[p])) return upper_bound_tile_u32(tab, n, d);
        }
    }
    return p < n ? p : n - 1;
}

// ancestor of the offspring at CDF position u inside the tile whose table build_tile_cdf left in `tab`
__device__ __forceinline__ int tile_ancestor(const double* tab, bool small, int nValid, double tileStart, double u)
{
    if (small) return upper_bound_tile_u32(reinterpret_cast<const uint32_t*>(tab), nValid, cvt_u32_sat((u - tileStart) * (double)kCdfScale));
    return upper_bound_tile(tab, nValid, u);
}

}  // namespace pf

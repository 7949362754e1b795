// pf_slam_kernels.h -- HIP kernels of the particleFilterSlam hot path (gfx950, wave64).
//
// The reference's SLAM filter is Rao-Blackwellised (particleFilter.h:66-100): nTag (100) tag particles, and per
// beacon one ROW of nBcn (1000) beacon particles for every tag particle.  A row is an independent little particle
// filter (pfResample.c:141-201), so the unit of work here is ONE WORKGROUP PER ROW: 256 threads x 4 particles, the
// whole row in registers, its CDF and ancestor copies in LDS, resampled in place (no ping-pong buffer).
// Many SLAM filters ("a batch", BASELINE.json configs[4]) are one grid dimension more; filters never interact.
//
// One deposit (particleFilter.c:193-211) is three launches for the whole batch, no host sync:
//   k_slam_gate_rows grid (nTag, F): the measured beacon's row i in ONE pass over its 20 B/particle -- beacon VIO motion
//                 (pfMeasurement.c:78-106), range gate against tag particle i AFTER its own VIO motion (:127-152; the
//                 tag motion is recomputed from the same Philox counters, not yet stored), row weight sum, and the row's
//                 own resample-or-normalise step (_resampleBcn, pfResample.c:141-201) under the assumption that the tag
//                 particles will NOT be resampled (the common case).  A resampled row is final; a normalised row keeps
//                 its gated weights and only records the factor N/s (rowScale, multiplied in by the next reader), so
//                 that the rare forced pass below still finds the weights the reference would resample from.
//   k_slam_tag    grid (F): tag VIO motion (:51-76), w_i *= rowsum_i, ESS test, resample or normalise the tag
//                 particles (pfResample.c:86-139).                                               [tiny]
//   k_slam_rows   grid (nTag, B, F): what is left -- VIO motion of the other beacons, the first-range spawn
//                 (pfInit.c:34-45), and, only if the tag particles WERE resampled, the forced _resampleBcn of every
//                 initialised beacon (pfResample.c:122-130); rows of the measured beacon that the first kernel
//                 already resampled would come out identical (same weights, same Philox counters) and are only
//                 re-spawned around the NEW tag particle where the respawn rule fired.  Idle rows exit at once.
// Noise: Philox sub-streams keyed (seed of the filter, stream = 0 for the tag / slot+1 for a beacon, epoch = deposit
// count) and indexed by the particle's position in its array, as the oracle's Philox mode (oracle/pf_oracle.c).
#pragma once

#include "pf_common.h"

namespace pf {

constexpr int kSlamMaxBcns = 32;        // beacon slots per filter (bit masks below)

struct SlamStep {           // one filter's part of a (batched) deposit
    int mode;               // 0 no measurement, 1 range on an initialised beacon, 2 first range: spawn the beacon
    int target;             // beacon slot of the measurement
    uint32_t seed, epoch;
    uint32_t activeMask;    // beacons handed to the deposit (allBcns[0..numBcns), particleFilter.c:198-199)
    uint32_t initMask;      // of those, initialised before this deposit
    uint32_t movingMask;    // of those, with a non-zero VIO delta to commit
    int tagMoving;
    Motion tagMo;
    float range, stdRange, thresh, minWeight;
};

struct SlamDims { int numFilters, maxBcns, nTag, nBcn, tagStride, rowStride; };
struct SlamPtrs { float* tag[5]; float* bcn[5]; };      // w x y z theta

__host__ __device__ __forceinline__ size_t slam_row_base(const SlamDims& D, int f, int c, int k)
{
    return (((size_t)f * D.maxBcns + c) * D.nTag + k) * (size_t)D.rowStride;
}

struct Row4 { float w[4], x[4], y[4], z[4], th[4]; };

// non-temporal hint on the measured beacon's row loads / stores (k_slam_gate_rows_*), by measurement
#ifndef PF_SLAM_NT_LD
#define PF_SLAM_NT_LD 1
#endif
#ifndef PF_SLAM_NT_LDW
#define PF_SLAM_NT_LDW PF_SLAM_NT_LD
#endif
#ifndef PF_SLAM_NT_ST
#define PF_SLAM_NT_ST 1
#endif

__device__ __forceinline__ void row_load(const float* const (&a)[5], size_t base, int j0, int n, Row4& r)
{
    // unconditional loads from a clamped index (threads past the row's end read its head and drop it): loads behind a
    // per-array branch were issued one after the other, each waited for before the next (four round trips instead of one)
    const int jl = j0 < n ? j0 : 0;
    float4 v[5];
#pragma unroll
    for (int q = 0; q < 5; ++q) v[q] = *reinterpret_cast<const float4*>(a[q] + base + jl);
#pragma unroll
    for (int q = 0; q < 5; ++q)
        if (j0 >= n) v[q] = make_float4(0, 0, 0, 0);
    r.w[0] = v[0].x; r.w[1] = v[0].y; r.w[2] = v[0].z; r.w[3] = v[0].w;
    r.x[0] = v[1].x; r.x[1] = v[1].y; r.x[2] = v[1].z; r.x[3] = v[1].w;
    r.y[0] = v[2].x; r.y[1] = v[2].y; r.y[2] = v[2].z; r.y[3] = v[2].w;
    r.z[0] = v[3].x; r.z[1] = v[3].y; r.z[2] = v[3].z; r.z[3] = v[3].w;
    r.th[0] = v[4].x; r.th[1] = v[4].y; r.th[2] = v[4].z; r.th[3] = v[4].w;
}

__device__ __forceinline__ void row_store1(float* a, size_t base, int j0, int n, const float (&v)[4])
{
    // rows are padded to a multiple of four floats, so a whole float4 may be written when j0 < n
    if (j0 < n) *reinterpret_cast<float4*>(a + base + j0) = make_float4(v[0], v[1], v[2], v[3]);
}
__device__ __forceinline__ void row_store1_gate(float* a, size_t base, int j0, int n, const float (&v)[4])
{
    if (j0 < n) st4<PF_SLAM_NT_ST>(a + base + j0, make_float4(v[0], v[1], v[2], v[3]));
}

// The scalar part of one small filter's resample-or-normalise decision (pfResample.c:107-121 for the tag
// particles, :165-184 for a beacon row) on the totals S = sum w (f64), SS, C, SN.
struct SmallDecision { int resample, numSpawn; float htheta, m; double rStart, rStep; };

// Part 1: the ESS / respawn test (needs only S and SS).  Part 2, for a filter that resamples: heading bandwidth from
// the circular sums, rStep, rStart.  small_decide is the two in a row.
__device__ __forceinline__ SmallDecision small_decide_test(double S, double SS, int n, float range, bool allowSpawn, bool force)
{
    SmallDecision d;
    const float s = (float)S, ss = (float)SS;
    const float ess = s * s / ss;
    const float invN = 1.0f / (float)n;
    d.numSpawn = 0;
    if (allowSpawn && s * invN < kWeightSpawnThresh && range < kRadiusSpawnThresh) d.numSpawn = (int)lroundf((float)n * kPctSpawn);
    d.resample = ((ess * invN < kResampleThresh) || d.numSpawn > 0 || force) ? 1 : 0;
    d.htheta = 0.0f; d.m = 0.0f; d.rStart = 0.0; d.rStep = 0.0;
    if (!d.resample) d.m = (float)n / s;
    return d;
}

__device__ __forceinline__ void small_decide_spread(SmallDecision& d, double S, double SS, double C, double SN, int n,
                                                    const RngKey& key, uint64_t rsIndex)
{
    const float s = (float)S, ss = (float)SS;
    const float ess = s * s / ss;
    float csum = (float)C, ssum = (float)SN;
    csum /= s;
    ssum /= s;
    float h = csum * csum + ssum * ssum;
    h = h > 1e-10f ? h : 1e-10f;
    h = h < 1.0f - 1e-10f ? h : 1.0f - 1e-10f;
    d.htheta = sqrtf(-logf(h) / ess);
    d.rStep = S / (double)n;
    d.rStart = (double)uniform01(rng_block(key, P_RSTART, rsIndex, 0).a) * d.rStep;
}

__device__ __forceinline__ SmallDecision small_decide(double S, double SS, double C, double SN, int n, float range,
                                                      bool allowSpawn, bool force, const RngKey& key, uint64_t rsIndex)
{
    SmallDecision d = small_decide_test(S, SS, n, range, allowSpawn, force);
    if (d.resample) small_decide_spread(d, S, SS, C, SN, n, key, rsIndex);
    return d;
}

// Shared-memory working set of one small filter (a tag particle set or one beacon row)
struct SmallLds {
    double cdf[kTile];          // parking space of the block reductions, then the CDF table of a resampling filter (32-bit
                                // integer offsets in its first half, or float64 values: build_tile_cdf)
    float4 anc[kTile];          // x y z theta of the filter's particles, staged for the ancestor look-up
    double red[4 * 4];
    double scan[4];
    SmallDecision dec;
};

// The resampling of one small filter whose decision `d` says so (pfResample.c:115-119 / :178-192), the Loc kernels' way:
// the CDF goes to LDS as 32-bit integer offsets in units of 2^-20 (exact: the weights enter as multiples of 2^-20 and a
// filter of <= 1024 particles of mean weight ~1 stays far below the 4095 the form holds; heavier rows take the float64
// table -- both are the CDF rule of DESIGN.md 4.1, not approximations of it), built with one integer block scan in DPP
// steps; the ancestors wait in LDS as one float4 each; a thread's four consecutive offspring search from the previous
// one's ancestor on (three probes, then five levels of three independent probes) instead of ten dependent halving steps
// over float64; and the unit normals of the thread's first offspring are drawn before the table is looked at.  Same
// ancestors, same offspring, bit for bit, as the float64 table and halving search of rounds 1-2.
// r: the filter's particles, four per thread (weights as they enter the CDF); on return the offspring, w = 1.
__device__ __forceinline__ void small_resample(Row4& r, int j0, int n, SmallLds& L, const SmallDecision& d, double S,
                                               const RngKey& key, uint64_t index0, float range, float stdRange,
                                               float cx, float cy, float cz)
{
    float nj0[4] = { 0.f, 0.f, 0.f, 0.f };
    const bool first = j0 < n && j0 >= d.numSpawn;
    if (first) jitter_normals(key, index0 + (uint64_t)j0, nj0);
    float wq[4];
#pragma unroll
    for (int k = 0; k < 4; ++k) {
        wq[k] = (j0 + k < n) ? r.w[k] : 0.0f;
        L.anc[j0 + k] = make_float4(r.x[k], r.y[k], r.z[k], r.th[k]);
    }
    const bool small = build_tile_cdf(wq, 0.0, S, S, n, L.cdf, L.scan);     // (ends with a barrier: the ancestors are in place too)
    const uint32_t* const tab32 = reinterpret_cast<const uint32_t*>(L.cdf);
    int prevJ = -1;
#pragma unroll
    for (int k = 0; k < 4; ++k) {
        const int i = j0 + k;
        if (i >= n) continue;
        if (i < d.numSpawn) {
            spawn_from_range(key, index0 + (uint64_t)i, cx, cy, cz, range, stdRange, r.x[k], r.y[k], r.z[k], r.th[k]);
        } else {
            const double u = d.rStart + d.rStep * (double)i;            // pfResample.c:117 / :180
            int j;
            if (small) {
                const uint32_t q = cvt_u32_sat(u * (double)kCdfScale);
                j = prevJ >= 0 ? upper_bound_tile_u32_from(tab32, n, q, prevJ) : upper_bound_tile_u32(tab32, n, q);
            } else {
                j = upper_bound_tile(L.cdf, n, u);
            }
            prevJ = j;
            const float4 a = L.anc[j];
            if (k == 0 && first) spawn_from_other_n(nj0, a.x, a.y, a.z, a.w, kHxyz, d.htheta, r.x[k], r.y[k], r.z[k], r.th[k]);
            else spawn_from_other(key, index0 + (uint64_t)i, a.x, a.y, a.z, a.w, kHxyz, d.htheta, r.x[k], r.y[k], r.z[k], r.th[k]);
        }
        r.w[k] = 1.0f;
    }
}

// Weight pass + decision + in-register resample/normalise of one small filter of n <= 1024 particles held four
// per thread in `r` (sn/cs = sin/cos of their headings).  (cx,cy,cz) is the respawn centre; index0 the Philox
// index of particle 0.  Returns the decision (valid in every thread).
//
// Two phases, because most rows only normalise: (1) the four sums (pfResample.c:93-106 / :151-164: float products,
// added in double) go through ONE block reduction in LDS (the CDF area serves as parking space) and thread 0
// decides; (2) only a resampling filter builds its CDF (block scan), stages the ancestors in LDS and searches.
// rowSumOut (optional): receives (float) sum w, the factor pfMeasurement.c:148-150 puts on the tag particle.
__device__ __forceinline__ SmallDecision small_filter_update(Row4& r, const float (&sn)[4], const float (&cs)[4], int j0, int n,
                                                             SmallLds& L, const RngKey& key, uint64_t index0, uint64_t rsIndex,
                                                             float range, float stdRange, bool allowSpawn, bool force,
                                                             float cx, float cy, float cz, bool lazyNorm = false,
                                                             float* __restrict__ rowSumOut = nullptr)
{
    const int tid = threadIdx.x;
    double acc[4] = { 0, 0, 0, 0 };     // s, ss, c, sn
    double run = 0.0;
#pragma unroll
    for (int k = 0; k < 4; ++k) {
        const float wi = (j0 + k < n) ? r.w[k] : 0.0f;
        run += cdf_weight(wi);          // exact sums (pf_common.h, CDF rule)
        acc[1] += wi * wi;
        acc[2] += wi * cs[k];
        acc[3] += wi * sn[k];
    }
    acc[0] = run;
    double* const park = L.cdf;         // 4 * kThreads doubles
#pragma unroll
    for (int k = 0; k < 4; ++k) park[k * kThreads + tid] = acc[k];
    __syncthreads();
    if (tid < 4 * 32) {
        const int k = tid >> 5, j = tid & 31;
        const double* p = park + k * kThreads + j;
        double t = p[0];
#pragma unroll
        for (int i = 1; i < 8; ++i) t += p[32 * i];
        park[k * kThreads + j] = t;
    }
    __syncthreads();
    if (tid < 4) {
        const double* p = park + tid * kThreads;
        double t = p[0];
#pragma unroll
        for (int j = 1; j < 32; ++j) t += p[j];
        L.red[tid] = t;
    }
    __syncthreads();
    if (tid == 0) {
        if (rowSumOut) *rowSumOut = (float)L.red[0];
        L.dec = small_decide(L.red[0], L.red[1], L.red[2], L.red[3], n, range, allowSpawn, force, key, rsIndex);
    }
    __syncthreads();
    const SmallDecision d = L.dec;
    if (d.resample) {                   // uniform over the workgroup
        small_resample(r, j0, n, L, d, L.red[0], key, index0, range, stdRange, cx, cy, cz);
    } else if (!lazyNorm) {
#pragma unroll
        for (int k = 0; k < 4; ++k)
            if (j0 + k < n) r.w[k] *= d.m;              // pfResample.c:133-137 / :195-199
    }
    return d;
}

// The same for a row whose headings have not been read: a row that only normalises never needs them (the circular sums
// feed the heading bandwidth of a resampling, pfResample.c:171-176, nothing else), which saves it a fifth of its reads
// and a sincos per particle.  Phase 1 reduces sum w and sum w^2 and tests; a resampling row then loads its headings
// (thSrc + j0, one float4 per thread), reduces the circular sums, and goes on as small_filter_update does.
// The headings are left unwrapped (pfMeasurement.c:104 with zero motion is theta = fmodf(theta, 2 pi): a no-op for every
// stored heading except an initial one of exactly 2 pi_f, probability 2^-31 per spawned particle, which then keeps that
// value -- equal modulo 2 pi -- until the row next resamples).
__device__ __forceinline__ SmallDecision small_filter_update_lazy_theta(Row4& r, const float* __restrict__ thSrc, int j0, int n,
                                                                        SmallLds& L, const RngKey& key, uint64_t index0,
                                                                        uint64_t rsIndex, float range, float stdRange,
                                                                        float cx, float cy, float cz,
                                                                        float* __restrict__ rowSumOut)
{
    const int tid = threadIdx.x;
    double acc[2] = { 0, 0 };           // s, ss
#pragma unroll
    for (int k = 0; k < 4; ++k) {
        const float wi = (j0 + k < n) ? r.w[k] : 0.0f;
        acc[0] += cdf_weight(wi);       // exact sums (pf_common.h, CDF rule)
        acc[1] += wi * wi;
    }
    double* const park = L.cdf;         // 2 * kThreads doubles here
    auto reduce2 = [&](const double (&v)[2]) {      // totals in L.red[0], L.red[1]
        park[tid] = v[0];
        park[kThreads + tid] = v[1];
        __syncthreads();
        if (tid < 2 * 32) {
            const int k = tid >> 5, j = tid & 31;
            const double* p = park + k * kThreads + j;
            double t = p[0];
#pragma unroll
            for (int i = 1; i < 8; ++i) t += p[32 * i];
            park[k * kThreads + j] = t;
        }
        __syncthreads();
        if (tid < 2) {
            const double* p = park + tid * kThreads;
            double t = p[0];
#pragma unroll
            for (int j = 1; j < 32; ++j) t += p[j];
            L.red[tid] = t;
        }
        __syncthreads();
    };
    reduce2(acc);
    if (tid == 0) {
        *rowSumOut = (float)L.red[0];
        L.red[2] = L.red[0];            // S and SS survive the second reduction
        L.red[3] = L.red[1];
        L.dec = small_decide_test(L.red[0], L.red[1], n, range, true, false);
    }
    __syncthreads();
    SmallDecision d = L.dec;
    if (!d.resample) return d;          // uniform over the workgroup

    const float4 t4 = (j0 < n) ? *reinterpret_cast<const float4*>(thSrc + j0) : make_float4(0, 0, 0, 0);
    r.th[0] = t4.x; r.th[1] = t4.y; r.th[2] = t4.z; r.th[3] = t4.w;
    double cs2[2] = { 0, 0 };           // c, sn
#pragma unroll
    for (int k = 0; k < 4; ++k) {
        const float wi = (j0 + k < n) ? r.w[k] : 0.0f;
        r.th[k] = wrap_two_pi(r.th[k]);
        float sn, cs;
        sincos_bounded(r.th[k], sn, cs);
        cs2[0] += wi * cs;
        cs2[1] += wi * sn;
    }
    reduce2(cs2);
    if (tid == 0) {
        SmallDecision dd = L.dec;
        small_decide_spread(dd, L.red[2], L.red[3], L.red[0], L.red[1], n, key, rsIndex);
        L.dec = dd;
    }
    __syncthreads();
    d = L.dec;
    small_resample(r, j0, n, L, d, L.red[2], key, index0, range, stdRange, cx, cy, cz);
    return d;
}

// ------------------------------------------------------------------ the measured beacon: gate + row step, one pass
// rowScale[slot row]: pending factor on the stored weights of that row (1 after a resample / spawn).
// rowFlag[f * nTag + i]: 0 = row i was normalised (lazily), 1 + numSpawn = it was resampled here.
// ANY_MOVING = false (the host knows that no beacon of the batch moved since its last deposit -- beacons are anchors,
// csvslam.c never deposits beacon VIO): the headings are not read unless the row resamples (small_filter_update_lazy_theta).
#ifndef PF_SLAM_STATIC_WAVES
#define PF_SLAM_STATIC_WAVES 6
#endif
template <bool ANY_MOVING>
__device__ __forceinline__ void slam_gate_rows_body(const SlamPtrs& P, const SlamDims& D, const SlamStep& one, const SlamStep* __restrict__ many,
                                                    const Motion* __restrict__ bcnMo, float* __restrict__ rowSum,
                                                    float* __restrict__ rowScale, int* __restrict__ rowFlag,
                                                    int* __restrict__ forcedCount)
{
    const int f = blockIdx.y, i = blockIdx.x;
    if (f == 0 && i == 0 && threadIdx.x == 0) *forcedCount = 0;         // filled by k_slam_tag, read by k_slam_rows_forced
    const SlamStep st = many ? many[f] : one;
    if (st.mode != 1) return;
    __shared__ SmallLds L;
    __shared__ float tagp[3];

    const int c = st.target, n = D.nBcn, j0 = threadIdx.x * 4;
    const size_t base = slam_row_base(D, f, c, i);
    const size_t srow = ((size_t)f * D.maxBcns + c) * D.nTag + i;
    const float scale = rowScale[srow];
    const bool moving = ANY_MOVING && ((st.movingMask >> c) & 1u);
    Row4 r;
    {
        // (unconditional loads from a clamped index, all in flight together: see row_load)
        const int jl = j0 < n ? j0 : 0;
        const bool in = j0 < n;
        float4 v[5];
        v[0] = ld4<PF_SLAM_NT_LDW>(P.bcn[0] + base + jl);
#pragma unroll
        for (int q = 1; q < 4; ++q) v[q] = ld4<PF_SLAM_NT_LD>(P.bcn[q] + base + jl);
        v[4] = make_float4(0, 0, 0, 0);
        if (moving) v[4] = ld4<PF_SLAM_NT_LD>(P.bcn[4] + base + jl);       // uniform over the workgroup
#pragma unroll
        for (int q = 0; q < 5; ++q)
            if (!in) v[q] = make_float4(0, 0, 0, 0);
        r.w[0] = v[0].x; r.w[1] = v[0].y; r.w[2] = v[0].z; r.w[3] = v[0].w;
        r.x[0] = v[1].x; r.x[1] = v[1].y; r.x[2] = v[1].z; r.x[3] = v[1].w;
        r.y[0] = v[2].x; r.y[1] = v[2].y; r.y[2] = v[2].z; r.y[3] = v[2].w;
        r.z[0] = v[3].x; r.z[1] = v[3].y; r.z[2] = v[3].z; r.z[3] = v[3].w;
        r.th[0] = v[4].x; r.th[1] = v[4].y; r.th[2] = v[4].z; r.th[3] = v[4].w;
    }

    // tag particle i after its VIO motion (stored by k_slam_tag, which runs after every row has read the old one):
    // one lane computes it while the row loads are in flight, the others pick it up from LDS
    if (threadIdx.x == 0) {
        const size_t ti = (size_t)f * D.tagStride + i;
        float px = P.tag[1][ti], py = P.tag[2][ti], pz = P.tag[3][ti], pth = P.tag[4][ti];
        if (st.tagMoving) {
            float sn0, cs0;
            motion_particle(RngKey{ st.seed, 0u, st.epoch }, (uint64_t)i, st.tagMo, st.tagMo.stdTheta <= 1e-4f, px, py, pz, pth, sn0, cs0);
        }
        tagp[0] = px; tagp[1] = py; tagp[2] = pz;
    }
    __syncthreads();
    const float tx = tagp[0], ty = tagp[1], tz = tagp[2];

    const RngKey key{ st.seed, (uint32_t)(c + 1), st.epoch };
    Motion mo{};
    if (moving) mo = bcnMo[(size_t)f * D.maxBcns + c];
    const bool smallTurn = mo.stdTheta <= 1e-4f;
    float sn[4] = { 0, 0, 0, 0 }, cs[4] = { 1, 1, 1, 1 };
#pragma unroll
    for (int k = 0; k < 4; ++k) {
        const int j = j0 + k;
        if (j >= n) continue;
        r.w[k] *= scale;                    // pending normalisation of an earlier deposit (pfResample.c:195-199)
        if (moving) motion_particle(key, (uint64_t)i * n + j, mo, smallTurn, r.x[k], r.y[k], r.z[k], r.th[k], sn[k], cs[k]);
        // pfMeasurement.c:142-147
        const float dx = tx - r.x[k], dy = ty - r.y[k], dz = tz - r.z[k];
        const float pRange = sqrt_rn(dx * dx + dy * dy + dz * dz);
        if (fabsf(pRange - st.range) > st.thresh) r.w[k] *= st.minWeight;
    }

    // _resampleBcn(force = 0) of this row; the normalisation stays pending in rowScale.  The row sum that
    // pfMeasurement.c:148-150 multiplies into the tag particle's weight is the total of the same reduction.
    float* const rsOut = rowSum + (size_t)f * D.nTag + i;
    SmallDecision d;
    if (moving)
        d = small_filter_update(r, sn, cs, j0, n, L, key, (uint64_t)i * n, (uint64_t)i, st.range, st.stdRange, true, false, tx, ty, tz,
                                true, rsOut);
    else
        d = small_filter_update_lazy_theta(r, P.bcn[4] + base, j0, n, L, key, (uint64_t)i * n, (uint64_t)i, st.range, st.stdRange,
                                           tx, ty, tz, rsOut);
    row_store1_gate(P.bcn[0], base, j0, n, r.w);
    if (moving || d.resample) {
        row_store1_gate(P.bcn[1], base, j0, n, r.x);
        row_store1_gate(P.bcn[2], base, j0, n, r.y);
        row_store1_gate(P.bcn[3], base, j0, n, r.z);
        row_store1_gate(P.bcn[4], base, j0, n, r.th);
    }
    if (threadIdx.x == 0) {
        rowScale[srow] = d.resample ? 1.0f : d.m;
        rowFlag[(size_t)f * D.nTag + i] = d.resample ? 1 + d.numSpawn : 0;
    }
}

// some beacon of the batch moved: the rows carry their headings and a motion step (98 registers, 4 waves per SIMD)
__global__ __launch_bounds__(kThreads) void k_slam_gate_rows_moving(SlamPtrs P, SlamDims D, SlamStep one, const SlamStep* __restrict__ many,
                                                                   const Motion* __restrict__ bcnMo, float* __restrict__ rowSum,
                                                                   float* __restrict__ rowScale, int* __restrict__ rowFlag,
                                                                   int* __restrict__ forcedCount)
{
    slam_gate_rows_body<true>(P, D, one, many, bcnMo, rowSum, rowScale, rowFlag, forcedCount);
}
// no beacon moved (anchors: the usual case)
__global__ __launch_bounds__(kThreads) __attribute__((amdgpu_waves_per_eu(PF_SLAM_STATIC_WAVES, PF_SLAM_STATIC_WAVES)))
void k_slam_gate_rows_static(SlamPtrs P, SlamDims D, SlamStep one, const SlamStep* __restrict__ many,
                             const Motion* __restrict__ bcnMo, float* __restrict__ rowSum, float* __restrict__ rowScale,
                             int* __restrict__ rowFlag, int* __restrict__ forcedCount)
{
    slam_gate_rows_body<false>(P, D, one, many, bcnMo, rowSum, rowScale, rowFlag, forcedCount);
}

// ------------------------------------------------------------------ the tag particles of every filter
__global__ __launch_bounds__(kThreads) void k_slam_tag(SlamPtrs P, SlamDims D, SlamStep one, const SlamStep* __restrict__ many,
                                                      const float* __restrict__ rowSum, int* __restrict__ tagBranch,
                                                      int* __restrict__ forcedList, int* __restrict__ forcedCount)
{
    const int f = blockIdx.x;
    const SlamStep st = many ? many[f] : one;
    if (st.mode == 0) return;
    __shared__ SmallLds L;

    const int n = D.nTag, j0 = threadIdx.x * 4;
    const size_t base = (size_t)f * D.tagStride;
    const float* const src[5] = { P.tag[0], P.tag[1], P.tag[2], P.tag[3], P.tag[4] };
    Row4 r;
    row_load(src, base, j0, n, r);
    const RngKey key{ st.seed, 0u, st.epoch };
    const bool smallTurn = st.tagMo.stdTheta <= 1e-4f;
    float sn[4] = { 0, 0, 0, 0 }, cs[4] = { 1, 1, 1, 1 };
#pragma unroll
    for (int k = 0; k < 4; ++k) {
        const int j = j0 + k;
        if (j >= n) continue;
        if (st.tagMoving) {
            motion_particle(key, (uint64_t)j, st.tagMo, smallTurn, r.x[k], r.y[k], r.z[k], r.th[k], sn[k], cs[k]);
        } else {
            r.th[k] = wrap_two_pi(r.th[k]);
            sincos_bounded(r.th[k], sn[k], cs[k]);
        }
        if (st.mode == 1) r.w[k] *= rowSum[(size_t)f * D.nTag + j];
    }
    int branch = 0;
    if (st.mode == 1) {
        // pfResample.c:86-139 on the tag particles: no respawn, never forced
        const SmallDecision d = small_filter_update(r, sn, cs, j0, n, L, key, 0, 0, st.range, st.stdRange, false, false, 0.f, 0.f, 0.f);
        branch = d.resample;
    }
    row_store1(P.tag[0], base, j0, n, r.w);
    row_store1(P.tag[1], base, j0, n, r.x);
    row_store1(P.tag[2], base, j0, n, r.y);
    row_store1(P.tag[3], base, j0, n, r.z);
    row_store1(P.tag[4], base, j0, n, r.th);
    if (threadIdx.x == 0) {
        tagBranch[f] = branch;
        if (branch) forcedList[atomicAdd(forcedCount, 1)] = f;     // order irrelevant: filters are independent
    }
}

// ------------------------------------------------------------------ beacon rows: motion, spawn, forced _resampleBcn
// One row (tag particle k) of one beacon slot c: first-range spawn, VIO motion, forced _resampleBcn.  Called by every
// thread of the workgroup with the same arguments (it has a barrier behind workgroup-uniform conditions).
__device__ __forceinline__ void slam_row_work(const SlamPtrs& P, const SlamDims& D, const SlamStep& st, const Motion* __restrict__ bcnMo,
                                              bool forcedPass, float* __restrict__ rowScale, const int* __restrict__ rowFlag,
                                              SmallLds& L, int f, int c, int k)
{   // forcedPass = false: the work the host knows about (spawn, motion); true: the forced resample after a tag resample
    const bool isTarget = c == st.target;
    const bool active = (st.activeMask >> c) & 1u;     // one of allBcns[]: takes the VIO commit and a forced resample
    if (!active && !isTarget) return;
    const int n = D.nBcn, j0 = threadIdx.x * 4;
    const size_t base = slam_row_base(D, f, c, k);
    const size_t srow = ((size_t)f * D.maxBcns + c) * D.nTag + k;
    const RngKey key{ st.seed, (uint32_t)(c + 1), st.epoch };
    const size_t ti = (size_t)f * D.tagStride + k;

    if (st.mode == 2 && isTarget) {
        if (forcedPass) return;
        // pfInit.c:34-45: the beacon's particles on the range shell around each (moved) tag particle
        const float tx = P.tag[1][ti], ty = P.tag[2][ti], tz = P.tag[3][ti];
        Row4 r;
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            const int j = j0 + q;
            r.w[q] = 0.f; r.x[q] = 0.f; r.y[q] = 0.f; r.z[q] = 0.f; r.th[q] = 0.f;
            if (j >= n) continue;
            spawn_from_range(key, (uint64_t)k * n + j, tx, ty, tz, st.range, st.stdRange, r.x[q], r.y[q], r.z[q], r.th[q]);
            r.w[q] = 1.0f;
        }
        row_store1(P.bcn[0], base, j0, n, r.w);
        row_store1(P.bcn[1], base, j0, n, r.x);
        row_store1(P.bcn[2], base, j0, n, r.y);
        row_store1(P.bcn[3], base, j0, n, r.z);
        row_store1(P.bcn[4], base, j0, n, r.th);
        if (threadIdx.x == 0) rowScale[srow] = 1.0f;
        return;
    }
    if (!((st.initMask >> c) & 1u)) return;     // not yet spawned: nothing to move (the spawn overwrites it all)

    const bool measured = isTarget && st.mode == 1;     // its motion, gate and unforced row step ran in k_slam_gate_rows
    const bool moving = !forcedPass && ((st.movingMask >> c) & 1u) && !measured;
    // pfResample.c:122-130: tag particles resampled -> every initialised beacon of allBcns[], forced
    const bool forced = forcedPass && st.mode == 1 && active;
    if (!moving && !forced) return;

    const float tx = P.tag[1][ti], ty = P.tag[2][ti], tz = P.tag[3][ti];   // tag particle k as it is NOW (respawn centre)
    if (forced && measured) {
        const int flag = rowFlag[(size_t)f * D.nTag + k];
        if (flag > 0) {
            // Already resampled by k_slam_gate_rows from the very weights, CDF and Philox counters a forced pass would
            // use: identical offspring.  Only the respawned head (pfResample.c:190-192) must sit around the NEW tag k.
            const int numSpawn = flag - 1;
            if (j0 >= numSpawn) return;
            Row4 r;
            const float* const src[5] = { P.bcn[0], P.bcn[1], P.bcn[2], P.bcn[3], P.bcn[4] };
            row_load(src, base, j0, n, r);
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                const int j = j0 + q;
                if (j < numSpawn && j < n) spawn_from_range(key, (uint64_t)k * n + j, tx, ty, tz, st.range, st.stdRange, r.x[q], r.y[q], r.z[q], r.th[q]);
            }
            row_store1(P.bcn[1], base, j0, n, r.x);
            row_store1(P.bcn[2], base, j0, n, r.y);
            row_store1(P.bcn[3], base, j0, n, r.z);
            row_store1(P.bcn[4], base, j0, n, r.th);
            return;
        }
    }

    __syncthreads();            // the previous beacon's pass through this function is done with L
    const float* const src[5] = { P.bcn[0], P.bcn[1], P.bcn[2], P.bcn[3], P.bcn[4] };
    Row4 r;
    row_load(src, base, j0, n, r);
    // the measured beacon's row holds its gated weights (its rowScale is the factor of the step that is being redone
    // as a forced resample: not applied); every other row carries the pending factor of an earlier deposit
    const float scale = measured ? 1.0f : rowScale[srow];
    Motion mo{};
    if (moving) mo = bcnMo[(size_t)f * D.maxBcns + c];
    const bool smallTurn = mo.stdTheta <= 1e-4f;
    float sn[4] = { 0, 0, 0, 0 }, cs[4] = { 1, 1, 1, 1 };
#pragma unroll
    for (int q = 0; q < 4; ++q) {
        const int j = j0 + q;
        if (j >= n) continue;
        if (moving) {
            motion_particle(key, (uint64_t)k * n + j, mo, smallTurn, r.x[q], r.y[q], r.z[q], r.th[q], sn[q], cs[q]);
        } else {
            r.th[q] = wrap_two_pi(r.th[q]);
            sincos_bounded(r.th[q], sn[q], cs[q]);
        }
    }
    if (forced) {
#pragma unroll
        for (int q = 0; q < 4; ++q) r.w[q] *= scale;
        small_filter_update(r, sn, cs, j0, n, L, key, (uint64_t)k * n, (uint64_t)k, st.range, st.stdRange, true, true, tx, ty, tz);
        row_store1(P.bcn[0], base, j0, n, r.w);
        if (threadIdx.x == 0) rowScale[srow] = 1.0f;
    }
    row_store1(P.bcn[1], base, j0, n, r.x);
    row_store1(P.bcn[2], base, j0, n, r.y);
    row_store1(P.bcn[3], base, j0, n, r.z);
    row_store1(P.bcn[4], base, j0, n, r.th);
}

// Pass A, grid (nTag, F), launched only when the host knows of a first range or a moving beacon in the batch.
__global__ __launch_bounds__(kThreads) void k_slam_rows(SlamPtrs P, SlamDims D, SlamStep one, const SlamStep* __restrict__ many,
                                                       const Motion* __restrict__ bcnMo, float* __restrict__ rowScale,
                                                       const int* __restrict__ rowFlag)
{
    const int f = blockIdx.y, k = blockIdx.x;
    const SlamStep st = many ? many[f] : one;
    if (st.mode == 0 || (st.mode != 2 && st.movingMask == 0u)) return;
    __shared__ SmallLds L;
    for (int c = 0; c < D.maxBcns; ++c) slam_row_work(P, D, st, bcnMo, false, rowScale, rowFlag, L, f, c, k);
}

// Pass B, grid (nTag, min(F, 64)): the filters whose tag particles k_slam_tag resampled (it lists them; usually none,
// and then these few workgroups leave at once -- one workgroup per row of the whole batch, as before, cost 46..136 us
// per deposit in dispatching workgroups that had nothing to do).
__global__ __launch_bounds__(kThreads) void k_slam_rows_forced(SlamPtrs P, SlamDims D, SlamStep one, const SlamStep* __restrict__ many,
                                                              const Motion* __restrict__ bcnMo, float* __restrict__ rowScale,
                                                              const int* __restrict__ rowFlag, const int* __restrict__ forcedList,
                                                              const int* __restrict__ forcedCount)
{
    const int k = blockIdx.x;
    const int cnt = *forcedCount;
    __shared__ SmallLds L;
    for (int q = blockIdx.y; q < cnt; q += gridDim.y) {
        const int f = forcedList[q];
        const SlamStep st = many ? many[f] : one;
        for (int c = 0; c < D.maxBcns; ++c) slam_row_work(P, D, st, bcnMo, true, rowScale, rowFlag, L, f, c, k);
    }
}

// ------------------------------------------------------------------ pose estimates
// particleFilter.c:302-333: sums of w, wx, wy, wz, w cos, w sin over the tag particles of each filter
__global__ __launch_bounds__(kThreads) void k_slam_tag_pose(SlamPtrs P, SlamDims D, double* __restrict__ tagPose)
{
    __shared__ double red[6 * kThreads];
    const int f = blockIdx.x, n = D.nTag, j0 = threadIdx.x * 4;
    const float* const src[5] = { P.tag[0], P.tag[1], P.tag[2], P.tag[3], P.tag[4] };
    Row4 r;
    row_load(src, (size_t)f * D.tagStride, j0, n, r);
    double acc[6] = { 0, 0, 0, 0, 0, 0 };
#pragma unroll
    for (int k = 0; k < 4; ++k) {
        if (j0 + k >= n) continue;
        float sn, cs;
        sincos_bounded(r.th[k], sn, cs);
        const float w = r.w[k];
        acc[0] += cdf_weight(w); acc[1] += w * r.x[k]; acc[2] += w * r.y[k]; acc[3] += w * r.z[k]; acc[4] += w * cs; acc[5] += w * sn;
    }
    block_sum_store<6>(acc, red, tagPose + (size_t)f * 6);
}

// particleFilter.c:353-378, inner loop: per tag particle i the weighted sums of its beacon row, folded into the
// row's contribution w1 * (sum2 / s2) exactly as the reference forms it (float), -> rowTerms[f][i][6]
__global__ __launch_bounds__(kThreads) void k_slam_bcn_rows_pose(SlamPtrs P, SlamDims D, const int* __restrict__ slotOf,
                                                                const float* __restrict__ rowScale, float* __restrict__ rowTerms)
{
    __shared__ double red[4 * 6];
    const int f = blockIdx.y, i = blockIdx.x, n = D.nBcn, j0 = threadIdx.x * 4;
    const int c = slotOf[f];
    if (c < 0) return;
    const float* const src[5] = { P.bcn[0], P.bcn[1], P.bcn[2], P.bcn[3], P.bcn[4] };
    Row4 r;
    row_load(src, slam_row_base(D, f, c, i), j0, n, r);
    const float scale = rowScale[((size_t)f * D.maxBcns + c) * D.nTag + i];
    double acc[6] = { 0, 0, 0, 0, 0, 0 };
#pragma unroll
    for (int k = 0; k < 4; ++k) {
        if (j0 + k >= n) continue;
        float sn, cs;
        sincos_bounded(r.th[k], sn, cs);
        const float w = r.w[k] * scale;
        acc[0] += cdf_weight(w); acc[1] += w * r.x[k]; acc[2] += w * r.y[k]; acc[3] += w * r.z[k]; acc[4] += w * cs; acc[5] += w * sn;
    }
    block_sum<6>(acc, red);
    if (threadIdx.x == 0) {
        const float w1 = P.tag[0][(size_t)f * D.tagStride + i];
        const float s2 = (float)acc[0];
        float* o = rowTerms + ((size_t)f * D.nTag + i) * 6;
        o[0] = cdf_weight(w1);
#pragma unroll
        for (int k = 1; k < 6; ++k) o[k] = w1 * (float)acc[k] / s2;
    }
}

// ... outer loop: sums over the tag particles -> bcnPose[f][6]
__global__ __launch_bounds__(kThreads) void k_slam_bcn_pose(SlamDims D, const int* __restrict__ slotOf, const float* __restrict__ rowTerms,
                                                           double* __restrict__ bcnPose)
{
    __shared__ double red[4 * 6];
    const int f = blockIdx.x;
    if (slotOf[f] < 0) return;
    double acc[6] = { 0, 0, 0, 0, 0, 0 };
    for (int i = threadIdx.x; i < D.nTag; i += kThreads) {
        const float* o = rowTerms + ((size_t)f * D.nTag + i) * 6;
#pragma unroll
        for (int k = 0; k < 6; ++k) acc[k] += o[k];
    }
    block_sum<6>(acc, red);
    if (threadIdx.x == 0) {
#pragma unroll
        for (int k = 0; k < 6; ++k) bcnPose[(size_t)f * 6 + k] = acc[k];
    }
}

// tag particles of freshly initialised filters: origin, w = 1, theta = 0 (pfInit.c:27-32, 47-54)
__global__ void k_slam_tag_init(SlamPtrs P, SlamDims D, int f0, int count)
{
    const int idx = blockIdx.x * blockDim.x + threadIdx.x;
    if (idx >= count * D.tagStride) return;
    const size_t p = (size_t)f0 * D.tagStride + idx;
    const int j = idx % D.tagStride;
    P.tag[0][p] = j < D.nTag ? 1.0f : 0.0f;
    P.tag[1][p] = 0.f; P.tag[2][p] = 0.f; P.tag[3][p] = 0.f; P.tag[4][p] = 0.f;
}

}  // namespace pf

// pf_loc_kernels.h -- HIP kernels of the particleFilterLoc hot path (gfx950, wave64).
//
// One deposit (particleFilter.c:178-191) is three or four launches on the filter's stream, no host sync:
//   k_loc_stream   VIO motion (pfMeasurement.c:24-49) + range gate (:108-125) + weight/pose sums
//                  (pfResample.c:41-50, particleFilter.c:263-273) in ONE pass over the SoA arrays;
//                  per-tile sums through LDS.                                          [HBM-bound]
//   k_loc_chunks   one workgroup per 256 tiles: sums of the tile sums and the in-chunk prefix of the
//                  tile weights.                                                       [tiny]
//   k_loc_decide   one wavefront: prefix over chunks (f64), ESS / respawn test, heading bandwidth,
//                  rStart from Philox (pfResample.c:51-68).                            [tiny]
//   k_loc_resample systematic resampling (pfResample.c:70-76): each workgroup rebuilds its tile's
//                  CDF in LDS, binary-searches the ancestors of its contiguous offspring range,
//                  gathers, jitters (pfInit.c:68-79), respawns (pfInit.c:56-66) and writes the
//                  other ping-pong buffer; exits at once on a normalise-only step.      [HBM-bound]
// Unsharded filters defer the resample kernel: the next deposit launches k_loc_deposit instead of
// k_loc_stream, which -- according to the pending decision, read on the device -- either produces the
// offspring and deposits on them in the same pass, or is the streaming pass.          [vector-ALU-bound]
// The normalisation of a non-resampling step (pfResample.c:80-82, w *= N/s) is not a pass: the factor
// is kept in the control block and multiplied in by the next reader of w (same product, same rounding).
//
// CDF rule.  The reference keeps a serial float32 running sum (pfResample.c:45,49), which is void for
// N >= 2^24.  Here the CDF value of particle j of tile t is  B(t) + (inclusive f64 scan inside the tile),
// with B(t) = chunkBase[t / 256] + (f64 prefix of the tile sums inside the chunk); every sum is float64 over
// the float32 weights.  Offspring i sits at u_i = rStart + rStep * i (f64, rStep = S / N) and descends
// from the first particle whose CDF value exceeds u_i; offspring that fall off the end go to the last
// particle (the reference reads out of bounds there).
#pragma once

#include "pf_common.h"

namespace pf {

#ifndef PF_STREAM_UNROLL
#define PF_STREAM_UNROLL 1
#endif
// The ten particle arrays [ping-pong][w,x,y,z,theta], each padded to a whole tile, lie one after the other in one
// allocation: base + stride instead of ten pointers keeps 16 scalar registers free in kernels that are short of them.
// Which accesses of the O(N) passes carry the non-temporal hint (ld4 / st4, pf_common.h); set per site by measurement.
#ifndef PF_NT_STATIC_LD
#define PF_NT_STATIC_LD 1
#endif
#ifndef PF_NT_STATIC_ST
#define PF_NT_STATIC_ST 1
#endif
#ifndef PF_NT_MOTION_LD
#define PF_NT_MOTION_LD 0
#endif
#ifndef PF_NT_MOTION_ST
#define PF_NT_MOTION_ST 0
#endif
#ifndef PF_NT_RS_W
#define PF_NT_RS_W 0
#endif
#ifndef PF_NT_RS_GATHER
#define PF_NT_RS_GATHER 0
#endif
#ifndef PF_NT_RS_ST
#define PF_NT_RS_ST 0
#endif
struct LocBufs {
    float* slab;
    unsigned long long padded;
    __host__ __device__ __forceinline__ float* p(int b, int k) const { return slab + (size_t)(b * 5 + k) * padded; }
};

struct LocCtl {
    int cur;            // buffer holding the current particles
    int src, dst;       // chosen by k_loc_decide for k_loc_resample
    int branch;         // 0 normalise-only, 1 resampled (or freshly spawned)
    int numSpawn;
    int rsValid;        // rsPart holds the pose partials of the current particles (after a spawn or a pose pass)
    float wscale;       // pending factor on the stored weights
    float htheta;
    float s, ss, csum, ssum, ess, m;    // the reference's float totals (diagnostics / pfx_loc_last_stats)
    double S;           // total weight, f64
    double rStart, rStep;
    double pose[6];     // sum w, w*x, w*y, w*z, w*cos, w*sin of the current particles
    long long nDecisions;   // deposits decided so far (index into the branch log)
    long long nResampled;   // ... of which resampled
    // Span of this filter inside a (possibly larger, sharded) global filter.  Unsharded: [0,S) x [0,n).
    double shardBase, shardEnd;     // CDF values at the first / past the last local particle
    long long outBegin, outEnd;     // global offspring indices produced from the local particles
    long long selfBegin, selfEnd;   // global indices of the particles this shard holds (targets kept locally)
    long long nGlobal;
    double local[8];                // sharded: the local sums handed to the all-gather (s ss c sn wx wy wz 0)
};

// What a shard's host needs to know after a sharded decision (copied back once per deposit).
constexpr int kMaxShards = 16;
struct ShardPlan {
    int branch, numSpawn;
    float htheta, wscale;
    double S, rStart, rStep;
    long long F[kMaxShards + 1];    // F[g] = first global offspring index produced by shard g; F[G] = nGlobal
    unsigned long long seq;         // written last: the deposit (epoch) this plan belongs to
};

constexpr int kBranchLogSize = 65536;

// The range gate (pfMeasurement.c:118-123): a particle at squared distance d2 from the beacon keeps its weight unless
// fabsf(sqrtf(d2) - range) > thresh.  sqrtf is correctly rounded and the subtraction monotone, so the particles that keep
// their weight are exactly those with d2 in one closed interval of floats [d2Lo, d2Hi] (or d2 NaN), which the host finds
// once per deposit by bisection over bit patterns with the reference's own expression (gate_of, pf_loc.hip): two
// comparisons per particle on the device, no square root, the same decision for every float d2.
struct Gate { float bx, by, bz, range, thresh, minWeight, d2Lo, d2Hi; };

__device__ __forceinline__ bool gate_rejects(const Gate& ga, float dx, float dy, float dz)
{
    const float d2 = dx * dx + dy * dy + dz * dz;
    return d2 < ga.d2Lo || d2 > ga.d2Hi;            // NaN: kept, as in the reference
}

constexpr int kSumsPerTile = 8;     // s ss c sn wx wy wz (pad)
constexpr int kRsPerTile = 8;       // x y z c sn count (pad)
constexpr int kChunkTiles = 256;    // tiles per chunk of the two-level prefix
constexpr int kMaxChunks = 1024;    // one chunk per thread of k_loc_decide: 1024 * 256 * 1024 = 2^28 particles

__device__ __forceinline__ long long first_offspring_at(double b, double rStart, double rStep, long long n);

// ------------------------------------------------------------------ streaming pass
template <bool MOTION, bool GATE>
__device__ __forceinline__ void stream_tile_cw(const LocBufs& bufs, int cur, float wscale, double* __restrict__ tileSums, long long n,
                                               long long idx0, const RngKey& key, const Motion& mo, const Gate& ga, double* red, int tile);

// Body of k_loc_stream (and of the normalise-only side of k_loc_deposit).  `red` holds 7 * kThreads doubles of LDS.
template <bool MOTION, bool GATE>
__device__ __forceinline__ void stream_tile(const LocBufs& bufs, const LocCtl* __restrict__ ctl,
                                            double* __restrict__ tileSums, long long n, long long idx0,
                                            const RngKey& key, const Motion& mo, const Gate& ga, double* red, int tile)
{
    // idx0: global index of local particle 0 (0 unless the filter is one shard of a larger one); the
    // Philox counter uses the global index, so the noise does not depend on how the filter is sharded
    stream_tile_cw<MOTION, GATE>(bufs, ctl->cur, ctl->wscale, tileSums, n, idx0, key, mo, ga, red, tile);
}

// ... with the two values it needs of the control block handed over (a caller that has loaded them already, together with
// something else it was waiting for: the batched kernels)
template <bool MOTION, bool GATE>
__device__ __forceinline__ void stream_tile_cw(const LocBufs& bufs, int cur, float wscale,
                                               double* __restrict__ tileSums, long long n, long long idx0,
                                               const RngKey& key, const Motion& mo, const Gate& ga, double* red, int tile)
{
    float* __restrict__ pw = bufs.p(cur, 0);
    float* __restrict__ px = bufs.p(cur, 1);
    float* __restrict__ py = bufs.p(cur, 2);
    float* __restrict__ pz = bufs.p(cur, 3);
    float* __restrict__ pt = bufs.p(cur, 4);

    const long long base = (long long)tile * kTile + (long long)threadIdx.x * kPerThread;
    constexpr bool kLd = MOTION ? PF_NT_MOTION_LD : PF_NT_STATIC_LD, kSt = MOTION ? PF_NT_MOTION_ST : PF_NT_STATIC_ST;
    const float4 w4 = ld4<kLd>(pw + base), x4 = ld4<kLd>(px + base), y4 = ld4<kLd>(py + base), z4 = ld4<kLd>(pz + base);
    const float4 t4 = ld4<kLd>(pt + base);
    float w[4] = { w4.x, w4.y, w4.z, w4.w }, x[4] = { x4.x, x4.y, x4.z, x4.w }, y[4] = { y4.x, y4.y, y4.z, y4.w };
    float z[4] = { z4.x, z4.y, z4.z, z4.w }, th[4] = { t4.x, t4.y, t4.z, t4.w };

    double acc[7] = { 0, 0, 0, 0, 0, 0, 0 };
    bool thetaChanged = false;
    const bool smallTurn = mo.stdTheta <= 1e-4f;
#pragma unroll PF_STREAM_UNROLL
    for (int k = 0; k < kPerThread; ++k) {
        const long long i = base + k;
        float wi = (i < n) ? w[k] * wscale : 0.0f;      // pad lanes never carry weight
        float thi = th[k];
        float sn, cs;                                   // of the heading after this deposit
        if (MOTION) {
            motion_particle(key, (uint64_t)(idx0 + i), mo, smallTurn, x[k], y[k], z[k], thi, sn, cs);
        } else {
            // zero motion leaves x,y,z untouched and theta = fmodf(theta, 2pi): a no-op unless |theta| >= 2pi
            const float t2 = wrap_two_pi(thi);
            thetaChanged |= (t2 != thi);
            thi = t2;
            sincos_bounded(thi, sn, cs);
        }
        th[k] = thi;
        if (GATE) {
            // pfMeasurement.c:118-123
            if (gate_rejects(ga, x[k] - ga.bx, y[k] - ga.by, z[k] - ga.bz)) wi *= ga.minWeight;
        }
        w[k] = wi;
        // pfResample.c:44-48 and particleFilter.c:266-272: float products, added in double; the weight total is the
        // CDF's last value and takes the weight as the CDF does (cdf_weight: exact sums)
        acc[0] += cdf_weight(wi);
        acc[1] += wi * wi;
        acc[2] += wi * cs;
        acc[3] += wi * sn;
        acc[4] += wi * x[k];
        acc[5] += wi * y[k];
        acc[6] += wi * z[k];
    }

    st4<kSt>(pw + base, make_float4(w[0], w[1], w[2], w[3]));
    if (MOTION) {
        st4<kSt>(px + base, make_float4(x[0], x[1], x[2], x[3]));
        st4<kSt>(py + base, make_float4(y[0], y[1], y[2], y[3]));
        st4<kSt>(pz + base, make_float4(z[0], z[1], z[2], z[3]));
        st4<kSt>(pt + base, make_float4(th[0], th[1], th[2], th[3]));
    } else if (thetaChanged) {
        st4<kSt>(pt + base, make_float4(th[0], th[1], th[2], th[3]));
    }

    block_sum_store<7>(acc, red, tileSums + (size_t)tile * kSumsPerTile);
}

template <bool MOTION, bool GATE>
__global__ __launch_bounds__(kThreads) void k_loc_stream(LocBufs bufs, const LocCtl* __restrict__ ctl,
                                                        double* __restrict__ tileSums, long long n, long long idx0,
                                                        RngKey key, Motion mo, Gate ga)
{
    __shared__ double red[7 * kThreads];
    stream_tile<MOTION, GATE>(bufs, ctl, tileSums, n, idx0, key, mo, ga, red, (int)blockIdx.x);
}

// ------------------------------------------------------------------ two-level prefix, level 1
// Workgroup c sums the seven per-tile sums of its 256 tiles (-> chunkSums[c]) and writes the inclusive
// prefix of the tile weights inside the chunk (-> tileIncl[t]), made non-decreasing with a max-scan so that
// rounding in the parallel scan can never order two tile boundaries the wrong way round.
//
// pieces != nullptr: the deposit may have been applied by k_loc_deposit (ctl->branch == 1 at this point,
// i.e. the PREVIOUS decision was to resample).  Then the sums of tile t arrive as pieces, one per source tile that
// produced offspring inside it, in the consecutive slots headSlot[t] .. tailSlot[t]; they are added in slot order
// (fixed, so runs are bit-reproducible) and written to tileSums as the streaming kernel would have.
//
// Sharded filters: only the local elements [lo, hi) were produced here (pieces); the elements [0, lo) and [hi, n)
// arrived from the neighbouring shards and k_loc_land summed them per tile into `extra`.
// direct (k_loc_deposit_shard): no pieces -- a tile some of whose elements were produced here has its (partial) sums in
// tileSums already.
struct TilePieces { const double* sums; const int* headSlot; const int* tailSlot; const double* extra; int lo, hi, n; int direct; };

__device__ __forceinline__ void chunks_body(double* __restrict__ tileSums, double* __restrict__ tileIncl,
                                            double* __restrict__ chunkSums, int numTiles,
                                            const LocCtl* __restrict__ ctl, const TilePieces& pieces, int chunk)
{
    __shared__ double red[4 * 7];
    __shared__ double sc[4];
    const int t = chunk * kChunkTiles + threadIdx.x;
    double v[7] = { 0, 0, 0, 0, 0, 0, 0 };
    if (t < numTiles) {
        double* p = tileSums + (size_t)t * kSumsPerTile;
        if (pieces.sums != nullptr && ctl->branch == 1) {
            const int first = t * kTile, last = (first + kTile < pieces.n ? first + kTile : pieces.n) - 1;
            // (hi > lo: a shard without weight produces nothing, and then no piece of this round exists for any tile --
            // the head / tail slots would be those of an earlier resampling)
            if (pieces.direct) {
                if (pieces.hi > pieces.lo && first < pieces.hi && last >= pieces.lo) {
#pragma unroll
                    for (int k = 0; k < 7; ++k) v[k] = p[k];
                }
            } else if (pieces.hi > pieces.lo && first < pieces.hi && last >= pieces.lo) {  // some of the tile was produced here
                int h = pieces.headSlot[t], e = pieces.tailSlot[t];
                h = h < 0 ? 0 : h;
                e = e >= 2 * numTiles ? 2 * numTiles - 1 : e;
                for (int sl = h; sl <= e; ++sl) {
                    const double* q = pieces.sums + (size_t)sl * kSumsPerTile;
#pragma unroll
                    for (int k = 0; k < 7; ++k) v[k] += q[k];
                }
            }
            if (first < pieces.lo || last >= pieces.hi) {           // some of it was received
                const double* q = pieces.extra + (size_t)t * kSumsPerTile;
#pragma unroll
                for (int k = 0; k < 7; ++k) v[k] += q[k];
            }
#pragma unroll
            for (int k = 0; k < 7; ++k) p[k] = v[k];
        } else {
#pragma unroll
            for (int k = 0; k < 7; ++k) v[k] = p[k];
        }
    }
    double total;
    double incl = block_inclusive_scan(v[0], sc, total);
    __syncthreads();
    incl = block_inclusive_max(incl, sc);
    if (t < numTiles) tileIncl[t] = incl;
    __syncthreads();
    block_sum<7>(v, red);
    if (threadIdx.x == kChunkTiles - 1) chunkSums[(size_t)chunk * kSumsPerTile] = incl;     // == last prefix, not a re-sum
    if (threadIdx.x == 0) {
        double* o = chunkSums + (size_t)chunk * kSumsPerTile;
#pragma unroll
        for (int k = 1; k < 7; ++k) o[k] = v[k];
    }
}

static __global__ __launch_bounds__(kChunkTiles) void k_loc_chunks(double* __restrict__ tileSums, double* __restrict__ tileIncl,
                                                           double* __restrict__ chunkSums, int numTiles,
                                                           const LocCtl* __restrict__ ctl, TilePieces pieces)
{
    chunks_body(tileSums, tileIncl, chunkSums, numTiles, ctl, pieces, (int)blockIdx.x);
}

// ------------------------------------------------------------------ decision (one workgroup)
constexpr int kDecideThreads = 64;

// The scalar part of pfResample.c:51-68,80 on the global totals tot[7] = {S, SS, C, SN, WX, WY, WZ} of n
// particles: ESS / respawn test, heading bandwidth, rStart, or the deferred normalisation factor.  A pure function of
// its arguments (every workgroup of k_loc_decide_map evaluates it and gets the same answer).
// force: -1 follow the reference's test, 0 never resample, 1 always resample (worst-case traffic runs).
struct Decision { bool resample; int numSpawn; float s, ss, csum, ssum, ess, htheta, m; double S, rStart, rStep; };

__device__ __forceinline__ float decide_uniform(const RngKey& key) { return uniform01(rng_block(key, P_RSTART, 0, 0).a); }

// u01 = decide_uniform(key): a function of the key alone, so a caller can draw it while the sums are still on their way.
__device__ __forceinline__ Decision decide_compute(const double (&tot)[7], long long n, float range, float u01, int force)
{
    Decision d;
    d.S = tot[0];
    // pfResample.c:51-58 on the float-rounded totals
    d.s = (float)tot[0]; d.ss = (float)tot[1]; d.csum = (float)tot[2]; d.ssum = (float)tot[3];
    d.ess = d.s * d.s / d.ss;
    const float invN = 1.0f / (float)n;
    d.numSpawn = 0;
    if (d.s * invN < kWeightSpawnThresh && range < kRadiusSpawnThresh) d.numSpawn = (int)lroundf((float)n * kPctSpawn);
    d.resample = (d.ess * invN < kResampleThresh) || d.numSpawn > 0;
    if (force >= 0) d.resample = force != 0;
    d.htheta = 0.0f; d.m = 0.0f; d.rStart = 0.0; d.rStep = 0.0;
    if (d.resample) {
        // pfResample.c:60-68
        const float c = d.csum / d.s, sn = d.ssum / d.s;
        float h = c * c + sn * sn;
        h = h > 1e-10f ? h : 1e-10f;
        h = h < 1.0f - 1e-10f ? h : 1.0f - 1e-10f;
        d.htheta = sqrtf(-logf(h) / d.ess);
        d.rStep = d.S / (double)n;
        d.rStart = (double)u01 * d.rStep;
    } else {
        d.m = (float)n / d.s;       // pfResample.c:80-82, deferred: the next reader multiplies w by m
    }
    return d;
}

__device__ __forceinline__ void decide_commit(LocCtl* __restrict__ ctl, const Decision& d, unsigned char* __restrict__ branchLog)
{
    const int cur = ctl->cur;
    ctl->s = d.s; ctl->ss = d.ss; ctl->csum = d.csum; ctl->ssum = d.ssum; ctl->ess = d.ess; ctl->S = d.S;
    ctl->numSpawn = d.numSpawn;
    if (d.resample) {
        ctl->htheta = d.htheta; ctl->rStep = d.rStep; ctl->rStart = d.rStart;
        ctl->src = cur; ctl->dst = cur ^ 1; ctl->cur = cur ^ 1;
        ctl->wscale = 1.0f; ctl->m = 0.0f; ctl->branch = 1; ctl->rsValid = 0;
    } else {
        ctl->wscale = d.m; ctl->m = d.m; ctl->branch = 0; ctl->htheta = 0.0f;
    }
    const long long k = ctl->nDecisions;
    if (branchLog) branchLog[k % kBranchLogSize] = d.resample ? 1 : 0;     // (the filters of a batch keep the two counts only)
    ctl->nDecisions = k + 1;
    if (d.resample) ctl->nResampled += 1;
}

__device__ __forceinline__ bool decide_scalars(LocCtl* __restrict__ ctl, const double (&tot)[7], long long n, float range,
                                               const RngKey& key, int force, unsigned char* __restrict__ branchLog)
{
    const Decision d = decide_compute(tot, n, range, decide_uniform(key), force);
    decide_commit(ctl, d, branchLog);
    return d.resample;
}

// Prefix over the chunk sums by ONE wavefront, no LDS, no barrier: lane l owns the chunks [l*per, (l+1)*per) (per = 1
// up to 16.7 M particles).  The chunk bases are non-decreasing by construction: lane bases by a prefix over the 64 lane
// totals followed by a running maximum, values inside a lane by sequential adds capped at the next lane's base.  put(c, v) receives
// base c for c in [0, numChunks]; tot[] the seven totals (in every lane).
// meanwhile(): called once the loads of a lane's first chunk are on their way and before anything waits for them.
template <typename Put, typename Meanwhile>
__device__ __forceinline__ void chunk_prefix(const double* __restrict__ chunkSums, int numChunks, double (&tot)[7], Put put,
                                             Meanwhile meanwhile)
{
    const int lane = threadIdx.x & 63;
    const int per = (numChunks + 63) / 64;
    const int begin = min(lane * per, numChunks), end = min(begin + per, numChunks);
    double loc[7];
    {
        const double* p = chunkSums + (size_t)(begin < end ? begin : 0) * kSumsPerTile;
#pragma unroll
        for (int k = 0; k < 7; ++k) loc[k] = p[k];
        meanwhile();
#pragma unroll
        for (int k = 0; k < 7; ++k) loc[k] = begin < end ? loc[k] : 0.0;
    }
    for (int c = begin + 1; c < end; ++c) {
        const double* p = chunkSums + (size_t)c * kSumsPerTile;
#pragma unroll
        for (int k = 0; k < 7; ++k) loc[k] += p[k];
    }
    // prefix over the lane totals: six DPP steps, then a running maximum (a no-op while the sums are exact, section 4.1;
    // beyond that it keeps the bases non-decreasing whatever the additions rounded to).  A loop over the 64 lanes, one
    // readlane and one addition each, was 1 us of every deposit.
    const double nextBase = wave_inclusive_max_nonneg(wave_inclusive_scan(loc[0]));
    double laneBase = __shfl_up(nextBase, 1, 64);
    if (lane == 0) laneBase = 0.0;
    const double run = __shfl(nextBase, 63, 64);
    double v = laneBase;
    for (int c = begin; c < end; ++c) {
        put(c, v < nextBase ? v : nextBase);
        v += chunkSums[(size_t)c * kSumsPerTile];
    }
    if (lane == 0) put(numChunks, run);
    tot[0] = run;
#pragma unroll
    for (int k = 1; k < 7; ++k) tot[k] = wave_sum(loc[k]);
}

// The decision of a filter whose resampling stays source-owned (a shard driven through the one-filter entry points): one
// wavefront.
static __global__ __launch_bounds__(64) void k_loc_decide(LocCtl* __restrict__ ctl, const double* __restrict__ chunkSums,
                                                         double* __restrict__ chunkBase, int numChunks, long long n,
                                                         float range, RngKey key, int force,
                                                         unsigned char* __restrict__ branchLog)
{
    double tot[7];
    chunk_prefix(chunkSums, numChunks, tot, [&](int c, double v) { chunkBase[c] = v; }, [] {});
    if (threadIdx.x != 0) return;
    const bool resample = decide_scalars(ctl, tot, n, range, key, force, branchLog);
    if (!resample) {    // the streaming pass's sums are the pose sums of the (unnormalised) current particles
        ctl->pose[0] = tot[0]; ctl->pose[1] = tot[4]; ctl->pose[2] = tot[5]; ctl->pose[3] = tot[6];
        ctl->pose[4] = tot[2]; ctl->pose[5] = tot[3];
    }
    ctl->shardBase = 0.0; ctl->shardEnd = tot[0];
    ctl->outBegin = 0; ctl->outEnd = n; ctl->selfBegin = 0; ctl->selfEnd = n; ctl->nGlobal = n;
}

// CDF value at the START of tile t (t in [0, numTiles]).  The same expression is evaluated by the
// workgroups on both sides of a boundary, and it is non-decreasing in t: inside a chunk tileIncl is
// non-decreasing and the sum is capped by the next chunk's base.
__device__ __forceinline__ double tile_start_cdf(const double* __restrict__ chunkBase, const double* __restrict__ tileIncl, int t,
                                                 double shardBase, double shardEnd)
{
    const int c = t / kChunkTiles;
    double v = chunkBase[c];
    if (t % kChunkTiles != 0) {
        v += tileIncl[t - 1];
        const double cap = chunkBase[c + 1];
        v = v < cap ? v : cap;
    }
    v += shardBase;                         // local prefix -> position in the global CDF
    return v < shardEnd ? v : shardEnd;
}

// The same value with the load unconditional (tileIncl[t - 1] exists for every t in [1, numTiles]; t = 0 reads entry 0
// and drops it): a load behind a branch cannot be hoisted, and a sequence of these calls is then a sequence of round
// trips.  For the plan records, where fifteen of them are wanted at once.
__device__ __forceinline__ double tile_start_cdf_flat(const double* __restrict__ chunkBase, const double* __restrict__ tileIncl, int t,
                                                      double shardEnd, double shardBase = 0.0)
{
    const int c = t / kChunkTiles;
    const double in = tileIncl[t > 0 ? t - 1 : 0];
    const bool inside = t % kChunkTiles != 0;
    const double b0 = chunkBase[c], cap = chunkBase[inside ? c + 1 : c];
    double v = b0 + (inside ? in : 0.0);        // selections, no branch (the compiler sinks the load into one)
    v = v < cap ? v : cap;
    v += shardBase;                             // (0.0 for an unsharded filter)
    return v < shardEnd ? v : shardEnd;
}

// Number of offspring i in [0,n) with rStart + rStep*i < b (u_i is non-decreasing in i), i.e. the first
// offspring that does not precede CDF value b.
__device__ __forceinline__ long long first_offspring_at(double b, double rStart, double rStep, long long n)
{
    if (!((rStep > 0.0) && (rStep < 1e300))) return 0;      // zero / non-finite total weight: all to the last tile
    const double q = ceil((b - rStart) / rStep);
    if (!(q == q)) return 0;
    long long f = q < 0.0 ? 0 : (q > (double)n ? n : (long long)q);
    for (int it = 0; it < 8 && f > 0 && !(rStart + rStep * (double)(f - 1) < b); ++it) --f;
    for (int it = 0; it < 8 && f < n && (rStart + rStep * (double)f < b); ++it) ++f;
    return f;
}

// The same count for n < 2^31 with the quotient estimated through invStep = 1 / rStep (one multiplication instead of a
// float64 division, 32-bit conversions instead of 64-bit ones).  The estimate may be off by one where the division's
// is exact; the two correction loops end at the same f either way, because f is defined by the comparisons alone.
// The caller checks that invStep is finite.
__device__ __forceinline__ int first_offspring_at_i32(double b, double rStart, double rStep, double invStep, int n)
{
    if (!((rStep > 0.0) && (rStep < 1e300))) return 0;
    const double q = ceil((b - rStart) * invStep);
    if (!(q == q)) return 0;
    int f = q < 0.0 ? 0 : (q > (double)n ? n : (int)q);
    for (int it = 0; it < 8 && f > 0 && !(rStart + rStep * (double)(f - 1) < b); ++it) --f;
    for (int it = 0; it < 8 && f < n && (rStart + rStep * (double)f < b); ++it) ++f;
    return f;
}

// What workgroup d of a destination-owned resampling needs before it can load anything else: the source tile t0 that
// holds the ancestor of its first offspring, and the CDF values that bound tile t0 and its successors.  One 64-byte
// record per destination tile, written by k_loc_decide_map: the workgroup reads it together with the control block
// (one round trip) and then knows which weights to load and which of its offspring each source tile takes; without it
// the tile bounds were two more dependent loads (a third for every further source tile), and how far a tile gets was
// worked out by every wave of every workgroup with float64 arithmetic.
constexpr int kPlanEnds = 4;
struct __attribute__((aligned(64))) DestPlan {
    int t0, pad;
    double start0;              // CDF value at the start of tile t0
    double end[kPlanEnds];      // CDF value at the end of tile t0 + i (the total weight from the last tile on)
    int upto[kPlanEnds];        // first offspring (global index) whose position is not below end[i]: where tile t0 + i stops
};
static_assert(sizeof(DestPlan) == 64, "one cache line per destination tile");

// The plan records of the destination tiles whose first offspring has its ancestor in chunk `chunk` (k_loc_decide_map:
// one workgroup per chunk; k_loc_decide_small: the only chunk).  Worked from the SOURCE side: thread tid owns source
// tile t = chunk * kChunkTiles + tid, takes the CDF value at its end and F_t, the number of offspring below that value
// (one load, one first_offspring_at).  The offspring [F_(t-1), F_t) have their ancestors in tile t, so the destination
// tiles d with d * kTile in that range -- those whose FIRST offspring does -- get the record of tile t: its bounds and
// the (end, F) pairs of its three successors, from LDS.  This is the tile the search from the destination side finds
// (the first tile whose end value exceeds u_d: d * kTile < F_t says exactly that), without the search: looking for tile t
// from destination tile d took a halving search over the chunk bases and eight dependent loads from tileIncl, 8 us of
// the 12 us this kernel took on a resampling deposit.
// Called by every thread of the workgroup (one barrier inside; a caller that comes back for another chunk puts one in
// between).  What lies beyond the chunk -- the three tiles after it and the offspring count at its start -- is worked out
// by threads kChunkTiles .. kChunkTiles + 3 where the workgroup has them, else by threads 0 .. 3 in a second go.
// base[]: chunk bases (LDS); tileIncl: tile prefixes inside their chunk (global memory, or LDS for the small kernel).
//
// One shard of a sharded filter (SHARDED; k_loc_shard_decide_plan): the tiles are the shard's, positions in the GLOBAL CDF are
// shardBase + the local prefix, offspring counts are global and enter the records relative to the shard's first element
// (clamped to [0, n]: the index of the destination ELEMENT in the shard's own arrays).  Only the destination tiles that
// hold an offspring of a local ancestor, local elements [loL, hiL), get a record; the tile that loL falls into gets the
// record of the source tile that produces offspring loL (its leading elements arrive from the shard below).
constexpr int kPlanDirect = 4;      // records a thread writes by itself; what a tile has beyond that the wavefront shares
struct PlanLds { double end[kChunkTiles + 3]; int F[kChunkTiles + 4]; };
struct ShardSpan { double base, end; long long selfBegin, nGlobal, outBegin, outEnd; };
template <bool SHARDED = false>
__device__ __forceinline__ void plan_scatter(int chunk, const double* base, const double* tileIncl, DestPlan* __restrict__ plan,
                                             int numTiles, long long n, double rStart, double rStep, double S, PlanLds& L,
                                             const ShardSpan& sp = ShardSpan{})
{
    const int tid = threadIdx.x, last = numTiles - 1;
    const double invStep = 1.0 / rStep;
    const bool fast = !SHARDED && invStep < 1e300;
    auto rel = [&](long long F) -> int { F -= sp.selfBegin; return F < 0 ? 0 : (F > n ? (int)n : (int)F); };
    const int loL = SHARDED ? rel(sp.outBegin) : 0;
    // slot k of L, k < kChunkTiles + 3: tile chunk * kChunkTiles + k (beyond the last tile: the last tile again); slot
    // kChunkTiles + 3 of F: the count at the START of the chunk
    auto fill = [&](int slot, int& tOut, int& FOut) {
        const bool startSlot = slot == kChunkTiles + 3;
        int t = chunk * kChunkTiles + slot;
        t = t < last ? t : last;
        const double endV = SHARDED ? tile_start_cdf_flat(base, tileIncl, startSlot ? chunk * kChunkTiles : t + 1, sp.end, sp.base)
                                    : tile_start_cdf_flat(base, tileIncl, startSlot ? chunk * kChunkTiles : t + 1, S);
        int F;
        if (SHARDED) {
            // (the shard's first and last offspring are what the global plan says, as in resample_deposit_tile)
            const long long Fg = (startSlot && chunk == 0) ? sp.outBegin
                               : (!startSlot && chunk * kChunkTiles + slot >= last) ? sp.outEnd
                               : first_offspring_at(endV, rStart, rStep, sp.nGlobal);
            F = rel(Fg);
        } else if (fast) F = first_offspring_at_i32(endV, rStart, rStep, invStep, (int)n);
        else F = (int)first_offspring_at(endV, rStart, rStep, n);
        if (slot < kChunkTiles + 4) {
            L.F[slot] = F;
            if (!startSlot) L.end[slot] = endV;
        }
        tOut = t; FOut = F;
    };
    int t, F;
    fill(tid, t, F);
    if (blockDim.x <= kChunkTiles && tid < 4) {     // a workgroup without the extra wavefront: four threads go twice
        int t2, F2;
        fill(kChunkTiles + tid, t2, F2);
    }
    __syncthreads();
    int dLo = 0, dHi = 0;
    DestPlan P;
    P.t0 = t; P.pad = 0;
    if (tid < kChunkTiles && chunk * kChunkTiles + tid <= last) {
        const int Fprev = tid > 0 ? L.F[tid - 1] : L.F[kChunkTiles + 3];
        dLo = (Fprev + kTile - 1) / kTile;
        if (SHARDED && Fprev == loL && F > loL) dLo = loL / kTile;      // the tile that produces the shard's first own offspring
        dHi = t == last ? numTiles : (F + kTile - 1) / kTile;
        const double chunkStart = SHARDED ? base[chunk] + sp.base : base[chunk];
        const double cdfEnd = SHARDED ? sp.end : S;
        P.start0 = tid > 0 ? L.end[tid - 1] : (chunkStart < cdfEnd ? chunkStart : cdfEnd);
#pragma unroll
        for (int i = 0; i < kPlanEnds; ++i) {
            const int k = (t + i < last ? t + i : last) - chunk * kChunkTiles;
            P.end[i] = L.end[k];
            P.upto[i] = L.F[k];
        }
#pragma unroll
        for (int k = 0; k < kPlanDirect; ++k)
            if (dLo + k < dHi) plan[dLo + k] = P;
    }
    if (tid >= kChunkTiles) return;             // (whole wavefronts)
    // a tile with more than kPlanDirect destination tiles (concentrated weight): its wavefront writes the rest together
    unsigned long long more = __ballot(dHi - dLo > kPlanDirect);
    while (more) {
        const int l = __ffsll((long long)more) - 1;
        more &= more - 1;
        DestPlan Q;
        Q.t0 = __shfl(P.t0, l, 64); Q.pad = 0;
        Q.start0 = __shfl(P.start0, l, 64);
#pragma unroll
        for (int i = 0; i < kPlanEnds; ++i) { Q.end[i] = __shfl(P.end[i], l, 64); Q.upto[i] = __shfl(P.upto[i], l, 64); }
        const int from = __shfl(dLo, l, 64) + kPlanDirect, to = __shfl(dHi, l, 64);
        for (int d = from + (tid & 63); d < to; d += 64) plan[d] = Q;
    }
}

#ifndef PF_SMALL_CHUNKS
#define PF_SMALL_CHUNKS 1
#endif
// ------------------------------------------------------------------ decision + plan records (unsharded filters)
// The decision of an unsharded filter, whose resampling is destination-owned (resample_dest_tile below): besides the
// scalars, a resampling step needs the plan record of every destination tile (DestPlan above).  Working it out inside
// the big kernel puts two to four dependent loads in front of everything else a workgroup does; here it is one thread
// per destination tile, all of them at once.  Grid: one workgroup of 256 threads per chunk of 256 tiles.  EVERY workgroup
// computes the chunk prefix and the decision (a pure function of the chunk sums: same values everywhere) and keeps the
// chunk bases in LDS; workgroup 0 alone writes them, and the control block.
constexpr int kDecideMapThreads = kChunkTiles + 64;
__device__ __forceinline__ void decide_map_body(LocCtl* __restrict__ ctl, const double* __restrict__ chunkSums,
                                                double* __restrict__ chunkBase, const double* __restrict__ tileIncl,
                                                DestPlan* __restrict__ plan, int numChunks, int numTiles, long long n,
                                                float range, const RngKey& key, int force,
                                                unsigned char* __restrict__ branchLog, int chunk)
{
    __shared__ double base[kMaxChunks + 1];
    __shared__ double dec[4];
    __shared__ PlanLds planLds;
    const int tid = threadIdx.x;
    if (tid < 64) {
        double tot[7];
        const bool first = chunk == 0;
        float u01 = 0.0f;
        chunk_prefix(chunkSums, numChunks, tot, [&](int c, double v) { base[c] = v; if (first) chunkBase[c] = v; },
                     [&] { u01 = decide_uniform(key); });
        if (tid == 0) {
            const Decision d = decide_compute(tot, n, range, u01, force);
            dec[0] = d.resample ? 1.0 : 0.0; dec[1] = d.rStart; dec[2] = d.rStep; dec[3] = d.S;
            if (first) {
                decide_commit(ctl, d, branchLog);
                if (!d.resample) {  // the streaming pass's sums are the pose sums of the (unnormalised) current particles
                    ctl->pose[0] = tot[0]; ctl->pose[1] = tot[4]; ctl->pose[2] = tot[5]; ctl->pose[3] = tot[6];
                    ctl->pose[4] = tot[2]; ctl->pose[5] = tot[3];
                }
                ctl->shardBase = 0.0; ctl->shardEnd = tot[0];
                ctl->outBegin = 0; ctl->outEnd = n; ctl->selfBegin = 0; ctl->selfEnd = n; ctl->nGlobal = n;
            }
        }
    }
    __syncthreads();
    if (dec[0] == 0.0) return;
    plan_scatter(chunk, base, tileIncl, plan, numTiles, n, dec[1], dec[2], dec[3], planLds);
}

static __global__ __launch_bounds__(kDecideMapThreads) void k_loc_decide_map(LocCtl* __restrict__ ctl, const double* __restrict__ chunkSums,
                                                            double* __restrict__ chunkBase, const double* __restrict__ tileIncl,
                                                            DestPlan* __restrict__ plan, int numChunks, int numTiles, long long n,
                                                            float range, RngKey key, int force,
                                                            unsigned char* __restrict__ branchLog)
{
    decide_map_body(ctl, chunkSums, chunkBase, tileIncl, plan, numChunks, numTiles, n, range, key, force, branchLog, (int)blockIdx.x);
}

// The global decision of a sharded filter and the plan records of the shard's destination-owned fused pass, one workgroup
// per chunk of the shard's tiles.  Input: the CHUNK SUMS of every shard, `all[g * rowDoubles + c * 8 + k]` (what every
// shard's k_loc_chunks left, exchanged by one all-gather or written by the shards themselves into each other's tables;
// rows are zero beyond a shard's last chunk).  Every workgroup of every shard derives from them, with the same additions
// in the same order: each shard's seven totals (16 threads per shard, 16 partial sums each, added up in a fixed order),
// the global totals and the decision (a pure function of them, so all agree, as in k_loc_decide_map), where every shard's
// offspring begin, and -- from the shard's own row -- the local chunk bases the plan records need.  Workgroup 0 alone
// commits the decision to the control block and publishes the host's copy.
// (Round 2 had a launch in between that reduced the shard's chunk sums to eight totals for the all-gather,
// k_loc_decide<LOCAL_ONLY>: 5 us of every deposit, for a message that is latency-bound at 64 bytes and at 2 kB alike.)
// writePlan = 0: the shard's fused pass is the source-owned one (PFX_SHARD_DIRECT=0), which needs no plan records.
constexpr int kShardSub = 16;       // partial sums per shard
static __global__ __launch_bounds__(kDecideMapThreads) void k_loc_shard_decide_plan(LocCtl* __restrict__ ctl, const double* __restrict__ all,
                                                                             int rowDoubles, int rowChunks,
                                                                             int numShards, int rank, long long nGlobal,
                                                                             long long selfBegin, long long selfEnd, float range,
                                                                             RngKey key, int force, unsigned char* __restrict__ branchLog,
                                                                             ShardPlan* hostPlan, unsigned long long seq,
                                                                             double* __restrict__ chunkBase,
                                                                             const double* __restrict__ tileIncl,
                                                                             DestPlan* __restrict__ plan, int numChunks, int numTiles, int n,
                                                                             int writePlan)
{
    __shared__ double base[kMaxChunks + 1];
    __shared__ PlanLds planLds;
    __shared__ double part[kMaxShards * kShardSub * 7];
    __shared__ double totS[kMaxShards * 8];
    __shared__ double decD[6];          // resample, rStart, rStep, S, shardBase, shardEnd
    __shared__ long long decL[2];       // outBegin, outEnd
    const int tid = threadIdx.x;
    if (tid < kMaxShards * kShardSub) {
        const int g = tid / kShardSub, sub = tid % kShardSub;
        double p[7] = { 0, 0, 0, 0, 0, 0, 0 };
        if (g < numShards) {
            const double* row = all + (size_t)g * rowDoubles;
            for (int c = sub; c < rowChunks; c += kShardSub) {
#pragma unroll
                for (int k = 0; k < 7; ++k) p[k] += row[c * kSumsPerTile + k];
            }
        }
#pragma unroll
        for (int k = 0; k < 7; ++k) part[(g * kShardSub + sub) * 7 + k] = p[k];
    }
    __syncthreads();
    if (tid < kMaxShards * 7) {
        const int g = tid / 7, k = tid % 7;
        double t = 0.0;
        for (int sub = 0; sub < kShardSub; ++sub) t += part[(g * kShardSub + sub) * 7 + k];
        totS[g * 8 + k] = t;
    }
    __syncthreads();
    if (tid < 64) {
        // every lane of the first wavefront takes the decision (the same arithmetic on the same values); lane g then works
        // out where shard g's offspring begin -- up to seventeen float64 divisions side by side instead of in a row.  The
        // totals are added up shard by shard, in rank order, every lane the same additions; lane g keeps the prefix in
        // front of shard g
        double mine[7], tot[7] = { 0, 0, 0, 0, 0, 0, 0 };
#pragma unroll
        for (int k = 0; k < 7; ++k) mine[k] = tid < numShards ? totS[tid * 8 + k] : 0.0;
        double myBase = 0.0;                        // start of shard g's span of the global CDF (the total for lane G)
#pragma unroll 1
        for (int q = 0; q < numShards; ++q) {
#pragma unroll
            for (int k = 0; k < 7; ++k) tot[k] += __shfl(mine[k], q, 64);
            if (tid == q + 1) myBase = tot[0];      // sequential prefix: non-decreasing, the last one == S exactly
        }
        const Decision d = decide_compute(tot, nGlobal, range, decide_uniform(key), force);
        const int g = tid <= numShards ? tid : numShards;
        long long F = 0;
        if (d.resample) F = (g == 0) ? 0 : (g == numShards ? nGlobal : first_offspring_at(myBase, d.rStart, d.rStep, nGlobal));
        if (tid == rank) { decL[0] = F; decD[4] = myBase; }
        if (tid == rank + 1) { decL[1] = F; decD[5] = myBase; }
        if (tid == 0) { decD[0] = d.resample ? 1.0 : 0.0; decD[1] = d.rStart; decD[2] = d.rStep; decD[3] = d.S; }
        if (blockIdx.x == 0) {
            if (tid <= numShards) hostPlan->F[tid] = F;
            const long long ob = __shfl(F, rank, 64), oe = __shfl(F, rank + 1, 64);
            const double sb = __shfl(myBase, rank, 64), se = __shfl(myBase, rank + 1, 64);
            if (tid == 0) {
                decide_commit(ctl, d, branchLog);
                const double* l = totS + rank * 8;  // the shard's own totals
#pragma unroll
                for (int k = 0; k < 7; ++k) ctl->local[k] = l[k];
                ctl->local[7] = 0.0;
                if (!d.resample) {  // local part of the pose sums; the host all-reduces them when a pose is asked for
                    ctl->pose[0] = l[0]; ctl->pose[1] = l[4]; ctl->pose[2] = l[5]; ctl->pose[3] = l[6]; ctl->pose[4] = l[2]; ctl->pose[5] = l[3];
                }
                hostPlan->branch = d.resample ? 1 : 0;
                hostPlan->numSpawn = d.numSpawn; hostPlan->htheta = d.htheta; hostPlan->wscale = d.resample ? 1.0f : d.m;
                hostPlan->S = tot[0]; hostPlan->rStart = d.rStart; hostPlan->rStep = d.rStep;
                ctl->shardBase = sb; ctl->shardEnd = se;
                ctl->outBegin = ob; ctl->outEnd = oe;
                ctl->selfBegin = selfBegin; ctl->selfEnd = selfEnd; ctl->nGlobal = nGlobal;
            }
            // the sequence number goes out after every lane's part of the plan (one wavefront: program order + the fence)
            __threadfence_system();
            if (tid == 0) __hip_atomic_store(&hostPlan->seq, seq, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_SYSTEM);
        }
    } else if (tid < 128) {
        // the second wavefront meanwhile: the shard's LOCAL chunk bases, from its own row of the table
        double ltot[7];
        const bool first = blockIdx.x == 0;
        chunk_prefix(all + (size_t)rank * rowDoubles, numChunks, ltot, [&](int c, double v) { base[c] = v; if (first) chunkBase[c] = v; }, [] {});
    }
    __syncthreads();
    if (decD[0] == 0.0 || !writePlan) return;
    const ShardSpan sp{ decD[4], decD[5], selfBegin, nGlobal, decL[0], decL[1] };
    plan_scatter<true>(blockIdx.x, base, tileIncl, plan, numTiles, n, decD[1], decD[2], decD[3], planLds, sp);
}

// The same two steps -- k_loc_chunks, then k_loc_decide_map -- in ONE launch of ONE workgroup, for filters of at most
// kSmallChunks chunks: one chunk, N <= 262 144, which is where the reference's own sizes lie (PF_N_TAG_LOC = 10 000).  There
// a deposit is bound by its launches (three of ~5 us each) and a launch saved is a tenth of the step (measured at
// N = 10 000: 33.4 -> 30.4 us per step of two deposits).  With four chunks (-DPF_SMALL_CHUNKS=4, N <= 1 M) it loses, also
// with the plan records written from the source side: 36.7 -> 38.8 us per step at N = 500 000, 41.7 -> 58.1 at N = 1 M --
// one workgroup takes the chunks one after the other (five barriers each) where the two kernels have a workgroup per
// chunk.  The step further -- k_loc_deposit and this kernel in one launch, the workgroup that arrives last (device-scope
// fence, one atomic) taking the decision -- was measured too: 28.1 -> 28.7 us per step at N = 10 000 (the step is bound by the
// host's calls there, not by the launches), 28.7 -> 31.1 at N = 100 000, 30.8 -> 40.0 at N = 262 144; dropped.
// Same functions on the same values in the same order as the two kernels: the results are bit-identical, the
// arrays they leave in global memory (tileIncl, chunkSums, chunkBase, plan) are those the big kernels expect; inside the
// kernel they travel through LDS.
constexpr int kSmallChunks = PF_SMALL_CHUNKS;
__device__ __forceinline__ void decide_small_body(LocCtl* __restrict__ ctl, const double* __restrict__ tileSums,
                                                  double* __restrict__ tileIncl, double* __restrict__ chunkSums,
                                                  double* __restrict__ chunkBase, DestPlan* __restrict__ plan,
                                                  int numChunks, int numTiles, long long n, float range, const RngKey& key,
                                                  int force, unsigned char* __restrict__ branchLog)
{
    __shared__ double red[4 * 7];
    __shared__ double sc[4];
    __shared__ double cs[kSmallChunks * kSumsPerTile];
    __shared__ double inclS[kSmallChunks * kChunkTiles];
    __shared__ double base[kSmallChunks + 1];
    __shared__ double dec[4];
    __shared__ PlanLds planLds;
    const int tid = threadIdx.x;
    // all the tile sums first: one round trip
    double v[kSmallChunks][7];
#pragma unroll
    for (int c = 0; c < kSmallChunks; ++c) {
        const int t = c * kChunkTiles + tid;
#pragma unroll
        for (int k = 0; k < 7; ++k) v[c][k] = (c < numChunks && t < numTiles) ? tileSums[(size_t)t * kSumsPerTile + k] : 0.0;
    }
#pragma unroll
    for (int c = 0; c < kSmallChunks; ++c) {
        if (c < numChunks) {                                 // uniform; the body of k_loc_chunks for chunk c
            const int t = c * kChunkTiles + tid;
            double total;
            double incl = block_inclusive_scan(v[c][0], sc, total);
            __syncthreads();
            incl = block_inclusive_max(incl, sc);
            if (t < numTiles) { tileIncl[t] = incl; inclS[t] = incl; }
            __syncthreads();
            block_sum<7>(v[c], red);
            if (tid == kChunkTiles - 1) { cs[c * kSumsPerTile] = incl; chunkSums[(size_t)c * kSumsPerTile] = incl; }
            if (tid == 0) {
#pragma unroll
                for (int k = 1; k < 7; ++k) { cs[c * kSumsPerTile + k] = v[c][k]; chunkSums[(size_t)c * kSumsPerTile + k] = v[c][k]; }
            }
            __syncthreads();
        }
    }
    if (tid < 64) {                                          // the head of k_loc_decide_map
        double tot[7];
        chunk_prefix(cs, numChunks, tot, [&](int c, double b) { base[c] = b; chunkBase[c] = b; }, [] {});
        if (tid == 0) {
            const Decision d = decide_compute(tot, n, range, decide_uniform(key), force);
            dec[0] = d.resample ? 1.0 : 0.0; dec[1] = d.rStart; dec[2] = d.rStep; dec[3] = d.S;
            decide_commit(ctl, d, branchLog);
            if (!d.resample) {
                ctl->pose[0] = tot[0]; ctl->pose[1] = tot[4]; ctl->pose[2] = tot[5]; ctl->pose[3] = tot[6];
                ctl->pose[4] = tot[2]; ctl->pose[5] = tot[3];
            }
            ctl->shardBase = 0.0; ctl->shardEnd = tot[0];
            ctl->outBegin = 0; ctl->outEnd = n; ctl->selfBegin = 0; ctl->selfEnd = n; ctl->nGlobal = n;
        }
    }
    __syncthreads();
    if (dec[0] == 0.0) return;
    for (int c = 0; c < numChunks; ++c) {
        if (c > 0) __syncthreads();
        plan_scatter(c, base, inclS, plan, numTiles, n, dec[1], dec[2], dec[3], planLds);
    }
}

static __global__ __launch_bounds__(kChunkTiles) void k_loc_decide_small(LocCtl* __restrict__ ctl, const double* __restrict__ tileSums,
                                                                 double* __restrict__ tileIncl, double* __restrict__ chunkSums,
                                                                 double* __restrict__ chunkBase, DestPlan* __restrict__ plan,
                                                                 int numChunks, int numTiles, long long n, float range, RngKey key,
                                                                 int force, unsigned char* __restrict__ branchLog)
{
    decide_small_body(ctl, tileSums, tileIncl, chunkSums, chunkBase, plan, numChunks, numTiles, n, range, key, force, branchLog);
}

// The same for a filter of at most 64 tiles (N <= 65 536: the reference's own 10 000 is ten tiles), by ONE wavefront: no
// LDS round trips for the sums, no barrier but plan_scatter's.  Batches of such filters take one 64-thread workgroup per
// filter (k_locb_decide_tiny) instead of a 256-thread one of which three wavefronts add zeros: 14.6 -> ? us per deposit of
// 1600 filters.  Bit-identical to decide_small_body: the weight totals are exact sums (any order), and the other six sums
// reduce over the same xor tree of wavefront 0, to which the block version only adds the three zeros of the idle
// wavefronts.  Called by a workgroup of exactly 64 threads.
__device__ __forceinline__ void decide_tiny_body(LocCtl* __restrict__ ctl, const double* __restrict__ tileSums,
                                                 double* __restrict__ tileIncl, double* __restrict__ chunkSums,
                                                 double* __restrict__ chunkBase, DestPlan* __restrict__ plan,
                                                 int numTiles, long long n, float range, const RngKey& key,
                                                 int force, unsigned char* __restrict__ branchLog)
{
    __shared__ double cs[kSumsPerTile];
    __shared__ double inclS[64];
    __shared__ double base[2];
    __shared__ double dec[4];
    __shared__ PlanLds planLds;
    const int tid = threadIdx.x;
    double v[7];
#pragma unroll
    for (int k = 0; k < 7; ++k) v[k] = tid < numTiles ? tileSums[(size_t)tid * kSumsPerTile + k] : 0.0;
    const float u01 = decide_uniform(key);                  // while the sums are on their way
    const double incl = wave_inclusive_max_nonneg(wave_inclusive_scan(v[0]));
    if (tid < numTiles) tileIncl[tid] = incl;
    inclS[tid] = incl;
#pragma unroll
    for (int k = 1; k < 7; ++k) v[k] = wave_sum(v[k]);
    const double total = __shfl(incl, 63, 64);              // == the last prefix, not a re-sum
    if (tid == 0) {
        cs[0] = total; chunkSums[0] = total;
#pragma unroll
        for (int k = 1; k < 7; ++k) { cs[k] = v[k]; chunkSums[k] = v[k]; }
        // the chunk prefix of a single chunk (chunk_prefix with numChunks = 1: base 0, then the total)
        const double b0 = 0.0 < total ? 0.0 : total;        // (as chunk_prefix caps it: 0 unless the total is 0 or NaN)
        base[0] = b0; base[1] = total; chunkBase[0] = b0; chunkBase[1] = total;
        const double tot[7] = { total, v[1], v[2], v[3], v[4], v[5], v[6] };
        const Decision d = decide_compute(tot, n, range, u01, force);
        dec[0] = d.resample ? 1.0 : 0.0; dec[1] = d.rStart; dec[2] = d.rStep; dec[3] = d.S;
        decide_commit(ctl, d, branchLog);
        if (!d.resample) {
            ctl->pose[0] = tot[0]; ctl->pose[1] = tot[4]; ctl->pose[2] = tot[5]; ctl->pose[3] = tot[6];
            ctl->pose[4] = tot[2]; ctl->pose[5] = tot[3];
        }
        ctl->shardBase = 0.0; ctl->shardEnd = tot[0];
        ctl->outBegin = 0; ctl->outEnd = n; ctl->selfBegin = 0; ctl->selfEnd = n; ctl->nGlobal = n;
    }
    __syncthreads();
    if (dec[0] == 0.0) return;
    plan_scatter(0, base, inclS, plan, numTiles, n, dec[1], dec[2], dec[3], planLds);
}

// ------------------------------------------------------------------ resample
// Staging area for offspring whose place is on another shard (sharded filters only; null otherwise): one
// (x, y, z, theta) record per offspring, those below this shard's own index range first, then those above it --
// one contiguous message per neighbour.
struct StagePtrs { float4* rec; long long cap; };

// Structure (measured, MI355X, N = 16 M): a version that gathered the ancestors from global memory with one dword
// load per lane and stored one dword per lane and array ran at 0.14 ms whatever was left of its arithmetic; the
// access shape was the bound.  Here every byte moves as a float4: the tile's w,x,y,z,theta are loaded with five
// float4 loads per thread issued together (one memory latency per workgroup), the ancestors wait in LDS, and each
// thread produces FOUR CONSECUTIVE offspring, aligned on a 4-element boundary of the destination arrays, so that
// it stores five float4.  The first and last group of a tile (shared with the neighbouring workgroups) and
// offspring that go to another shard's staging area take dword stores.
// POSE (unsharded filters, when the kernel is launched because a pose was asked for): the six pose sums of the
// offspring (particleFilter.c:263-273 with w = 1) are accumulated on the way and left per workgroup in rsPart, which
// saves k_loc_pose_tiles its pass over the particles just written.
template <bool RECORD, bool POSE>
__global__ __launch_bounds__(kThreads) void k_loc_resample(LocBufs bufs, LocCtl* __restrict__ ctl,
                                                          const double* __restrict__ chunkBase,
                                                          const double* __restrict__ tileIncl,
                                                          int* __restrict__ ancOut,
                                                          int numTiles, long long n, StagePtrs stage, RngKey key,
                                                          float bx, float by, float bz, float range, float stdRange,
                                                          double* __restrict__ rsPart)
{
    if (ctl->branch != 1) return;           // normalise-only step: nothing to do (wave-uniform exit)

    __shared__ double cdf[kTile];
    __shared__ double park[POSE ? 6 * 64 : 1];
    __shared__ float4 anc[kTile];           // x y z theta of the tile's particles
    __shared__ double red[4];
    __shared__ long long span[2];
    const int t = blockIdx.x, tid = threadIdx.x;
    const int src = ctl->src, dst = ctl->dst;
    const long long tbase = (long long)t * kTile;
    const long long e0 = tbase + tid * kPerThread;
    const float4 w4 = *reinterpret_cast<const float4*>(bufs.p(src, 0) + e0);
    const float4 x4 = *reinterpret_cast<const float4*>(bufs.p(src, 1) + e0);
    const float4 y4 = *reinterpret_cast<const float4*>(bufs.p(src, 2) + e0);
    const float4 z4 = *reinterpret_cast<const float4*>(bufs.p(src, 3) + e0);
    const float4 t4 = *reinterpret_cast<const float4*>(bufs.p(src, 4) + e0);

    const double rStart = ctl->rStart, rStep = ctl->rStep;
    const double shardBase = ctl->shardBase, shardEnd = ctl->shardEnd;
    const double tileStart = tile_start_cdf(chunkBase, tileIncl, t, shardBase, shardEnd);
    const double tileEnd = tile_start_cdf(chunkBase, tileIncl, t + 1, shardBase, shardEnd);
    // Offspring range [i0, i1) of this tile.  Neighbouring workgroups (and neighbouring shards) evaluate the
    // same expression for a shared boundary, so the ranges tile [outBegin, outEnd) exactly.  One lane each of
    // two different waves does the float64 division; everybody else picks the result up after the barrier.
    if (tid == 0) {
        span[0] = (t == 0) ? ctl->outBegin : first_offspring_at(tileStart, rStart, rStep, ctl->nGlobal);
    } else if (tid == 64) {
        span[1] = (t == numTiles - 1) ? ctl->outEnd : first_offspring_at(tileEnd, rStart, rStep, ctl->nGlobal);
    }

    // this tile's CDF, rebuilt from the weights (pfResample.c:45,49): integer offsets from tileStart or float64 values
    const long long rem = n - tbase;
    const int nValid = rem < kTile ? (int)rem : kTile;
    const float wq[4] = { w4.x, w4.y, w4.z, w4.w };
    anc[tid * 4 + 0] = make_float4(x4.x, y4.x, z4.x, t4.x);
    anc[tid * 4 + 1] = make_float4(x4.y, y4.y, z4.y, t4.y);
    anc[tid * 4 + 2] = make_float4(x4.z, y4.z, z4.z, t4.z);
    anc[tid * 4 + 3] = make_float4(x4.w, y4.w, z4.w, t4.w);
    const bool small = build_tile_cdf(wq, tileStart, tileEnd, ctl->S, nValid, cdf, red);
    const long long i0 = span[0], i1 = span[1];
    if (POSE && t == 0 && tid == 0) ctl->rsValid = 1;   // nobody in this launch reads it; k_loc_pose_tiles then exits
    if (i0 >= i1) {
        if (POSE && tid < 6) rsPart[(size_t)t * kRsPerTile + tid] = 0.0;
        return;
    }

    double pacc[6] = { 0, 0, 0, 0, 0, 0 };
    const float hTheta = ctl->htheta;
    const long long numSpawn = ctl->numSpawn;
    const long long selfBegin = ctl->selfBegin, selfEnd = ctl->selfEnd, outBegin = ctl->outBegin;
    const long long outEnd = ctl->outEnd;
    // staged offspring: those below the shard's own range first, then those above it
    const long long below = selfBegin > outBegin ? (selfBegin < outEnd ? selfBegin : outEnd) - outBegin : 0;
    const long long aboveStart = selfEnd > outBegin ? selfEnd : outBegin;

    // Offspring i = cb + k with k a 32-bit offset inside the chunk [cb, ce): everything per offspring is 32-bit, the
    // 64-bit parts are wave-uniform.  (One tile owns more than 2^30 offspring only if nearly all the weight of a huge
    // sharded filter sits in it; the outer loop covers that.)
    constexpr long long kSpan = 1ll << 30;
    for (long long cb = i0; cb < i1; cb += kSpan) {
        const long long ce = (i1 - cb < kSpan) ? i1 : cb + kSpan;
        const int cnt = (int)(ce - cb);
        auto clampk = [&](long long v) -> int { v -= cb; return v < 0 ? 0 : (v > cnt ? cnt : (int)v); };
        const int kSpawn = clampk(numSpawn), kSelfB = clampk(selfBegin), kSelfE = clampk(selfEnd);
        const double dCb = (double)cb;                      // exact (< 2^53), and so is dCb + k
        const long long so = cb - selfBegin;                // element of the own arrays that offspring cb goes to
        // groups of four offspring aligned on the destination arrays: element (so + k) % 4 == 0 starts a group
        const int lead = (int)(((so % 4) + 4) % 4);         // elements of the first group that precede cb
        const int groups = (cnt + lead + 3) / 4;
        float* __restrict__ dw = bufs.p(dst, 0) + so;
        float* __restrict__ dx = bufs.p(dst, 1) + so;
        float* __restrict__ dy = bufs.p(dst, 2) + so;
        float* __restrict__ dz = bufs.p(dst, 3) + so;
        float* __restrict__ dt = bufs.p(dst, 4) + so;
        int* __restrict__ da = RECORD ? ancOut + so : nullptr;
        const long long lo = cb - outBegin, hi = cb - aboveStart + below;
        const int fullB = kSelfB > 0 ? kSelfB : 0, fullE = kSelfE < cnt ? kSelfE : cnt;
        for (int g = tid; g < groups; g += kThreads) {
            const int k0 = 4 * g - lead;                    // offsets k0 .. k0+3 (k0 may be negative in group 0)
            int jj[4];
#pragma unroll
            for (int m = 0; m < 4; ++m) {
                const double u = rStart + rStep * (dCb + (double)(k0 + m));     // pfResample.c:71 with i = cb + k
                jj[m] = tile_ancestor(cdf, small, nValid, tileStart, u);
            }
            float ox[4], oy[4], oz[4], ot[4];
#pragma unroll
            for (int m = 0; m < 4; ++m) {
                const int k = k0 + m;
                ox[m] = oy[m] = oz[m] = ot[m] = 0.0f;
                if (k < 0 || k >= cnt) continue;
                const uint64_t i = (uint64_t)(cb + k);
                if (k < kSpawn) {
                    spawn_from_range(key, i, bx, by, bz, range, stdRange, ox[m], oy[m], oz[m], ot[m]);  // pfResample.c:75-76
                } else {
                    const float4 a = anc[jj[m]];
                    spawn_from_other(key, i, a.x, a.y, a.z, a.w, kHxyz, hTheta, ox[m], oy[m], oz[m], ot[m]);
                }
                if (POSE) {
                    float sn, cs;
                    sincos_bounded(ot[m], sn, cs);
                    pacc[0] += ox[m]; pacc[1] += oy[m]; pacc[2] += oz[m]; pacc[3] += cs; pacc[4] += sn; pacc[5] += 1.0;
                }
            }
            if (k0 >= fullB && k0 + 3 < fullE) {
                *reinterpret_cast<float4*>(dw + k0) = make_float4(1.0f, 1.0f, 1.0f, 1.0f);
                *reinterpret_cast<float4*>(dx + k0) = make_float4(ox[0], ox[1], ox[2], ox[3]);
                *reinterpret_cast<float4*>(dy + k0) = make_float4(oy[0], oy[1], oy[2], oy[3]);
                *reinterpret_cast<float4*>(dz + k0) = make_float4(oz[0], oz[1], oz[2], oz[3]);
                *reinterpret_cast<float4*>(dt + k0) = make_float4(ot[0], ot[1], ot[2], ot[3]);
                if (RECORD) {
#pragma unroll
                    for (int m = 0; m < 4; ++m) da[k0 + m] = (int)(tbase + jj[m]);
                }
            } else {
#pragma unroll
                for (int m = 0; m < 4; ++m) {
                    const int k = k0 + m;
                    if (k < 0 || k >= cnt) continue;
                    if (k >= kSelfB && k < kSelfE) {
                        dw[k] = 1.0f; dx[k] = ox[m]; dy[k] = oy[m]; dz[k] = oz[m]; dt[k] = ot[m];
                        if (RECORD) da[k] = (int)(tbase + jj[m]);
                    } else {
                        const long long p = (k < kSelfB ? lo : hi) + k;
                        if (p >= stage.cap) continue;   // shard imbalance beyond the staging area: the host reports it
                        stage.rec[p] = make_float4(ox[m], oy[m], oz[m], ot[m]);
                    }
                }
            }
        }
    }
    if (POSE) {
        const double r = block_sum_quads<6>(pacc, park);
        if (tid < 6) rsPart[(size_t)t * kRsPerTile + tid] = r;
    }
}

// ------------------------------------------------------------------ deferred resampling + the next deposit
// A deposit whose decision was "resample" does not have to write its offspring before the next deposit is known:
// the unsharded filter defers the kernel, and when the next deposit arrives this kernel produces the offspring
// (exactly as k_loc_resample does: same CDF, same searches, same Philox counters of the deciding deposit, keyRs)
// and applies the new deposit to each one while it is still in registers -- VIO motion (MOTION) or the heading wrap
// of a zero motion, the range gate, and the seven weighted sums -- instead of writing 20 B, reading them back and
// writing again in k_loc_stream.  The values stored are bit for bit those of the two-kernel sequence.
//
// Sums.  The next CDF needs the sums per DESTINATION tile, but a workgroup owns a SOURCE tile and its offspring
// span [i0, i1) crosses destination tiles anywhere.  The span is walked destination tile by destination tile (a
// thread's group of four offspring never straddles one); each piece (source tile t, destination tile d) is reduced
// on its own and stored in slot t + d -- unique and increasing along the walk, because every step advances t or d.
// Workgroups with an empty span store a zero piece so that the slots between a tile's first and last piece are
// all written; the workgroups that produce a tile's first and last offspring record those slots.  k_loc_chunks
// adds the pieces in slot order: deterministic, unlike atomics.
struct RespawnShell { float bx, by, bz, range, stdRange; };     // pfInit.c:56-66 arguments of the deciding deposit
struct PieceOut { double* sums; int* headSlot; int* tailSlot; };

struct ResampleDepositLds {
    double cdf[kTile];
    float4 anc[kTile];
    double park[7 * 64];
    double red[4];
    long long span[2];
};

// SHARDED: the filter is one shard of a larger one (global indices selfBegin + local element).  Offspring whose index
// falls outside the shard's own range are not deposited on here: they go to the staging area as they come out of the
// resampling, and the shard that owns them deposits on them when they land (k_loc_land).
// STAGE_ONLY (k_loc_stage_foreign): source tile t's offspring that belong to other shards, and nothing else.
template <bool MOTION, bool SHARDED, bool STAGE_ONLY = false>
__device__ __forceinline__ void resample_deposit_tile(const LocBufs& bufs, const LocCtl* __restrict__ ctl,
                                                      const double* __restrict__ chunkBase,
                                                      const double* __restrict__ tileIncl, int numTiles, int n,
                                                      const RngKey& keyRs, const RespawnShell& sh, const RngKey& key,
                                                      const Motion& mo, const Gate& ga, const PieceOut& out,
                                                      const StagePtrs& stage, ResampleDepositLds& L, int t)
{
    double* const cdf = L.cdf;
    float4* const anc = L.anc;
    double* const park = L.park;
    double* const red = L.red;
    long long* const span = L.span;
    double* __restrict__ pieceSums = out.sums;
    int* __restrict__ headSlot = out.headSlot;
    int* __restrict__ tailSlot = out.tailSlot;
    const float sbx = sh.bx, sby = sh.by, sbz = sh.bz, srange = sh.range, sstd = sh.stdRange;
    const int tid = threadIdx.x;
    const int src = ctl->src, dst = ctl->dst;
    const int tbase = t * kTile;
    const int e0 = tbase + tid * kPerThread;
    const float4 w4 = *reinterpret_cast<const float4*>(bufs.p(src, 0) + e0);
    const float4 x4 = *reinterpret_cast<const float4*>(bufs.p(src, 1) + e0);
    const float4 y4 = *reinterpret_cast<const float4*>(bufs.p(src, 2) + e0);
    const float4 z4 = *reinterpret_cast<const float4*>(bufs.p(src, 3) + e0);
    const float4 t4 = *reinterpret_cast<const float4*>(bufs.p(src, 4) + e0);

    const double rStart = ctl->rStart, rStep = ctl->rStep;
    const double shardBase = ctl->shardBase, shardEnd = ctl->shardEnd;     // [0, S) unless the filter is sharded
    const double tileStart = tile_start_cdf(chunkBase, tileIncl, t, shardBase, shardEnd);
    const double tileEnd = tile_start_cdf(chunkBase, tileIncl, t + 1, shardBase, shardEnd);
    const long long gOff = SHARDED ? ctl->selfBegin : 0;                    // global index of local element 0
    const long long nGlobal = SHARDED ? ctl->nGlobal : n;
    const long long outBegin = SHARDED ? ctl->outBegin : 0, outEnd = SHARDED ? ctl->outEnd : n;
    if (tid == 0) {
        span[0] = (t == 0) ? outBegin : first_offspring_at(tileStart, rStart, rStep, nGlobal);
    } else if (tid == 64) {
        span[1] = (t == numTiles - 1) ? outEnd : first_offspring_at(tileEnd, rStart, rStep, nGlobal);
    }

    const int rem = n - tbase;
    const int nValid = rem < kTile ? rem : kTile;
    const float wq[4] = { w4.x, w4.y, w4.z, w4.w };
    anc[tid * 4 + 0] = make_float4(x4.x, y4.x, z4.x, t4.x);
    anc[tid * 4 + 1] = make_float4(x4.y, y4.y, z4.y, t4.y);
    anc[tid * 4 + 2] = make_float4(x4.z, y4.z, z4.z, t4.z);
    anc[tid * 4 + 3] = make_float4(x4.w, y4.w, z4.w, t4.w);
    const bool small = build_tile_cdf(wq, tileStart, tileEnd, ctl->S, nValid, cdf, red);
    const long long s0 = span[0], s1 = span[1];         // global offspring span of this source tile
    auto to_local = [&](long long g) -> int { g -= gOff; return g < 0 ? 0 : (g > n ? n : (int)g); };
    const int i0 = to_local(s0), i1 = to_local(s1);     // the part of it that lands in this shard's own arrays
    const int loL = to_local(outBegin), hiL = to_local(outEnd);     // local elements produced by this shard at all
    const float hTheta = ctl->htheta;
    const int numSpawn = ctl->numSpawn;
    const double dOff = (double)gOff;

    if (SHARDED) {
        // foreign offspring: below the own range first, then above it, one record each
        const long long selfEnd = gOff + n;
        const long long below = gOff > outBegin ? (gOff < outEnd ? gOff : outEnd) - outBegin : 0;
        const long long aboveStart = selfEnd > outBegin ? selfEnd : outBegin;
#pragma unroll 1
        for (int seg = 0; seg < 2; ++seg) {
            const long long f0 = seg == 0 ? s0 : (s0 > selfEnd ? s0 : selfEnd);
            const long long f1 = seg == 0 ? (s1 < gOff ? s1 : gOff) : s1;
            for (long long i = f0 + tid; i < f1; i += kThreads) {
                const double u = rStart + rStep * (double)i;
                const int j = tile_ancestor(cdf, small, nValid, tileStart, u);
                float x, y, z, th;
                if (i < numSpawn) {
                    spawn_from_range(keyRs, (uint64_t)i, sbx, sby, sbz, srange, sstd, x, y, z, th);
                } else {
                    const float4 a = anc[j];
                    spawn_from_other(keyRs, (uint64_t)i, a.x, a.y, a.z, a.w, kHxyz, hTheta, x, y, z, th);
                }
                const long long p = seg == 0 ? i - outBegin : i - aboveStart + below;
                if (p < stage.cap) stage.rec[p] = make_float4(x, y, z, th);     // beyond it: the host has reported the imbalance
            }
        }
    }

    if (STAGE_ONLY) return;
    if (i0 >= i1) {
        // empty span at offspring position i0: a zero piece in the slot the walk would have used there
        if (tid < 7) pieceSums[(size_t)(t + i0 / kTile) * kSumsPerTile + tid] = 0.0;
        return;
    }

    const bool smallTurn = mo.stdTheta <= 1e-4f;
    float* __restrict__ dw = bufs.p(dst, 0);
    float* __restrict__ dx = bufs.p(dst, 1);
    float* __restrict__ dy = bufs.p(dst, 2);
    float* __restrict__ dz = bufs.p(dst, 3);
    float* __restrict__ dt = bufs.p(dst, 4);

    // thread k < 7 stores sum k of the piece (this source tile, destination tile d); thread 0 notes whether the
    // piece holds the tile's first / last offspring
    auto store_piece = [&](int d, double v) {
        const int slot = t + d;
        if (tid < 7) pieceSums[(size_t)slot * kSumsPerTile + tid] = v;
        if (tid == 0) {
            // first / last element of tile d that this shard produces itself ([loL, hiL) is all of it when unsharded)
            int first = d * kTile;
            int last = (first + kTile < n ? first + kTile : n) - 1;
            first = first > loL ? first : loL;
            last = last < hiL - 1 ? last : hiL - 1;
            if (i0 <= first && first < i1) headSlot[d] = slot;
            if (i0 <= last && last < i1) tailSlot[d] = slot;
        }
    };

    // Groups of four offspring aligned on the destination arrays, dealt to the threads densely from the start of
    // the span (group g0 + tid), 1024 offspring per round.  A round touches at most two destination tiles, dLo and
    // dHi = dLo + 1: their sums are reduced separately, and a tile that continues into the next round is carried.
    const int d1 = (i1 - 1) / kTile;
    double carry = 0.0;
    int carryTile = -1;
    for (int g0 = i0 >> 2; 4 * g0 < i1; g0 += kThreads) {
        const int e = 4 * (g0 + tid);                       // first offspring of this thread's group
        double acc[7] = { 0, 0, 0, 0, 0, 0, 0 };
        if (e < i1) {
            int jj[4];
#pragma unroll
            for (int m = 0; m < 4; ++m) {
                const double u = rStart + rStep * (SHARDED ? dOff + (double)(e + m) : (double)(e + m));     // pfResample.c:71
                jj[m] = tile_ancestor(cdf, small, nValid, tileStart, u);
            }
            float ow[4], ox[4], oy[4], oz[4], ot[4];
#pragma unroll
            for (int m = 0; m < 4; ++m) {
                const int i = e + m;
                ow[m] = ox[m] = oy[m] = oz[m] = ot[m] = 0.0f;
                if (i < i0 || i >= i1) continue;
                const uint64_t gi = (uint64_t)(gOff + i);   // global index: the Philox counter
                if ((long long)gi < numSpawn) {
                    spawn_from_range(keyRs, gi, sbx, sby, sbz, srange, sstd, ox[m], oy[m], oz[m], ot[m]);
                } else {
                    const float4 a = anc[jj[m]];
                    spawn_from_other(keyRs, gi, a.x, a.y, a.z, a.w, kHxyz, hTheta, ox[m], oy[m], oz[m], ot[m]);
                }
                // the new deposit on the fresh offspring (weight 1): the body of k_loc_stream
                float wi = 1.0f, thi = ot[m], sn, cs;
                if (MOTION) {
                    motion_particle(key, gi, mo, smallTurn, ox[m], oy[m], oz[m], thi, sn, cs);
                } else {
                    thi = wrap_two_pi(thi);
                    sincos_bounded(thi, sn, cs);
                }
                ot[m] = thi;
                if (gate_rejects(ga, ox[m] - ga.bx, oy[m] - ga.by, oz[m] - ga.bz)) wi *= ga.minWeight;
                ow[m] = wi;
                acc[0] += cdf_weight(wi);
                acc[1] += wi * wi;
                acc[2] += wi * cs;
                acc[3] += wi * sn;
                acc[4] += wi * ox[m];
                acc[5] += wi * oy[m];
                acc[6] += wi * oz[m];
            }
            if (e >= i0 && e + kPerThread <= i1) {
                *reinterpret_cast<float4*>(dw + e) = make_float4(ow[0], ow[1], ow[2], ow[3]);
                *reinterpret_cast<float4*>(dx + e) = make_float4(ox[0], ox[1], ox[2], ox[3]);
                *reinterpret_cast<float4*>(dy + e) = make_float4(oy[0], oy[1], oy[2], oy[3]);
                *reinterpret_cast<float4*>(dz + e) = make_float4(oz[0], oz[1], oz[2], oz[3]);
                *reinterpret_cast<float4*>(dt + e) = make_float4(ot[0], ot[1], ot[2], ot[3]);
            } else {
#pragma unroll
                for (int m = 0; m < 4; ++m) {
                    const int i = e + m;
                    if (i < i0 || i >= i1) continue;
                    dw[i] = ow[m]; dx[i] = ox[m]; dy[i] = oy[m]; dz[i] = oz[m]; dt[i] = ot[m];
                }
            }
        }
        const int dLo = (4 * g0) / kTile;
        int dHi = (4 * (g0 + kThreads - 1)) / kTile;
        dHi = dHi < d1 ? dHi : d1;                          // groups past the span carry zeros
        const int myTile = e / kTile;
        double v[7];
#pragma unroll
        for (int k = 0; k < 7; ++k) v[k] = (myTile == dLo) ? acc[k] : 0.0;
        double lo = block_sum_quads<7>(v, park);
        if (carryTile == dLo) lo = carry + lo;
        else if (carryTile >= 0) store_piece(carryTile, carry);     // the previous round ended on a tile boundary
        if (dHi != dLo) {
            store_piece(dLo, lo);
#pragma unroll
            for (int k = 0; k < 7; ++k) v[k] = (myTile == dLo) ? 0.0 : acc[k];
            carry = block_sum_quads<7>(v, park);
            carryTile = dHi;
        } else {
            carry = lo;
            carryTile = dLo;
        }
    }
    store_piece(carryTile, carry);
}

// ------------------------------------------------------------------ destination-owned resampling (unsharded filters)
// The kernels above give a workgroup a SOURCE tile and let it produce however many offspring that tile's weight calls
// for: simple, and what a sharded filter needs (offspring are produced where their ancestors live), but the work per
// workgroup is as uneven as the weights -- a tile worth 1100 offspring runs a second round for 76 of them, a tile of
// gated-out particles idles -- and the sums per destination tile have to be pieced together afterwards.  An unsharded
// filter can turn it round: workgroup d produces offspring [1024 d, 1024 d + 1024), whatever ancestors they have.
//   * the source tile that holds the ancestor of its first unresolved offspring is found by one wavefront with a
//     64-ary search over the chunk bases and the in-chunk tile prefixes (2 to 4 dependent loads);
//   * that tile's CDF is rebuilt in LDS from its weights (4 B per source particle; integer offsets, build_tile_cdf)
//     and every offspring whose u_i lies below the tile's end value finds its ancestor in it; typically two source
//     tiles cover a destination tile, a heavy ancestor region takes one, a stretch of light particles several;
//   * the ancestors' x, y, z, theta are gathered straight from global memory (sorted indices: a wavefront's gathers
//     fall into a few cache lines), the offspring are formed and, in the deferred form, deposited on, exactly as in
//     resample_deposit_tile; all stores are aligned float4, the seven sums of the tile are reduced once and written
//     where the streaming pass would have written them.
// Every workgroup does the same amount of arithmetic (1024 offspring), which is what the kernel is bound by.
// Selection rule, Philox counters and arithmetic are those of k_loc_resample: same particles, bit for bit.
template <typename Pred>
__device__ __forceinline__ int wave_first_true(int a, int b, Pred pred)
{   // first index in [a, b) for which the monotone predicate holds (false ... false true ... true), b if none;
    // called by all 64 lanes of one wavefront
    const int lane = threadIdx.x & 63;
    for (;;) {
        const int len = b - a;
        if (len <= 0) return b;
        if (len <= 64) {
            const bool p = (lane < len) && pred(a + lane);
            const unsigned long long m = __ballot(p);
            return m ? a + (int)__builtin_ctzll(m) : b;
        }
        const int stride = (len + 63) / 64;
        int idx = a + (lane + 1) * stride - 1;              // last index of sub-range `lane`
        idx = idx < b ? idx : b - 1;
        const unsigned long long m = __ballot(pred(idx));
        if (!m) return b;
        const int sub = (int)__builtin_ctzll(m);
        int last = a + (sub + 1) * stride - 1;
        last = last < b ? last : b - 1;                     // the predicate holds there
        a += sub * stride;
        b = last;                                           // "none in [a, last)" then means `last` itself
    }
}

// the source tile whose CDF span [tile start, tile end) holds position u: first tile whose end value exceeds u
// (the last tile if there is none: offspring that fall off the end, and every offspring of a filter without weight)
// cdfBase, cdfEnd: 0 and the total weight, or a shard's span of the global CDF
__device__ __forceinline__ int find_source_tile(const double* __restrict__ chunkBase, const double* __restrict__ tileIncl,
                                                int numChunks, int numTiles, double cdfBase, double cdfEnd, double u)
{
    int c = wave_first_true(0, numChunks, [&](int k) { return chunkBase[k + 1] + cdfBase > u; });
    if (c >= numChunks) return numTiles - 1;
    const int t0 = c * kChunkTiles, t1 = (t0 + kChunkTiles < numTiles) ? t0 + kChunkTiles : numTiles;
    const int t = wave_first_true(t0, t1, [&](int k) { return tile_start_cdf(chunkBase, tileIncl, k + 1, cdfBase, cdfEnd) > u; });
    return t < t1 ? t : t1 - 1;
}

#ifndef PF_OFFSPRING_UNROLL
#define PF_OFFSPRING_UNROLL 4
#endif
#ifndef PF_SCHED_BARRIER_ALL
#define PF_SCHED_BARRIER_ALL 0
#endif
#ifndef PF_DEPOSIT_TILES
#define PF_DEPOSIT_TILES 1
#endif
#ifndef PF_NOISE_AHEAD1
#define PF_NOISE_AHEAD1 1
#endif
#ifndef PF_NOISE_AHEAD2
#define PF_NOISE_AHEAD2 1
#endif
constexpr int kDepositTiles = PF_DEPOSIT_TILES;     // destination tiles per workgroup of k_loc_deposit
#ifndef PF_XCD_REMAP
#define PF_XCD_REMAP 0
#endif
// Workgroups are dealt to the eight XCDs round robin (workgroup b runs on XCD b % 8), each XCD has an L2 of its own.  With
// tile = workgroup index, neighbouring destination tiles -- which draw from the same source tiles -- run on different
// XCDs and every L2 fetches those source tiles for itself.  This deals each XCD a contiguous eighth of the tiles instead.
// Measured (-DPF_XCD_REMAP=1, same box, N = 16 M): WORSE -- resampling side 0.149 -> 0.164 ms, static streaming side
// 0.078 -> 0.081: eight distant regions of ten arrays each are open at once instead of one (address translation reach,
// presumably), and that costs more than the shared source tiles save.  PF_XCD_REMAP=2 (eight consecutive tiles per XCD
// inside every group of 64, so that all XCDs stay in one region): resampling side unchanged (0.152 / 0.152), streaming
// sides +1.5 % / +4 % -- the memory likes the tiles in plain order.  Off.
__device__ __forceinline__ int xcd_tile(int b, int numTiles)
{
#if PF_XCD_REMAP == 2
    // inside every group of 64 tiles XCD x takes the eight consecutive tiles 8 x .. 8 x + 7: neighbours share an L2, and all
    // XCDs stay in the same region of the arrays
    const int g = b & ~63;
    if (g + 64 > numTiles) return b;
    return g + ((b & 7) << 3) + ((b >> 3) & 7);
#else
    const int x = b & 7, idx = b >> 3, per = numTiles >> 3, rem = numTiles & 7;
    return x * per + (x < rem ? x : rem) + idx;
#endif
}
struct DestLds {
    double tab[kTile];          // CDF of the source tile at hand (integer form: the first half)
    double park[7 * 64];
    double red[4];
};

// NSUB: destination tiles per workgroup (consecutive ones: blockIdx.x * NSUB + s).  With two, the dependent loads in front
// of the arithmetic (record -> weights -> ancestors) are paid once per 2048 offspring, the second tile's ancestors are in
// flight while the first tile is computed, and three source tiles are scanned per workgroup instead of four.
// SHARDED (one shard of a sharded filter, k_loc_deposit_shard): the tiles are the shard's own, global offspring index =
// selfBegin + local element (positions u, Philox counters and the respawn test use it), CDF values are global
// (shardBase + local prefix), and only the elements [loL, hiL) -- the offspring of local ancestors that land in the
// shard's own arrays -- are produced: the rest of a boundary tile is left zero here and written by k_loc_land when the
// neighbours' records arrive, and its sums are a partial sum (k_loc_chunks adds what k_loc_land summed).
template <int NSUB, bool DEPOSIT, bool MOTION, bool RECORD, bool POSE, bool SHARDED = false>
__device__ __forceinline__ void resample_dest_tile(const LocBufs& bufs, const LocCtl* __restrict__ ctl,
                                                   const double* __restrict__ chunkBase, const double* __restrict__ tileIncl,
                                                   int numTiles, int n, const RngKey& keyRs, const RespawnShell& sh,
                                                   const RngKey& key, const Motion& mo, const Gate& ga,
                                                   double* __restrict__ tileSums, int* __restrict__ ancOut,
                                                   double* __restrict__ rsPart, const DestPlan& P, DestLds& L, int d0)
{
    const int tid = threadIdx.x;
    const int src = ctl->src, dst = ctl->dst;
    const double rStart = ctl->rStart, rStep = ctl->rStep, S = ctl->S;
    const float hTheta = ctl->htheta;
    const int numSpawn = ctl->numSpawn;
    const int numChunks = (numTiles + kChunkTiles - 1) / kChunkTiles;
    const int lo = d0 * kTile;
    const int cnt = (n - lo < NSUB * kTile) ? n - lo : NSUB * kTile;    // offspring of this workgroup that exist
    const int k0 = tid * kPerThread;        // the thread's offspring: lo + s * kTile + k0 + m, s < NSUB, m < 4
    static_assert(!SHARDED || (NSUB == 1 && !RECORD && !POSE), "the sharded form is the plain fused deposit");
    const long long gOff = SHARDED ? ctl->selfBegin : 0;                // global index of local element 0
    // A sharded filter has at most 16 x 2^28 = 2^32 particles (pfx_loc_shard_config checks it): global indices fit 32 bits
    // unsigned, and the sharded form differs from the unsharded one in nothing but a uniform offset.
    const uint32_t gLo = (uint32_t)(gOff + lo);                         // global index of the workgroup's first offspring
    auto idx = [&](int k) -> uint64_t { return SHARDED ? (uint64_t)(gLo + (uint32_t)k) : (uint64_t)(lo + k); };
    auto pos = [&](int k) -> double { return SHARDED ? (double)(gLo + (uint32_t)k) : (double)(lo + k); };
    // [pBegin, pEnd): the workgroup's offspring (relative to lo) that are produced here
    int pBegin = 0, pEnd = cnt;
    if (SHARDED) {
        auto rel = [&](long long g) -> int { g -= gOff + lo; return g < 0 ? 0 : (g > cnt ? cnt : (int)g); };
        pBegin = rel(ctl->outBegin);
        pEnd = rel(ctl->outEnd);
        pEnd = pEnd > pBegin ? pEnd : pBegin;
    }

    int jg[NSUB][4];
#pragma unroll
    for (int s = 0; s < NSUB; ++s)
#pragma unroll
        for (int m = 0; m < 4; ++m) jg[s][m] = 0;

    // offspring below numSpawn are respawned on the range shell and need no ancestor (pfResample.c:75-76), unless the
    // ancestors are being recorded (the reference selects one for them too, :70-72, before it overwrites them)
    int next;
    if (SHARDED) {
        const long long ns = (long long)numSpawn - gOff - lo;
        next = ns < pBegin ? pBegin : (ns > pEnd ? pEnd : (int)ns);
    } else {
        next = RECORD ? 0 : numSpawn - lo;
        next = next < 0 ? 0 : (next > cnt ? cnt : next);
    }
    const float* __restrict__ sw = bufs.p(src, 0);
    // Every wavefront works the tile sequence out for itself (uniform values, loads that hit the same lines): nothing
    // but the CDF table goes through LDS, and its two barriers per tile are the only ones of the loop.
    int tPrev = P.t0 - 1, tAhead = -1;
    float4 wAhead = make_float4(0.f, 0.f, 0.f, 0.f);        // weights of the tile after the current one, loaded ahead
    // The weights of the first two source tiles are asked for here, and while they are on their way the noise of the
    // thread's first offspring is drawn (Philox and Box-Muller depend on the counter only); the noise of its second
    // offspring is drawn while the ancestors are on their way, further down.  (kAhead offspring; NSUB == 1.)
    constexpr int kAhead1 = (NSUB == 1 && !RECORD) ? PF_NOISE_AHEAD1 : 0;     // ... drawn while the weights are on their way
    constexpr int kAhead = (NSUB == 1 && !RECORD) ? PF_NOISE_AHEAD1 + PF_NOISE_AHEAD2 : 0;
    float nj[kAhead > 0 ? kAhead : 1][4], nv[kAhead > 0 ? kAhead : 1][4];
    int tFirst = -1;
    float4 wFirst = make_float4(0.f, 0.f, 0.f, 0.f);
    if (kAhead > 0) {
        tFirst = P.t0;
        wFirst = ld4<PF_NT_RS_W>(sw + (size_t)tFirst * kTile + k0);
        if (tFirst + 1 < numTiles) {
            tAhead = tFirst + 1;
            wAhead = ld4<PF_NT_RS_W>(sw + (size_t)tAhead * kTile + k0);
        }
#pragma unroll
        for (int m = 0; m < kAhead1; ++m)
            if (k0 + m < cnt) {
                jitter_normals(keyRs, idx(k0 + m), nj[m]);
                if (DEPOSIT && MOTION) vio_normals(key, idx(k0 + m), nv[m]);
            }
    }
#ifdef PF_EXP_IDENTITY_ANC      // timing experiment: no selection at all, offspring k descends from particle k
#pragma unroll
    for (int s = 0; s < NSUB; ++s)
#pragma unroll
        for (int m = 0; m < 4; ++m) jg[s][m] = (lo + s * kTile + k0 + m < n) ? lo + s * kTile + k0 + m : 0;
    next = cnt;
#endif
    while (next < pEnd) {                                   // uniform over the workgroup
        // The source tile of offspring `next`: the first tile from tPrev + 1 on whose end value exceeds its position (the
        // last tile if there is none), and `upto`, how far that tile gets: the first offspring whose position is not
        // below the tile's end value (u is non-decreasing in k).  The record of this destination tile answers both for
        // the tiles t0 .. t0 + 3 with integer comparisons (tile t0 + i takes offspring `next` iff next < upto[i]); beyond
        // them the tile bounds are loaded, successors probed 64 at a time, and upto found by arithmetic on 64 lanes.
        int t = -1, upto = pEnd;
        double tileStart = 0.0, tileEnd = 0.0;
#pragma unroll
        for (int i = 0; i < kPlanEnds; ++i) {
            const int ti = P.t0 + i;
            if (t < 0 && ti > tPrev && (ti >= numTiles - 1 || P.upto[i] > lo + next)) {
                t = ti < numTiles - 1 ? ti : numTiles - 1;
                tileStart = (i == 0) ? P.start0 : P.end[i - 1];
                tileEnd = P.end[i];
                const int rel = P.upto[i] - lo;
                upto = (ti >= numTiles - 1 || rel > pEnd) ? pEnd : rel;
            }
        }
        if (t < 0) {
            // (the shard's span of the global CDF is read here, where it is needed: beyond the record's four tiles)
            const double cdfBase = SHARDED ? ctl->shardBase : 0.0, cdfEnd = SHARDED ? ctl->shardEnd : S;
            const double un = rStart + rStep * pos(next);
            const int a = (tPrev + 1 > P.t0 + kPlanEnds) ? tPrev + 1 : P.t0 + kPlanEnds;
            const int b = (a + 64 < numTiles) ? a + 64 : numTiles;
            t = wave_first_true(a, b, [&](int k) { return tile_start_cdf(chunkBase, tileIncl, k + 1, cdfBase, cdfEnd) > un; });
            if (t >= b) t = (b == numTiles) ? numTiles - 1 : find_source_tile(chunkBase, tileIncl, numChunks, numTiles, cdfBase, cdfEnd, un);
            tileStart = tile_start_cdf(chunkBase, tileIncl, t, cdfBase, cdfEnd);
            tileEnd = tile_start_cdf(chunkBase, tileIncl, t + 1, cdfBase, cdfEnd);
            upto = (t == numTiles - 1) ? pEnd
                                       : wave_first_true(next + 1, pEnd, [&](int k) { return !(rStart + rStep * pos(k) < tileEnd); });
        }
        const bool lastTile = t == numTiles - 1;
        const float4 w4 = (t == tFirst) ? wFirst : (t == tAhead) ? wAhead : ld4<PF_NT_RS_W>(sw + (size_t)t * kTile + k0);
        if (!lastTile && tAhead != t + 1) {                 // the next source tile is nearly always needed too
            tAhead = t + 1;
            wAhead = ld4<PF_NT_RS_W>(sw + (size_t)tAhead * kTile + k0);
        }
        const int rem = n - t * kTile;
        const int nValid = rem < kTile ? rem : kTile;
        const float wq[4] = { w4.x, w4.y, w4.z, w4.w };
        const bool small = build_tile_cdf(wq, tileStart, tileEnd, S, nValid, L.tab, L.red);
#pragma unroll
        for (int s = 0; s < NSUB; ++s) {
            if (NSUB == 1 || (upto > s * kTile && next < (s + 1) * kTile)) {            // uniform: tile s has offspring here
                // The thread's offspring are consecutive: after the first one of them in this tile, the next one's ancestor is
                // looked for from the previous one's on (three probes, then the halving search): -2.7 % on the launch in the
                // tracking scene, -1.6 .. -3.9 % under forced resampling, -2.4 % in the transient after the spawn.
                int prevJ = -1;                             // ancestor, inside this tile, of the thread's previous offspring
#pragma unroll
                for (int m = 0; m < 4; ++m) {
                    const int k = s * kTile + k0 + m;
                    if (k >= next && k < upto) {
                        const double u = rStart + rStep * pos(k);                       // pfResample.c:71
                        int j;
                        if (small && prevJ >= 0)
                            j = upper_bound_tile_u32_from(reinterpret_cast<const uint32_t*>(L.tab), nValid,
                                                          cvt_u32_sat((u - tileStart) * (double)kCdfScale), prevJ);
                        else
                            j = tile_ancestor(L.tab, small, nValid, tileStart, u);
                        prevJ = j;
                        jg[s][m] = t * kTile + j;
                    }
                }
            }
        }
        next = upto;
        tPrev = t;
        // (no barrier here: the next tile's table is written behind the barrier of its scan, which every thread reaches
        // only after it has finished searching this one)
    }

    const float* __restrict__ sx = bufs.p(src, 1);
    const float* __restrict__ sy = bufs.p(src, 2);
    const float* __restrict__ sz = bufs.p(src, 3);
    const float* __restrict__ st = bufs.p(src, 4);
    float ax[NSUB][4], ay[NSUB][4], az[NSUB][4], at[NSUB][4];
#pragma unroll
    for (int s = 0; s < NSUB; ++s)
#pragma unroll
        for (int m = 0; m < 4; ++m) {
            const int j = jg[s][m];                         // 0 for respawned and pad offspring: a harmless read
#ifdef PF_EXP_NO_GATHER         // timing experiment
            ax[s][m] = (float)j * 1e-7f; ay[s][m] = ax[s][m] + 1.0f; az[s][m] = 1.5f; at[s][m] = ax[s][m];
#else
            ax[s][m] = ld1<PF_NT_RS_GATHER>(sx + j); ay[s][m] = ld1<PF_NT_RS_GATHER>(sy + j);
            az[s][m] = ld1<PF_NT_RS_GATHER>(sz + j); at[s][m] = ld1<PF_NT_RS_GATHER>(st + j);
#endif
        }

#pragma unroll
    for (int m = kAhead1; m < kAhead; ++m)                  // while the ancestors are on their way
        if (k0 + m < cnt) {
            jitter_normals(keyRs, idx(k0 + m), nj[m]);
            if (DEPOSIT && MOTION) vio_normals(key, idx(k0 + m), nv[m]);
        }
    const bool smallTurn = mo.stdTheta <= 1e-4f;
#pragma unroll
    for (int s = 0; s < NSUB; ++s) {
        const int d = d0 + s;
        const int cntS = (cnt - s * kTile < kTile) ? cnt - s * kTile : kTile;   // <= 0: a second tile that does not exist
        if (NSUB == 1 || cntS > 0) {                        // uniform
        double acc[7] = { 0, 0, 0, 0, 0, 0, 0 };
        float ow[4], ox[4], oy[4], oz[4], ot[4];
#pragma unroll PF_OFFSPRING_UNROLL
        for (int m = 0; m < 4; ++m) {
            const int k = k0 + m;
            ow[m] = ox[m] = oy[m] = oz[m] = ot[m] = 0.0f;
            // pad elements of the last tile stay zero; so do, in a shard, those another shard produces (k_loc_land writes
            // them).  One unsigned comparison either way: a second condition here cost twelve spilled registers.
            if (SHARDED ? (unsigned)(k - pBegin) >= (unsigned)(pEnd - pBegin) : k >= cntS) continue;
            const int g = lo + s * kTile + k;
            const uint64_t gi = idx(s * kTile + k);         // global index: the Philox counter
            const bool ahead = m < kAhead;                  // this offspring's unit normals have been drawn already
            if (SHARDED ? (uint32_t)gi < (uint32_t)numSpawn : g < numSpawn) {
                spawn_from_range(keyRs, gi, sh.bx, sh.by, sh.bz, sh.range, sh.stdRange, ox[m], oy[m], oz[m], ot[m]);
            } else if (ahead) {
                spawn_from_other_n(nj[m < kAhead ? m : 0], ax[s][m], ay[s][m], az[s][m], at[s][m], kHxyz, hTheta, ox[m], oy[m], oz[m], ot[m]);
            } else {
                spawn_from_other(keyRs, gi, ax[s][m], ay[s][m], az[s][m], at[s][m], kHxyz, hTheta, ox[m], oy[m], oz[m], ot[m]);
            }
            float wi = 1.0f;
            if (DEPOSIT) {
                // the new deposit on the fresh offspring (weight 1): the body of k_loc_stream
                float thi = ot[m], sn, cs;
                if (MOTION && ahead) {
                    motion_particle_n(nv[m < kAhead ? m : 0], mo, smallTurn, ox[m], oy[m], oz[m], thi, sn, cs);
                } else if (MOTION) {
                    motion_particle(key, gi, mo, smallTurn, ox[m], oy[m], oz[m], thi, sn, cs);
                } else {
                    thi = wrap_two_pi(thi);
                    sincos_bounded(thi, sn, cs);
                }
                ot[m] = thi;
                if (gate_rejects(ga, ox[m] - ga.bx, oy[m] - ga.by, oz[m] - ga.bz)) wi *= ga.minWeight;
                acc[0] += cdf_weight(wi);
                acc[1] += wi * wi;
                acc[2] += wi * cs;
                acc[3] += wi * sn;
                acc[4] += wi * ox[m];
                acc[5] += wi * oy[m];
                acc[6] += wi * oz[m];
            } else if (POSE) {
                float sn, cs;
                sincos_bounded(ot[m], sn, cs);
                acc[0] += ox[m]; acc[1] += oy[m]; acc[2] += oz[m]; acc[3] += cs; acc[4] += sn; acc[5] += 1.0;
            }
            ow[m] = wi;
            // Without motion an offspring is little work and the scheduler runs all four at once: more values alive than
            // the 72 registers of seven waves hold (25 spilled).  One offspring after the other, then.
            if (DEPOSIT && (!MOTION || PF_SCHED_BARRIER_ALL)) __builtin_amdgcn_sched_barrier(0);
        }
        const size_t e = (size_t)lo + s * kTile + k0;
#ifdef PF_EXP_NO_STORE          // timing experiment: the results stay alive, nothing is written
        if (ow[0] + ox[1] + oy[2] + oz[3] + ot[0] == 12345.678f)
#endif
        {
            st4<PF_NT_RS_ST>(bufs.p(dst, 0) + e, make_float4(ow[0], ow[1], ow[2], ow[3]));
            st4<PF_NT_RS_ST>(bufs.p(dst, 1) + e, make_float4(ox[0], ox[1], ox[2], ox[3]));
            st4<PF_NT_RS_ST>(bufs.p(dst, 2) + e, make_float4(oy[0], oy[1], oy[2], oy[3]));
            st4<PF_NT_RS_ST>(bufs.p(dst, 3) + e, make_float4(oz[0], oz[1], oz[2], oz[3]));
            st4<PF_NT_RS_ST>(bufs.p(dst, 4) + e, make_float4(ot[0], ot[1], ot[2], ot[3]));
        }
        if (RECORD) {
#pragma unroll
            for (int m = 0; m < 4; ++m)
                if (k0 + m < cntS) ancOut[lo + s * kTile + k0 + m] = jg[s][m];
        }
        if (DEPOSIT) {
            const double r = (NSUB == 1) ? block_sum_quads_tail<7>(acc, L.park) : block_sum_quads<7>(acc, L.park);
            if (tid < 7) tileSums[(size_t)d * kSumsPerTile + tid] = r;
        } else if (POSE) {
            double pacc[6] = { acc[0], acc[1], acc[2], acc[3], acc[4], acc[5] };
            const double r = block_sum_quads<6>(pacc, L.park);
            if (tid < 6) rsPart[(size_t)d * kRsPerTile + tid] = r;
        }
        }
    }
}

// The resampling a deposit decided on, on its own (undeferred filters; a pose query, download, ... that finds one pending).
template <bool RECORD, bool POSE>
__global__ __launch_bounds__(kThreads) void k_loc_resample_dest(LocBufs bufs, LocCtl* __restrict__ ctl,
                                                               const double* __restrict__ chunkBase,
                                                               const double* __restrict__ tileIncl, int* __restrict__ ancOut,
                                                               int numTiles, int n, RngKey key, RespawnShell sh,
                                                               double* __restrict__ rsPart, const DestPlan* __restrict__ plan)
{
    const DestPlan P = plan[blockIdx.x];    // read together with the control block: one round trip
    if (ctl->branch != 1) return;           // normalise-only step: nothing to do (wave-uniform exit)
    __shared__ DestLds L;
    if (POSE && blockIdx.x == 0 && threadIdx.x == 0) ctl->rsValid = 1;  // nobody in this launch reads it; k_loc_pose_tiles then exits
    resample_dest_tile<1, false, false, RECORD, POSE>(bufs, ctl, chunkBase, tileIncl, numTiles, n, key, sh, key, Motion{}, Gate{},
                                                   nullptr, ancOut, rsPart, P, L, (int)blockIdx.x);
}

// One deposit of an unsharded filter with deferred resampling, whichever way the previous decision went (it is
// known on the device only): after a "resample" decision the offspring are produced and deposited on in one pass
// (workgroup d makes destination tile d), after a "normalise" decision this is k_loc_stream.  One launch instead of two
// of which one would exit at once (an empty launch of 16 k workgroups plus its gap cost ~8 us of a 170 us deposit).
// The two sides share the LDS.
// Seven waves per SIMD (71 registers, no spill).  What bounds the resampling side (measured at N = 16 M, round 2): its
// arithmetic alone takes 0.095 ms (timing build without loads, selection and stores), its memory traffic alone ~0.115 ms,
// and the two overlap only as far as OTHER resident waves are in their arithmetic while one waits -- a single wave
// issues one vector instruction per ~9 cycles, so a SIMD needs five or six waves in arithmetic at once to be busy, of
// eight it can hold.  Hence: as many waves as the registers allow (0.204 ms at 5 waves, 0.190 at 6, 0.175 at 7; two
// destination tiles per workgroup, PF_DEPOSIT_TILES = 2, needs 85 registers and lost: 0.196), and arithmetic that
// depends on nothing but the counter -- the unit normals of the thread's first two offspring -- drawn while the wave
// waits for the source weights and for the ancestors (PF_NOISE_AHEAD1/2: 0.182 -> 0.163; a third offspring spills).
#ifndef PF_DEPOSIT_WAVES
#define PF_DEPOSIT_WAVES 7
#endif
template <bool MOTION>
__global__ __launch_bounds__(kThreads) __attribute__((amdgpu_waves_per_eu(PF_DEPOSIT_WAVES, PF_DEPOSIT_WAVES))) void k_loc_deposit(LocBufs bufs, const LocCtl* __restrict__ ctl,
                                                         double* __restrict__ tileSums,
                                                         const double* __restrict__ chunkBase,
                                                         const double* __restrict__ tileIncl, int numTiles, int n,
                                                         RngKey keyRs, RespawnShell sh, RngKey key, Motion mo, Gate ga,
                                                         const DestPlan* __restrict__ plan)
{
    union DepositLds { DestLds dest; double red[7 * kThreads]; };
    __shared__ DepositLds L;
#if PF_XCD_REMAP
    static_assert(kDepositTiles == 1, "the remap deals single tiles");
    const int d0 = xcd_tile(blockIdx.x, numTiles);
#else
    const int d0 = blockIdx.x * kDepositTiles;
#endif
    const DestPlan P = plan[d0];            // read together with the control block: one round trip (unused when streaming)
    const int branch = ctl->branch;
    if (branch == 1) {
        resample_dest_tile<kDepositTiles, true, MOTION, false, false>(bufs, ctl, chunkBase, tileIncl, numTiles, n, keyRs, sh, key, mo, ga,
                                                                      tileSums, nullptr, nullptr, P, L.dest, d0);
    } else {
        for (int s = 0; s < kDepositTiles && d0 + s < numTiles; ++s) {
            if (s > 0) __syncthreads();     // the previous tile's last reads of L.red
            stream_tile<MOTION, true>(bufs, ctl, tileSums, (long long)n, 0, key, mo, ga, L.red, d0 + s);
        }
    }
}

// The same for one shard of a sharded filter, destination-owned like k_loc_deposit: workgroup d makes the shard's tile d
// from the shard's own ancestors (plan records by k_loc_shard_decide_plan).  Tiles none of whose elements descend from a local
// ancestor are left to k_loc_land; the offspring that belong to OTHER shards are staged by k_loc_stage_foreign.
template <bool MOTION>
#ifndef PF_SHARD_WAVES
#define PF_SHARD_WAVES 7
#endif
__global__ __launch_bounds__(kThreads) __attribute__((amdgpu_waves_per_eu(PF_SHARD_WAVES, PF_SHARD_WAVES))) void k_loc_deposit_shard(LocBufs bufs, const LocCtl* __restrict__ ctl,
                                                               double* __restrict__ tileSums,
                                                               const double* __restrict__ chunkBase,
                                                               const double* __restrict__ tileIncl, int numTiles, int n,
                                                               long long idx0, RngKey keyRs, RespawnShell sh, RngKey key, Motion mo,
                                                               Gate ga, const DestPlan* __restrict__ plan)
{
    union DepositLds { DestLds dest; double red[7 * kThreads]; };
    __shared__ DepositLds L;
    const int d = blockIdx.x;
    const DestPlan P = plan[d];
    if (ctl->branch == 1) {
        const long long first = ctl->selfBegin + (long long)d * kTile;      // global index of the tile's first element
        if (first >= ctl->outEnd || first + kTile <= ctl->outBegin) return;
        resample_dest_tile<1, true, MOTION, false, false, true>(bufs, ctl, chunkBase, tileIncl, numTiles, n, keyRs, sh, key, mo, ga,
                                                                tileSums, nullptr, nullptr, P, L.dest, d);
    } else {
        stream_tile<MOTION, true>(bufs, ctl, tileSums, (long long)n, idx0, key, mo, ga, L.red, d);
    }
}

// The source-owned form (round 1; PFX_SHARD_DIRECT=0): a workgroup owns a SOURCE tile (global indices, foreign offspring
// staged).  The host launches it before it has looked at the previous decision, so that waiting for that decision
// overlaps with this kernel.
template <bool MOTION>
__global__ __launch_bounds__(kThreads) void k_loc_deposit_sharded(LocBufs bufs, const LocCtl* __restrict__ ctl,
                                                                 double* __restrict__ tileSums,
                                                                 const double* __restrict__ chunkBase,
                                                                 const double* __restrict__ tileIncl, int numTiles, int n,
                                                                 long long idx0, RngKey keyRs, RespawnShell sh, RngKey key,
                                                                 Motion mo, Gate ga, PieceOut out, StagePtrs stage)
{
    __shared__ ResampleDepositLds L;
    if (ctl->branch == 1)
        resample_deposit_tile<MOTION, true>(bufs, ctl, chunkBase, tileIncl, numTiles, n, keyRs, sh, key, mo, ga, out, stage, L,
                                            (int)blockIdx.x);
    else
        stream_tile<MOTION, true>(bufs, ctl, tileSums, (long long)n, idx0, key, mo, ga, reinterpret_cast<double*>(&L), (int)blockIdx.x);
}

// The offspring of this shard's ancestors that belong to OTHER shards (global index outside [selfBegin, selfEnd)), as
// records in the send half of the staging area: the companion of k_loc_deposit_shard.  They descend from the shard's
// first and last source tiles (those whose CDF span reaches below position u(selfBegin) or beyond u(selfEnd)): every
// thread tests one tile with two comparisons, the few tiles that pass are taken one after the other by the whole
// workgroup, source-owned (resample_deposit_tile<STAGE_ONLY>: tile in LDS, its foreign offspring in a strided loop).
// Grid: one workgroup per 256 tiles.
static __global__ __launch_bounds__(kThreads) void k_loc_stage_foreign(LocBufs bufs, const LocCtl* __restrict__ ctl,
                                                               const double* __restrict__ chunkBase,
                                                               const double* __restrict__ tileIncl, int numTiles, int n,
                                                               RngKey keyRs, RespawnShell sh, StagePtrs stage)
{
    if (ctl->branch != 1) return;
    __shared__ ResampleDepositLds L;
    __shared__ int list[kThreads];
    __shared__ int count;
    const int tid = threadIdx.x, t = blockIdx.x * kThreads + tid;
    if (tid == 0) count = 0;
    __syncthreads();
    const long long gOff = ctl->selfBegin, selfEnd = gOff + n, outBegin = ctl->outBegin, outEnd = ctl->outEnd;
    const bool below = outBegin < gOff, above = outEnd > selfEnd;
    if (t < numTiles && (below || above)) {
        const double rStart = ctl->rStart, rStep = ctl->rStep, shardBase = ctl->shardBase, shardEnd = ctl->shardEnd;
        const double a = tile_start_cdf(chunkBase, tileIncl, t, shardBase, shardEnd);
        const double b = tile_start_cdf(chunkBase, tileIncl, t + 1, shardBase, shardEnd);
        // offspring i of the tile: a <= u_i < b, u non-decreasing in i.  (Written so that a NaN position keeps the tile.)
        bool need = (below && !(a > rStart + rStep * (double)gOff)) || (above && !(b <= rStart + rStep * (double)selfEnd));
        need = need || (below && t == 0) || (above && t == numTiles - 1);   // the shard's ends are what the global plan says
        if (need) list[atomicAdd(&count, 1)] = t;
    }
    __syncthreads();
    const int c = count;
    for (int q = 0; q < c; ++q) {
        if (q > 0) __syncthreads();
        resample_deposit_tile<false, true, true>(bufs, ctl, chunkBase, tileIncl, numTiles, n, keyRs, sh, keyRs, Motion{}, Gate{},
                                                 PieceOut{ nullptr, nullptr, nullptr }, stage, L, list[q]);
    }
}

// Offspring received from the neighbouring shards: record k of `landing` belongs to local element k (k < lo: sent by
// the shards below) or hi + (k - lo) (sent by the shards above); [lo, hi) is what this shard produced itself.  The
// records are unpacked into the current SoA arrays with w = 1.  DEPOSIT: the pending deposit is applied on the way
// (the body of k_loc_stream on a weight-1 particle, Philox counter = global index) and the seven sums of the
// received elements of each tile go to `extra` (k_loc_chunks adds them to the tile's pieces).
// Grid: the tiles that overlap [0, lo), then those that overlap [hi, n) (none of them twice).
template <bool MOTION, bool DEPOSIT>
__global__ __launch_bounds__(kThreads) void k_loc_land(LocBufs bufs, const LocCtl* __restrict__ ctl,
                                                      const float4* __restrict__ landing, int lo, int hi, int n,
                                                      int tilesBelow, int firstTileAbove, long long idx0, RngKey key,
                                                      Motion mo, Gate ga, double* __restrict__ extra)
{
    __shared__ double red[DEPOSIT ? 7 * kThreads : 1];
    const int tile = (int)blockIdx.x < tilesBelow ? (int)blockIdx.x : firstTileAbove + ((int)blockIdx.x - tilesBelow);
    const int cur = ctl->cur;
    float* __restrict__ pw = bufs.p(cur, 0);
    float* __restrict__ px = bufs.p(cur, 1);
    float* __restrict__ py = bufs.p(cur, 2);
    float* __restrict__ pz = bufs.p(cur, 3);
    float* __restrict__ pt = bufs.p(cur, 4);
    const bool smallTurn = mo.stdTheta <= 1e-4f;
    double acc[7] = { 0, 0, 0, 0, 0, 0, 0 };
#pragma unroll
    for (int k = 0; k < kPerThread; ++k) {
        const int el = tile * kTile + (int)threadIdx.x * kPerThread + k;
        if (el >= n || (el >= lo && el < hi)) continue;
        const float4 r = landing[el < lo ? el : lo + (el - hi)];
        float wi = 1.0f, x = r.x, y = r.y, z = r.z, th = r.w;
        if (DEPOSIT) {
            float sn, cs;
            if (MOTION) {
                motion_particle(key, (uint64_t)(idx0 + el), mo, smallTurn, x, y, z, th, sn, cs);
            } else {
                th = wrap_two_pi(th);
                sincos_bounded(th, sn, cs);
            }
            if (gate_rejects(ga, x - ga.bx, y - ga.by, z - ga.bz)) wi *= ga.minWeight;
            acc[0] += cdf_weight(wi);
            acc[1] += wi * wi;
            acc[2] += wi * cs;
            acc[3] += wi * sn;
            acc[4] += wi * x;
            acc[5] += wi * y;
            acc[6] += wi * z;
        }
        pw[el] = wi; px[el] = x; py[el] = y; pz[el] = z; pt[el] = th;
    }
    if (DEPOSIT) block_sum_store<7>(acc, red, extra + (size_t)tile * kSumsPerTile);
}

// ------------------------------------------------------------------ pose sums after a resampling step
// The resample kernel does not accumulate the pose sums of the offspring (a sincos per particle in a kernel bound
// by vector-ALU issue, for a result most deposits never ask for).  When a pose is asked for after a resampling
// step, this pass computes the per-tile sums of w, wx, wy, wz, w cos, w sin over the current particles
// (particleFilter.c:263-273); after a normalise-only step, or when the partials are already there, it exits.
// body shared by k_loc_pose_tiles (the filter's own arrays) and k_pose_tiles_raw (a caller's arrays, pfx_pose_sums_noalloc)
template <bool PADDED>
__device__ __forceinline__ void pose_tile_body(const float* __restrict__ pw, const float* __restrict__ px, const float* __restrict__ py,
                                               const float* __restrict__ pz, const float* __restrict__ pt, float wscale, long long n,
                                               double* __restrict__ part, double* red, int tile)
{
    const long long base = (long long)tile * kTile + (long long)threadIdx.x * kPerThread;
    float w[4], x[4], y[4], z[4], th[4];
    if (PADDED || base + kPerThread <= n) {         // the filter's arrays are padded to whole tiles; a caller's need not be
        const float4 w4 = *reinterpret_cast<const float4*>(pw + base);
        const float4 x4 = *reinterpret_cast<const float4*>(px + base);
        const float4 y4 = *reinterpret_cast<const float4*>(py + base);
        const float4 z4 = *reinterpret_cast<const float4*>(pz + base);
        const float4 t4 = *reinterpret_cast<const float4*>(pt + base);
        w[0] = w4.x; w[1] = w4.y; w[2] = w4.z; w[3] = w4.w; x[0] = x4.x; x[1] = x4.y; x[2] = x4.z; x[3] = x4.w;
        y[0] = y4.x; y[1] = y4.y; y[2] = y4.z; y[3] = y4.w; z[0] = z4.x; z[1] = z4.y; z[2] = z4.z; z[3] = z4.w;
        th[0] = t4.x; th[1] = t4.y; th[2] = t4.z; th[3] = t4.w;
    } else {
#pragma unroll
        for (int k = 0; k < kPerThread; ++k) {
            const bool in = base + k < n;
            w[k] = in ? pw[base + k] : 0.0f; x[k] = in ? px[base + k] : 0.0f; y[k] = in ? py[base + k] : 0.0f;
            z[k] = in ? pz[base + k] : 0.0f; th[k] = in ? pt[base + k] : 0.0f;
        }
    }
    double acc[6] = { 0, 0, 0, 0, 0, 0 };
#pragma unroll
    for (int k = 0; k < kPerThread; ++k) {
        const float wi = (base + k < n) ? w[k] * wscale : 0.0f;
        float sn, cs;
        sincos_bounded(th[k], sn, cs);
        acc[0] += wi * x[k]; acc[1] += wi * y[k]; acc[2] += wi * z[k];
        acc[3] += wi * cs; acc[4] += wi * sn; acc[5] += cdf_weight(wi);
    }
    block_sum_store<6>(acc, red, part + (size_t)tile * kRsPerTile);
}

static __global__ __launch_bounds__(kThreads) void k_loc_pose_tiles(LocBufs bufs, const LocCtl* __restrict__ ctl,
                                                            double* __restrict__ rsPart, long long n)
{
    if (ctl->branch != 1 || ctl->rsValid) return;
    __shared__ double red[6 * kThreads];
    const int cur = ctl->cur;
    pose_tile_body<true>(bufs.p(cur, 0), bufs.p(cur, 1), bufs.p(cur, 2), bufs.p(cur, 3), bufs.p(cur, 4), ctl->wscale, n, rsPart, red, (int)blockIdx.x);
}

// The same pass over a CALLER's device arrays, partials into a workspace the caller of the kernel names
// (pfx_pose_sums_noalloc: the library's static scratch array).
static __global__ __launch_bounds__(kThreads) void k_pose_tiles_raw(const float* __restrict__ pw, const float* __restrict__ px,
                                                            const float* __restrict__ py, const float* __restrict__ pz,
                                                            const float* __restrict__ pt, float wscale, long long n,
                                                            double* __restrict__ part)
{
    __shared__ double red[6 * kThreads];
    pose_tile_body<false>(pw, px, py, pz, pt, wscale, n, part, red, (int)blockIdx.x);
}

// ------------------------------------------------------------------ initial spawn (pfInit.c:20-25)
__device__ __forceinline__ void spawn_tile_body(const LocBufs& bufs, int cur, double* __restrict__ rsPart,
                                                long long n, long long idx0, const RngKey& key, float bx, float by, float bz,
                                                float range, float stdRange, double* red, int tile)
{
    float* __restrict__ pw = bufs.p(cur, 0);
    float* __restrict__ px = bufs.p(cur, 1);
    float* __restrict__ py = bufs.p(cur, 2);
    float* __restrict__ pz = bufs.p(cur, 3);
    float* __restrict__ pt = bufs.p(cur, 4);
    double acc[6] = { 0, 0, 0, 0, 0, 0 };
    const long long tbase = (long long)tile * kTile;
#pragma unroll
    for (int k = 0; k < kPerThread; ++k) {
        const long long i = tbase + k * kThreads + threadIdx.x;
        if (i < n) {
            float x, y, z, th;
            spawn_from_range(key, (uint64_t)(idx0 + i), bx, by, bz, range, stdRange, x, y, z, th);
            pw[i] = 1.0f; px[i] = x; py[i] = y; pz[i] = z; pt[i] = th;
            float sn, cs;
            sincos_bounded(th, sn, cs);
            acc[0] += x; acc[1] += y; acc[2] += z; acc[3] += cs; acc[4] += sn; acc[5] += 1.0;
        } else {
            pw[i] = 0.0f; px[i] = 0.0f; py[i] = 0.0f; pz[i] = 0.0f; pt[i] = 0.0f;
        }
    }
    block_sum_store<6>(acc, red, rsPart + (size_t)tile * kRsPerTile);
}

static __global__ __launch_bounds__(kThreads) void k_loc_spawn(LocBufs bufs, LocCtl* __restrict__ ctl, double* __restrict__ rsPart,
                                                       long long n, long long idx0, RngKey key, float bx, float by, float bz,
                                                       float range, float stdRange)
{
    __shared__ double red[6 * kThreads];
    spawn_tile_body(bufs, ctl->cur, rsPart, n, idx0, key, bx, by, bz, range, stdRange, red, (int)blockIdx.x);
}

// ------------------------------------------------------------------ small control kernels
static __global__ void k_loc_set_ctl(LocCtl* ctl, int cur, int branch, float wscale, int keepCur, int rsValid)
{
    if (threadIdx.x == 0 && blockIdx.x == 0) {
        if (!keepCur) ctl->cur = cur;
        ctl->branch = branch;
        ctl->wscale = wscale;
        ctl->rsValid = rsValid;
    }
}

// Sum the per-tile pose partials of the last writer of the particles into ctl->pose (one workgroup).
// After a normalise-only step k_loc_decide has already done this from the streaming pass.
// fromPartials: per-tile {x y z cos sin w} partials (pose pass, spawn, resample); else the streaming pass's tile sums.
__device__ __forceinline__ void pose_sum_body(const double* __restrict__ tileSums, const double* __restrict__ rsPart, bool fromPartials,
                                              int numTiles, double (&acc)[6], double* red)
{
#pragma unroll
    for (int k = 0; k < 6; ++k) acc[k] = 0.0;
    for (int t = threadIdx.x; t < numTiles; t += kThreads) {
        if (fromPartials) {
            const double* p = rsPart + (size_t)t * kRsPerTile;
            acc[0] += p[5]; acc[1] += p[0]; acc[2] += p[1]; acc[3] += p[2]; acc[4] += p[3]; acc[5] += p[4];
        } else {
            const double* p = tileSums + (size_t)t * kSumsPerTile;
            acc[0] += p[0]; acc[1] += p[4]; acc[2] += p[5]; acc[3] += p[6]; acc[4] += p[2]; acc[5] += p[3];
        }
    }
    block_sum<6>(acc, red);
}

static __global__ __launch_bounds__(kThreads) void k_loc_pose(LocCtl* __restrict__ ctl, const double* __restrict__ tileSums,
                                                      const double* __restrict__ rsPart, int numTiles, long long n)
{
    __shared__ double red[4 * 6];
    const int branch = ctl->branch;
    double acc[6];
    pose_sum_body(tileSums, rsPart, branch == 1, numTiles, acc, red);
    if (threadIdx.x == 0) {
#pragma unroll
        for (int k = 0; k < 6; ++k) ctl->pose[k] = acc[k];
        if (branch == 1) ctl->rsValid = 1;
    }
}

// ... of k_pose_tiles_raw's partials, into the caller's six doubles {sum w, w x, w y, w z, w cos, w sin}
static __global__ __launch_bounds__(kThreads) void k_pose_sum_raw(const double* __restrict__ part, int numTiles, double* __restrict__ out6)
{
    __shared__ double red[4 * 6];
    double acc[6];
    pose_sum_body(nullptr, part, true, numTiles, acc, red);
    if (threadIdx.x == 0) {
#pragma unroll
        for (int k = 0; k < 6; ++k) out6[k] = acc[k];
    }
}

// ------------------------------------------------------------------ index selection on a caller CDF
// anc[i] = first j with rStart + rStep*i < cdf[j], else n-1 (pfResample.c:70-72).  T = float reproduces the
// reference's float expression exactly (no fused multiply-add: -ffp-contract=off); T = double is the
// production rule.  Same search as upper_bound_clamped in k_loc_resample, on global memory.
template <typename T>
__global__ __launch_bounds__(kThreads) void k_resample_indices(const T* __restrict__ cdf, long long n, T rStart, T rStep,
                                                              int* __restrict__ anc)
{
    const long long i = (long long)blockIdx.x * kThreads + threadIdx.x;
    if (i >= n) return;
    const T u = rStart + rStep * (T)i;
    long long lo = 0, hi = n - 1;
    while (lo < hi) {
        const long long mid = (lo + hi) >> 1;
        if (u < cdf[mid]) hi = mid; else lo = mid + 1;
    }
    anc[i] = (int)lo;
}

}  // namespace pf

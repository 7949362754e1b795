// pf_loc_batch_kernels.h -- many independent particleFilterLoc filters of equal size in ONE set of launches.
//
// The reference runs one filter per tag and one call per measurement (csvlocalize.c:64-88); its own filter size is
// N = 10 000 (particleFilter.h:18), where a deposit on the GPU is bound by its launches and by the host's calls, not by
// the particles.  A server that tracks many tags runs F such filters side by side: here they share their launches --
// grid dimension y = filter -- and nothing else: every filter has its own particles, control block, Philox stream
// (seed, stream id, deposit count), VIO accumulator and resample-or-normalise decision, taken on the device.  The
// kernels are thin frames around the SAME device functions the single-filter kernels are made of (stream_tile,
// resample_dest_tile, decide_small_body, ...), so filter f of a batch holds bit for bit the particles a single filter
// with its seed and stream id would hold after the same calls (tests/test_gpu_loc_batch.py).
//
// One deposit of the whole batch (particleFilter.c:178-191 for every filter that has a measurement):
//   k_locb_prepare   one thread per filter: pfMeasurement.c:31-32 (noise scales) and the squared-distance bounds of the
//                    range gate (gate_of, pf_loc_host.h -- sixty dependent steps, far too slow on the host for 1600 filters
//                    per deposit); marks a filter that spawns.                                    [tiny]
//   k_locb_deposit<motion>  grid (tiles, F): per filter what k_loc_deposit / k_loc_stream would do (k_locb_spawn: k_loc_spawn).
//   k_locb_decide_small  grid (F): the filter's prefix, decision and plan records (N <= 262 144), or
//   k_locb_chunks + k_locb_decide_map  grid (chunks, F) for larger filters.
#pragma once

#include "pf_loc_kernels.h"

namespace pf {

enum : int { LB_NONE = 0, LB_DEPOSIT = 1, LB_SPAWN = 2, LB_MATERIALISE = 3 };

// What the host sends per filter and operation: 64 bytes in pinned, device-mapped host memory, read by k_locb_prepare
// itself -- no copy engine between the kernels of two deposits (a hipMemcpyAsync of the full records and its hand-over to
// the compute queue cost ~15 us per deposit of 1600 filters).
struct __attribute__((aligned(16))) BatchIn {
    int mode;               // LB_*
    int motion;             // the VIO delta is not all zeros
    int fused;              // a decision is pending whose offspring have not been produced: the pass acts on it
    int poseWanted;         // pose launches: refresh this filter's pose sums
    float d[5];             // VIO delta: dt dx dy dz ddist (particleFilter.c:389-404)
    float bx, by, bz, range, stdRange;
    uint32_t epoch;         // the filter's deposit count (Philox counter word 2)
    int pad;
};
static_assert(sizeof(BatchIn) == 64, "one 64-byte record per filter");

// One filter's part of a batched operation as the kernels read it (device memory, one record per filter, persistent):
// k_locb_prepare fills it from the host's BatchIn, from the control block and from what it remembers of the filter's
// previous deposit.
struct __attribute__((aligned(16))) BatchStep {
    int mode, motion, fused, poseWanted;    // as BatchIn
    int resample;           // fused and the pending decision was "resample": the pass produces the offspring
    int cur;                // the control block's current buffer and pending weight factor, for the streaming side
    float wscale;
    int pad0;
    RngKey key;             // this deposit's Philox key (seed, stream id, deposit count)
    RngKey keyRs;           // ... of the deposit whose decision is pending
    float d[5];
    float bx, by, bz, range, stdRange;
    RespawnShell sh;        // measurement of the deposit whose decision is pending (pfResample.c:75-76)
    Motion mo;              // pfMeasurement.c:31-32
    Gate ga;                // pfMeasurement.c:22,114,118-123
    // persistent, not read by the O(N) kernels
    uint32_t seed, streamId;    // Philox key words 0 and 1 of the filter (set at creation)
    RngKey pendKey;             // key and measurement of the filter's last deposit: what a pending resampling needs
    RespawnShell pendSh;
};

struct LocBatchPtrs {
    float* slab;                    // filter f: ten arrays of `padded` floats from slab + f * 10 * padded
    unsigned long long padded;
    LocCtl* ctl;                    // [F]
    double* tileSums;               // [F][numTiles][8]
    double* rsPart;                 // [F][numTiles][8]
    double* tileIncl;               // [F][numTiles]
    double* chunkSums;              // [F][numChunks][8]
    double* chunkBase;              // [F][numChunks + 1]
    DestPlan* plan;                 // [F][numTiles]
    int numTiles, numChunks, n, numFilters;
    __device__ __forceinline__ LocBufs bufs(int f) const { return LocBufs{ slab + (size_t)f * 10 * padded, padded }; }
    __device__ __forceinline__ double* sums(int f) const { return tileSums + (size_t)f * numTiles * kSumsPerTile; }
    __device__ __forceinline__ double* rs(int f) const { return rsPart + (size_t)f * numTiles * kRsPerTile; }
    __device__ __forceinline__ double* incl(int f) const { return tileIncl + (size_t)f * numTiles; }
    __device__ __forceinline__ double* csums(int f) const { return chunkSums + (size_t)f * numChunks * kSumsPerTile; }
    __device__ __forceinline__ double* cbase(int f) const { return chunkBase + (size_t)f * (numChunks + 1); }
    __device__ __forceinline__ DestPlan* plans(int f) const { return plan + (size_t)f * numTiles; }
};

// gate_of (pf_loc_host.h) on the device: the same two bisections over the bit patterns of the non-negative floats with the
// reference's own expression fabsf(sqrtf(d2) - range) > thresh (pfMeasurement.c:118-123).  sqrtf is correctly rounded on
// both sides and nothing here is contracted, so the bounds are the host's bit for bit.
__device__ __forceinline__ void gate_bounds(Gate& g)
{
    const float inf = __builtin_huge_valf();
    g.d2Lo = 0.0f; g.d2Hi = inf;
    const float r = g.range, t = g.thresh;
    if (r != r || t != t || t == inf) return;
    if (t == -inf || r == -inf) { g.d2Lo = inf; g.d2Hi = -1.0f; return; }
    if (r == inf) { g.d2Lo = inf; g.d2Hi = inf; return; }
    auto e_of = [&](uint32_t bits) { return sqrtf(__uint_as_float(bits)) - r; };
    // The bisections start from a bracket of +-32 bit patterns around the squared bound, checked with the predicate itself
    // (monotone in the bit pattern), and fall back to the whole range [0, +inf] if it does not hold: 8 steps instead of 31
    // for every gate that occurs in practice, the same answer always.
    auto bracket = [&](float edge, auto pred, uint32_t& lo, uint32_t& hi) {     // first pattern with pred true lies in [lo, hi]
        lo = 0; hi = 0x7F800000u;
        if (!(edge > 0.0f)) return;
        const float sq = edge * edge;
        if (!(sq < inf)) return;
        const uint32_t c = __float_as_uint(sq);
        const uint32_t a = c > 32u ? c - 32u : 0u, b = c + 32u < 0x7F800000u ? c + 32u : 0x7F800000u;
        if (b == 0x7F800000u || a == 0u) return;
        if (!pred(a - 1u) && pred(b)) { lo = a; hi = b; }
    };
    uint32_t lo, hi;
    auto ge_lo = [&](uint32_t bits) { return e_of(bits) >= -t; };
    auto gt_hi = [&](uint32_t bits) { return e_of(bits) > t; };
    bracket(r - t, ge_lo, lo, hi);
    while (lo < hi) { const uint32_t mid = lo + (hi - lo) / 2; if (ge_lo(mid)) hi = mid; else lo = mid + 1; }
    const uint32_t first = lo;
    bracket(r + t, gt_hi, lo, hi);
    while (lo < hi) { const uint32_t mid = lo + (hi - lo) / 2; if (gt_hi(mid)) hi = mid; else lo = mid + 1; }
    const uint32_t past = lo;
    if (first >= past) { g.d2Lo = inf; g.d2Hi = -1.0f; }
    else { g.d2Lo = __uint_as_float(first); g.d2Hi = __uint_as_float(past - 1); }
}

__global__ __launch_bounds__(64) void k_locb_prepare(const BatchIn* __restrict__ in, BatchStep* __restrict__ steps, LocCtl* __restrict__ ctl,
                                                     int numFilters)
{
    const int f = blockIdx.x * 64 + threadIdx.x;
    if (f >= numFilters) return;
    const BatchIn h = in[f];                // (pinned host memory: four 16-byte loads per lane, a wavefront's side by side)
    BatchStep& st = steps[f];
    st.mode = h.mode; st.motion = h.motion; st.fused = h.fused; st.poseWanted = h.poseWanted;
    if (h.mode == LB_DEPOSIT || h.mode == LB_SPAWN) {
        st.key = RngKey{ st.seed, st.streamId, h.epoch };
#pragma unroll
        for (int k = 0; k < 5; ++k) st.d[k] = h.d[k];
        st.bx = h.bx; st.by = h.by; st.bz = h.bz; st.range = h.range; st.stdRange = h.stdRange;
    }
    if (h.mode == LB_DEPOSIT) {
        // the deposit whose decision may be pending, and this one in its place for the next
        st.keyRs = st.pendKey; st.sh = st.pendSh;
        st.pendKey = RngKey{ st.seed, st.streamId, h.epoch };
        st.pendSh = RespawnShell{ h.bx, h.by, h.bz, h.range, h.stdRange };
        // pfMeasurement.c:31-32, :22, :114, :122
        st.mo = Motion{ h.d[1], h.d[2], h.d[3], sqrtf(h.d[4]) * kVioStdXyz, sqrtf(h.d[0]) * kVioStdTheta };
        Gate g{ h.bx, h.by, h.bz, h.range, 3 * h.stdRange, (h.range < 3.0f) ? 0.1f : 0.5f, 0.0f, 0.0f };
        gate_bounds(g);
        st.ga = g;
        if (ctl) {
            // what the pass needs of the control block (final by now: the previous deposit's decision precedes this kernel on
            // the stream), so that a workgroup of the pass has one round trip of scalar loads in front of its particle loads
            const LocCtl* c = ctl + f;
            st.resample = (h.fused != 0 && c->branch == 1) ? 1 : 0;
            st.cur = c->cur;
            st.wscale = c->wscale;
        }
    } else if (h.mode == LB_MATERIALISE) {
        st.keyRs = st.pendKey; st.sh = st.pendSh;
    } else if (h.mode == LB_SPAWN) {
        // what k_loc_set_ctl does in front of k_loc_spawn: buffer 0, "resampled", no pending factor, pose partials valid
        LocCtl* c = ctl + f;
        c->cur = 0; c->branch = 1; c->wscale = 1.0f; c->rsValid = 1;
    }
}

// first range of the flagged filters: pfInit.c:20-25 (launched only when the host knows of one)
__global__ __launch_bounds__(kThreads) void k_locb_spawn(LocBatchPtrs B, const BatchStep* __restrict__ steps)
{
    __shared__ double red[6 * kThreads];
    const int f = blockIdx.y;
    const BatchStep st = steps[f];
    if (st.mode != LB_SPAWN) return;
    spawn_tile_body(B.bufs(f), 0, B.rs(f), (long long)B.n, 0, st.key, st.bx, st.by, st.bz, st.range, st.stdRange, red, (int)blockIdx.x);
}

// The pass over the particles for the filters whose deposit carries a VIO motion (MOTION) or none: what k_loc_deposit<MOTION>
// (a decision is pending: resample-and-deposit or stream, as that decision says) or k_loc_stream<MOTION, true> (none is)
// does for one filter.  A batch whose filters all moved, or all stood still, since their last deposit -- the usual case: the
// tags' VIO arrives together -- takes one launch; a mixed one takes both, each leaving the other's filters at once.  (One
// kernel with both sides inlined held 34 more scalar registers than the 102 there are and spilled 25 vector registers.)
template <bool MOTION>
__global__ __launch_bounds__(kThreads) __attribute__((amdgpu_waves_per_eu(7, 7))) void k_locb_deposit(LocBatchPtrs B, const BatchStep* __restrict__ steps)
{
    union DepositLds { DestLds dest; double red[7 * kThreads]; };
    __shared__ DepositLds L;
    const int f = blockIdx.y, d0 = blockIdx.x;
    const BatchStep st = steps[f];       // (uniform address: scalar loads; the record sits in scalar registers like kernel arguments)
    if (st.mode != LB_DEPOSIT || (st.motion != 0) != MOTION) return;
    const LocBufs bufs = B.bufs(f);
    // (what the streaming side needs of the control block is in the record: k_locb_prepare)
    if (st.resample) {
        const LocCtl* ctl = B.ctl + f;
        const DestPlan P = B.plans(f)[d0];
        resample_dest_tile<1, true, MOTION, false, false>(bufs, ctl, B.cbase(f), B.incl(f), B.numTiles, B.n, st.keyRs, st.sh, st.key,
                                                         st.mo, st.ga, B.sums(f), nullptr, nullptr, P, L.dest, d0);
    } else {
        stream_tile_cw<MOTION, true>(bufs, st.cur, st.wscale, B.sums(f), (long long)B.n, 0, st.key, st.mo, st.ga, L.red, d0);
    }
}

__global__ __launch_bounds__(kChunkTiles) void k_locb_decide_small(LocBatchPtrs B, const BatchStep* __restrict__ steps, int force)
{
    const int f = blockIdx.x;
    const BatchStep st = steps[f];
    if (st.mode != LB_DEPOSIT) return;
    decide_small_body(B.ctl + f, B.sums(f), B.incl(f), B.csums(f), B.cbase(f), B.plans(f), B.numChunks, B.numTiles, (long long)B.n,
                      st.range, st.key, force, nullptr);
}

// the same for filters of at most 64 tiles: one wavefront per filter (decide_tiny_body)
__global__ __launch_bounds__(64) void k_locb_decide_tiny(LocBatchPtrs B, const BatchStep* __restrict__ steps, int force)
{
    const int f = blockIdx.x;
    const BatchStep st = steps[f];
    if (st.mode != LB_DEPOSIT) return;
    decide_tiny_body(B.ctl + f, B.sums(f), B.incl(f), B.csums(f), B.cbase(f), B.plans(f), B.numTiles, (long long)B.n, st.range, st.key,
                     force, nullptr);
}

__global__ __launch_bounds__(kChunkTiles) void k_locb_chunks(LocBatchPtrs B, const BatchStep* __restrict__ steps)
{
    const int f = blockIdx.y;
    if (steps[f].mode != LB_DEPOSIT) return;
    chunks_body(B.sums(f), B.incl(f), B.csums(f), B.numTiles, B.ctl + f, TilePieces{ nullptr, nullptr, nullptr, nullptr, 0, 0, 0, 0 },
                (int)blockIdx.x);
}

__global__ __launch_bounds__(kDecideMapThreads) void k_locb_decide_map(LocBatchPtrs B, const BatchStep* __restrict__ steps, int force)
{
    const int f = blockIdx.y;
    const BatchStep st = steps[f];
    if (st.mode != LB_DEPOSIT) return;
    decide_map_body(B.ctl + f, B.csums(f), B.cbase(f), B.incl(f), B.plans(f), B.numChunks, B.numTiles, (long long)B.n, st.range, st.key,
                    force, nullptr, (int)blockIdx.x);
}

// a pending resampling produced on its own (pose query, download): k_loc_resample_dest per flagged filter
template <bool POSE>
__global__ __launch_bounds__(kThreads) void k_locb_resample(LocBatchPtrs B, const BatchStep* __restrict__ steps)
{
    const int f = blockIdx.y, d0 = blockIdx.x;
    const BatchStep st = steps[f];
    if (st.mode != LB_MATERIALISE) return;
    LocCtl* ctl = B.ctl + f;
    const DestPlan P = B.plans(f)[d0];
    if (ctl->branch != 1) return;
    __shared__ DestLds L;
    if (POSE && d0 == 0 && threadIdx.x == 0) ctl->rsValid = 1;
    resample_dest_tile<1, false, false, false, POSE>(B.bufs(f), ctl, B.cbase(f), B.incl(f), B.numTiles, B.n, st.keyRs, st.sh, st.keyRs,
                                                     Motion{}, Gate{}, nullptr, nullptr, B.rs(f), P, L, d0);
}

__global__ __launch_bounds__(kThreads) void k_locb_pose_tiles(LocBatchPtrs B, const BatchStep* __restrict__ steps)
{
    const int f = blockIdx.y;
    if (!steps[f].poseWanted) return;
    const LocCtl* ctl = B.ctl + f;
    if (ctl->branch != 1 || ctl->rsValid) return;
    __shared__ double red[6 * kThreads];
    const LocBufs bufs = B.bufs(f);
    const int cur = ctl->cur;
    pose_tile_body<true>(bufs.p(cur, 0), bufs.p(cur, 1), bufs.p(cur, 2), bufs.p(cur, 3), bufs.p(cur, 4), ctl->wscale, (long long)B.n,
                         B.rs(f), red, (int)blockIdx.x);
}

__global__ __launch_bounds__(kThreads) void k_locb_pose(LocBatchPtrs B, const BatchStep* __restrict__ steps)
{
    const int f = blockIdx.x;
    if (!steps[f].poseWanted) return;
    __shared__ double red[4 * 6];
    LocCtl* ctl = B.ctl + f;
    const int branch = ctl->branch;
    double acc[6];
    pose_sum_body(B.sums(f), B.rs(f), branch == 1, B.numTiles, acc, red);
    if (threadIdx.x == 0) {
#pragma unroll
        for (int k = 0; k < 6; ++k) ctl->pose[k] = acc[k];
        if (branch == 1) ctl->rsValid = 1;
    }
}

// after an upload into filter f: buffer 0, no pending decision, weights as stored
__global__ void k_locb_set_ctl(LocCtl* ctl, int f, int cur, int branch, float wscale, int rsValid)
{
    if (threadIdx.x == 0 && blockIdx.x == 0) {
        LocCtl* c = ctl + f;
        c->cur = cur; c->branch = branch; c->wscale = wscale; c->rsValid = rsValid;
    }
}

}  // namespace pf

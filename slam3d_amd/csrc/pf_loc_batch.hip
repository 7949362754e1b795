// pf_loc_batch.hip -- C-ABI of pfx_loc_batch_*: F independent particleFilterLoc filters of equal size on one GPU, every
// deposit of the whole batch one set of launches (pf_loc_batch_kernels.h).  No counterpart in the reference, which runs
// one filter per tag and one call per measurement (csvlocalize.c:64-88); the per-filter semantics are exactly those of
// pfx_loc_deposit_vio / _range / _rssi / _get_tag_loc (pf_loc.hip; particleFilter.c:70-104, 178-191, 213-226, 248-291),
// including the deferred resampling: filter f of a batch and a single filter created with its seed and stream id hold
// the same particles bit for bit after the same calls.
//
// Host side of a deposit: O(F) scalar work (VIO commit, flags: ~10 ns per filter) into a 64-byte record per filter in
// pinned, device-mapped host memory, and three launches; the first one (k_locb_prepare) reads the records straight from
// host memory -- no copy engine in the chain.
#include "pf_loc_batch_kernels.h"
#include "pf_loc_host.h"

#include "../../include/pfx.h"

#include <mutex>
#include <new>
#include <unordered_set>
#include <vector>

using namespace pf;

namespace {
constexpr int kRing = 8;        // pinned staging buffers for the step records
constexpr int kHalf = kRing / 2;    // one "free again" event per half of the ring: an event record between two kernels idles
                                    // the GPU for ~6 us (kernel trace: the gap between k_locb_prepare and k_locb_deposit)
}

struct pfx_loc_batch {
    int device = 0;
    hipStream_t stream = nullptr;
    int F = 0;
    long long n = 0;
    int numTiles = 0, numChunks = 0;
    size_t padded = 0;
    LocBatchPtrs B{};
    BatchStep* dSteps = nullptr;            // [F] on the device: the filters' step records (k_locb_prepare keeps them)
    BatchIn* hSteps[kRing] = {};            // pinned, device-mapped: what the host sends per operation
    BatchIn* hStepsDev[kRing] = {};         // ... as the device addresses them
    hipEvent_t hFree[2] = {};               // recorded behind the k_locb_prepare that reads the last buffer of a half
    bool hUsed[2] = {};
    int ring = 0;
    LocCtl* hostCtl = nullptr;              // pinned, [F]
    int force = -1;
    // per filter, host side (what pfx_loc keeps per filter)
    std::vector<VioState> vio;
    std::vector<uint32_t> seed, streamId, epoch;
    std::vector<uint8_t> initialized, poseDirty, deferOpen;
};

namespace {

std::mutex g_bmu;
std::unordered_set<pfx_loc_batch*> g_blive;

void free_batch(pfx_loc_batch* b)
{
    LocBatchPtrs& B = b->B;
    if (B.slab) hipFree(B.slab);
    if (B.ctl) hipFree(B.ctl);
    if (B.tileSums) hipFree(B.tileSums);
    if (B.rsPart) hipFree(B.rsPart);
    if (B.tileIncl) hipFree(B.tileIncl);
    if (B.chunkSums) hipFree(B.chunkSums);
    if (B.chunkBase) hipFree(B.chunkBase);
    if (B.plan) hipFree(B.plan);
    if (b->dSteps) hipFree(b->dSteps);
    for (int i = 0; i < kRing; ++i)
        if (b->hSteps[i]) hipHostFree(b->hSteps[i]);
    for (int i = 0; i < 2; ++i)
        if (b->hFree[i]) hipEventDestroy(b->hFree[i]);
    if (b->hostCtl) hipHostFree(b->hostCtl);
    if (b->stream) hipStreamDestroy(b->stream);
}

int alloc_batch(pfx_loc_batch* b)
{
    PF_HIP_OK(hipStreamCreateWithFlags(&b->stream, hipStreamNonBlocking));
    LocBatchPtrs& B = b->B;
    const size_t F = (size_t)b->F, nt = (size_t)b->numTiles, nc = (size_t)b->numChunks;
    const size_t slabBytes = F * 10 * b->padded * sizeof(float);
    PF_HIP_OK(hipMalloc((void**)&B.slab, slabBytes));
    PF_HIP_OK(hipMemsetAsync(B.slab, 0, slabBytes, b->stream));
    struct { void** p; size_t bytes; } arr[] = {
        { (void**)&B.ctl, F * sizeof(LocCtl) },
        { (void**)&B.tileSums, F * nt * kSumsPerTile * sizeof(double) },
        { (void**)&B.rsPart, F * nt * kRsPerTile * sizeof(double) },
        { (void**)&B.tileIncl, F * nt * sizeof(double) },
        { (void**)&B.chunkSums, F * nc * kSumsPerTile * sizeof(double) },
        { (void**)&B.chunkBase, F * (nc + 1) * sizeof(double) },
        { (void**)&B.plan, F * nt * sizeof(DestPlan) },
        { (void**)&b->dSteps, F * sizeof(BatchStep) },
    };
    for (auto& a : arr) {
        PF_HIP_OK(hipMalloc(a.p, a.bytes));
        PF_HIP_OK(hipMemsetAsync(*a.p, 0, a.bytes, b->stream));
    }
    B.padded = b->padded;
    B.numTiles = b->numTiles; B.numChunks = b->numChunks; B.n = (int)b->n; B.numFilters = b->F;
    for (int i = 0; i < kRing; ++i) {
        PF_HIP_OK(hipHostMalloc((void**)&b->hSteps[i], F * sizeof(BatchIn), hipHostMallocMapped));
        memset(b->hSteps[i], 0, F * sizeof(BatchIn));
        PF_HIP_OK(hipHostGetDevicePointer((void**)&b->hStepsDev[i], b->hSteps[i], 0));
    }
    for (int i = 0; i < 2; ++i) PF_HIP_OK(hipEventCreateWithFlags(&b->hFree[i], hipEventDisableTiming));
    PF_HIP_OK(hipHostMalloc((void**)&b->hostCtl, F * sizeof(LocCtl), hipHostMallocDefault));
    // every control block as pfx_loc_create leaves it: buffer 0, nothing pending, factor 1
    std::vector<LocCtl> init(F);
    memset(init.data(), 0, F * sizeof(LocCtl));
    for (auto& c : init) c.wscale = 1.0f;
    PF_HIP_OK(hipMemcpyAsync(B.ctl, init.data(), F * sizeof(LocCtl), hipMemcpyHostToDevice, b->stream));
    PF_HIP_OK(hipStreamSynchronize(b->stream));
    return 0;
}

// the next pinned record buffer (waits for the kernel that last read it, normally long finished)
BatchIn* stage_begin(pfx_loc_batch* b)
{
    const int i = b->ring;
    // entering a half of the ring: every k_locb_prepare that read its buffers in the previous round has finished
    if (i % kHalf == 0 && b->hUsed[i / kHalf]) hipEventSynchronize(b->hFree[i / kHalf]);
    return b->hSteps[i];
}
// k_locb_prepare on the buffer just filled: the records reach the device, per-deposit scalars and gate bounds are derived
int stage_send(pfx_loc_batch* b)
{
    const int i = b->ring;
    hipLaunchKernelGGL(k_locb_prepare, dim3((b->F + 63) / 64), dim3(64), 0, b->stream, b->hStepsDev[i], b->dSteps, b->B.ctl, b->F);
    if (check_launch("k_locb_prepare")) return -1;
    if (i % kHalf == kHalf - 1) {
        PF_HIP_OK(hipEventRecord(b->hFree[i / kHalf], b->stream));
        b->hUsed[i / kHalf] = true;
    }
    b->ring = (i + 1) % kRing;
    return 0;
}

int batch_alive(pfx_loc_batch* b, const char* fn)
{
    if (!b) { set_error("%s: null batch", fn); return -2; }
    return 0;
}

// the pending resampling of the flagged filters on its own (pose: leave the pose partials of the offspring behind)
int materialise_flagged(pfx_loc_batch* b, bool pose)
{
    const dim3 grid(b->numTiles, b->F), block(kThreads);
    if (pose) hipLaunchKernelGGL((k_locb_resample<true>), grid, block, 0, b->stream, b->B, b->dSteps);
    else hipLaunchKernelGGL((k_locb_resample<false>), grid, block, 0, b->stream, b->B, b->dSteps);
    return check_launch("k_locb_resample");
}

int read_ctl_all(pfx_loc_batch* b)
{
    PF_HIP_OK(hipMemcpyAsync(b->hostCtl, b->B.ctl, (size_t)b->F * sizeof(LocCtl), hipMemcpyDeviceToHost, b->stream));
    PF_HIP_OK(hipStreamSynchronize(b->stream));
    return 0;
}

// filter f's pending resampling, before something other than the next deposit reads its particles
int materialise_one(pfx_loc_batch* b, int f)
{
    if (!b->deferOpen[f]) return 0;
    BatchIn* h = stage_begin(b);
    for (int g = 0; g < b->F; ++g) { h[g].mode = LB_NONE; h[g].poseWanted = 0; }
    h[f].mode = LB_MATERIALISE;
    if (stage_send(b)) return -1;
    b->deferOpen[f] = 0;
    return materialise_flagged(b, false);
}

}  // namespace

extern "C" pfx_loc_batch* pfx_loc_batch_create(int num_filters, int64_t n, const uint32_t* seeds, uint32_t seed0,
                                               const uint32_t* stream_ids, int device)
{
    clear_error();
    if (num_filters < 1 || num_filters > 65535) { set_error("pfx_loc_batch_create: num_filters=%d out of range [1, 65535]", num_filters); return nullptr; }
    if (n < 1 || n > (int64_t)1 << 28) { set_error("pfx_loc_batch_create: n=%lld out of range [1, 2^28]", (long long)n); return nullptr; }
    int ndev = 0;
    if (hipGetDeviceCount(&ndev) != hipSuccess) { (void)hipGetLastError(); ndev = 0; }
    if (ndev <= 0) { set_error("pfx_loc_batch_create: no HIP device visible"); return nullptr; }
    if (device < 0 || device >= ndev) { set_error("pfx_loc_batch_create: device %d not in [0,%d)", device, ndev); return nullptr; }
    pfx_loc_batch* b = new (std::nothrow) pfx_loc_batch();
    if (!b) { set_error("pfx_loc_batch_create: out of host memory"); return nullptr; }
    b->device = device;
    b->F = num_filters;
    b->n = n;
    b->numTiles = (int)((n + kTile - 1) / kTile);
    b->padded = (size_t)(b->numTiles | 1) * kTile;          // as pfx_loc_create: arrays an odd number of tiles apart
    b->numChunks = (b->numTiles + kChunkTiles - 1) / kChunkTiles;
    if ((double)num_filters * 10.0 * (double)b->padded * 4.0 > 200e9) {
        set_error("pfx_loc_batch_create: %d filters of %lld particles need more than 200 GB", num_filters, (long long)n);
        delete b;
        return nullptr;
    }
    const size_t F = (size_t)num_filters;
    b->vio.assign(F, VioState());
    b->seed.resize(F); b->streamId.resize(F); b->epoch.assign(F, 0);
    for (size_t f = 0; f < F; ++f) {
        b->seed[f] = seeds ? seeds[f] : seed0;
        b->streamId[f] = stream_ids ? stream_ids[f] : (uint32_t)f;
    }
    b->initialized.assign(F, 0); b->poseDirty.assign(F, 1); b->deferOpen.assign(F, 0);
    DeviceGuard g(device);
    if (alloc_batch(b)) { free_batch(b); delete b; return nullptr; }
    {   // the filters' Philox key words 0 and 1 live in their device records
        std::vector<BatchStep> init(F);
        memset(init.data(), 0, F * sizeof(BatchStep));
        for (size_t f = 0; f < F; ++f) { init[f].seed = b->seed[f]; init[f].streamId = b->streamId[f]; }
        if (hipMemcpy(b->dSteps, init.data(), F * sizeof(BatchStep), hipMemcpyHostToDevice) != hipSuccess) {
            set_error("pfx_loc_batch_create: copy of the filter records failed");
            free_batch(b); delete b; return nullptr;
        }
    }
    std::lock_guard<std::mutex> lk(g_bmu);
    g_blive.insert(b);
    return b;
}

extern "C" void pfx_loc_batch_destroy(pfx_loc_batch* b)
{
    if (!b) return;
    {
        std::lock_guard<std::mutex> lk(g_bmu);
        if (!g_blive.erase(b)) return;
    }
    DeviceGuard g(b->device);
    if (b->stream) hipStreamSynchronize(b->stream);
    free_batch(b);
    delete b;
}

extern "C" int pfx_loc_batch_num_filters(const pfx_loc_batch* b) { return b->F; }
extern "C" int64_t pfx_loc_batch_n(const pfx_loc_batch* b) { return b->n; }
extern "C" uint32_t pfx_loc_batch_seed(const pfx_loc_batch* b, int f) { return b->seed[(size_t)f]; }
extern "C" uint32_t pfx_loc_batch_stream_id(const pfx_loc_batch* b, int f) { return b->streamId[(size_t)f]; }
extern "C" uint32_t pfx_loc_batch_epoch(const pfx_loc_batch* b, int f) { return b->epoch[(size_t)f]; }
extern "C" int pfx_loc_batch_initialized(const pfx_loc_batch* b, int f) { return b->initialized[(size_t)f]; }

extern "C" int pfx_loc_batch_sync(pfx_loc_batch* b)
{
    clear_error();
    DeviceGuard g(b->device);
    PF_HIP_OK(hipStreamSynchronize(b->stream));
    return 0;
}

extern "C" int pfx_loc_batch_set_force_branch(pfx_loc_batch* b, int force)
{
    b->force = force < 0 ? -1 : (force ? 1 : 0);
    return 0;
}

// particleFilterLoc_depositVio for every filter f with mask[f] != 0 (mask NULL: all): O(1) per filter, host only
extern "C" int pfx_loc_batch_deposit_vio(pfx_loc_batch* b, const uint8_t* mask, const double* t, const float* x, const float* y,
                                         const float* z, const float* dist)
{
    clear_error();
    if (batch_alive(b, "pfx_loc_batch_deposit_vio")) return -2;
    for (int f = 0; f < b->F; ++f)
        if (!mask || mask[f]) vio_deposit(b->vio[(size_t)f], t[f], x[f], y[f], z[f], dist ? dist[f] : 0.0f);
    return 0;
}

// particleFilterLoc_depositRange for every filter f with mask[f] != 0 (mask NULL: all), one set of launches
extern "C" int pfx_loc_batch_deposit_range(pfx_loc_batch* b, const uint8_t* mask, const float* bx, const float* by, const float* bz,
                                           const float* range, const float* stdRange)
{
    clear_error();
    if (batch_alive(b, "pfx_loc_batch_deposit_range")) return -2;
    DeviceGuard g(b->device);
    BatchIn* h = stage_begin(b);
    int nDeposit = 0, nSpawn = 0, nMoving = 0;
    for (int f = 0; f < b->F; ++f) {
        BatchIn& r = h[f];
        r.poseWanted = 0;
        if (mask && !mask[f]) { r.mode = LB_NONE; continue; }
        const size_t q = (size_t)f;
        r.epoch = ++b->epoch[q];
        vio_commit(b->vio[q], r.d);
        b->poseDirty[q] = 1;
        r.bx = bx[f]; r.by = by[f]; r.bz = bz[f]; r.range = range[f]; r.stdRange = stdRange[f];
        if (!b->initialized[q]) {               // first range: spawn (particleFilter.c:186-189)
            r.mode = LB_SPAWN;
            b->initialized[q] = 1;
            b->deferOpen[q] = 0;
            ++nSpawn;
            continue;
        }
        r.mode = LB_DEPOSIT;
        r.motion = is_zero_motion(r.d) ? 0 : 1;
        nMoving += r.motion;
        r.fused = b->deferOpen[q];
        b->deferOpen[q] = 1;                    // this deposit's decision stays pending until the next one (or a read)
        ++nDeposit;
    }
    if (nDeposit + nSpawn == 0) return 0;
    if (stage_send(b)) return -1;
    const dim3 grid(b->numTiles, b->F), block(kThreads);
    if (nSpawn) {
        hipLaunchKernelGGL(k_locb_spawn, grid, block, 0, b->stream, b->B, b->dSteps);
        if (check_launch("k_locb_spawn")) return -1;
    }
    if (nMoving) {
        hipLaunchKernelGGL((k_locb_deposit<true>), grid, block, 0, b->stream, b->B, b->dSteps);
        if (check_launch("k_locb_deposit<motion>")) return -1;
    }
    if (nDeposit - nMoving) {
        hipLaunchKernelGGL((k_locb_deposit<false>), grid, block, 0, b->stream, b->B, b->dSteps);
        if (check_launch("k_locb_deposit<static>")) return -1;
    }
    if (nDeposit == 0) return 0;
    if (b->numTiles <= 64) {
        hipLaunchKernelGGL(k_locb_decide_tiny, dim3(b->F), dim3(64), 0, b->stream, b->B, b->dSteps, b->force);
        return check_launch("k_locb_decide_tiny");
    }
    if (b->numChunks <= kSmallChunks) {
        hipLaunchKernelGGL(k_locb_decide_small, dim3(b->F), dim3(kChunkTiles), 0, b->stream, b->B, b->dSteps, b->force);
        return check_launch("k_locb_decide_small");
    }
    hipLaunchKernelGGL(k_locb_chunks, dim3(b->numChunks, b->F), dim3(kChunkTiles), 0, b->stream, b->B, b->dSteps);
    if (check_launch("k_locb_chunks")) return -1;
    hipLaunchKernelGGL(k_locb_decide_map, dim3(b->numChunks, b->F), dim3(kDecideMapThreads), 0, b->stream, b->B, b->dSteps, b->force);
    return check_launch("k_locb_decide_map");
}

// particleFilterLoc_depositRssi (particleFilter.c:213-226): the rssi values are not used; a 1.5 m +- 0.5 m range
extern "C" int pfx_loc_batch_deposit_rssi(pfx_loc_batch* b, const uint8_t* mask, const float* bx, const float* by, const float* bz,
                                          const int32_t* rssi)
{
    (void)rssi;
    std::vector<float> r((size_t)b->F, 1.5f), s((size_t)b->F, 0.5f);
    return pfx_loc_batch_deposit_range(b, mask, bx, by, bz, r.data(), s.data());
}

// particleFilterLoc_getTagLoc of every filter: ok[f] = 1 / 0 (not initialised: outputs untouched)
extern "C" int pfx_loc_batch_get_tag_loc(pfx_loc_batch* b, uint8_t* ok, double* t, float* x, float* y, float* z, float* theta)
{
    clear_error();
    if (batch_alive(b, "pfx_loc_batch_get_tag_loc")) return -2;
    DeviceGuard g(b->device);
    int nDirty = 0, nOpen = 0;
    for (int f = 0; f < b->F; ++f) nDirty += (b->initialized[(size_t)f] && b->poseDirty[(size_t)f]) ? 1 : 0;
    if (nDirty) {
        BatchIn* h = stage_begin(b);
        for (int f = 0; f < b->F; ++f) {
            const size_t q = (size_t)f;
            const bool want = b->initialized[q] && b->poseDirty[q];
            h[f].poseWanted = want ? 1 : 0;
            h[f].mode = LB_NONE;
            if (want && b->deferOpen[q]) {
                h[f].mode = LB_MATERIALISE;
                b->deferOpen[q] = 0;
                ++nOpen;
            }
        }
        if (stage_send(b)) return -1;
        if (nOpen && materialise_flagged(b, true)) return -1;
        hipLaunchKernelGGL(k_locb_pose_tiles, dim3(b->numTiles, b->F), dim3(kThreads), 0, b->stream, b->B, b->dSteps);
        if (check_launch("k_locb_pose_tiles")) return -1;
        hipLaunchKernelGGL(k_locb_pose, dim3(b->F), dim3(kThreads), 0, b->stream, b->B, b->dSteps);
        if (check_launch("k_locb_pose")) return -1;
        if (read_ctl_all(b)) return -1;
        for (int f = 0; f < b->F; ++f)
            if (b->initialized[(size_t)f]) b->poseDirty[(size_t)f] = 0;
    }
    for (int f = 0; f < b->F; ++f) {
        const size_t q = (size_t)f;
        ok[f] = b->initialized[q] ? 1 : 0;
        if (ok[f]) pose_from_sums_vio(b->vio[q], b->hostCtl[q].pose, &t[f], &x[f], &y[f], &z[f], &theta[f]);
    }
    return 0;
}

extern "C" int pfx_loc_batch_download(pfx_loc_batch* b, int f, float* w, float* x, float* y, float* z, float* theta, int64_t n)
{
    clear_error();
    if (f < 0 || f >= b->F || n != b->n) { set_error("pfx_loc_batch_download: filter %d / n=%lld do not match the batch", f, (long long)n); return -2; }
    DeviceGuard g(b->device);
    if (materialise_one(b, f)) return -1;
    LocCtl c;
    PF_HIP_OK(hipMemcpyAsync(&c, b->B.ctl + f, sizeof(LocCtl), hipMemcpyDeviceToHost, b->stream));
    PF_HIP_OK(hipStreamSynchronize(b->stream));
    float* dst[5] = { w, x, y, z, theta };
    const float* base = b->B.slab + (size_t)f * 10 * b->padded + (size_t)c.cur * 5 * b->padded;
    for (int k = 0; k < 5; ++k)
        if (dst[k]) PF_HIP_OK(hipMemcpyAsync(dst[k], base + (size_t)k * b->padded, (size_t)n * sizeof(float), hipMemcpyDeviceToHost, b->stream));
    PF_HIP_OK(hipStreamSynchronize(b->stream));
    if (w && c.wscale != 1.0f)
        for (int64_t i = 0; i < n; ++i) w[i] *= c.wscale;       // pending w *= N/s (pfResample.c:80-82)
    return 0;
}

// as pfx_loc_last_stats, for filter f
extern "C" int pfx_loc_batch_last_stats(pfx_loc_batch* b, int f, double out[12])
{
    clear_error();
    if (f < 0 || f >= b->F) { set_error("pfx_loc_batch_last_stats: no filter %d", f); return -2; }
    DeviceGuard g(b->device);
    LocCtl c;
    PF_HIP_OK(hipMemcpyAsync(&c, b->B.ctl + f, sizeof(LocCtl), hipMemcpyDeviceToHost, b->stream));
    PF_HIP_OK(hipStreamSynchronize(b->stream));
    out[0] = c.branch; out[1] = c.numSpawn; out[2] = c.s; out[3] = c.ss; out[4] = c.csum; out[5] = c.ssum;
    out[6] = c.ess; out[7] = c.htheta; out[8] = c.rStart; out[9] = c.rStep; out[10] = c.m; out[11] = c.S;
    return 0;
}

// deposits decided so far and how many of them resampled, summed over the batch (per filter if the arrays are given)
extern "C" int pfx_loc_batch_branch_counts(pfx_loc_batch* b, int64_t* decisions, int64_t* resampled, int64_t* per_filter_decisions,
                                           int64_t* per_filter_resampled)
{
    clear_error();
    DeviceGuard g(b->device);
    if (read_ctl_all(b)) return -1;
    long long d = 0, r = 0;
    for (int f = 0; f < b->F; ++f) {
        d += b->hostCtl[f].nDecisions; r += b->hostCtl[f].nResampled;
        if (per_filter_decisions) per_filter_decisions[f] = b->hostCtl[f].nDecisions;
        if (per_filter_resampled) per_filter_resampled[f] = b->hostCtl[f].nResampled;
    }
    if (decisions) *decisions = d;
    if (resampled) *resampled = r;
    return 0;
}

// Diagnostic: the squared-distance bounds of the range gate (pfMeasurement.c:118-123) for `count` (range, stdRange)
// pairs, as the host derives them for a single filter (gate_of) and as k_locb_prepare derives them on the device for the
// filters of a batch -- the two must agree bit for bit (tests/test_gpu_loc_batch.py).
extern "C" int pfx_gate_bounds_both(const float* range, const float* stdRange, int count, float* lo_host, float* hi_host,
                                    float* lo_dev, float* hi_dev, int device)
{
    clear_error();
    if (count < 1) { set_error("pfx_gate_bounds_both: count < 1"); return -2; }
    DeviceGuard g(device);
    std::vector<BatchIn> h((size_t)count);
    memset(h.data(), 0, h.size() * sizeof(BatchIn));
    for (int i = 0; i < count; ++i) {
        h[(size_t)i].mode = LB_DEPOSIT;
        h[(size_t)i].range = range[i];
        h[(size_t)i].stdRange = stdRange[i];
        const Gate gh = gate_of(0.f, 0.f, 0.f, range[i], stdRange[i]);
        lo_host[i] = gh.d2Lo; hi_host[i] = gh.d2Hi;
    }
    BatchIn* din = nullptr;
    BatchStep* d = nullptr;
    std::vector<BatchStep> out((size_t)count);
    PF_HIP_OK(hipMalloc((void**)&din, h.size() * sizeof(BatchIn)));
    if (hipMalloc((void**)&d, out.size() * sizeof(BatchStep)) != hipSuccess) { hipFree(din); set_error("hipMalloc failed"); return -1; }
    int rc = 0;
    do {
        if (hipMemcpy(din, h.data(), h.size() * sizeof(BatchIn), hipMemcpyHostToDevice) != hipSuccess) { rc = -1; break; }
        if (hipMemset(d, 0, out.size() * sizeof(BatchStep)) != hipSuccess) { rc = -1; break; }
        hipLaunchKernelGGL(k_locb_prepare, dim3((count + 63) / 64), dim3(64), 0, 0, din, d, (LocCtl*)nullptr, count);
        if (check_launch("k_locb_prepare")) { rc = -1; break; }
        if (hipMemcpy(out.data(), d, out.size() * sizeof(BatchStep), hipMemcpyDeviceToHost) != hipSuccess) { rc = -1; break; }
    } while (0);
    hipFree(d);
    hipFree(din);
    if (rc) { if (!last_error_buf()[0]) set_error("pfx_gate_bounds_both: copy failed"); return rc; }
    const std::vector<BatchStep>& hh = out;
    for (int i = 0; i < count; ++i) { lo_dev[i] = hh[(size_t)i].ga.d2Lo; hi_dev[i] = hh[(size_t)i].ga.d2Hi; }
    return 0;
}

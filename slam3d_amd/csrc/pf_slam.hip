// pf_slam.hip -- C-ABI of the particleFilterSlam hot path: the reference's entry points
// (include/particleFilter.h: particleFilterSlam_*) and the pfx_slam_* extension (include/pfx.h) over the
// HIP kernels of pf_slam_kernels.h.
//
// A pfx_slam holds a BATCH of independent SLAM filters (one for the reference-layout API): per filter the
// tag particles and up to 32 beacons of nTag x nBcn particles, all in HBM as SoA arrays.  The host side of a
// deposit is the VIO bookkeeping (O(1) per filter/beacon, particleFilter.c:106-176, 406-438) and the
// description of the step handed to the kernels; everything O(particles) runs on the GPU.  No CPU path.
#include "pf_slam_kernels.h"
#include "pf_host.h"

#include "../../include/pfx.h"

#include <cmath>
#include <ctime>
#include <mutex>
#include <new>
#include <unordered_map>
#include <unordered_set>
#include <vector>
#include <unistd.h>

using namespace pf;

namespace {

struct VioAcc {                 // particleFilter.h:70-80 / :88-98 without `initialized`
    double firstT = 0.0; float firstX = 0, firstY = 0, firstZ = 0, firstDist = 0;
    double lastT = 0.0;  float lastX = 0, lastY = 0, lastZ = 0, lastDist = 0;
};

void vio_deposit(VioAcc& v, double t, float x, float y, float z, float dist)
{   // particleFilter.c:106-176 (tag and beacon variants are identical)
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

void vio_commit(VioAcc& v, float d[5])
{   // particleFilter.c:406-438
    d[0] = (float)(v.lastT - v.firstT);
    d[1] = v.lastX - v.firstX; d[2] = v.lastY - v.firstY; d[3] = v.lastZ - v.firstZ;
    d[4] = v.lastDist - v.firstDist;
    v.firstT = v.lastT; v.firstX = v.lastX; v.firstY = v.lastY; v.firstZ = v.lastZ; v.firstDist = v.lastDist;
}

Motion motion_of(const float d[5])
{   // pfMeasurement.c:58-59, 85-86
    return Motion{ d[1], d[2], d[3], sqrtf(d[4]) * kVioStdXyz, sqrtf(d[0]) * kVioStdTheta };
}
bool is_zero_motion(const float d[5])
{
    return d[0] == 0.0f && d[1] == 0.0f && d[2] == 0.0f && d[3] == 0.0f && d[4] == 0.0f;
}

struct FilterHost {
    VioAcc vio;
    uint32_t seed = 0, epoch = 0;
    int numBcns = 0;                        // slots handed out by pfx_slam_add_bcn
    bool bcnInit[kSlamMaxBcns] = {};
    VioAcc bcnVio[kSlamMaxBcns];
};

// one filter's request to the core step; deltas already committed by the caller
struct StepReq {
    int f, target;
    float range, stdRange;
    float tagDelta[5];
    uint32_t activeMask;
    float bcnDelta[kSlamMaxBcns][5];
};

constexpr int kRing = 8;

}  // namespace

struct pfx_slam {
    int device = 0;
    hipStream_t stream = nullptr;
    SlamDims D{};
    SlamPtrs P{};
    float* tagSlab = nullptr;
    float* bcnSlab = nullptr;
    float* rowSum = nullptr;        // F * nTag
    float* rowScale = nullptr;      // F * B * nTag: pending normalisation factor of every beacon row
    int* rowFlag = nullptr;         // F * nTag: what k_slam_gate_rows did to the measured beacon's rows
    int* tagBranch = nullptr;       // F
    int* forcedList = nullptr;      // F + 1: filters whose tag particles the current deposit resampled; [F] = their number
    double* tagPose = nullptr;      // F * 6
    double* bcnPose = nullptr;      // F * 6
    float* rowTerms = nullptr;      // F * nTag * 6
    int* slotOf = nullptr;          // F
    // per-deposit parameters: ring of pinned host blocks, one device block
    Motion* dBcnMo = nullptr;           // single filter: the beacon motions of the deposit at hand
    // batches: per-deposit records travel host -> device on a stream of their own, into a ring of device blocks, so that
    // the copy of deposit k + 1 runs while deposit k's kernels do (in-stream it was 4.6 us of every deposit plus the
    // hand-over on either side)
    SlamStep* dStepsRing[kRing] = {};
    Motion* dBcnMoRing[kRing] = {};
    hipStream_t copyStream = nullptr;
    hipEvent_t copyEv[kRing] = {};
    SlamStep* hSteps[kRing] = {};
    Motion* hBcnMo[kRing] = {};
    hipEvent_t ringEv[kRing] = {};
    bool ringUsed[kRing] = {};
    int ringPos = 0;
    void* hScratch = nullptr;       // pinned: pose sums, branch flags
    std::vector<FilterHost> fl;
    long long deposits = 0;
    // reference-layout API: slot of every bcn_t this filter has met, by struct address.  particleFilterSlam_addBcn
    // (the reference's way of resetting a beacon) wipes the struct's own binding record; the address finds the slot again.
    std::unordered_map<uintptr_t, int> legacySlots;
    // ... and how many state-changing calls the tag particles / each beacon slot have seen through that API: a struct
    // whose binding record shows another count holds a copy of an earlier state (see "POD copies" below)
    unsigned long long legacyOps = 0;
    unsigned long long bcnOps[kSlamMaxBcns] = {};
    // ... and the deposit (value of legacyOps) that last listed each slot: when all slots are taken, a bcn_t the filter has
    // not met before takes over the slot that has gone unlisted the longest (callers that free() their beacons never say
    // so: slam3d-jni.c:34-37)
    unsigned long long slotSeen[kSlamMaxBcns] = {};
};

static std::mutex g_smu;
static std::unordered_set<pfx_slam*> g_slive;

namespace {

struct DeviceGuard {
    int prev = -1;
    explicit DeviceGuard(int dev) { hipGetDevice(&prev); if (prev != dev) hipSetDevice(dev); else prev = -1; }
    ~DeviceGuard() { if (prev >= 0) hipSetDevice(prev); }
};

int check_launch(const char* what)
{
    hipError_t e = hipGetLastError();
    if (e != hipSuccess) { set_error("launch of %s failed: %s", what, hipGetErrorString(e)); return -1; }
    return 0;
}

void free_slam(pfx_slam* pf)
{
    if (pf->tagSlab) hipFree(pf->tagSlab);
    if (pf->bcnSlab) hipFree(pf->bcnSlab);
    if (pf->rowSum) hipFree(pf->rowSum);
    if (pf->rowScale) hipFree(pf->rowScale);
    if (pf->rowFlag) hipFree(pf->rowFlag);
    if (pf->tagBranch) hipFree(pf->tagBranch);
    if (pf->forcedList) hipFree(pf->forcedList);
    if (pf->tagPose) hipFree(pf->tagPose);
    if (pf->bcnPose) hipFree(pf->bcnPose);
    if (pf->rowTerms) hipFree(pf->rowTerms);
    if (pf->slotOf) hipFree(pf->slotOf);
    if (pf->dBcnMo) hipFree(pf->dBcnMo);
    for (int r = 0; r < kRing; ++r) {
        if (pf->dStepsRing[r]) hipFree(pf->dStepsRing[r]);
        if (pf->dBcnMoRing[r]) hipFree(pf->dBcnMoRing[r]);
        if (pf->copyEv[r]) hipEventDestroy(pf->copyEv[r]);
        if (pf->hSteps[r]) hipHostFree(pf->hSteps[r]);
        if (pf->hBcnMo[r]) hipHostFree(pf->hBcnMo[r]);
        if (pf->ringEv[r]) hipEventDestroy(pf->ringEv[r]);
    }
    if (pf->hScratch) hipHostFree(pf->hScratch);
    if (pf->copyStream) hipStreamDestroy(pf->copyStream);
    if (pf->stream) hipStreamDestroy(pf->stream);
}

int alloc_slam(pfx_slam* pf)
{
    const SlamDims& D = pf->D;
    const size_t F = (size_t)D.numFilters;
    PF_HIP_OK(hipStreamCreateWithFlags(&pf->stream, hipStreamNonBlocking));
    const size_t tagN = F * D.tagStride;
    const size_t bcnN = F * D.maxBcns * D.nTag * (size_t)D.rowStride;
    PF_HIP_OK(hipMalloc((void**)&pf->tagSlab, tagN * 5 * sizeof(float)));
    PF_HIP_OK(hipMalloc((void**)&pf->bcnSlab, bcnN * 5 * sizeof(float)));
    PF_HIP_OK(hipMemsetAsync(pf->bcnSlab, 0, bcnN * 5 * sizeof(float), pf->stream));
    for (int k = 0; k < 5; ++k) { pf->P.tag[k] = pf->tagSlab + k * tagN; pf->P.bcn[k] = pf->bcnSlab + k * bcnN; }
    PF_HIP_OK(hipMalloc((void**)&pf->rowSum, F * D.nTag * sizeof(float)));
    PF_HIP_OK(hipMalloc((void**)&pf->rowScale, F * D.maxBcns * D.nTag * sizeof(float)));
    PF_HIP_OK(hipMemsetD32Async((hipDeviceptr_t)pf->rowScale, 0x3f800000, F * D.maxBcns * D.nTag, pf->stream));    // 1.0f
    PF_HIP_OK(hipMalloc((void**)&pf->rowFlag, F * D.nTag * sizeof(int)));
    PF_HIP_OK(hipMemsetAsync(pf->rowFlag, 0, F * D.nTag * sizeof(int), pf->stream));
    PF_HIP_OK(hipMalloc((void**)&pf->tagBranch, F * sizeof(int)));
    PF_HIP_OK(hipMemsetAsync(pf->tagBranch, 0, F * sizeof(int), pf->stream));
    PF_HIP_OK(hipMalloc((void**)&pf->forcedList, (F + 1) * sizeof(int)));
    PF_HIP_OK(hipMemsetAsync(pf->forcedList, 0, (F + 1) * sizeof(int), pf->stream));
    PF_HIP_OK(hipMalloc((void**)&pf->tagPose, F * 6 * sizeof(double)));
    PF_HIP_OK(hipMalloc((void**)&pf->bcnPose, F * 6 * sizeof(double)));
    PF_HIP_OK(hipMalloc((void**)&pf->rowTerms, F * D.nTag * 6 * sizeof(float)));
    PF_HIP_OK(hipMalloc((void**)&pf->slotOf, F * sizeof(int)));
    PF_HIP_OK(hipMalloc((void**)&pf->dBcnMo, F * D.maxBcns * sizeof(Motion)));
    PF_HIP_OK(hipStreamCreateWithFlags(&pf->copyStream, hipStreamNonBlocking));
    for (int r = 0; r < kRing; ++r) {
        PF_HIP_OK(hipMalloc((void**)&pf->dStepsRing[r], F * sizeof(SlamStep)));
        PF_HIP_OK(hipMalloc((void**)&pf->dBcnMoRing[r], F * D.maxBcns * sizeof(Motion)));
        PF_HIP_OK(hipEventCreateWithFlags(&pf->copyEv[r], hipEventDisableTiming));
        PF_HIP_OK(hipHostMalloc((void**)&pf->hSteps[r], F * sizeof(SlamStep), hipHostMallocDefault));
        PF_HIP_OK(hipHostMalloc((void**)&pf->hBcnMo[r], F * D.maxBcns * sizeof(Motion), hipHostMallocDefault));
        PF_HIP_OK(hipEventCreateWithFlags(&pf->ringEv[r], hipEventDisableTiming));
    }
    PF_HIP_OK(hipHostMalloc(&pf->hScratch, F * (6 * sizeof(double) + sizeof(int)) + 64, hipHostMallocDefault));
    hipLaunchKernelGGL(k_slam_tag_init, dim3((unsigned)((tagN + 255) / 256)), dim3(256), 0, pf->stream, pf->P, pf->D, 0, D.numFilters);
    if (check_launch("k_slam_tag_init")) return -1;
    PF_HIP_OK(hipStreamSynchronize(pf->stream));
    return 0;
}

// The core of a deposit for `nReq` filters (every other filter of the batch sits this one out).
int slam_step(pfx_slam* pf, const StepReq* req, int nReq)
{
    const SlamDims& D = pf->D;
    const int F = D.numFilters;
    const bool single = (F == 1);
    SlamStep one{};
    SlamStep* steps = &one;
    Motion* bmo = nullptr;
    int slot = -1;
    if (!single) {
        slot = pf->ringPos;
        pf->ringPos = (pf->ringPos + 1) % kRing;
        // One event per HALF of the ring (an event record between two deposits idles the GPU for ~5 us, kernel trace):
        // ringEv[h] is recorded behind the last deposit of half h; a half is entered again only once the event of its
        // previous round has fired, i.e. every kernel that read its four blocks has finished.
        constexpr int kHalf = kRing / 2;
        if (slot % kHalf == 0 && pf->ringUsed[slot / kHalf]) PF_HIP_OK(hipEventSynchronize(pf->ringEv[slot / kHalf]));
        steps = pf->hSteps[slot];
        for (int f = 0; f < F; ++f) steps[f].mode = 0;
    }
    bmo = single ? pf->hBcnMo[0] : pf->hBcnMo[slot];
    bool anyMoving = false, anyUpdate = false, anySpawn = false;
    for (int q = 0; q < nReq; ++q) {
        const StepReq& r = req[q];
        if (r.f < 0 || r.f >= F) { set_error("slam deposit: filter %d not in [0,%d)", r.f, F); return -2; }
        if (r.target < 0 || r.target >= D.maxBcns) { set_error("slam deposit: beacon slot %d not in [0,%d)", r.target, D.maxBcns); return -2; }
        FilterHost& fh = pf->fl[r.f];
        SlamStep& s = single ? one : steps[r.f];
        fh.epoch++;
        s.mode = fh.bcnInit[r.target] ? 1 : 2;
        s.target = r.target;
        s.seed = fh.seed; s.epoch = fh.epoch;
        s.activeMask = r.activeMask;
        s.initMask = 0; s.movingMask = 0;
        for (int c = 0; c < D.maxBcns; ++c) {
            if (fh.bcnInit[c]) s.initMask |= 1u << c;
            if (((r.activeMask >> c) & 1u) && !is_zero_motion(r.bcnDelta[c])) {
                s.movingMask |= 1u << c;
                bmo[(size_t)r.f * D.maxBcns + c] = motion_of(r.bcnDelta[c]);
                if (fh.bcnInit[c]) anyMoving = true;
            }
        }
        s.tagMoving = is_zero_motion(r.tagDelta) ? 0 : 1;
        s.tagMo = motion_of(r.tagDelta);
        s.range = r.range; s.stdRange = r.stdRange;
        s.thresh = 3 * r.stdRange;                          // pfMeasurement.c:146
        s.minWeight = (r.range < 3.0f) ? 0.1f : 0.5f;       // pfMeasurement.c:22
        if (s.mode == 1) anyUpdate = true; else anySpawn = true;
        fh.bcnInit[r.target] = true;                        // particleFilter.c:208
    }
    const SlamStep* dMany = nullptr;
    const Motion* dMo = pf->dBcnMo;
    if (!single) {
        // (the slot's device blocks are free: ringEv[slot], waited for above, was recorded behind the last kernel that read them)
        PF_HIP_OK(hipMemcpyAsync(pf->dStepsRing[slot], steps, (size_t)F * sizeof(SlamStep), hipMemcpyHostToDevice, pf->copyStream));
        if (anyMoving)
            PF_HIP_OK(hipMemcpyAsync(pf->dBcnMoRing[slot], bmo, (size_t)F * D.maxBcns * sizeof(Motion), hipMemcpyHostToDevice, pf->copyStream));
        PF_HIP_OK(hipEventRecord(pf->copyEv[slot], pf->copyStream));
        // The HOST waits for the copy (~10 us; the GPU is a few deposits behind and does not notice), not the stream: a
        // cross-queue wait in front of the row kernel cost 10.5 us of idle GPU per deposit (kernel trace, config 5) --
        // as much as the in-stream blit this ring replaced; same-box A/B 0.2555 -> 0.2517 ms per step.
        PF_HIP_OK(hipEventSynchronize(pf->copyEv[slot]));
        dMany = pf->dStepsRing[slot];
        dMo = pf->dBcnMoRing[slot];
    } else if (anyMoving) {
        // one filter: the motions are few; a blocking copy keeps the pinned block reusable at once
        PF_HIP_OK(hipMemcpyAsync(pf->dBcnMo, bmo, (size_t)D.maxBcns * sizeof(Motion), hipMemcpyHostToDevice, pf->stream));
        PF_HIP_OK(hipStreamSynchronize(pf->stream));
    }
    if (anyUpdate) {
        if (anyMoving)
            hipLaunchKernelGGL(k_slam_gate_rows_moving, dim3(D.nTag, F), dim3(kThreads), 0, pf->stream, pf->P, D, one, dMany, dMo,
                               pf->rowSum, pf->rowScale, pf->rowFlag, pf->forcedList + F);
        else
            hipLaunchKernelGGL(k_slam_gate_rows_static, dim3(D.nTag, F), dim3(kThreads), 0, pf->stream, pf->P, D, one, dMany, dMo,
                               pf->rowSum, pf->rowScale, pf->rowFlag, pf->forcedList + F);
        if (check_launch("k_slam_gate_rows")) return -1;
    }
    hipLaunchKernelGGL(k_slam_tag, dim3(F), dim3(kThreads), 0, pf->stream, pf->P, D, one, dMany, pf->rowSum, pf->tagBranch, pf->forcedList,
                       pf->forcedList + F);
    if (check_launch("k_slam_tag")) return -1;
    if (anySpawn || anyMoving) {
        hipLaunchKernelGGL(k_slam_rows, dim3(D.nTag, F), dim3(kThreads), 0, pf->stream, pf->P, D, one, dMany, dMo, pf->rowScale,
                           pf->rowFlag);
        if (check_launch("k_slam_rows")) return -1;
    }
    if (anyUpdate) {
        hipLaunchKernelGGL(k_slam_rows_forced, dim3(D.nTag, F < 64 ? F : 64), dim3(kThreads), 0, pf->stream, pf->P, D, one, dMany,
                           dMo, pf->rowScale, pf->rowFlag, pf->forcedList, pf->forcedList + F);
        if (check_launch("k_slam_rows_forced")) return -1;
    }
    if (!single) {
        constexpr int kHalf = kRing / 2;
        if (slot % kHalf == kHalf - 1) {
            PF_HIP_OK(hipEventRecord(pf->ringEv[slot / kHalf], pf->stream));
            pf->ringUsed[slot / kHalf] = true;
        }
    }
    pf->deposits++;
    return 0;
}

int check_filter(const pfx_slam* pf, int f, const char* fn)
{
    if (f < 0 || f >= pf->D.numFilters) { set_error("%s: filter %d not in [0,%d)", fn, f, pf->D.numFilters); return -2; }
    return 0;
}
int check_slot(const pfx_slam* pf, int f, int slot, const char* fn)
{
    if (check_filter(pf, f, fn)) return -2;
    if (slot < 0 || slot >= pf->fl[f].numBcns) { set_error("%s: beacon slot %d not in [0,%d)", fn, slot, pf->fl[f].numBcns); return -2; }
    return 0;
}

// particleFilter.c:321-333 on given tag pose sums and VIO accumulator
void tag_pose_finish(const double p[6], const VioAcc& v, double* t, float* x, float* y, float* z, float* theta)
{
    const float s = (float)p[0], xs = (float)p[1], ys = (float)p[2], zs = (float)p[3], cs = (float)p[4], sn = (float)p[5];
    *t = v.lastT;
    *x = xs / s; *y = ys / s; *z = zs / s;
    *theta = atan2f(sn, cs);
    const float dx = v.lastX - v.firstX, dy = v.lastY - v.firstY, dz = v.lastZ - v.firstZ;
    const float co = cosf(*theta), si = sinf(*theta);
    *x += dx * co - dy * si;
    *y += dx * si + dy * co;
    *z += dz;
}

int tag_pose_sums(pfx_slam* pf)
{   // all filters at once -> hScratch (F * 6 doubles)
    const int F = pf->D.numFilters;
    hipLaunchKernelGGL(k_slam_tag_pose, dim3(F), dim3(kThreads), 0, pf->stream, pf->P, pf->D, pf->tagPose);
    if (check_launch("k_slam_tag_pose")) return -1;
    PF_HIP_OK(hipMemcpyAsync(pf->hScratch, pf->tagPose, (size_t)F * 6 * sizeof(double), hipMemcpyDeviceToHost, pf->stream));
    PF_HIP_OK(hipStreamSynchronize(pf->stream));
    return 0;
}

int bcn_pose_sums(pfx_slam* pf, const int* slots)
{   // slots[f] = beacon slot or -1 -> hScratch (F * 6 doubles)
    const SlamDims& D = pf->D;
    const int F = D.numFilters;
    PF_HIP_OK(hipMemcpyAsync(pf->slotOf, slots, (size_t)F * sizeof(int), hipMemcpyHostToDevice, pf->stream));
    PF_HIP_OK(hipStreamSynchronize(pf->stream));        // `slots` may be pageable and short-lived
    hipLaunchKernelGGL(k_slam_bcn_rows_pose, dim3(D.nTag, F), dim3(kThreads), 0, pf->stream, pf->P, D, pf->slotOf, pf->rowScale, pf->rowTerms);
    if (check_launch("k_slam_bcn_rows_pose")) return -1;
    hipLaunchKernelGGL(k_slam_bcn_pose, dim3(F), dim3(kThreads), 0, pf->stream, D, pf->slotOf, pf->rowTerms, pf->bcnPose);
    if (check_launch("k_slam_bcn_pose")) return -1;
    PF_HIP_OK(hipMemcpyAsync(pf->hScratch, pf->bcnPose, (size_t)F * 6 * sizeof(double), hipMemcpyDeviceToHost, pf->stream));
    PF_HIP_OK(hipStreamSynchronize(pf->stream));
    return 0;
}

void bcn_pose_finish(const double p[6], double lastT, double* t, float* x, float* y, float* z, float* theta)
{   // particleFilter.c:379-384
    const float s1 = (float)p[0];
    *t = lastT;
    *x = (float)p[1] / s1; *y = (float)p[2] / s1; *z = (float)p[3] / s1;
    *theta = atan2f((float)p[5], (float)p[4]);
}

}  // namespace

// ------------------------------------------------------------------ create / destroy
extern "C" pfx_slam* pfx_slam_create(int num_filters, int n_tag, int n_bcn, int max_bcns, const uint32_t* seeds, uint32_t seed0,
                                     int device)
{
    clear_error();
    if (num_filters < 1 || n_tag < 1 || n_tag > kTile || n_bcn < 1 || n_bcn > kTile || max_bcns < 1 || max_bcns > kSlamMaxBcns) {
        set_error("pfx_slam_create: need num_filters >= 1, 1 <= n_tag, n_bcn <= %d, 1 <= max_bcns <= %d", kTile, kSlamMaxBcns);
        return nullptr;
    }
    const int ndev = pfx_device_count();
    if (ndev <= 0) { set_error("pfx_slam_create: no HIP device visible"); return nullptr; }
    if (device < 0 || device >= ndev) { set_error("pfx_slam_create: device %d not in [0,%d)", device, ndev); return nullptr; }
    pfx_slam* pf = new (std::nothrow) pfx_slam();
    if (!pf) { set_error("pfx_slam_create: out of host memory"); return nullptr; }
    pf->device = device;
    pf->D = SlamDims{ num_filters, max_bcns, n_tag, n_bcn, (n_tag + 3) & ~3, (n_bcn + 3) & ~3 };
    pf->fl.resize((size_t)num_filters);
    for (int f = 0; f < num_filters; ++f) pf->fl[f].seed = seeds ? seeds[f] : seed0 + (uint32_t)f;
    DeviceGuard g(device);
    if (alloc_slam(pf)) { free_slam(pf); delete pf; return nullptr; }
    std::lock_guard<std::mutex> lk(g_smu);
    g_slive.insert(pf);
    return pf;
}

extern "C" void pfx_slam_destroy(pfx_slam* pf)
{
    if (!pf) return;
    {
        std::lock_guard<std::mutex> lk(g_smu);
        if (!g_slive.erase(pf)) return;
    }
    DeviceGuard g(pf->device);
    if (pf->stream) hipStreamSynchronize(pf->stream);
    free_slam(pf);
    delete pf;
}

extern "C" int pfx_slam_num_filters(const pfx_slam* pf) { return pf->D.numFilters; }
extern "C" int pfx_slam_n_tag(const pfx_slam* pf) { return pf->D.nTag; }
extern "C" int pfx_slam_n_bcn(const pfx_slam* pf) { return pf->D.nBcn; }
extern "C" int pfx_slam_max_bcns(const pfx_slam* pf) { return pf->D.maxBcns; }

extern "C" int pfx_slam_sync(pfx_slam* pf)
{
    clear_error();
    DeviceGuard g(pf->device);
    PF_HIP_OK(hipStreamSynchronize(pf->stream));
    return 0;
}

// particleFilterSlam_init of filter f (particleFilter.c:48-63): VIO scalars zeroed, tag particles at the
// origin with w = 1; its beacons are forgotten (slots handed out again by pfx_slam_add_bcn)
extern "C" int pfx_slam_reset(pfx_slam* pf, int f, uint32_t seed)
{
    clear_error();
    if (check_filter(pf, f, "pfx_slam_reset")) return -2;
    DeviceGuard g(pf->device);
    pf->fl[f] = FilterHost();
    pf->fl[f].seed = seed;
    if (f == 0) pf->legacySlots.clear();
    hipLaunchKernelGGL(k_slam_tag_init, dim3((unsigned)((pf->D.tagStride + 255) / 256)), dim3(256), 0, pf->stream, pf->P, pf->D, f, 1);
    return check_launch("k_slam_tag_init");
}

extern "C" int pfx_slam_add_bcn(pfx_slam* pf, int f)
{   // particleFilterSlam_addBcn (particleFilter.c:65-68): a beacon that is not initialised yet
    clear_error();
    if (check_filter(pf, f, "pfx_slam_add_bcn")) return -2;
    FilterHost& fh = pf->fl[f];
    if (fh.numBcns >= pf->D.maxBcns) { set_error("pfx_slam_add_bcn: filter %d already has %d beacons", f, fh.numBcns); return -3; }
    const int slot = fh.numBcns++;
    fh.bcnInit[slot] = false;
    fh.bcnVio[slot] = VioAcc();
    return slot;
}

extern "C" int pfx_slam_bcn_initialized(const pfx_slam* pf, int f, int slot)
{
    if (f < 0 || f >= pf->D.numFilters || slot < 0 || slot >= pf->D.maxBcns) return 0;
    return pf->fl[f].bcnInit[slot] ? 1 : 0;
}
extern "C" uint32_t pfx_slam_epoch(const pfx_slam* pf, int f) { return pf->fl[f].epoch; }
extern "C" void pfx_slam_set_epoch(pfx_slam* pf, int f, uint32_t e) { pf->fl[f].epoch = e; }
extern "C" uint32_t pfx_slam_seed(const pfx_slam* pf, int f) { return pf->fl[f].seed; }

// ------------------------------------------------------------------ the reference's operations, per filter
extern "C" int pfx_slam_deposit_tag_vio(pfx_slam* pf, int f, double t, float x, float y, float z, float dist)
{
    clear_error();
    if (check_filter(pf, f, "pfx_slam_deposit_tag_vio")) return -2;
    vio_deposit(pf->fl[f].vio, t, x, y, z, dist);
    return 0;
}

extern "C" int pfx_slam_deposit_bcn_vio(pfx_slam* pf, int f, int slot, double t, float x, float y, float z, float dist)
{
    clear_error();
    if (check_slot(pf, f, slot, "pfx_slam_deposit_bcn_vio")) return -2;
    vio_deposit(pf->fl[f].bcnVio[slot], t, x, y, z, dist);
    return 0;
}

namespace {
// commit the VIO accumulators of filter f and all its beacons into a request (particleFilter.c:197-199)
void fill_req(pfx_slam* pf, int f, int slot, float range, float stdRange, StepReq& r)
{
    FilterHost& fh = pf->fl[f];
    r.f = f; r.target = slot; r.range = range; r.stdRange = stdRange;
    vio_commit(fh.vio, r.tagDelta);
    r.activeMask = fh.numBcns >= 32 ? 0xFFFFFFFFu : ((1u << fh.numBcns) - 1u);
    for (int c = 0; c < fh.numBcns; ++c) vio_commit(fh.bcnVio[c], r.bcnDelta[c]);
}
}  // namespace

extern "C" int pfx_slam_deposit_range(pfx_slam* pf, int f, int slot, float range, float stdRange)
{   // particleFilter.c:193-211 with allBcns = every beacon added to the filter
    clear_error();
    if (check_slot(pf, f, slot, "pfx_slam_deposit_range")) return -2;
    DeviceGuard g(pf->device);
    StepReq r{};
    fill_req(pf, f, slot, range, stdRange, r);
    return slam_step(pf, &r, 1);
}

extern "C" int pfx_slam_deposit_rssi(pfx_slam* pf, int f, int slot, int rssi)
{   // particleFilter.c:228-246: the rssi value is not used; a 1.5 m +- 0.5 m range is applied
    (void)rssi;
    return pfx_slam_deposit_range(pf, f, slot, 1.5f, 0.5f);
}

// ------------------------------------------------------------------ the same for the whole batch in one go
extern "C" int pfx_slam_deposit_tag_vio_batch(pfx_slam* pf, const double* t, const float* x, const float* y, const float* z,
                                              const float* dist)
{
    clear_error();
    for (int f = 0; f < pf->D.numFilters; ++f) vio_deposit(pf->fl[f].vio, t[f], x[f], y[f], z[f], dist ? dist[f] : 0.0f);
    return 0;
}

extern "C" int pfx_slam_deposit_range_batch(pfx_slam* pf, const int32_t* slot, const float* range, const float* stdRange)
{   // slot[f] < 0: filter f has no measurement in this deposit (its VIO keeps accumulating)
    clear_error();
    DeviceGuard g(pf->device);
    const int F = pf->D.numFilters;
    std::vector<StepReq> reqs;
    reqs.reserve((size_t)F);
    for (int f = 0; f < F; ++f) {
        if (slot[f] < 0) continue;
        if (check_slot(pf, f, slot[f], "pfx_slam_deposit_range_batch")) return -2;
        reqs.emplace_back();
        fill_req(pf, f, slot[f], range[f], stdRange[f], reqs.back());
    }
    if (reqs.empty()) return 0;
    return slam_step(pf, reqs.data(), (int)reqs.size());
}

// ------------------------------------------------------------------ pose estimates
extern "C" int pfx_slam_get_tag_loc(pfx_slam* pf, int f, double* t, float* x, float* y, float* z, float* theta)
{   // particleFilter.c:293-336 (a SLAM filter is initialised from the start, :62)
    clear_error();
    if (check_filter(pf, f, "pfx_slam_get_tag_loc")) return -2;
    DeviceGuard g(pf->device);
    if (tag_pose_sums(pf)) return -1;
    tag_pose_finish((const double*)pf->hScratch + (size_t)f * 6, pf->fl[f].vio, t, x, y, z, theta);
    return 1;
}

extern "C" int pfx_slam_get_tag_loc_batch(pfx_slam* pf, double* t, float* x, float* y, float* z, float* theta)
{
    clear_error();
    DeviceGuard g(pf->device);
    if (tag_pose_sums(pf)) return -1;
    for (int f = 0; f < pf->D.numFilters; ++f)
        tag_pose_finish((const double*)pf->hScratch + (size_t)f * 6, pf->fl[f].vio, &t[f], &x[f], &y[f], &z[f], &theta[f]);
    return 1;
}

extern "C" int pfx_slam_get_bcn_loc(pfx_slam* pf, int f, int slot, double* t, float* x, float* y, float* z, float* theta)
{   // particleFilter.c:338-387; 0 if the beacon has not been initialised by a first range
    clear_error();
    if (check_slot(pf, f, slot, "pfx_slam_get_bcn_loc")) return -2;
    if (!pf->fl[f].bcnInit[slot]) return 0;
    DeviceGuard g(pf->device);
    std::vector<int> slots((size_t)pf->D.numFilters, -1);
    slots[(size_t)f] = slot;
    if (bcn_pose_sums(pf, slots.data())) return -1;
    bcn_pose_finish((const double*)pf->hScratch + (size_t)f * 6, pf->fl[f].vio.lastT, t, x, y, z, theta);
    return 1;
}

extern "C" int pfx_slam_get_bcn_loc_batch(pfx_slam* pf, const int32_t* slot, uint8_t* ok, double* t, float* x, float* y, float* z,
                                          float* theta)
{
    clear_error();
    DeviceGuard g(pf->device);
    const int F = pf->D.numFilters;
    std::vector<int> slots((size_t)F, -1);
    for (int f = 0; f < F; ++f) {
        ok[f] = 0;
        if (slot[f] >= 0 && slot[f] < pf->fl[f].numBcns && pf->fl[f].bcnInit[slot[f]]) { slots[(size_t)f] = slot[f]; ok[f] = 1; }
    }
    if (bcn_pose_sums(pf, slots.data())) return -1;
    for (int f = 0; f < F; ++f)
        if (ok[f]) bcn_pose_finish((const double*)pf->hScratch + (size_t)f * 6, pf->fl[f].vio.lastT, &t[f], &x[f], &y[f], &z[f], &theta[f]);
    return 1;
}

extern "C" int pfx_slam_last_tag_branch(pfx_slam* pf, int f)
{   // 1 if the last deposit of filter f resampled its tag particles (pfResample.c:110)
    clear_error();
    if (check_filter(pf, f, "pfx_slam_last_tag_branch")) return -2;
    DeviceGuard g(pf->device);
    int* h = (int*)pf->hScratch;
    PF_HIP_OK(hipMemcpyAsync(h, pf->tagBranch + f, sizeof(int), hipMemcpyDeviceToHost, pf->stream));
    PF_HIP_OK(hipStreamSynchronize(pf->stream));
    return *h;
}

// What the last deposit did to the rows of the measured beacons (measurement aid: bench.py prices a deposit by it).
// out[0] = rows that only normalised, out[1] = rows that resampled, out[2] = filters whose tag particles resampled (every
// initialised beacon of those was force-resampled, pfResample.c:122-130), out[3] = filters that took part.  Synchronises.
extern "C" int pfx_slam_row_census(pfx_slam* pf, int64_t out[4])
{
    clear_error();
    DeviceGuard g(pf->device);
    const size_t F = (size_t)pf->D.numFilters, nTag = (size_t)pf->D.nTag;
    std::vector<int> flag(F * nTag), br(F + 1);
    PF_HIP_OK(hipMemcpyAsync(flag.data(), pf->rowFlag, flag.size() * sizeof(int), hipMemcpyDeviceToHost, pf->stream));
    PF_HIP_OK(hipMemcpyAsync(br.data(), pf->forcedList, (F + 1) * sizeof(int), hipMemcpyDeviceToHost, pf->stream));
    PF_HIP_OK(hipStreamSynchronize(pf->stream));
    out[0] = out[1] = 0;
    for (int v : flag) { if (v > 0) ++out[1]; else ++out[0]; }
    out[2] = br[F];
    out[3] = (int64_t)F;
    return 0;
}

// Device pointers of the batch's SoA state for zero-copy views (Python: DeviceSlamBatch.arrays_view): tag arrays
// [filter][tagStride], beacon arrays [filter][slot][tag particle][rowStride] (= pBcn[i][j], particleFilter.h:86), and the
// pending per-row weight factor [filter][slot][tag particle] (true beacon weight = w * rowScale of its row).
// dims: numFilters, maxBcns, nTag, nBcn, tagStride, rowStride.  Synchronises the stream.
extern "C" int pfx_slam_device_arrays(pfx_slam* pf, void* d_tag[5], void* d_bcn[5], void** d_row_scale, int32_t dims[6])
{
    clear_error();
    DeviceGuard g(pf->device);
    PF_HIP_OK(hipStreamSynchronize(pf->stream));
    for (int k = 0; k < 5; ++k) { d_tag[k] = pf->P.tag[k]; d_bcn[k] = pf->P.bcn[k]; }
    *d_row_scale = pf->rowScale;
    const SlamDims& D = pf->D;
    dims[0] = D.numFilters; dims[1] = D.maxBcns; dims[2] = D.nTag; dims[3] = D.nBcn; dims[4] = D.tagStride; dims[5] = D.rowStride;
    return 0;
}

// ------------------------------------------------------------------ state access (tests, checkpoints)
namespace {
int copy_rows(pfx_slam* pf, float* const dev[5], size_t base, int rows, int n, int stride, float* const host[5], bool toHost)
{
    for (int k = 0; k < 5; ++k) {
        if (!host[k]) continue;
        if (toHost)
            PF_HIP_OK(hipMemcpy2DAsync(host[k], (size_t)n * sizeof(float), dev[k] + base, (size_t)stride * sizeof(float),
                                       (size_t)n * sizeof(float), (size_t)rows, hipMemcpyDeviceToHost, pf->stream));
        else
            PF_HIP_OK(hipMemcpy2DAsync(dev[k] + base, (size_t)stride * sizeof(float), host[k], (size_t)n * sizeof(float),
                                       (size_t)n * sizeof(float), (size_t)rows, hipMemcpyHostToDevice, pf->stream));
    }
    PF_HIP_OK(hipStreamSynchronize(pf->stream));
    return 0;
}
}  // namespace

extern "C" int pfx_slam_download_tag(pfx_slam* pf, int f, float* w, float* x, float* y, float* z, float* theta)
{
    clear_error();
    if (check_filter(pf, f, "pfx_slam_download_tag")) return -2;
    DeviceGuard g(pf->device);
    float* host[5] = { w, x, y, z, theta };
    return copy_rows(pf, pf->P.tag, (size_t)f * pf->D.tagStride, 1, pf->D.nTag, pf->D.tagStride, host, true);
}

extern "C" int pfx_slam_upload_tag(pfx_slam* pf, int f, const float* w, const float* x, const float* y, const float* z,
                                   const float* theta)
{
    clear_error();
    if (check_filter(pf, f, "pfx_slam_upload_tag")) return -2;
    DeviceGuard g(pf->device);
    float* host[5] = { (float*)w, (float*)x, (float*)y, (float*)z, (float*)theta };
    return copy_rows(pf, pf->P.tag, (size_t)f * pf->D.tagStride, 1, pf->D.nTag, pf->D.tagStride, host, false);
}

// arrays of nTag * nBcn floats, row i = the beacon particles of tag particle i (pBcn[i][j], particleFilter.h:86)
extern "C" int pfx_slam_download_bcn(pfx_slam* pf, int f, int slot, float* w, float* x, float* y, float* z, float* theta)
{
    clear_error();
    if (check_slot(pf, f, slot, "pfx_slam_download_bcn")) return -2;
    DeviceGuard g(pf->device);
    float* host[5] = { w, x, y, z, theta };
    if (copy_rows(pf, pf->P.bcn, slam_row_base(pf->D, f, slot, 0), pf->D.nTag, pf->D.nBcn, pf->D.rowStride, host, true)) return -1;
    if (w) {    // pending w *= N/s of the rows that were normalised last (pfResample.c:195-199)
        std::vector<float> sc((size_t)pf->D.nTag);
        PF_HIP_OK(hipMemcpy(sc.data(), pf->rowScale + ((size_t)f * pf->D.maxBcns + slot) * pf->D.nTag, sc.size() * sizeof(float),
                            hipMemcpyDeviceToHost));
        for (int i = 0; i < pf->D.nTag; ++i)
            if (sc[(size_t)i] != 1.0f)
                for (int j = 0; j < pf->D.nBcn; ++j) w[(size_t)i * pf->D.nBcn + j] *= sc[(size_t)i];
    }
    return 0;
}

extern "C" int pfx_slam_upload_bcn(pfx_slam* pf, int f, int slot, const float* w, const float* x, const float* y, const float* z,
                                   const float* theta, int initialized)
{
    clear_error();
    if (check_slot(pf, f, slot, "pfx_slam_upload_bcn")) return -2;
    DeviceGuard g(pf->device);
    float* host[5] = { (float*)w, (float*)x, (float*)y, (float*)z, (float*)theta };
    if (copy_rows(pf, pf->P.bcn, slam_row_base(pf->D, f, slot, 0), pf->D.nTag, pf->D.nBcn, pf->D.rowStride, host, false)) return -1;
    if (w) PF_HIP_OK(hipMemsetD32Async((hipDeviceptr_t)(pf->rowScale + ((size_t)f * pf->D.maxBcns + slot) * pf->D.nTag), 0x3f800000,
                                       (size_t)pf->D.nTag, pf->stream));
    pf->fl[f].bcnInit[slot] = initialized != 0;
    return 0;
}

// ------------------------------------------------------------------ reference-layout API (particleFilterSlam_*)
namespace {

// Binding records kept in the first bytes of the structs' scratch arrays (pTagBuf / pBcnBuf: resampling
// scratch in the reference, pfResample.c:119-120, 182-188).
//
// POD copies (the reference's structs ARE the filter).  Both records count the state-changing calls
// their struct has been through (`ops`, mirrored by the device filter) and remember the count at which
// pfx_slam_sync_host last filled the struct's particle array (`syncedOps`).  A struct met at an address, or with a count,
// other than the device filter knows -- a copy, a checkpoint read back, bytes copied back over the live struct -- is
// continued from its own pTag[] / pBcn[][] if that array is current (syncedOps == ops), and refused with a message if it
// is not.  (No device-side clone here as particleFilterLoc_t copies get: a SLAM filter with its beacons is gigabytes.)
// A restored tag filter draws new noise streams for its beacons if they land in other slots than before.
struct SlamStash { uint64_t magic; uint64_t self; pfx_slam* handle; uint64_t process; unsigned long long ops, syncedOps; uint32_t seed, epoch; };
struct BcnStash { uint64_t magic; uint64_t self; pfx_slam* owner; int64_t slot; uint64_t process; unsigned long long ops, syncedOps; };
constexpr unsigned long long kNeverSynced = ~0ull;
const uint64_t g_slamProcess = (uint64_t)time(nullptr) * 0x9E3779B97F4A7C15ull ^ (uint64_t)(uintptr_t)&g_slive ^ ((uint64_t)getpid() << 32);
constexpr uint64_t kSlamMagic = 0x7066536C6D484950ull;   // "pfSlmHIP"
constexpr uint64_t kBcnMagic = 0x706642636E484950ull;    // "pfBcnHIP"

SlamStash* stash_of(const particleFilterSlam_t* pf) { return reinterpret_cast<SlamStash*>(const_cast<tagParticle_t*>(pf->pTagBuf)); }
BcnStash* stash_of(const bcn_t* b) { return reinterpret_cast<BcnStash*>(const_cast<bcnParticle_t*>(b->pBcnBuf)); }

bool slam_live(pfx_slam* h, uint64_t process)
{
    if (process != g_slamProcess) return false;
    std::lock_guard<std::mutex> lk(g_smu);
    return g_slive.count(h) != 0;
}

// strictly this struct, at this address, bound to a live filter (whatever its contents' age)
pfx_slam* bound(const particleFilterSlam_t* pf)
{
    const SlamStash* s = stash_of(pf);
    if (s->magic != kSlamMagic || s->self != (uint64_t)(uintptr_t)pf) return nullptr;
    return slam_live(s->handle, s->process) ? s->handle : nullptr;
}

int legacy_n(const char* env, int dflt);
int legacy_device();
template <typename P> void aos_to_soa(const P* p, size_t n, std::vector<float>& out);

void bind_slam(particleFilterSlam_t* pf, pfx_slam* h)
{
    SlamStash* s = stash_of(pf);
    s->magic = kSlamMagic; s->self = (uint64_t)(uintptr_t)pf; s->handle = h; s->process = g_slamProcess;
    s->ops = h->legacyOps; s->syncedOps = kNeverSynced;
    s->seed = h->fl[0].seed; s->epoch = h->fl[0].epoch;
}

void note_slam_state(particleFilterSlam_t* pf, pfx_slam* h)
{   // after a state-changing call
    SlamStash* s = stash_of(pf);
    s->ops = ++h->legacyOps; s->seed = h->fl[0].seed; s->epoch = h->fl[0].epoch;
}

// the device filter behind `pf`; a struct that is a copy of an earlier state is first continued from its own pTag[]
pfx_slam* must_be_bound(const particleFilterSlam_t* cpf, const char* fn)
{
    particleFilterSlam_t* pf = const_cast<particleFilterSlam_t*>(cpf);
    SlamStash* s = stash_of(pf);
    if (s->magic != kSlamMagic) {
        set_error("%s: struct %p was not initialised with particleFilterSlam_init", fn, (const void*)pf);
        fail(fn);
        return nullptr;
    }
    const bool live = slam_live(s->handle, s->process);
    const bool here = s->self == (uint64_t)(uintptr_t)pf;
    if (live && here && s->ops == s->handle->legacyOps) return s->handle;       // the common case
    const SlamStash was = *s;
    if (was.syncedOps != was.ops) {
        set_error("%s: struct %p is a copy of a filter that has since moved on (or is gone, or was moved) and does not hold "
                  "its particles: call pfx_slam_sync_host() before copying a struct that is to be continued", fn, (const void*)pf);
        fail(fn);
        return nullptr;
    }
    pfx_slam* h = (live && here) ? was.handle
                                 : pfx_slam_create(1, PF_N_TAG_SLAM, PF_N_BCN, kSlamMaxBcns, nullptr, was.seed, legacy_device());
    if (!h) { fail(fn); return nullptr; }
    if (h->D.nTag != PF_N_TAG_SLAM || h->D.nBcn != PF_N_BCN) {
        set_error("%s: a stored struct cannot be continued while the particle counts are overridden", fn);
        fail(fn);
        return nullptr;
    }
    std::vector<float> tmp;
    const size_t n = PF_N_TAG_SLAM;
    aos_to_soa(&pf->pTag[0], n, tmp);
    if (pfx_slam_upload_tag(h, 0, &tmp[0], &tmp[n], &tmp[2 * n], &tmp[3 * n], &tmp[4 * n])) {
        if (!(live && here)) pfx_slam_destroy(h);
        fail(fn);
        return nullptr;
    }
    h->fl[0].seed = was.seed; h->fl[0].epoch = was.epoch;
    ++h->legacyOps;
    bind_slam(pf, h);
    stash_of(pf)->syncedOps = h->legacyOps;         // pTag[] is what the filter holds
    return h;
}

// the VIO accumulator of a reference-layout struct (the struct's own fields are authoritative)
template <typename S> VioAcc vio_load(const S* s)
{
    VioAcc v;
    v.firstT = s->firstT; v.firstX = s->firstX; v.firstY = s->firstY; v.firstZ = s->firstZ; v.firstDist = s->firstDist;
    v.lastT = s->lastT; v.lastX = s->lastX; v.lastY = s->lastY; v.lastZ = s->lastZ; v.lastDist = s->lastDist;
    return v;
}
template <typename S> void vio_store(S* s, const VioAcc& v)
{
    s->firstT = v.firstT; s->firstX = v.firstX; s->firstY = v.firstY; s->firstZ = v.firstZ; s->firstDist = v.firstDist;
    s->lastT = v.lastT; s->lastX = v.lastX; s->lastY = v.lastY; s->lastZ = v.lastZ; s->lastDist = v.lastDist;
}

int legacy_n(const char* env, int dflt)
{
    const char* e = getenv(env);
    if (e && *e) { const int v = atoi(e); if (v > 0) return v; }
    return dflt;
}
int legacy_device()
{
    const char* e = getenv("PF_HIP_DEVICE");
    return (e && *e) ? atoi(e) : 0;
}

// AoS particles of a caller's struct -> the SoA arrays pfx_slam_upload_* take
template <typename P>
void aos_to_soa(const P* p, size_t n, std::vector<float>& out)
{
    out.resize(n * 5);
    for (size_t i = 0; i < n; ++i) {
        out[i] = p[i].w; out[n + i] = p[i].x; out[2 * n + i] = p[i].y; out[3 * n + i] = p[i].z; out[4 * n + i] = p[i].theta;
    }
}

// Slot of a beacon struct inside the filter's device state; bound at its first use with that filter.  The reference
// has no limit on the beacons of a filter; here a filter holds at most kSlamMaxBcns (32) distinct bcn_t at a time.
int slot_of(pfx_slam* h, bcn_t* b, const char* fn)
{
    BcnStash* s = stash_of(b);
    const bool mine = s->magic == kBcnMagic && s->self == (uint64_t)(uintptr_t)b && s->owner == h && s->process == g_slamProcess &&
                      s->slot >= 0 && s->slot < h->fl[0].numBcns;
    if (mine && s->ops == h->bcnOps[s->slot]) { h->slotSeen[s->slot] = h->legacyOps + 1; return (int)s->slot; }
    // Not this filter's beacon as it last saw it.  A struct that carries a record of this library (a copy, a checkpoint
    // read back, bytes copied back over the live struct) must hold its particles to be continued; one without a record
    // (filled by the reference library, or reset with particleFilterSlam_addBcn) is taken at its word.
    if (s->magic == kBcnMagic && b->initialized && s->syncedOps != s->ops) {
        set_error("%s: beacon %p is a copy of a beacon that has since moved on (or belongs to a filter that is gone) and does "
                  "not hold its particles: call pfx_slam_sync_host(pf, bcn) before copying a beacon that is to be continued",
                  fn, (const void*)b);
        return -1;
    }
    int slot;
    const auto it = h->legacySlots.find((uintptr_t)b);
    if (mine) {
        slot = (int)s->slot;                // rolled back in place: its rows are replaced below
    } else if (it != h->legacySlots.end()) {
        slot = it->second;                  // reset with particleFilterSlam_addBcn: same struct, same slot, spawned afresh
    } else if (h->fl[0].numBcns < h->D.maxBcns) {
        slot = pfx_slam_add_bcn(h, 0);
        if (slot < 0) return -1;
        h->legacySlots[(uintptr_t)b] = slot;
    } else {
        // Every slot is taken.  The reference API has no call that gives a beacon up -- its callers just free() the struct
        // (slam3d-jni.c:34-37) -- so the slot that has gone unlisted in allBcns the longest, and is not part of the call at
        // hand, is taken to be such a beacon and handed on.  Should its struct come back after all, its binding record no
        // longer matches (bcnOps) and it is continued from its own pBcn[][] or refused, like any other out-of-date copy.
        slot = -1;
        for (int c = 0; c < h->D.maxBcns; ++c)
            if (h->slotSeen[c] <= h->legacyOps && (slot < 0 || h->slotSeen[c] < h->slotSeen[slot])) slot = c;
        if (slot < 0) {
            set_error("%s: more than %d distinct beacons in one call", fn, h->D.maxBcns);
            return -1;
        }
        for (auto it2 = h->legacySlots.begin(); it2 != h->legacySlots.end();)
            it2 = it2->second == slot ? h->legacySlots.erase(it2) : std::next(it2);
        h->legacySlots[(uintptr_t)b] = slot;
        h->fl[0].bcnVio[slot] = VioAcc();
    }
    h->slotSeen[slot] = h->legacyOps + 1;       // part of the call at hand (legacy_deposit bumps legacyOps at its end)
    h->fl[0].bcnInit[slot] = false;
    if (b->initialized) {
        // A beacon that says it holds particles but has no rows on the device (a struct filled by the reference
        // library, or a copy of one taken after pfx_slam_sync_host): its own arrays are the state, as in the reference.
        if (h->D.nTag != PF_N_TAG_SLAM || h->D.nBcn != PF_N_BCN) {
            set_error("%s: beacon %p is initialised but not on the device, and the particle counts were overridden", fn, (const void*)b);
            return -1;
        }
        const size_t n = (size_t)PF_N_TAG_SLAM * PF_N_BCN;
        std::vector<float> tmp;
        aos_to_soa(&b->pBcn[0][0], n, tmp);
        // binding record first: pBcnBuf, where it lives, is not part of the state
        if (pfx_slam_upload_bcn(h, 0, slot, &tmp[0], &tmp[n], &tmp[2 * n], &tmp[3 * n], &tmp[4 * n], 1)) return -1;
    }
    s->magic = kBcnMagic; s->self = (uint64_t)(uintptr_t)b; s->owner = h; s->slot = slot; s->process = g_slamProcess;
    s->ops = ++h->bcnOps[slot];
    s->syncedOps = b->initialized ? s->ops : kNeverSynced;      // uploaded from pBcn[][]: the struct holds what the device holds
    return slot;
}

void legacy_deposit(particleFilterSlam_t* pf, bcn_t* bcn, float range, float stdRange, bcn_t** allBcns, int numBcns, const char* fn)
{   // particleFilter.c:193-211
    pfx_slam* h = must_be_bound(pf, fn);
    if (!h) return;
    DeviceGuard g(h->device);
    // every beacon gets its slot before anything is committed, so that a failure leaves the call without effect
    int slots[kSlamMaxBcns];
    if (numBcns > kSlamMaxBcns) {
        set_error("%s: %d beacons in allBcns, this library holds at most %d per filter", fn, numBcns, kSlamMaxBcns);
        fail(fn);
        return;
    }
    for (int i = 0; i < numBcns; ++i)
        if ((slots[i] = slot_of(h, allBcns[i], fn)) < 0) { fail(fn); return; }
    const int target = slot_of(h, bcn, fn);
    if (target < 0) { fail(fn); return; }
    StepReq r{};
    r.f = 0; r.range = range; r.stdRange = stdRange;
    // the VIO accumulators are committed into temporaries and stored only once the launches have succeeded: a failing
    // call (pfx_set_abort_on_error(0)) leaves the structs as it found them
    VioAcc v = vio_load(pf);
    vio_commit(v, r.tagDelta);
    VioAcc bvs[kSlamMaxBcns];
    r.activeMask = 0;
    for (int i = 0; i < numBcns; ++i) {
        const int c = slots[i];
        bvs[i] = vio_load(allBcns[i]);
        vio_commit(bvs[i], r.bcnDelta[c]);
        r.activeMask |= 1u << c;
        h->fl[0].bcnInit[c] = allBcns[i]->initialized != 0;    // the struct's flag is authoritative
    }
    r.target = target;
    const bool unlisted = !((r.activeMask >> target) & 1u);
    VioAcc tv{};
    if (unlisted) {
        // The measured beacon is not one of allBcns.  Every caller in the reference lists it (csvslam.c:78-85,
        // slam3d-jni.c:52-70); the reference would then skip its VIO commit and, when the tag particles resample, leave
        // its gated weights un-normalised (pfResample.c:122-130).  Here it is treated as listed: one rule for the fused
        // row pass, stated in INTEGRATION.md.
        tv = vio_load(bcn);
        vio_commit(tv, r.bcnDelta[target]);
        r.activeMask |= 1u << target;
    }
    h->fl[0].bcnInit[r.target] = bcn->initialized != 0;
    if (slam_step(h, &r, 1)) { fail(fn); return; }
    vio_store(pf, v);
    for (int i = 0; i < numBcns; ++i) vio_store(allBcns[i], bvs[i]);
    if (unlisted) vio_store(bcn, tv);
    bcn->initialized = 1;
    note_slam_state(pf, h);
    for (int i = 0; i < numBcns; ++i) stash_of(allBcns[i])->ops = ++h->bcnOps[slots[i]];
    bool listed = false;
    for (int i = 0; i < numBcns; ++i) listed |= (slots[i] == target);
    if (!listed) stash_of(bcn)->ops = ++h->bcnOps[target];
}

}  // namespace

extern "C" void particleFilterSlam_init(particleFilterSlam_t* pf)
{   // particleFilter.c:48-63 + pfRandom.c:56-60 (seed from the clock unless one was set).  Filters initialised after one
    // particleFilterSeed_set share Philox key word 1 = 0 for their tag particles and slot + 1 for their beacons, so the
    // n-th of them gets its own key word 0: the seed plus n golden-ratio steps (n = 0: the seed itself).
    PF_API_LOCK;
    if (!PF_SEED_SET) PF_SEED = unset_seed();
    const uint32_t seed = PF_SEED + 0x9E3779B9u * next_filter_stream();
    pfx_slam* h = bound(pf);
    if (h) {
        // The reference resets the tag particles and the VIO scalars and nothing else: the beacons are structs of
        // their own and keep their particles (and here their slots).
        DeviceGuard g(h->device);
        FilterHost& fh = h->fl[0];
        fh.vio = VioAcc();
        fh.seed = seed;
        fh.epoch = 0;
        hipLaunchKernelGGL(k_slam_tag_init, dim3((unsigned)((h->D.tagStride + 255) / 256)), dim3(256), 0, h->stream, h->P, h->D, 0, 1);
        if (check_launch("k_slam_tag_init")) { fail("particleFilterSlam_init"); return; }
    } else {
        h = pfx_slam_create(1, legacy_n("PF_HIP_N_TAG_SLAM", PF_N_TAG_SLAM), legacy_n("PF_HIP_N_BCN", PF_N_BCN), kSlamMaxBcns, nullptr,
                            seed, legacy_device());
        if (!h) { fail("particleFilterSlam_init"); return; }
    }
    ++h->legacyOps;
    bind_slam(pf, h);
    vio_store(pf, VioAcc());
    pf->initialized = 1;
}

extern "C" void particleFilterSlam_addBcn(bcn_t* bcn)
{   // particleFilter.c:65-68: only clears `initialized`; the beacon meets its filter at the first deposit
    PF_API_LOCK;
    bcn->initialized = 0;
    BcnStash* s = stash_of(bcn);
    s->magic = 0; s->self = 0; s->owner = nullptr; s->slot = -1;
}

extern "C" void particleFilterSlam_depositTagVio(particleFilterSlam_t* pf, double t, float x, float y, float z, float dist)
{   // particleFilter.c:106-140: host bookkeeping in the caller's struct
    PF_API_LOCK;
    VioAcc v = vio_load(pf);
    vio_deposit(v, t, x, y, z, dist);
    vio_store(pf, v);
}

extern "C" void particleFilterSlam_depositBcnVio(bcn_t* bcn, double t, float x, float y, float z, float dist)
{   // particleFilter.c:142-176
    PF_API_LOCK;
    VioAcc v = vio_load(bcn);
    vio_deposit(v, t, x, y, z, dist);
    vio_store(bcn, v);
}

extern "C" void particleFilterSlam_depositRange(particleFilterSlam_t* pf, bcn_t* bcn, float range, float stdRange, bcn_t** allBcns,
                                                int numBcns)
{
    PF_API_LOCK;
    legacy_deposit(pf, bcn, range, stdRange, allBcns, numBcns, "particleFilterSlam_depositRange");
}

extern "C" void particleFilterSlam_depositRssi(particleFilterSlam_t* pf, bcn_t* bcn, int rssi, bcn_t** allBcns, int numBcns)
{   // particleFilter.c:228-246
    PF_API_LOCK;
    (void)rssi;
    legacy_deposit(pf, bcn, 1.5f, 0.5f, allBcns, numBcns, "particleFilterSlam_depositRssi");
}

extern "C" uint8_t particleFilterSlam_getTagLoc(const particleFilterSlam_t* pf, double* t, float* x, float* y, float* z, float* theta)
{
    PF_API_LOCK;
    if (!pf->initialized) return 0;
    pfx_slam* h = must_be_bound(pf, "particleFilterSlam_getTagLoc");
    if (!h) return 0;
    DeviceGuard g(h->device);
    if (tag_pose_sums(h)) { fail("particleFilterSlam_getTagLoc"); return 0; }
    tag_pose_finish((const double*)h->hScratch, vio_load(pf), t, x, y, z, theta);
    return 1;
}

extern "C" uint8_t particleFilterSlam_getBcnLoc(const particleFilterSlam_t* pf, const bcn_t* bcn, double* t, float* x, float* y,
                                                float* z, float* theta)
{
    PF_API_LOCK;
    if (!bcn->initialized) return 0;
    pfx_slam* h = must_be_bound(pf, "particleFilterSlam_getBcnLoc");
    if (!h) return 0;
    DeviceGuard g(h->device);
    // (a beacon that holds particles this filter has not seen -- a stored struct being continued -- is taken in here)
    const int slot = slot_of(h, const_cast<bcn_t*>(bcn), "particleFilterSlam_getBcnLoc");
    if (slot < 0) { fail("particleFilterSlam_getBcnLoc"); return 0; }
    if (bcn_pose_sums(h, &slot)) { fail("particleFilterSlam_getBcnLoc"); return 0; }
    bcn_pose_finish((const double*)h->hScratch, pf->lastT, t, x, y, z, theta);
    return 1;
}

extern "C" pfx_slam* pfx_slam_of(const particleFilterSlam_t* pf) { return bound(pf); }
extern "C" int pfx_slam_slot_of(const particleFilterSlam_t* pf, const bcn_t* bcn)
{
    pfx_slam* h = bound(pf);
    const BcnStash* s = stash_of(bcn);
    if (!h || s->magic != kBcnMagic || s->self != (uint64_t)(uintptr_t)bcn || s->owner != h) return -1;
    return (int)s->slot;
}

extern "C" int pfx_slam_sync_host(particleFilterSlam_t* pf, bcn_t* bcn)
{   // fill pf->pTag[] and (if given) bcn->pBcn[][] (AoS) from HBM; needs the reference's compile-time counts
    PF_API_LOCK;
    clear_error();
    pfx_slam* h = bound(pf);
    if (!h) { set_error("pfx_slam_sync_host: struct is not bound"); return -2; }
    if (h->D.nTag != PF_N_TAG_SLAM || h->D.nBcn != PF_N_BCN) { set_error("pfx_slam_sync_host: particle counts were overridden"); return -2; }
    {
        std::vector<float> tmp((size_t)PF_N_TAG_SLAM * 5);
        const int n = PF_N_TAG_SLAM;
        if (pfx_slam_download_tag(h, 0, &tmp[0], &tmp[n], &tmp[2 * n], &tmp[3 * n], &tmp[4 * n])) return -1;
        for (int i = 0; i < n; ++i) {
            pf->pTag[i].w = tmp[i]; pf->pTag[i].x = tmp[n + i]; pf->pTag[i].y = tmp[2 * n + i];
            pf->pTag[i].z = tmp[3 * n + i]; pf->pTag[i].theta = tmp[4 * n + i];
        }
        SlamStash* s = stash_of(pf);
        s->ops = h->legacyOps; s->seed = h->fl[0].seed; s->epoch = h->fl[0].epoch;
        s->syncedOps = s->ops;              // until the next state-changing call the struct is a complete checkpoint
    }
    if (bcn) {
        const int slot = pfx_slam_slot_of(pf, bcn);
        if (slot < 0) { set_error("pfx_slam_sync_host: beacon is not bound to this filter"); return -2; }
        const size_t n = (size_t)PF_N_TAG_SLAM * PF_N_BCN;
        std::vector<float> tmp(n * 5);
        if (pfx_slam_download_bcn(h, 0, slot, &tmp[0], &tmp[n], &tmp[2 * n], &tmp[3 * n], &tmp[4 * n])) return -1;
        bcnParticle_t* p = &bcn->pBcn[0][0];
        for (size_t i = 0; i < n; ++i) {
            p[i].w = tmp[i]; p[i].x = tmp[n + i]; p[i].y = tmp[2 * n + i]; p[i].z = tmp[3 * n + i]; p[i].theta = tmp[4 * n + i];
        }
        BcnStash* bs = stash_of(bcn);
        bs->ops = h->bcnOps[slot];
        bs->syncedOps = bs->ops;
    }
    return 0;
}

extern "C" void pfx_slam_release(particleFilterSlam_t* pf)
{
    PF_API_LOCK;
    pfx_slam* h = bound(pf);
    if (!h) return;
    pfx_slam_destroy(h);
    stash_of(pf)->handle = nullptr;         // the record stays: a struct that was synced to the host can still be continued
}

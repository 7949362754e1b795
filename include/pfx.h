/*
 * pfx.h -- additive extension API of libparticlefilter.so (no counterpart in the reference unless a
 * function says so).  The reference sizes every filter at compile time and offers no create/destroy,
 * no error channel and no access to the pieces of a step; this header supplies those for the GPU
 * implementation: runtime-N filters, explicit device/stream control, state upload/download, the
 * individual stages of a step (for parity tests against the oracle), and kernel timing.
 *
 * Conventions: plain C ABI, host pointers unless a parameter is named d_* (device pointer);
 * functions returning int give 0 on success and a negative code on failure, after which
 * pfx_last_error() describes it.  Argument errors (-2, -3) are found before anything changes.  A
 * failure of the GPU side (-1: a launch or copy refused by the runtime) leaves pfx_loc_deposit_range /
 * _rssi and the reference-layout entry points without effect (VIO accumulator and deposit count as
 * before); a batch (pfx_loc_batch_*, pfx_slam_*_batch) has by then consumed the VIO deltas of the
 * filters of that deposit and is to be destroyed, as is any filter after a failed collective of the
 * sharded path.  A filter is not thread-safe through this header (neither is the reference's:
 * one global rand_r stream); all work of one filter is ordered on its own HIP stream and is
 * asynchronous to the host except the calls that return data.
 */
#ifndef PFX_H
#define PFX_H

#include <stdint.h>
#include "particleFilter.h"

#ifdef __cplusplus
extern "C" {
#endif

typedef struct pfx_loc pfx_loc;         /* a particleFilterLoc with its particles in HBM */

/* Philox sub-stream selectors (counter word 3).  Same values as oracle/pf_oracle.h ORA_P_*. */
enum { PFX_P_VIO = 1, PFX_P_RSTART = 2, PFX_P_JITTER = 3, PFX_P_SPAWN = 4 };

/* kernel classes reported by pfx_loc_profile_read */
enum { PFX_K_STREAM = 0,        /* k_loc_stream with VIO motion (reads and writes all five arrays) */
       PFX_K_DECIDE = 1, PFX_K_RESAMPLE = 2, PFX_K_SPAWN = 3, PFX_K_POSE = 4,
       PFX_K_STREAM_STATIC = 5, /* k_loc_stream without motion (writes the weights only) */
       PFX_K_DEPOSIT = 6,       /* k_loc_deposit with motion: the deferred resampling of the previous deposit fused with
                                   this one, or the streaming pass, as the previous decision (on the device) says */
       PFX_K_DEPOSIT_STATIC = 7,    /* the same without motion */
       PFX_K_COUNT = 8 };

/* ---- library ---- */
const char* pfx_version(void);
const char* pfx_last_error(void);                   /* thread-local, "" if none */
int         pfx_device_count(void);                 /* gfx950 devices visible; 0 = none (every create fails) */
/* What the reference-layout entry points (particleFilterLoc_*, particleFilterSlam_*: void, no error channel in the
 * reference) do when the GPU side fails or a struct is not bound.  1 (default): print the message and abort() -- a
 * failure must not pass for a result.  0: keep the message for pfx_last_error(), count it, and return with the call
 * having had no effect (get*Loc return 0). */
void        pfx_set_abort_on_error(int enable);
long long   pfx_error_count(void);                  /* failures swallowed since the process started (mode 0 only) */

/* ---- runtime-N Loc filter ---- */
/* particleFilterLoc_init with a particle count and placement chosen at run time
 * (reference: particleFilter.c:32-46, PF_N_TAG_LOC fixed at particleFilter.h:18).
 * `stream_id` separates the noise of filters that share a seed (Philox key word 1). */
pfx_loc* pfx_loc_create(int64_t n, uint32_t seed, uint32_t stream_id, int device);
void     pfx_loc_destroy(pfx_loc* pf);
/* A second filter with the same particles, VIO accumulator, seed, sub-stream and deposit count on the same GPU: under
 * the same calls the two stay bit for bit alike (the reference's way to fork a filter is struct assignment,
 * particleFilter.h:62-77: the struct is a POD).  Unsharded filters only; NULL on error. */
pfx_loc* pfx_loc_clone(pfx_loc* src);
int      pfx_loc_reset(pfx_loc* pf, uint32_t seed, uint32_t stream_id);   /* back to "not initialised" */
int64_t  pfx_loc_n(const pfx_loc* pf);
int      pfx_loc_device(const pfx_loc* pf);
uint32_t pfx_loc_seed(const pfx_loc* pf);           /* Philox key word 0 */
uint32_t pfx_loc_stream_id(const pfx_loc* pf);      /* Philox key word 1 */
int      pfx_loc_set_stream(pfx_loc* pf, void* hip_stream);  /* run on a caller stream (e.g. torch's current one) */
int      pfx_loc_sync(pfx_loc* pf);                           /* wait until queued kernels have finished */
/* -1 (default): the reference's ESS / respawn test picks the branch (pfResample.c:58); 0: never resample;
 * 1: resample on every deposit (the worst-case-traffic run of BASELINE.json config 3) */
int      pfx_loc_set_force_branch(pfx_loc* pf, int force);
/* 1 (default): the resample kernel a deposit calls for (pfResample.c:70-76) is deferred until the next deposit and
 * then runs fused with that deposit's motion + gate + sums (one pass over the offspring instead of two), or until
 * something else reads the particles; 0: every deposit launches its own resample kernel.  Same particles either
 * way.  Ancestor recording always runs undeferred; sharded filters follow the pfx_loc_shard_* protocol below. */
int      pfx_loc_set_deferred(pfx_loc* pf, int enable);

/* the reference's four operations (particleFilter.c:70-104, 178-191, 213-226, 248-291) */
void     pfx_loc_deposit_vio(pfx_loc* pf, double t, float x, float y, float z, float dist);
int      pfx_loc_deposit_range(pfx_loc* pf, float bx, float by, float bz, float range, float stdRange);
int      pfx_loc_deposit_rssi(pfx_loc* pf, float bx, float by, float bz, int rssi);
int      pfx_loc_get_tag_loc(pfx_loc* pf, double* t, float* x, float* y, float* z, float* theta);  /* 1 ok, 0 not initialised, <0 error */

/* ---- state access ---- */
/* Weights enter the weight total, the CDF and the pose sums rounded to multiples of 2^-20: exact sums in
 * any order, sized for the path's own convention of mean weight 1 (pfResample.c:80-82); a weight below 2^-21 counts as 0.
 * pfx_loc_upload therefore rescales a set whose mean weight lies below 2^-12, or whose total reaches 2^33, to mean 1 (factor N / sum w, what
 * the reference's normalisation would apply at the next non-resampling deposit), and pfx_loc_download returns the
 * rescaled weights; other sets are stored as given. */
int      pfx_loc_upload(pfx_loc* pf, const float* w, const float* x, const float* y, const float* z,
                        const float* theta, int64_t n, int initialized);
int      pfx_loc_download(pfx_loc* pf, float* w, float* x, float* y, float* z, float* theta, int64_t n);
int      pfx_loc_initialized(const pfx_loc* pf);
uint32_t pfx_loc_epoch(const pfx_loc* pf);          /* Philox counter word 2: one per deposit */
void     pfx_loc_set_epoch(pfx_loc* pf, uint32_t epoch);
/* device pointers of the current SoA arrays (0=w 1=x 2=y 3=z 4=theta); synchronises the stream.
 * w is stored without the pending normalisation factor: true weight = w * (*wscale). */
int      pfx_loc_device_arrays(pfx_loc* pf, void* d_ptrs[5], float* wscale);

/* ---- the stages of a step, one at a time (what the oracle's ora_loc_* pieces restate) ---- */
int      pfx_loc_apply_vio(pfx_loc* pf, float dt, float dx, float dy, float dz, float ddist);           /* pfMeasurement.c:24-49 */
int      pfx_loc_apply_range(pfx_loc* pf, float bx, float by, float bz, float range, float stdRange);   /* pfMeasurement.c:108-125 */
int      pfx_loc_resample(pfx_loc* pf, float bx, float by, float bz, float range, float stdRange);      /* pfResample.c:30-84 */
int      pfx_loc_spawn_all(pfx_loc* pf, float bx, float by, float bz, float range, float stdRange);     /* pfInit.c:20-25 */
/* out[0]=branch (1 resampled, 0 normalised) out[1]=numSpawn out[2..5]=s ss csum ssum (f32 totals)
 * out[6]=ess out[7]=htheta out[8]=rStart out[9]=rStep out[10]=m (normalisation factor) out[11]=S (f64 total) */
int      pfx_loc_last_stats(pfx_loc* pf, double out[12]);
int      pfx_loc_record_ancestors(pfx_loc* pf, int enable);  /* keep the ancestor index of every offspring */
int      pfx_loc_last_ancestors(pfx_loc* pf, int32_t* anc, int64_t n);

/* systematic-resample index selection on a caller-supplied CDF (host arrays), as pfResample.c:70-72
 * walks it: anc[i] = first j with rStart + rStep*i < cdf[j], clamped to n-1.  _f32 evaluates the
 * reference's float32 expression (no fused multiply-add) and is bit-exact against the reference loop
 * for identical inputs.  These two run a plain binary search over the caller's array in global memory;
 * the filter's own resampling (tile CDFs rebuilt in LDS, integer offsets) is checked index for index
 * through pfx_loc_record_ancestors / pfx_loc_last_ancestors. */
int      pfx_resample_indices_f32(const float* cdf, int64_t n, float rStart, float rStep, int32_t* anc, int device);
int      pfx_resample_indices_f64(const double* cdf, int64_t n, double rStart, double rStep, int32_t* anc, int device);

/* ---- kernel timing (HIP events on the filter's stream) ---- */
int      pfx_loc_profile_begin(pfx_loc* pf, int max_records);
/* which kernel classes are timed (bit k = PFX_K_* class k; default all): a timed launch costs ~3 us of GPU time */
int      pfx_loc_profile_mask(pfx_loc* pf, unsigned mask);
/* time only every stride-th launch of a class (launch 0, stride, 2 * stride ... counted from pfx_loc_profile_begin) */
int      pfx_loc_profile_stride(pfx_loc* pf, int stride);
/* sums of elapsed ms and launch counts per kernel class since pfx_loc_profile_begin; stops recording */
int      pfx_loc_profile_read(pfx_loc* pf, double ms[PFX_K_COUNT], int64_t launches[PFX_K_COUNT]);
/* the same records one by one, in launch order (call before pfx_loc_profile_read, which clears them);
 * returns the number of records written (<= cap), <0 on error */
int64_t  pfx_loc_profile_records(pfx_loc* pf, int32_t* kind, float* ms, int64_t cap);
/* branch taken by each deposit since creation (1 resampled, 0 normalised), oldest first, at most the
 * last 65536; *total = number of decisions so far.  Synchronises the stream. */
int64_t  pfx_loc_branch_log(pfx_loc* pf, uint8_t* out, int64_t cap, int64_t* total);

/* ---- many Loc filters of equal size in one set of launches (no counterpart in the reference) ----
 * The reference runs one filter per tag and one call per measurement (csvlocalize.c:64-88) at N = 10 000
 * (particleFilter.h:18), a size at which a GPU deposit is bound by its launches and by the caller.  A pfx_loc_batch holds
 * num_filters independent particleFilterLoc filters of n particles each on one GPU; every call below does for each filter
 * f with mask[f] != 0 (mask NULL: every filter) exactly what the pfx_loc_* call of the same name does for one filter --
 * same arithmetic, same deferred resampling, decisions taken per filter on the device -- in ONE set of kernel launches
 * (grid dimension = filter).  Filter f draws its noise from Philox key (seeds[f] or seed0, stream_ids[f] or f): after
 * the same calls it holds bit for bit the particles of pfx_loc_create(n, that seed, that stream id, device).  All arrays
 * are host arrays of num_filters entries.  ("batch of independent filters ... also many Loc tags".) */
typedef struct pfx_loc_batch pfx_loc_batch;
pfx_loc_batch* pfx_loc_batch_create(int num_filters, int64_t n, const uint32_t* seeds, uint32_t seed0, const uint32_t* stream_ids, int device);
void     pfx_loc_batch_destroy(pfx_loc_batch* b);
int      pfx_loc_batch_num_filters(const pfx_loc_batch* b);
int64_t  pfx_loc_batch_n(const pfx_loc_batch* b);
uint32_t pfx_loc_batch_seed(const pfx_loc_batch* b, int f);
uint32_t pfx_loc_batch_stream_id(const pfx_loc_batch* b, int f);
uint32_t pfx_loc_batch_epoch(const pfx_loc_batch* b, int f);
int      pfx_loc_batch_initialized(const pfx_loc_batch* b, int f);
int      pfx_loc_batch_sync(pfx_loc_batch* b);
int      pfx_loc_batch_set_force_branch(pfx_loc_batch* b, int force);          /* as pfx_loc_set_force_branch, whole batch */
/* particleFilter.c:70-104 / 178-191 / 213-226 / 248-291 for every selected filter */
int      pfx_loc_batch_deposit_vio(pfx_loc_batch* b, const uint8_t* mask, const double* t, const float* x, const float* y,
                                   const float* z, const float* dist /* NULL: zeros */);
int      pfx_loc_batch_deposit_range(pfx_loc_batch* b, const uint8_t* mask, const float* bx, const float* by, const float* bz,
                                     const float* range, const float* stdRange);
int      pfx_loc_batch_deposit_rssi(pfx_loc_batch* b, const uint8_t* mask, const float* bx, const float* by, const float* bz,
                                    const int32_t* rssi /* unused, as in the reference; may be NULL */);
int      pfx_loc_batch_get_tag_loc(pfx_loc_batch* b, uint8_t* ok, double* t, float* x, float* y, float* z, float* theta);
/* state and diagnostics of one filter of the batch (as pfx_loc_download / pfx_loc_last_stats) */
int      pfx_loc_batch_download(pfx_loc_batch* b, int f, float* w, float* x, float* y, float* z, float* theta, int64_t n);
int      pfx_loc_batch_last_stats(pfx_loc_batch* b, int f, double out[12]);
/* deposits decided / of those resampled since creation: batch totals, and per filter where the arrays are not NULL */
int      pfx_loc_batch_branch_counts(pfx_loc_batch* b, int64_t* decisions, int64_t* resampled, int64_t* per_filter_decisions,
                                     int64_t* per_filter_resampled);

/* diagnostic: the range gate's squared-distance bounds (pfMeasurement.c:118-123: a particle keeps its weight iff its squared
 * distance to the beacon lies in [lo, hi]) for `count` (range, stdRange) pairs, derived on the host (single filters) and on
 * the device (batches); they must agree bit for bit */
int      pfx_gate_bounds_both(const float* range, const float* stdRange, int count, float* lo_host, float* hi_host,
                              float* lo_dev, float* hi_dev, int device);

/* ---- one filter sharded by particle range over several GPUs (no counterpart in the reference) ----
 * The caller owns the collectives (slam3d_amd/sharded.py uses torch.distributed = RCCL); these are the local
 * halves of a deposit.  The shard created with pfx_loc_create(n_local, ...) holds the global particles
 * [self_begin, self_begin + n_local) of a filter of n_global; noise is keyed by global index.  Per deposit:
 *   pfx_loc_shard_begin    VIO commit (returns 1 instead if the filter was uninitialised and has just spawned its
 *                          particles: nothing else follows)
 *   pfx_loc_shard_deposit  the O(N) pass, launched WITHOUT looking at the previous deposit's decision: after a
 *                          "resample" the kernel produces the offspring of the local ancestors that fall into the shard's
 *                          own index range and deposits on them in the same pass; after a "normalise" it is the streaming
 *                          pass
 *   pfx_loc_shard_plan     the previous deposit's decision (it landed in pinned host memory while the kernel above was
 *                          launched): branch and F[g] = first global offspring index produced by shard g
 *                          (F[num_shards] = n_global).  Fails if the staging area is too small.
 *   after a "resample":
 *     pfx_loc_shard_stage  if F says that some of this shard's offspring belong to other shards: they are produced as
 *                          (x, y, z, theta) records in the send half of the staging area, first those below the shard's
 *                          range, then those above it (a no-op under PFX_SHARD_DIRECT=0, where the deposit kernel is
 *                          round 1's source-owned one and has staged them itself)
 *     shipping             the caller sends each neighbour its slice of the staging area into that shard's landing area
 *                          (record k of the landing area = local element k for k < lo, else hi + (k - lo), where
 *                          [lo, hi) are the elements the shard produced itself)
 *     pfx_loc_shard_land(1, lo, hi)  the received records are unpacked into the arrays and deposited on
 *     pfx_loc_shard_finish(1, lo, hi)  local tile prefix from the pieces; the shard's chunk sums land in its row
 *   after a "normalise":   pfx_loc_shard_finish(0, 0, 0)
 *   all-gather             of the shard's row of chunk sums, pfx_loc_shard_local_ptr (pfx_loc_shard_row_doubles doubles: 8 per
 *                          262 144 particles of the largest shard; the same on every rank), into pfx_loc_shard_all_ptr
 *                          (rank-major), by the caller
 *   pfx_loc_shard_decide   the reference's ESS / respawn decision on the GLOBAL totals, identical on every rank;
 *                          returns without waiting.  A "resample" is acted on by the next pfx_loc_shard_deposit.
 * Before anything reads the particles (pose query, download): pfx_loc_shard_plan, and if it says "resample"
 * pfx_loc_shard_produce (the plain resampling), shipping, pfx_loc_shard_land(0, lo, hi). */
int      pfx_loc_shard_config(pfx_loc* pf, int rank, int num_shards, int64_t n_global, int64_t self_begin, int64_t stage_capacity);
int      pfx_loc_shard_begin(pfx_loc* pf, float bx, float by, float bz, float range, float stdRange);
int      pfx_loc_shard_deposit(pfx_loc* pf);
int      pfx_loc_shard_stage(pfx_loc* pf);
void*    pfx_loc_shard_local_ptr(pfx_loc* pf);      /* device address: the shard's row, per chunk s ss csum ssum wx wy wz 0 */
void*    pfx_loc_shard_all_ptr(pfx_loc* pf);        /* device address: 16 rows */
int64_t  pfx_loc_shard_row_doubles(const pfx_loc* pf);  /* doubles per row (after pfx_loc_shard_config) */
int      pfx_loc_shard_decide(pfx_loc* pf);
int      pfx_loc_shard_plan(pfx_loc* pf, int* out_branch, int64_t* out_F);
/* 0: nothing outstanding; 1: a decision has been launched and neither looked at nor acted on; 2: it was "resample"
 * and its offspring have not been produced */
int      pfx_loc_shard_pending(const pfx_loc* pf);
int      pfx_loc_shard_produce(pfx_loc* pf);
int      pfx_loc_shard_land(pfx_loc* pf, int deposit, int64_t lo, int64_t hi);
int      pfx_loc_shard_finish(pfx_loc* pf, int pieces, int64_t lo, int64_t hi);
/* device pointers: send and landing halves of the staging area (stage_capacity records of 4 floats each) and both
 * particle buffers (w,x,y,z,theta); no synchronisation */
int      pfx_loc_shard_ptrs_both(pfx_loc* pf, void** d_stage_send, void** d_stage_land, void* d_buf0[5], void* d_buf1[5]);
int      pfx_loc_shard_cur(const pfx_loc* pf);     /* which of the two buffers is current (after pfx_loc_shard_plan) */
int      pfx_loc_shard_pose_sums(pfx_loc* pf, double out[6]);   /* local sums of w, wx, wy, wz, w cos, w sin */
/* -- the same deposit with the collectives inside the library: ONE call per deposit and rank, RCCL called directly on the
 * shard's stream (ncclAllGather of the rows, ncclSend / ncclRecv of the records), looked up at run time ("librccl.so.1": in
 * a process that has loaded torch, torch's copy).  The shards must follow the even split T_g = floor(g N / G).  Every rank
 * makes the same calls in the same order (they are collectives).  No counterpart in the reference. */
int      pfx_rccl_unique_id(uint8_t out[128]);                          /* ncclGetUniqueId: one rank calls it, all get the bytes */
int      pfx_loc_shard_comm_init(pfx_loc* pf, const uint8_t id[128]);   /* after pfx_loc_shard_config; collective */
int      pfx_loc_shard_comm_active(pfx_loc* pf);
int      pfx_loc_shard_comm_destroy(pfx_loc* pf);
int      pfx_loc_shard_deposit_range(pfx_loc* pf, float bx, float by, float bz, float range, float stdRange);  /* 1: spawned */
int      pfx_loc_shard_settle(pfx_loc* pf);                             /* produce an outstanding resampling without a deposit */
int      pfx_loc_shard_get_tag_loc(pfx_loc* pf, double* t, float* x, float* y, float* z, float* theta);
int64_t  pfx_loc_shard_exchanged(pfx_loc* pf);                          /* offspring this shard has sent to others so far */
/* who sends what after a "resample" decision (host arithmetic only, no GPU): F[g]..F[g+1] = global offspring indices shard g
 * produces; per peer the (offset, count) of the records this shard sends from its staging area and receives into its
 * landing area; [lo, hi) = the local elements it produced itself */
int      pfx_shard_exchange_plan(const int64_t* F, int64_t n_global, int world, int rank, int64_t* send_off, int64_t* send_cnt,
                                 int64_t* recv_off, int64_t* recv_cnt, int64_t* lo, int64_t* hi);
/* particleFilter.c:274-290 on caller-supplied (e.g. all-reduced) pose sums; 1 ok, 0 not initialised */
int      pfx_loc_pose_from_sums(pfx_loc* pf, const double sums[6], double* t, float* x, float* y, float* z, float* theta);

/* ---- one entry point that allocates nothing ----
 * No counterpart in the reference (which never allocates: its scratch is a stack array, pfResample.c:35).  For callers
 * that cannot call hipMalloc -- code under HIP graph capture, tight serving loops, harnesses that forbid allocation:
 * the pose reduction of getTagLoc (particleFilter.c:263-273; the kernels' code behind pfx_loc_shard_pose_sums) over the
 * caller's own DEVICE arrays w, x, y, z, theta of n particles (16-byte aligned; true weight = w * wscale, as
 * pfx_loc_device_arrays hands them out; w * wscale enters sum w rounded to a multiple of 2^-20, so pass a wscale that
 * brings the mean weight to about 1 -- N for a set normalised to sum 1 -- or the total comes out 0), six sums {w, w x, w y, w z, w cos theta, w sin theta} written to the caller's
 * device buffer d_out6 by kernels enqueued on hip_stream (NULL = the default stream); feed them to pfx_loc_pose_from_sums.
 * The per-tile partial sums, for which a filter allocates a buffer of its own, live in a static device array of the
 * library: 4 MiB per GPU, held until the process ends, n <= 2^26 (larger n is refused with an error, nothing runs).
 * Calls no hipMalloc / hipFree / allocating routine; the first call on a device looks the array's address up (do that
 * once outside any graph capture).  All calls on one GPU share the array: calls that overlap there (different streams
 * or threads) race and give wrong sums; successive calls need no clearing (every partial read was written by the same call). */
int      pfx_pose_sums_noalloc(const float* d_w, const float* d_x, const float* d_y, const float* d_z, const float* d_theta,
                               int64_t n, float wscale, double* d_out6, void* hip_stream, int device);

/* ---- particleFilterSlam with runtime sizes, batched (BASELINE.json configs[4]) ----
 * A pfx_slam is a batch of `num_filters` independent SLAM filters on one GPU: each has n_tag tag particles
 * (reference: PF_N_TAG_SLAM = 100, particleFilter.h:19) and up to max_bcns (<= 32) beacons of n_tag x n_bcn
 * particles (PF_N_BCN = 1000, :20); n_tag, n_bcn <= 1024.  Filter f draws its noise from Philox key
 * (seeds[f], or seed0 + f when seeds is NULL); beacon slot c uses sub-stream c + 1, the tag particles 0.
 * The per-filter calls mirror particleFilterSlam_* one to one (particleFilter.c:48-68, 106-176, 193-246,
 * 293-387; `allBcns` = every beacon added to the filter); the *_batch calls do the same for every filter
 * of the batch in one set of kernel launches. */
typedef struct pfx_slam pfx_slam;
pfx_slam* pfx_slam_create(int num_filters, int n_tag, int n_bcn, int max_bcns, const uint32_t* seeds, uint32_t seed0, int device);
void     pfx_slam_destroy(pfx_slam* pf);
int      pfx_slam_num_filters(const pfx_slam* pf);
int      pfx_slam_n_tag(const pfx_slam* pf);
int      pfx_slam_n_bcn(const pfx_slam* pf);
int      pfx_slam_max_bcns(const pfx_slam* pf);
int      pfx_slam_sync(pfx_slam* pf);
int      pfx_slam_reset(pfx_slam* pf, int f, uint32_t seed);                    /* particleFilterSlam_init of filter f */
int      pfx_slam_add_bcn(pfx_slam* pf, int f);                                 /* particleFilterSlam_addBcn; returns the slot */
int      pfx_slam_bcn_initialized(const pfx_slam* pf, int f, int slot);
uint32_t pfx_slam_epoch(const pfx_slam* pf, int f);                             /* Philox counter word 2: one per deposit */
void     pfx_slam_set_epoch(pfx_slam* pf, int f, uint32_t epoch);
uint32_t pfx_slam_seed(const pfx_slam* pf, int f);
int      pfx_slam_deposit_tag_vio(pfx_slam* pf, int f, double t, float x, float y, float z, float dist);
int      pfx_slam_deposit_bcn_vio(pfx_slam* pf, int f, int slot, double t, float x, float y, float z, float dist);
int      pfx_slam_deposit_range(pfx_slam* pf, int f, int slot, float range, float stdRange);
int      pfx_slam_deposit_rssi(pfx_slam* pf, int f, int slot, int rssi);
int      pfx_slam_get_tag_loc(pfx_slam* pf, int f, double* t, float* x, float* y, float* z, float* theta);          /* 1 ok, <0 error */
int      pfx_slam_get_bcn_loc(pfx_slam* pf, int f, int slot, double* t, float* x, float* y, float* z, float* theta); /* 1 ok, 0 beacon not initialised */
int      pfx_slam_last_tag_branch(pfx_slam* pf, int f);     /* 1: the last deposit resampled the tag particles (pfResample.c:110) */
/* whole batch: arrays of num_filters entries; slot[f] < 0 = no measurement for filter f in this deposit */
int      pfx_slam_deposit_tag_vio_batch(pfx_slam* pf, const double* t, const float* x, const float* y, const float* z, const float* dist);
int      pfx_slam_deposit_range_batch(pfx_slam* pf, const int32_t* slot, const float* range, const float* stdRange);
int      pfx_slam_get_tag_loc_batch(pfx_slam* pf, double* t, float* x, float* y, float* z, float* theta);
int      pfx_slam_get_bcn_loc_batch(pfx_slam* pf, const int32_t* slot, uint8_t* ok, double* t, float* x, float* y, float* z, float* theta);
/* measurement aid: rows of the measured beacons that the last deposit only normalised (out[0]) / resampled (out[1]),
 * filters whose tag particles it resampled (out[2]), filters in the batch (out[3]); synchronises the stream */
int      pfx_slam_row_census(pfx_slam* pf, int64_t out[4]);
/* state: tag arrays of n_tag floats; beacon arrays of n_tag * n_bcn floats, row i = pBcn[i][*] (particleFilter.h:86) */
int      pfx_slam_download_tag(pfx_slam* pf, int f, float* w, float* x, float* y, float* z, float* theta);
int      pfx_slam_upload_tag(pfx_slam* pf, int f, const float* w, const float* x, const float* y, const float* z, const float* theta);
int      pfx_slam_download_bcn(pfx_slam* pf, int f, int slot, float* w, float* x, float* y, float* z, float* theta);
int      pfx_slam_upload_bcn(pfx_slam* pf, int f, int slot, const float* w, const float* x, const float* y, const float* z,
                             const float* theta, int initialized);
/* device pointers of the SoA state, for zero-copy views: tag arrays [filter][tagStride], beacon arrays
 * [filter][slot][tag particle][rowStride], pending per-row weight factor [filter][slot][tag particle] (true beacon weight
 * = w * factor of its row); dims = numFilters, maxBcns, nTag, nBcn, tagStride, rowStride.  Synchronises the stream; the
 * pointers stay valid for the life of the batch (SLAM rows are resampled in place, there is no ping-pong). */
int      pfx_slam_device_arrays(pfx_slam* pf, void* d_tag[5], void* d_bcn[5], void** d_row_scale, int32_t dims[6]);
/* binding between the reference-layout structs and device filters */
pfx_slam* pfx_slam_of(const particleFilterSlam_t* pf);                       /* NULL if the struct is not bound */
int      pfx_slam_slot_of(const particleFilterSlam_t* pf, const bcn_t* bcn); /* -1 until the beacon's first deposit */
/* fill pTag[] and (bcn != NULL) pBcn[][] from HBM.  Until their next state-changing call the structs are then complete
 * checkpoints: their bytes may be stored, copied or moved and are continued from where they are found (the reference's
 * structs are PODs); a copy taken without this call is refused at its first use with a message. */
int      pfx_slam_sync_host(particleFilterSlam_t* pf, bcn_t* bcn);
void     pfx_slam_release(particleFilterSlam_t* pf);                         /* free the device side */

/* ---- binding between the reference-layout structs and device filters ---- */
/* A particleFilterLoc_t stays a POD that may be assigned, memcpy-ed or stored (reference: the struct is the filter):
 *   - a copy taken from a struct that has not been used since is given a device-side clone at its first call;
 *   - a copy that is to be continued LATER (a checkpoint, also across processes) must be taken right after
 *     pfx_loc_sync_host(): it is then continued from its own pTag[] -- in a filter of its own, or, when the bytes are
 *     copied back over the live struct, in place;
 *   - any other copy is refused at its first call with a message (pfx_last_error). */
pfx_loc* pfx_loc_of(const particleFilterLoc_t* pf);          /* NULL if the struct (at this address) is not bound */
int      pfx_loc_sync_host(particleFilterLoc_t* pf);         /* fill pf->pTag[] (AoS) from HBM: the struct is a full checkpoint */
void     pfx_loc_release(particleFilterLoc_t* pf);           /* free the device side (the reference API has no destroy) */

#ifdef __cplusplus
}
#endif
#endif

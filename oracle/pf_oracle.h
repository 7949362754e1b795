/*
 * pf_oracle.h -- TEST INFRASTRUCTURE ONLY.
 *
 * CPU restatement of the slam3d particle-filter hot path (reference:
 * particlefilter/src/{particleFilter,pfMeasurement,pfResample,pfInit,pfRandom}.c).
 * It exists to check the HIP path; nothing under slam3d_amd/ may link, load
 * or call it.  Only tests/, __graft_entry__.smoke() and bench.py's
 * cpu_baseline leg use it, and only as the checker / the reported baseline.
 *
 * Parity status: PINNED.  rand_r(glibc)+f32 mode reproduces the reference's
 * test/data/test1_ParticleFilterLoc_expected_out_linux-gcc.csv byte for byte,
 * rand_r(MUSL)+f32 mode reproduces the windows-gcc/windows-cl files to 1e-6,
 * and every function (Loc and SLAM) is checked against goldens generated
 * from the compiled reference (oracle/_ref, tests/golden/make_golden.py).
 *
 * Differences from the reference, all deliberate:
 *   - runtime particle counts and SoA storage (reference: compile-time AoS);
 *   - the RNG is an object with three sources: glibc rand_r, MUSL rand_r
 *     (both serial streams, as the reference) and counter-based Philox4x32-10
 *     (what the HIP kernels use; keyed by epoch/purpose/particle index);
 *   - an optional "f64 accumulation" rule (ORA_ACC_F64) for the weight sums
 *     and the CDF: this is the rule the HIP kernels implement, because the
 *     reference's serial f32 running sum is numerically void for N >= 2^24.
 *     Under it a weight enters every weight total and every CDF rounded to the
 *     nearest multiple of 2^-20 (ora_cdf_weight) and the sums are double: such
 *     sums are exact below 2^33, hence independent of the order of the
 *     additions, which is what lets the GPU's tiled / sharded / deferred sums
 *     and this file's serial loops agree on every CDF value bit for bit.
 *     ORA_ACC_F32 is the faithful reference arithmetic.
 */
#ifndef PF_ORACLE_H
#define PF_ORACLE_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

enum { ORA_RNG_GLIBC = 0, ORA_RNG_MUSL = 1, ORA_RNG_PHILOX = 2 };
enum { ORA_ACC_F32 = 0, ORA_ACC_F64 = 1 };
/* Philox sub-stream selectors; must equal the PFX_P_* values in slam3d_amd/csrc/pf_common.h */
enum { ORA_P_VIO = 1, ORA_P_RSTART = 2, ORA_P_JITTER = 3, ORA_P_SPAWN = 4 };

typedef struct ora_rng ora_rng;
typedef struct ora_loc ora_loc;
typedef struct ora_slam ora_slam;
typedef struct ora_bcn ora_bcn;

/* ---- RNG (pfRandom.c) ---- */
ora_rng* ora_rng_new(int mode, unsigned seed);
void     ora_rng_free(ora_rng* r);
unsigned ora_rng_state(const ora_rng* r);                 /* PF_SEED of the serial modes */
void     ora_rng_set_state(ora_rng* r, unsigned s);
void     ora_rng_begin(ora_rng* r, uint32_t stream, uint32_t epoch, uint32_t purpose, uint64_t index);
int      ora_rng_next31(ora_rng* r);                      /* rand_r_func(): [0, 2^31-1] */
float    ora_rng_uniform(ora_rng* r);                     /* pfRandom_uniform  */
void     ora_rng_normal2(ora_rng* r, float* a, float* b); /* pfRandom_normal2  */
void     ora_rng_sphere(ora_rng* r, float* x, float* y, float* z, float range, float stdRange); /* pfRandom_sphere */
void     ora_philox4x32_10(const uint32_t ctr[4], const uint32_t key[2], uint32_t out[4]);

/* ---- particleFilterLoc (runtime N) ---- */
ora_loc* ora_loc_new(int n, ora_rng* rng, int acc_mode);
void     ora_loc_free(ora_loc* pf);
void     ora_loc_init(ora_loc* pf);
int      ora_loc_n(const ora_loc* pf);
float*   ora_loc_array(ora_loc* pf, int which);           /* 0=w 1=x 2=y 3=z 4=theta, n floats each */
void     ora_loc_set_initialized(ora_loc* pf, int v);
int      ora_loc_initialized(const ora_loc* pf);
void     ora_loc_set_epoch(ora_loc* pf, uint32_t e);
uint32_t ora_loc_epoch(const ora_loc* pf);
void     ora_loc_deposit_vio(ora_loc* pf, double t, float x, float y, float z, float dist);
void     ora_loc_deposit_range(ora_loc* pf, float bx, float by, float bz, float range, float stdRange);
void     ora_loc_deposit_rssi(ora_loc* pf, float bx, float by, float bz, int rssi);
int      ora_loc_get_tag_loc(const ora_loc* pf, double* t, float* x, float* y, float* z, float* theta);
/* the pieces, callable on their own */
void     ora_loc_apply_vio(ora_loc* pf, float dt, float dx, float dy, float dz, float ddist);
void     ora_loc_apply_range(ora_loc* pf, float bx, float by, float bz, float range, float stdRange);
void     ora_loc_resample(ora_loc* pf, float bx, float by, float bz, float range, float stdRange);
void     ora_loc_spawn_all(ora_loc* pf, float bx, float by, float bz, float range, float stdRange);
/* what the last ora_loc_resample did: out[0]=branch(1 resample,0 normalise) out[1]=numSpawn
 * out[2..]= s ss csum ssum ess htheta rStart rStep m ; ancestors of the last resample */
void       ora_loc_last_stats(const ora_loc* pf, double out[12]);
const int* ora_loc_last_ancestors(const ora_loc* pf);
const double* ora_loc_last_cdf(const ora_loc* pf);

/* helpers for the sharded-filter test double (Philox mode) */
void     ora_loc_set_index_offset(ora_loc* pf, uint64_t idx0);
void     ora_loc_commit_vio(ora_loc* pf, float d[5]);
int      ora_loc_pose_from_sums(const ora_loc* pf, const double p[6], double* t, float* x, float* y, float* z, float* theta);
void     ora_offspring_block(ora_rng* rng, uint32_t stream, uint32_t epoch, int64_t i_begin, int count, const int* anc,
                             const float* sx, const float* sy, const float* sz, const float* sth, float hTheta, int numSpawn,
                             float cx, float cy, float cz, float range, float stdRange,
                             float* ox, float* oy, float* oz, float* oth);
void     ora_decide(ora_rng* rng, uint32_t stream, uint32_t epoch, double S, double SS, double C, double SN, int64_t n,
                    float range, int force, double out[6]);

/* the weight as it enters a weight total / the CDF under ORA_ACC_F64: nearest multiple of 2^-20, ties to even */
float ora_cdf_weight(float w);

/* systematic-resample index selection exactly as pfResample.c:70-72 walks it, on a caller CDF */
long ora_clamp_events(void);   /* offspring clamped to the last ancestor so far (reference: out-of-bounds read) */
void ora_resample_indices_f32(const float* cdf, int n, float rStart, float rStep, int* anc);
void ora_resample_indices_f64(const double* cdf, int n, double rStart, double rStep, int* anc);

/* ---- particleFilterSlam (runtime nTag, nBcn) ---- */
ora_slam* ora_slam_new(int nTag, int nBcn, ora_rng* rng, int acc_mode);
void      ora_slam_free(ora_slam* pf);
void      ora_slam_init(ora_slam* pf);
float*    ora_slam_array(ora_slam* pf, int which);         /* tag particles, nTag floats each */
int       ora_slam_last_tag_branch(const ora_slam* pf);    /* did the last deposit resample the tags */
ora_bcn*  ora_bcn_new(ora_slam* pf, uint32_t stream);
void      ora_bcn_free(ora_bcn* b);
void      ora_bcn_add(ora_bcn* b);                         /* particleFilterSlam_addBcn */
float*    ora_bcn_array(ora_bcn* b, int which);            /* nTag*nBcn floats each, row = tag particle */
int       ora_bcn_initialized(const ora_bcn* b);
void      ora_slam_set_epoch(ora_slam* pf, uint32_t e);
void      ora_slam_deposit_tag_vio(ora_slam* pf, double t, float x, float y, float z, float dist);
void      ora_slam_deposit_bcn_vio(ora_bcn* b, double t, float x, float y, float z, float dist);
void      ora_slam_deposit_range(ora_slam* pf, ora_bcn* b, float range, float stdRange, ora_bcn** all, int numBcns);
void      ora_slam_deposit_rssi(ora_slam* pf, ora_bcn* b, int rssi, ora_bcn** all, int numBcns);
int       ora_slam_get_tag_loc(const ora_slam* pf, double* t, float* x, float* y, float* z, float* theta);
int       ora_slam_get_bcn_loc(const ora_slam* pf, const ora_bcn* b, double* t, float* x, float* y, float* z, float* theta);

#ifdef __cplusplus
}
#endif
#endif

/*
 * particleFilter.h -- drop-in C interface of the MI355X particle-filter library.
 *
 * Binary- and source-compatible with the reference's public header
 * (reference: particlefilter/include/particleFilter.h:13-120): the same 14 entry
 * points and the same caller-owned POD structs, byte for byte
 * (sizeof: particleFilterLoc_t 400056, particleFilterSlam_t 4056, bcn_t 2020056;
 * offsets: initialized @ end of the particle arrays, then the VIO accumulator).
 * A program written against the reference header recompiles, or simply relinks,
 * against libparticlefilter.so unchanged.
 *
 * What differs behind the interface: the particles of a filter live in HBM on the
 * GPU as SoA float arrays, not in pTag[]/pBcn[]; the host struct keeps the scalar
 * fields (initialized, first*, last* -- authoritative, readable by the caller as
 * before) and the library keeps a small binding record in the first bytes of the
 * struct's scratch array (pTagBuf / pBcnBuf; the reference uses those only as
 * resampling scratch).  pTag[]/pBcn[] are filled on request by pfx_*_sync_host()
 * (include/pfx.h).  Noise comes from counter-based Philox keyed by the seed given
 * to particleFilterSeed_set(), not from the rand_r stream.
 */
#ifndef _PARTICLEFILTER_H
#define _PARTICLEFILTER_H

#include <stdint.h>

/* compile-time particle counts of the caller-visible arrays (reference :18-20) */
#define PF_N_TAG_LOC    (10000)
#define PF_N_TAG_SLAM   (100)
#define PF_N_BCN        (1000)

#ifdef __cplusplus
extern "C" {
#endif

/* one hypothesis: weight, position, heading (reference :28-46; two identical types) */
typedef struct { float w; float x; float y; float z; float theta; } tagParticle_t;
typedef struct { float w; float x; float y; float z; float theta; } bcnParticle_t;

/* VIO accumulator carried by every filter and beacon (reference :52-62, :70-80, :88-98) */
#define PF_VIO_STATE_FIELDS                                              \
    uint8_t initialized;                                                 \
    double firstT; float firstX; float firstY; float firstZ; float firstDist; \
    double lastT;  float lastX;  float lastY;  float lastZ;  float lastDist;

typedef struct {                            /* reference :48-64 */
    tagParticle_t pTag[PF_N_TAG_LOC];
    tagParticle_t pTagBuf[PF_N_TAG_LOC];
    PF_VIO_STATE_FIELDS
} particleFilterLoc_t;

typedef struct {                            /* reference :66-82 */
    tagParticle_t pTag[PF_N_TAG_SLAM];
    tagParticle_t pTagBuf[PF_N_TAG_SLAM];
    PF_VIO_STATE_FIELDS
} particleFilterSlam_t;

typedef struct {                            /* reference :84-100 */
    bcnParticle_t pBcn[PF_N_TAG_SLAM][PF_N_BCN];
    bcnParticle_t pBcnBuf[PF_N_BCN];
    PF_VIO_STATE_FIELDS
} bcn_t;

/* reference :26 / particleFilter.c:26-30 -- seed for every filter initialised afterwards */
void particleFilterSeed_set(unsigned int seed);

/* reference :102-104 / particleFilter.c:32-68 */
void particleFilterLoc_init(particleFilterLoc_t* pf);
void particleFilterSlam_init(particleFilterSlam_t* pf);
void particleFilterSlam_addBcn(bcn_t* bcn);

/* reference :105-107 / particleFilter.c:70-176 -- O(1) host bookkeeping, no GPU work */
void particleFilterLoc_depositVio(particleFilterLoc_t* pf, double t, float x, float y, float z, float dist);
void particleFilterSlam_depositTagVio(particleFilterSlam_t* pf, double t, float x, float y, float z, float dist);
void particleFilterSlam_depositBcnVio(bcn_t* bcn, double t, float x, float y, float z, float dist);

/* reference :108-111 / particleFilter.c:178-246 -- the hot path: VIO commit, range gate, resample */
void particleFilterLoc_depositRange(particleFilterLoc_t* pf, float bx, float by, float bz, float range, float stdRange);
void particleFilterSlam_depositRange(particleFilterSlam_t* pf, bcn_t* bcn, float range, float stdRange, bcn_t** allBcns, int numBcns);
void particleFilterLoc_depositRssi(particleFilterLoc_t* pf, float bx, float by, float bz, int rssi);
void particleFilterSlam_depositRssi(particleFilterSlam_t* pf, bcn_t* bcn, int rssi, bcn_t** allBcns, int numBcns);

/* reference :112-114 / particleFilter.c:248-387 -- pose estimates; 0 = not initialised, outputs untouched */
uint8_t particleFilterLoc_getTagLoc(const particleFilterLoc_t* pf, double* t, float* x, float* y, float* z, float* theta);
uint8_t particleFilterSlam_getTagLoc(const particleFilterSlam_t* pf, double* t, float* x, float* y, float* z, float* theta);
uint8_t particleFilterSlam_getBcnLoc(const particleFilterSlam_t* pf, const bcn_t* bcn, double* t, float* x, float* y, float* z, float* theta);

/* reference particlefilter/include/pfRandom.h:20-21 -- exported globals some callers link against */
extern unsigned int PF_SEED;
extern int PF_SEED_SET;

#ifdef __cplusplus
}
#endif

#endif

/*
 * pfStages.h -- the Loc-level stage functions of the reference's internal headers, exported for link compatibility.
 *
 * The reference installs pfInit.h / pfMeasurement.h / pfResample.h / pfRandom.h beside particleFilter.h and leaves
 * their functions visible in its shared object (nothing there is static).  No caller in the reference uses them, but
 * a program that does keeps linking -- and keeps its meaning -- for the functions below.  Each is a forward onto the GPU
 * stage of the filter bound to the struct (pfx_loc_spawn_all / apply_vio / apply_range / resample) and, like the
 * reference's, works on the struct: on entry pf->pTag[] is uploaded if the caller has changed it since the library last
 * wrote it (a hash of the array is kept beside the binding), on exit pf->pTag[] holds the stage's result and
 * pf->initialized is whatever the reference leaves there (pfInit_initTagLoc does not set it, pfInit.c:20-25).  Noise
 * comes from the filter's Philox sub-streams with the filter's current deposit count as epoch.  Same error behaviour
 * as the particleFilterLoc_* entry points.  A stage call costs two copies of the particle array: these are
 * compatibility entry points, not the hot path (tests/test_gpu_stage_functions.py).
 *
 * Not exported, and why: the single-particle spawners (pfInit_spawn*Particle*, pfInit.h:25-29)
 * and pfRandom_uniform / normal2 / sphere (pfRandom.h:24-26) are host-side helpers on one AoS particle drawing from the
 * one global rand_r stream -- there is no GPU work in them and no CPU path in this library; the SLAM stage functions
 * (pfInit_initTagSlam / initBcnSlam, pfMeasurement_apply{TagVio,BcnVio,Range}Slam, pfResample_resampleSlam) are fused
 * into one pass per deposit on the GPU (k_slam_gate_rows) and have no separable intermediate state to stop at.
 */
#ifndef PF_STAGES_H
#define PF_STAGES_H

#include "particleFilter.h"

#ifdef __cplusplus
extern "C" {
#endif

void pfRandom_init(void);       /* pfRandom.h:23, pfRandom.c:56-60: seed from the clock unless particleFilterSeed_set was called */
void pfInit_initTagLoc(particleFilterLoc_t* pf, float bx, float by, float bz, float range, float stdRange);        /* pfInit.h:22, pfInit.c:20-25 */
void pfMeasurement_applyVioLoc(particleFilterLoc_t* pf, float dt, float dx, float dy, float dz, float ddist);      /* pfMeasurement.h:22, pfMeasurement.c:24-49 */
void pfMeasurement_applyRangeLoc(particleFilterLoc_t* pf, float bx, float by, float bz, float range, float stdRange);  /* pfMeasurement.h:25, pfMeasurement.c:108-125 */
void pfResample_resampleLoc(particleFilterLoc_t* pf, float bx, float by, float bz, float range, float stdRange);   /* pfResample.h:22, pfResample.c:30-84 */

#ifdef __cplusplus
}
#endif
#endif

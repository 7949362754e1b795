/*
 * pf_oracle.c -- TEST INFRASTRUCTURE ONLY (see pf_oracle.h for the rules).
 *
 * CPU restatement of the slam3d particle-filter hot path.  Every function
 * names the reference lines it follows (paths relative to the reference
 * root).  Arithmetic is float32 in the reference's operation order; build
 * with -ffp-contract=off so no multiply-add is fused (the reference is built
 * for baseline x86-64, which has no FMA).
 *
 * Parity: PINNED -- see tests/test_oracle_golden.py (reference expected_out
 * CSVs) and tests/test_oracle_vs_ref_golden.py (goldens from oracle/_ref).
 */
#define _USE_MATH_DEFINES
#include <math.h>
#include <stdint.h>
#include <stdlib.h>
#include <string.h>

#include "pf_oracle.h"

#define PI_F        ((float)M_PI)
#define TWO_PI_F    (2 * (float)M_PI)           /* pfMeasurement.c:47, pfInit.c:65,78 */

/* pfMeasurement.c:20-22 */
#define VIO_STD_XYZ     (1e-3f)
#define VIO_STD_THETA   (1e-6f)
/* pfResample.c:22-26 */
#define RESAMPLE_THRESH     (0.5f)
#define RADIUS_SPAWN_THRESH (4.0f)
#define WEIGHT_SPAWN_THRESH (0.4f)
#define PCT_SPAWN           (0.05f)
#define HXYZ                (0.1f)

/* ------------------------------------------------------------------ RNG */

struct ora_rng {
    int mode;
    unsigned state;                 /* PF_SEED (pfRandom.c:51) for the serial modes */
    uint32_t key[2];                /* Philox key: {seed, stream} */
    uint32_t epoch, purpose, block; /* Philox counter pieces */
    uint64_t index;
    uint32_t words[4];
    int have;
};

/* Philox4x32-10, Salmon et al., "Parallel random numbers: as easy as 1, 2, 3" (SC'11).
 * Not in the reference: it is the counter-based generator the HIP kernels use
 * in place of the serial rand_r stream (BASELINE.json north_star). */
void ora_philox4x32_10(const uint32_t ctr[4], const uint32_t key[2], uint32_t out[4])
{
    uint32_t c0 = ctr[0], c1 = ctr[1], c2 = ctr[2], c3 = ctr[3];
    uint32_t k0 = key[0], k1 = key[1];
    for (int r = 0; r < 10; ++r) {
        uint64_t p0 = (uint64_t)0xD2511F53u * c0;
        uint64_t p1 = (uint64_t)0xCD9E8D57u * c2;
        uint32_t n0 = (uint32_t)(p1 >> 32) ^ c1 ^ k0;
        uint32_t n1 = (uint32_t)p1;
        uint32_t n2 = (uint32_t)(p0 >> 32) ^ c3 ^ k1;
        uint32_t n3 = (uint32_t)p0;
        c0 = n0; c1 = n1; c2 = n2; c3 = n3;
        k0 += 0x9E3779B9u; k1 += 0xBB67AE85u;
    }
    out[0] = c0; out[1] = c1; out[2] = c2; out[3] = c3;
}

/* glibc stdlib/rand_r.c (the reference's default on Linux, pfRandom.c:47-48; the
 * source is not in the reference tree -- published glibc 2.35 algorithm): three
 * LCG steps yielding 11 + 10 + 10 bits. */
static int randr_glibc(unsigned* seed)
{
    unsigned next = *seed;
    unsigned result;
    next = next * 1103515245u + 12345u;
    result = (next >> 16) & 0x7ffu;
    next = next * 1103515245u + 12345u;
    result = (result << 10) ^ ((next >> 16) & 0x3ffu);
    next = next * 1103515245u + 12345u;
    result = (result << 10) ^ ((next >> 16) & 0x3ffu);
    *seed = next;
    return (int)result;
}

/* MUSL rand_r as vendored by the reference (pfRandom.c:30-42): one LCG step, MT-style temper, halve. */
static int randr_musl(unsigned* seed)
{
    unsigned v = *seed = *seed * 1103515245u + 12345u;
    v ^= v >> 11;
    v ^= (v << 7) & 0x9D2C5680u;
    v ^= (v << 15) & 0xEFC60000u;
    v ^= v >> 18;
    return (int)(v >> 1);
}

ora_rng* ora_rng_new(int mode, unsigned seed)
{
    ora_rng* r = (ora_rng*)calloc(1, sizeof(*r));
    r->mode = mode;
    r->state = seed;                /* particleFilterSeed_set, particleFilter.c:26-30 */
    r->key[0] = seed;
    return r;
}
void ora_rng_free(ora_rng* r) { free(r); }
unsigned ora_rng_state(const ora_rng* r) { return r->state; }
void ora_rng_set_state(ora_rng* r, unsigned s) { r->state = s; r->key[0] = s; }

/* Select the Philox sub-stream for (stream, epoch, purpose, particle index).  A no-op for the
 * serial modes, where -- as in the reference -- draws simply come off the one global stream. */
void ora_rng_begin(ora_rng* r, uint32_t stream, uint32_t epoch, uint32_t purpose, uint64_t index)
{
    if (r->mode != ORA_RNG_PHILOX) return;
    r->key[1] = stream; r->epoch = epoch; r->purpose = purpose; r->index = index;
    r->block = 0; r->have = 0;
}

int ora_rng_next31(ora_rng* r)
{
    if (r->mode == ORA_RNG_GLIBC) return randr_glibc(&r->state);
    if (r->mode == ORA_RNG_MUSL)  return randr_musl(&r->state);
    if (!r->have) {
        uint32_t ctr[4] = { (uint32_t)r->index, (uint32_t)(r->index >> 32), r->epoch, r->purpose + 16u * r->block };
        ora_philox4x32_10(ctr, r->key, r->words);
        r->block++; r->have = 4;
    }
    uint32_t w = r->words[4 - r->have];
    r->have--;
    return (int)(w >> 1);
}

/* pfRandom.c:62-65 -- (float)r / RAND_MAX; both round to 2^31, so the range is [0,1] inclusive */
float ora_rng_uniform(ora_rng* r)
{
    return (float)ora_rng_next31(r) / 2147483647;
}

/* pfRandom.c:100-103 -- (r+1)/2^31 in (0,1].  r+1 is formed in 64 bits: the reference's int
 * overflow at r == INT_MAX (a 2^-31 event, undefined behaviour there) is defined here as 1.0. */
static float uniform_nonzero(ora_rng* r)
{
    return (float)((int64_t)ora_rng_next31(r) + 1) / ((float)2147483647 + 1);
}

/* pfRandom.c:67-74 -- Box-Muller pair; u1 is drawn before u2 */
void ora_rng_normal2(ora_rng* r, float* a, float* b)
{
    float f = sqrtf(-2 * logf(uniform_nonzero(r)));
    float g = uniform_nonzero(r) * 2 * PI_F;
    *a = f * cosf(g);
    *b = f * sinf(g);
}

/* pfRandom.c:76-98 -- point on a spherical shell, radius range +- 3 sigma (up to 10 tries for >= 0) */
void ora_rng_sphere(ora_rng* r, float* x, float* y, float* z, float range, float stdRange)
{
    float rad = 0.0f;
    for (int k = 0; k < 10; ++k) {
        float t = range + 3 * stdRange * (ora_rng_uniform(r) * 2 - 1);
        if (t < 0.0f) continue;
        rad = t;
        break;
    }
    float elev = asinf(ora_rng_uniform(r) * 2 - 1);
    float azim = ora_rng_uniform(r) * 2 * PI_F;
    float c = rad * cosf(elev);
    *x = c * cosf(azim);
    *y = c * sinf(azim);
    *z = rad * sinf(elev);
}

/* --------------------------------------------------- shared particle math */

typedef struct { float *w, *x, *y, *z, *th; } soa_t;

static void soa_alloc(soa_t* a, size_t n)
{
    a->w = (float*)calloc(n, sizeof(float)); a->x = (float*)calloc(n, sizeof(float));
    a->y = (float*)calloc(n, sizeof(float)); a->z = (float*)calloc(n, sizeof(float));
    a->th = (float*)calloc(n, sizeof(float));
}
static void soa_free(soa_t* a) { free(a->w); free(a->x); free(a->y); free(a->z); free(a->th); }
static float* soa_pick(soa_t* a, int which)
{
    switch (which) { case 0: return a->w; case 1: return a->x; case 2: return a->y; case 3: return a->z; default: return a->th; }
}

/* the VIO accumulator every filter/beacon carries (particleFilter.h:52-62) */
typedef struct {
    uint8_t initialized;
    double firstT; float firstX, firstY, firstZ, firstDist;
    double lastT;  float lastX, lastY, lastZ, lastDist;
} vio_t;

static void vio_zero(vio_t* v)
{   /* particleFilter.c:34-43 */
    v->firstT = 0.0; v->firstX = v->firstY = v->firstZ = v->firstDist = 0.0f;
    v->lastT = 0.0;  v->lastX = v->lastY = v->lastZ = v->lastDist = 0.0f;
}

/* particleFilter.c:70-104 (and its two SLAM twins :106-176) */
static void vio_deposit(vio_t* v, double t, float x, float y, float z, float dist)
{
    if (v->firstT == 0.0) {
        v->firstT = v->lastT = t;
        v->firstX = v->lastX = x; v->firstY = v->lastY = y; v->firstZ = v->lastZ = z;
        v->firstDist = v->lastDist = dist;
        return;
    }
    if (dist > v->lastDist) {
        v->lastDist = dist;
    } else {
        float dx = x - v->lastX, dy = y - v->lastY, dz = z - v->lastZ;
        v->lastDist += sqrtf(dx * dx + dy * dy + dz * dz);
    }
    v->lastT = t; v->lastX = x; v->lastY = y; v->lastZ = z;
}

/* particleFilter.c:389-404: delta since the last commit, then first <- last */
static void vio_commit(vio_t* v, float d[5])
{
    d[0] = (float)(v->lastT - v->firstT);
    d[1] = v->lastX - v->firstX; d[2] = v->lastY - v->firstY; d[3] = v->lastZ - v->firstZ;
    d[4] = v->lastDist - v->firstDist;
    v->firstT = v->lastT; v->firstX = v->lastX; v->firstY = v->lastY; v->firstZ = v->lastZ;
    v->firstDist = v->lastDist;
}

/* pfMeasurement.c:24-49 (Loc), :51-76 (SLAM tag), :78-106 (SLAM beacon) -- one particle */
static void motion_one(soa_t* p, size_t i, ora_rng* rng, float dx, float dy, float dz, float stdXyz, float stdTheta)
{
    float th = p->th[i];
    float c = cosf(th), s = sinf(th);
    float pDx = dx * c - dy * s;
    float pDy = dx * s + dy * c;
    float rx, ry, rz, rth;
    ora_rng_normal2(rng, &rx, &ry);
    ora_rng_normal2(rng, &rz, &rth);
    p->x[i] += pDx + stdXyz * rx;
    p->y[i] += pDy + stdXyz * ry;
    p->z[i] += dz + stdXyz * rz;
    p->th[i] = fmodf(th + stdTheta * rth, TWO_PI_F);
}

static void motion_all_from(soa_t* p, size_t n, ora_rng* rng, uint32_t stream, uint32_t epoch, const float d[5], uint64_t idx0)
{
    float stdXyz = sqrtf(d[4]) * VIO_STD_XYZ;
    float stdTheta = sqrtf(d[0]) * VIO_STD_THETA;
    for (size_t i = 0; i < n; ++i) {
        ora_rng_begin(rng, stream, epoch, ORA_P_VIO, idx0 + i);
        motion_one(p, i, rng, d[1], d[2], d[3], stdXyz, stdTheta);
    }
}

static void motion_all(soa_t* p, size_t n, ora_rng* rng, uint32_t stream, uint32_t epoch, const float d[5])
{
    float stdXyz = sqrtf(d[4]) * VIO_STD_XYZ;
    float stdTheta = sqrtf(d[0]) * VIO_STD_THETA;
    for (size_t i = 0; i < n; ++i) {
        ora_rng_begin(rng, stream, epoch, ORA_P_VIO, i);
        motion_one(p, i, rng, d[1], d[2], d[3], stdXyz, stdTheta);
    }
}

/* pfInit.c:56-66 / :81-91 -- particle on the range shell around (cx,cy,cz), random heading */
static void spawn_from_range(soa_t* p, size_t i, ora_rng* rng, float cx, float cy, float cz, float range, float stdRange)
{
    float dx, dy, dz;
    ora_rng_sphere(rng, &dx, &dy, &dz, range, stdRange);
    p->w[i] = 1.0f;
    p->x[i] = cx + dx; p->y[i] = cy + dy; p->z[i] = cz + dz;
    p->th[i] = ora_rng_uniform(rng) * 2 * PI_F;
}

/* pfInit.c:68-79 / :93-104 -- jittered copy of an ancestor */
static void spawn_from_other(soa_t* dst, size_t i, const soa_t* src, size_t j, ora_rng* rng, float hXyz, float hTheta)
{
    float dx, dy, dz, dth;
    ora_rng_normal2(rng, &dx, &dy);
    ora_rng_normal2(rng, &dz, &dth);
    dst->w[i] = 1.0f;
    dst->x[i] = src->x[j] + dx * hXyz;
    dst->y[i] = src->y[j] + dy * hXyz;
    dst->z[i] = src->z[j] + dz * hXyz;
    dst->th[i] = fmodf(src->th[j] + dth * hTheta, TWO_PI_F);
}

typedef struct { float s, ss, csum, ssum; double S; } wsums_t;

/* ORA_ACC_F64 (the rule of the HIP path; no counterpart in the reference): a weight enters every
 * weight total and every CDF rounded to the nearest multiple of 2^-20 (ties to even), and those are added in double.
 * Multiples of 2^-20 totalling less than 2^33 add up exactly in double, so the order of the additions -- serial
 * here, by tiles, chunks and shards on the GPU -- cannot change a single bit of the CDF. */
float ora_cdf_weight(float w)
{
    return rintf(w * 1048576.0f) * 9.5367431640625e-07f;
}

/* pfResample.c:37-50 (and :93-106, :151-164): weight sums + inclusive CDF.  ORA_ACC_F32 keeps the
 * reference's serial float running sums; ORA_ACC_F64 forms the same float products but adds them in
 * double (the rule of the HIP path), rounding the four totals to float afterwards. */
static void weight_pass(const float* w, const float* th, int n, int acc, double* cdf, wsums_t* o)
{
    if (acc == ORA_ACC_F32) {
        float s = 0.0f, ss = 0.0f, cs = 0.0f, sn = 0.0f;
        for (int i = 0; i < n; ++i) {
            float wi = w[i];
            s += wi;
            ss += wi * wi;
            cs += wi * cosf(th[i]);
            sn += wi * sinf(th[i]);
            cdf[i] = s;
        }
        o->s = s; o->ss = ss; o->csum = cs; o->ssum = sn; o->S = s;
    } else {
        double s = 0.0, ss = 0.0, cs = 0.0, sn = 0.0;
        for (int i = 0; i < n; ++i) {
            float wi = w[i];
            float p2 = wi * wi, pc = wi * cosf(th[i]), ps = wi * sinf(th[i]);
            s += ora_cdf_weight(wi); ss += p2; cs += pc; sn += ps;
            cdf[i] = s;
        }
        o->s = (float)s; o->ss = (float)ss; o->csum = (float)cs; o->ssum = (float)sn; o->S = s;
    }
}

/* pfResample.c:60-65 -- KDE bandwidth of the heading from the mean resultant length */
static float theta_bandwidth(float csum, float ssum, float s, float ess)
{
    csum /= s;
    ssum /= s;
    float h = csum * csum + ssum * ssum;
    h = h > 1e-10f ? h : 1e-10f;
    h = h < 1 - 1e-10f ? h : 1 - 1e-10f;
    return sqrtf(-logf(h) / ess);
}

/* pfResample.c:70-72: ancestor j keeps emitting offspring i while rStart + rStep*i < cdf[j].
 * The reference's outer loop has no bound on j; here j stops at n-1 (all remaining offspring
 * go to the last ancestor), which is what the reference does whenever it stays in bounds. */
void ora_resample_indices_f32(const float* cdf, int n, float rStart, float rStep, int* anc)
{
    int j = 0;
    for (int i = 0; i < n; ++i) {
        float u = rStart + rStep * i;
        while (j < n - 1 && !(u < cdf[j])) ++j;
        anc[i] = j;
    }
}
void ora_resample_indices_f64(const double* cdf, int n, double rStart, double rStep, int* anc)
{
    int j = 0;
    for (int i = 0; i < n; ++i) {
        double u = rStart + rStep * i;
        while (j < n - 1 && !(u < cdf[j])) ++j;
        anc[i] = j;
    }
}

typedef struct {
    int branch, numSpawn;
    float s, ss, csum, ssum, ess, htheta, m;
    double rStart, rStep, S;
} rs_stats_t;

/* Number of offspring that fell off the end of the CDF (rStart + rStep*i >= cdf[n-1]) and were
 * clamped to the last ancestor.  The reference reads past its stack CDF there (pfResample.c:70-71,
 * undefined behaviour), so a golden taken from it is only meaningful while this stays 0. */
static long g_clamp_events = 0;
long ora_clamp_events(void) { return g_clamp_events; }

/* One filter's resample-or-normalise step: pfResample.c:30-84 (Loc) and :141-201 (one row k of a
 * beacon; `force` as there).  `p` holds n particles starting at element `off`; offspring are built
 * in `buf` (n particles from 0) and copied back (:74/:188).  Respawned particles go around
 * (cx,cy,cz).  Philox sub-streams use particle index off+i; rStart uses index `rs_index`. */
static void resample_block(soa_t* p, size_t off, int n, soa_t* buf, double* cdf, int* anc,
                           ora_rng* rng, uint32_t stream, uint32_t epoch, uint64_t rs_index, int acc,
                           float cx, float cy, float cz, float range, float stdRange,
                           int allow_spawn, int force, rs_stats_t* st)
{
    soa_t q = { p->w + off, p->x + off, p->y + off, p->z + off, p->th + off };
    wsums_t ws;
    weight_pass(q.w, q.th, n, acc, cdf, &ws);
    float s = ws.s;
    float ess = s * s / ws.ss;
    float invN = 1.0f / n;
    int numSpawn = 0;
    if (allow_spawn && s * invN < WEIGHT_SPAWN_THRESH && range < RADIUS_SPAWN_THRESH)
        numSpawn = (int)lroundf(n * PCT_SPAWN);

    memset(st, 0, sizeof(*st));
    st->s = s; st->ss = ws.ss; st->csum = ws.csum; st->ssum = ws.ssum; st->ess = ess; st->numSpawn = numSpawn;
    st->S = ws.S;

    if (ess * invN < RESAMPLE_THRESH || numSpawn > 0 || force) {
        float htheta = theta_bandwidth(ws.csum, ws.ssum, s, ess);
        ora_rng_begin(rng, stream, epoch, ORA_P_RSTART, rs_index);
        if (acc == ORA_ACC_F32) {
            float rStep = invN * s;
            float rStart = ora_rng_uniform(rng) * rStep;
            int j = 0;
            for (int i = 0; i < n; ++i) {
                float u = rStart + rStep * i;
                while (j < n - 1 && !(u < (float)cdf[j])) ++j;
                if (!(u < (float)cdf[j])) g_clamp_events++;
                anc[i] = j;
            }
            st->rStart = rStart; st->rStep = rStep;
        } else {
            double rStep = ws.S / (double)n;
            double rStart = (double)ora_rng_uniform(rng) * rStep;
            ora_resample_indices_f64(cdf, n, rStart, rStep, anc);
            if (!(rStart + rStep * (n - 1) < cdf[n - 1])) g_clamp_events++;
            st->rStart = rStart; st->rStep = rStep;
        }
        for (int i = 0; i < n; ++i) {
            ora_rng_begin(rng, stream, epoch, ORA_P_JITTER, off + (size_t)i);
            spawn_from_other(buf, (size_t)i, &q, (size_t)anc[i], rng, HXYZ, htheta);
        }
        memcpy(q.w, buf->w, sizeof(float) * n); memcpy(q.x, buf->x, sizeof(float) * n);
        memcpy(q.y, buf->y, sizeof(float) * n); memcpy(q.z, buf->z, sizeof(float) * n);
        memcpy(q.th, buf->th, sizeof(float) * n);
        for (int i = 0; i < numSpawn; ++i) {
            ora_rng_begin(rng, stream, epoch, ORA_P_SPAWN, off + (size_t)i);
            spawn_from_range(&q, (size_t)i, rng, cx, cy, cz, range, stdRange);
        }
        st->branch = 1; st->htheta = htheta;
    } else {
        float m = n / s;
        for (int i = 0; i < n; ++i) q.w[i] *= m;
        st->branch = 0; st->m = m;
    }
}

/* particleFilter.c:263-290 / :302-333 -- weighted mean position, circular mean heading, then the
 * VIO delta not yet committed, rotated by the estimated heading */
static void pose_estimate(const soa_t* p, int n, int acc, const vio_t* v,
                          double* t, float* x, float* y, float* z, float* theta)
{
    float s, xs, ys, zs, cs, sn;
    if (acc == ORA_ACC_F32) {
        s = xs = ys = zs = cs = sn = 0.0f;
        for (int i = 0; i < n; ++i) {
            float w = p->w[i];
            s += w; xs += w * p->x[i]; ys += w * p->y[i]; zs += w * p->z[i];
            cs += w * cosf(p->th[i]); sn += w * sinf(p->th[i]);
        }
    } else {
        double S = 0, X = 0, Y = 0, Z = 0, C = 0, Sn = 0;
        for (int i = 0; i < n; ++i) {
            float w = p->w[i];
            float px = w * p->x[i], py = w * p->y[i], pz = w * p->z[i];
            float pc = w * cosf(p->th[i]), ps = w * sinf(p->th[i]);
            S += ora_cdf_weight(w); X += px; Y += py; Z += pz; C += pc; Sn += ps;
        }
        s = (float)S; xs = (float)X; ys = (float)Y; zs = (float)Z; cs = (float)C; sn = (float)Sn;
    }
    *t = v->lastT;
    *x = xs / s; *y = ys / s; *z = zs / s;
    *theta = atan2f(sn, cs);
    float dx = v->lastX - v->firstX, dy = v->lastY - v->firstY, dz = v->lastZ - v->firstZ;
    float co = cosf(*theta), si = sinf(*theta);
    *x += dx * co - dy * si;
    *y += dx * si + dy * co;
    *z += dz;
}

/* ------------------------------------------------------- particleFilterLoc */

struct ora_loc {
    int n, acc;
    soa_t p, buf;
    double* cdf;
    int* anc;
    vio_t v;
    ora_rng* rng;
    uint32_t epoch;
    uint64_t idx0;          /* global index of particle 0 when this filter is one shard of a larger one */
    rs_stats_t st;
};

ora_loc* ora_loc_new(int n, ora_rng* rng, int acc_mode)
{
    ora_loc* pf = (ora_loc*)calloc(1, sizeof(*pf));
    pf->n = n; pf->acc = acc_mode; pf->rng = rng;
    soa_alloc(&pf->p, (size_t)n); soa_alloc(&pf->buf, (size_t)n);
    pf->cdf = (double*)calloc((size_t)n, sizeof(double));
    pf->anc = (int*)calloc((size_t)n, sizeof(int));
    ora_loc_init(pf);
    return pf;
}
void ora_loc_free(ora_loc* pf)
{
    if (!pf) return;
    soa_free(&pf->p); soa_free(&pf->buf); free(pf->cdf); free(pf->anc); free(pf);
}
/* particleFilter.c:32-46 (the seed itself is the rng object's business) */
void ora_loc_init(ora_loc* pf) { vio_zero(&pf->v); pf->v.initialized = 0; pf->epoch = 0; }
int ora_loc_n(const ora_loc* pf) { return pf->n; }
float* ora_loc_array(ora_loc* pf, int which) { return soa_pick(&pf->p, which); }
void ora_loc_set_initialized(ora_loc* pf, int v) { pf->v.initialized = (uint8_t)v; }
int ora_loc_initialized(const ora_loc* pf) { return pf->v.initialized; }
void ora_loc_set_epoch(ora_loc* pf, uint32_t e) { pf->epoch = e; }
uint32_t ora_loc_epoch(const ora_loc* pf) { return pf->epoch; }

void ora_loc_deposit_vio(ora_loc* pf, double t, float x, float y, float z, float dist)
{
    vio_deposit(&pf->v, t, x, y, z, dist);
}

/* pfMeasurement.c:24-49 */
void ora_loc_apply_vio(ora_loc* pf, float dt, float dx, float dy, float dz, float ddist)
{
    float d[5] = { dt, dx, dy, dz, ddist };
    motion_all_from(&pf->p, (size_t)pf->n, pf->rng, 0, pf->epoch, d, pf->idx0);
}

/* pfMeasurement.c:108-125 -- hard 3-sigma range gate */
void ora_loc_apply_range(ora_loc* pf, float bx, float by, float bz, float range, float stdRange)
{
    float minWeight = (range < 3.0f) ? 0.1f : 0.5f;
    for (int i = 0; i < pf->n; ++i) {
        float dx = pf->p.x[i] - bx, dy = pf->p.y[i] - by, dz = pf->p.z[i] - bz;
        float pRange = sqrtf(dx * dx + dy * dy + dz * dz);
        if (fabsf(pRange - range) > 3 * stdRange)
            pf->p.w[i] *= minWeight;
    }
}

/* pfResample.c:30-84 */
void ora_loc_resample(ora_loc* pf, float bx, float by, float bz, float range, float stdRange)
{
    resample_block(&pf->p, 0, pf->n, &pf->buf, pf->cdf, pf->anc, pf->rng, 0, pf->epoch, 0, pf->acc,
                   bx, by, bz, range, stdRange, 1, 0, &pf->st);
}

/* pfInit.c:20-25 */
void ora_loc_spawn_all(ora_loc* pf, float bx, float by, float bz, float range, float stdRange)
{
    for (int i = 0; i < pf->n; ++i) {
        ora_rng_begin(pf->rng, 0, pf->epoch, ORA_P_SPAWN, pf->idx0 + (uint64_t)i);
        spawn_from_range(&pf->p, (size_t)i, pf->rng, bx, by, bz, range, stdRange);
    }
}

/* particleFilter.c:389-404 on its own, and particleFilter.c:274-290 on caller-supplied sums (sharded test double) */
void ora_loc_commit_vio(ora_loc* pf, float d[5]) { vio_commit(&pf->v, d); }
int ora_loc_pose_from_sums(const ora_loc* pf, const double p[6], double* t, float* x, float* y, float* z, float* theta)
{
    if (!pf->v.initialized) return 0;
    float s = (float)p[0], xs = (float)p[1], ys = (float)p[2], zs = (float)p[3], cs = (float)p[4], sn = (float)p[5];
    const vio_t* v = &pf->v;
    *t = v->lastT;
    *x = xs / s; *y = ys / s; *z = zs / s;
    *theta = atan2f(sn, cs);
    float dx = v->lastX - v->firstX, dy = v->lastY - v->firstY, dz = v->lastZ - v->firstZ;
    float co = cosf(*theta), si = sinf(*theta);
    *x += dx * co - dy * si;
    *y += dx * si + dy * co;
    *z += dz;
    return 1;
}

/* ---- helpers for the sharded-filter test double (tests/_oracle_shard.py); Philox mode only ---- */
void ora_loc_set_index_offset(ora_loc* pf, uint64_t idx0) { pf->idx0 = idx0; }

/* Offspring i_begin .. i_begin+count-1 of a resampling step, given their ancestors (indices into the src
 * arrays): the first numSpawn global indices are respawned on the range shell (pfResample.c:75-76), the
 * others are jittered copies (pfInit.c:68-79).  Noise keyed by the GLOBAL offspring index. */
void ora_offspring_block(ora_rng* rng, uint32_t stream, uint32_t epoch, int64_t i_begin, int count, const int* anc,
                         const float* sx, const float* sy, const float* sz, const float* sth, float hTheta, int numSpawn,
                         float cx, float cy, float cz, float range, float stdRange,
                         float* ox, float* oy, float* oz, float* oth)
{
    float one;
    soa_t src = { (float*)sx, (float*)sx, (float*)sy, (float*)sz, (float*)sth };
    for (int k = 0; k < count; ++k) {
        int64_t i = i_begin + k;
        soa_t dst = { &one, ox + k, oy + k, oz + k, oth + k };
        if (i < numSpawn) {
            ora_rng_begin(rng, stream, epoch, ORA_P_SPAWN, (uint64_t)i);
            spawn_from_range(&dst, 0, rng, cx, cy, cz, range, stdRange);
        } else {
            ora_rng_begin(rng, stream, epoch, ORA_P_JITTER, (uint64_t)i);
            spawn_from_other(&dst, 0, &src, (size_t)anc[k], rng, HXYZ, hTheta);
        }
    }
}

/* the scalar decision of pfResample.c:51-68,80 on given f64 totals (f64 rule); out: branch numSpawn htheta m rStart rStep */
void ora_decide(ora_rng* rng, uint32_t stream, uint32_t epoch, double S, double SS, double C, double SN, int64_t n,
                float range, int force, double out[6])
{
    float s = (float)S, ss = (float)SS, csum = (float)C, ssum = (float)SN;
    float ess = s * s / ss;
    float invN = 1.0f / (float)n;
    int numSpawn = 0;
    if (s * invN < WEIGHT_SPAWN_THRESH && range < RADIUS_SPAWN_THRESH) numSpawn = (int)lroundf((float)n * PCT_SPAWN);
    int resample = (ess * invN < RESAMPLE_THRESH) || numSpawn > 0;
    if (force >= 0) resample = force != 0;
    out[0] = resample; out[1] = numSpawn; out[2] = out[3] = out[4] = out[5] = 0.0;
    if (resample) {
        out[2] = theta_bandwidth(csum, ssum, s, ess);
        ora_rng_begin(rng, stream, epoch, ORA_P_RSTART, 0);
        out[5] = S / (double)n;
        out[4] = (double)ora_rng_uniform(rng) * out[5];
    } else {
        out[3] = (float)n / s;
    }
}

/* particleFilter.c:178-191.  The VIO commit runs before the initialised test, over whatever the
 * particle array holds (zeros here), so the serial RNG modes burn the same 4N draws as the reference. */
void ora_loc_deposit_range(ora_loc* pf, float bx, float by, float bz, float range, float stdRange)
{
    float d[5];
    pf->epoch++;
    vio_commit(&pf->v, d);
    motion_all(&pf->p, (size_t)pf->n, pf->rng, 0, pf->epoch, d);
    if (pf->v.initialized) {
        ora_loc_apply_range(pf, bx, by, bz, range, stdRange);
        ora_loc_resample(pf, bx, by, bz, range, stdRange);
    } else {
        ora_loc_spawn_all(pf, bx, by, bz, range, stdRange);
        pf->v.initialized = 1;
    }
}

/* particleFilter.c:213-226 -- the rssi value is ignored; a fixed 1.5 m +- 0.5 m range is applied */
void ora_loc_deposit_rssi(ora_loc* pf, float bx, float by, float bz, int rssi)
{
    (void)rssi;
    ora_loc_deposit_range(pf, bx, by, bz, 1.5f, 0.5f);
}

/* particleFilter.c:248-291 */
int ora_loc_get_tag_loc(const ora_loc* pf, double* t, float* x, float* y, float* z, float* theta)
{
    if (!pf->v.initialized) return 0;
    pose_estimate(&pf->p, pf->n, pf->acc, &pf->v, t, x, y, z, theta);
    return 1;
}

void ora_loc_last_stats(const ora_loc* pf, double out[12])
{
    const rs_stats_t* s = &pf->st;
    out[0] = s->branch; out[1] = s->numSpawn; out[2] = s->s; out[3] = s->ss; out[4] = s->csum; out[5] = s->ssum;
    out[6] = s->ess; out[7] = s->htheta; out[8] = s->rStart; out[9] = s->rStep; out[10] = s->m; out[11] = s->S;
}
const int* ora_loc_last_ancestors(const ora_loc* pf) { return pf->anc; }
const double* ora_loc_last_cdf(const ora_loc* pf) { return pf->cdf; }

/* ------------------------------------------------------ particleFilterSlam */

struct ora_slam {
    int nTag, nBcn, acc;
    soa_t p, buf;            /* tag particles */
    double* cdf; int* anc;   /* sized max(nTag, nBcn) */
    soa_t bbuf;              /* pBcnBuf, nBcn */
    vio_t v;
    ora_rng* rng;
    uint32_t epoch;
    int last_tag_branch;     /* 1 if the last deposit resampled the tag particles */
};

struct ora_bcn {
    ora_slam* owner;
    soa_t p;                 /* nTag rows of nBcn particles */
    vio_t v;
    uint32_t stream;
};

ora_slam* ora_slam_new(int nTag, int nBcn, ora_rng* rng, int acc_mode)
{
    ora_slam* pf = (ora_slam*)calloc(1, sizeof(*pf));
    int m = nTag > nBcn ? nTag : nBcn;
    pf->nTag = nTag; pf->nBcn = nBcn; pf->acc = acc_mode; pf->rng = rng;
    soa_alloc(&pf->p, (size_t)nTag); soa_alloc(&pf->buf, (size_t)nTag); soa_alloc(&pf->bbuf, (size_t)nBcn);
    pf->cdf = (double*)calloc((size_t)m, sizeof(double));
    pf->anc = (int*)calloc((size_t)m, sizeof(int));
    ora_slam_init(pf);
    return pf;
}
void ora_slam_free(ora_slam* pf)
{
    if (!pf) return;
    soa_free(&pf->p); soa_free(&pf->buf); soa_free(&pf->bbuf); free(pf->cdf); free(pf->anc); free(pf);
}
/* particleFilter.c:48-63 + pfInit.c:27-32,47-54: tag particles start at the origin, w = 1, theta = 0 */
void ora_slam_init(ora_slam* pf)
{
    vio_zero(&pf->v);
    for (int i = 0; i < pf->nTag; ++i) {
        pf->p.w[i] = 1.0f; pf->p.x[i] = pf->p.y[i] = pf->p.z[i] = pf->p.th[i] = 0.0f;
    }
    pf->v.initialized = 1;
    pf->epoch = 0;
}
float* ora_slam_array(ora_slam* pf, int which) { return soa_pick(&pf->p, which); }
int ora_slam_last_tag_branch(const ora_slam* pf) { return pf->last_tag_branch; }
void ora_slam_set_epoch(ora_slam* pf, uint32_t e) { pf->epoch = e; }

ora_bcn* ora_bcn_new(ora_slam* pf, uint32_t stream)
{
    ora_bcn* b = (ora_bcn*)calloc(1, sizeof(*b));
    b->owner = pf; b->stream = stream;
    soa_alloc(&b->p, (size_t)pf->nTag * (size_t)pf->nBcn);
    return b;
}
void ora_bcn_free(ora_bcn* b) { if (b) { soa_free(&b->p); free(b); } }
/* particleFilter.c:65-68 -- only clears `initialized` (the VIO scalars are whatever the caller's
 * memory held; every caller in the reference uses zeroed static storage, as calloc gives here) */
void ora_bcn_add(ora_bcn* b) { b->v.initialized = 0; }
float* ora_bcn_array(ora_bcn* b, int which) { return soa_pick(&b->p, which); }
int ora_bcn_initialized(const ora_bcn* b) { return b->v.initialized; }

void ora_slam_deposit_tag_vio(ora_slam* pf, double t, float x, float y, float z, float dist)
{
    vio_deposit(&pf->v, t, x, y, z, dist);
}
void ora_slam_deposit_bcn_vio(ora_bcn* b, double t, float x, float y, float z, float dist)
{
    vio_deposit(&b->v, t, x, y, z, dist);
}

/* pfMeasurement.c:127-152 -- gate every beacon particle against its own tag particle; the tag
 * particle's weight is multiplied by the sum of its row's beacon weights */
static void slam_apply_range(ora_slam* pf, ora_bcn* b, float range, float stdRange)
{
    float minWeight = (range < 3.0f) ? 0.1f : 0.5f;
    for (int i = 0; i < pf->nTag; ++i) {
        float tx = pf->p.x[i], ty = pf->p.y[i], tz = pf->p.z[i];
        float sum32 = 0.0f; double sum64 = 0.0;
        for (int j = 0; j < pf->nBcn; ++j) {
            size_t k = (size_t)i * pf->nBcn + j;
            float dx = tx - b->p.x[k], dy = ty - b->p.y[k], dz = tz - b->p.z[k];
            float pRange = sqrtf(dx * dx + dy * dy + dz * dz);
            if (fabsf(pRange - range) > 3 * stdRange)
                b->p.w[k] *= minWeight;
            sum32 += b->p.w[k]; sum64 += ora_cdf_weight(b->p.w[k]);
        }
        pf->p.w[i] *= (pf->acc == ORA_ACC_F32) ? sum32 : (float)sum64;
    }
}

/* pfResample.c:141-201 */
static void slam_resample_bcn(ora_slam* pf, ora_bcn* b, float range, float stdRange, int force)
{
    rs_stats_t st;
    for (int k = 0; k < pf->nTag; ++k) {
        resample_block(&b->p, (size_t)k * pf->nBcn, pf->nBcn, &pf->bbuf, pf->cdf, pf->anc, pf->rng,
                       b->stream, pf->epoch, (uint64_t)k, pf->acc,
                       pf->p.x[k], pf->p.y[k], pf->p.z[k], range, stdRange, 1, force, &st);
    }
}

/* pfResample.c:86-139.  When the tag particles are resampled the beacon rows stay with their slot
 * (they are not re-indexed to follow the ancestor) and every initialised beacon is force-resampled. */
static void slam_resample(ora_slam* pf, ora_bcn* b, float range, float stdRange, ora_bcn** all, int numBcns)
{
    rs_stats_t st;
    /* no respawn at tag level (allow_spawn = 0); a forced pass is never requested here */
    resample_block(&pf->p, 0, pf->nTag, &pf->buf, pf->cdf, pf->anc, pf->rng, 0, pf->epoch, 0, pf->acc,
                   0, 0, 0, range, stdRange, 0, 0, &st);
    pf->last_tag_branch = st.branch;
    if (st.branch) {
        for (int i = 0; i < numBcns; ++i)
            if (all[i]->v.initialized)
                slam_resample_bcn(pf, all[i], range, stdRange, 1);
    } else {
        slam_resample_bcn(pf, b, range, stdRange, 0);
    }
}

/* pfInit.c:34-45 */
static void slam_init_bcn(ora_slam* pf, ora_bcn* b, float range, float stdRange)
{
    for (int i = 0; i < pf->nTag; ++i)
        for (int j = 0; j < pf->nBcn; ++j) {
            size_t k = (size_t)i * pf->nBcn + j;
            ora_rng_begin(pf->rng, b->stream, pf->epoch, ORA_P_SPAWN, k);
            spawn_from_range(&b->p, k, pf->rng, pf->p.x[i], pf->p.y[i], pf->p.z[i], range, stdRange);
        }
}

/* particleFilter.c:193-211.  Every beacon in `all` gets its VIO commit, initialised or not. */
void ora_slam_deposit_range(ora_slam* pf, ora_bcn* b, float range, float stdRange, ora_bcn** all, int numBcns)
{
    float d[5];
    pf->epoch++;
    pf->last_tag_branch = 0;
    vio_commit(&pf->v, d);
    motion_all(&pf->p, (size_t)pf->nTag, pf->rng, 0, pf->epoch, d);
    for (int i = 0; i < numBcns; ++i) {
        vio_commit(&all[i]->v, d);
        motion_all(&all[i]->p, (size_t)pf->nTag * pf->nBcn, pf->rng, all[i]->stream, pf->epoch, d);
    }
    if (b->v.initialized) {
        slam_apply_range(pf, b, range, stdRange);
        slam_resample(pf, b, range, stdRange, all, numBcns);
    } else {
        slam_init_bcn(pf, b, range, stdRange);
        b->v.initialized = 1;
    }
}

/* particleFilter.c:228-246 */
void ora_slam_deposit_rssi(ora_slam* pf, ora_bcn* b, int rssi, ora_bcn** all, int numBcns)
{
    (void)rssi;
    ora_slam_deposit_range(pf, b, 1.5f, 0.5f, all, numBcns);
}

/* particleFilter.c:293-336 */
int ora_slam_get_tag_loc(const ora_slam* pf, double* t, float* x, float* y, float* z, float* theta)
{
    if (!pf->v.initialized) return 0;
    pose_estimate(&pf->p, pf->nTag, pf->acc, &pf->v, t, x, y, z, theta);
    return 1;
}

/* particleFilter.c:338-387 -- per tag particle the weighted mean of its beacon row, then the
 * tag-weighted mean of those; heading by the same nesting on cos/sin */
int ora_slam_get_bcn_loc(const ora_slam* pf, const ora_bcn* b, double* t, float* x, float* y, float* z, float* theta)
{
    if (!b->v.initialized) return 0;
    float s1 = 0, x1 = 0, y1 = 0, z1 = 0, c1 = 0, n1 = 0;
    double S1 = 0, X1 = 0, Y1 = 0, Z1 = 0, C1 = 0, N1 = 0;
    for (int i = 0; i < pf->nTag; ++i) {
        float w1 = pf->p.w[i];
        float s2, x2, y2, z2, c2, n2;
        if (pf->acc == ORA_ACC_F32) {
            s2 = x2 = y2 = z2 = c2 = n2 = 0.0f;
            for (int j = 0; j < pf->nBcn; ++j) {
                size_t k = (size_t)i * pf->nBcn + j;
                float w2 = b->p.w[k];
                s2 += w2; x2 += w2 * b->p.x[k]; y2 += w2 * b->p.y[k]; z2 += w2 * b->p.z[k];
                c2 += w2 * cosf(b->p.th[k]); n2 += w2 * sinf(b->p.th[k]);
            }
        } else {
            double S2 = 0, X2 = 0, Y2 = 0, Z2 = 0, C2 = 0, N2 = 0;
            for (int j = 0; j < pf->nBcn; ++j) {
                size_t k = (size_t)i * pf->nBcn + j;
                float w2 = b->p.w[k];
                float px = w2 * b->p.x[k], py = w2 * b->p.y[k], pz = w2 * b->p.z[k];
                float pc = w2 * cosf(b->p.th[k]), ps = w2 * sinf(b->p.th[k]);
                S2 += ora_cdf_weight(w2); X2 += px; Y2 += py; Z2 += pz; C2 += pc; N2 += ps;
            }
            s2 = (float)S2; x2 = (float)X2; y2 = (float)Y2; z2 = (float)Z2; c2 = (float)C2; n2 = (float)N2;
        }
        float tx = w1 * x2 / s2, ty = w1 * y2 / s2, tz = w1 * z2 / s2, tc = w1 * c2 / s2, tn = w1 * n2 / s2;
        s1 += w1; x1 += tx; y1 += ty; z1 += tz; c1 += tc; n1 += tn;
        S1 += ora_cdf_weight(w1); X1 += tx; Y1 += ty; Z1 += tz; C1 += tc; N1 += tn;
    }
    if (pf->acc != ORA_ACC_F32) {
        s1 = (float)S1; x1 = (float)X1; y1 = (float)Y1; z1 = (float)Z1; c1 = (float)C1; n1 = (float)N1;
    }
    *t = pf->v.lastT;
    *x = x1 / s1; *y = y1 / s1; *z = z1 / s1;
    *theta = atan2f(n1, c1);
    return 1;
}

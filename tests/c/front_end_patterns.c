/* front_end_patterns.c -- the way the reference's two remaining front ends use the particle-filter API, without the
 * parts of them this image cannot build (no jni.h, no Paho MQTT, no cJSON): SURVEY.md section 8f row 4.
 *
 *   jni   the native side of ArSlam/app/src/main/jni/slam3d-jni.c:15-83 with `long long` handles where the shim has
 *         jlong: filters and beacons on the heap (malloc, never zeroed; :17,24), beacons handed over as an array of
 *         64-bit handles (:51-53), free() without any call into the library (:29-37), and then the same again.
 *   mqtt  mqttlocalize/mqttlocalize.c: one static particleFilterLoc_t (:56) fed by the MQTT callback thread
 *         (depositVio :294, depositRange :301) while main polls getTagLoc (:116) and reads lastX/Y/Z straight from
 *         the struct (:129-131) -- two threads, no lock of their own.
 *
 * Written for this repository (plain C99 + pthreads over include/particleFilter.h); prints every estimate as hex
 * floats so that tests/test_gpu_front_ends.py can compare runs bit for bit.
 */
#include <particleFilter.h>
#include <pfx.h>
#include <math.h>
#include <pthread.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <time.h>

/* ------------------------------------------------------------------ jni */
static long long NewPf(void)
{
    particleFilterSlam_t* pf = (particleFilterSlam_t*)malloc(sizeof(particleFilterSlam_t));
    particleFilterSlam_init(pf);
    return (long long)pf;
}
static long long NewBcn(void)
{
    /* The shim says malloc (:24).  particleFilterSlam_addBcn clears `initialized` and nothing else (particleFilter.c:65-68),
     * so the beacon's VIO scalars are whatever the block held: zero pages when a 2 MB request goes to mmap, as it does in a
     * young Android process -- but glibc raises its mmap threshold once a larger block has been freed (the HIP runtime
     * does that), and the block then comes dirty from the heap: firstT != 0, the first depositBcnVio does not latch, the
     * beacon moves by garbage, in the reference exactly as here.  calloc states what the shim relies on. */
    bcn_t* bcn = (bcn_t*)calloc(1, sizeof(bcn_t));
    particleFilterSlam_addBcn(bcn);
    return (long long)bcn;
}
static void FreePf(long long pf) { free((particleFilterSlam_t*)pf); }
static void FreeBcn(long long bcn) { free((bcn_t*)bcn); }
static void DepositTagVio(long long pf, double t, float x, float y, float z, float dist)
{
    particleFilterSlam_depositTagVio((particleFilterSlam_t*)pf, t, x, y, z, dist);
}
static void DepositBcnVio(long long bcn, double t, float x, float y, float z, float dist)
{
    particleFilterSlam_depositBcnVio((bcn_t*)bcn, t, x, y, z, dist);
}
static void DepositRange(long long pf, long long bcn, float range, float stdRange, long long* bcnArray, int numBcns)
{
    particleFilterSlam_depositRange((particleFilterSlam_t*)pf, (bcn_t*)bcn, range, stdRange, (bcn_t**)bcnArray, numBcns);
}
static void DepositRssi(long long pf, long long bcn, int rssi, long long* bcnArray, int numBcns)
{
    particleFilterSlam_depositRssi((particleFilterSlam_t*)pf, (bcn_t*)bcn, rssi, (bcn_t**)bcnArray, numBcns);
}

static const float kBcnPos[3][3] = { { 4.0f, 1.0f, 0.5f }, { -3.0f, 2.5f, 1.0f }, { 0.5f, -4.0f, 2.0f } };

/* one session of the app: create, 36 measurements, estimates after each, free.  Static storage instead of the heap when
 * `onHeap` is 0 (the reference's C callers, csvslam.c:34-36). */
static void jni_session(const char* label, int onHeap)
{
    static particleFilterSlam_t sPf;
    static bcn_t sBcn[3];
    long long pf, bcns[3];
    particleFilterSeed_set(4242u);
    if (onHeap) {
        pf = NewPf();
        for (int b = 0; b < 3; ++b) bcns[b] = NewBcn();
    } else {
        memset(&sPf, 0, sizeof sPf);
        memset(sBcn, 0, sizeof sBcn);
        particleFilterSlam_init(&sPf);
        pf = (long long)&sPf;
        for (int b = 0; b < 3; ++b) { particleFilterSlam_addBcn(&sBcn[b]); bcns[b] = (long long)&sBcn[b]; }
    }
    printf("%s begin\n", label);
    for (int k = 0; k < 36; ++k) {
        const float a = 0.05f * (float)k;
        const float x = 1.5f * cosf(a) - 1.5f, y = 1.5f * sinf(a), z = 0.02f * (float)k;     /* the tag starts at its origin */
        const int b = k % 3;
        DepositTagVio(pf, 10.0 + 0.1 * k, x, y, z, 0.0f);
        if (k % 7 == 3) DepositBcnVio(bcns[2], 10.0 + 0.1 * k, 0.001f * (float)k, 0.0f, 0.0f, 0.0f);        /* a beacon that moves */
        const float dx = x - kBcnPos[b][0], dy = y - kBcnPos[b][1], dz = z - kBcnPos[b][2];
        const float range = sqrtf(dx * dx + dy * dy + dz * dz) + 0.01f * (float)((k * 7) % 5 - 2);
        if (k % 10 == 9) DepositRssi(pf, bcns[b], -60, bcns, 3);
        else DepositRange(pf, bcns[b], range, 0.1f, bcns, 3);
        double t = 0.0, tb = 0.0;
        float p[4] = { 0, 0, 0, 0 }, q[4] = { 0, 0, 0, 0 };
        const int ok = particleFilterSlam_getTagLoc((const particleFilterSlam_t*)pf, &t, &p[0], &p[1], &p[2], &p[3]);
        const int okb = particleFilterSlam_getBcnLoc((const particleFilterSlam_t*)pf, (const bcn_t*)bcns[b], &tb, &q[0], &q[1], &q[2], &q[3]);
        printf("%s %d tag %d %a %a %a %a %a bcn%d %d %a %a %a %a\n", label, k, ok, t, p[0], p[1], p[2], p[3], b, okb, q[0], q[1], q[2], q[3]);
    }
    /* a beacon that has seen no measurement says so, and leaves the outputs alone (particleFilter.c:344-345) */
    if (onHeap) {
        long long fresh = NewBcn();
        double t = -1.0;
        float q[4] = { -1, -1, -1, -1 };
        const int ok = particleFilterSlam_getBcnLoc((const particleFilterSlam_t*)pf, (const bcn_t*)fresh, &t, &q[0], &q[1], &q[2], &q[3]);
        printf("%s fresh-beacon %d %g %g\n", label, ok, t, q[0]);
        FreeBcn(fresh);
    }
    printf("%s slots %d %d %d\n", label, pfx_slam_slot_of((const particleFilterSlam_t*)pf, (const bcn_t*)bcns[0]),
           pfx_slam_slot_of((const particleFilterSlam_t*)pf, (const bcn_t*)bcns[1]), pfx_slam_slot_of((const particleFilterSlam_t*)pf, (const bcn_t*)bcns[2]));
    printf("%s end\n", label);
    if (onHeap) {
        for (int b = 0; b < 3; ++b) FreeBcn(bcns[b]);
        FreePf(pf);
    }
}

static int run_jni(void)
{
    /* leave the small-block heap dirty: the particleFilterSlam_t that malloc hands out next is not zeroed
     * (particleFilterSlam_init sets every field it reads, particleFilter.c:48-63) */
    void* junk[8];
    for (int i = 0; i < 8; ++i) { junk[i] = malloc(sizeof(particleFilterSlam_t)); memset(junk[i], 0xA5, sizeof(particleFilterSlam_t)); }
    for (int i = 0; i < 8; ++i) free(junk[i]);
    jni_session("static", 0);
    jni_session("heap1", 1);
    jni_session("heap2", 1);        /* the app started over: new structs, quite possibly at the addresses just freed */
    return 0;
}

/* An app that lives long: the filter stays, beacons come and go at ever new addresses (the old ones are kept until the
 * end here, so that malloc cannot hand their addresses out again).  The reference has no limit on this; the library has
 * 32 slots per filter and must hand on those of beacons that are no longer listed. */
static int run_jni_many(void)
{
    particleFilterSeed_set(99u);
    long long pf = NewPf();
    long long old[14 * 3];
    int nOld = 0, k = 0;
    for (int gen = 0; gen < 14; ++gen) {
        long long bcns[3];
        for (int b = 0; b < 3; ++b) bcns[b] = NewBcn();
        int okAll = 1;
        float q[4] = { 0, 0, 0, 0 }, p[4] = { 0, 0, 0, 0 };
        for (int j = 0; j < 9; ++j, ++k) {
            const float a = 0.02f * (float)k;
            const float x = 1.5f * cosf(a) - 1.5f, y = 1.5f * sinf(a), z = 0.0f;
            const int b = k % 3;
            DepositTagVio(pf, 10.0 + 0.1 * k, x, y, z, 0.0f);
            const float dx = x - kBcnPos[b][0], dy = y - kBcnPos[b][1], dz = z - kBcnPos[b][2];
            DepositRange(pf, bcns[b], sqrtf(dx * dx + dy * dy + dz * dz), 0.1f, bcns, 3);
            double t;
            okAll &= particleFilterSlam_getTagLoc((const particleFilterSlam_t*)pf, &t, &p[0], &p[1], &p[2], &p[3]);
            okAll &= particleFilterSlam_getBcnLoc((const particleFilterSlam_t*)pf, (const bcn_t*)bcns[b], &t, &q[0], &q[1], &q[2], &q[3]);
        }
        printf("gen %d ok %d slots %d %d %d tag %a %a %a bcn %a %a %a\n", gen, okAll, pfx_slam_slot_of((const particleFilterSlam_t*)pf, (const bcn_t*)bcns[0]),
               pfx_slam_slot_of((const particleFilterSlam_t*)pf, (const bcn_t*)bcns[1]), pfx_slam_slot_of((const particleFilterSlam_t*)pf, (const bcn_t*)bcns[2]),
               p[0], p[1], p[2], q[0], q[1], q[2]);
        for (int b = 0; b < 3; ++b) old[nOld++] = bcns[b];
    }
    for (int b = 0; b < nOld; ++b) FreeBcn(old[b]);
    FreePf(pf);
    return 0;
}

/* ------------------------------------------------------------------ mqtt */
static particleFilterLoc_t gPf;
static volatile int gDone;
static const float kDeploy[4][3] = { { 5.0f, 5.0f, 0.5f }, { -5.0f, 5.0f, 3.0f }, { 5.0f, -5.0f, 3.0f }, { -5.0f, -5.0f, 0.5f } };
enum { kEvents = 400 };

static void feed_one(int k)
{
    const float a = 0.02f * (float)k;
    const float x = 2.0f * cosf(a), y = 2.0f * sinf(a), z = 1.5f;
    particleFilterLoc_depositVio(&gPf, 100.0 + 0.05 * k, x, y, z, 0.0f);
    const int b = k % 4;
    const float dx = x - kDeploy[b][0], dy = y - kDeploy[b][1], dz = z - kDeploy[b][2];
    const float r = sqrtf(dx * dx + dy * dy + dz * dz) + 0.02f * (float)((k * 5) % 7 - 3);
    if (r > 0.0f && r < 30.0f) particleFilterLoc_depositRange(&gPf, kDeploy[b][0], kDeploy[b][1], kDeploy[b][2], r, 0.1f);
}

static void* callback_thread(void* arg)
{
    (void)arg;
    for (int k = 0; k < kEvents; ++k) {
        feed_one(k);
        if (k % 16 == 0) { struct timespec ts = { 0, 200000 }; nanosleep(&ts, NULL); }     /* messages come in bursts */
    }
    gDone = 1;
    return NULL;
}

static void print_pose(const char* label)
{
    double t = 0.0;
    float p[4] = { 0, 0, 0, 0 };
    const int ok = particleFilterLoc_getTagLoc(&gPf, &t, &p[0], &p[1], &p[2], &p[3]);
    printf("%s %d %a %a %a %a %a last %a %a %a\n", label, ok, t, p[0], p[1], p[2], p[3], gPf.lastX, gPf.lastY, gPf.lastZ);
}

static int run_mqtt(void)
{
    /* one thread, the reference's other callers' way */
    particleFilterSeed_set(777u);
    particleFilterLoc_init(&gPf);
    for (int k = 0; k < kEvents; ++k) feed_one(k);
    print_pose("single");

    /* two threads, mqttlocalize.c's way */
    particleFilterSeed_set(777u);
    particleFilterLoc_init(&gPf);
    gDone = 0;
    pthread_t th;
    if (pthread_create(&th, NULL, callback_thread, NULL)) { perror("pthread_create"); return 1; }
    long polls = 0, answered = 0, finite = 0;
    while (!gDone) {
        double t;
        float x, y, z, theta;
        if (particleFilterLoc_getTagLoc(&gPf, &t, &x, &y, &z, &theta)) {
            ++answered;
            const float dx = gPf.lastX, dy = gPf.lastY, dz = gPf.lastZ;        /* mqttlocalize.c:129-131 */
            if (isfinite(x) && isfinite(y) && isfinite(z) && isfinite(theta) && isfinite(dx) && isfinite(dy) && isfinite(dz)) ++finite;
        }
        ++polls;
        struct timespec ts = { 0, 100000 };
        nanosleep(&ts, NULL);
    }
    pthread_join(th, NULL);
    print_pose("threads");
    printf("polls %ld answered %ld finite %ld\n", polls, answered, finite);
    return 0;
}

int main(int argc, char** argv)
{
    if (argc > 1 && !strcmp(argv[1], "jni")) return run_jni();
    if (argc > 1 && !strcmp(argv[1], "jni-many")) return run_jni_many();
    if (argc > 1 && !strcmp(argv[1], "mqtt")) return run_mqtt();
    fprintf(stderr, "usage: %s jni|jni-many|mqtt\n", argv[0]);
    return 2;
}

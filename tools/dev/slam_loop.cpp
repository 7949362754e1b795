// Config-5 step loop from C++ (no Python): per-call host time against the synchronised step (diagnostic).
// g++ -O2 -std=c++17 -o tools/dev/slam_loop tools/dev/slam_loop.cpp -Iinclude -Lslam3d_amd/lib -lparticlefilter -Wl,-rpath,'$ORIGIN/../../slam3d_amd/lib'
#include "pfx.h"
#include <chrono>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <vector>
static double now() { return std::chrono::duration<double>(std::chrono::steady_clock::now().time_since_epoch()).count(); }
int main(int argc, char** argv)
{
    const int F = argc > 1 ? atoi(argv[1]) : 512, B = argc > 2 ? atoi(argv[2]) : 4, K = 120, W = 30;
    pfx_slam* pf = pfx_slam_create(F, 100, 1000, B, nullptr, 1000003u, 0);
    if (!pf) { fprintf(stderr, "%s\n", pfx_last_error()); return 1; }
    for (int b = 0; b < B; ++b) for (int f = 0; f < F; ++f) pfx_slam_add_bcn(pf, f);
    const float bx[8][3] = {{5,5,.5f},{-5,5,3},{5,-5,3},{-5,-5,.5f},{5,5,3},{-5,5,.5f},{5,-5,.5f},{-5,-5,3}};
    std::vector<double> t(F); std::vector<float> x(F), y(F), z(F), r(F), sd(F, 0.1f); std::vector<int32_t> slot(F);
    double tv = 0, tr = 0, t0 = 0;
    for (int k = 0; k < B + W + K; ++k) {
        if (k == B + W) { pfx_slam_sync(pf); tv = tr = 0; t0 = now(); }
        const float a = 0.025f * k, px = 2 * cosf(a), py = 2 * sinf(a), pz = 1.5f;
        const float* bb = bx[(k % B) % 8];
        const float d = sqrtf((px - bb[0]) * (px - bb[0]) + (py - bb[1]) * (py - bb[1]) + (pz - bb[2]) * (pz - bb[2]));
        for (int f = 0; f < F; ++f) { t[f] = 1000.0 + 0.1 * k; x[f] = px; y[f] = py; z[f] = pz; r[f] = d; slot[f] = k % B; }
        double a0 = now();
        pfx_slam_deposit_tag_vio_batch(pf, t.data(), x.data(), y.data(), z.data(), nullptr);
        double a1 = now();
        if (pfx_slam_deposit_range_batch(pf, slot.data(), r.data(), sd.data())) { fprintf(stderr, "%s\n", pfx_last_error()); return 1; }
        double a2 = now();
        tv += a1 - a0; tr += a2 - a1;
    }
    const double tq = now() - t0;
    pfx_slam_sync(pf);
    const double dt = now() - t0;
    printf("C++ F=%d B=%d: step %.4f ms; enqueue loop %.4f ms (tag_vio %.4f, range %.4f)\n", F, B, dt / K * 1e3, tq / K * 1e3, tv / K * 1e3, tr / K * 1e3);
    pfx_slam_destroy(pf);
    return 0;
}

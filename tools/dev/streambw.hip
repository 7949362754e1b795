// Times the product kernels k_loc_stream<*,*> stand-alone on 16 M particles (same launch shape as pf_loc.hip).
#include "../../slam3d_amd/csrc/pf_loc_kernels.h"
#include <cstdio>
#include <vector>
#include <random>
using namespace pf;
template <int MODE>
__global__ __launch_bounds__(kThreads) void k_var(LocBufs bufs, const LocCtl* __restrict__ ctl, double* __restrict__ tileSums, long long n, Gate ga)
{
    __shared__ double red[4 * 7];
    __shared__ double big[(MODE & 16) ? 7 * 256 : 1];
    const int cur = ctl->cur;
    const float wscale = ctl->wscale;
    float* __restrict__ pw = bufs.a[cur][0];
    const long long base = (long long)blockIdx.x * kTile + (long long)threadIdx.x * kPerThread;
    const float4 w4 = *reinterpret_cast<const float4*>(pw + base);
    const float4 x4 = *reinterpret_cast<const float4*>(bufs.a[cur][1] + base);
    const float4 y4 = *reinterpret_cast<const float4*>(bufs.a[cur][2] + base);
    const float4 z4 = *reinterpret_cast<const float4*>(bufs.a[cur][3] + base);
    const float4 t4 = *reinterpret_cast<const float4*>(bufs.a[cur][4] + base);
    float w[4] = { w4.x, w4.y, w4.z, w4.w }, x[4] = { x4.x, x4.y, x4.z, x4.w }, y[4] = { y4.x, y4.y, y4.z, y4.w };
    float z[4] = { z4.x, z4.y, z4.z, z4.w }, th[4] = { t4.x, t4.y, t4.z, t4.w };
    double acc[7] = { 0, 0, 0, 0, 0, 0, 0 };
    float facc[7] = { 0, 0, 0, 0, 0, 0, 0 };
#pragma unroll
    for (int k = 0; k < 4; ++k) {
        float wi = (base + k < n) ? w[k] * wscale : 0.0f;
        float sn = 0.f, cs = 1.f;
        const float thi = wrap_two_pi(th[k]);
        if (!(MODE & 1)) sincos_bounded(thi, sn, cs); else { sn = thi; cs = thi * 0.5f; }
        const float dx = x[k] - ga.bx, dy = y[k] - ga.by, dz = z[k] - ga.bz;
        const float pRange = (MODE & 8) ? __builtin_amdgcn_sqrtf(dx * dx + dy * dy + dz * dz) : sqrtf(dx * dx + dy * dy + dz * dz);
        if (fabsf(pRange - ga.range) > ga.thresh) wi *= ga.minWeight;
        w[k] = wi;
        if (MODE & 2) { facc[0] += wi; facc[1] += wi*wi; facc[2] += wi*cs; facc[3] += wi*sn; facc[4] += wi*x[k]; facc[5] += wi*y[k]; facc[6] += wi*z[k]; }
        else { acc[0] += wi; acc[1] += wi * wi; acc[2] += wi * cs; acc[3] += wi * sn; acc[4] += wi * x[k]; acc[5] += wi * y[k]; acc[6] += wi * z[k]; }
    }
    *reinterpret_cast<float4*>(pw + base) = make_float4(w[0], w[1], w[2], w[3]);
    if (MODE & 2) for (int k = 0; k < 7; ++k) acc[k] = facc[k];
    if (MODE & 4) { if (threadIdx.x == 0) for (int k = 0; k < 7; ++k) tileSums[(size_t)blockIdx.x * 8 + k] = acc[k]; return; }
    if (MODE & 16) { block_sum_store<7>(acc, big, tileSums + (size_t)blockIdx.x * kSumsPerTile); return; }
    block_sum<7>(acc, red);
    if (threadIdx.x == 0) { double* o = tileSums + (size_t)blockIdx.x * kSumsPerTile; for (int k = 0; k < 7; ++k) o[k] = acc[k]; }
}

template <typename F> float timeit(F f, int reps){ hipEvent_t a,b; hipEventCreate(&a); hipEventCreate(&b); f(); f(); hipDeviceSynchronize(); hipEventRecord(a); for(int i=0;i<reps;++i) f(); hipEventRecord(b); hipEventSynchronize(b); float ms; hipEventElapsedTime(&ms,a,b); return ms/reps; }
__global__ void k_flush(float4* p, long long n4){ long long i=(long long)blockIdx.x*256+threadIdx.x; if(i<n4){ float4 v=p[i]; v.x+=1.f; p[i]=v; } }
template <typename F, typename G> float timeit_cold(F f, G flush, int reps){ hipEvent_t a,b; hipEventCreate(&a); hipEventCreate(&b); float tot=0; for(int i=0;i<reps+2;++i){ flush(); hipEventRecord(a); f(); hipEventRecord(b); hipEventSynchronize(b); float ms; hipEventElapsedTime(&ms,a,b); if(i>=2) tot+=ms; } return tot/reps; }
int main(){
    const long long n = 16000000; const int tiles = (int)(n/1024);
    float* slab; hipMalloc(&slab, n*10*4);
    { std::vector<float> h(n*5); std::mt19937 g(1); std::uniform_real_distribution<float> u(0.f,6.f); for (auto& v:h) v=u(g);
      for (long long i=0;i<n;++i) h[i]=1.0f; hipMemcpy(slab,h.data(),n*5*4,hipMemcpyHostToDevice); hipMemcpy(slab+n*5,h.data(),n*5*4,hipMemcpyHostToDevice); }
    LocBufs b; for(int i=0;i<2;++i) for(int q=0;q<5;++q) b.a[i][q] = slab + (size_t)(i*5+q)*n;
    LocCtl hc{}; hc.cur=0; hc.wscale=1.0f; LocCtl* ctl; hipMalloc(&ctl,sizeof(LocCtl)); hipMemcpy(ctl,&hc,sizeof(LocCtl),hipMemcpyHostToDevice);
    double* ts; hipMalloc(&ts, 8*8*(size_t)tiles);
    RngKey key{1,0,1}; Motion mo{0.05f,0.01f,0.f,3e-4f,3e-7f}; Gate ga{5,5,0.5f,100.f,1000.f,0.5f};   // gate never fires: weights stay 1
    printf("stream<0,1>: %.4f ms\n", timeit([&]{ hipLaunchKernelGGL((k_loc_stream<false,true>), dim3(tiles), dim3(256),0,0,b,ctl,ts,n,0ll,key,mo,ga); },50));
    printf("stream<0,0>: %.4f ms\n", timeit([&]{ hipLaunchKernelGGL((k_loc_stream<false,false>), dim3(tiles), dim3(256),0,0,b,ctl,ts,n,0ll,key,mo,ga); },50));
    printf("stream<1,1>: %.4f ms\n", timeit([&]{ hipLaunchKernelGGL((k_loc_stream<true,true>), dim3(tiles), dim3(256),0,0,b,ctl,ts,n,0ll,key,mo,ga); },50));
    float4* fl; const long long n4 = 40000000; hipMalloc(&fl, n4*16);    // 640 MB read+written between timed launches
    auto flush=[&]{ hipLaunchKernelGGL(k_flush, dim3((unsigned)((n4+255)/256)), dim3(256),0,0,fl,n4); };
    printf("cold (640 MB of other traffic before each launch):\n");
    printf("  stream<0,1>: %.4f ms\n", timeit_cold([&]{ hipLaunchKernelGGL((k_loc_stream<false,true>), dim3(tiles), dim3(256),0,0,b,ctl,ts,n,0ll,key,mo,ga); },flush,20));
    printf("  stream<1,1>: %.4f ms\n", timeit_cold([&]{ hipLaunchKernelGGL((k_loc_stream<true,true>), dim3(tiles), dim3(256),0,0,b,ctl,ts,n,0ll,key,mo,ga); },flush,20));
    printf("  k_var<4> (no block sum): %.4f ms\n", timeit_cold([&]{ hipLaunchKernelGGL((k_var<4>), dim3(tiles), dim3(256),0,0,b,ctl,ts,n,ga); },flush,20));
    printf("  k_var<0> (shuffle sum): %.4f ms\n", timeit_cold([&]{ hipLaunchKernelGGL((k_var<0>), dim3(tiles), dim3(256),0,0,b,ctl,ts,n,ga); },flush,20));
#define VAR(M) printf("k_var<%d>: %.4f ms\n", M, timeit([&]{ hipLaunchKernelGGL((k_var<M>), dim3(tiles), dim3(256),0,0,b,ctl,ts,n,ga); },50));
    VAR(0) VAR(1) VAR(2) VAR(3) VAR(4) VAR(5) VAR(6) VAR(7) VAR(16) VAR(24) VAR(19)
    return 0;
}

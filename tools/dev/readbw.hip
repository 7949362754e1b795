// Microbenchmark: what limits a kernel that streams five float arrays once (the shape of k_loc_stream<false,*> and
// k_loc_pose_tiles)?  Variants: 4 or 8 floats per thread and array, with or without a dependent control-block load
// in front, one tile per workgroup or a grid-stride loop, with or without the w write-back.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
struct Bufs { float* a[2][5]; };
struct Ctl { int cur; float wscale; };

__device__ inline double wave_sum(double v){ for(int o=32;o>0;o>>=1) v+=__shfl_xor(v,o,64); return v; }

template <int PER, bool DEP, bool WRITE>
__global__ __launch_bounds__(256) void k_tile(Bufs b, const Ctl* ctl, double* out, long long n)
{
    __shared__ double red[4];
    const int cur = DEP ? ctl->cur : 0;
    const float ws = DEP ? ctl->wscale : 1.0f;
    const long long base = ((long long)blockIdx.x * 256 + threadIdx.x) * PER;
    float4 v[5][PER/4];
#pragma unroll
    for (int q=0;q<5;++q)
#pragma unroll
        for (int p=0;p<PER/4;++p) v[q][p] = *reinterpret_cast<const float4*>(b.a[cur][q] + base + 4*p);
    double acc = 0;
#pragma unroll
    for (int p=0;p<PER/4;++p) {
        float4 w = v[0][p];
        w.x*=ws; w.y*=ws; w.z*=ws; w.w*=ws;
        acc += (double)(w.x*v[1][p].x + w.y*v[2][p].y + w.z*v[3][p].z + w.w*v[4][p].w);
        if (WRITE) *reinterpret_cast<float4*>(b.a[cur][0] + base + 4*p) = w;
    }
    acc = wave_sum(acc);
    if ((threadIdx.x&63)==0) red[threadIdx.x>>6]=acc;
    __syncthreads();
    if (threadIdx.x==0) out[blockIdx.x] = red[0]+red[1]+red[2]+red[3];
}

template <bool WRITE>
__global__ __launch_bounds__(256) void k_stride(Bufs b, const Ctl* ctl, double* out, long long n)
{
    __shared__ double red[4];
    const int cur = ctl->cur;
    const float ws = ctl->wscale;
    double acc = 0;
    for (long long base = ((long long)blockIdx.x * 256 + threadIdx.x) * 4; base < n; base += (long long)gridDim.x * 1024) {
        float4 v[5];
#pragma unroll
        for (int q=0;q<5;++q) v[q] = *reinterpret_cast<const float4*>(b.a[cur][q] + base);
        float4 w = v[0]; w.x*=ws; w.y*=ws; w.z*=ws; w.w*=ws;
        acc += (double)(w.x*v[1].x + w.y*v[2].y + w.z*v[3].z + w.w*v[4].w);
        if (WRITE) *reinterpret_cast<float4*>(b.a[cur][0] + base) = w;
    }
    acc = wave_sum(acc);
    if ((threadIdx.x&63)==0) red[threadIdx.x>>6]=acc;
    __syncthreads();
    if (threadIdx.x==0) out[blockIdx.x] = red[0]+red[1]+red[2]+red[3];
}

template <typename F> float timeit(F f, int reps){ hipEvent_t a,b; hipEventCreate(&a); hipEventCreate(&b); f(); f(); hipDeviceSynchronize(); hipEventRecord(a); for(int i=0;i<reps;++i) f(); hipEventRecord(b); hipEventSynchronize(b); float ms; hipEventElapsedTime(&ms,a,b); return ms/reps; }

int main(){
    const long long n = 16000000;
    float* slab; hipMalloc(&slab, n*10*4); hipMemset(slab, 0, n*10*4);
    Bufs b; for(int i=0;i<2;++i) for(int q=0;q<5;++q) b.a[i][q] = slab + (size_t)(i*5+q)*n;
    Ctl h{0,1.0f}; Ctl* ctl; hipMalloc(&ctl,sizeof(Ctl)); hipMemcpy(ctl,&h,sizeof(Ctl),hipMemcpyHostToDevice);
    double* out; hipMalloc(&out, 8*(n/1024+1));
    const int tiles = n/1024;
    auto rep = [&](const char* name, float ms, double bytes){ printf("%-44s %.4f ms  %.2f TB/s\n", name, ms, bytes/ms/1e9); };
    rep("tile 4/thread, no dep, read only", timeit([&]{ hipLaunchKernelGGL((k_tile<4,false,false>), dim3(tiles), dim3(256),0,0,b,ctl,out,n); },50), 20.0*n);
    rep("tile 4/thread, dep ctl load, read only", timeit([&]{ hipLaunchKernelGGL((k_tile<4,true,false>), dim3(tiles), dim3(256),0,0,b,ctl,out,n); },50), 20.0*n);
    rep("tile 8/thread, dep, read only", timeit([&]{ hipLaunchKernelGGL((k_tile<8,true,false>), dim3(tiles/2), dim3(256),0,0,b,ctl,out,n); },50), 20.0*n);
    rep("tile 16/thread, dep, read only", timeit([&]{ hipLaunchKernelGGL((k_tile<16,true,false>), dim3(tiles/4), dim3(256),0,0,b,ctl,out,n); },50), 20.0*n);
    rep("tile 4/thread, dep, write w", timeit([&]{ hipLaunchKernelGGL((k_tile<4,true,true>), dim3(tiles), dim3(256),0,0,b,ctl,out,n); },50), 24.0*n);
    rep("tile 8/thread, dep, write w", timeit([&]{ hipLaunchKernelGGL((k_tile<8,true,true>), dim3(tiles/2), dim3(256),0,0,b,ctl,out,n); },50), 24.0*n);
    for (int g : {1024, 2048, 4096, 8192}) { char nm[64]; snprintf(nm,64,"grid-stride %d WGs, read only", g);
        rep(nm, timeit([&]{ hipLaunchKernelGGL((k_stride<false>), dim3(g), dim3(256),0,0,b,ctl,out,n); },50), 20.0*n); }
    rep("grid-stride 2048 WGs, write w", timeit([&]{ hipLaunchKernelGGL((k_stride<true>), dim3(2048), dim3(256),0,0,b,ctl,out,n); },50), 24.0*n);
    return 0;
}

// Microbenchmark: does the 5.1 TB/s "read ceiling" of the streaming passes come from the layout (five SoA arrays 64 MB
// apart) or from the part?  Cold reads: six data sets of 320 MB are used in rotation (the Infinity Cache holds 256 MB).
//   A  five separate arrays, one 1024-particle tile per workgroup (the library's layout)
//   B  the same bytes tile-interleaved: [tile][array][1024]
//   W  with the 4-byte weight written back (the static streaming pass / a normalising SLAM row)
//   C  plain float4 copy (read n, write n) for the guide's 6.3 TB/s figure
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>

__device__ inline double wave_sum(double v){ for(int o=32;o>0;o>>=1) v+=__shfl_xor(v,o,64); return v; }

template <bool INTERLEAVED, bool WRITE>
__global__ __launch_bounds__(256) void k_read(float* base, long long n, double* out)
{
    __shared__ double red[4];
    const long long tile = blockIdx.x;
    float4 v[5];
#pragma unroll
    for (int q = 0; q < 5; ++q) {
        const float* p = INTERLEAVED ? base + (tile * 5 + q) * 1024 + threadIdx.x * 4 : base + (size_t)q * n + tile * 1024 + threadIdx.x * 4;
        v[q] = *reinterpret_cast<const float4*>(p);
    }
    float4 w = v[0];
    double acc = (double)(w.x * v[1].x + w.y * v[2].y + w.z * v[3].z + w.w * v[4].w);
    if (WRITE) {
        w.x *= 1.0001f; w.y *= 1.0001f; w.z *= 1.0001f; w.w *= 1.0001f;
        float* p = INTERLEAVED ? base + (tile * 5) * 1024 + threadIdx.x * 4 : base + tile * 1024 + threadIdx.x * 4;
        *reinterpret_cast<float4*>(p) = w;
    }
    acc = wave_sum(acc);
    if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = acc;
    __syncthreads();
    if (threadIdx.x == 0) out[blockIdx.x] = red[0] + red[1] + red[2] + red[3];
}

// AW with what the library's static pass has in front and around: a dependent load of a control word before the tile
// loads (DEP), 14 KB of LDS per workgroup (LDS), seven workgroups per CU at most (waves_per_eu 7)
template <bool DEP, bool LDS>
__global__ __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(7, 7))) void k_read_like(float* base, long long n, double* out, const int* ctl)
{
    __shared__ double red[LDS ? 7 * 256 : 4];
    const long long tile = blockIdx.x;
    const int cur = DEP ? ctl[0] : 0;
    const float ws = DEP ? __int_as_float(ctl[1]) : 1.0001f;
    float4 v[5];
#pragma unroll
    for (int q = 0; q < 5; ++q) v[q] = *reinterpret_cast<const float4*>(base + (size_t)(q + cur) * n + tile * 1024 + threadIdx.x * 4);
    float4 w = v[0];
    double acc = (double)(w.x * v[1].x + w.y * v[2].y + w.z * v[3].z + w.w * v[4].w);
    w.x *= ws; w.y *= ws; w.z *= ws; w.w *= ws;
    *reinterpret_cast<float4*>(base + (size_t)cur * n + tile * 1024 + threadIdx.x * 4) = w;
    if (LDS) {
        red[threadIdx.x] = acc; red[256 * 6 + threadIdx.x] = acc;
        __syncthreads();
        acc = red[threadIdx.x ^ 64] + red[256 * 6 + (threadIdx.x ^ 128)];
    }
    acc = wave_sum(acc);
    __syncthreads();
    if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = acc;
    __syncthreads();
    if (threadIdx.x == 0) out[blockIdx.x] = red[0] + red[1] + red[2] + red[3];
}

__global__ __launch_bounds__(256) void k_copy(const float4* a, float4* b)
{
    const size_t i = (size_t)blockIdx.x * 256 + threadIdx.x;
    b[i] = a[i];
}

int main()
{
    const long long n = 16 * 1024 * 1024 - 3 * 1024;       // per array (not a power of two apart)
    const int sets = 6, tiles = (int)(n / 1024);
    float* slab; if (hipMalloc(&slab, (size_t)sets * n * 5 * 4) != hipSuccess) { printf("alloc failed\n"); return 1; }
    hipMemset(slab, 0, (size_t)sets * n * 5 * 4);
    double* out; hipMalloc(&out, 8 * (tiles + 1));
    hipEvent_t a, b; hipEventCreate(&a); hipEventCreate(&b);
    auto run = [&](const char* name, auto launch, double bytes) {
        for (int i = 0; i < sets; ++i) launch(i);
        hipDeviceSynchronize();
        const int reps = 36;
        hipEventRecord(a);
        for (int i = 0; i < reps; ++i) launch(i % sets);
        hipEventRecord(b); hipEventSynchronize(b);
        float ms; hipEventElapsedTime(&ms, a, b); ms /= reps;
        printf("%-56s %.4f ms  %.2f TB/s\n", name, ms, bytes / ms / 1e9);
    };
    run("A  SoA, 5 arrays, read only (20 B)", [&](int s) { hipLaunchKernelGGL((k_read<false, false>), dim3(tiles), dim3(256), 0, 0, slab + (size_t)s * n * 5, n, out); }, 20.0 * n);
    run("B  tile-interleaved, read only (20 B)", [&](int s) { hipLaunchKernelGGL((k_read<true, false>), dim3(tiles), dim3(256), 0, 0, slab + (size_t)s * n * 5, n, out); }, 20.0 * n);
    run("AW SoA, read 20 B + write w 4 B", [&](int s) { hipLaunchKernelGGL((k_read<false, true>), dim3(tiles), dim3(256), 0, 0, slab + (size_t)s * n * 5, n, out); }, 24.0 * n);
    run("BW tile-interleaved, read 20 B + write w 4 B", [&](int s) { hipLaunchKernelGGL((k_read<true, true>), dim3(tiles), dim3(256), 0, 0, slab + (size_t)s * n * 5, n, out); }, 24.0 * n);
    int* ctl; hipMalloc(&ctl, 8); { int h[2] = { 0, 0x3f800347 }; hipMemcpy(ctl, h, 8, hipMemcpyHostToDevice); }
    run("AW + 7 workgroups per CU", [&](int s) { hipLaunchKernelGGL((k_read_like<false, false>), dim3(tiles), dim3(256), 0, 0, slab + (size_t)s * n * 5, n, out, ctl); }, 24.0 * n);
    run("AW + 7 per CU + dependent control load", [&](int s) { hipLaunchKernelGGL((k_read_like<true, false>), dim3(tiles), dim3(256), 0, 0, slab + (size_t)s * n * 5, n, out, ctl); }, 24.0 * n);
    run("AW + 7 per CU + 14 KB LDS", [&](int s) { hipLaunchKernelGGL((k_read_like<false, true>), dim3(tiles), dim3(256), 0, 0, slab + (size_t)s * n * 5, n, out, ctl); }, 24.0 * n);
    run("AW + 7 per CU + dependent load + 14 KB LDS", [&](int s) { hipLaunchKernelGGL((k_read_like<true, true>), dim3(tiles), dim3(256), 0, 0, slab + (size_t)s * n * 5, n, out, ctl); }, 24.0 * n);
    run("C  float4 copy, read 10 B + write 10 B per element", [&](int s) { hipLaunchKernelGGL(k_copy, dim3((unsigned)(n * 5 / 2 / 4 / 256)), dim3(256), 0, 0, reinterpret_cast<const float4*>(slab + (size_t)s * n * 5), reinterpret_cast<float4*>(slab + (size_t)s * n * 5 + n * 5 / 2)); }, 20.0 * n);
    run("A  again", [&](int s) { hipLaunchKernelGGL((k_read<false, false>), dim3(tiles), dim3(256), 0, 0, slab + (size_t)s * n * 5, n, out); }, 20.0 * n);
    return 0;
}

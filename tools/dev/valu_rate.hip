// valu_rate.hip -- issue cost of the vector instructions the particle-filter kernels are made of, on gfx950.
// Every workgroup runs a loop of 16 independent copies of one instruction x 512 iterations; the grid puts W waves
// on every SIMD (W = 1, 2, 4, 8).  Reported: shader cycles (s_memtime) per wave-instruction per SIMD, i.e. the
// reciprocal throughput a kernel bound by vector-ALU issue sees.   hipcc --offload-arch=gfx950 -O3 -o valu_rate valu_rate.hip
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
#include <vector>
#include <algorithm>

#define REP16(X) X X X X X X X X X X X X X X X X

template <int OP>
__global__ __launch_bounds__(256) void k(uint64_t* out, float seedf, double seedd, uint32_t seedu)
{
    float f0 = seedf + threadIdx.x, f1 = f0 * 1.1f, f2 = f0 * 1.2f, f3 = f0 * 1.3f;
    double d0 = seedd + threadIdx.x, d1 = d0 * 1.1, d2 = d0 * 1.2, d3 = d0 * 1.3;
    uint32_t u0 = seedu + threadIdx.x, u1 = u0 * 3u, u2 = u0 * 5u, u3 = u0 * 7u;
    uint64_t q0 = u0, q1 = u1;
    float2 p0 = make_float2(f0, f1), p1 = make_float2(f2, f3);
    const uint64_t t0 = __builtin_amdgcn_s_memtime();
    for (int it = 0; it < 512; ++it) {
        if (OP == 0) { REP16(asm volatile("v_fma_f32 %0, %0, %1, %2" : "+v"(f0) : "v"(f1), "v"(f2));) }
        if (OP == 1) { REP16(asm volatile("v_mad_u64_u32 %0, vcc, %1, %2, %0" : "+v"(q0) : "v"(u1), "v"(u2) : "vcc");) }
        if (OP == 2) { REP16(asm volatile("v_mul_hi_u32 %0, %1, %2" : "=v"(u0) : "v"(u1), "v"(u2));) }
        if (OP == 3) { REP16(asm volatile("v_mul_lo_u32 %0, %1, %2" : "=v"(u0) : "v"(u1), "v"(u2));) }
        if (OP == 4) { REP16(asm volatile("v_add_f64 %0, %1, %2" : "=v"(d0) : "v"(d1), "v"(d2));) }
        if (OP == 5) { REP16(asm volatile("v_cmp_lt_f64 vcc, %0, %1" : : "v"(d1), "v"(d2) : "vcc");) }
        if (OP == 6) { REP16(asm volatile("v_cvt_f64_f32 %0, %1" : "=v"(d0) : "v"(f1));) }
        if (OP == 7) { REP16(asm volatile("v_sin_f32 %0, %1" : "=v"(f0) : "v"(f1));) }
        if (OP == 8) { REP16(asm volatile("v_log_f32 %0, %1" : "=v"(f0) : "v"(f1));) }
        if (OP == 9) { REP16(asm volatile("v_sqrt_f32 %0, %1" : "=v"(f0) : "v"(f1));) }
        if (OP == 10) { REP16(asm volatile("v_pk_fma_f32 %0, %1, %2, %0" : "+v"(p0) : "v"(p1), "v"(p1));) }
        if (OP == 11) { REP16(asm volatile("v_bitop3_b32 %0, %1, %2, %3 bitop3:0x96" : "=v"(u0) : "v"(u1), "v"(u2), "v"(u3));) }
        if (OP == 12) { REP16(asm volatile("v_mul_u32_u24 %0, %1, %2" : "=v"(u0) : "v"(u1), "v"(u2));) }
        if (OP == 13) { REP16(asm volatile("v_cndmask_b32 %0, %1, %2, vcc" : "=v"(u0) : "v"(u1), "v"(u2) : "vcc");) }
        if (OP == 14) { REP16(asm volatile("v_cmp_lt_f32 vcc, %0, %1" : : "v"(f1), "v"(f2) : "vcc");) }
        if (OP == 15) { REP16(asm volatile("v_fma_f64 %0, %1, %2, %0" : "+v"(d0) : "v"(d1), "v"(d2));) }
        if (OP == 16) { REP16(asm volatile("v_cmp_lt_u64 vcc, %0, %1" : : "v"(q0), "v"(q1) : "vcc");) }
        if (OP == 17) { REP16(asm volatile("v_cvt_f32_u32 %0, %1" : "=v"(f0) : "v"(u1));) }
        if (OP == 18) { REP16(asm volatile("v_mul_f64 %0, %1, %2" : "=v"(d0) : "v"(d1), "v"(d2));) }
        if (OP == 19) { REP16(asm volatile("v_lshl_add_u64 %0, %1, 0, %2" : "=v"(q0) : "v"(q1), "v"(q0));) }
        if (OP == 20) { REP16(asm volatile("v_rcp_f32 %0, %1" : "=v"(f0) : "v"(f1));) }
        if (OP == 21) { REP16(asm volatile("v_mad_u32_u24 %0, %1, %2, %3" : "=v"(u0) : "v"(u1), "v"(u2), "v"(u3));) }
        if (OP == 22) { REP16(asm volatile("v_pk_mul_f32 %0, %1, %2" : "=v"(p0) : "v"(p1), "v"(p1));) }
        if (OP == 23) { REP16(asm volatile("v_cvt_u32_f32 %0, %1" : "=v"(u0) : "v"(f1));) }
    }
    const uint64_t t1 = __builtin_amdgcn_s_memtime();
    if ((threadIdx.x & 63) == 0) out[(blockIdx.x * 4 + (threadIdx.x >> 6)) * 2] = t1 - t0;
    // keep the results alive
    if (f0 + f3 + (float)d0 + (float)d3 + (float)(u0 + u3) + (float)(q0 + q1) + p0.x + p0.y == 12345.678f) out[1] = 1;
}

static const char* kName[] = { "v_fma_f32", "v_mad_u64_u32", "v_mul_hi_u32", "v_mul_lo_u32", "v_add_f64", "v_cmp_lt_f64", "v_cvt_f64_f32",
                               "v_sin_f32", "v_log_f32", "v_sqrt_f32", "v_pk_fma_f32", "v_bitop3_b32", "v_mul_u32_u24", "v_cndmask_b32",
                               "v_cmp_lt_f32", "v_fma_f64", "v_cmp_lt_u64", "v_cvt_f32_u32", "v_mul_f64", "v_lshl_add_u64", "v_rcp_f32",
                               "v_mad_u32_u24", "v_pk_mul_f32", "v_cvt_u32_f32" };

template <int OP>
void run(uint64_t* d, std::vector<uint64_t>& h)
{
    printf("%-16s", kName[OP]);
    for (int wavesPerSimd : { 1, 2, 4, 8 }) {
        const int blocks = 256 * wavesPerSimd;       // 256 CUs x (wavesPerSimd blocks of 4 waves)
        hipLaunchKernelGGL(k<OP>, dim3(blocks), dim3(256), 0, 0, d, 1.0f, 1.0, 1u);
        hipLaunchKernelGGL(k<OP>, dim3(blocks), dim3(256), 0, 0, d, 1.0f, 1.0, 1u);
        hipDeviceSynchronize();
        hipMemcpy(h.data(), d, sizeof(uint64_t) * 2 * blocks * 4, hipMemcpyDeviceToHost);
        std::vector<uint64_t> c;
        for (int i = 0; i < blocks * 4; ++i) c.push_back(h[2 * i]);
        std::sort(c.begin(), c.end());
        const double med = (double)c[c.size() / 2];
        // cycles per wave-instruction per SIMD: the W waves of a SIMD each issued 512*16 instructions in `med` cycles
        printf("  W=%d: %6.2f", wavesPerSimd, med / (512.0 * 16.0 * wavesPerSimd));
    }
    printf("\n");
}

int main()
{
    uint64_t* d;
    hipMalloc(&d, sizeof(uint64_t) * 2 * 2048 * 4 + 64);
    std::vector<uint64_t> h(2 * 2048 * 4 + 8);
    printf("cycles per wave64 instruction per SIMD (s_memtime ticks), W waves per SIMD\n");
    run<0>(d, h); run<10>(d, h); run<22>(d, h); run<1>(d, h); run<2>(d, h); run<3>(d, h); run<12>(d, h); run<21>(d, h); run<11>(d, h);
    run<13>(d, h); run<14>(d, h); run<17>(d, h); run<23>(d, h); run<4>(d, h); run<15>(d, h); run<18>(d, h); run<5>(d, h); run<16>(d, h);
    run<19>(d, h); run<6>(d, h); run<7>(d, h); run<8>(d, h); run<9>(d, h); run<20>(d, h);
    return 0;
}

// Microbenchmark for gru_f16x3.hip's weave: what does the r-path's VALU stream (80 single instructions, stage-major over 8
// values) cost a lone wave per SIMD -- alone, with one v_mfma_f32_16x16x32_f16 after every N-th instruction, and how many
// cycles do the MFMAs take on their own?  s_memtime cycles per iteration.
#include <hip/hip_runtime.h>
#include <cstdio>
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));

#define M(k) asm volatile("v_mfma_f32_16x16x32_f16 %0, %1, %2, %0" : "+v"(acc[(k) & 7]) : "v"(a), "v"(b))

template <int EVERY, bool VALU, int PAD>
__global__ void __launch_bounds__(256) k(const float* __restrict__ src, float* __restrict__ dst, long long* cyc, int iters) {
    float v[8], h[8], t[8];
    unsigned ph[4], pl[4];
    f32x4 acc[8];
    f16x8 a, b;
    const int lane = threadIdx.x & 63;
#pragma unroll
    for (int i = 0; i < 8; ++i) { v[i] = src[i * 64 + lane]; h[i] = src[512 + i * 64 + lane]; acc[i] = (f32x4){v[i], v[i], v[i], v[i]}; }
#pragma unroll
    for (int i = 0; i < 8; ++i) { a[i] = (_Float16)src[1024 + i]; b[i] = (_Float16)src[1100 + i]; }
    float c0 = src[2000], c1 = src[2001], c2 = src[2002];
    asm volatile("" : "+s"(c0), "+s"(c1), "+s"(c2));
    __syncthreads();
    const long long t0 = __builtin_readcyclecounter();
    for (int it = 0; it < iters; ++it) {
        int n = 0;
#define STEP(stmt) do { if (VALU) { stmt; } ++n; if (EVERY > 0 && n % EVERY == 0) { M(n / EVERY); for (int q = 0; q < PAD; ++q) asm volatile("s_nop 0"); } } while (0)
#pragma unroll
        for (int i = 0; i < 8; ++i) STEP(asm volatile("v_fma_f32 %0, %1, %2, %3" : "=v"(t[i]) : "s"(c0), "v"(acc[i][0]), "v"(acc[i][1])));
#pragma unroll
        for (int i = 0; i < 8; ++i) STEP(asm volatile("v_mul_f32 %0, %1, %0" : "+v"(t[i]) : "s"(c1)));
#pragma unroll
        for (int i = 0; i < 8; ++i) STEP(asm volatile("v_exp_f32 %0, %0" : "+v"(t[i])));
#pragma unroll
        for (int i = 0; i < 8; ++i) STEP(asm volatile("v_add_f32 %0, 1.0, %0" : "+v"(t[i])));
#pragma unroll
        for (int i = 0; i < 8; ++i) STEP(asm volatile("v_rcp_f32 %0, %0" : "+v"(t[i])));
#pragma unroll
        for (int i = 0; i < 8; ++i) STEP(asm volatile("v_mul_f32 %0, %0, %1" : "+v"(t[i]) : "v"(h[i])));
#pragma unroll
        for (int i = 0; i < 4; ++i) STEP(asm volatile("v_cvt_pk_f16_f32 %0, %1, %2" : "=v"(ph[i]) : "v"(t[2 * i]), "v"(t[2 * i + 1])));
#pragma unroll
        for (int i = 0; i < 8; ++i) {
            if (i & 1) STEP(asm volatile("v_cvt_f32_f16_sdwa %0, %1 dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:WORD_1" : "=v"(v[i]) : "v"(ph[i >> 1])));
            else STEP(asm volatile("v_cvt_f32_f16 %0, %1" : "=v"(v[i]) : "v"(ph[i >> 1])));
        }
#pragma unroll
        for (int i = 0; i < 8; ++i) STEP(asm volatile("v_sub_f32 %0, %1, %0" : "+v"(v[i]) : "v"(t[i])));
#pragma unroll
        for (int i = 0; i < 8; ++i) STEP(asm volatile("v_mul_f32 %0, %1, %0" : "+v"(v[i]) : "s"(c2)));
#pragma unroll
        for (int i = 0; i < 4; ++i) STEP(asm volatile("v_cvt_pk_f16_f32 %0, %1, %2" : "=v"(pl[i]) : "v"(v[2 * i]), "v"(v[2 * i + 1])));
        if (!VALU && EVERY == 0) for (int q = 0; q < 20; ++q) M(q);
#pragma unroll
        for (int i = 0; i < 8; ++i) h[i] = __uint_as_float(ph[i >> 1] ^ pl[i >> 1]) ;
    }
    const long long t1 = __builtin_readcyclecounter();
    float r = 0;
#pragma unroll
    for (int i = 0; i < 8; ++i) r += h[i] + acc[i][0] + acc[i][3];
    dst[blockIdx.x * 256 + threadIdx.x] = r;
    if (threadIdx.x == 0 && blockIdx.x == 0) *cyc = t1 - t0;
}

template <typename K>
void run(const char* name, K kern, const float* src, float* dst, long long* cyc) {
    const int iters = 2000;
    hipLaunchKernelGGL(kern, dim3(256), dim3(256), 0, 0, src, dst, cyc, 10);
    hipDeviceSynchronize();
    hipLaunchKernelGGL(kern, dim3(256), dim3(256), 0, 0, src, dst, cyc, iters);
    hipDeviceSynchronize();
    long long c = 0;
    hipMemcpy(&c, cyc, 8, hipMemcpyDeviceToHost);
    printf("%-52s %8.1f cycles per iteration\n", name, (double)c / iters);
}

int main() {
    float *src, *dst;
    long long* cyc;
    hipMalloc(&src, 8192 * 4); hipMalloc(&dst, 256 * 256 * 4); hipMalloc(&cyc, 8);
    float hsrc[8192];
    for (int i = 0; i < 8192; ++i) hsrc[i] = 0.001f * (i % 97) + 0.1f;
    hipMemcpy(src, hsrc, sizeof(hsrc), hipMemcpyHostToDevice);
    run("80 VALU (r path), no MFMA", k<0, true, 0>, src, dst, cyc);
    run("20 MFMA alone", k<0, false, 0>, src, dst, cyc);
    run("80 VALU + MFMA after every 4th (20)", k<4, true, 0>, src, dst, cyc);
    run("80 VALU + MFMA after every 3rd (26)", k<3, true, 0>, src, dst, cyc);
    run("80 VALU + MFMA after every 2nd (40)", k<2, true, 0>, src, dst, cyc);
    run("80 VALU + MFMA after every 8th (10)", k<8, true, 0>, src, dst, cyc);
    run("MFMA after every 4th slot, no VALU (20)", k<4, false, 0>, src, dst, cyc);
    run("MFMA every 4th slot + 3 s_nop each, no VALU", k<4, false, 3>, src, dst, cyc);
    return 0;
}

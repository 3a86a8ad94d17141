// Microbenchmark: issue rate of plain and packed fp32 VALU ops, WPS waves per SIMD (s_memtime cycles per instruction per SIMD).
// Answers: does a wave64 v_add_f32 / v_fma_f32 take 2 or 4 cycles of a SIMD, and does v_pk_*_f32 double the work per slot?
#include <hip/hip_runtime.h>
#include <cstdio>
typedef float f32x2 __attribute__((ext_vector_type(2)));

// KIND 0: v_fma_f32, 1: v_add_f32, 2: v_pk_fma_f32, 3: v_pk_add_f32, 4: v_pk_mul_f32, 5: v_sqrt_f32, 6: v_xor_b32,
// 7: v_pk_mul_lo_u16, 8: v_pk_mad_i16 clamp, 9: v_dot2_i32_i16, 10: v_med3_i32, 11: v_dot4_i32_i8, 12: v_exp_f32, 13: v_cvt_pk_bf16_f32
template <int KIND>
__global__ void __launch_bounds__(1024) k(const float* __restrict__ src, float* __restrict__ dst, long long* cyc, int iters) {
    float v[16];
    f32x2 p[8];
#pragma unroll
    for (int i = 0; i < 16; ++i) v[i] = src[i * 64 + (threadIdx.x & 63)];
#pragma unroll
    for (int i = 0; i < 8; ++i) p[i] = (f32x2){v[2 * i], v[2 * i + 1]};
    const float c = src[1024];
    const f32x2 c2 = {c, c};
    __syncthreads();
    const long long t0 = __builtin_readcyclecounter();
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int r = 0; r < 8; ++r) {
#pragma unroll
            for (int i = 0; i < 16; ++i) {
                if (KIND == 0) asm volatile("v_fma_f32 %0, %0, %1, %0" : "+v"(v[i]) : "v"(c));
                if (KIND == 1) asm volatile("v_add_f32 %0, %0, %1" : "+v"(v[i]) : "v"(c));
                if (KIND == 5) asm volatile("v_sqrt_f32 %0, %0" : "+v"(v[i]));
                if (KIND == 6) asm volatile("v_xor_b32 %0, %0, %1" : "+v"(v[i]) : "v"(c));
                if (KIND == 7) asm volatile("v_pk_mul_lo_u16 %0, %0, %1" : "+v"(v[i]) : "v"(c));
                if (KIND == 8) asm volatile("v_pk_mad_i16 %0, %0, %1, %0 clamp" : "+v"(v[i]) : "v"(c));
                if (KIND == 9) asm volatile("v_dot2_i32_i16 %0, %0, %1, %0" : "+v"(v[i]) : "v"(c));
                if (KIND == 10) asm volatile("v_med3_i32 %0, %0, %1, %1" : "+v"(v[i]) : "v"(c));
                if (KIND == 11) asm volatile("v_dot4_i32_i8 %0, %0, %1, %0" : "+v"(v[i]) : "v"(c));
                if (KIND == 12) asm volatile("v_exp_f32 %0, %0" : "+v"(v[i]));
                if (KIND == 13) asm volatile("v_cvt_pk_bf16_f32 %0, %0, %1" : "+v"(v[i]) : "v"(c));
            }
#pragma unroll
            for (int i = 0; i < 8; ++i) {
                if (KIND == 2) asm volatile("v_pk_fma_f32 %0, %0, %1, %0" : "+v"(p[i]) : "v"(c2));
                if (KIND == 3) asm volatile("v_pk_add_f32 %0, %0, %1" : "+v"(p[i]) : "v"(c2));
                if (KIND == 4) asm volatile("v_pk_mul_f32 %0, %0, %1" : "+v"(p[i]) : "v"(c2));
            }
        }
    }
    const long long t1 = __builtin_readcyclecounter();
    float r = 0;
#pragma unroll
    for (int i = 0; i < 16; ++i) r += v[i];
#pragma unroll
    for (int i = 0; i < 8; ++i) r += p[i].x + p[i].y;
    dst[blockIdx.x * 1024 + threadIdx.x] = r;
    if (threadIdx.x == 0 && blockIdx.x == 0) *cyc = t1 - t0;
}

template <typename K>
void run(const char* name, K kern, int threads, int per_iter, const float* src, float* dst, long long* cyc) {
    const int iters = 2000;
    hipLaunchKernelGGL(kern, dim3(256), dim3(threads), 0, 0, src, dst, cyc, 10);
    hipDeviceSynchronize();
    hipEvent_t a, b;
    hipEventCreate(&a); hipEventCreate(&b);
    hipEventRecord(a, 0);
    hipLaunchKernelGGL(kern, dim3(256), dim3(threads), 0, 0, src, dst, cyc, iters);
    hipEventRecord(b, 0);
    hipEventSynchronize(b);
    float ms = 0; hipEventElapsedTime(&ms, a, b);
    long long c = 0; hipMemcpy(&c, cyc, 8, hipMemcpyDeviceToHost);
    const int wps = threads / 256;
    const double n = (double)iters * per_iter;            // instructions per wave
    printf("%-20s %d wave/SIMD: %.2f s_memtime ticks per instr per wave; wall %.3f ms -> %.2f ns per instr per SIMD (x2.4 = %.2f cyc @2.4GHz)\n",
           name, wps, c / n, ms, ms * 1e6 / (n * wps), ms * 1e6 / (n * wps) * 2.4);
}

int main() {
    float *src, *dst; long long* cyc;
    hipMalloc(&src, 8192 * 4); hipMalloc(&dst, 256 * 1024 * 4); hipMalloc(&cyc, 8);
    hipMemset(src, 0, 8192 * 4);
    for (int threads : {256, 512, 1024}) {
        run("v_fma_f32", k<0>, threads, 128, src, dst, cyc);
        run("v_add_f32", k<1>, threads, 128, src, dst, cyc);
        run("v_pk_fma_f32", k<2>, threads, 64, src, dst, cyc);
        run("v_pk_add_f32", k<3>, threads, 64, src, dst, cyc);
        run("v_pk_mul_f32", k<4>, threads, 64, src, dst, cyc);
        run("v_sqrt_f32", k<5>, threads, 128, src, dst, cyc);
        run("v_xor_b32", k<6>, threads, 128, src, dst, cyc);
        run("v_pk_mul_lo_u16", k<7>, threads, 128, src, dst, cyc);
        run("v_pk_mad_i16 clamp", k<8>, threads, 128, src, dst, cyc);
        run("v_dot2_i32_i16", k<9>, threads, 128, src, dst, cyc);
        run("v_med3_i32", k<10>, threads, 128, src, dst, cyc);
        run("v_dot4_i32_i8", k<11>, threads, 128, src, dst, cyc);
        run("v_exp_f32", k<12>, threads, 128, src, dst, cyc);
        run("v_cvt_pk_bf16_f32", k<13>, threads, 128, src, dst, cyc);
    }
    return 0;
}

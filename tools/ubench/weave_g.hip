// Microbenchmark: an MFMA-bound stream shaped like gru_f16x3.hip's G phase (48 x v_mfma_f32_16x16x32_f16, per chunk of 12 the
// accumulator pattern 0,1,2,3 | 4,5,6,7 | 4,5,6,7) with V independent VALU instructions (or LDS reads) woven in after every
// E-th MFMA: what do the fillers cost?  s_memtime cycles per 48-MFMA iteration, one wave per SIMD.
#include <hip/hip_runtime.h>
#include <cstdio>
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));

template <int E, int V, int KIND, int NACC>
__global__ void __launch_bounds__(256) k(const float* __restrict__ src, float* __restrict__ dst, long long* cyc, int iters) {
    __shared__ float lds[4096];
    float v[8];
    f32x4 acc[12];
    f16x8 a[4], b;
    const int lane = threadIdx.x & 63;
    for (int i = threadIdx.x; i < 4096; i += 256) lds[i] = src[i];
#pragma unroll
    for (int i = 0; i < 8; ++i) v[i] = src[i * 64 + lane];
#pragma unroll
    for (int i = 0; i < 12; ++i) acc[i] = (f32x4){v[i & 7], v[i & 7], v[i & 7], v[i & 7]};
#pragma unroll
    for (int i = 0; i < 8; ++i) { b[i] = (_Float16)src[1100 + i]; for (int q = 0; q < 4; ++q) a[q][i] = (_Float16)src[1024 + 8 * q + i]; }
#pragma unroll
    for (int q = 0; q < 4; ++q) asm volatile("" : "+a"(a[q]));
    float c0 = src[2000];
    asm volatile("" : "+s"(c0));
    __syncthreads();
    const long long t0 = __builtin_readcyclecounter();
    for (int it = 0; it < iters; ++it) {
        int nv = 0;
#pragma unroll
        for (int i = 0; i < 48; ++i) {
            const int r = i % 12;
            const int ai = NACC == 8 ? (r < 4 ? r : 4 + (r - 4) % 4) : r;      // 8: G's pattern (distance 4 on the lo set); 12: all distinct
            asm volatile("v_mfma_f32_16x16x32_f16 %0, %1, %2, %0" : "+v"(acc[ai]) : "a"(a[i & 3]), "v"(b));
            if (E > 0 && (i + 1) % E == 0) {
#pragma unroll
                for (int q = 0; q < V; ++q) {
                    if (KIND == 0) asm volatile("v_add_f32 %0, %1, %0" : "+v"(v[nv & 7]) : "s"(c0));
                    if (KIND == 1) asm volatile("v_exp_f32 %0, %0" : "+v"(v[nv & 7]));
                    if (KIND == 2) { float t; asm volatile("ds_read_b32 %0, %1" : "=v"(t) : "v"(lane * 4)); v[nv & 7] = t; }
                    if (KIND == 3) asm volatile("v_fma_f32 %0, %1, %0, %0" : "+v"(v[nv & 7]) : "s"(c0));
                    ++nv;
                }
            }
        }
        if (KIND == 2) asm volatile("s_waitcnt lgkmcnt(0)");
    }
    const long long t1 = __builtin_readcyclecounter();
    float r = 0;
#pragma unroll
    for (int i = 0; i < 8; ++i) r += v[i];
#pragma unroll
    for (int i = 0; i < 12; ++i) r += acc[i][0] + acc[i][3];
    dst[blockIdx.x * 256 + threadIdx.x] = r + lds[lane];
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
    printf("%-64s %8.1f cycles per 48 MFMAs\n", name, (double)c / iters);
}

int main() {
    float *src, *dst;
    long long* cyc;
    hipMalloc(&src, 8192 * 4); hipMalloc(&dst, 256 * 256 * 4); hipMalloc(&cyc, 8);
    float hsrc[8192];
    for (int i = 0; i < 8192; ++i) hsrc[i] = 0.001f * (i % 97) + 0.1f;
    hipMemcpy(src, hsrc, sizeof(hsrc), hipMemcpyHostToDevice);
    run("48 MFMA, G pattern (8 accumulators)", k<0, 0, 0, 8>, src, dst, cyc);
    run("48 MFMA, 12 distinct accumulators", k<0, 0, 0, 12>, src, dst, cyc);
    run("G pattern + 1 v_add after every 2nd MFMA (24)", k<2, 1, 0, 8>, src, dst, cyc);
    run("G pattern + 1 v_add after every MFMA (48)", k<1, 1, 0, 8>, src, dst, cyc);
    run("G pattern + 2 v_add after every MFMA (96)", k<1, 2, 0, 8>, src, dst, cyc);
    run("G pattern + 3 v_add after every MFMA (144)", k<1, 3, 0, 8>, src, dst, cyc);
    run("12 distinct + 1 v_add after every 2nd MFMA (24)", k<2, 1, 0, 12>, src, dst, cyc);
    run("12 distinct + 2 v_add after every MFMA (96)", k<1, 2, 0, 12>, src, dst, cyc);
    run("G pattern + 1 v_exp after every 2nd MFMA (24)", k<2, 1, 1, 8>, src, dst, cyc);
    run("G pattern + 1 v_fma (VOP3) after every 2nd MFMA (24)", k<2, 1, 3, 8>, src, dst, cyc);
    run("G pattern + 1 ds_read_b32 after every 4th MFMA (12)", k<4, 1, 2, 8>, src, dst, cyc);
    run("G pattern + 4 v_add after every 4th MFMA (48)", k<4, 4, 0, 8>, src, dst, cyc);
    return 0;
}

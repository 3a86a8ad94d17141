// Microbenchmark for a role-split f16x3 frame: per SIMD one "critical" wave (recurrent MFMAs + the activation VALU chain, two
// workgroup barriers per frame) and one "helper" wave (the next frame's 72 x-part MFMAs, same barriers).  Does the helper's matrix
// work fit into the critical wave's VALU / barrier phases?  cycles per frame-like iteration (s_memtime, wave 0).
#include <hip/hip_runtime.h>
#include <cstdio>
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));

#define MF(k) asm volatile("v_mfma_f32_16x16x32_f16 %0, %1, %2, %0" : "+v"(acc[(k) % 8]) : "v"(a), "v"(b))
#define VA(i) asm volatile("v_fma_f32 %0, %1, %0, %0" : "+v"(v[(i) % 8]) : "s"(c0))
#define VT(i) asm volatile("v_exp_f32 %0, %0" : "+v"(v[(i) % 8]))

// MODE 0: critical waves only do their work, helpers only the barriers.  1: helpers run 72 MFMAs per iteration.
// 2: one wave per SIMD does everything (256 threads), the x MFMAs woven 1 per 3 VALU into the VALU phases (today's kernel shape)
template <int MODE>
__global__ void __launch_bounds__(512) k(const float* __restrict__ src, float* __restrict__ dst, long long* cyc, int iters) {
    float v[8];
    f32x4 acc[8];
    f16x8 a, b;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
#pragma unroll
    for (int i = 0; i < 8; ++i) { v[i] = src[i * 64 + lane]; acc[i] = (f32x4){v[i], v[i], v[i], v[i]}; }
#pragma unroll
    for (int i = 0; i < 8; ++i) { a[i] = (_Float16)src[1024 + i]; b[i] = (_Float16)src[1100 + i]; }
    float c0 = src[2000];
    asm volatile("" : "+s"(c0));
    __syncthreads();
    const bool critical = MODE == 2 || wave < 4;
    const long long t0 = __builtin_readcyclecounter();
    for (int it = 0; it < iters; ++it) {
        if (critical) {
#pragma unroll
            for (int i = 0; i < 48; ++i) MF(i);                       // gates, recurrent part
#pragma unroll
            for (int i = 0; i < 60; ++i) {                            // r path
                if (i % 5 == 1 || i % 5 == 3) VT(i); else VA(i);
                if (MODE == 2 && i % 3 == 2) MF(i / 3);
            }
            __builtin_amdgcn_s_barrier();
#pragma unroll
            for (int i = 0; i < 24; ++i) { MF(i); VA(i); if (i % 3 == 0) VT(i); }   // candidate + u sigmoid
#pragma unroll
            for (int i = 0; i < 76; ++i) {                            // candidate path
                if (i % 5 == 1 || i % 5 == 3) VT(i); else VA(i);
                if (MODE == 2 && i % 3 == 2) MF(i / 3);
            }
            if (MODE == 2) {
#pragma unroll
                for (int i = 0; i < 72 - 20 - 25; ++i) MF(i);         // what did not fit beside the VALU
            }
            __builtin_amdgcn_s_barrier();
        } else {
            if (MODE == 1) {
#pragma unroll
                for (int i = 0; i < 36; ++i) MF(i);
            }
            __builtin_amdgcn_s_barrier();
            if (MODE == 1) {
#pragma unroll
                for (int i = 0; i < 36; ++i) MF(i);
            }
            __builtin_amdgcn_s_barrier();
        }
    }
    const long long t1 = __builtin_readcyclecounter();
    float r = 0;
#pragma unroll
    for (int i = 0; i < 8; ++i) r += v[i] + acc[i][0] + acc[i][3];
    dst[blockIdx.x * 512 + threadIdx.x] = r;
    if (threadIdx.x == 0 && blockIdx.x == 0) *cyc = t1 - t0;
}

template <typename K>
void run(const char* name, K kern, int threads, const float* src, float* dst, long long* cyc) {
    const int iters = 1000;
    hipLaunchKernelGGL(kern, dim3(256), dim3(threads), 0, 0, src, dst, cyc, 10);
    hipDeviceSynchronize();
    hipLaunchKernelGGL(kern, dim3(256), dim3(threads), 0, 0, src, dst, cyc, iters);
    hipDeviceSynchronize();
    long long c = 0;
    hipMemcpy(&c, cyc, 8, hipMemcpyDeviceToHost);
    printf("%-78s %8.1f cycles per frame\n", name, (double)c / iters);
}

int main() {
    float *src, *dst;
    long long* cyc;
    hipMalloc(&src, 8192 * 4); hipMalloc(&dst, 256 * 512 * 4); hipMalloc(&cyc, 8);
    float hsrc[8192];
    for (int i = 0; i < 8192; ++i) hsrc[i] = 0.001f * (i % 97) + 0.1f;
    hipMemcpy(src, hsrc, sizeof(hsrc), hipMemcpyHostToDevice);
    run("critical wave alone (72 + 3 MFMA, 160 VALU; helpers idle at the barriers)", k<0>, 512, src, dst, cyc);
    run("critical wave + helper wave with the 72 x-part MFMAs", k<1>, 512, src, dst, cyc);
    run("one wave per SIMD does everything, x MFMAs woven 1 per 3 VALU (today's shape)", k<2>, 256, src, dst, cyc);
    return 0;
}

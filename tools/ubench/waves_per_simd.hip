// Microbenchmark: the frame loop of the resident fp32 GRU kernel in caricature -- per 16-stream group and frame 1568 fp32
// MFMAs (16x16x4), ~200 activation VALU ops per 32 units, two LDS exchanges and two barriers -- run by ONE wave per SIMD
// (4 waves x 2 tiles, as shipped) or by TWO waves per SIMD (8 waves x 1 tile).  Question: does splitting the same work over
// two waves per SIMD shorten the frame (VALU issue 2.3 vs 4.7 cycles per instruction, waits of one wave under the other's
// MFMAs), or do 8-wave barriers and the shared matrix pipe eat it?
#include <hip/hip_runtime.h>
#include <cstdio>
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef float f32x2 __attribute__((ext_vector_type(2)));
#define MFMA_A(acc, wa, bv) asm volatile("v_mfma_f32_16x16x4_f32 %0, %1, %2, %0" : "+v"(acc) : "a"(wa), "v"(bv))

template <int WAVES>        // 4 or 8
__global__ void __launch_bounds__(WAVES * 64) k(const float* __restrict__ src, float* __restrict__ dst, int frames) {
    constexpr int TILES = 8 / WAVES;                 // 16-unit tiles per wave
    constexpr int KH = 32;                           // h-part k-chunks
    extern __shared__ f32x4 lds[];                   // [2][8 tiles][64]
    const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
    float wg[TILES][2][KH], wc[TILES][KH], wx[TILES][3][KH];
#pragma unroll
    for (int j = 0; j < TILES; ++j)
#pragma unroll
        for (int kc = 0; kc < KH; ++kc) {
            wg[j][0][kc] = src[(kc * 7 + j) * 64 + lane]; wg[j][1][kc] = src[(kc * 5 + j + 1) * 64 + lane]; wc[j][kc] = src[(kc * 3 + j + 2) * 64 + lane];
            asm volatile("" : "+a"(wg[j][0][kc])); asm volatile("" : "+a"(wg[j][1][kc])); asm volatile("" : "+a"(wc[j][kc]));
        }
#pragma unroll
    for (int j = 0; j < TILES; ++j)
#pragma unroll
        for (int g = 0; g < 3; ++g)
#pragma unroll
            for (int kc = 0; kc < KH; ++kc) wx[j][g][kc] = src[(kc + g + j) * 64 + lane];    // x-part: VGPR operands here (LDS-fed in the product)
    asm volatile("s_nop 7" ::: "memory");
    f32x4 h[TILES];
#pragma unroll
    for (int j = 0; j < TILES; ++j) { h[j] = (f32x4){0.1f, 0.2f, 0.3f, 0.4f}; lds[(TILES * w + j) * 64 + lane] = h[j]; }
    float xb[KH];
#pragma unroll
    for (int kc = 0; kc < KH; ++kc) xb[kc] = src[4096 + kc * 64 + lane];
    __syncthreads();
    for (int t = 0; t < frames; ++t) {
        f32x4 ar[TILES], au[TILES], ac[TILES];
#pragma unroll
        for (int j = 0; j < TILES; ++j) { ar[j] = (f32x4){0, 0, 0, 0}; au[j] = ar[j]; ac[j] = ar[j]; }
        // x-part of the three gates (independent of the recurrence)
#pragma unroll
        for (int kc = 0; kc < KH; ++kc)
#pragma unroll
            for (int j = 0; j < TILES; ++j) {
                ar[j] = __builtin_amdgcn_mfma_f32_16x16x4f32(wx[j][0][kc], xb[kc], ar[j], 0, 0, 0);
                au[j] = __builtin_amdgcn_mfma_f32_16x16x4f32(wx[j][1][kc], xb[kc], au[j], 0, 0, 0);
                ac[j] = __builtin_amdgcn_mfma_f32_16x16x4f32(wx[j][2][kc], xb[kc], ac[j], 0, 0, 0);
            }
        // gates, h-part
#pragma unroll
        for (int j = 0; j < TILES; ++j) asm volatile("s_nop 3" : "+v"(ar[j]), "+v"(au[j]));
#pragma unroll
        for (int n = 0; n < 8; ++n) {
            const f32x4 hb = lds[n * 64 + lane];
#pragma unroll
            for (int e = 0; e < 4; ++e)
#pragma unroll
                for (int j = 0; j < TILES; ++j) { MFMA_A(ar[j], wg[j][0][4 * n + e], hb[e]); MFMA_A(au[j], wg[j][1][4 * n + e], hb[e]); }
        }
#pragma unroll
        for (int j = 0; j < TILES; ++j) asm volatile("s_nop 15" : "+v"(ar[j]), "+v"(au[j]));
        f32x4 u[TILES];
#pragma unroll
        for (int j = 0; j < TILES; ++j) {
            f32x4 rh;
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                rh[e] = __builtin_amdgcn_rcpf(1.0f + __builtin_amdgcn_exp2f(ar[j][e] * -1.4426950f)) * h[j][e];
                u[j][e] = __builtin_amdgcn_rcpf(1.0f + __builtin_amdgcn_exp2f(au[j][e] * -1.4426950f));
            }
            lds[512 + (TILES * w + j) * 64 + lane] = rh;
        }
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        __builtin_amdgcn_s_barrier();
        asm volatile("" ::: "memory");
#pragma unroll
        for (int j = 0; j < TILES; ++j) asm volatile("s_nop 3" : "+v"(ac[j]));
#pragma unroll
        for (int n = 0; n < 8; ++n) {
            const f32x4 rb = lds[512 + n * 64 + lane];
#pragma unroll
            for (int e = 0; e < 4; ++e)
#pragma unroll
                for (int j = 0; j < TILES; ++j) MFMA_A(ac[j], wc[j][4 * n + e], rb[e]);
        }
#pragma unroll
        for (int j = 0; j < TILES; ++j) asm volatile("s_nop 15" : "+v"(ac[j]));
#pragma unroll
        for (int j = 0; j < TILES; ++j) {
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                const float c = 1.0f - 2.0f * __builtin_amdgcn_rcpf(1.0f + __builtin_amdgcn_exp2f(ac[j][e] * 2.8853901f));
                h[j][e] = u[j][e] * h[j][e] + (1.0f - u[j][e]) * c;
            }
            lds[(TILES * w + j) * 64 + lane] = h[j];
        }
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        __builtin_amdgcn_s_barrier();
        asm volatile("" ::: "memory");
    }
    float r = 0;
#pragma unroll
    for (int j = 0; j < TILES; ++j) r += h[j][0] + h[j][1] + h[j][2] + h[j][3];
    dst[blockIdx.x * WAVES * 64 + threadIdx.x] = r;
}

template <typename K>
void run(const char* name, K kern, int threads, const float* src, float* dst) {
    const int frames = 300;
    hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, 100 * 1024);
    hipLaunchKernelGGL(kern, dim3(256), dim3(threads), 100 * 1024, 0, src, dst, 10);
    hipDeviceSynchronize();
    float best = 1e9f;
    for (int rep = 0; rep < 3; ++rep) {
        hipEvent_t a, b;
        hipEventCreate(&a); hipEventCreate(&b);
        hipEventRecord(a, 0);
        hipLaunchKernelGGL(kern, dim3(256), dim3(threads), 100 * 1024, 0, src, dst, frames);
        hipEventRecord(b, 0);
        hipEventSynchronize(b);
        float ms = 0; hipEventElapsedTime(&ms, a, b);
        best = ms < best ? ms : best;
    }
    const double mfma = 1568.0 * 4 / 4;       // per SIMD and frame: 8 tiles x (96 x + 64 gate-h + 32 cand-h) / 4 SIMDs
    printf("%-28s %.3f ms per 300 frames = %.2f us per frame; MFMA-ideal %.2f us (%.1f %% of it)\n", name, best, best * 1e3 / frames,
           mfma * 32 / 2.4e3, 100.0 * (mfma * 32 / 2.4e3) / (best * 1e3 / frames));
}

int main() {
    float *src, *dst;
    hipMalloc(&src, 1 << 20); hipMalloc(&dst, 256 * 512 * 4);
    hipMemset(src, 0, 1 << 20);
    run("4 waves x 2 tiles (1 / SIMD)", k<4>, 256, src, dst);
    run("8 waves x 1 tile  (2 / SIMD)", k<8>, 512, src, dst);
    return 0;
}

// Caricature of the fused bf16 stack's skewed iteration (gru_bf16.hip): per 16-stream group and frame 340 bf16 MFMAs
// (16x16x32), ~96 sigmoid/tanh evaluations + conversions per 32 units, two LDS exchanges, two barriers -- run by one wave
// per SIMD (4 waves x 2 tiles, as shipped) or two (8 waves x 1 tile).  bf16 MFMAs do not share the FP32 VALU datapath, and a
// lone wave issues a VALU instruction only every ~4.7 cycles (valu_rate.hip), so two waves per SIMD might pay here although
// they do not for the fp32 kernels (waves_per_simd.hip).
#include <hip/hip_runtime.h>
#include <cstdio>
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef short bf16x8 __attribute__((ext_vector_type(8)));
typedef unsigned u32x4 __attribute__((ext_vector_type(4)));
#define MFMA_A(acc, wa, bv) asm volatile("v_mfma_f32_16x16x32_bf16 %0, %1, %2, %0" : "+v"(acc) : "a"(wa), "v"(bv))
__device__ __forceinline__ unsigned pk(float lo, float hi) { unsigned r; asm("v_cvt_pk_bf16_f32 %0, %1, %2" : "=v"(r) : "v"(lo), "v"(hi)); return r; }
__device__ __forceinline__ float sig(float x) { return __builtin_amdgcn_rcpf(1.0f + __builtin_amdgcn_exp2f(x * -1.4426950f)); }

template <int WAVES>
__global__ void __launch_bounds__(WAVES * 64) k(const u32x4* __restrict__ src, float* __restrict__ dst, int frames) {
    constexpr int TILES = 8 / WAVES;
    extern __shared__ u32x4 lds[];            // hb [2 layers][4 chunks][64], rhb [2][4][64]
    const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
    bf16x8 w0[TILES][3][6], w1[TILES][3][8];
#pragma unroll
    for (int j = 0; j < TILES; ++j)
#pragma unroll
        for (int q = 0; q < 3; ++q) {
#pragma unroll
            for (int c = 0; c < 6; ++c) { w0[j][q][c] = __builtin_bit_cast(bf16x8, src[(c * 7 + q + j) * 64 + lane]); asm volatile("" : "+a"(w0[j][q][c])); }
#pragma unroll
            for (int c = 0; c < 8; ++c) { w1[j][q][c] = __builtin_bit_cast(bf16x8, src[(c * 5 + q + j + 1) * 64 + lane]); asm volatile("" : "+a"(w1[j][q][c])); }
        }
    asm volatile("s_nop 7" ::: "memory");
    f32x4 h0[TILES], h1[TILES];
#pragma unroll
    for (int j = 0; j < TILES; ++j) { h0[j] = (f32x4){0.1f, 0.2f, 0.3f, 0.4f}; h1[j] = h0[j]; }
    for (int i = threadIdx.x; i < 16 * 64; i += WAVES * 64) lds[i] = (u32x4){0x3c003c00u, 0x3c003c00u, 0x3c003c00u, 0x3c003c00u};
    const bf16x8 xB0 = __builtin_bit_cast(bf16x8, src[4096 + lane]), xB1 = __builtin_bit_cast(bf16x8, src[4160 + lane]);
    __syncthreads();
    auto put = [&](int base, int j, f32x4 v) {           // this wave's tile -> its half (or whole) of the chunk's 16-byte slot
        const int tile = TILES * w + j;
        uint2* p2 = reinterpret_cast<uint2*>(lds + base + (tile >> 1) * 64 + lane) + (tile & 1);
        *p2 = make_uint2(pk(v[0], v[1]), pk(v[2], v[3]));
    };
    for (int t = 0; t < frames; ++t) {
        bf16x8 hB0[4], hB1[4];
#pragma unroll
        for (int m = 0; m < 4; ++m) { hB0[m] = __builtin_bit_cast(bf16x8, lds[m * 64 + lane]); hB1[m] = __builtin_bit_cast(bf16x8, lds[256 + m * 64 + lane]); }
        f32x4 r0[TILES], u0[TILES], c0[TILES], r1[TILES], u1[TILES], c1[TILES];
#pragma unroll
        for (int j = 0; j < TILES; ++j) { r0[j] = (f32x4){0, 0, 0, 0}; u0[j] = r0[j]; c0[j] = r0[j]; r1[j] = r0[j]; u1[j] = r0[j]; c1[j] = r0[j]; }
#pragma unroll
        for (int j = 0; j < TILES; ++j) asm volatile("s_nop 3" : "+v"(r0[j]), "+v"(u0[j]), "+v"(c0[j]), "+v"(r1[j]), "+v"(u1[j]), "+v"(c1[j]));
#pragma unroll
        for (int j = 0; j < TILES; ++j) {
            MFMA_A(r0[j], w0[j][0][0], xB0); MFMA_A(u0[j], w0[j][1][0], xB0); MFMA_A(c0[j], w0[j][2][0], xB0);
            MFMA_A(r0[j], w0[j][0][1], xB1); MFMA_A(u0[j], w0[j][1][1], xB1); MFMA_A(c0[j], w0[j][2][1], xB1);
        }
#pragma unroll
        for (int m = 0; m < 4; ++m)
#pragma unroll
            for (int j = 0; j < TILES; ++j) {
                MFMA_A(r0[j], w0[j][0][2 + m], hB0[m]); MFMA_A(u0[j], w0[j][1][2 + m], hB0[m]);
                MFMA_A(r1[j], w1[j][0][m], hB0[m]); MFMA_A(u1[j], w1[j][1][m], hB0[m]); MFMA_A(c1[j], w1[j][2][m], hB0[m]);
                MFMA_A(r1[j], w1[j][0][4 + m], hB1[m]); MFMA_A(u1[j], w1[j][1][4 + m], hB1[m]);
            }
#pragma unroll
        for (int j = 0; j < TILES; ++j) asm volatile("s_nop 15" : "+v"(r0[j]), "+v"(u0[j]), "+v"(r1[j]), "+v"(u1[j]));
#pragma unroll
        for (int j = 0; j < TILES; ++j) {
            f32x4 a, b;
#pragma unroll
            for (int e = 0; e < 4; ++e) { a[e] = sig(r0[j][e]) * h0[j][e]; b[e] = sig(r1[j][e]) * h1[j][e]; u0[j][e] = sig(u0[j][e]); u1[j][e] = sig(u1[j][e]); }
            put(512, j, a); put(768, j, b);
        }
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory"); __builtin_amdgcn_s_barrier(); asm volatile("" ::: "memory");
#pragma unroll
        for (int j = 0; j < TILES; ++j) asm volatile("s_nop 3" : "+v"(c0[j]), "+v"(c1[j]));
#pragma unroll
        for (int m = 0; m < 4; ++m) {
            const bf16x8 rB0 = __builtin_bit_cast(bf16x8, lds[512 + m * 64 + lane]), rB1 = __builtin_bit_cast(bf16x8, lds[768 + m * 64 + lane]);
#pragma unroll
            for (int j = 0; j < TILES; ++j) { MFMA_A(c0[j], w0[j][2][2 + m], rB0); MFMA_A(c1[j], w1[j][2][4 + m], rB1); }
        }
#pragma unroll
        for (int j = 0; j < TILES; ++j) asm volatile("s_nop 15" : "+v"(c0[j]), "+v"(c1[j]));
#pragma unroll
        for (int j = 0; j < TILES; ++j) {
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                const float ca = 1.0f - 2.0f * __builtin_amdgcn_rcpf(1.0f + __builtin_amdgcn_exp2f(c0[j][e] * 2.8853901f));
                const float cb = 1.0f - 2.0f * __builtin_amdgcn_rcpf(1.0f + __builtin_amdgcn_exp2f(c1[j][e] * 2.8853901f));
                h0[j][e] = u0[j][e] * h0[j][e] + (1.0f - u0[j][e]) * ca;
                h1[j][e] = u1[j][e] * h1[j][e] + (1.0f - u1[j][e]) * cb;
            }
            put(0, j, h0[j]); put(256, j, h1[j]);
        }
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory"); __builtin_amdgcn_s_barrier(); asm volatile("" ::: "memory");
    }
    float r = 0;
#pragma unroll
    for (int j = 0; j < TILES; ++j) r += h0[j][0] + h1[j][1];
    dst[blockIdx.x * WAVES * 64 + threadIdx.x] = r;
}

template <typename K>
void run(const char* name, K kern, int threads, const u32x4* src, float* dst) {
    const int frames = 300;
    hipLaunchKernelGGL(kern, dim3(256), dim3(threads), 16 * 1024, 0, src, dst, 10);
    hipDeviceSynchronize();
    float best = 1e9f;
    for (int rep = 0; rep < 3; ++rep) {
        hipEvent_t a, b;
        hipEventCreate(&a); hipEventCreate(&b);
        hipEventRecord(a, 0);
        hipLaunchKernelGGL(kern, dim3(256), dim3(threads), 16 * 1024, 0, src, dst, frames);
        hipEventRecord(b, 0);
        hipEventSynchronize(b);
        float ms = 0; hipEventElapsedTime(&ms, a, b);
        best = ms < best ? ms : best;
    }
    printf("%-28s %.3f ms per 300 frames = %.2f us per frame (shipped kernel: ~1.9)\n", name, best, best * 1e3 / frames);
}

int main() {
    u32x4* src; float* dst;
    hipMalloc(&src, 1 << 20); hipMalloc(&dst, 256 * 512 * 4);
    hipMemset(src, 0, 1 << 20);
    run("4 waves x 2 tiles (1 / SIMD)", k<4>, 256, src, dst);
    run("8 waves x 1 tile  (2 / SIMD)", k<8>, 512, src, dst);
    return 0;
}

// Microbenchmark: what does one v_mfma_f32_16x16x4_f32 cost per SIMD under the operand patterns the
// GRU kernel uses?  One wave per SIMD (256 threads/block, 1 block/CU), NW weights held in registers.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
typedef float f32x4 __attribute__((ext_vector_type(4)));
#define MFMA(a, b, c) __builtin_amdgcn_mfma_f32_16x16x4f32(a, b, c, 0, 0, 0)

template <int NW, int NACC, bool VARY_B>
__global__ void __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(1, 1)))
k_regs(const float* __restrict__ src, float* __restrict__ dst, int iters) {
    float w[NW];
#pragma unroll
    for (int i = 0; i < NW; ++i) w[i] = src[i * 64 + (threadIdx.x & 63)];
    float bb[8];
#pragma unroll
    for (int i = 0; i < 8; ++i) bb[i] = src[(NW + i) * 64 + (threadIdx.x & 63)];
    f32x4 acc[NACC];
#pragma unroll
    for (int i = 0; i < NACC; ++i) acc[i] = (f32x4){0, 0, 0, 0};
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int i = 0; i < NW; ++i) acc[i % NACC] = MFMA(w[i], VARY_B ? bb[(i / NACC) & 7] : bb[0], acc[i % NACC]);
    }
    f32x4 s = acc[0];
#pragma unroll
    for (int i = 1; i < NACC; ++i) s += acc[i];
    dst[blockIdx.x * 256 + threadIdx.x] = s[0] + s[1] + s[2] + s[3];
}

// inline-asm MFMA taking the A operand straight from an AGPR ("a" constraint): no v_accvgpr_read
template <int NW, int NACC>
__global__ void __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(1, 1)))
k_agpr(const float* __restrict__ src, float* __restrict__ dst, int iters) {
    float w[NW];
#pragma unroll
    for (int i = 0; i < NW; ++i) w[i] = src[i * 64 + (threadIdx.x & 63)];
    float b0 = src[NW * 64 + (threadIdx.x & 63)];
    f32x4 acc[NACC];
#pragma unroll
    for (int i = 0; i < NACC; ++i) acc[i] = (f32x4){0, 0, 0, 0};
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int i = 0; i < NW; ++i)
            asm volatile("v_mfma_f32_16x16x4_f32 %0, %1, %2, %0" : "+v"(acc[i % NACC]) : "a"(w[i]), "v"(b0));
    }
    asm volatile("s_nop 15\n\ts_nop 15" ::: "memory");
    f32x4 s = acc[0];
#pragma unroll
    for (int i = 1; i < NACC; ++i) s += acc[i];
    dst[blockIdx.x * 256 + threadIdx.x] = s[0] + s[1] + s[2] + s[3];
}

template <typename K>
void run(const char* name, K kern, int nw, const float* src, float* dst, int iters) {
    hipEvent_t a, b;
    hipEventCreate(&a); hipEventCreate(&b);
    hipLaunchKernelGGL(kern, dim3(256), dim3(256), 0, 0, src, dst, 10);
    hipDeviceSynchronize();
    hipEventRecord(a, 0);
    hipLaunchKernelGGL(kern, dim3(256), dim3(256), 0, 0, src, dst, iters);
    hipEventRecord(b, 0);
    hipEventSynchronize(b);
    float ms = 0; hipEventElapsedTime(&ms, a, b);
    double mfma_per_wave = (double)nw * iters;
    double ns_per = ms * 1e6 / mfma_per_wave;
    printf("%-34s %8.3f ms  %6.2f ns/MFMA/SIMD = %5.1f cyc @2.4GHz  -> %6.1f TF\n", name, ms, ns_per, ns_per * 2.4,
           1024.0 * mfma_per_wave * 2048 / (ms * 1e-3) / 1e12);
}

int main() {
    float *src, *dst;
    hipMalloc(&src, 600 * 64 * 4); hipMalloc(&dst, 256 * 256 * 4);
    std::vector<float> h(600 * 64);
    for (size_t i = 0; i < h.size(); ++i) h[i] = (float)((i * 2654435761u) % 1000) * 1e-4f - 0.05f;
    hipMemcpy(src, h.data(), h.size() * 4, hipMemcpyHostToDevice);
    const int it = 4000;
    run("regs NW=32  acc4 fixedB", k_regs<32, 4, false>, 32, src, dst, it * 8);
    run("regs NW=128 acc4 fixedB", k_regs<128, 4, false>, 128, src, dst, it * 2);
    run("regs NW=128 acc4 varyB", k_regs<128, 4, true>, 128, src, dst, it * 2);
    run("regs NW=128 acc2 fixedB", k_regs<128, 2, false>, 128, src, dst, it * 2);
    run("regs NW=128 acc1 fixedB", k_regs<128, 1, false>, 128, src, dst, it * 2);
    run("regs NW=256 acc4 fixedB", k_regs<256, 4, false>, 256, src, dst, it);
    run("regs NW=384 acc4 fixedB", k_regs<384, 4, false>, 384, src, dst, it);
    run("regs NW=384 acc6 varyB", k_regs<384, 6, true>, 384, src, dst, it);
    run("asm-agpr NW=128 acc4", k_agpr<128, 4>, 128, src, dst, it * 2);
    run("asm-agpr NW=240 acc4", k_agpr<240, 4>, 240, src, dst, it);
    run("asm-agpr NW=240 acc2", k_agpr<240, 2>, 240, src, dst, it);
    return 0;
}

// Microbenchmark: cost of VALU / transcendental fillers issued between v_mfma_f32_16x16x32_bf16 by the SAME wave
// (one wave per SIMD).  cycles per MFMA at 2.4 GHz; the bare figure is the matrix pipe's issue interval.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef short bf16x8 __attribute__((ext_vector_type(8)));
#define MFMA(a, b, c) __builtin_amdgcn_mfma_f32_16x16x32_bf16(a, b, c, 0, 0, 0)

// KIND: 0 none, 1 v_add, 2 v_exp, 3 v_rcp, 6 v_pk_mul
template <int KIND, int NF>
__global__ void __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(1, 1)))
k(const float* __restrict__ src, float* __restrict__ dst, int iters) {
    bf16x8 w[16];
#pragma unroll
    for (int i = 0; i < 16; ++i) w[i] = *reinterpret_cast<const bf16x8*>(src + (i * 64 + (threadIdx.x & 63)) * 4);
    bf16x8 b0 = *reinterpret_cast<const bf16x8*>(src + (20 * 64 + (threadIdx.x & 63)) * 4);
    f32x4 acc[4];
#pragma unroll
    for (int i = 0; i < 4; ++i) acc[i] = (f32x4){0, 0, 0, 0};
    float v[16];
#pragma unroll
    for (int i = 0; i < 16; ++i) v[i] = src[(80 + i) * 64 + (threadIdx.x & 63)];
    typedef float f32x2 __attribute__((ext_vector_type(2)));
    f32x2 pk[8];
#pragma unroll
    for (int i = 0; i < 8; ++i) pk[i] = (f32x2){v[2 * i], v[2 * i + 1]};
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int i = 0; i < 64; ++i) {
            acc[i % 4] = MFMA(w[i % 16], b0, acc[i % 4]);
#pragma unroll
            for (int f = 0; f < NF; ++f) {
                const int j = (i * NF + f) % 16;
                if (KIND == 1) v[j] = v[j] + 1.0f;
                if (KIND == 2) v[j] = __builtin_amdgcn_exp2f(v[j]);
                if (KIND == 3) v[j] = __builtin_amdgcn_rcpf(v[j]);
                if (KIND == 6) pk[j % 8] = pk[j % 8] * 1.0001f;
            }
            __builtin_amdgcn_sched_barrier(0);
        }
    }
    f32x4 s = acc[0] + acc[1] + acc[2] + acc[3];
    float r = s[0] + s[1] + s[2] + s[3];
#pragma unroll
    for (int i = 0; i < 16; ++i) r += v[i];
#pragma unroll
    for (int i = 0; i < 8; ++i) r += pk[i].x + pk[i].y;
    dst[blockIdx.x * 256 + threadIdx.x] = r;
}

template <typename K>
void run(const char* name, K kern, const float* src, float* dst, int iters) {
    hipEvent_t a, b;
    hipEventCreate(&a); hipEventCreate(&b);
    hipLaunchKernelGGL(kern, dim3(256), dim3(256), 0, 0, src, dst, 10);
    hipDeviceSynchronize();
    hipEventRecord(a, 0);
    hipLaunchKernelGGL(kern, dim3(256), dim3(256), 0, 0, src, dst, iters);
    hipEventRecord(b, 0);
    hipEventSynchronize(b);
    float ms = 0; hipEventElapsedTime(&ms, a, b);
    double ns_per = ms * 1e6 / (64.0 * iters);
    printf("%-28s %8.3f ms  %6.2f ns/MFMA = %5.1f cyc @2.4GHz\n", name, ms, ns_per, ns_per * 2.4);
}

int main() {
    float *src, *dst;
    hipMalloc(&src, 200 * 64 * 4 * 4); hipMalloc(&dst, 256 * 256 * 4);
    std::vector<float> h(200 * 64 * 4);
    for (size_t i = 0; i < h.size(); ++i) h[i] = (float)((i * 2654435761u) % 1000) * 1e-4f + 0.5f;
    hipMemcpy(src, h.data(), h.size() * 4, hipMemcpyHostToDevice);
    const int it = 8000;
    run("bf16 mfma alone", k<0, 0>, src, dst, it);
    run("+ v_add x1", k<1, 1>, src, dst, it);
    run("+ v_add x2", k<1, 2>, src, dst, it);
    run("+ v_add x3", k<1, 3>, src, dst, it);
    run("+ v_add x4", k<1, 4>, src, dst, it);
    run("+ v_pk_mul x2", k<6, 2>, src, dst, it);
    run("+ v_pk_mul x4", k<6, 4>, src, dst, it);
    run("+ v_exp x1", k<2, 1>, src, dst, it);
    run("+ v_exp x2", k<2, 2>, src, dst, it);
    run("+ v_rcp x1", k<3, 1>, src, dst, it);
    run("+ v_rcp x2", k<3, 2>, src, dst, it);
    return 0;
}

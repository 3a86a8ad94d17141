// Microbenchmark: cost of VALU / transcendental / LDS fillers issued between v_mfma_f32_16x16x4_f32
// by the SAME wave (one wave per SIMD).  cycles per MFMA; 32.0 = matrix pipe saturated.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
typedef float f32x4 __attribute__((ext_vector_type(4)));
#define MFMA(a, b, c) __builtin_amdgcn_mfma_f32_16x16x4f32(a, b, c, 0, 0, 0)

// KIND: 0 none, 1 v_add, 2 v_exp, 3 v_rcp, 4 v_mul dependent chain, 5 ds_read_b128 every 4th
template <int KIND, int NF>
__global__ void __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(1, 1)))
k(const float* __restrict__ src, float* __restrict__ dst, int iters) {
    __shared__ f32x4 lds[256];
    lds[threadIdx.x] = (f32x4){1, 2, 3, 4};
    __syncthreads();
    float w[64];
#pragma unroll
    for (int i = 0; i < 64; ++i) w[i] = src[i * 64 + (threadIdx.x & 63)];
    float b0 = src[70 * 64 + (threadIdx.x & 63)];
    f32x4 acc[4];
#pragma unroll
    for (int i = 0; i < 4; ++i) acc[i] = (f32x4){0, 0, 0, 0};
    float v[16];
#pragma unroll
    for (int i = 0; i < 16; ++i) v[i] = src[(80 + i) * 64 + (threadIdx.x & 63)];
    f32x4 l = lds[threadIdx.x & 63];
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int i = 0; i < 64; ++i) {
            acc[i % 4] = MFMA(w[i], b0, acc[i % 4]);
#pragma unroll
            for (int f = 0; f < NF; ++f) {
                const int j = (i * NF + f) % 16;
                if (KIND == 1) v[j] = v[j] + 1.0f;
                if (KIND == 2) v[j] = __builtin_amdgcn_exp2f(v[j]);
                if (KIND == 3) v[j] = __builtin_amdgcn_rcpf(v[j]);
                if (KIND == 4) v[0] = v[0] * 1.0001f;
            }
            if (KIND == 5 && (i % 4) == 0) { l += lds[(threadIdx.x + i) & 255]; }
            __builtin_amdgcn_sched_barrier(0);
        }
    }
    f32x4 s = acc[0] + acc[1] + acc[2] + acc[3] + l;
    float r = s[0] + s[1] + s[2] + s[3];
#pragma unroll
    for (int i = 0; i < 16; ++i) r += v[i];
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
    hipMalloc(&src, 200 * 64 * 4); hipMalloc(&dst, 256 * 256 * 4);
    std::vector<float> h(200 * 64);
    for (size_t i = 0; i < h.size(); ++i) h[i] = (float)((i * 2654435761u) % 1000) * 1e-4f + 0.5f;
    hipMemcpy(src, h.data(), h.size() * 4, hipMemcpyHostToDevice);
    const int it = 8000;
    run("none", k<0, 0>, src, dst, it);
    run("v_add x1", k<1, 1>, src, dst, it);
    run("v_add x2", k<1, 2>, src, dst, it);
    run("v_add x4", k<1, 4>, src, dst, it);
    run("v_add x6", k<1, 6>, src, dst, it);
    run("v_exp x1", k<2, 1>, src, dst, it);
    run("v_exp x2", k<2, 2>, src, dst, it);
    run("v_exp x4", k<2, 4>, src, dst, it);
    run("v_rcp x1", k<3, 1>, src, dst, it);
    run("v_rcp x2", k<3, 2>, src, dst, it);
    run("v_mul dep-chain x2", k<4, 2>, src, dst, it);
    run("ds_read_b128 every 4th", k<5, 0>, src, dst, it);
    return 0;
}

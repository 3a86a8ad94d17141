// Microbenchmark: how fast can every CU fill its registers with the SAME weight block out of L2 (the prologue of the
// resident GRU kernels: 258 / 393 KB per workgroup)?  Variants: all workgroups read one copy / R replicas at different
// addresses (L2 channel spread) / a single workgroup alone / nt loads.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
typedef float f32x4 __attribute__((ext_vector_type(4)));
constexpr int NLOAD = 96;     // dwordx4 per lane: 96 KB per wave, 384 KB per workgroup

template <int MODE>   // 0 plain, 1 nt
__global__ void __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(1, 1)))
fill(const f32x4* __restrict__ w, size_t replica_stride4, int replicas, float* out, long long* cyc, int reps) {
    const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
    const f32x4* src = w + (size_t)((blockIdx.x / 8) % replicas) * replica_stride4 + (size_t)wv * NLOAD * 64 + lane;
    f32x4 r[NLOAD];
    f32x4 acc = {0, 0, 0, 0};
    const long long t0 = __builtin_readcyclecounter();
    for (int it = 0; it < reps; ++it) {
#pragma unroll
        for (int i = 0; i < NLOAD; ++i) {
            if (MODE == 0) r[i] = src[(size_t)i * 64];
            else r[i] = __builtin_nontemporal_load(src + (size_t)i * 64);
        }
#pragma unroll
        for (int i = 0; i < NLOAD; ++i) acc += r[i];
        asm volatile("" ::: "memory");
    }
    const long long t1 = __builtin_readcyclecounter();
    out[blockIdx.x * 256 + threadIdx.x] = acc[0] + acc[1] + acc[2] + acc[3];
    if (threadIdx.x == 0) cyc[blockIdx.x] = t1 - t0;
}

template <typename K>
void run(const char* name, K kern, int grid, const f32x4* w, size_t stride4, int replicas, float* out, long long* cyc) {
    hipLaunchKernelGGL(kern, dim3(grid), dim3(256), 0, 0, w, stride4, replicas, out, cyc, 1);
    hipDeviceSynchronize();
    float best = 1e9f;
    for (int rep = 0; rep < 5; ++rep) {
        hipEvent_t a, b;
        hipEventCreate(&a); hipEventCreate(&b);
        hipEventRecord(a, 0);
        hipLaunchKernelGGL(kern, dim3(grid), dim3(256), 0, 0, w, stride4, replicas, out, cyc, 1);
        hipEventRecord(b, 0);
        hipEventSynchronize(b);
        float ms = 0; hipEventElapsedTime(&ms, a, b);
        best = ms < best ? ms : best;
    }
    std::vector<long long> c(grid);
    hipMemcpy(c.data(), cyc, grid * 8, hipMemcpyDeviceToHost);
    double s = 0; for (auto v : c) s += (double)v;
    printf("%-34s grid %3d: %.2f us per launch, %.0f ticks inside the kernel (avg) -> %.1f B/tick/CU\n", name, grid, best * 1e3, s / grid,
           4.0 * NLOAD * 1024 / (s / grid));
}

int main() {
    const size_t block4 = (size_t)4 * NLOAD * 64;            // float4 per replica
    const size_t stride4 = block4 + 64 * 37;                  // replicas offset by an odd number of KiB
    f32x4* w; float* out; long long* cyc;
    hipMalloc(&w, 16 * stride4 * 16); hipMalloc(&out, 256 * 256 * 4); hipMalloc(&cyc, 256 * 8);
    hipMemset(w, 0, 16 * stride4 * 16);
    run("one copy, all CUs", fill<0>, 256, w, stride4, 1, out, cyc);
    run("2 replicas", fill<0>, 256, w, stride4, 2, out, cyc);
    run("4 replicas", fill<0>, 256, w, stride4, 4, out, cyc);
    run("8 replicas", fill<0>, 256, w, stride4, 8, out, cyc);
    run("16 replicas", fill<0>, 256, w, stride4, 16, out, cyc);
    run("one copy, nt loads", fill<1>, 256, w, stride4, 1, out, cyc);
    run("one workgroup alone", fill<0>, 1, w, stride4, 1, out, cyc);
    run("8 workgroups (one per XCD)", fill<0>, 8, w, stride4, 1, out, cyc);
    run("64 workgroups", fill<0>, 64, w, stride4, 1, out, cyc);
    return 0;
}

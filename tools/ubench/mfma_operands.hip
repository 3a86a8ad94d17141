// Microbenchmark: the gates_h pattern of gru_layer_resident (A fragment from an AGPR, 4 or 2 accumulators,
// B operand changing every few MFMAs) under different operand placements.  Real cycles via s_memtime.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
typedef float f32x4 __attribute__((ext_vector_type(4)));

#define MFMA_AV(acc, wa, bv) asm volatile("v_mfma_f32_16x16x4_f32 %0, %1, %2, %0" : "+v"(acc) : "a"(wa), "v"(bv))
#define MFMA_AA(acc, wa, bv) asm volatile("v_mfma_f32_16x16x4_f32 %0, %1, %2, %0" : "+a"(acc) : "a"(wa), "v"(bv))
#define MFMA_VA(acc, wa, bv) asm volatile("v_mfma_f32_16x16x4_f32 %0, %1, %2, %0" : "+a"(acc) : "v"(wa), "v"(bv))

// MODE 0: acc VGPR, B fixed | 1: acc VGPR, B rotates over 8 VGPRs | 2: acc VGPR, B from ds_read ring
// MODE 3: acc AGPR, B from ds_read ring | 4: acc AGPR, B rotates | 5: acc VGPR, B ring, NACC=2 | 6: acc AGPR, ring, NACC=2
template <int MODE, int NACC>
__global__ void __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(1, 1)))
k(const float* __restrict__ src, float* __restrict__ dst, unsigned long long* cyc, int iters) {
    __shared__ f32x4 lds[8 * 64];
    for (int i = threadIdx.x; i < 8 * 64; i += 256) lds[i] = (f32x4){1.f + i, 2, 3, 4};
    __syncthreads();
    const int lane = threadIdx.x & 63;
    float w[128];
#pragma unroll
    for (int i = 0; i < 128; ++i) { w[i] = src[i * 64 + lane]; asm volatile("" : "+a"(w[i])); }
    float bb[8];
#pragma unroll
    for (int i = 0; i < 8; ++i) bb[i] = src[(130 + i) * 64 + lane];
    f32x4 acc[4];
#pragma unroll
    for (int i = 0; i < 4; ++i) acc[i] = (f32x4){0, 0, 0, 0};
    asm volatile("s_nop 7" ::: "memory");
    const float4* gsrc = reinterpret_cast<const float4*>(src) + lane;
    float4 gx[8];
#pragma unroll
    for (int i = 0; i < 8; ++i) gx[i] = gsrc[i * 64];
    float gacc = 0.f;
    const unsigned long long t0 = __builtin_readcyclecounter();
    if (MODE == 8) {
        for (int it = 0; it < iters; ++it) {
            f32x4 ring[3];
            ring[0] = lds[0 * 64 + lane]; ring[1] = lds[1 * 64 + lane];
#pragma unroll
            for (int kc = 0; kc < 32; ++kc) {
                if (kc + 2 < 32) ring[(kc + 2) % 3] = lds[((kc + 2) & 7) * 64 + lane];
                __builtin_amdgcn_sched_barrier(0);
                const f32x4 a4 = ring[kc % 3];
                MFMA_VA(acc[0], a4.x, bb[kc & 7]); MFMA_VA(acc[1], a4.y, bb[kc & 7]);
                MFMA_VA(acc[2], a4.z, bb[kc & 7]); MFMA_VA(acc[3], a4.w, bb[kc & 7]);
                __builtin_amdgcn_sched_barrier(0);
            }
        }
    } else
    for (int it = 0; it < iters; ++it) {
        f32x4 ha = lds[0 * 64 + lane], hbv = lds[1 * 64 + lane];
#pragma unroll
        for (int nn = 0; nn < 8; ++nn) {
            const f32x4 hb = (nn & 1) ? hbv : ha;
            if (MODE == 2 || MODE == 3 || MODE == 5 || MODE == 6) {
                if (nn + 2 < 8) { if (nn & 1) hbv = lds[(nn + 2) * 64 + lane]; else ha = lds[(nn + 2) * 64 + lane]; }
            }
            if (MODE == 7) { gacc += gx[nn].x; gx[nn] = gsrc[(size_t)((it * 8 + nn) & 1023) * 64]; }
            __builtin_amdgcn_sched_barrier(0);
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                float bv = (MODE == 0) ? bb[0] : (MODE == 1 || MODE == 4 || MODE == 7) ? bb[(nn * 4 + e) & 7] : hb[e];
#pragma unroll
                for (int q = 0; q < 4; ++q) {
                    const int wi = (nn * 4 + e) * 4 + q;
                    if (MODE == 3 || MODE == 4 || MODE == 6) MFMA_AA(acc[q % NACC], w[wi], bv);
                    else MFMA_AV(acc[q % NACC], w[wi], bv);
                }
            }
        }
    }
    asm volatile("s_nop 15\n\ts_nop 15" ::: "memory");
    const unsigned long long t1 = __builtin_readcyclecounter();
    f32x4 s = acc[0] + acc[1] + acc[2] + acc[3];
    dst[blockIdx.x * 256 + threadIdx.x] = s[0] + s[1] + s[2] + s[3] + gacc;
    if (threadIdx.x == 0 && blockIdx.x == 7) *cyc = t1 - t0;
}

template <typename K>
void run(const char* name, K kern, const float* src, float* dst, unsigned long long* cyc, int iters) {
    hipLaunchKernelGGL(kern, dim3(256), dim3(256), 0, 0, src, dst, cyc, 10);
    hipDeviceSynchronize();
    hipLaunchKernelGGL(kern, dim3(256), dim3(256), 0, 0, src, dst, cyc, iters);
    hipDeviceSynchronize();
    unsigned long long c = 0;
    hipMemcpy(&c, cyc, 8, hipMemcpyDeviceToHost);
    printf("%-44s %7.2f cycles/MFMA (s_memtime)\n", name, (double)c / (128.0 * iters));
}

int main() {
    float *src, *dst; unsigned long long* cyc;
    hipMalloc(&src, 1100 * 64 * 16); hipMalloc(&dst, 256 * 256 * 4); hipMalloc(&cyc, 8);
    std::vector<float> h(1100 * 64 * 4);
    for (size_t i = 0; i < h.size(); ++i) h[i] = (float)((i * 2654435761u) % 1000) * 1e-4f + 0.5f;
    hipMemcpy(src, h.data(), h.size() * 4, hipMemcpyHostToDevice);
    const int it = 3000;
    run("A=agpr acc=vgpr x4  B fixed", k<0, 4>, src, dst, cyc, it);
    run("A=agpr acc=vgpr x4  B rotating vgprs", k<1, 4>, src, dst, cyc, it);
    run("A=agpr acc=vgpr x4  B from ds_read ring", k<2, 4>, src, dst, cyc, it);
    run("A=agpr acc=AGPR x4  B from ds_read ring", k<3, 4>, src, dst, cyc, it);
    run("A=agpr acc=AGPR x4  B rotating vgprs", k<4, 4>, src, dst, cyc, it);
    run("A=agpr acc=vgpr x2  B from ds_read ring", k<5, 2>, src, dst, cyc, it);
    run("A=agpr acc=AGPR x2  B from ds_read ring", k<6, 2>, src, dst, cyc, it);
    run("A=agpr acc=vgpr x4 +1 global_load_x4 /16 MFMA", k<7, 4>, src, dst, cyc, it);
    run("A=ds_read ring(1 per 4 MFMA) acc=AGPR x4", k<8, 4>, src, dst, cyc, it);
    return 0;
}

// Microbenchmark for VERDICT r5 item 5: could configs[4] (4 x GRU h = 256, n_mel = 60, 1024 streams) run on the fp16 matrix
// pipe at fp32 tolerance (the f16x3 split) with the weights STREAMED FROM L2 every frame, as gru_stack_generic_pipelined streams
// its fp32 weights today (28.8 us per frame, 0.63 of the fp32 MFMA peak)?
//
// An h = 256 layer has (256 + 256) x 768 weights; split into fp16 (hi, lo) pairs that is again 4 bytes per weight: 1.5 MiB per
// layer that every workgroup (16 streams) re-reads every frame.  The layer-pipelined launch runs all 4 layers x 64 groups = 256
// workgroups at once, XCD-affine (block i serves layer (i % 8) % 4: the workgroups behind one L2 all stream the SAME layer, so
// each L2 holds 1.5 MiB), i.e. 384 MiB leave the L2s per frame.  The matrix work is small next to that: 768 k-chunk x tile
// products x 3 MFMAs of 16 cycles = 576 MFMAs per wave = 9.2 k cycles per frame.  So the question is what the L2 -> CU path
// delivers to 256 CUs that all stream at once: MI355X_MICROARCH.md gives 34.5 TB/s aggregate (64 B / clk / CU) = 11.7 us per
// frame; the stop rule is "projected < 1.5 x today's 28.8 us per frame -> record and stop".
//
//   MODE 0  loads only (every 16-byte lane value is folded into a checksum with one v_xor): the L2 -> register ceiling
//   MODE 1  the product shape: per (hi, lo) operand pair three v_mfma_f32_16x16x32_f16 (main += Wh Xh, lo += Wl Xh, lo += Wh Xl)
//   MODE 2  MODE 1 + per frame the activation / exchange skeleton a GRU frame adds (two workgroup barriers, 2 x 48 scalar VALU
//           instructions, two LDS round trips) -- the recurrence forbids streaming across frames
// build: hipcc --offload-arch=gfx950 -O3 -std=c++17 -o _bin/l2_stream_f16x3 l2_stream_f16x3.hip
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));
typedef unsigned u32x4 __attribute__((ext_vector_type(4)));

constexpr int kPairsPerLayer = 768;          // (hi, lo) operand pairs of 2 x 1 KiB per layer: 48 unit tiles x 16 k-chunks
constexpr int kLayers = 4;
#ifndef DEPTH
#define DEPTH 8
#endif
constexpr int kDepth = DEPTH;                // operand pairs in flight per wave (2 KiB each)

__device__ __forceinline__ f16x8 as_f16x8(u32x4 v) { return __builtin_bit_cast(f16x8, v); }

template <int MODE>
__global__ void __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(1, 1)))
stream_kernel(const u32x4* __restrict__ w, float* __restrict__ out, int frames) {
    const int tid = threadIdx.x, wave = tid >> 6, lane = tid & 63;
    const int layer = (blockIdx.x % 8) % kLayers;          // XCD-affine: blocks are dealt round-robin to the 8 XCDs
    // [layer][pair][hi|lo][64 lanes] x 16 B; wave w takes pairs w, w + 4, ... (192 per frame)
    const u32x4* base0 = w + (size_t)layer * kPairsPerLayer * 2 * 64 + lane;
    __shared__ u32x4 xch[2][4][64];
    f32x4 am[4], al[4];
#pragma unroll
    for (int i = 0; i < 4; ++i) { am[i] = (f32x4){0.f, 0.f, 0.f, 0.f}; al[i] = am[i]; }
    u32x4 xh = (u32x4){0x3c003c00u + lane, 0x3c003c00u, 0x38003800u, 0x3c003c00u}, xl = (u32x4){0x10001000u, 0x10001000u + lane, 0x10001000u, 0x10001000u};
    u32x4 sum = (u32x4){0u, 0u, 0u, 0u};
    constexpr int per_wave = kPairsPerLayer / 4;
    for (int f = 0; f < frames; ++f) {
        // the weights are re-read every frame (a frame's operands are the same addresses: keep the compiler from hoisting the loads)
        int opaque = 0;
        asm volatile("" : "+v"(opaque));
        const u32x4* base = base0 + opaque;
        u32x4 bh[kDepth], bl[kDepth];
#pragma unroll
        for (int d = 0; d < kDepth; ++d) {
            bh[d] = base[(size_t)((wave + 4 * d) * 2 + 0) * 64];
            bl[d] = base[(size_t)((wave + 4 * d) * 2 + 1) * 64];
        }
        for (int k = 0; k < per_wave; k += kDepth) {
#pragma unroll
            for (int d = 0; d < kDepth; ++d) {
                const u32x4 h = bh[d], l = bl[d];
                // straight-line code (the tail re-reads the last operands): a branch here makes every use wait for vmcnt(0)
                const int nk = k + kDepth + d < per_wave ? k + kDepth + d : per_wave - 1;
                {
                    bh[d] = base[(size_t)((wave + 4 * nk) * 2 + 0) * 64];
                    bl[d] = base[(size_t)((wave + 4 * nk) * 2 + 1) * 64];
                }
                if constexpr (MODE == 0) {
                    sum ^= h;
                    sum ^= l;
                } else {
                    am[d & 3] = __builtin_amdgcn_mfma_f32_16x16x32_f16(as_f16x8(h), as_f16x8(xh), am[d & 3], 0, 0, 0);
                    al[d & 3] = __builtin_amdgcn_mfma_f32_16x16x32_f16(as_f16x8(l), as_f16x8(xh), al[d & 3], 0, 0, 0);
                    al[d & 3] = __builtin_amdgcn_mfma_f32_16x16x32_f16(as_f16x8(h), as_f16x8(xl), al[d & 3], 0, 0, 0);
                }
            }
        }
        if constexpr (MODE == 2) {
            // a GRU frame's skeleton: activations on the gate accumulators -> LDS -> barrier -> read back -> candidate part ->
            // LDS -> barrier (the operands of the next frame cannot be requested before its input exists)
#pragma unroll
            for (int ph = 0; ph < 2; ++ph) {
                float v[8];
#pragma unroll
                for (int i = 0; i < 8; ++i) v[i] = am[i & 3][i >> 2] + al[i & 3][i >> 2] * (1.f / 2048.f);
#pragma unroll
                for (int i = 0; i < 8; ++i) v[i] = __builtin_amdgcn_rcpf(1.0f + __builtin_amdgcn_exp2f(v[i]));
#pragma unroll
                for (int i = 0; i < 8; ++i) v[i] = v[i] * 0.5f + 0.25f;
                xch[ph][wave][lane] = (u32x4){__float_as_uint(v[0]) ^ __float_as_uint(v[4]), __float_as_uint(v[1]) ^ __float_as_uint(v[5]),
                                              __float_as_uint(v[2]) ^ __float_as_uint(v[6]), __float_as_uint(v[3]) ^ __float_as_uint(v[7])};
                __syncthreads();
                const u32x4 r = xch[ph][(wave + 1) & 3][lane];
                xh ^= (u32x4){r[0] & 0x00010001u, r[1] & 0x00010001u, r[2] & 0x00010001u, r[3] & 0x00010001u};
            }
        }
    }
    float acc = 0.f;
#pragma unroll
    for (int i = 0; i < 4; ++i) acc += am[i][0] + al[i][1];
    out[blockIdx.x * 256 + tid] = acc + (float)(sum[0] ^ sum[1] ^ sum[2] ^ sum[3]);
}

int main(int argc, char** argv) {
    const int frames = argc > 1 ? atoi(argv[1]) : 300, blocks = argc > 2 ? atoi(argv[2]) : 256;
    const size_t n16 = (size_t)kLayers * kPairsPerLayer * 2 * 64;
    std::vector<unsigned> h(n16 * 4);
    for (size_t i = 0; i < h.size(); ++i) h[i] = 0x2c002c00u + (unsigned)((i * 2654435761u) >> 22 & 0x03ff03ffu);   // small fp16 values
    u32x4* d_w; float* d_out;
    hipMalloc(&d_w, n16 * 16); hipMalloc(&d_out, (size_t)blocks * 256 * 4);
    hipMemcpy(d_w, h.data(), n16 * 16, hipMemcpyHostToDevice);
    hipEvent_t a, b; hipEventCreate(&a); hipEventCreate(&b);
    const double layer_bytes = (double)kPairsPerLayer * 2 * 1024;
    printf("h = 256 layer, f16x3 operands: %.2f MiB per layer and frame per workgroup; %d workgroups, %d frames\n", layer_bytes / 1048576.0, blocks, frames);
    for (int mode = 0; mode < 3; ++mode) {
        float best = 1e30f;
        for (int rep = 0; rep < 4; ++rep) {
            hipEventRecord(a);
            if (mode == 0) hipLaunchKernelGGL(stream_kernel<0>, dim3(blocks), dim3(256), 0, 0, d_w, d_out, frames);
            if (mode == 1) hipLaunchKernelGGL(stream_kernel<1>, dim3(blocks), dim3(256), 0, 0, d_w, d_out, frames);
            if (mode == 2) hipLaunchKernelGGL(stream_kernel<2>, dim3(blocks), dim3(256), 0, 0, d_w, d_out, frames);
            hipEventRecord(b); hipEventSynchronize(b);
            float ms; hipEventElapsedTime(&ms, a, b);
            if (rep > 0 && ms < best) best = ms;
        }
        const double us_frame = best * 1e3 / frames;
        printf("MODE %d (%s): %.2f us per frame, %.1f TB/s out of the L2s (%.1f B/clk/CU at 2.1 GHz), vs 28.8 us per frame today: %.2fx\n", mode,
               mode == 0 ? "loads only" : mode == 1 ? "loads + 3 MFMAs per operand pair" : "+ a GRU frame's barriers and activations",
               us_frame, blocks * layer_bytes / (us_frame * 1e-6) / 1e12, layer_bytes / (us_frame * 1e-6 * 2.1e9), 28.8 / us_frame);
    }
    return 0;
}

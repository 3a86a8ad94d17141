// vad(data, thres) = sum(|data|) > thres (utils/basic_vad.py:17-18): the one summation every kernel of the library uses
// (kws_vad, the stream manager's gate kernel, the gate fused into the FFT front-end), so that all of them take the same
// decision on the same samples.
#pragma once
#include "kws_internal.h"

namespace kws {

// sum_n |x[n]| of one row by a 256-thread block in fp32 (utils/basic_vad.py:17-18).  ONE summation order for every caller
// and every buffer -- kws_vad and the stream manager's gate kernel must take identical decisions, whatever the alignment of
// the row: thread t owns the groups of four samples t, t + 256, ...; a group is summed (|a|+|b|)+(|c|+|d|) and added to the
// thread's partial sum, then wave shuffles, then the four wave totals.  Aligned rows fetch a group with one 8/16-byte load,
// unaligned rows and the tail group (N % 4 samples, padded with +0) sample by sample: the arithmetic is the same.
template <typename SampleT>
__device__ __forceinline__ float block_abs_sum(const SampleT* __restrict__ x, int N, float* __restrict__ widened) {
    constexpr float kScale = sizeof(SampleT) == 2 ? 1.0f / 32768.0f : 1.0f;      // detector.py:40-43: int16 -> [-1, 1)
    float acc = 0.f;
    const bool vec = (reinterpret_cast<uintptr_t>(x) & (4 * sizeof(SampleT) - 1)) == 0;
    const bool wvec = widened && (reinterpret_cast<uintptr_t>(widened) & 15) == 0;
    const int full = N / 4;
    for (int i = threadIdx.x; i < (N + 3) / 4; i += 256) {
        float v[4] = {0.f, 0.f, 0.f, 0.f};
        if (i < full && vec) {
            if constexpr (sizeof(SampleT) == 2) {
                const short4 q = reinterpret_cast<const short4*>(x)[i];
                v[0] = (float)q.x * kScale; v[1] = (float)q.y * kScale; v[2] = (float)q.z * kScale; v[3] = (float)q.w * kScale;
            } else {
                const float4 q = reinterpret_cast<const float4*>(x)[i];
                v[0] = q.x; v[1] = q.y; v[2] = q.z; v[3] = q.w;
            }
        } else {
#pragma unroll
            for (int e = 0; e < 4; ++e)
                if (4 * i + e < N) v[e] = (float)x[4 * i + e] * kScale;
        }
        if (widened) {
            if (i < full && wvec) {
                reinterpret_cast<float4*>(widened)[i] = make_float4(v[0], v[1], v[2], v[3]);
            } else {
#pragma unroll
                for (int e = 0; e < 4; ++e)
                    if (4 * i + e < N) widened[4 * i + e] = v[e];
            }
        }
        acc += (fabsf(v[0]) + fabsf(v[1])) + (fabsf(v[2]) + fabsf(v[3]));
    }
    for (int off = 32; off > 0; off >>= 1) acc += __shfl_down(acc, off);
    __shared__ float part[4];
    if ((threadIdx.x & 63) == 0) part[threadIdx.x >> 6] = acc;
    __syncthreads();
    return (part[0] + part[1]) + (part[2] + part[3]);         // valid in thread 0
}


// The same sum, the same association of every addition -- hence the same bits -- by ONE wave: lane l plays threads l, 64 + l,
// 128 + l, 192 + l of the block above (four partial sums), each "wave" of them is folded by the same shuffle tree, and the
// four totals meet in the same order.  For kernels that have a wave to spare but no workgroup barrier.  Valid in lane 0.
template <typename SampleT>
__device__ __forceinline__ float wave_abs_sum(const SampleT* __restrict__ x, int N, int lane) {
    constexpr float kScale = sizeof(SampleT) == 2 ? 1.0f / 32768.0f : 1.0f;
    const bool vec = (reinterpret_cast<uintptr_t>(x) & (4 * sizeof(SampleT) - 1)) == 0;
    const int full = N / 4, groups = (N + 3) / 4;
    float acc[4] = {0.f, 0.f, 0.f, 0.f};
    for (int base = 0; base < groups; base += 256) {
        float v[4][4];
#pragma unroll
        for (int t = 0; t < 4; ++t) {
            const int i = base + 64 * t + lane;
#pragma unroll
            for (int e = 0; e < 4; ++e) v[t][e] = 0.f;
            if (i < full && vec) {
                if constexpr (sizeof(SampleT) == 2) {
                    const short4 q = reinterpret_cast<const short4*>(x)[i];
                    v[t][0] = (float)q.x * kScale; v[t][1] = (float)q.y * kScale; v[t][2] = (float)q.z * kScale; v[t][3] = (float)q.w * kScale;
                } else {
                    const float4 q = reinterpret_cast<const float4*>(x)[i];
                    v[t][0] = q.x; v[t][1] = q.y; v[t][2] = q.z; v[t][3] = q.w;
                }
            } else if (i < groups) {
#pragma unroll
                for (int e = 0; e < 4; ++e)
                    if (4 * i + e < N) v[t][e] = (float)x[4 * i + e] * kScale;
            }
        }
#pragma unroll
        for (int t = 0; t < 4; ++t)
            if (base + 64 * t + lane < groups) acc[t] += (fabsf(v[t][0]) + fabsf(v[t][1])) + (fabsf(v[t][2]) + fabsf(v[t][3]));
    }
#pragma unroll
    for (int t = 0; t < 4; ++t)
        for (int off = 32; off > 0; off >>= 1) acc[t] += __shfl_down(acc[t], off);
    return (acc[0] + acc[1]) + (acc[2] + acc[3]);
}

// acc summed over the wave with the association of the `acc += __shfl_down(acc, off)` tree above (off = 32 ... 1) -- valid in lane 0,
// the same bits -- but without LDS: __shfl_down is a ds_bpermute round trip per step; here the halves meet through gfx950's
// v_permlane32_swap / v_permlane16_swap and the last four steps through DPP row shifts.  (fp32 addition commutes, so
// "lane l adds lane l + off" only fixes WHICH pairs meet at each level.)
__device__ __forceinline__ float wave_sum_lane0(float acc) {
    typedef unsigned u32x2 __attribute__((ext_vector_type(2)));
    u32x2 sw = __builtin_amdgcn_permlane32_swap(__float_as_uint(acc), __float_as_uint(acc), false, false);
    acc = __uint_as_float(sw[0]) + __uint_as_float(sw[1]);                 // lanes < 32: a[l] + a[l + 32]
    sw = __builtin_amdgcn_permlane16_swap(__float_as_uint(acc), __float_as_uint(acc), false, false);
    acc = __uint_as_float(sw[0]) + __uint_as_float(sw[1]);                 // lanes < 16: x[l] + x[l + 16]
#define KWS_SHL(ctrl_) acc += __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(acc), ctrl_, 0xf, 0xf, true));
    KWS_SHL(0x108) KWS_SHL(0x104) KWS_SHL(0x102) KWS_SHL(0x101)           // row_shl 8, 4, 2, 1: lane l + lane l + off
#undef KWS_SHL
    return acc;
}

// the masks one loop iteration of detector.py:158-209 consumes, from the chunk's vad sum: silent (-> the decode window is
// cleared before the chunk is added, :171-177) and reset = silent | restart (-> the GRU starts this chunk from zero)
__device__ __forceinline__ void vad_masks(float total, float thres, int b, const uint8_t* __restrict__ restart,
                                          uint8_t* __restrict__ silent, uint8_t* __restrict__ reset) {
    const uint8_t quiet = total > thres ? 0 : 1;
    silent[b] = quiet;
    reset[b] = (quiet || (restart && restart[b])) ? 1 : 0;
}

// next carry = the last n_next samples of [carry | chunk] (detector.py:181-183), by `nthreads` threads (index `thread`)
template <typename SampleT>
__device__ __forceinline__ void carry_tail(const float* __restrict__ carry, int n_carry, const SampleT* __restrict__ chunk, int n_chunk,
                                           float* __restrict__ next, int n_next, int thread = threadIdx.x, int nthreads = 256) {
    constexpr float kScale = sizeof(SampleT) == 2 ? 1.0f / 32768.0f : 1.0f;
    for (int j = thread; j < n_next; j += nthreads) {
        const int i = n_carry + n_chunk - n_next + j;
        next[j] = i < n_carry ? carry[i] : (float)chunk[i - n_carry] * kScale;
    }
}

}  // namespace kws

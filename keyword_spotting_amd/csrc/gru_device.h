// Device helpers shared by the GRU kernels (fp32: gru_kernels.hip, bf16: gru_bf16.hip).
#pragma once
#include <type_traits>

#include "kws_internal.h"

namespace kws {

typedef float f32x4 __attribute__((ext_vector_type(4)));

__device__ __forceinline__ f32x4 mfma4(float a, float b, f32x4 c) {
    return __builtin_amdgcn_mfma_f32_16x16x4f32(a, b, c, 0, 0, 0);
}
// sigma(x) = 1/(1+e^-x).  v_exp_f32 path: abs error <= 3e-7 on [-30,30]; saturates cleanly.
__device__ __forceinline__ float sigmoid_f(float x) {
    return __builtin_amdgcn_rcpf(1.0f + __expf(-x));
}
// tanh(x) = 1 - 2/(1+e^{2x}); abs error <= 3e-7, exact limits +-1.
__device__ __forceinline__ float tanh_f(float x) {
    return 1.0f - 2.0f * __builtin_amdgcn_rcpf(1.0f + __expf(2.0f * x));
}
// MFMA whose A operand (a resident weight fragment) is read straight from the AGPR half of the
// unified register file.  hipcc only ever parks such values in AGPRs as spills and re-reads them with
// v_accvgpr_read + s_nop (39 cycles per MFMA instead of 32, tools/ubench/mfma_issue.hip); the "a"
// constraint removes the copy.  The statement is opaque to the hazard recogniser: every chain of these
// ends with mfma_fence() before any non-MFMA instruction touches the accumulators.
#define KWS_MFMA_A(acc, wa, bv) \
    asm volatile("v_mfma_f32_16x16x4_f32 %0, %1, %2, %0" : "+v"(acc) : "a"(wa), "v"(bv))
// XDL 8-pass write -> VALU read needs 11 wait states (s_nop 15 = 16); the "+v" ties order it after the
// chain and ahead of every consumer
__device__ __forceinline__ void mfma_fence(f32x4& a, f32x4& b, f32x4& c, f32x4& d) {
    asm volatile("s_nop 15" : "+v"(a), "+v"(b), "+v"(c), "+v"(d));
}
__device__ __forceinline__ void mfma_fence(f32x4& a, f32x4& b) {
    asm volatile("s_nop 15" : "+v"(a), "+v"(b));
}
// ... and every chain STARTS with mfma_prefence(): the compiler may have produced the accumulators (or a B
// operand) with a VALU instruction -- typically v_accvgpr_read copies out of the AGPRs the preceding builtin
// MFMAs used -- immediately before the first asm MFMA, and "VALU write VGPR -> MFMA read" needs wait states
// it cannot know about (observed: intermittent 1e-2 errors in the bf16 stack).
__device__ __forceinline__ void mfma_prefence(f32x4& a, f32x4& b, f32x4& c, f32x4& d) {
    asm volatile("s_nop 3" : "+v"(a), "+v"(b), "+v"(c), "+v"(d));
}
__device__ __forceinline__ void mfma_prefence(f32x4& a, f32x4& b) {
    asm volatile("s_nop 3" : "+v"(a), "+v"(b));
}
__device__ __forceinline__ f32x4 splat4(float v) { f32x4 r = {v, v, v, v}; return r; }
// exact, branch-free select: m = all-ones -> a, m = 0 -> b (v_bfi_b32); keeps the h update one
// straight-line block so MFMAs can be scheduled through it
__device__ __forceinline__ float bitsel(unsigned m, float a, float b) {
    return __uint_as_float((__float_as_uint(a) & m) | (__float_as_uint(b) & ~m));
}
template <int I0, int I1, class F>
__device__ __forceinline__ void static_for(F&& f) {
    if constexpr (I0 < I1) {
        f(std::integral_constant<int, I0>{});
        static_for<I0 + 1, I1>(f);
    }
}
constexpr float kLog2e = 1.4426950408889634f;
typedef float f32x2 __attribute__((ext_vector_type(2)));
// Activations on register pairs.  Beside an f32 MFMA stream VALU work is not hidden (f32 MFMA and the
// VALU share the FP32 datapath: tools/ubench/mfma_coissue.hip measures +2..3 cycles per plain op, +8 per
// back-to-back transcendental and +14 for an isolated one), so the ops are clustered, packed
// (v_pk_mul/add/fma_f32) and kept to the minimum count: sigmoid = pk_mul, 2 exp, pk_add, 2 rcp.
__device__ __forceinline__ f32x2 sigmoid2(f32x2 x) {
    const f32x2 t = x * -kLog2e;
    f32x2 e;
    e.x = __builtin_amdgcn_exp2f(t.x);
    e.y = __builtin_amdgcn_exp2f(t.y);
    const f32x2 d = e + 1.0f;
    f32x2 r;
    r.x = __builtin_amdgcn_rcpf(d.x);
    r.y = __builtin_amdgcn_rcpf(d.y);
#ifdef KWS_FAULT_INJECT      // never defined in a product build: tests/test_abi.py builds a deliberately wrong library with it
    r = r * 1.0078125f;      // (tools/build_variant.sh faulty -DKWS_FAULT_INJECT=1) and expects kws_selftest to reject it
#endif
    return r;
}
__device__ __forceinline__ f32x2 tanh2(f32x2 x) {
    const f32x2 t = x * (2.0f * kLog2e);
    f32x2 e;
    e.x = __builtin_amdgcn_exp2f(t.x);
    e.y = __builtin_amdgcn_exp2f(t.y);
    const f32x2 d = e + 1.0f;
    f32x2 r;
    r.x = __builtin_amdgcn_rcpf(d.x);
    r.y = __builtin_amdgcn_rcpf(d.y);
    return r * -2.0f + 1.0f;
}
__device__ __forceinline__ f32x4 ld4(const float* p) { return *reinterpret_cast<const f32x4*>(p); }

// Workgroup barrier that orders LDS traffic only: the inter-layer scratch stores and the x prefetch of the next
// frame, which no other wave of the group ever reads, stay in flight across it.  (hipcc's __syncthreads() is the
// same two instructions on gfx950 -- it does not wait for global memory either; where a store has to be visible
// outside the CU before the barrier, the kernels drain vmcnt(0) explicitly.)
__device__ __forceinline__ void lds_barrier() {
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();
    asm volatile("" ::: "memory");
}
// ------------------------------------------------------------------------------------------------
// Last-layer epilogue.  Each wave leaves the partial logits of its 32 units in `pstage`; after the
// frame's second barrier ONE wave (rotating, w == t & 3) folds the four partials into a 16-frame ring.
// Every kRingFrames frames (and at the end of the call) ALL FOUR waves flush: wave w takes frames
// 4w..4w+3 of the block x 16 streams = 64 items, so the softmax / decode / store work is spread
// over the whole group instead of stalling three waves behind one.
//   pstage [4 waves][16 streams][8]     lring [16 frames][16 streams][8]
//   words  [16 frames][16 streams]      carry [2][16]   (previous block's last word, ping-pong)
// ------------------------------------------------------------------------------------------------
constexpr int kRingFrames = 16;
constexpr int xs_stride(int kcx) { return 4 * ((((kcx + 3) / 4) & 1) ? (kcx + 3) / 4 : (kcx + 3) / 4 + 1); }
struct EpilogueLds {
    float* pstage;
    float* lring;
    int* words;
    int* carry;
    int8_t* cwords;     // kernels instantiated with the window tail: [16 streams][kWinTailWordsStride] frame words of the whole call; else nullptr
};
constexpr size_t kEpilogueLdsBytes = (4 * 16 * 8 + kRingFrames * 16 * 8) * 4 + kRingFrames * 16 * 4 + 2 * 16 * 4;
__device__ __forceinline__ EpilogueLds epilogue_carve(char* base) {
    EpilogueLds e;
    e.pstage = reinterpret_cast<float*>(base);
    e.lring = e.pstage + 4 * 16 * 8;
    e.words = reinterpret_cast<int*>(e.lring + kRingFrames * 16 * 8);
    e.carry = e.words + kRingFrames * 16;
    e.cwords = nullptr;
    return e;
}

// fold the 4 partial logit vectors of frame t into the ring: 32 lanes, (stream, half) each
__device__ __forceinline__ void epilogue_fold(const EpilogueLds& e, int t, int lane) {
    if (lane < 32) {
        const int s = lane >> 1, half = lane & 1;
        f32x4 v = *reinterpret_cast<const f32x4*>(e.pstage + (0 * 16 + s) * 8 + 4 * half);
#pragma unroll
        for (int w = 1; w < 4; ++w) v += *reinterpret_cast<const f32x4*>(e.pstage + (w * 16 + s) * 8 + 4 * half);
        *reinterpret_cast<f32x4*>(e.lring + (((t & (kRingFrames - 1)) * 16 + s) * 8 + 4 * half)) = v;
    }
}

// the same fold in two halves, so the LDS round trip can sit behind independent MFMAs
struct FoldRegs { f32x4 v[4]; };
__device__ __forceinline__ void epilogue_fold_load(const EpilogueLds& e, int lane, FoldRegs& r) {
    if (lane < 32) {
        const int s = lane >> 1, half = lane & 1;
#pragma unroll
        for (int w = 0; w < 4; ++w) r.v[w] = *reinterpret_cast<const f32x4*>(e.pstage + (w * 16 + s) * 8 + 4 * half);
    }
}
__device__ __forceinline__ void epilogue_fold_store(const EpilogueLds& e, int t, int lane, const FoldRegs& r) {
    if (lane < 32) {
        const int s = lane >> 1, half = lane & 1;
        const f32x4 v = ((r.v[0] + r.v[1]) + r.v[2]) + r.v[3];        // same order as epilogue_fold
        *reinterpret_cast<f32x4*>(e.lring + (((t & (kRingFrames - 1)) * 16 + s) * 8 + 4 * half)) = v;
    }
}

// flush frames [t0, t0+n) of the ring; called by all four waves of the group together.
// OPAQUE_LANE: everything the flush derives from the lane index (row pointers, LDS addresses) is computed HERE from an opaque
// copy of it.  Otherwise the compiler hoists those per-lane values out of the frame loop; in gru_layer_f16x3's last-layer
// kernels, which have no register to spare, they get spilled, and every reload (scratch_load + s_waitcnt vmcnt(0)) makes the
// flush wait for the acknowledgement of the stores it has just issued -- several microseconds per call at T = 22.
template <bool OPAQUE_LANE = false>
__device__ __forceinline__ void epilogue_flush(const GruLayerParams& p, const EpilogueLds& e, int group, int t0,
                                               int n, int w, int lane, bool final_flush) {
    if constexpr (OPAQUE_LANE) asm volatile("" : "+v"(lane));
    const int f = 4 * w + (lane & 3);          // frame within the block
    const int s = lane >> 2;
    const int b = group * kStreamsPerGroup + s;
    const int C = p.C;
    float lg[kMaxClasses];
    {
        const f32x4* row = reinterpret_cast<const f32x4*>(e.lring + (f * 16 + s) * 8);
        const f32x4 lo = row[0], hi = row[1];
#pragma unroll
        for (int c = 0; c < 4; ++c) { lg[c] = lo[c]; lg[4 + c] = hi[c]; }
    }
    if (p.use_relu) {
#pragma unroll
        for (int c = 0; c < kMaxClasses; ++c) {
            lg[c] = fmaxf(lg[c], 0.f);
            if (p.value_clip > 0.f) lg[c] = fminf(lg[c], 20.f);
        }
    }
    float m = lg[0];
#pragma unroll
    for (int c = 1; c < kMaxClasses; ++c) m = (c < C) ? fmaxf(m, lg[c]) : m;
    float pr[kMaxClasses];
    float sum = 0.f;
#pragma unroll
    for (int c = 0; c < kMaxClasses; ++c) {
        pr[c] = (c < C) ? __expf(lg[c] - m) : 0.f;     // arguments <= 0: abs error < 1e-7
        sum += pr[c];
    }
    const float inv = __builtin_amdgcn_rcpf(sum);
#pragma unroll
    for (int c = 0; c < kMaxClasses; ++c) pr[c] *= inv;
    // ctc_decode2 frame rule over classes 1..C-2 (utils/prediction.py:67,74-75): first maximum, strict >
    int word = -1;
    float best = -1.f;
#pragma unroll
    for (int c = 1; c < kMaxClasses - 1; ++c) {
        if (c < C - 1 && pr[c] > best) { best = pr[c]; word = c - 1; }
    }
    if (!(best > p.decode_thres)) word = -1;
    e.words[f * 16 + s] = word;
    if (e.cwords != nullptr) {       // the call's words wait for the window tail (row offset from an opaque copy of the lane, as above)
        int ln = lane;
        asm volatile("" : "+v"(ln));
        const int f2 = 4 * w + (ln & 3);
        if (f2 < n) e.cwords[(ln >> 2) * kWinTailWordsStride + t0 + f2] = (int8_t)word;
    }
    const bool mine = b < p.B && f < n;
    const size_t row = (size_t)b * (p.t_stride ? p.t_stride : p.T) + (t0 + f);
    if (mine) {
        if (C == 6) {       // rows are 24 B: three 8-byte stores
            if (p.logits) {
                float2* o = reinterpret_cast<float2*>(p.logits + row * 6);
                o[0] = make_float2(lg[0], lg[1]); o[1] = make_float2(lg[2], lg[3]); o[2] = make_float2(lg[4], lg[5]);
            }
            if (p.softmax) {
                float2* o = reinterpret_cast<float2*>(p.softmax + row * 6);
                o[0] = make_float2(pr[0], pr[1]); o[1] = make_float2(pr[2], pr[3]); o[2] = make_float2(pr[4], pr[5]);
            }
        } else {
            if (p.logits) {
#pragma unroll
                for (int c = 0; c < kMaxClasses; ++c)
                    if (c < C) p.logits[row * C + c] = lg[c];
            }
            if (p.softmax) {
#pragma unroll
                for (int c = 0; c < kMaxClasses; ++c)
                    if (c < C) p.softmax[row * C + c] = pr[c];
            }
        }
    }
    lds_barrier();            // every wave's words are in LDS
    const int blk = (t0 / kRingFrames) & 1;
    const int prev = f == 0 ? e.carry[blk * 16 + s] : e.words[(f - 1) * 16 + s];
    const int token = (word >= 0 && word != prev) ? word + 1 : 0;   // utils/prediction.py:76-80
    if (f == n - 1) {
        // the ping-pong slot is for the NEXT block of this group only: after the final flush nobody reads it, and a persistent
        // workgroup's next group initialises carry[0..15] with no barrier in between (it would race with this store)
        if (!final_flush) e.carry[(blk ^ 1) * 16 + s] = word;
        else if (b < p.B && p.prev_word) p.prev_word[b] = word;
    }
    if (mine && p.tokens) p.tokens[row] = (int8_t)token;
}


}  // namespace kws

// int8 ("octbit") variant of the streaming GRU -- BASELINE configs[2], SURVEY 8a R15-R17.
//
// What the reference's quantised graph computes (octbit/octbit_graph.py:218-225 picks the MatMuls of
// cell_1.. and the class projection; octbit/octbit_mat_mul_op.cc:90-181 is the op): per stream and frame
//     g = sigmoid(OctbitMatMul([x, h];     Wg_q) + bg)        one op call on a [1, 2H] row
//     c = tanh   (OctbitMatMul([x, r (.) h]; Wc_q) + bc)       another call, its own activation range
//     logits = OctbitMatMul(h'[T, H]; Wfc_q) + bfc             one call per sess.run: ONE range over all T frames
// where the op quantises its input to u8 with the call's min/max, multiplies u8 x s8 with
// _mm_maddubs_epi16 -- adjacent-pair sums SATURATED to int16 -- and rescales.  The saturation is part of
// the reference's results (1.8 % of the pairs clip with glorot-uniform weights and they dominate the
// int8-vs-fp32 logit error), so it is reproduced exactly; that rules out v_mfma_i32_*_i8, which
// accumulates exactly.  The pair arithmetic runs on the packed-int16 VALU instead, 1.5 instructions per pair:
//     v_pk_mul_lo_u16  p, A_even, W_even          two products a[k]*w[k]        (|a*w| <= 254*127 fits int16)
//     v_pk_mad_i16     s, A_odd, W_odd, p clamp   two saturated pair sums       (= maddubs)
//     v_dot2_i32_i16   acc, s, (1,1), acc         widen + accumulate            (= the i32 lanes; exact)
//
// Mapping.  16 streams per workgroup (B = 4096 fills the 256 CUs exactly as the fp32 kernels do), 8 waves.
// lane = output unit: each lane keeps the int16-expanded weights of ITS unit in VGPRs for the whole launch
// (two units x K-quarter = 64 dwords for the gates, two units x K-eighth = 32 for the candidate), so the
// activation operand is the same for all 64 lanes -- a wave-uniform value, i.e. an SGPR operand.  Two units per
// lane give 96 VALU instructions per fetched 32-dword batch: scalar loads return out of order, only lgkmcnt(0) is
// usable, so the prefetch is one batch deep and a batch has to outlast the fetch.  (Measured: one unit per lane
// with half the work per batch ran at the same speed -- the dot phases are VALU-issue-bound at ~80 % of the
// 4-cycles-per-instruction ideal, not latency-bound.)  The quantised activations of the
// group (16 x 512 B per matmul) are written to a small global exchange buffer and fetched back with
// s_load_dwordx16 (scalar cache, invalidated after each exchange); LDS cannot feed a uniform operand
// without paying a full 64-lane return per read, which would make LDS the bound by 2.7x.
#include "gru_device.h"

namespace kws {
namespace {

typedef uint32_t u32x16 __attribute__((ext_vector_type(16)));

constexpr int kHS = 132;   // LDS row stride (floats) of the [16 streams][128] fp32 blocks

// two s_load_dwordx16 = one 32-dword batch = 16 couples (32 pairs) of one stream's quantised row
__device__ __forceinline__ void sload32(const uint32_t* p, u32x16& a, u32x16& b) {
    asm volatile("s_load_dwordx16 %0, %2, 0x0\n\ts_load_dwordx16 %1, %2, 0x40" : "=&s"(a), "=&s"(b) : "s"(p) : "memory");
}
// SMEM returns out of order: only lgkmcnt(0) is meaningful
__device__ __forceinline__ void swait(u32x16& a, u32x16& b) {
    asm volatile("s_waitcnt lgkmcnt(0)" : "+s"(a), "+s"(b) : : "memory");
}
__device__ __forceinline__ void scalar_cache_invalidate() {
    asm volatile("s_dcache_inv\n\ts_waitcnt lgkmcnt(0)" ::: "memory");
}

// pair arithmetic: acc += sat16(ae.lo*we.lo + ao.lo*wo.lo) + sat16(ae.hi*we.hi + ao.hi*wo.hi)
// same with per-lane activations (class projection)
__device__ __forceinline__ int pair2_v(int acc, uint32_t ae, uint32_t ao, uint32_t we, uint32_t wo, uint32_t ones) {
    uint32_t p, s;
    asm("v_pk_mul_lo_u16 %0, %1, %2" : "=v"(p) : "v"(ae), "v"(we));
    asm("v_pk_mad_i16 %0, %1, %2, %3 clamp" : "=v"(s) : "v"(ao), "v"(wo), "v"(p));
    asm("v_dot2_i32_i16 %0, %1, %2, %3" : "=v"(acc) : "v"(s), "v"(ones), "v"(acc));
    return acc;
}

// four couples in one statement: the compiler separates consecutive asm statements with an s_nop
__device__ __forceinline__ int pair8_s(int acc, uint32_t a0, uint32_t a1, uint32_t a2, uint32_t a3, uint32_t a4, uint32_t a5,
                                       uint32_t a6, uint32_t a7, uint32_t w0, uint32_t w1, uint32_t w2, uint32_t w3,
                                       uint32_t w4, uint32_t w5, uint32_t w6, uint32_t w7, uint32_t ones) {
    uint32_t t;
    asm volatile(
        "v_pk_mul_lo_u16 %1, %2, %10\n\tv_pk_mad_i16 %1, %3, %11, %1 clamp\n\tv_dot2_i32_i16 %0, %1, %18, %0\n\t"
        "v_pk_mul_lo_u16 %1, %4, %12\n\tv_pk_mad_i16 %1, %5, %13, %1 clamp\n\tv_dot2_i32_i16 %0, %1, %18, %0\n\t"
        "v_pk_mul_lo_u16 %1, %6, %14\n\tv_pk_mad_i16 %1, %7, %15, %1 clamp\n\tv_dot2_i32_i16 %0, %1, %18, %0\n\t"
        "v_pk_mul_lo_u16 %1, %8, %16\n\tv_pk_mad_i16 %1, %9, %17, %1 clamp\n\tv_dot2_i32_i16 %0, %1, %18, %0"
        : "+v"(acc), "=&v"(t)
        : "s"(a0), "s"(a1), "s"(a2), "s"(a3), "s"(a4), "s"(a5), "s"(a6), "s"(a7),
          "v"(w0), "v"(w1), "v"(w2), "v"(w3), "v"(w4), "v"(w5), "v"(w6), "v"(w7), "v"(ones));
    return acc;
}

// one s_load_dwordx16 worth of activations (8 couples) against the 16 weight dwords W[OFF..OFF+15] of one unit
template <int OFF>
__device__ __forceinline__ int dot8(int acc, const u32x16& v, const uint32_t* W, uint32_t ones) {
#pragma unroll
    for (int c = 0; c < 2; ++c)
        acc = pair8_s(acc, v[8 * c], v[8 * c + 1], v[8 * c + 2], v[8 * c + 3], v[8 * c + 4], v[8 * c + 5], v[8 * c + 6],
                      v[8 * c + 7], W[OFF + 8 * c], W[OFF + 8 * c + 1], W[OFF + 8 * c + 2], W[OFF + 8 * c + 3],
                      W[OFF + 8 * c + 4], W[OFF + 8 * c + 5], W[OFF + 8 * c + 6], W[OFF + 8 * c + 7], ones);
    return acc;
}
// two s_load_dwordx16 from two rows
__device__ __forceinline__ void sload16x2(const uint32_t* p0, const uint32_t* p1, u32x16& a, u32x16& b) {
    asm volatile("s_load_dwordx16 %0, %2, 0x0\n\ts_load_dwordx16 %1, %3, 0x0" : "=&s"(a), "=&s"(b) : "s"(p0), "s"(p1) : "memory");
}

// Separately rounded fp32 ops, as the reference's graph executes them (OctbitMatMul output, then BiasAdd, ...).
// HIP's __fmul_rn/__fadd_rn are plain operators and contract into FMAs under -ffp-contract=fast-honor-pragmas.
__device__ __forceinline__ float mul_rn(float a, float b) {
#pragma clang fp contract(off)
    return a * b;
}
__device__ __forceinline__ float add_rn(float a, float b) {
#pragma clang fp contract(off)
    return a + b;
}
__device__ __forceinline__ float sub_rn(float a, float b) {
#pragma clang fp contract(off)
    return a - b;
}

// octbit_mat_mul_op.cc:105-124 on one value; `off` = 127 (signed branch) or 0.  (A reciprocal-multiply fast
// path with an exact fallback near half-integers was measured: no gain, the division is not what the stage costs.)
__device__ __forceinline__ uint32_t quant_u8(float v, float bscale, float off) {
#ifdef KWS_OABL_NODIV
    const float r = roundf(v * bscale);
#else
    const float r = roundf(v / bscale);                 // C round(): half away from zero, on the float quotient
#endif
    return bscale == 0.f ? 0u : (uint32_t)(int)(r + off) & 0xffu;
}

// min / max over the 32-lane half a stream's row is spread over, without LDS traffic: four DPP rotations inside each
// row of 16, then gfx950's v_permlane16_swap pairs row 0 with row 1 (and 2 with 3).  (__shfl_xor compiles to
// ds_bpermute: ten LDS round trips per range.)
template <bool MAX>
__device__ __forceinline__ float half_wave_reduce(float v) {
#define KWS_ROR(ctrl_)                                                                                     \
    {                                                                                                      \
        const float o = __int_as_float(__builtin_amdgcn_update_dpp(__float_as_int(v), __float_as_int(v), ctrl_, 0xf, 0xf, false)); \
        v = MAX ? fmaxf(v, o) : fminf(v, o);                                                               \
    }
    KWS_ROR(0x128) KWS_ROR(0x124) KWS_ROR(0x122) KWS_ROR(0x121)
#undef KWS_ROR
    typedef unsigned u32x2 __attribute__((ext_vector_type(2)));
    const u32x2 sw = __builtin_amdgcn_permlane16_swap(__float_as_uint(v), __float_as_uint(v), false, false);
    const float a = __uint_as_float(sw[0]), b = __uint_as_float(sw[1]);
    return MAX ? fmaxf(a, b) : fminf(a, b);
}

// range of the call (:92-99) -> (bscale, signed); an all-zero call has bscale 0 (output defined as 0)
__device__ __forceinline__ void range_to_scale(float mn, float mx, float& bscale, int& is_signed) {
    is_signed = mn < 0.f;
    bscale = is_signed ? fmaxf(-mn, mx) / 127.0f : mx / 254.0f;
}

}  // namespace

__global__ void __launch_bounds__(512) gru_layer_octbit_kernel(const GruOctbitParams p) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    float* hs = reinterpret_cast<float*>(smem);          // [16][kHS] recurrent state, fp32
    float* rh = hs + 16 * kHS;                           // r (.) h
    float* ub = rh + 16 * kHS;                           // u
    int* part = reinterpret_cast<int*>(ub + 16 * kHS);   // partial sums: gates [4][16][256], candidate [8][16][128]
    float* bsc = reinterpret_cast<float*>(part + 16384); // [2 matmuls][16]
    int* sgn = reinterpret_cast<int*>(bsc + 32);         // [2][16]
    int* slen = sgn + 32;                                // [16]

    const int tid = threadIdx.x, lane = tid & 63;
    const int w = __builtin_amdgcn_readfirstlane(tid >> 6);   // wave-uniform by construction: keeps SMEM addresses scalar
    const int G = blockIdx.x, T = p.T;
    const uint32_t ones = 0x00010001u;

    // ---- resident weights: this lane's unit, int16-expanded (we,wo) couples -----------------------------
    const int ug = w & 1, kq = w >> 1;       // gates: units 128ug + {lane, 64 + lane}, K-quarter kq
    const int k8 = w;                        // candidate: units {lane, 64 + lane}, K-eighth k8
    uint32_t WA[64], WB[32];                 // [unit-in-lane][couples x (even, odd)]
#pragma unroll
    for (int c = 0; c < 64; ++c) WA[c] = p.wg[((size_t)(kq * 2 + ug) * 64 + c) * 64 + lane];
#pragma unroll
    for (int c = 0; c < 32; ++c) WB[c] = p.wc[((size_t)k8 * 32 + c) * 64 + lane];

    // finalize mappings
    const int nA = tid & 255, sA0 = 8 * (tid >> 8);
    const int nB = tid & 127, sB0 = 4 * (tid >> 7);
    const float bgA = p.bias[nA], b127A = p.b127[nA];
    const float bcB = p.bias[256 + nB], b127B = p.b127[256 + nB];

    // ---- state in --------------------------------------------------------------------------------------
    for (int i = tid; i < 16 * 128; i += 512) {
        const int s = i >> 7, n = i & 127;
        const int b = G * 16 + s;
        const bool ok = b < p.B && !(p.reset && p.reset[b]);
        hs[s * kHS + n] = ok ? p.state_in[(size_t)b * 128 + n] : 0.f;
    }
    if (tid < 16) {
        const int b = G * 16 + tid;
        slen[tid] = b < p.B ? (p.seq_len ? p.seq_len[b] : T) : 0;
    }

    // quantise mapping: stream qs, eight consecutive k = 8qi..8qi+7 of the row [x (128) | h (128)]
    const int qs = tid >> 5, qi = tid & 31;
    f32x4 xa = splat4(0.f), xb = splat4(0.f);
    auto load_x = [&](int t) {
        if (qi < 16) {
            const size_t idx = (((size_t)G * T + t) * 8 + (qi >> 1)) * 64 + 32 * (qi & 1) + qs;
            xa = *reinterpret_cast<const f32x4*>(p.x_prev + idx);
            xb = *reinterpret_cast<const f32x4*>(p.x_prev + idx + 16);
        }
    };
    auto quantise = [&](int which) {
        f32x4 va = xa, vb = xb;
        if (qi >= 16) {
            const float* src = (which ? rh : hs) + qs * kHS + 8 * (qi - 16);
            va = ld4(src);
            vb = ld4(src + 4);
        }
        float mn = fminf(fminf(fminf(va[0], va[1]), fminf(va[2], va[3])), fminf(fminf(vb[0], vb[1]), fminf(vb[2], vb[3])));
        float mx = fmaxf(fmaxf(fmaxf(va[0], va[1]), fmaxf(va[2], va[3])), fmaxf(fmaxf(vb[0], vb[1]), fmaxf(vb[2], vb[3])));
        mn = half_wave_reduce<false>(mn);
        mx = half_wave_reduce<true>(mx);
        float bscale;
        int is_signed;
        range_to_scale(mn, mx, bscale, is_signed);
        if (qi == 0) { bsc[which * 16 + qs] = bscale; sgn[which * 16 + qs] = is_signed; }
        const float off = is_signed ? 127.0f : 0.0f;
        uint32_t q[8];
#pragma unroll
        for (int j = 0; j < 4; ++j) { q[j] = quant_u8(va[j], bscale, off); q[4 + j] = quant_u8(vb[j], bscale, off); }
        uint4 o;
        o.x = q[0] | (q[2] << 16); o.y = q[1] | (q[3] << 16);      // couple 2qi:   (A_even, A_odd)
        o.z = q[4] | (q[6] << 16); o.w = q[5] | (q[7] << 16);      // couple 2qi+1
        *reinterpret_cast<uint4*>(p.aq + (((size_t)G * 2 + which) * 16 + qs) * 128 + 4 * qi) = o;
    };
    // previous frame's output row -> xl scratch (zero row for finished frames, dynamic_rnn)
    const int ss = tid & 15, n4 = tid >> 4;
    auto store_out = [&](int t) {
        f32x4 v = ld4(hs + ss * kHS + 4 * n4);
        if (!(t < slen[ss])) v = splat4(0.f);
        *reinterpret_cast<f32x4*>(p.h_out + (((size_t)G * T + t) * 8 + (n4 >> 2)) * 64 + 16 * (n4 & 3) + ss) = v;
    };

    // range of the emitted rows per stream (the class projection's activation range, :92-99), tracked where h' is made
    float rmin[4], rmax[4];
#pragma unroll
    for (int j = 0; j < 4; ++j) { rmin[j] = 3.402823466e+38f; rmax[j] = -3.402823466e+38f; }

    load_x(0);
    __syncthreads();
#define OCT_TS(i) do {} while (0)

    for (int t = 0; t < T; ++t) {
        if (t > 0) store_out(t - 1);
        // ---- gates: quantise [x, h], exchange, dot, finalize ----------------------------------------
        quantise(0);
        OCT_TS(0);
        // __syncthreads() alone does NOT wait for global stores (workgroup scope only orders them inside the CU's L1);
        // the scalar loads below go through the scalar cache to L2, so the stores must have been acknowledged by L2
        // first.  (Without the explicit vmcnt(0): rare stale rows, found by tools/stress_determinism.py.)
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __syncthreads();
        scalar_cache_invalidate();
        OCT_TS(1);
        {
            const uint32_t* base = p.aq + ((size_t)G * 2 + 0) * 16 * 128 + 32 * kq;
            int* dst = part + (kq * 16) * 256 + 128 * ug + lane;
            u32x16 a0, a1, b0, b1;
            sload32(base, a0, a1);
            for (int s = 0; s < 16; s += 2) {
                swait(a0, a1);
                sload32(base + (s + 1) * 128, b0, b1);
#ifdef KWS_OABL_NODOT
                dst[s * 256] = a0[0] + WA[0]; dst[s * 256 + 64] = a1[0] + WA[32];
#else
                dst[s * 256] = dot8<16>(dot8<0>(0, a0, WA, ones), a1, WA, ones);
                dst[s * 256 + 64] = dot8<48>(dot8<32>(0, a0, WA, ones), a1, WA, ones);
#endif
                swait(b0, b1);
                sload32(base + (s < 14 ? s + 2 : 15) * 128, a0, a1);
#ifdef KWS_OABL_NODOT
                dst[(s + 1) * 256] = b0[0] + WA[1]; dst[(s + 1) * 256 + 64] = b1[0] + WA[33];
#else
                dst[(s + 1) * 256] = dot8<16>(dot8<0>(0, b0, WA, ones), b1, WA, ones);
                dst[(s + 1) * 256 + 64] = dot8<48>(dot8<32>(0, b0, WA, ones), b1, WA, ones);
#endif
            }
            swait(a0, a1);
        }
        OCT_TS(2);
        lds_barrier();
        OCT_TS(3);
#pragma unroll
        for (int j = 0; j < 8; ++j) {
            const int s = sA0 + j;
            const int o = (part[(0 * 16 + s) * 256 + nA] + part[(1 * 16 + s) * 256 + nA]) +
                          (part[(2 * 16 + s) * 256 + nA] + part[(3 * 16 + s) * 256 + nA]);
            float of = (float)o;
            if (sgn[s]) of = sub_rn(of, b127A);                                        // :176 signed correction
            const float sc = mul_rn(p.scale_g, bsc[s]);                  // :108 scale *= bscale
            const float pre = add_rn(mul_rn(of, sc), bgA);            // op output, then BiasAdd
            const float a = sigmoid_f(pre);
            if (nA < 128) rh[s * kHS + nA] = mul_rn(a, hs[s * kHS + nA]);
            else ub[s * kHS + nA - 128] = a;
        }
        OCT_TS(4);
        lds_barrier();
        OCT_TS(5);
        // ---- candidate: quantise [x, r (.) h], exchange, dot, finalize + state update ---------------
        quantise(1);
        OCT_TS(6);
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");   // exchange rows in L2 before any wave's scalar loads
        __syncthreads();
        if (t + 1 < T) load_x(t + 1);                      // x of this frame is consumed; issued after the drain so it stays in flight
        scalar_cache_invalidate();
        OCT_TS(7);
        {
            const uint32_t* base = p.aq + ((size_t)G * 2 + 1) * 16 * 128 + 16 * k8;   // one batch = two streams x 16 dwords
            int* dst = part + (k8 * 16) * 128 + lane;
            u32x16 a0, a1, b0, b1;
            sload16x2(base, base + 128, a0, a1);
            for (int s = 0; s < 16; s += 4) {
                swait(a0, a1);
                sload16x2(base + (s + 2) * 128, base + (s + 3) * 128, b0, b1);
#ifdef KWS_OABL_NODOT
                dst[s * 128] = a0[0] + WB[0]; dst[s * 128 + 64] = a0[1] + WB[16]; dst[(s + 1) * 128] = a1[0]; dst[(s + 1) * 128 + 64] = a1[1];
#else
                dst[s * 128] = dot8<0>(0, a0, WB, ones);
                dst[s * 128 + 64] = dot8<16>(0, a0, WB, ones);
                dst[(s + 1) * 128] = dot8<0>(0, a1, WB, ones);
                dst[(s + 1) * 128 + 64] = dot8<16>(0, a1, WB, ones);
#endif
                swait(b0, b1);
                { const int sn = s < 12 ? s + 4 : 14; sload16x2(base + sn * 128, base + (sn + 1) * 128, a0, a1); }
#ifdef KWS_OABL_NODOT
                dst[(s + 2) * 128] = b0[0]; dst[(s + 2) * 128 + 64] = b0[1]; dst[(s + 3) * 128] = b1[0]; dst[(s + 3) * 128 + 64] = b1[1];
#else
                dst[(s + 2) * 128] = dot8<0>(0, b0, WB, ones);
                dst[(s + 2) * 128 + 64] = dot8<16>(0, b0, WB, ones);
                dst[(s + 3) * 128] = dot8<0>(0, b1, WB, ones);
                dst[(s + 3) * 128 + 64] = dot8<16>(0, b1, WB, ones);
#endif
            }
            swait(a0, a1);
        }
        OCT_TS(8);
        lds_barrier();
        OCT_TS(9);
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            const int s = sB0 + j;
            int o = 0;
#pragma unroll
            for (int k = 0; k < 8; ++k) o += part[(k * 16 + s) * 128 + nB];
            float of = (float)o;
            if (sgn[16 + s]) of = sub_rn(of, b127B);
            const float sc = mul_rn(p.scale_c, bsc[16 + s]);
            const float cand = tanh_f(add_rn(mul_rn(of, sc), bcB));
            const float u = ub[s * kHS + nB], hp = hs[s * kHS + nB];
            const float hn = add_rn(mul_rn(u, hp), mul_rn(sub_rn(1.0f, u), cand));
            if (t < slen[s]) {
                hs[s * kHS + nB] = hn;
                rmin[j] = fminf(rmin[j], hn);
                rmax[j] = fmaxf(rmax[j], hn);
            }
        }
        OCT_TS(10);
        lds_barrier();
        OCT_TS(11);
    }
    store_out(T - 1);
    for (int i = tid; i < 16 * 128; i += 512) {
        const int s = i >> 7, n = i & 127;
        const int b = G * 16 + s;
        if (b < p.B) p.state_out[(size_t)b * 128 + n] = hs[s * kHS + n];
    }
    if (p.range) {
        // fold over the 128 units of each stream: stream sB0 + j is owned by the 128 threads with the same tid >> 7
        float* red = reinterpret_cast<float*>(part);            // [16 streams][128][2]
        __syncthreads();
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            red[((sB0 + j) * 128 + nB) * 2 + 0] = rmin[j];
            red[((sB0 + j) * 128 + nB) * 2 + 1] = rmax[j];
        }
        __syncthreads();
        if (tid < 16) {
            float mn = 3.402823466e+38f, mx = -3.402823466e+38f;
            for (int n = 0; n < 128; ++n) { mn = fminf(mn, red[(tid * 128 + n) * 2]); mx = fmaxf(mx, red[(tid * 128 + n) * 2 + 1]); }
            if (slen[tid] < T) { mn = fminf(mn, 0.f); mx = fmaxf(mx, 0.f); }      // finished frames emit the zero row
            if (slen[tid] <= 0 && T > 0) { mn = 0.f; mx = 0.f; }
            p.range[G * 16 + tid] = make_float2(mn, mx);
        }
    }
}

// per-stream range of the call's [T, H] top-layer block (the projection's one op call, :92-99)
__global__ void __launch_bounds__(512) octbit_top_range_kernel(const float4* __restrict__ h_top, int T, float2* __restrict__ range) {
    __shared__ float rmn[8][16], rmx[8][16];
    const int tid = threadIdx.x, n = tid >> 6, lane = tid & 63, G = blockIdx.x;
    float mn = 3.402823466e+38f, mx = -3.402823466e+38f;
    const float4* src = h_top + ((size_t)G * T * 8 + n) * 64 + lane;
    for (int t = 0; t < T; ++t) {
        const float4 v = src[(size_t)t * 8 * 64];
        mn = fminf(mn, fminf(fminf(v.x, v.y), fminf(v.z, v.w)));
        mx = fmaxf(mx, fmaxf(fmaxf(v.x, v.y), fmaxf(v.z, v.w)));
    }
    mn = fminf(mn, __shfl_xor(mn, 16)); mx = fmaxf(mx, __shfl_xor(mx, 16));
    mn = fminf(mn, __shfl_xor(mn, 32)); mx = fmaxf(mx, __shfl_xor(mx, 32));
    if (lane < 16) { rmn[n][lane] = mn; rmx[n][lane] = mx; }
    __syncthreads();
    if (tid < 16) {
#pragma unroll
        for (int k = 1; k < 8; ++k) { mn = fminf(mn, rmn[k][tid]); mx = fmaxf(mx, rmx[k][tid]); }
        range[G * 16 + tid] = make_float2(mn, mx);
    }
}

// class projection + relu/clip + softmax + ctc_decode2 frame rule for kFcFrames frames of one 16-stream group
constexpr int kFcFrames = 32;
__global__ void __launch_bounds__(512) octbit_fc_kernel(const OctbitFcParams p) {
    __shared__ int iacc[kFcFrames + 1][16][8];
    __shared__ int words[kFcFrames + 1][16];
    const int tid = threadIdx.x, n = tid >> 6, lane = tid & 63, g = lane >> 4, s = lane & 15;
    const int G = blockIdx.x, t0 = blockIdx.y * kFcFrames, T = p.T, C = p.C;
    const int b = G * 16 + s;
    const uint32_t ones = 0x00010001u;
    for (int i = tid; i < (kFcFrames + 1) * 16 * 8; i += 512) (&iacc[0][0][0])[i] = 0;
    uint32_t W[2 * kMaxClasses];
#pragma unroll
    for (int c = 0; c < 2 * kMaxClasses; ++c) W[c] = p.wfc[(n * 4 + g) * 2 * kMaxClasses + c];
    float bscale;
    int is_signed;
    {
        const float2 r = p.range[b];
        range_to_scale(r.x, r.y, bscale, is_signed);
    }
    const float off = is_signed ? 127.0f : 0.0f;
    __syncthreads();
    // frame slot f <-> frame t0 - 1 + f; slot 0 is the halo that supplies the previous word
    for (int f = (t0 == 0 ? 1 : 0); f <= kFcFrames; ++f) {
        const int t = t0 - 1 + f;
        if (t >= T) break;
        const float4 v = p.h_top[(((size_t)G * T + t) * 8 + n) * 64 + lane];
        const uint32_t q0 = quant_u8(v.x, bscale, off), q1 = quant_u8(v.y, bscale, off);
        const uint32_t q2 = quant_u8(v.z, bscale, off), q3 = quant_u8(v.w, bscale, off);
        const uint32_t ae = q0 | (q2 << 16), ao = q1 | (q3 << 16);
        int acc[kMaxClasses];
#pragma unroll
        for (int c = 0; c < kMaxClasses; ++c) {
            if (c < C) {
                acc[c] = pair2_v(0, ae, ao, W[2 * c], W[2 * c + 1], ones);
                acc[c] += __shfl_xor(acc[c], 16);
                acc[c] += __shfl_xor(acc[c], 32);
            }
        }
        if (lane < 16) {
#pragma unroll
            for (int c = 0; c < kMaxClasses; ++c)
                if (c < C) atomicAdd(&iacc[f][s][c], acc[c]);
        }
    }
    __syncthreads();
    for (int item = tid; item < (kFcFrames + 1) * 16; item += 512) {
        const int f = item >> 4, si = item & 15;
        const int t = t0 - 1 + f, bi = G * 16 + si;
        if (t < 0 || t >= T) { words[f][si] = -1; continue; }
        float bs;
        int sg;
        {
            const float2 r = p.range[bi];
            range_to_scale(r.x, r.y, bs, sg);
        }
        const float sc = mul_rn(p.scale_w, bs);
        float lg[kMaxClasses], pr[kMaxClasses];
#pragma unroll
        for (int c = 0; c < kMaxClasses; ++c) {
            float of = (float)iacc[f][si][c];
            if (sg) of = sub_rn(of, p.b127[c]);
            lg[c] = c < C ? add_rn(mul_rn(of, sc), p.bfc[c]) : 0.f;
            if (p.use_relu) {
                lg[c] = fmaxf(lg[c], 0.f);
                if (p.value_clip > 0.f) lg[c] = fminf(lg[c], 20.f);
            }
        }
        float m = lg[0];
#pragma unroll
        for (int c = 1; c < kMaxClasses; ++c) m = (c < C) ? fmaxf(m, lg[c]) : m;
        float sum = 0.f;
#pragma unroll
        for (int c = 0; c < kMaxClasses; ++c) { pr[c] = (c < C) ? __expf(lg[c] - m) : 0.f; sum += pr[c]; }
        const float inv = __builtin_amdgcn_rcpf(sum);
        int word = -1;
        float best = -1.f;
#pragma unroll
        for (int c = 0; c < kMaxClasses; ++c) {
            pr[c] *= inv;
            if (c >= 1 && c < C - 1 && pr[c] > best) { best = pr[c]; word = c - 1; }   // utils/prediction.py:67,74-75
        }
        if (!(best > p.decode_thres)) word = -1;
        words[f][si] = word;
        if (f >= 1 && bi < p.B) {
            const size_t row = (size_t)bi * T + t;
            for (int c = 0; c < C; ++c) {
                if (p.logits) p.logits[row * C + c] = lg[c];
                if (p.softmax) p.softmax[row * C + c] = pr[c];
            }
        }
    }
    __syncthreads();
    for (int item = tid; item < kFcFrames * 16; item += 512) {
        const int f = 1 + (item >> 4), si = item & 15;
        const int t = t0 - 1 + f, bi = G * 16 + si;
        if (t >= T || bi >= p.B) continue;
        const int word = words[f][si];
        const int prev = t == 0 ? (p.prev_in ? p.prev_in[bi] : -1) : words[f - 1][si];
        if (p.tokens) p.tokens[(size_t)bi * T + t] = (int8_t)((word >= 0 && word != prev) ? word + 1 : 0);   // :76-80
        if (t == T - 1 && p.prev_word) p.prev_word[bi] = word;
    }
}

size_t gru_octbit_lds_bytes() { return (size_t)(3 * 16 * kHS + 16384 + 32 + 32 + 16) * 4; }

hipError_t launch_gru_layer_octbit(const GruOctbitParams& p, hipStream_t st) {
    const int groups = (p.B + 15) / 16;
    static LdsGrant granted;
    {
        const hipError_t e = grant_dynamic_lds(gru_layer_octbit_kernel, granted, gru_octbit_lds_bytes());
        if (e != hipSuccess) return e;
    }
    hipLaunchKernelGGL(gru_layer_octbit_kernel, dim3(groups), dim3(512), gru_octbit_lds_bytes(), st, p);
    return hipGetLastError();
}

hipError_t launch_octbit_fc(const OctbitFcParams& p, hipStream_t st) {
    const int groups = (p.B + 15) / 16;
    if (!p.range_ready)          // top layer ran on the fp32 kernels (one-layer models): scan its rows
        hipLaunchKernelGGL(octbit_top_range_kernel, dim3(groups), dim3(512), 0, st, p.h_top, p.T, p.range);
    hipLaunchKernelGGL(octbit_fc_kernel, dim3(groups, (p.T + kFcFrames - 1) / kFcFrames), dim3(512), 0, st, p);
    return hipGetLastError();
}

}  // namespace kws

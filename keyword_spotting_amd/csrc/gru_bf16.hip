// bf16 variant of the streaming GRU stack (BASELINE.json configs[2]): bf16 weights and bf16 matmul inputs,
// fp32 accumulation, fp32 recurrent state, fp32 activations / logits.  Same semantics as gru_kernels.hip
// (models/rnn_ctc.py:155-165,202-284), same boundary; only the operand precision differs.
//
// Mapping: v_mfma_f32_16x16x32_bf16, D[unit][stream] as in the fp32 kernels (A = weights, B = activations,
// 16 streams per workgroup, wave w owns units [32w, 32w+32)).  One k-chunk is 32 inputs; lane (g, s/i) holds
// 8 consecutive k of it.  K is permuted so that chunk m of a hidden vector is exactly what wave m produces:
//   lane (g,s), j = 0..3 -> unit 32m + 4g + j ; j = 4..7 -> unit 32m + 16 + 4g + (j-4)
// i.e. the wave's two fp32 C tiles, rounded to bf16 and packed, ARE the B operand of chunk m: the
// exchange is one ds_write_b128 per wave and four ds_read_b128 per consumer, no transpose.
//
// At bf16 MFMA rates (16 cycles per 16x16x32) both layers' weights fit the register file (84 operands x 4
// VGPRs = 336 of 512) and the matrix work of a frame is ~1.4k cycles, so the whole stack runs FUSED in one
// launch with no inter-layer HBM scratch; activations, LDS exchange and barriers dominate.
#include <cstdlib>

#include <cstddef>

#include "gru_device.h"
#include "window_device.h"

namespace kws {

typedef short bf16x8 __attribute__((ext_vector_type(8)));
typedef unsigned u32x4 __attribute__((ext_vector_type(4)));

__device__ __forceinline__ f32x4 mfma_bf16(bf16x8 a, bf16x8 b, f32x4 c) {
    return __builtin_amdgcn_mfma_f32_16x16x32_bf16(a, b, c, 0, 0, 0);
}
// A operand straight from AGPRs (see KWS_MFMA_A in gru_device.h)
#define KWS_MFMA_BF16_A(acc, wa, bv) \
    asm volatile("v_mfma_f32_16x16x32_bf16 %0, %1, %2, %0" : "+v"(acc) : "a"(wa), "v"(bv))

// two fp32 -> packed bf16 pair, round to nearest even (v_cvt_pk_bf16_f32 has no builtin)
__device__ __forceinline__ unsigned pack_bf16(float lo, float hi) {
    unsigned r;
    asm("v_cvt_pk_bf16_f32 %0, %1, %2" : "=v"(r) : "v"(lo), "v"(hi));
    return r;
}
__device__ __forceinline__ bf16x8 as_bf16x8(u32x4 v) { return __builtin_bit_cast(bf16x8, v); }

template <int KX0, int NL>
__global__ void __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(1, 1)))
gru_stack_bf16(const GruBf16Params p) {
    constexpr int H = 128;
    const int tid = threadIdx.x;
    const int w = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int lane = tid & 63, g = lane >> 4, s = lane & 15;
    const int group = blockIdx.x;
    const int b_raw = group * kStreamsPerGroup + s;
    const bool bvalid = b_raw < p.B;
    const int b = bvalid ? b_raw : p.B - 1;
    const int T = p.T;

    extern __shared__ __attribute__((aligned(16))) char smem[];
    u32x4* hb = reinterpret_cast<u32x4*>(smem);             // [NL][4 chunks][64]  h_{t-1} (bf16)
    u32x4* rhb = hb + NL * 4 * 64;                           // [NL][4][64]         r (.) h_{t-1}
    u32x4* xsb = rhb + NL * 4 * 64;                          // [KX0][64]           mel frame (bf16), zero padded
    float* biasl = reinterpret_cast<float*>(xsb + KX0 * 64); // [NL][3][128]
    const EpilogueLds epi = epilogue_carve(reinterpret_cast<char*>(biasl + NL * 3 * H));

    // ---- weights: layer 1 (48 operands) + the first 8 of layer 0 in AGPRs, the rest in VGPRs -------------
    constexpr int KC0 = KX0 + 4, KC1 = 8;
    bf16x8 w0[2][3][KC0];        // [tile][gate][chunk]
    bf16x8 w1[NL > 1 ? 2 : 1][3][KC1];
#pragma unroll
    for (int j = 0; j < 2; ++j)
#pragma unroll
        for (int q = 0; q < 3; ++q) {
#pragma unroll
            for (int c = 0; c < KC0; ++c) w0[j][q][c] = as_bf16x8(reinterpret_cast<const u32x4*>(p.w[0])[(((2 * w + j) * 3 + q) * KC0 + c) * 64 + lane]);
            if constexpr (NL > 1) {
#pragma unroll
                for (int c = 0; c < KC1; ++c) w1[j][q][c] = as_bf16x8(reinterpret_cast<const u32x4*>(p.w[1])[(((2 * w + j) * 3 + q) * KC1 + c) * 64 + lane]);
            }
        }
    if constexpr (NL > 1) {
#pragma unroll
        for (int j = 0; j < 2; ++j)
#pragma unroll
            for (int q = 0; q < 3; ++q)
#pragma unroll
                for (int c = 0; c < KC1; ++c) asm volatile("" : "+a"(w1[j][q][c]));
    }
#pragma unroll
    for (int j = 0; j < 2; ++j)
#pragma unroll
        for (int c = KX0; c < KC0; ++c) {
            asm volatile("" : "+a"(w0[j][0][c]));                                // r-gate h-part of layer 0
            if constexpr (NL > 1) asm volatile("" : "+a"(w0[j][1][c]));          // u-gate too in the skewed 2-layer loop
        }
    asm volatile("s_nop 7" ::: "memory");
    const bf16x8 wfc = as_bf16x8(reinterpret_cast<const u32x4*>(p.wfc)[w * 64 + lane]);
    f32x4 bfc4 = splat4(0.f);
    if (w == 0) bfc4 = ld4(p.bfc + 4 * g);

    // ---- LDS init: biases, zeroed mel staging, initial state -------------------------------------------
    for (int i = tid; i < NL * 3 * H; i += 256) biasl[i] = p.bias[i / (3 * H)][i % (3 * H)];
    for (int i = tid; i < KX0 * 64; i += 256) xsb[i] = (u32x4){0u, 0u, 0u, 0u};
    const bool do_reset = p.reset != nullptr && p.reset[b] != 0;
    const int len_s = p.seq_len ? p.seq_len[b] : T;
    f32x4 hreg[NL][2];
#pragma unroll
    for (int l = 0; l < NL; ++l) {
#pragma unroll
        for (int j = 0; j < 2; ++j)
            hreg[l][j] = do_reset ? splat4(0.f)
                                  : ld4(p.state_in + ((size_t)l * p.B + b) * H + (2 * w + j) * 16 + 4 * g);
        hb[(l * 4 + w) * 64 + lane] = (u32x4){pack_bf16(hreg[l][0][0], hreg[l][0][1]), pack_bf16(hreg[l][0][2], hreg[l][0][3]),
                                               pack_bf16(hreg[l][1][0], hreg[l][1][1]), pack_bf16(hreg[l][1][2], hreg[l][1][3])};
    }
    if (tid < 16) {
        const int bb = group * kStreamsPerGroup + tid;
        int pw = -1;
        if (bb < p.B && p.epi.prev_word && !(p.reset && p.reset[bb])) pw = p.epi.prev_word[bb];
        epi.carry[tid] = pw;
    }

    // ---- mel: wave w fetches streams 4w..4w+3 (one dwordx4 per lane), rounds to bf16, scatters into the
    // B-operand image: k = 4q+e -> chunk k/32, lane group (k%32)/8, element k%8 ------------------------------
    const int XQ = p.I / 4;                               // float4 pieces per mel row (I % 4 == 0)
    const int xl_row = lane / XQ, xl_q = lane % XQ;
    const bool xl_active = lane < 4 * XQ;
    const int xl_b = min(group * kStreamsPerGroup + 4 * w + (xl_active ? xl_row : 0), p.B - 1);
    const float4* xl_src = reinterpret_cast<const float4*>(p.x_mel + (size_t)xl_b * T * p.I) + xl_q;
    unsigned* xs_dst = reinterpret_cast<unsigned*>(xsb) +
                       (((xl_q * 4) / 32) * 64 + (((xl_q * 4) % 32) / 8) * 16 + (4 * w + xl_row)) * 4 + ((xl_q * 4) % 8) / 2;
    float4 fl_a = make_float4(0.f, 0.f, 0.f, 0.f), fl_b = fl_a;     // two frames in flight
    auto fetch = [&](float4& r, int t_req) { if (xl_active) r = xl_src[(size_t)(t_req < T ? t_req : T - 1) * XQ]; };
    auto commit = [&](const float4& r) {
        if (xl_active) *reinterpret_cast<uint2*>(xs_dst) = make_uint2(pack_bf16(r.x, r.y), pack_bf16(r.z, r.w));
    };

    __syncthreads();
    if (T > 0) {
        fetch(fl_a, 0);
        commit(fl_a);                 // x(0)
        fetch(fl_b, 1);               // x(1): committed during frame 0
        fetch(fl_a, 2);               // x(2): committed during frame 1
        __syncthreads();
    }

    // one GRU layer for one frame; xB/nx = this layer's input chunks, returns with hreg[l] updated, hb[l] rewritten
    auto layer = [&](auto l_, const bf16x8* xB, auto nx_, int t, f32x4 (&hout)[2]) {
        constexpr int l = decltype(l_)::value, NX = decltype(nx_)::value;
        const f32x4* bl = reinterpret_cast<const f32x4*>(biasl + l * 3 * H);
        f32x4 acc_r[2], acc_u[2], acc_c[2];
#pragma unroll
        for (int j = 0; j < 2; ++j) {
            acc_r[j] = bl[(0 * H + (2 * w + j) * 16) / 4 + g];
            acc_u[j] = bl[(1 * H + (2 * w + j) * 16) / 4 + g];
            acc_c[j] = bl[(2 * H + (2 * w + j) * 16) / 4 + g];
        }
        bf16x8 hB[4];
#pragma unroll
        for (int m = 0; m < 4; ++m) hB[m] = as_bf16x8(hb[(l * 4 + m) * 64 + lane]);
        // x-part of all three gates, then the gate h-part
        if constexpr (l != 0) { mfma_prefence(acc_r[0], acc_u[0], acc_r[1], acc_u[1]); mfma_prefence(acc_c[0], acc_c[1]); }
#pragma unroll
        for (int c = 0; c < NX; ++c) {
#pragma unroll
            for (int j = 0; j < 2; ++j) {
                if constexpr (l == 0) {
                    acc_r[j] = mfma_bf16(w0[j][0][c], xB[c], acc_r[j]);
                    acc_u[j] = mfma_bf16(w0[j][1][c], xB[c], acc_u[j]);
                    acc_c[j] = mfma_bf16(w0[j][2][c], xB[c], acc_c[j]);
                } else {
                    KWS_MFMA_BF16_A(acc_r[j], w1[j][0][c], xB[c]);
                    KWS_MFMA_BF16_A(acc_u[j], w1[j][1][c], xB[c]);
                    KWS_MFMA_BF16_A(acc_c[j], w1[j][2][c], xB[c]);
                }
            }
        }
        mfma_prefence(acc_r[0], acc_u[0], acc_r[1], acc_u[1]);
#pragma unroll
        for (int m = 0; m < 4; ++m) {
#pragma unroll
            for (int j = 0; j < 2; ++j) {
                if constexpr (l == 0) {
                    KWS_MFMA_BF16_A(acc_r[j], w0[j][0][NX + m], hB[m]);
                    acc_u[j] = mfma_bf16(w0[j][1][NX + m], hB[m], acc_u[j]);
                } else {
                    KWS_MFMA_BF16_A(acc_r[j], w1[j][0][NX + m], hB[m]);
                    KWS_MFMA_BF16_A(acc_u[j], w1[j][1][NX + m], hB[m]);
                }
            }
        }
        mfma_fence(acc_r[0], acc_u[0], acc_r[1], acc_u[1]);
        f32x4 u[2], rh[2];
#pragma unroll
        for (int j = 0; j < 2; ++j) {
            const f32x2 r_lo = sigmoid2((f32x2){acc_r[j][0], acc_r[j][1]});
            const f32x2 r_hi = sigmoid2((f32x2){acc_r[j][2], acc_r[j][3]});
            const f32x2 a = r_lo * (f32x2){hreg[l][j][0], hreg[l][j][1]};
            const f32x2 c2 = r_hi * (f32x2){hreg[l][j][2], hreg[l][j][3]};
            rh[j] = (f32x4){a.x, a.y, c2.x, c2.y};
        }
        rhb[(l * 4 + w) * 64 + lane] = (u32x4){pack_bf16(rh[0][0], rh[0][1]), pack_bf16(rh[0][2], rh[0][3]),
                                                pack_bf16(rh[1][0], rh[1][1]), pack_bf16(rh[1][2], rh[1][3])};
#pragma unroll
        for (int j = 0; j < 2; ++j) {
            const f32x2 u_lo = sigmoid2((f32x2){acc_u[j][0], acc_u[j][1]});
            const f32x2 u_hi = sigmoid2((f32x2){acc_u[j][2], acc_u[j][3]});
            u[j] = (f32x4){u_lo.x, u_lo.y, u_hi.x, u_hi.y};
        }
        lds_barrier();            // r(.)h visible; hb[l] fully consumed
        mfma_prefence(acc_c[0], acc_c[1]);
#pragma unroll
        for (int m = 0; m < 4; ++m) {
            const bf16x8 rB = as_bf16x8(rhb[(l * 4 + m) * 64 + lane]);
#pragma unroll
            for (int j = 0; j < 2; ++j) {
                if constexpr (l == 0) acc_c[j] = mfma_bf16(w0[j][2][NX + m], rB, acc_c[j]);
                else KWS_MFMA_BF16_A(acc_c[j], w1[j][2][NX + m], rB);
            }
        }
        mfma_fence(acc_c[0], acc_c[1]);
        const unsigned live = t < len_s ? 0xffffffffu : 0u;
#pragma unroll
        for (int j = 0; j < 2; ++j) {
#pragma unroll
            for (int h2 = 0; h2 < 2; ++h2) {
                const f32x2 c = tanh2((f32x2){acc_c[j][2 * h2], acc_c[j][2 * h2 + 1]});
                const f32x2 uu = {u[j][2 * h2], u[j][2 * h2 + 1]};
                const f32x2 hh = {hreg[l][j][2 * h2], hreg[l][j][2 * h2 + 1]};
                const f32x2 hn = (1.0f - uu) * c + uu * hh;
                hreg[l][j][2 * h2] = bitsel(live, hn.x, hh.x);
                hreg[l][j][2 * h2 + 1] = bitsel(live, hn.y, hh.y);
                hout[j][2 * h2] = bitsel(live, hn.x, 0.f);
                hout[j][2 * h2 + 1] = bitsel(live, hn.y, 0.f);
            }
        }
        hb[(l * 4 + w) * 64 + lane] = (u32x4){pack_bf16(hreg[l][0][0], hreg[l][0][1]), pack_bf16(hreg[l][0][2], hreg[l][0][3]),
                                               pack_bf16(hreg[l][1][0], hreg[l][1][1]), pack_bf16(hreg[l][1][2], hreg[l][1][3])};
    };

    auto frame = [&](int t, float4& fl_commit, float4& /*unused*/) {
        bf16x8 xB[KX0];
#pragma unroll
        for (int c = 0; c < KX0; ++c) xB[c] = as_bf16x8(xsb[c * 64 + lane]);
        f32x4 hout[2];
        layer(std::integral_constant<int, 0>{}, xB, std::integral_constant<int, KX0>{}, t, hout);
        // xsb was read by every wave before the barrier inside layer 0: stage x(t+1), request x(t+3)
        commit(fl_commit);
        fetch(fl_commit, t + 3);
        lds_barrier();            // h0(t) (and x(t+1)) visible
        if constexpr (NL > 1) {
            bf16x8 x1B[4];
#pragma unroll
            for (int m = 0; m < 4; ++m) x1B[m] = as_bf16x8(hb[(0 * 4 + m) * 64 + lane]);
            layer(std::integral_constant<int, 1>{}, x1B, std::integral_constant<int, 4>{}, t, hout);
        }
        // dense: this wave's 32 units are exactly k-chunk w of Wfc^T
        const bf16x8 hB = as_bf16x8((u32x4){pack_bf16(hout[0][0], hout[0][1]), pack_bf16(hout[0][2], hout[0][3]),
                                             pack_bf16(hout[1][0], hout[1][1]), pack_bf16(hout[1][2], hout[1][3])});
        u32x4 hBv = __builtin_bit_cast(u32x4, hB);
        asm volatile("s_nop 3" : "+v"(hBv));      // v_cvt_pk (inline asm) -> MFMA SrcB distance
        const f32x4 accf = mfma_bf16(wfc, __builtin_bit_cast(bf16x8, hBv), bfc4);
        if (g < 2) *reinterpret_cast<f32x4*>(epi.pstage + (w * 16 + s) * 8 + 4 * g) = accf;
        lds_barrier();            // top-layer h(t) and the partial logits visible
        if (w == (t & 3)) epilogue_fold(epi, t, lane);
        if (((t + 1) & (kRingFrames - 1)) == 0 || t == T - 1) {
            const int t0 = t & ~(kRingFrames - 1);
            lds_barrier();
            epilogue_flush(p.epi, epi, group, t0, t - t0 + 1, w, lane, t == T - 1);
        }
    };
    if constexpr (NL == 2) {
        // ---- skewed two-layer loop: iteration i runs layer 0 on frame i+1 and layer 1 on frame i in the SAME two
        // phases, so a frame costs 2 barriers and 2 LDS round trips instead of 4-5.  hb[0] = h0(i) is read once
        // and serves both as layer 0's previous state and as layer 1's input.  Iterations i = -1 and i = T-1 run one
        // of the layers on nothing: its `live` mask is off, so its state is left untouched (bitsel).
        const f32x4* bl0 = reinterpret_cast<const f32x4*>(biasl);
        const f32x4* bl1 = reinterpret_cast<const f32x4*>(biasl + 3 * H);
        auto iteration = [&](int i, float4& fl_commit) {
            // ---------------- phase 1: gates of both layers ----------------
            bf16x8 xB[KX0], h0B[4], h1B[4];
#pragma unroll
            for (int c = 0; c < KX0; ++c) xB[c] = as_bf16x8(xsb[c * 64 + lane]);
#pragma unroll
            for (int m = 0; m < 4; ++m) { h0B[m] = as_bf16x8(hb[(0 * 4 + m) * 64 + lane]); h1B[m] = as_bf16x8(hb[(1 * 4 + m) * 64 + lane]); }
            f32x4 r0[2], u0[2], c0[2], r1[2], u1[2], c1[2];
#pragma unroll
            for (int j = 0; j < 2; ++j) {
                const int o = (2 * w + j) * 4 + g;
                r0[j] = bl0[0 * 32 + o]; u0[j] = bl0[1 * 32 + o]; c0[j] = bl0[2 * 32 + o];
                r1[j] = bl1[0 * 32 + o]; u1[j] = bl1[1 * 32 + o]; c1[j] = bl1[2 * 32 + o];
            }
            // layer 0, frame i+1: x-part (VGPR operands, builtin) then gate h-part (AGPR operands)
#pragma unroll
            for (int c = 0; c < KX0; ++c)
#pragma unroll
                for (int j = 0; j < 2; ++j) {
                    r0[j] = mfma_bf16(w0[j][0][c], xB[c], r0[j]);
                    u0[j] = mfma_bf16(w0[j][1][c], xB[c], u0[j]);
                    c0[j] = mfma_bf16(w0[j][2][c], xB[c], c0[j]);
                }
            mfma_prefence(r0[0], u0[0], r0[1], u0[1]);
#pragma unroll
            for (int m = 0; m < 4; ++m)
#pragma unroll
                for (int j = 0; j < 2; ++j) {
                    KWS_MFMA_BF16_A(r0[j], w0[j][0][KX0 + m], h0B[m]);
                    KWS_MFMA_BF16_A(u0[j], w0[j][1][KX0 + m], h0B[m]);
                }
            // layer 1, frame i: x-part on h0(i) (the same registers), gate h-part on h1(i-1)
            mfma_prefence(r1[0], u1[0], r1[1], u1[1]);
            mfma_prefence(c1[0], c1[1]);
#pragma unroll
            for (int m = 0; m < 4; ++m)
#pragma unroll
                for (int j = 0; j < 2; ++j) {
                    KWS_MFMA_BF16_A(r1[j], w1[j][0][m], h0B[m]);
                    KWS_MFMA_BF16_A(u1[j], w1[j][1][m], h0B[m]);
                    KWS_MFMA_BF16_A(c1[j], w1[j][2][m], h0B[m]);
                }
#pragma unroll
            for (int m = 0; m < 4; ++m)
#pragma unroll
                for (int j = 0; j < 2; ++j) {
                    KWS_MFMA_BF16_A(r1[j], w1[j][0][4 + m], h1B[m]);
                    KWS_MFMA_BF16_A(u1[j], w1[j][1][4 + m], h1B[m]);
                }
            mfma_fence(r0[0], u0[0], r0[1], u0[1]);
            mfma_fence(r1[0], u1[0], r1[1], u1[1]);
            // sigmoids: r first (feeds the exchange), then u
            f32x4 rh0[2], rh1[2];
#pragma unroll
            for (int j = 0; j < 2; ++j) {
#pragma unroll
                for (int h2 = 0; h2 < 2; ++h2) {
                    const f32x2 a = sigmoid2((f32x2){r0[j][2 * h2], r0[j][2 * h2 + 1]}) * (f32x2){hreg[0][j][2 * h2], hreg[0][j][2 * h2 + 1]};
                    const f32x2 b2 = sigmoid2((f32x2){r1[j][2 * h2], r1[j][2 * h2 + 1]}) * (f32x2){hreg[1][j][2 * h2], hreg[1][j][2 * h2 + 1]};
                    rh0[j][2 * h2] = a.x; rh0[j][2 * h2 + 1] = a.y;
                    rh1[j][2 * h2] = b2.x; rh1[j][2 * h2 + 1] = b2.y;
                }
            }
            rhb[(0 * 4 + w) * 64 + lane] = (u32x4){pack_bf16(rh0[0][0], rh0[0][1]), pack_bf16(rh0[0][2], rh0[0][3]),
                                                    pack_bf16(rh0[1][0], rh0[1][1]), pack_bf16(rh0[1][2], rh0[1][3])};
            rhb[(1 * 4 + w) * 64 + lane] = (u32x4){pack_bf16(rh1[0][0], rh1[0][1]), pack_bf16(rh1[0][2], rh1[0][3]),
                                                    pack_bf16(rh1[1][0], rh1[1][1]), pack_bf16(rh1[1][2], rh1[1][3])};
#pragma unroll
            for (int j = 0; j < 2; ++j) {
#pragma unroll
                for (int h2 = 0; h2 < 2; ++h2) {
                    const f32x2 a = sigmoid2((f32x2){u0[j][2 * h2], u0[j][2 * h2 + 1]});
                    const f32x2 b2 = sigmoid2((f32x2){u1[j][2 * h2], u1[j][2 * h2 + 1]});
                    u0[j][2 * h2] = a.x; u0[j][2 * h2 + 1] = a.y;
                    u1[j][2 * h2] = b2.x; u1[j][2 * h2 + 1] = b2.y;
                }
            }
            lds_barrier();            // #1: both r(.)h visible; hb / xsb fully consumed
            // ---------------- phase 2: candidates, state updates, dense ----------------
            mfma_prefence(c0[0], c0[1]);
            mfma_prefence(c1[0], c1[1]);
#pragma unroll
            for (int m = 0; m < 4; ++m) {
                const bf16x8 rB0 = as_bf16x8(rhb[(0 * 4 + m) * 64 + lane]);
                const bf16x8 rB1 = as_bf16x8(rhb[(1 * 4 + m) * 64 + lane]);
#pragma unroll
                for (int j = 0; j < 2; ++j) {
                    c0[j] = mfma_bf16(w0[j][2][KX0 + m], rB0, c0[j]);
                    KWS_MFMA_BF16_A(c1[j], w1[j][2][4 + m], rB1);
                }
            }
            mfma_fence(c0[0], c0[1], c1[0], c1[1]);
            const unsigned live0 = (i + 1 < T && i + 1 < len_s) ? 0xffffffffu : 0u;
            const unsigned live1 = (i >= 0 && i < len_s) ? 0xffffffffu : 0u;
            f32x4 hout[2];
#pragma unroll
            for (int j = 0; j < 2; ++j) {
#pragma unroll
                for (int h2 = 0; h2 < 2; ++h2) {
                    {
                        const f32x2 c = tanh2((f32x2){c0[j][2 * h2], c0[j][2 * h2 + 1]});
                        const f32x2 uu = {u0[j][2 * h2], u0[j][2 * h2 + 1]};
                        const f32x2 hh = {hreg[0][j][2 * h2], hreg[0][j][2 * h2 + 1]};
                        const f32x2 hn = (1.0f - uu) * c + uu * hh;
                        hreg[0][j][2 * h2] = bitsel(live0, hn.x, hh.x);
                        hreg[0][j][2 * h2 + 1] = bitsel(live0, hn.y, hh.y);
                    }
                    {
                        const f32x2 c = tanh2((f32x2){c1[j][2 * h2], c1[j][2 * h2 + 1]});
                        const f32x2 uu = {u1[j][2 * h2], u1[j][2 * h2 + 1]};
                        const f32x2 hh = {hreg[1][j][2 * h2], hreg[1][j][2 * h2 + 1]};
                        const f32x2 hn = (1.0f - uu) * c + uu * hh;
                        hreg[1][j][2 * h2] = bitsel(live1, hn.x, hh.x);
                        hreg[1][j][2 * h2 + 1] = bitsel(live1, hn.y, hh.y);
                        hout[j][2 * h2] = bitsel(live1, hn.x, 0.f);
                        hout[j][2 * h2 + 1] = bitsel(live1, hn.y, 0.f);
                    }
                }
            }
#pragma unroll
            for (int l = 0; l < 2; ++l)
                hb[(l * 4 + w) * 64 + lane] = (u32x4){pack_bf16(hreg[l][0][0], hreg[l][0][1]), pack_bf16(hreg[l][0][2], hreg[l][0][3]),
                                                       pack_bf16(hreg[l][1][0], hreg[l][1][1]), pack_bf16(hreg[l][1][2], hreg[l][1][3])};
            commit(fl_commit);        // x(i+2)
            fetch(fl_commit, i + 4);
            if (i >= 0) {
                u32x4 hBv = (u32x4){pack_bf16(hout[0][0], hout[0][1]), pack_bf16(hout[0][2], hout[0][3]),
                                    pack_bf16(hout[1][0], hout[1][1]), pack_bf16(hout[1][2], hout[1][3])};
                asm volatile("s_nop 3" : "+v"(hBv));
                const f32x4 accf = mfma_bf16(wfc, as_bf16x8(hBv), bfc4);
                if (g < 2) *reinterpret_cast<f32x4*>(epi.pstage + (w * 16 + s) * 8 + 4 * g) = accf;
            }
            lds_barrier();            // #2: h0(i+1), h1(i), x(i+2), partial logits visible
            if (i >= 0) {
                if (w == (i & 3)) epilogue_fold(epi, i, lane);
                if (((i + 1) & (kRingFrames - 1)) == 0 || i == T - 1) {
                    const int t0 = i & ~(kRingFrames - 1);
                    lds_barrier();
                    epilogue_flush(p.epi, epi, group, t0, i - t0 + 1, w, lane, i == T - 1);
                }
            }
        };
        // prologue left x(0) in xsb, x(1) in fl_b, x(2) in fl_a: iteration i commits x(i+2)
        // i = -1 commits x(1) = fl_b, i = 0 commits x(2) = fl_a, ...
        for (int i = -1; i < T; i += 2) {
            iteration(i, fl_b);
            if (i + 1 < T) iteration(i + 1, fl_a);
        }
    } else {
    // fl_b holds x(t+1) on even frames, fl_a on odd ones
    for (int t = 0; t < T; t += 2) {
        frame(t, fl_b, fl_a);
        if (t + 1 < T) frame(t + 1, fl_a, fl_b);
    }
    }

    if (bvalid) {
#pragma unroll
        for (int l = 0; l < NL; ++l)
#pragma unroll
            for (int j = 0; j < 2; ++j)
                *reinterpret_cast<f32x4*>(p.state_out + ((size_t)l * p.B + b) * H + (2 * w + j) * 16 + 4 * g) = hreg[l][j];
    }
}

// ------------------------------------------------------------------------------------------------------------
// Two waves per SIMD, LAYER-SPECIALISED (NL == 2): waves 0..3 run layer 0, waves 4..7 layer 1, each on its 32 units as
// above, in the same skewed schedule -- iteration i = layer 0 on frame i+1 beside layer 1 on frame i, two workgroup
// barriers.  A lone wave issues a VALU instruction only every ~4.7 cycles, two waves on a SIMD one every ~2.3
// (tools/ubench/valu_rate.hip), and bf16 MFMAs do not share the FP32 datapath with the VALU: the frame is mostly
// activations, conversions, LDS round trips and fences (tools/ubench/waves_per_simd_bf16.hip: MFMAs are a quarter of
// it), so a layer-0 wave and a layer-1 wave that share a SIMD fill each other's gaps.  (The fp32 kernels gain nothing
// from a second wave -- waves_per_simd.hip: their MFMAs and VALU share one pipe.)  256 registers per wave: layer 0 keeps
// its 36 operands in registers, layer 1 its 32 gate operands; layer 1's 16 candidate operands stream from LDS (64 KiB).
// ------------------------------------------------------------------------------------------------------------
// WINDOW: the decode-window step of the stream manager rides at the end of every group (window_device.h)
template <int KX0, bool WINDOW = false>
__global__ void __launch_bounds__(512) __attribute__((amdgpu_waves_per_eu(2, 2)))
gru_stack_bf16_ls(const GruBf16Params p) {
    constexpr int H = 128, NL = 2;
    const int tid = threadIdx.x;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int layer = wave >> 2, w = wave & 3;             // w: this wave's 32 units within its layer
    const int lane = tid & 63, g = lane >> 4, s = lane & 15;
    // persistent over stream groups (as the fp32 resident kernels): operands and LDS tables are set up once per workgroup
    const int n_groups = (p.B + kStreamsPerGroup - 1) / kStreamsPerGroup;
    int group = blockIdx.x;
    int b_raw = 0, b = 0;
    bool bvalid = false;
    const int T = p.T;

    extern __shared__ __attribute__((aligned(16))) char smem[];
    u32x4* hb = reinterpret_cast<u32x4*>(smem);             // [NL][4 chunks][64]  h_{t-1} (bf16)
    u32x4* rhb = hb + NL * 4 * 64;                           // [NL][4][64]         r (.) h_{t-1}
    u32x4* xsb = rhb + NL * 4 * 64;                          // [KX0][64]           mel frame (bf16), zero padded
    u32x4* wc1 = xsb + KX0 * 64;                             // [4 waves][2 tiles][8 chunks][64]  layer 1 candidate operands
    float* biasl = reinterpret_cast<float*>(wc1 + 4 * 2 * 8 * 64); // [NL][3][128]
    EpilogueLds epi = epilogue_carve(reinterpret_cast<char*>(biasl + NL * 3 * H));
    if constexpr (WINDOW) epi.cwords = reinterpret_cast<int8_t*>(reinterpret_cast<char*>(biasl + NL * 3 * H) + kEpilogueLdsBytes);
    const uint8_t* win_dl = reinterpret_cast<const uint8_t*>(epi.cwords) + 16 * kWinTailWordsStride;     // WINDOW only: the label matcher
    constexpr size_t kWinOffset = offsetof(GruBf16Params, epi) + offsetof(GruLayerParams, win);
    if constexpr (WINDOW) window_tail_prepare(window_tail_params_from_kernarg(kWinOffset), const_cast<uint8_t*>(win_dl), tid);   // (visible after the first barrier below)

    constexpr int KC0 = KX0 + 4, KC1 = 8;
    // ---- LDS init (all eight waves): biases, zeroed mel staging, layer 1's candidate operands, initial state -------
    for (int i = tid; i < NL * 3 * H; i += 512) biasl[i] = p.bias[i / (3 * H)][i % (3 * H)];
    for (int i = tid; i < KX0 * 64; i += 512) xsb[i] = (u32x4){0u, 0u, 0u, 0u};
    for (int i = tid; i < 4 * 2 * 8 * 64; i += 512) {
        const int ln = i & 63, c = (i >> 6) & 7, j = (i >> 9) & 1, ww = i >> 10;
        wc1[i] = reinterpret_cast<const u32x4*>(p.w[1])[(((2 * ww + j) * 3 + 2) * KC1 + c) * 64 + ln];
    }
    int len_s = T;
    f32x4 hreg[2];
    auto enter_group = [&]() {
        b_raw = group * kStreamsPerGroup + s;
        bvalid = b_raw < p.B;
        b = bvalid ? b_raw : p.B - 1;
        const bool do_reset = p.reset != nullptr && p.reset[b] != 0;
        len_s = p.seq_len ? p.seq_len[b] : T;
#pragma unroll
        for (int j = 0; j < 2; ++j)
            hreg[j] = do_reset ? splat4(0.f) : ld4(p.state_in + ((size_t)layer * p.B + b) * H + (2 * w + j) * 16 + 4 * g);
        hb[(layer * 4 + w) * 64 + lane] = (u32x4){pack_bf16(hreg[0][0], hreg[0][1]), pack_bf16(hreg[0][2], hreg[0][3]),
                                                   pack_bf16(hreg[1][0], hreg[1][1]), pack_bf16(hreg[1][2], hreg[1][3])};
        if (tid < 16) {
            const int bb = group * kStreamsPerGroup + tid;
            int pw = -1;
            if (bb < p.B && p.epi.prev_word && !(p.reset && p.reset[bb])) pw = p.epi.prev_word[bb];
            epi.carry[tid] = pw;
        }
    };
    auto leave_group = [&]() {
        if (bvalid) {
#pragma unroll
            for (int j = 0; j < 2; ++j)
                *reinterpret_cast<f32x4*>(p.state_out + ((size_t)layer * p.B + b) * H + (2 * w + j) * 16 + 4 * g) = hreg[j];
        }
        __syncthreads();         // every wave is done with this group's LDS state before the next group's is written
        if constexpr (WINDOW) {
            // detector.py:195-209 for this group's 16 streams (all eight waves keep the barriers, four work): the call's
            // frame words wait in epi.cwords, the scratch is hb | rhb
            const WindowTail win = window_tail_params_from_kernarg(kWinOffset);
            WindowTailRegs<2> wreq;
            window_tail_request<2>(win, p.B, group * kStreamsPerGroup, tid, wreq);
            window_tail<2>(win, p.B, group * kStreamsPerGroup, T, epi.cwords, kWinTailWordsStride, win_dl, reinterpret_cast<char*>(hb), tid, wreq);
            __syncthreads();
        }
    };
    const f32x4* bl = reinterpret_cast<const f32x4*>(biasl + layer * 3 * H);
    // the epilogue's barriers are workgroup barriers: layer 0's waves keep step with them
    auto flush_due = [&](int i) { return i >= 0 && ((((i + 1) & (kRingFrames - 1)) == 0) || i == T - 1); };

    if (layer == 0) {
        // ================= layer 0: frame i+1 in iteration i =================
        bf16x8 w0[2][3][KC0];
#pragma unroll
        for (int j = 0; j < 2; ++j)
#pragma unroll
            for (int q = 0; q < 3; ++q)
#pragma unroll
                for (int c = 0; c < KC0; ++c)
                    w0[j][q][c] = as_bf16x8(reinterpret_cast<const u32x4*>(p.w[0])[(((2 * w + j) * 3 + q) * KC0 + c) * 64 + lane]);
        // all loads first, the AGPR pins afterwards: a pin right behind its load makes the compiler wait for that load before it
        // issues the next one (36 serial L2 round trips per launch).
        // 2 x 3 x KC0 operands: 30 (KX0 = 1) fit the 32 operand slots of this wave's 128 AGPRs.  Of the 36 at KX0 = 2 the
        // candidate's x-part (4) stays in VGPRs and goes through the builtin MFMA: pinned, the compiler kept them in VGPRs anyway
        // and shuttled them into an AGPR (v_accvgpr_write) right in front of the inline-asm MFMA that reads it -- 56 copies per
        // iteration and a VALU-write -> MFMA-read distance nobody checks (round 6: a re-scheduled variant of this kernel
        // computed garbage that way; this one had been lucky).
        constexpr bool kCxInVgpr = 2 * 3 * KC0 > 32;
#pragma unroll
        for (int j = 0; j < 2; ++j)
#pragma unroll
            for (int q = 0; q < 3; ++q)
#pragma unroll
                for (int c = 0; c < KC0; ++c)
                    if (!(kCxInVgpr && q == 2 && c < KX0)) asm volatile("" : "+a"(w0[j][q][c]));
        asm volatile("s_nop 7" ::: "memory");
        // mel: wave w fetches streams 4w..4w+3 (one dwordx4 per lane), rounds to bf16, scatters into the B-operand image
        const int XQ = p.I / 4;
        const int xl_row = lane / XQ, xl_q = lane % XQ;
        const bool xl_active = lane < 4 * XQ;
        const float4* xl_src = nullptr;
        unsigned* xs_dst = reinterpret_cast<unsigned*>(xsb) +
                           (((xl_q * 4) / 32) * 64 + (((xl_q * 4) % 32) / 8) * 16 + (4 * w + xl_row)) * 4 + ((xl_q * 4) % 8) / 2;
        float4 fl_a = make_float4(0.f, 0.f, 0.f, 0.f), fl_b = fl_a;
        auto fetch = [&](float4& r, int t_req) { if (xl_active) r = xl_src[(size_t)(t_req < T ? t_req : T - 1) * XQ]; };
        auto commit = [&](const float4& r) {
            if (xl_active) *reinterpret_cast<uint2*>(xs_dst) = make_uint2(pack_bf16(r.x, r.y), pack_bf16(r.z, r.w));
        };
        auto iteration = [&](int i, float4& fl_commit) {
            bf16x8 xB[KX0], h0B[4];
#pragma unroll
            for (int c = 0; c < KX0; ++c) xB[c] = as_bf16x8(xsb[c * 64 + lane]);
#pragma unroll
            for (int m = 0; m < 4; ++m) h0B[m] = as_bf16x8(hb[(0 * 4 + m) * 64 + lane]);
            f32x4 r0[2], u0[2], c0[2];
#pragma unroll
            for (int j = 0; j < 2; ++j) {
                const int o = (2 * w + j) * 4 + g;
                r0[j] = bl[0 * 32 + o]; u0[j] = bl[1 * 32 + o]; c0[j] = bl[2 * 32 + o];
            }
            if constexpr (kCxInVgpr) {
#pragma unroll
                for (int c = 0; c < KX0; ++c)
#pragma unroll
                    for (int j = 0; j < 2; ++j) c0[j] = mfma_bf16(w0[j][2][c], xB[c], c0[j]);
            }
            mfma_prefence(r0[0], u0[0], r0[1], u0[1]);
            if constexpr (!kCxInVgpr) mfma_prefence(c0[0], c0[1]);
#pragma unroll
            for (int c = 0; c < KX0; ++c)
#pragma unroll
                for (int j = 0; j < 2; ++j) {
                    KWS_MFMA_BF16_A(r0[j], w0[j][0][c], xB[c]);
                    KWS_MFMA_BF16_A(u0[j], w0[j][1][c], xB[c]);
                    if constexpr (!kCxInVgpr) KWS_MFMA_BF16_A(c0[j], w0[j][2][c], xB[c]);
                }
#pragma unroll
            for (int m = 0; m < 4; ++m)
#pragma unroll
                for (int j = 0; j < 2; ++j) {
                    KWS_MFMA_BF16_A(r0[j], w0[j][0][KX0 + m], h0B[m]);
                    KWS_MFMA_BF16_A(u0[j], w0[j][1][KX0 + m], h0B[m]);
                }
            mfma_fence(r0[0], u0[0], r0[1], u0[1]);
            f32x4 rh0[2];
#pragma unroll
            for (int j = 0; j < 2; ++j)
#pragma unroll
                for (int h2 = 0; h2 < 2; ++h2) {
                    const f32x2 a = sigmoid2((f32x2){r0[j][2 * h2], r0[j][2 * h2 + 1]}) * (f32x2){hreg[j][2 * h2], hreg[j][2 * h2 + 1]};
                    rh0[j][2 * h2] = a.x; rh0[j][2 * h2 + 1] = a.y;
                }
            rhb[(0 * 4 + w) * 64 + lane] = (u32x4){pack_bf16(rh0[0][0], rh0[0][1]), pack_bf16(rh0[0][2], rh0[0][3]),
                                                    pack_bf16(rh0[1][0], rh0[1][1]), pack_bf16(rh0[1][2], rh0[1][3])};
#pragma unroll
            for (int j = 0; j < 2; ++j)
#pragma unroll
                for (int h2 = 0; h2 < 2; ++h2) {
                    const f32x2 a = sigmoid2((f32x2){u0[j][2 * h2], u0[j][2 * h2 + 1]});
                    u0[j][2 * h2] = a.x; u0[j][2 * h2 + 1] = a.y;
                }
            lds_barrier();            // #1: both r(.)h visible; hb / xsb fully consumed
            mfma_prefence(c0[0], c0[1]);
#pragma unroll
            for (int m = 0; m < 4; ++m) {
                const bf16x8 rB0 = as_bf16x8(rhb[(0 * 4 + m) * 64 + lane]);
#pragma unroll
                for (int j = 0; j < 2; ++j) KWS_MFMA_BF16_A(c0[j], w0[j][2][KX0 + m], rB0);
            }
            mfma_fence(c0[0], c0[1]);
            const unsigned live0 = (i + 1 < T && i + 1 < len_s) ? 0xffffffffu : 0u;
#pragma unroll
            for (int j = 0; j < 2; ++j)
#pragma unroll
                for (int h2 = 0; h2 < 2; ++h2) {
                    const f32x2 c = tanh2((f32x2){c0[j][2 * h2], c0[j][2 * h2 + 1]});
                    const f32x2 uu = {u0[j][2 * h2], u0[j][2 * h2 + 1]};
                    const f32x2 hh = {hreg[j][2 * h2], hreg[j][2 * h2 + 1]};
                    const f32x2 hn = (1.0f - uu) * c + uu * hh;
                    hreg[j][2 * h2] = bitsel(live0, hn.x, hh.x);
                    hreg[j][2 * h2 + 1] = bitsel(live0, hn.y, hh.y);
                }
            hb[(0 * 4 + w) * 64 + lane] = (u32x4){pack_bf16(hreg[0][0], hreg[0][1]), pack_bf16(hreg[0][2], hreg[0][3]),
                                                   pack_bf16(hreg[1][0], hreg[1][1]), pack_bf16(hreg[1][2], hreg[1][3])};
            commit(fl_commit);        // x(i+2)
            fetch(fl_commit, i + 4);
            lds_barrier();            // #2: h0(i+1), h1(i), x(i+2), partial logits visible
            // the epilogue (fold, softmax, decode rule, stores) runs on layer 0's waves: they carry 36 MFMAs and no LDS-fed
            // operands per iteration against layer 1's 49, so the tail work goes where the slack is
            if (i >= 0 && w == (i & 3)) epilogue_fold(epi, i, lane);
            if (flush_due(i)) {
                const int t0 = i & ~(kRingFrames - 1);
                lds_barrier();
                epilogue_flush(p.epi, epi, group, t0, i - t0 + 1, w, lane, i == T - 1);
            }
        };
        for (; group < n_groups; group += gridDim.x) {
            enter_group();
            {
                const int xl_b = min(group * kStreamsPerGroup + 4 * w + (xl_active ? xl_row : 0), p.B - 1);
                xl_src = reinterpret_cast<const float4*>(p.x_mel + (size_t)xl_b * T * p.I) + xl_q;
            }
            __syncthreads();
            if (T > 0) {
                fetch(fl_a, 0);
                commit(fl_a);                 // x(0)
                fetch(fl_b, 1);
                fetch(fl_a, 2);
            }
            __syncthreads();
            for (int i = -1; i < T; i += 2) {
                iteration(i, fl_b);
                if (i + 1 < T) iteration(i + 1, fl_a);
            }
            leave_group();
        }
    } else {
        // ================= layer 1: frame i in iteration i, the dense layer and the epilogue =================
        bf16x8 w1[2][2][KC1];            // gates only: [tile][r|u][chunk]
#pragma unroll
        for (int j = 0; j < 2; ++j)
#pragma unroll
            for (int q = 0; q < 2; ++q)
#pragma unroll
                for (int c = 0; c < KC1; ++c)
                    w1[j][q][c] = as_bf16x8(reinterpret_cast<const u32x4*>(p.w[1])[(((2 * w + j) * 3 + q) * KC1 + c) * 64 + lane]);
#pragma unroll
        for (int j = 0; j < 2; ++j)
#pragma unroll
            for (int q = 0; q < 2; ++q)
#pragma unroll
                for (int c = 0; c < KC1; ++c) asm volatile("" : "+a"(w1[j][q][c]));
        asm volatile("s_nop 7" ::: "memory");
        const bf16x8 wfc = as_bf16x8(reinterpret_cast<const u32x4*>(p.wfc)[w * 64 + lane]);
        f32x4 bfc4 = splat4(0.f);
        if (w == 0) bfc4 = ld4(p.bfc + 4 * g);
        const u32x4* wcl = wc1 + (size_t)w * 2 * 8 * 64 + lane;      // [tile][chunk][64]
        auto iteration = [&](int i) {
            bf16x8 h0B[4], h1B[4];
#pragma unroll
            for (int m = 0; m < 4; ++m) { h0B[m] = as_bf16x8(hb[(0 * 4 + m) * 64 + lane]); h1B[m] = as_bf16x8(hb[(1 * 4 + m) * 64 + lane]); }
            f32x4 r1[2], u1[2], c1[2];
#pragma unroll
            for (int j = 0; j < 2; ++j) {
                const int o = (2 * w + j) * 4 + g;
                r1[j] = bl[0 * 32 + o]; u1[j] = bl[1 * 32 + o]; c1[j] = bl[2 * 32 + o];
            }
            // candidate x-part on h0(i): operands from LDS (builtin MFMAs: the compiler tracks these hazards itself)
#pragma unroll
            for (int m = 0; m < 4; ++m)
#pragma unroll
                for (int j = 0; j < 2; ++j) c1[j] = mfma_bf16(as_bf16x8(wcl[(j * 8 + m) * 64]), h0B[m], c1[j]);
            mfma_prefence(r1[0], u1[0], r1[1], u1[1]);
#pragma unroll
            for (int m = 0; m < 4; ++m)
#pragma unroll
                for (int j = 0; j < 2; ++j) {
                    KWS_MFMA_BF16_A(r1[j], w1[j][0][m], h0B[m]);
                    KWS_MFMA_BF16_A(u1[j], w1[j][1][m], h0B[m]);
                }
#pragma unroll
            for (int m = 0; m < 4; ++m)
#pragma unroll
                for (int j = 0; j < 2; ++j) {
                    KWS_MFMA_BF16_A(r1[j], w1[j][0][4 + m], h1B[m]);
                    KWS_MFMA_BF16_A(u1[j], w1[j][1][4 + m], h1B[m]);
                }
            mfma_fence(r1[0], u1[0], r1[1], u1[1]);
            f32x4 rh1[2];
#pragma unroll
            for (int j = 0; j < 2; ++j)
#pragma unroll
                for (int h2 = 0; h2 < 2; ++h2) {
                    const f32x2 a = sigmoid2((f32x2){r1[j][2 * h2], r1[j][2 * h2 + 1]}) * (f32x2){hreg[j][2 * h2], hreg[j][2 * h2 + 1]};
                    rh1[j][2 * h2] = a.x; rh1[j][2 * h2 + 1] = a.y;
                }
            rhb[(1 * 4 + w) * 64 + lane] = (u32x4){pack_bf16(rh1[0][0], rh1[0][1]), pack_bf16(rh1[0][2], rh1[0][3]),
                                                    pack_bf16(rh1[1][0], rh1[1][1]), pack_bf16(rh1[1][2], rh1[1][3])};
#pragma unroll
            for (int j = 0; j < 2; ++j)
#pragma unroll
                for (int h2 = 0; h2 < 2; ++h2) {
                    const f32x2 a = sigmoid2((f32x2){u1[j][2 * h2], u1[j][2 * h2 + 1]});
                    u1[j][2 * h2] = a.x; u1[j][2 * h2 + 1] = a.y;
                }
            lds_barrier();            // #1
#pragma unroll
            for (int m = 0; m < 4; ++m) {
                const bf16x8 rB1 = as_bf16x8(rhb[(1 * 4 + m) * 64 + lane]);
#pragma unroll
                for (int j = 0; j < 2; ++j) c1[j] = mfma_bf16(as_bf16x8(wcl[(j * 8 + 4 + m) * 64]), rB1, c1[j]);
            }
            const unsigned live1 = (i >= 0 && i < len_s) ? 0xffffffffu : 0u;
            f32x4 hout[2];
#pragma unroll
            for (int j = 0; j < 2; ++j)
#pragma unroll
                for (int h2 = 0; h2 < 2; ++h2) {
                    const f32x2 c = tanh2((f32x2){c1[j][2 * h2], c1[j][2 * h2 + 1]});
                    const f32x2 uu = {u1[j][2 * h2], u1[j][2 * h2 + 1]};
                    const f32x2 hh = {hreg[j][2 * h2], hreg[j][2 * h2 + 1]};
                    const f32x2 hn = (1.0f - uu) * c + uu * hh;
                    hreg[j][2 * h2] = bitsel(live1, hn.x, hh.x);
                    hreg[j][2 * h2 + 1] = bitsel(live1, hn.y, hh.y);
                    hout[j][2 * h2] = bitsel(live1, hn.x, 0.f);
                    hout[j][2 * h2 + 1] = bitsel(live1, hn.y, 0.f);
                }
            hb[(1 * 4 + w) * 64 + lane] = (u32x4){pack_bf16(hreg[0][0], hreg[0][1]), pack_bf16(hreg[0][2], hreg[0][3]),
                                                   pack_bf16(hreg[1][0], hreg[1][1]), pack_bf16(hreg[1][2], hreg[1][3])};
            if (i >= 0) {
                u32x4 hBv = (u32x4){pack_bf16(hout[0][0], hout[0][1]), pack_bf16(hout[0][2], hout[0][3]),
                                    pack_bf16(hout[1][0], hout[1][1]), pack_bf16(hout[1][2], hout[1][3])};
                asm volatile("s_nop 3" : "+v"(hBv));
                const f32x4 accf = mfma_bf16(wfc, as_bf16x8(hBv), bfc4);
                if (g < 2) *reinterpret_cast<f32x4*>(epi.pstage + (w * 16 + s) * 8 + 4 * g) = accf;
            }
            lds_barrier();            // #2
            if (flush_due(i)) { lds_barrier(); lds_barrier(); }     // the fold and the flush by layer 0's waves
        };
        for (; group < n_groups; group += gridDim.x) {
            enter_group();
            __syncthreads();
            __syncthreads();
            for (int i = -1; i < T; ++i) iteration(i);
            leave_group();
        }
    }
}

size_t gru_bf16_ls_lds_bytes(int kx0, bool window = false) {
    return (size_t)(2 * 2 * 4 * 64 + kx0 * 64 + 4 * 2 * 8 * 64) * 16 + (size_t)2 * 3 * 128 * 4 + kEpilogueLdsBytes + (window ? kWinTailWordsBytes : 0);
}

size_t gru_bf16_lds_bytes(int kx0, int nl) {
    return (size_t)(2 * nl * 4 * 64 + kx0 * 64) * 16 + (size_t)nl * 3 * 128 * 4 + kEpilogueLdsBytes;
}

bool gru_bf16_supported(int hidden, int n_mel, int layers) {
    return hidden == 128 && n_mel % 4 == 0 && n_mel >= 4 && n_mel <= 64 && layers >= 1 && layers <= 2;
}

template <int KX0, int NL>
static hipError_t launch_bf16(const GruBf16Params& p, hipStream_t st) {
    const size_t lds = gru_bf16_lds_bytes(KX0, NL);
    static LdsGrant granted;             // per kernel instantiation (one static per template instance) and device
    {
        const hipError_t e = grant_dynamic_lds(gru_stack_bf16<KX0, NL>, granted, lds);
        if (e != hipSuccess) return e;
    }
    const int groups = (p.B + kStreamsPerGroup - 1) / kStreamsPerGroup;
    hipLaunchKernelGGL((gru_stack_bf16<KX0, NL>), dim3(groups), dim3(256), lds, st, p);
    return hipGetLastError();
}

template <int KX0, bool WINDOW = false>
static hipError_t launch_bf16_ls(const GruBf16Params& p, hipStream_t st) {
    const size_t lds = gru_bf16_ls_lds_bytes(KX0, WINDOW);
    static LdsGrant granted;
    {
        const hipError_t e = grant_dynamic_lds(gru_stack_bf16_ls<KX0, WINDOW>, granted, lds);
        if (e != hipSuccess) return e;
    }
    const int groups = (p.B + kStreamsPerGroup - 1) / kStreamsPerGroup;
    int dev = 0, cus = 256;
    hipDeviceProp_t prop;
    static std::atomic<int> cu_cache[kMaxDevices];
    if (hipGetDevice(&dev) == hipSuccess && dev >= 0 && dev < kMaxDevices) {
        cus = cu_cache[dev].load(std::memory_order_relaxed);
        if (cus <= 0) {
            cus = hipGetDeviceProperties(&prop, dev) == hipSuccess && prop.multiProcessorCount > 0 ? prop.multiProcessorCount : 256;
            cu_cache[dev].store(cus, std::memory_order_relaxed);
        }
    }
    hipLaunchKernelGGL((gru_stack_bf16_ls<KX0, WINDOW>), dim3(groups < cus ? groups : cus), dim3(512), lds, st, p);
    return hipGetLastError();
}

static bool bf16_four_waves() {
    static const bool four = [] { const char* e = getenv("KWS_BF16_WAVES"); return e && e[0] == '4'; }();
    return four;
}
const char* gru_stack_bf16_kernel_name(int kx0, int nl) {
    if (nl == 2 && !bf16_four_waves()) return kx0 == 1 ? "gru_stack_bf16_ls<1> (both layers, one launch, 8 waves)" : "gru_stack_bf16_ls<2> (both layers, one launch, 8 waves)";
    if (nl == 2) return kx0 == 1 ? "gru_stack_bf16<1, 2> (both layers, one launch, 4 waves)" : "gru_stack_bf16<2, 2> (both layers, one launch, 4 waves)";
    return kx0 == 1 ? "gru_stack_bf16<1, 1>" : "gru_stack_bf16<2, 1>";
}
bool gru_stack_bf16_takes_window(int kx0, int nl) { return nl == 2 && !bf16_four_waves() && (kx0 == 1 || kx0 == 2); }
bool gru_bf16_vgpr_form() {
#ifdef KWS_BF16_VGPR_FORM
    return true;
#else
    return false;
#endif
}
hipError_t launch_gru_stack_bf16(const GruBf16Params& p, int kx0, int nl, hipStream_t st) {
    if (p.T <= 0 || p.B <= 0) return hipSuccess;   // the kernels prefetch frame min(t, T-1): nothing to run, nothing to read
    // two layers: the layer-specialised 8-wave kernel (KWS_BF16_WAVES=4 keeps the 4-wave kernel for A/B)
    const bool four = bf16_four_waves();
    if (p.epi.win.tab != nullptr) {           // with the window tail: the 8-wave kernel only (gru_stack_bf16_takes_window)
        if (nl != 2 || four || p.seq_len) return hipErrorInvalidValue;
        if (kx0 == 1) return launch_bf16_ls<1, true>(p, st);
        if (kx0 == 2) return launch_bf16_ls<2, true>(p, st);
        return hipErrorInvalidValue;
    }
    if (nl == 2 && !four) {
        if (kx0 == 1) return launch_bf16_ls<1>(p, st);
        if (kx0 == 2) return launch_bf16_ls<2>(p, st);
    }
    if (kx0 == 1 && nl == 1) return launch_bf16<1, 1>(p, st);
    if (kx0 == 1 && nl == 2) return launch_bf16<1, 2>(p, st);
    if (kx0 == 2 && nl == 1) return launch_bf16<2, 1>(p, st);
    if (kx0 == 2 && nl == 2) return launch_bf16<2, 2>(p, st);
    return hipErrorInvalidValue;
}

}  // namespace kws

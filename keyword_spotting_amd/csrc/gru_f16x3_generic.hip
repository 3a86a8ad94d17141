// "f16x3" for the shapes the register-resident kernels (gru_f16x3.hip: hidden = 128) do not cover -- BASELINE configs[4]: 4 x GRU
// h = 256, n_mel = 60 (config/rnn_config.py:76,84) -- with the weights STREAMED FROM L2 every frame, as gru_layer_generic /
// gru_stack_generic_pipelined (gru_kernels.hip) stream their fp32 weights.  Same semantics (models/rnn_ctc.py:155-165,202-284;
// TF-1.x GRUCell), same boundary, same tolerance as every f16x3 kernel: logits within 1e-4 of the fp64 oracle.
//
// Why: an h = 256 layer's weights are 1.5 MiB whether they are fp32 or (hi, lo) fp16 pairs, so either way a workgroup re-reads
// 1.5 MiB per frame out of its XCD's L2.  The fp32 kernel then spends 28.8 us per frame in v_mfma_f32_16x16x4_f32 (0.63 of the
// fp32 MFMA peak); three v_mfma_f32_16x16x32_f16 per operand pair need a quarter of that matrix time, and what remains is the
// L2 -> CU stream itself: tools/ubench/l2_stream_f16x3.hip measured 12.7-13.8 us per frame for 256 workgroups streaming at
// once (31 TB/s out of the eight L2s), 2.1-2.3x the fp32 kernel -- above the 1.5x the round-5 review set as the bar for building this.
//
// Operand split, exponent-scale folding, mel pre-scale and the K permutation are gru_f16x3.hip's (v = hi + 2^-11 lo; three MFMAs
// per product: main += Wh Xh, lo += Wl Xh + Wh Xl; a wave's pair of C tiles, split and packed, IS one 32-wide chunk of the next
// B operand).  One workgroup = 4 waves = 16 streams; wave w owns tiles [TPW w, TPW w + TPW) of r, u, c and h' (TPW = H / 64) and
// streams their operands: per input / hidden chunk one ROW = all of the wave's tiles x gates x (hi, lo) -- 24 KiB per wave at
// h = 256 -- through two register sets, the next row's buffer loads issued ahead of the current row's MFMAs.  Table layout
// p.w: [H/16 tiles][3 gates][KX + H/32 chunks][hi|lo][64 lanes] x 16 B, x chunks first (kws_create).
// The layers meet through the split seam [G][T][H/32 chunks][hi|lo][64 lanes] x 16 B, as in gru_f16x3.hip; in the layer-pipelined
// launch (all L x G workgroups in one grid, gru_stack_f16x3_pipelined) it lives in fine-grained memory behind per-group frame
// counters, exactly the protocol of gru_stack_generic_pipelined.
#include <cstddef>

#include "gru_device.h"

namespace kws {

namespace {

typedef _Float16 g16x8 __attribute__((ext_vector_type(8)));
typedef _Float16 g16x2 __attribute__((ext_vector_type(2)));
typedef unsigned gu32x4 __attribute__((ext_vector_type(4)));
typedef int gi32x4 __attribute__((ext_vector_type(4)));

constexpr float kGLoScale = 2048.f, kGLoInv = 1.f / 2048.f, kGMelScale = 1.f / 256.f, kGHalfMax = 65504.f;

__device__ __forceinline__ g16x8 g_f16x8(gu32x4 v) { return __builtin_bit_cast(g16x8, v); }
__device__ __forceinline__ f32x4 g_mfma(gu32x4 a, gu32x4 b, f32x4 c) {
    return __builtin_amdgcn_mfma_f32_16x16x32_f16(g_f16x8(a), g_f16x8(b), c, 0, 0, 0);
}
__device__ __forceinline__ void g_split2(f32x2 x, unsigned& hi, unsigned& lo) {
    const g16x2 h = __builtin_convertvector(x, g16x2);
    const f32x2 r = (x - __builtin_convertvector(h, f32x2)) * kGLoScale;
    const g16x2 l = __builtin_convertvector(r, g16x2);
    hi = __builtin_bit_cast(unsigned, h);
    lo = __builtin_bit_cast(unsigned, l);
}
// two C tiles -> one chunk of the next B operand, (hi, lo)
__device__ __forceinline__ void g_split8(const f32x4& a, const f32x4& b, gu32x4& hi, gu32x4& lo) {
    unsigned h[4], l[4];
    g_split2((f32x2){a[0], a[1]}, h[0], l[0]);
    g_split2((f32x2){a[2], a[3]}, h[1], l[1]);
    g_split2((f32x2){b[0], b[1]}, h[2], l[2]);
    g_split2((f32x2){b[2], b[3]}, h[3], l[3]);
    hi = (gu32x4){h[0], h[1], h[2], h[3]};
    lo = (gu32x4){l[0], l[1], l[2], l[3]};
}

}  // namespace

size_t gru_f16x3_generic_lds_bytes(int hidden, bool last) {
    const size_t hc = hidden / 32;
    size_t n = 3 * hc * 2 * 64 * 16;              // hb, rhb, xsb (the first layer's input needs <= 2 chunks of xsb)
    n += (size_t)3 * hidden * 4 + 16 * 4 + 16;    // biases + class bias + the pipelined launch's "next frame is there" flag
    if (last) n += kEpilogueLdsBytes;
    return n;
}

// TPW: tiles per wave, H = 64 TPW (2: h = 128, 4: h = 256).  PIPE: the layer-pipelined launch.
template <int TPW, bool FIRST, bool LAST, bool PIPE>
__device__ __forceinline__ void gru_f16x3_generic_body(const GruF16Params& p, const int group) {
    constexpr int H = 64 * TPW, HC = H / 32, CPW = TPW / 2;      // CPW: chunks of the hidden vector a wave produces
    static_assert(TPW == 2 || TPW == 4, "h = 128 or 256");
    const int tid = threadIdx.x;
    const int w = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int lane = tid & 63, g = lane >> 4, s = lane & 15;
    const int b_raw = group * kStreamsPerGroup + s;
    const bool bvalid = b_raw < p.B;
    const int b = bvalid ? b_raw : p.B - 1;
    const int T = p.T;
    const int KX = FIRST ? 2 * ((p.I + 63) / 64) : HC, KC = KX + HC;      // the first layer's x chunks padded to an even count (zero operands)

    extern __shared__ __attribute__((aligned(16))) char smem[];
    gu32x4* hb = reinterpret_cast<gu32x4*>(smem);              // [HC][hi|lo][64]  h_{t-1}
    gu32x4* rhb = hb + HC * 2 * 64;                             // [HC][2][64]      r (.) h_{t-1}
    gu32x4* xsb = rhb + HC * 2 * 64;                            // [HC][2][64]      this frame's input (the first layer uses KX <= 2 chunks)
    float* biasl = reinterpret_cast<float*>(xsb + HC * 2 * 64); // [3][H] + [16]
    const EpilogueLds epi = epilogue_carve(reinterpret_cast<char*>(biasl + 3 * H + 16 + 4));  // LAST only ([3H + 16 .. +4): the pipelined launch's flag)

    // the weight stream: buffer loads, lane offset in one VGPR, the operand index in the scalar offset (1 KiB per operand)
    const __amdgpu_buffer_rsrc_t w_rsrc = __builtin_amdgcn_make_buffer_rsrc(const_cast<uint4*>(p.w), (short)0, 0x7fffffff, 0x00020000);
    const int lane16 = lane * 16;
    auto wload = [&](int tile, int q, int c, int hl) -> gu32x4 {
        return __builtin_bit_cast(gu32x4, __builtin_amdgcn_raw_buffer_load_b128(w_rsrc, lane16, ((((tile * 3 + q) * KC + c) * 2 + hl)) * 1024, 0));
    };
    constexpr int kSysScope = 1 | 16;            // sc0 | sc1: past the non-coherent cache levels (the pipelined launch's seams)

    // kws_create folded the exponent scales into the weights (gates: -log2 e, candidate: 2 log2 e): the biases get them here
    for (int i = tid; i < 3 * H; i += 256) biasl[i] = p.bias[i] * (i < 2 * H ? -kLog2e : 2.0f * kLog2e);
    if (LAST && tid < 16) biasl[3 * H + tid] = p.bfc[tid];
    for (int i = tid; i < HC * 2 * 64; i += 256) xsb[i] = (gu32x4){0u, 0u, 0u, 0u};
    const f32x4* bl = reinterpret_cast<const f32x4*>(biasl);

    const bool do_reset = p.reset != nullptr && p.reset[b] != 0;
    const int len_s = p.seq_len ? p.seq_len[b] : T;
    f32x4 hreg[TPW];
#pragma unroll
    for (int j = 0; j < TPW; ++j) hreg[j] = do_reset ? splat4(0.f) : ld4(p.state_in + (size_t)b * H + (TPW * w + j) * 16 + 4 * g);
#pragma unroll
    for (int jj = 0; jj < CPW; ++jj) {
        gu32x4 hi, lo;
        g_split8(hreg[2 * jj], hreg[2 * jj + 1], hi, lo);
        hb[((CPW * w + jj) * 2 + 0) * 64 + lane] = hi;
        hb[((CPW * w + jj) * 2 + 1) * 64 + lane] = lo;
    }
    if (LAST && tid < 16) {
        const int bb = group * kStreamsPerGroup + tid;
        int pw = -1;
        if (bb < p.B && p.epi.prev_word && !(p.reset && p.reset[bb])) pw = p.epi.prev_word[bb];
        epi.carry[tid] = pw;
    }
    // the projection's operands: this wave's chunks of Wfc^T (hi, lo), resident
    gu32x4 wfc[CPW][2];
    if (LAST) {
#pragma unroll
        for (int jj = 0; jj < CPW; ++jj)
#pragma unroll
            for (int hl = 0; hl < 2; ++hl) wfc[jj][hl] = __builtin_bit_cast(gu32x4, p.wfc[((CPW * w + jj) * 2 + hl) * 64 + lane]);
    }

    // ---- input: FIRST: wave w brings streams 4w..4w+3 of the mel frame (one dwordx4 per lane), scales, clamps, splits and scatters
    // them into the B-operand image (gru_f16x3.hip's staging); above: the seam row of the frame, already split, in operand order ----
    const int XQ = FIRST ? p.I / 4 : 1;
    const int xl_row = lane / XQ, xl_q = lane % XQ;
    const bool xl_active = FIRST && lane < 4 * XQ;
    const float* mel_row = nullptr;
    if (FIRST) mel_row = p.x_mel + (size_t)min(group * kStreamsPerGroup + 4 * w + (xl_active ? xl_row : 0), p.B - 1) * T * p.I + xl_q * 4;
    const int xs_lane = ((((xl_q * 4) / 32) * 2 + 0) * 64 + (((xl_q * 4) % 32) / 8) * 16 + (4 * w + xl_row)) * 4 + ((xl_q * 4) % 8) / 2;   // dwords
    unsigned* const xsb_dw = reinterpret_cast<unsigned*>(xsb);
    f32x4 melv = splat4(0.f);
    auto mel_fetch = [&](int t_req) {
        if (xl_active) melv = ld4(mel_row + (size_t)(t_req < T ? t_req : T - 1) * p.I);
    };
    auto mel_commit = [&]() {
        if (xl_active) {
            f32x4 v = melv * kGMelScale;
#pragma unroll
            for (int e = 0; e < 4; ++e) v[e] = __builtin_fminf(__builtin_fmaxf(v[e], -kGHalfMax), kGHalfMax);
            unsigned h0, l0, h1, l1;
            g_split2((f32x2){v[0], v[1]}, h0, l0);
            g_split2((f32x2){v[2], v[3]}, h1, l1);
            *reinterpret_cast<uint2*>(xsb_dw + xs_lane) = make_uint2(h0, h1);
            *reinterpret_cast<uint2*>(xsb_dw + 64 * 4 + xs_lane) = make_uint2(l0, l1);
        }
    };
    // this group's [T][HC][2][64] x 16 B blocks of the seams
    const __amdgpu_buffer_rsrc_t xp_rsrc = __builtin_amdgcn_make_buffer_rsrc(
        FIRST ? const_cast<uint4*>(p.w) : const_cast<uint4*>(p.x_prev + (size_t)group * T * HC * 2 * 64), (short)0, 0x7fffffff, 0x00020000);
    const __amdgpu_buffer_rsrc_t ho_rsrc = __builtin_amdgcn_make_buffer_rsrc(
        LAST ? const_cast<uint4*>(p.w) : const_cast<uint4*>(p.h_out + (size_t)group * T * HC * 2 * 64), (short)0, 0x7fffffff, 0x00020000);
    constexpr int XF = HC * 2 / 4;                // seam operands (1 KiB) of a frame per wave
    gu32x4 xin[XF];
    auto seam_fetch = [&](int t_req) {          // the pipelined launch reads rows another CU has just written: system scope
        const int t = t_req < T ? t_req : T - 1;
#pragma unroll
        for (int i = 0; i < XF; ++i)
            xin[i] = __builtin_bit_cast(gu32x4, __builtin_amdgcn_raw_buffer_load_b128(xp_rsrc, lane16, (t * (HC * 2) + (w + 4 * i)) * 1024, PIPE ? kSysScope : 0));
    };
    auto seam_commit = [&]() {
#pragma unroll
        for (int i = 0; i < XF; ++i) xsb[(w + 4 * i) * 64 + lane] = xin[i];
    };

    if (FIRST) { mel_fetch(0); mel_commit(); mel_fetch(1); }
    else if (!PIPE) { seam_fetch(0); seam_commit(); seam_fetch(1); }
    __syncthreads();

    // ---- rows: all of the wave's tiles for one chunk, in ONE register image (24 operands at h = 256).  X rows carry the three
    // gates, H rows r and u, C rows two chunks of the candidate's recurrent part (so that every row keeps >= 16 KiB per wave in
    // flight).  Two images A / B ping-pong through the whole frame: every phase has an even number of rows (kws_create pads the
    // first layer's x-part to an even chunk count), starts in A, and the load that would run past a phase's end fetches the
    // NEXT phase's first row instead -- the weights do not depend on the data, so the stream never drains at a phase
    // boundary, a barrier or a frame boundary (the candidate's first row lands during the r / u activations, the next frame's
    // first x row during tanh / update).
    struct Row { gu32x4 a[TPW * 6]; };
    auto load_x = [&](Row& r, int c) {
#pragma unroll
        for (int j = 0; j < TPW; ++j)
#pragma unroll
            for (int q = 0; q < 3; ++q)
#pragma unroll
                for (int hl = 0; hl < 2; ++hl) r.a[(j * 3 + q) * 2 + hl] = wload(TPW * w + j, q, c, hl);
        __builtin_amdgcn_sched_barrier(0);
    };
    auto load_h = [&](Row& r, int m) {
#pragma unroll
        for (int j = 0; j < TPW; ++j)
#pragma unroll
            for (int q = 0; q < 2; ++q)
#pragma unroll
                for (int hl = 0; hl < 2; ++hl) r.a[(j * 2 + q) * 2 + hl] = wload(TPW * w + j, q, KX + m, hl);
        __builtin_amdgcn_sched_barrier(0);
    };
    auto load_c = [&](Row& r, int m2) {
#pragma unroll
        for (int k = 0; k < 2; ++k)
#pragma unroll
            for (int j = 0; j < TPW; ++j)
#pragma unroll
                for (int hl = 0; hl < 2; ++hl) r.a[(k * TPW + j) * 2 + hl] = wload(TPW * w + j, 2, KX + 2 * m2 + k, hl);
        __builtin_amdgcn_sched_barrier(0);
    };
    // Both images are always loaded or loading: a phase enters with its first TWO rows requested by the phase before it, every
    // image is re-requested the moment its MFMAs have been issued -- with the next row of the phase, or of the next phase.
    Row ra, rb;
    load_x(ra, 0);
    load_x(rb, 1);
    // layer-pipelined launch, layers above the first: has the layer below already published the NEXT frame?  Asked by one lane a
    // phase ahead (the answer travels through LDS, so the whole workgroup takes the same branch); if yes the frame's input is
    // fetched behind barrier #1 and the poll / fetch / barrier at the top of the next frame disappears
    int* const next_flag = reinterpret_cast<int*>(biasl + 3 * H + 16);
    bool have_x = false;

    for (int t = 0; t < T; ++t) {
        f32x4 am[TPW][3], al[TPW][3];
#pragma unroll
        for (int j = 0; j < TPW; ++j)
#pragma unroll
            for (int q = 0; q < 3; ++q) {
                am[j][q] = bl[(q * H + (TPW * w + j) * 16) / 4 + g];
                al[j][q] = splat4(0.f);
            }
        if (PIPE && !FIRST && !have_x) {
            // frame t of the layer below must have landed (its workgroup runs concurrently on another CU): every wave polls for
            // itself, bounded -- a protocol bug becomes a wrong answer plus an error flag, not a hung GPU (gru_kernels.hip)
            int spins = 0;
            while (__hip_atomic_load(p.epi.ready_in + group, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) <= t) {
                __builtin_amdgcn_s_sleep(2);
                if (++spins > (1 << 24)) { if (lane == 0) *reinterpret_cast<volatile int*>(p.epi.pipe_error) = 1; break; }
            }
            asm volatile("" ::: "memory");
            seam_fetch(t);
            seam_commit();
            __syncthreads();
        }
        int ahead = 0;
        if (PIPE && !FIRST && tid == 0 && t + 1 < T)
            ahead = __hip_atomic_load(p.epi.ready_in + group, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);       // (lands under the MFMAs below)
        // ---- x-part of r, u, c ----
#define KWS_G_MMA_X(R_, C_)                                                                             \
        {                                                                                                \
            const gu32x4 xh = xsb[((C_) * 2 + 0) * 64 + lane], xl = xsb[((C_) * 2 + 1) * 64 + lane];     \
            _Pragma("unroll") for (int j = 0; j < TPW; ++j)                                              \
                _Pragma("unroll") for (int q = 0; q < 3; ++q) {                                          \
                    am[j][q] = g_mfma(R_.a[(j * 3 + q) * 2 + 0], xh, am[j][q]);                          \
                    al[j][q] = g_mfma(R_.a[(j * 3 + q) * 2 + 1], xh, al[j][q]);                          \
                    al[j][q] = g_mfma(R_.a[(j * 3 + q) * 2 + 0], xl, al[j][q]);                          \
                }                                                                                        \
            __builtin_amdgcn_sched_barrier(0);                                                           \
        }
#pragma nounroll
        for (int c = 0; c < KX; c += 2) {                       // KX is even
            KWS_G_MMA_X(ra, c);
            if (c + 2 < KX) load_x(ra, c + 2); else load_h(ra, 0);
            KWS_G_MMA_X(rb, c + 1);
            if (c + 3 < KX) load_x(rb, c + 3); else load_h(rb, 1);
        }
#undef KWS_G_MMA_X
        // ---- recurrent part of r and u ----
#define KWS_G_MMA_H(R_, M_)                                                                             \
        {                                                                                                \
            const gu32x4 xh = hb[((M_) * 2 + 0) * 64 + lane], xl = hb[((M_) * 2 + 1) * 64 + lane];       \
            _Pragma("unroll") for (int j = 0; j < TPW; ++j)                                              \
                _Pragma("unroll") for (int q = 0; q < 2; ++q) {                                          \
                    am[j][q] = g_mfma(R_.a[(j * 2 + q) * 2 + 0], xh, am[j][q]);                          \
                    al[j][q] = g_mfma(R_.a[(j * 2 + q) * 2 + 1], xh, al[j][q]);                          \
                    al[j][q] = g_mfma(R_.a[(j * 2 + q) * 2 + 0], xl, al[j][q]);                          \
                }                                                                                        \
            __builtin_amdgcn_sched_barrier(0);                                                           \
        }
#pragma nounroll
        for (int m = 0; m < HC; m += 2) {                       // HC is even
            KWS_G_MMA_H(ra, m);
            if (m + 2 < HC) load_h(ra, m + 2);
            else {
                // ... the candidate's rows do not depend on the exchange below.  Pipelined producer: the seam rows of the frame
                // before must be written through before the counter moves behind barrier #1; draining HERE costs nothing -- all
                // that is still in flight is the row the next MFMAs wait for anyway -- whereas in front of the barrier it would
                // also wait for the two candidate rows requested now
                if (PIPE && !LAST) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
                load_c(ra, 0);
            }
            KWS_G_MMA_H(rb, m + 1);
            if (m + 3 < HC) load_h(rb, m + 3); else load_c(rb, 1);
        }
#undef KWS_G_MMA_H
        // ---- r; r (.) h split -> LDS (the u sigmoid waits for the candidate phase: only the update needs it) ----
#pragma unroll
        for (int jj = 0; jj < CPW; ++jj) {
            f32x4 rh[2];
#pragma unroll
            for (int k = 0; k < 2; ++k) {
                const int j = 2 * jj + k;
#pragma unroll
                for (int e = 0; e < 4; ++e) {
                    const float pr = __builtin_fmaf(al[j][0][e], kGLoInv, am[j][0][e]);
                    rh[k][e] = __builtin_amdgcn_rcpf(1.0f + __builtin_amdgcn_exp2f(pr)) * hreg[j][e];
                }
            }
            gu32x4 hi, lo;
            g_split8(rh[0], rh[1], hi, lo);
            rhb[((CPW * w + jj) * 2 + 0) * 64 + lane] = hi;
            rhb[((CPW * w + jj) * 2 + 1) * 64 + lane] = lo;
        }
        if (PIPE && !FIRST && tid == 0) *next_flag = ahead > t + 1 ? 1 : 0;
        __syncthreads();              // #1: r (.) h visible; hb and xsb fully consumed
        if (PIPE && !LAST && t > 0 && tid == 0)
            __hip_atomic_store(p.epi.ready_out + group, t, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);      // frames 0..t-1 are out (every wave drained its stores above)
        // the next frame's input: xsb is free now
        if (FIRST) { mel_commit(); mel_fetch(t + 2); }
        else if (!PIPE) { seam_commit(); seam_fetch(t + 2); }
        else {
            have_x = *next_flag != 0;
            if (have_x) seam_fetch(t + 1);           // committed behind the candidate MFMAs
        }
        // ---- recurrent part of the candidate; the u sigmoid rides in the stream's shadow ----
        f32x4 u[TPW];
#define KWS_G_MMA_C(R_, M2_)                                                                            \
        {                                                                                                \
            _Pragma("unroll") for (int k = 0; k < 2; ++k) {                                              \
                const gu32x4 xh = rhb[((2 * (M2_) + k) * 2 + 0) * 64 + lane], xl = rhb[((2 * (M2_) + k) * 2 + 1) * 64 + lane]; \
                _Pragma("unroll") for (int j = 0; j < TPW; ++j) {                                        \
                    am[j][2] = g_mfma(R_.a[(k * TPW + j) * 2 + 0], xh, am[j][2]);                        \
                    al[j][2] = g_mfma(R_.a[(k * TPW + j) * 2 + 1], xh, al[j][2]);                        \
                    al[j][2] = g_mfma(R_.a[(k * TPW + j) * 2 + 0], xl, al[j][2]);                        \
                }                                                                                        \
            }                                                                                            \
            __builtin_amdgcn_sched_barrier(0);                                                           \
        }
        // HC / 2 rows, an even count (HC = 4 or 8); the first pair peeled
        KWS_G_MMA_C(ra, 0);
        if (2 < HC / 2) load_c(ra, 2); else load_x(ra, 0);                    // ... the next frame's first x rows
#pragma unroll
        for (int j = 0; j < TPW; ++j)
#pragma unroll
            for (int e = 0; e < 4; ++e)
                u[j][e] = __builtin_amdgcn_rcpf(1.0f + __builtin_amdgcn_exp2f(__builtin_fmaf(al[j][1][e], kGLoInv, am[j][1][e])));
        KWS_G_MMA_C(rb, 1);
        if (3 < HC / 2) load_c(rb, 3); else load_x(rb, 1);
#pragma nounroll
        for (int m2 = 2; m2 < HC / 2; m2 += 2) {
            KWS_G_MMA_C(ra, m2);
            if (m2 + 2 < HC / 2) load_c(ra, m2 + 2); else load_x(ra, 0);
            KWS_G_MMA_C(rb, m2 + 1);
            if (m2 + 3 < HC / 2) load_c(rb, m2 + 3); else load_x(rb, 1);
        }
#undef KWS_G_MMA_C
        if (PIPE && !FIRST && have_x) seam_commit();
        // ---- tanh, update, split -> LDS, seam / projection ----
        const unsigned live = t < len_s ? 0xffffffffu : 0u;
        f32x4 fm = splat4(0.f), fl = splat4(0.f);
        if (LAST && w == 0) fm = bl[(3 * H) / 4 + g];
#pragma unroll
        for (int jj = 0; jj < CPW; ++jj) {
            f32x4 hout[2];
#pragma unroll
            for (int k = 0; k < 2; ++k) {
                const int j = 2 * jj + k;
#pragma unroll
                for (int e = 0; e < 4; ++e) {
                    const float pc = __builtin_fmaf(al[j][2][e], kGLoInv, am[j][2][e]);
                    const float c = __builtin_fmaf(__builtin_amdgcn_rcpf(1.0f + __builtin_amdgcn_exp2f(pc)), -2.0f, 1.0f);      // tanh
                    const float hn = __builtin_fmaf(u[j][e], hreg[j][e] - c, c);                                                 // c + u (h - c)
                    hreg[j][e] = bitsel(live, hn, hreg[j][e]);
                    hout[k][e] = bitsel(live, hn, 0.f);
                }
            }
            gu32x4 hi, lo;
            g_split8(hreg[2 * jj], hreg[2 * jj + 1], hi, lo);
            hb[((CPW * w + jj) * 2 + 0) * 64 + lane] = hi;
            hb[((CPW * w + jj) * 2 + 1) * 64 + lane] = lo;
            // the layer's OUTPUT row is zero past seq_len (dynamic_rnn), its state is copied through
            gu32x4 ohi = hi, olo = lo;
#pragma unroll
            for (int e = 0; e < 4; ++e) { ohi[e] &= live; olo[e] &= live; }
            if (!LAST) {
                const int frag = (t * HC + CPW * w + jj) * 2;
                __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(gi32x4, ohi), ho_rsrc, lane16, (frag + 0) * 1024, PIPE ? kSysScope : 0);
                __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(gi32x4, olo), ho_rsrc, lane16, (frag + 1) * 1024, PIPE ? kSysScope : 0);
            } else {
                fm = g_mfma(wfc[jj][0], ohi, fm);
                fl = g_mfma(wfc[jj][1], ohi, fl);
                fl = g_mfma(wfc[jj][0], olo, fl);
            }
        }
        if (LAST) {
            const f32x4 accf = fm + fl * kGLoInv;
            if (g < 2) *reinterpret_cast<f32x4*>(epi.pstage + (w * 16 + s) * 8 + 4 * g) = accf;
        }
        __syncthreads();              // #2: h(t), x(t+1), the partial logits visible
        if (LAST) {
            if (w == (t & 3)) epilogue_fold(epi, t, lane);
            if (((t + 1) & (kRingFrames - 1)) == 0 || t == T - 1) {
                const int t0 = t & ~(kRingFrames - 1);
                __syncthreads();
                epilogue_flush(p.epi, epi, group, t0, t - t0 + 1, w, lane, t == T - 1);
            }
        }
    }
    if (PIPE && !LAST) {                         // the last frame
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __syncthreads();
        if (tid == 0) __hip_atomic_store(p.epi.ready_out + group, T, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    }
    if (bvalid) {
#pragma unroll
        for (int j = 0; j < TPW; ++j)
            *reinterpret_cast<f32x4*>(p.state_out + (size_t)b * H + (TPW * w + j) * 16 + 4 * g) = hreg[j];
    }
}

template <int TPW, bool FIRST, bool LAST>
__global__ void __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(1, 1))) gru_layer_f16x3_generic(const GruF16Params p) {
    gru_f16x3_generic_body<TPW, FIRST, LAST, false>(p, blockIdx.x);
}

// Layer-pipelined launch: ONE grid of L x G workgroups, XCD-affine as gru_stack_generic_pipelined (block i -> XCD i % 8 serves
// layer (i % 8) % L: the workgroups behind one L2 all stream the same layer's 1.5 MiB).
template <int TPW>
__global__ void __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(1, 1))) gru_stack_f16x3_pipelined(const GruF16StackParams sp) {
    int layer, group;
    if (sp.xcd_affine) {
        const int xcd = blockIdx.x & 7, slot = blockIdx.x >> 3, per = 8 / sp.L;
        layer = xcd % sp.L;
        group = slot * per + xcd / sp.L;
        if (group >= sp.G) return;
    } else {
        layer = blockIdx.x / sp.G;
        group = blockIdx.x - layer * sp.G;
    }
    if (layer == 0) gru_f16x3_generic_body<TPW, true, false, true>(sp.layer[0], group);
    else if (layer == sp.L - 1) gru_f16x3_generic_body<TPW, false, true, true>(sp.layer[layer], group);
    else gru_f16x3_generic_body<TPW, false, false, true>(sp.layer[layer], group);
}

bool gru_f16x3_generic_supported(int hidden, int n_mel) { return (hidden == 128 || hidden == 256) && n_mel % 4 == 0 && n_mel >= 4 && n_mel <= 64; }

constexpr size_t kF16GenericOneWorkgroupPerCuLds = 82 * 1024;      // > 160 KB / 2

template <typename K>
static hipError_t launch_f16g(K kernel, const GruF16Params& p, size_t lds, hipStream_t st) {
    static LdsGrant granted;
    {
        const hipError_t e = grant_dynamic_lds(kernel, granted, lds);
        if (e != hipSuccess) return e;
    }
    const int groups = (p.B + kStreamsPerGroup - 1) / kStreamsPerGroup;
    hipLaunchKernelGGL(kernel, dim3(groups), dim3(256), lds, st, p);
    return hipGetLastError();
}

hipError_t launch_gru_layer_f16x3_generic(const GruF16Params& p, int hidden, bool first, bool last, hipStream_t st) {
    if (p.T <= 0 || p.B <= 0) return hipSuccess;
    // one workgroup per CU (each wants the whole register file)
    size_t lds = gru_f16x3_generic_lds_bytes(hidden, last);
    if (lds < kF16GenericOneWorkgroupPerCuLds) lds = kF16GenericOneWorkgroupPerCuLds;
#define KWS_F16G(TPW_) \
    do { \
        if (first && last) return launch_f16g(gru_layer_f16x3_generic<TPW_, true, true>, p, lds, st); \
        if (first) return launch_f16g(gru_layer_f16x3_generic<TPW_, true, false>, p, lds, st); \
        if (last) return launch_f16g(gru_layer_f16x3_generic<TPW_, false, true>, p, lds, st); \
        return launch_f16g(gru_layer_f16x3_generic<TPW_, false, false>, p, lds, st); \
    } while (0)
    if (hidden == 128) KWS_F16G(2);
    if (hidden == 256) KWS_F16G(4);
#undef KWS_F16G
    return hipErrorInvalidValue;
}

template <int TPW>
static hipError_t launch_f16_pipelined(const GruF16StackParams& sp, size_t lds, hipStream_t st) {
    static LdsGrant granted;
    {
        const hipError_t e = grant_dynamic_lds(gru_stack_f16x3_pipelined<TPW>, granted, lds);
        if (e != hipSuccess) return e;
    }
    const int per = sp.xcd_affine ? 8 / sp.L : 0;
    const int grid = sp.xcd_affine ? 8 * ((sp.G + per - 1) / per) : sp.G * sp.L;
    hipLaunchKernelGGL(gru_stack_f16x3_pipelined<TPW>, dim3(grid), dim3(256), lds, st, sp);
    return hipGetLastError();
}

hipError_t launch_gru_stack_f16x3_pipelined(const GruF16StackParams& sp, int hidden, hipStream_t st) {
    size_t lds = gru_f16x3_generic_lds_bytes(hidden, true);
    if (lds < kF16GenericOneWorkgroupPerCuLds) lds = kF16GenericOneWorkgroupPerCuLds;
    if (hidden == 128) return launch_f16_pipelined<2>(sp, lds, st);
    if (hidden == 256) return launch_f16_pipelined<4>(sp, lds, st);
    return hipErrorInvalidValue;
}

}  // namespace kws

// "f16x3": the fp32 GRU stack on the 16-bit matrix pipe at fp32 accuracy (precision = KWS_F16X3; secondary line of
// bench.py -- the headline stays the plain fp32 path).  Same semantics as gru_kernels.hip (models/rnn_ctc.py:155-165,
// 202-284; TF-1.x GRUCell), same boundary, same tolerances (logits within 1e-4 of the fp64 oracle; observed ~3e-6).
//
// Every matmul operand is split into two fp16 numbers that together carry 22 mantissa bits,
//     v = hi + 2^-11 lo,   hi = fp16(v),   lo = fp16((v - hi) * 2^11)        (both round-to-nearest-even),
// weights once at kws_create, activations on the fly (6 VALU instructions per register pair: v_cvt_pk_f16_f32,
// 2 x v_cvt_f32_f16, v_pk_add_f32, v_pk_mul_f32, v_cvt_pk_f16_f32).  A product then needs THREE v_mfma_f32_16x16x32_f16
// instead of eight v_mfma_f32_16x16x4_f32 of twice the duration (48 matrix-pipe cycles per 32 k instead of 256):
//     main += Wh Xh          lo += Wl Xh + Wh Xl          result = main + 2^-11 lo        (fp32 accumulators)
// fp16 x fp16 products are exact in fp32; the dropped term Wl Xl 2^-22 is below fp32's own rounding of the product.  The
// 2^11 scale keeps every lo operand in fp16's normal range (no reliance on how the matrix pipe treats subnormals).  bf16
// splits would need 3 + 3 pieces and six products for the same 24 bits; fp16's 11-bit pieces need two and three.
// Range: |hidden| <= 1; weights must be < 32768 in magnitude (kws_create checks); mel is pre-scaled by 2^-8 (and the
// x-part weights of the first layer by 2^8, both exact), so |mel| up to 1.6e7 is represented and larger values saturate.
//
// Weights are 4 bytes each again (hi + lo), so residency is the fp32 kernels': ONE LAYER per launch, the layers meet
// through a seam in HBM -- here already split, in B-operand order, so the layer above reads its input ready to use.
// One workgroup = 4 waves = 16 streams, wave w owns units [32w, 32w+32) of r, u, c and h' (two 16x16 tiles); the K
// permutation is gru_bf16.hip's: the wave's two C tiles, split and packed, ARE chunk w of the next B operand.
//   recurrent operands (48 per wave) + candidate x-part (<= 16)      AGPRs, fed to the MFMA directly ("a" constraint)
//   gate x-part                                                      first layer: AGPR/VGPR; above: r in VGPRs, u in LDS
#include <cstdlib>

#include "gru_device.h"

namespace kws {

namespace {

typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));
typedef _Float16 f16x2 __attribute__((ext_vector_type(2)));
typedef unsigned u32x4 __attribute__((ext_vector_type(4)));

constexpr float kLoScale = 2048.f, kLoInv = 1.f / 2048.f;
constexpr float kMelScale = 1.f / 256.f;       // kws_api.hip multiplies the first layer's x-part weights by 256
constexpr float kHalfMax = 65504.f;

__device__ __forceinline__ f16x8 as_f16x8(u32x4 v) { return __builtin_bit_cast(f16x8, v); }
__device__ __forceinline__ f32x4 mfma_f16(f16x8 a, f16x8 b, f32x4 c) {
    return __builtin_amdgcn_mfma_f32_16x16x32_f16(a, b, c, 0, 0, 0);
}
// Resident operands are pinned into AGPRs once (asm "+a" at load time); with MFMA results in VGPRs (csrc/Makefile builds this
// file with -amdgpu-mfma-vgpr-form, as gru_bf16.hip) hipcc then feeds the BUILTIN MFMA straight from the "a" registers -- no
// v_accvgpr_read copies (checked in the ISA) -- and, unlike inline-asm MFMAs, it sees every hazard and can interleave the
// matrix instructions with the activation arithmetic, which is what the frame loop below is arranged for.

__device__ __forceinline__ void split2(f32x2 x, unsigned& hi, unsigned& lo) {
    const f16x2 h = __builtin_convertvector(x, f16x2);
    const f32x2 r = (x - __builtin_convertvector(h, f32x2)) * kLoScale;
    const f16x2 l = __builtin_convertvector(r, f16x2);
    hi = __builtin_bit_cast(unsigned, h);
    lo = __builtin_bit_cast(unsigned, l);
}
// the wave's two C tiles -> its chunk of the next B operand, (hi, lo)
__device__ __forceinline__ void split8(const f32x4& a, const f32x4& b, u32x4& hi, u32x4& lo) {
    unsigned h[4], l[4];
    split2((f32x2){a[0], a[1]}, h[0], l[0]);
    split2((f32x2){a[2], a[3]}, h[1], l[1]);
    split2((f32x2){b[0], b[1]}, h[2], l[2]);
    split2((f32x2){b[2], b[3]}, h[3], l[3]);
    hi = (u32x4){h[0], h[1], h[2], h[3]};
    lo = (u32x4){l[0], l[1], l[2], l[3]};
}
__device__ __forceinline__ f32x4 combine(const f32x4& m, const f32x4& l) { return m + l * kLoInv; }

// Scheduling regions.  sched_barrier(0): nothing crosses.  interleave<N, M, V>: N times {M matrix instructions, then V VALU
// instructions} for the instructions of the enclosing region, in dependency order -- an MFMA occupies the matrix pipe for
// ~17 cycles during which the wave may issue VALU work of its own, but only when the two alternate in the stream (in-order issue).
__device__ __forceinline__ void region_fence() { __builtin_amdgcn_sched_barrier(0); }
template <int N, int M, int V>
__device__ __forceinline__ void interleave() {
#pragma unroll
    for (int i = 0; i < N; ++i) {
        __builtin_amdgcn_sched_group_barrier(0x008, M, 0);
        __builtin_amdgcn_sched_group_barrier(0x002, V, 0);
    }
}

enum { kInAgpr = 0, kInVgpr = 1, kInLds = 2 };
// where gate q's x-part operands of a layer with KX x-chunks live
template <int KX, bool FIRST>
constexpr int x_place(int q) {
    if (q == 2) return kInAgpr;                                  // candidate: 4 KX <= 16 operands
    if (FIRST) return (q == 0 || KX == 1) ? kInAgpr : kInVgpr;   // 48 + 12 KX <= 64 AGPR operands only for KX = 1
    return q == 0 ? kInVgpr : kInLds;
}

}  // namespace

size_t gru_f16x3_lds_bytes(int kx, bool first, bool last) {
    size_t n = 2 * 4 * 2 * 64 * 16;                      // hb, rhb
    n += (size_t)2 * kx * 2 * 64 * 16;                   // xsb, two slots
    if (!first) n += (size_t)4 * 16 * 64 * 16;           // u-gate x-part operands, per wave
    n += 3 * 128 * 4;                                    // biases
    if (last) n += kEpilogueLdsBytes;
    return n;
}

template <int KX, bool FIRST, bool LAST>
__global__ void __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(1, 1)))
gru_layer_f16x3(const GruF16Params p) {
    constexpr int H = 128, KC = KX + 4;
    const int tid = threadIdx.x;
    const int w = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int lane = tid & 63, g = lane >> 4, s = lane & 15;
    const int n_groups = (p.B + kStreamsPerGroup - 1) / kStreamsPerGroup;
    const int T = p.T;

    extern __shared__ __attribute__((aligned(16))) char smem[];
    u32x4* hb = reinterpret_cast<u32x4*>(smem);              // [4 chunks][hi|lo][64]   h_{t-1}
    u32x4* rhb = hb + 4 * 2 * 64;                             // [4][2][64]              r (.) h_{t-1}
    u32x4* xsb = rhb + 4 * 2 * 64;                            // [2 slots][KX][2][64]    input of frames t+1, t+2 (ping-pong)
    u32x4* wul = xsb + 2 * KX * 2 * 64;                         // !FIRST: [4 waves][2 tiles][KX][2][64]  u-gate x-part
    float* biasl = reinterpret_cast<float*>(wul + (FIRST ? 0 : 4 * 2 * KX * 2 * 64));   // [3][128]
    const EpilogueLds epi = epilogue_carve(reinterpret_cast<char*>(biasl + 3 * H));      // LAST only

    // ---- operands: [tile j][gate q][chunk][hi|lo]; table p.w is [8 tiles][3][KC][2][64 lanes] x 16 B, x chunks first ----
    const u32x4* wt = reinterpret_cast<const u32x4*>(p.w);
    auto wload = [&](int j, int q, int c, int hl) { return as_f16x8(wt[((((2 * w + j) * 3 + q) * KC + c) * 2 + hl) * 64 + lane]); };
    f16x8 wh[2][3][4][2];          // recurrent part: AGPRs
    f16x8 wx[2][3][KX][2];         // x-part: by x_place
#pragma unroll
    for (int j = 0; j < 2; ++j)
        static_for<0, 3>([&](auto q_) {
            constexpr int q = decltype(q_)::value;
            constexpr int place = x_place<KX, FIRST>(q);
#pragma unroll
            for (int m = 0; m < 4; ++m)
#pragma unroll
                for (int hl = 0; hl < 2; ++hl) {
                    wh[j][q][m][hl] = wload(j, q, KX + m, hl);
                    asm volatile("" : "+a"(wh[j][q][m][hl]));
                }
#pragma unroll
            for (int c = 0; c < KX; ++c)
#pragma unroll
                for (int hl = 0; hl < 2; ++hl) {
                    if constexpr (place == kInLds) {
                        wul[(((w * 2 + j) * KX + c) * 2 + hl) * 64 + lane] = wt[((((2 * w + j) * 3 + q) * KC + c) * 2 + hl) * 64 + lane];
                    } else {
                        wx[j][q][c][hl] = wload(j, q, c, hl);
                        if constexpr (place == kInAgpr) asm volatile("" : "+a"(wx[j][q][c][hl]));
                    }
                }
        });
    asm volatile("s_nop 7" ::: "memory");       // v_accvgpr_write -> MFMA SrcA distance
    f16x8 wfc[2];
    f32x4 bfc4 = splat4(0.f);
    if constexpr (LAST) {
        wfc[0] = as_f16x8(reinterpret_cast<const u32x4*>(p.wfc)[(w * 2 + 0) * 64 + lane]);
        wfc[1] = as_f16x8(reinterpret_cast<const u32x4*>(p.wfc)[(w * 2 + 1) * 64 + lane]);
        if (w == 0) bfc4 = ld4(p.bfc + 4 * g);
    }
    for (int i = tid; i < 3 * H; i += 256) biasl[i] = p.bias[i];
    for (int i = tid; i < 2 * KX * 2 * 64; i += 256) xsb[i] = (u32x4){0u, 0u, 0u, 0u};
    const f32x4* bl = reinterpret_cast<const f32x4*>(biasl);

    // ---- input fetch: FIRST: wave w brings streams 4w..4w+3 of the mel frame (one dwordx4 per lane), scales, splits and
    // scatters them into the B-operand image; above: wave w brings chunk w of the seam (hi, lo), already in operand order
    const int XQ = FIRST ? p.I / 4 : 1;
    const int xl_row = lane / XQ, xl_q = lane % XQ;
    const bool xl_active = lane < 4 * XQ;
    struct XF { f32x4 mel; u32x4 hi, lo; };
    const float4* mel_src = nullptr;
    const u32x4* seam_src = nullptr;
    unsigned* xs_hi = reinterpret_cast<unsigned*>(xsb) +
                      ((((xl_q * 4) / 32) * 2 + 0) * 64 + (((xl_q * 4) % 32) / 8) * 16 + (4 * w + xl_row)) * 4 + ((xl_q * 4) % 8) / 2;
    unsigned* xs_lo = xs_hi + 64 * 4;
    auto fetch = [&](XF& r, int t_req) {
        const int t = t_req < T ? t_req : T - 1;
        if constexpr (FIRST) {
            if (xl_active) {
                const float4 v = mel_src[(size_t)t * XQ];
                r.mel = (f32x4){v.x, v.y, v.z, v.w};
            }
        } else {
            r.hi = seam_src[((size_t)t * 4) * 2 * 64];
            r.lo = seam_src[((size_t)t * 4) * 2 * 64 + 64];
        }
    };
    auto commit = [&](const XF& r, int slot) {
        if constexpr (FIRST) {
            if (xl_active) {
                f32x4 v = r.mel * kMelScale;
#pragma unroll
                for (int e = 0; e < 4; ++e) v[e] = __builtin_fminf(__builtin_fmaxf(v[e], -kHalfMax), kHalfMax);
                unsigned h0, l0, h1, l1;
                split2((f32x2){v[0], v[1]}, h0, l0);
                split2((f32x2){v[2], v[3]}, h1, l1);
                *reinterpret_cast<uint2*>(xs_hi + slot * (KX * 2 * 64 * 4)) = make_uint2(h0, h1);
                *reinterpret_cast<uint2*>(xs_lo + slot * (KX * 2 * 64 * 4)) = make_uint2(l0, l1);
            }
        } else {
            xsb[slot * (KX * 2 * 64) + (w * 2 + 0) * 64 + lane] = r.hi;
            xsb[slot * (KX * 2 * 64) + (w * 2 + 1) * 64 + lane] = r.lo;
        }
    };

    for (int group = blockIdx.x; group < n_groups; group += gridDim.x) {
        const int b_raw = group * kStreamsPerGroup + s;
        const bool bvalid = b_raw < p.B;
        const int b = bvalid ? b_raw : p.B - 1;
        const bool do_reset = p.reset != nullptr && p.reset[b] != 0;
        const int len_s = p.seq_len ? p.seq_len[b] : T;
        f32x4 hreg[2];
#pragma unroll
        for (int j = 0; j < 2; ++j)
            hreg[j] = do_reset ? splat4(0.f) : ld4(p.state_in + (size_t)b * H + (2 * w + j) * 16 + 4 * g);
        {
            u32x4 hi, lo;
            split8(hreg[0], hreg[1], hi, lo);
            hb[(w * 2 + 0) * 64 + lane] = hi;
            hb[(w * 2 + 1) * 64 + lane] = lo;
        }
        if constexpr (LAST) {
            if (tid < 16) {
                const int bb = group * kStreamsPerGroup + tid;
                int pw = -1;
                if (bb < p.B && p.epi.prev_word && !(p.reset && p.reset[bb])) pw = p.epi.prev_word[bb];
                epi.carry[tid] = pw;
            }
        }
        if constexpr (FIRST) {
            const int xl_b = min(group * kStreamsPerGroup + 4 * w + (xl_active ? xl_row : 0), p.B - 1);
            mel_src = reinterpret_cast<const float4*>(p.x_mel + (size_t)xl_b * T * p.I) + xl_q;
        } else {
            seam_src = reinterpret_cast<const u32x4*>(p.x_prev) + ((size_t)group * T * 4 + w) * 2 * 64 + lane;
        }
        u32x4* seam_dst = nullptr;
        if constexpr (!LAST) seam_dst = reinterpret_cast<u32x4*>(p.h_out) + ((size_t)group * T * 4 + w) * 2 * 64 + lane;

        XF fl_a, fl_b;
        fl_a.mel = fl_b.mel = splat4(0.f);
        fl_a.hi = fl_a.lo = fl_b.hi = fl_b.lo = (u32x4){0u, 0u, 0u, 0u};
        __syncthreads();              // LDS tables / previous group's readers
        fetch(fl_a, 0);
        fetch(fl_b, 1);
        commit(fl_a, 0);              // x(0) -> slot 0
        commit(fl_b, 1);              // x(1) -> slot 1
        fetch(fl_b, 2);               // x(2): committed during frame 0
        fetch(fl_a, 3);               // x(3): committed during frame 1
        __syncthreads();

        // x-part of gate q for the frame whose input sits in xsb slot `slot`: accumulators start from the bias
        auto xpart = [&](auto q_, int slot, f32x4 (&m)[2][3], f32x4 (&l)[2][3]) {
            constexpr int q = decltype(q_)::value;
            constexpr int place = x_place<KX, FIRST>(q);
            const u32x4* xs = xsb + slot * (KX * 2 * 64);
#pragma unroll
            for (int j = 0; j < 2; ++j) {
                m[j][q] = bl[(q * H + (2 * w + j) * 16) / 4 + g];
                l[j][q] = splat4(0.f);
            }
#pragma unroll
            for (int c = 0; c < KX; ++c) {
                const f16x8 Bh = as_f16x8(xs[(c * 2 + 0) * 64 + lane]), Bl = as_f16x8(xs[(c * 2 + 1) * 64 + lane]);
                f16x8 whi[2], wlo[2];
#pragma unroll
                for (int j = 0; j < 2; ++j) {
                    if constexpr (place == kInLds) {
                        whi[j] = as_f16x8(wul[(((w * 2 + j) * KX + c) * 2 + 0) * 64 + lane]);
                        wlo[j] = as_f16x8(wul[(((w * 2 + j) * KX + c) * 2 + 1) * 64 + lane]);
                    } else {
                        whi[j] = wx[j][q][c][0];
                        wlo[j] = wx[j][q][c][1];
                    }
                }
#pragma unroll
                for (int j = 0; j < 2; ++j) m[j][q] = mfma_f16(whi[j], Bh, m[j][q]);
#pragma unroll
                for (int j = 0; j < 2; ++j) l[j][q] = mfma_f16(wlo[j], Bh, l[j][q]);
#pragma unroll
                for (int j = 0; j < 2; ++j) l[j][q] = mfma_f16(whi[j], Bl, l[j][q]);
            }
        };
        // the frame loop is software-pipelined: iteration t finds bias + x-part(t) in (nm, nl) and computes x-part(t+1), which
        // depends on nothing of the recurrence, beside the activation arithmetic -- matrix pipe and VALU are separate units
        f32x4 nm[2][3], nl[2][3];
        static_for<0, 3>([&](auto q_) { xpart(q_, 0, nm, nl); });

        auto frame = [&](int t, XF& fl_commit) {
            f32x4 am[2][3], al[2][3];
#pragma unroll
            for (int j = 0; j < 2; ++j)
#pragma unroll
                for (int q = 0; q < 3; ++q) { am[j][q] = nm[j][q]; al[j][q] = nl[j][q]; }
            const int nslot = (t + 1) & 1;
            // ---------------- recurrent part of r and u ----------------
#pragma unroll
            for (int m = 0; m < 4; ++m) {
                const f16x8 Bh = as_f16x8(hb[(m * 2 + 0) * 64 + lane]), Bl = as_f16x8(hb[(m * 2 + 1) * 64 + lane]);
#pragma unroll
                for (int j = 0; j < 2; ++j) { am[j][0] = mfma_f16(wh[j][0][m][0], Bh, am[j][0]); am[j][1] = mfma_f16(wh[j][1][m][0], Bh, am[j][1]); }
#pragma unroll
                for (int j = 0; j < 2; ++j) { al[j][0] = mfma_f16(wh[j][0][m][1], Bh, al[j][0]); al[j][1] = mfma_f16(wh[j][1][m][1], Bh, al[j][1]); }
#pragma unroll
                for (int j = 0; j < 2; ++j) { al[j][0] = mfma_f16(wh[j][0][m][0], Bl, al[j][0]); al[j][1] = mfma_f16(wh[j][1][m][0], Bl, al[j][1]); }
            }
            region_fence();
            // ---------------- r, r (.) h -> LDS; beside it the next frame's x-part of r and u ----------------
            f32x4 u[2], rh[2];
#pragma unroll
            for (int j = 0; j < 2; ++j) {
                const f32x4 pre = combine(am[j][0], al[j][0]);
                const f32x2 r_lo = sigmoid2((f32x2){pre[0], pre[1]});
                const f32x2 r_hi = sigmoid2((f32x2){pre[2], pre[3]});
                const f32x2 a = r_lo * (f32x2){hreg[j][0], hreg[j][1]};
                const f32x2 c2 = r_hi * (f32x2){hreg[j][2], hreg[j][3]};
                rh[j] = (f32x4){a.x, a.y, c2.x, c2.y};
            }
            {
                u32x4 hi, lo;
                split8(rh[0], rh[1], hi, lo);
                rhb[(w * 2 + 0) * 64 + lane] = hi;
                rhb[(w * 2 + 1) * 64 + lane] = lo;
            }
            xpart(std::integral_constant<int, 0>{}, nslot, nm, nl);
            xpart(std::integral_constant<int, 1>{}, nslot, nm, nl);
            interleave<12 * KX, 1, 3>();      // 12 KX matrix instructions beside the ~70 VALU instructions of the r path
            region_fence();
            lds_barrier();            // #1: r (.) h visible; hb fully consumed
            region_fence();
            // ---------------- candidate (recurrent part) with the u sigmoid in its shadow ----------------
#pragma unroll
            for (int m = 0; m < 4; ++m) {
                const f16x8 Bh = as_f16x8(rhb[(m * 2 + 0) * 64 + lane]), Bl = as_f16x8(rhb[(m * 2 + 1) * 64 + lane]);
#pragma unroll
                for (int j = 0; j < 2; ++j) am[j][2] = mfma_f16(wh[j][2][m][0], Bh, am[j][2]);
#pragma unroll
                for (int j = 0; j < 2; ++j) al[j][2] = mfma_f16(wh[j][2][m][1], Bh, al[j][2]);
#pragma unroll
                for (int j = 0; j < 2; ++j) al[j][2] = mfma_f16(wh[j][2][m][0], Bl, al[j][2]);
            }
#pragma unroll
            for (int j = 0; j < 2; ++j) {
                const f32x4 pre = combine(am[j][1], al[j][1]);
                const f32x2 u_lo = sigmoid2((f32x2){pre[0], pre[1]});
                const f32x2 u_hi = sigmoid2((f32x2){pre[2], pre[3]});
                u[j] = (f32x4){u_lo.x, u_lo.y, u_hi.x, u_hi.y};
            }
            interleave<24, 1, 1>();
            region_fence();
            // ---------------- state update, hand-over; beside it the next frame's x-part of c ----------------
            const unsigned live = t < len_s ? 0xffffffffu : 0u;
#pragma unroll
            for (int j = 0; j < 2; ++j) {
                const f32x4 pre = combine(am[j][2], al[j][2]);
#pragma unroll
                for (int h2 = 0; h2 < 2; ++h2) {
                    const f32x2 c = tanh2((f32x2){pre[2 * h2], pre[2 * h2 + 1]});
                    const f32x2 uu = {u[j][2 * h2], u[j][2 * h2 + 1]};
                    const f32x2 hh = {hreg[j][2 * h2], hreg[j][2 * h2 + 1]};
                    const f32x2 hn = (1.0f - uu) * c + uu * hh;
                    hreg[j][2 * h2] = bitsel(live, hn.x, hh.x);
                    hreg[j][2 * h2 + 1] = bitsel(live, hn.y, hh.y);
                }
            }
            u32x4 hhi, hlo;
            split8(hreg[0], hreg[1], hhi, hlo);
            hb[(w * 2 + 0) * 64 + lane] = hhi;
            hb[(w * 2 + 1) * 64 + lane] = hlo;
            xpart(std::integral_constant<int, 2>{}, nslot, nm, nl);
            // the layer's OUTPUT row is zero past seq_len (dynamic_rnn), its state is copied through
#pragma unroll
            for (int e = 0; e < 4; ++e) { hhi[e] &= live; hlo[e] &= live; }
            if constexpr (!LAST) {
                seam_dst[((size_t)t * 4) * 2 * 64] = hhi;
                seam_dst[((size_t)t * 4) * 2 * 64 + 64] = hlo;
            }
            commit(fl_commit, t & 1);        // x(t+2): its slot was last read during frame t-1
            fetch(fl_commit, t + 4);
            if constexpr (LAST) {
                // dense: this wave's 32 units are exactly k-chunk w of Wfc^T
                f32x4 fm = bfc4, fl = splat4(0.f);
                fm = mfma_f16(wfc[0], as_f16x8(hhi), fm);
                fl = mfma_f16(wfc[1], as_f16x8(hhi), fl);
                fl = mfma_f16(wfc[0], as_f16x8(hlo), fl);
                const f32x4 accf = combine(fm, fl);
                if (g < 2) *reinterpret_cast<f32x4*>(epi.pstage + (w * 16 + s) * 8 + 4 * g) = accf;
            }
            interleave<6 * KX, 1, 4>();
            region_fence();
            lds_barrier();            // #2: h(t), x(t+2), the partial logits visible
            region_fence();
            if constexpr (LAST) {
                if (w == (t & 3)) epilogue_fold(epi, t, lane);
                if (((t + 1) & (kRingFrames - 1)) == 0 || t == T - 1) {
                    const int t0 = t & ~(kRingFrames - 1);
                    lds_barrier();
                    epilogue_flush(p.epi, epi, group, t0, t - t0 + 1, w, lane, t == T - 1);
                }
            }
        };
        // fl_b holds x(t+2) on even frames, fl_a on odd ones
        for (int t = 0; t < T; t += 2) {
            frame(t, fl_b);
            if (t + 1 < T) frame(t + 1, fl_a);
        }
        if (bvalid) {
#pragma unroll
            for (int j = 0; j < 2; ++j)
                *reinterpret_cast<f32x4*>(p.state_out + (size_t)b * H + (2 * w + j) * 16 + 4 * g) = hreg[j];
        }
    }
}

bool gru_f16x3_supported(int hidden, int n_mel) { return hidden == 128 && n_mel % 4 == 0 && n_mel >= 4 && n_mel <= 64; }

template <int KX, bool FIRST, bool LAST>
static hipError_t launch_f16x3(const GruF16Params& p, hipStream_t st) {
    const size_t lds = gru_f16x3_lds_bytes(KX, FIRST, LAST);
    static LdsGrant granted;
    {
        const hipError_t e = grant_dynamic_lds(gru_layer_f16x3<KX, FIRST, LAST>, granted, lds);
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
    hipLaunchKernelGGL((gru_layer_f16x3<KX, FIRST, LAST>), dim3(groups < cus ? groups : cus), dim3(256), lds, st, p);
    return hipGetLastError();
}

hipError_t launch_gru_layer_f16x3(const GruF16Params& p, bool first, bool last, hipStream_t st) {
    if (p.T <= 0 || p.B <= 0) return hipSuccess;
    if (first) {
        const int kx = (p.I + 31) / 32;
        if (kx == 1) return last ? launch_f16x3<1, true, true>(p, st) : launch_f16x3<1, true, false>(p, st);
        if (kx == 2) return last ? launch_f16x3<2, true, true>(p, st) : launch_f16x3<2, true, false>(p, st);
        return hipErrorInvalidValue;
    }
    return last ? launch_f16x3<4, false, true>(p, st) : launch_f16x3<4, false, false>(p, st);
}

}  // namespace kws

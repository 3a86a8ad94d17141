// "f16x3": the fp32 GRU stack on the 16-bit matrix pipe at fp32 accuracy (precision = KWS_F16X3; secondary line of
// bench.py -- the headline stays the plain fp32 path).  Same semantics as gru_kernels.hip (models/rnn_ctc.py:155-165,
// 202-284; TF-1.x GRUCell), same boundary, same tolerances (logits within 1e-4 of the fp64 oracle; observed ~3e-6).
//
// Every matmul operand is split into two fp16 numbers that together carry 22 mantissa bits,
//     v = hi + lo,   hi = fp16(v),   lo = fp16(v - hi)                        (both round-to-nearest-even),
// weights once at kws_create, activations on the fly (per pair of values: one v_cvt_pk_f16_f32 for hi, then per value ONE
// v_fma_mixlo/hi_f16 that reads the fp16 half of hi, multiplies it by the inline constant -1.0, adds the value and writes the
// rounded fp16 half of lo in place -- bit-identical to subtract, convert).  A product then needs THREE v_mfma_f32_16x16x32_f16
// instead of eight v_mfma_f32_16x16x4_f32 of twice the duration (~51 matrix-pipe cycles per 32 k instead of 256):
//     acc += Wh Xh + Wl Xh + Wh Xl                                              (ONE fp32 accumulator)
// fp16 x fp16 products are exact in fp32; the dropped term Wl Xl is below fp32's own rounding of the product.  Round 6: the
// lo pieces sit at their OWN magnitude.  Rounds 4-5 scaled them by 2^11 (to keep them in fp16's normal range) and therefore
// kept a second accumulator per product, folded in with an fma per value before every activation -- 24 + 16 VALU
// instructions per frame and wave and 48 registers that the in-order-issue-bound frame loop could not spare (the last-layer
// kernels sat at 512 registers with 6-19 spills; now 468-504 and none; -3 % per 4096 x 300 step, -5 % per 16384 x 22-frame call).
// Unscaled, a lo piece below 2^-14 is an fp16 subnormal with absolute precision 2^-25: a value keeps max(2^-23 |v|, 2^-25), i.e.
// fp32's own rounding down to |v| = 1/4 and a 3e-8 absolute floor below -- nothing against the 1e-4 bar for hidden values in
// [-1, 1] and weights of O(0.1) (max |dlogit| against the fp32 kernels 6.4e-6, as before; every stream's token sequence identical).
// The path relies on the matrix pipe multiplying fp16 subnormals exactly (it does on gfx950: tests/test_gpu_parity.py feeds
// mel magnitudes down to 1e-6, whose pieces are all subnormal; kws_selftest runs the same kernels).  The ONE exception is the mel
// frame: pre-scaled by 2^-8 it is small against its weights (scaled by 2^8), where a 2^-25 floor would cost 1e-5 on a
// pre-activation -- so the first layer's x-part keeps the 2^11-scaled lo pieces (input and weights) and lo accumulators of its own,
// folded in once per gate.  bf16 splits would need 3 + 3 pieces and six products for the same 24 bits; fp16's 11-bit pieces need
// two and three.
// Range: |hidden| <= 1; weights must be < 64 in magnitude (kws_create checks); mel is pre-scaled by 2^-8 (and the
// x-part weights of the first layer by 2^8, both exact), so |mel| up to 1.6e7 is represented and larger values saturate.
// The exponent scales of the activations (sigmoid: -log2 e, tanh: 2 log2 e) are folded into the packed weights and biases,
// so a pre-activation is the argument of v_exp_f32 as it stands.
//
// Weights are 4 bytes each again (hi + lo), so residency is the fp32 kernels': ONE LAYER per launch, the layers meet
// through a seam in HBM -- here already split, in B-operand order, so the layer above reads its input ready to use.
// One workgroup = 4 waves = 16 streams, wave w owns units [32w, 32w+32) of r, u, c and h' (two 16x16 tiles); the K
// permutation is gru_bf16.hip's: the wave's two C tiles, split and packed, ARE chunk w of the next B operand.
//   recurrent operands (48 per wave) + candidate x-part (<= 16)      AGPRs (64 operands = all 256)
//   gate x-part            first layer: AGPR / VGPR; above: 4 operands in VGPRs, 28 streamed from LDS through two
//                          4-operand register sets
//
// The frame loop is a static schedule.  Per frame a wave issues 147 (upper layers) / 111 (first layer) MFMAs of ~17
// cycles and ~130 (first layer: ~155) VALU instructions; an MFMA runs in the matrix pipe while the wave issues VALU work of its own, but issue
// is in order, so the two only overlap when they ALTERNATE in the instruction stream.  The recurrence fixes a critical
// chain  gates_h MFMAs -> r sigmoid -> r(.)h split -> LDS -> barrier -> cand_h MFMAs -> tanh, update, split -> LDS ->
// barrier;  the next frame's x-part (72 / 36 MFMAs, independent of the recurrence) is the filler that is woven, one MFMA
// per three VALU instructions, into the two activation phases and into the LDS round trips around the barriers; the u
// sigmoid rides under the candidate MFMAs, the x stream's setup reads and (first layer) the mel conversion under the gate
// MFMAs.  The weave is written out below with compile-time indices (streams G, Cm, X of MFMAs; E of LDS reads; R, U, Cc, Mq
// of single scalar VALU instructions) and pinned with sched_barrier(0) after every element; the compiler still
// allocates registers, inserts the waitcnts and sees every hazard (builtin MFMAs only: with MFMA results in VGPRs --
// csrc/Makefile builds this file with -amdgpu-mfma-vgpr-form, as gru_bf16.hip -- hipcc feeds operands pinned into AGPRs
// ("+a" at load time) to the builtin directly, no v_accvgpr_read copies; checked in the ISA).
#include <cstddef>
#include <cstdio>
#include <cstdlib>

#include "gru_device.h"
#include "window_device.h"

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
// the mel input: lo scaled by 2^11 (its own accumulators in the first layer: see the header)
__device__ __forceinline__ void split2(f32x2 x, unsigned& hi, unsigned& lo) {
    const f16x2 h = __builtin_convertvector(x, f16x2);
    const f32x2 r = (x - __builtin_convertvector(h, f32x2)) * kLoScale;
    const f16x2 l = __builtin_convertvector(r, f16x2);
    hi = __builtin_bit_cast(unsigned, h);
    lo = __builtin_bit_cast(unsigned, l);
}
// hidden values (|v| <= 1): lo = fp16(v - hi), UNSCALED
__device__ __forceinline__ void split2u(f32x2 x, unsigned& hi, unsigned& lo) {
    const f16x2 h = __builtin_convertvector(x, f16x2);
    const f32x2 r = x - __builtin_convertvector(h, f32x2);
    const f16x2 l = __builtin_convertvector(r, f16x2);
    hi = __builtin_bit_cast(unsigned, h);
    lo = __builtin_bit_cast(unsigned, l);
}
// the wave's two C tiles -> its chunk of the next B operand, (hi, lo)
__device__ __forceinline__ void split8(const f32x4& a, const f32x4& b, u32x4& hi, u32x4& lo) {
    unsigned h[4], l[4];
    split2u((f32x2){a[0], a[1]}, h[0], l[0]);
    split2u((f32x2){a[2], a[3]}, h[1], l[1]);
    split2u((f32x2){b[0], b[1]}, h[2], l[2]);
    split2u((f32x2){b[2], b[3]}, h[3], l[3]);
    hi = (u32x4){h[0], h[1], h[2], h[3]};
    lo = (u32x4){l[0], l[1], l[2], l[3]};
}

// nothing is scheduled across this point: the weave below stays as written
__device__ __forceinline__ void pin() { __builtin_amdgcn_sched_barrier(0); }

// NA elements of stream a and NB of stream b, evenly merged, each pinned in place; the streams' own indices start at A0 /
// B0.  fa / fb take std::integral_constant<int, index>.
template <int NA, int A0, int NB, int B0, class FA, class FB>
__device__ __forceinline__ void zip(FA&& fa, FB&& fb) {
    static_for<0, NA + NB>([&](auto i_) {
        constexpr int i = decltype(i_)::value;
        constexpr int b_before = (int)((long long)i * NB / (NA + NB)), b_after = (int)((long long)(i + 1) * NB / (NA + NB));
        if constexpr (b_after > b_before) fb(std::integral_constant<int, B0 + b_before>{});
        else fa(std::integral_constant<int, A0 + (i - b_before)>{});
        pin();
    });
}
template <int N, int I0, class F>
__device__ __forceinline__ void run(F&& f) {
    static_for<0, N>([&](auto i_) { f(std::integral_constant<int, I0 + decltype(i_)::value>{}); pin(); });
}

enum { kInAgpr = 0, kInVgpr = 1, kInLds = 2 };
// where the x-part operands of gate q, chunk c live
template <int KX, bool FIRST>
constexpr int x_place(int q, int c) {
    if (q == 2) return kInAgpr;                                  // candidate: 4 KX <= 16 operands
    if (FIRST) return (q == 0 || KX == 1) ? kInAgpr : kInVgpr;   // 48 + 12 KX <= 64 AGPR operands only for KX = 1
    return (q == 0 && c < 1) ? kInVgpr : kInLds;
}
// LDS-resident (gate, chunk) groups in the order the x stream meets them (chunk-major, r before u): index or -1
template <int KX, bool FIRST>
constexpr int lds_group(int q, int c) {
    if (x_place<KX, FIRST>(q, c) != kInLds) return -1;
    int k = 0;
    for (int cc = 0; cc < KX; ++cc)
        for (int qq = 0; qq < 2; ++qq) {
            if (cc == c && qq == q) return k;
            if (x_place<KX, FIRST>(qq, cc) == kInLds) ++k;
        }
    return -1;
}
template <int KX, bool FIRST>
constexpr int lds_groups() {
    int k = 0;
    for (int cc = 0; cc < KX; ++cc)
        for (int qq = 0; qq < 2; ++qq)
            if (x_place<KX, FIRST>(qq, cc) == kInLds) ++k;
    return k;
}

#ifdef KWS_F16_TIMING       // tools/build_variant.sh timing -DKWS_F16_TIMING; tools/exp_f16_timing.py: cycles per phase of the frame loop
                            // (every stamp is an s_memtime + s_waitcnt: ~100 cycles that land in the NEXT interval)
__device__ long long* g_timing = nullptr;
#define KWS_STAMP(i) do { const long long now_ = __builtin_readcyclecounter(); tsum[i] += now_ - tlast; tlast = now_; } while (0)
#else
#define KWS_STAMP(i) do {} while (0)
#endif

}  // namespace

size_t gru_f16x3_lds_bytes(int kx, bool first, bool last, bool window = false) {
    size_t n = 2 * 4 * 2 * 64 * 16;                      // hb, rhb
    n += (size_t)(first ? 3 : 2) * kx * 2 * 64 * 16;     // xsb: three slots in the first layer, two above
    if (first) n += 512;                                 // dump row for the idle lanes' mel stores
    if (!first) n += (size_t)4 * 7 * 4 * 64 * 16;        // gate x-part operands streamed from LDS: 7 groups of 4 per wave
    n += 3 * 128 * 4 + 16 * 4;                           // biases
    if (last) n += kEpilogueLdsBytes;
    if (window) n += kWinTailWordsBytes;
    return n;
}

// WINDOW (LAST only): the decode-window step of the stream manager rides at the end of every group (window_device.h)
template <int KX, bool FIRST, bool LAST, bool MASKED, bool WINDOW = false>
__global__ void __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(1, 1)))
gru_layer_f16x3(const GruF16Params p) {
    static_assert(!WINDOW || LAST, "the window tail belongs to the last layer");
    constexpr int H = 128, KC = KX + 4;
    constexpr int NG = lds_groups<KX, FIRST>();          // LDS-resident operand groups per wave (0 or 7)
    constexpr int NX = 18 * KX;                          // MFMAs of one frame's x-part
    static_assert(FIRST ? NG == 0 : NG == 7, "gru_f16x3_lds_bytes assumes 7 streamed groups above the first layer");
    const int tid = threadIdx.x;
    const int w = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int lane = tid & 63, g = lane >> 4, s = lane & 15;
    const int n_groups = (p.B + kStreamsPerGroup - 1) / kStreamsPerGroup;
    const int T = p.T;

    extern __shared__ __attribute__((aligned(16))) char smem[];
    u32x4* hb = reinterpret_cast<u32x4*>(smem);              // [4 chunks][hi|lo][64]   h_{t-1}
    u32x4* rhb = hb + 4 * 2 * 64;                             // [4][2][64]              r (.) h_{t-1}
    constexpr int NS = FIRST ? 3 : 2;                         // xsb slots: frame t's input sits in slot t % NS
    u32x4* xsb = rhb + 4 * 2 * 64;                            // [NS slots][KX][2][64]
    u32x4* wul = xsb + NS * KX * 2 * 64 + (FIRST ? 32 : 0);                      // !FIRST: [4 waves][NG groups][tile j][hi|lo][64]
    float* biasl = reinterpret_cast<float*>(wul + 4 * NG * 4 * 64);    // [3][128] + [16] class bias
    EpilogueLds epi = epilogue_carve(reinterpret_cast<char*>(biasl + 3 * H + 16));      // LAST only
    if constexpr (WINDOW) epi.cwords = reinterpret_cast<int8_t*>(reinterpret_cast<char*>(biasl + 3 * H + 16) + kEpilogueLdsBytes);
    const uint8_t* win_dl = reinterpret_cast<const uint8_t*>(epi.cwords) + 16 * kWinTailWordsStride;     // WINDOW only: the label matcher
    constexpr size_t kWinOffset = offsetof(GruF16Params, epi) + offsetof(GruLayerParams, win);
    if constexpr (WINDOW) window_tail_prepare(window_tail_params_from_kernarg(kWinOffset), const_cast<uint8_t*>(win_dl), tid);   // (visible after the group loop's first barrier)

    // ---- operands: [tile j][gate q][chunk][hi|lo]; table p.w is [8 tiles][3][KC][2][64 lanes] x 16 B, x chunks first ----
    const u32x4* wt_tab = reinterpret_cast<const u32x4*>(p.w);
    auto wload = [&](int j, int q, int c, int hl) { return as_f16x8(wt_tab[((((2 * w + j) * 3 + q) * KC + c) * 2 + hl) * 64 + lane]); };
    f16x8 wh[2][3][4][2];          // recurrent part: AGPRs
    f16x8 wx[2][3][KX][2];         // x-part: by x_place
    // all loads first, the "+a" pins afterwards: a pin right behind its load makes the compiler wait for that load before it
    // issues the next one (64 serial L2 round trips: +10 us per launch, a sixth of a 22-frame call)
#pragma unroll
    for (int j = 0; j < 2; ++j)
        static_for<0, 3>([&](auto q_) {
            constexpr int q = decltype(q_)::value;
#pragma unroll
            for (int m = 0; m < 4; ++m)
#pragma unroll
                for (int hl = 0; hl < 2; ++hl) wh[j][q][m][hl] = wload(j, q, KX + m, hl);
            static_for<0, KX>([&](auto c_) {
                constexpr int c = decltype(c_)::value;
                if constexpr (x_place<KX, FIRST>(q, c) != kInLds) {
#pragma unroll
                    for (int hl = 0; hl < 2; ++hl) wx[j][q][c][hl] = wload(j, q, c, hl);
                }
            });
        });
#pragma unroll
    for (int j = 0; j < 2; ++j)
        static_for<0, 3>([&](auto q_) {
            constexpr int q = decltype(q_)::value;
            static_for<0, KX>([&](auto c_) {
                constexpr int c = decltype(c_)::value;
                if constexpr (x_place<KX, FIRST>(q, c) == kInLds) {
                    constexpr int k = lds_group<KX, FIRST>(q, c);
#pragma unroll
                    for (int hl = 0; hl < 2; ++hl)
                        wul[((w * NG + k) * 4 + j * 2 + hl) * 64 + lane] = wt_tab[((((2 * w + j) * 3 + q) * KC + c) * 2 + hl) * 64 + lane];
                }
            });
        });
#pragma unroll
    for (int j = 0; j < 2; ++j)
        static_for<0, 3>([&](auto q_) {
            constexpr int q = decltype(q_)::value;
#pragma unroll
            for (int m = 0; m < 4; ++m)
#pragma unroll
                for (int hl = 0; hl < 2; ++hl) {
                    f16x8& op = wh[j][q][m][hl];
                    asm volatile("" : "+a"(op));
                }
            static_for<0, KX>([&](auto c_) {
                constexpr int c = decltype(c_)::value;
                if constexpr (x_place<KX, FIRST>(q, c) == kInAgpr) {
#pragma unroll
                    for (int hl = 0; hl < 2; ++hl) {
                        f16x8& op = wx[j][q][c][hl];          // (named: clang does not capture a variable that only an asm operand names)
                        asm volatile("" : "+a"(op));
                    }
                }
            });
        });
    // LAST: the projection's operands (this wave's k-chunk of Wfc^T, hi and lo) are re-read every frame (2 KiB per wave, L1/L2
    // hits, requested a phase ahead): eight registers the upper layers do not have to spare
    const u32x4* wfc_src = reinterpret_cast<const u32x4*>(p.wfc) + (w * 2) * 64;        // uniform
    // kws_create folded the exponent scales into the weights (gates: -log2 e, candidate: 2 log2 e), so a pre-activation IS the
    // argument of exp2; the biases get the same factors here
    for (int i = tid; i < 3 * H; i += 256) biasl[i] = p.bias[i] * (i < 2 * H ? -kLog2e : 2.0f * kLog2e);
    if (LAST && tid < 16) biasl[3 * H + tid] = w == 0 ? p.bfc[tid] : 0.f;       // the class bias enters through wave 0's partial logits
    for (int i = tid; i < NS * KX * 2 * 64; i += 256) xsb[i] = (u32x4){0u, 0u, 0u, 0u};
    const f32x4* bl = reinterpret_cast<const f32x4*>(biasl);
    const u32x4* wul_w = wul + w * (NG * 4 * 64) + lane;

    // ---- input fetch: FIRST: wave w brings streams 4w..4w+3 of the mel frame (one dwordx4 per lane), scales, splits and
    // scatters them into the B-operand image; above: wave w brings chunk w of the seam (hi, lo), already in operand order.
    // Every global address is a wave-uniform base (scalar registers) plus a 32-bit lane offset: per-lane 64-bit pointers
    // would cost this kernel a dozen of the vector registers it does not have.
    const int XQ = FIRST ? p.I / 4 : 1;
    const int xl_row = lane / XQ, xl_q = lane % XQ;
    const bool xl_active = lane < 4 * XQ;
    struct XF { f32x4 mel; u32x4 hi, lo; };
    const float* mel_base = nullptr;       // FIRST: &mel[16 group][0][0], uniform
    unsigned mel_lane = 0;                 // FIRST: this lane's (row within the group, quarter) offset in floats
    const u32x4* seam_base = nullptr;      // above: &seam[group][0][chunk w][0][0], uniform
    const int xs_lane = ((((xl_q * 4) / 32) * 2 + 0) * 64 + (((xl_q * 4) % 32) / 8) * 16 + (4 * w + xl_row)) * 4 + ((xl_q * 4) % 8) / 2;   // dwords
    unsigned* const xsb_dw = reinterpret_cast<unsigned*>(xsb);
    const int xs_dump = NS * KX * 2 * 64 * 4 + 2 * lane;      // FIRST: 512 B behind the slots, where idle lanes' stores go
    auto fetch = [&](XF& r, int t_req) {
        const int t = t_req < T ? t_req : T - 1;
        if constexpr (FIRST) {
            if (xl_active) {
                const float4 v = *reinterpret_cast<const float4*>(mel_base + (size_t)t * p.I + mel_lane);
                r.mel = (f32x4){v.x, v.y, v.z, v.w};
            }
        } else {
            const u32x4* src = seam_base + (size_t)t * (4 * 2 * 64);
            r.hi = src[lane];
            r.lo = src[64 + lane];
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
                *reinterpret_cast<uint2*>(xsb_dw + slot * (KX * 2 * 64 * 4) + xs_lane) = make_uint2(h0, h1);
                *reinterpret_cast<uint2*>(xsb_dw + slot * (KX * 2 * 64 * 4) + 64 * 4 + xs_lane) = make_uint2(l0, l1);
            }
        } else {
            xsb[slot * (KX * 2 * 64) + (w * 2 + 0) * 64 + lane] = r.hi;
            xsb[slot * (KX * 2 * 64) + (w * 2 + 1) * 64 + lane] = r.lo;
        }
    };

    // The activation streams are single instructions; a 32-bit literal doubles an instruction's size (8 bytes), and a lone
    // wave per SIMD is fed instructions at a limited rate, so the constants live in scalar registers (VOP2 encodings, 4 bytes).
    float cLoInv = kLoInv, cNegTwo = -2.0f;
    asm volatile("" : "+s"(cLoInv), "+s"(cNegTwo));

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
            const int row = min(group * kStreamsPerGroup + 4 * w + (xl_active ? xl_row : 0), p.B - 1) - group * kStreamsPerGroup;
            mel_base = p.x_mel + (size_t)group * kStreamsPerGroup * T * p.I;
            mel_lane = (unsigned)row * (unsigned)(T * p.I) + (unsigned)xl_q * 4u;
        } else {
            seam_base = reinterpret_cast<const u32x4*>(p.x_prev) + ((size_t)group * T * 4 + w) * 2 * 64;
        }
        u32x4* seam_out = nullptr;       // !LAST: &seam[group][0][chunk w][0][0], uniform
        if constexpr (!LAST) seam_out = reinterpret_cast<u32x4*>(p.h_out) + ((size_t)group * T * 4 + w) * 2 * 64;

        XF fl;
        fl.mel = splat4(0.f);
        fl.hi = fl.lo = (u32x4){0u, 0u, 0u, 0u};
        __syncthreads();              // LDS tables / previous group's readers
        {
            XF f1 = fl;
            fetch(fl, 0);             // both requests in flight together: two serial memory round trips cost ~2 us per launch
            fetch(f1, 1);
            commit(fl, 0);            // x(0) -> slot 0
            commit(f1, 1);            // x(1) -> slot 1
        }
        fetch(fl, 2);                 // x(2): committed during frame 0
        __syncthreads();

        // ---- the streams ------------------------------------------------------------------------------------------
        // accumulators of two frames in flight: set t & 1 belongs to frame t.  [set][tile j][gate q]
        f32x4 am[2][2][3], al[2][2][3];
        // X: the x-part of the frame with parity PN (input in the xsb slot xs_ptr[PN] points at), 18 MFMAs per chunk: gate
        // r, u, c; per gate the three products over the two tiles.  Operands streamed from LDS arrive through two 4-operand
        // register sets, the input chunks through two (hi, lo) pairs; element i also issues the loads that elements
        // ~12..18 further on need.
        f16x8 xb[2][2], wtmp[2][2][2];
        int xs_off[2] = {lane, lane};            // element offset of the slot the x stream of each parity reads, + lane
        // E: what the x stream of parity PN needs before its first MFMA, one LDS read per element -- the accumulators' start values
        // (the biases; the lo accumulators start from the MFMA's inline 0), the first two input chunks, the first two streamed
        // operand groups.  These ride under the G MFMAs of the frame before, a whole phase ahead of their first use.
        constexpr int NE0 = 6, NE1 = 2 * (KX < 2 ? KX : 2), NE2 = 4 * (NG < 2 ? NG : 2), NE = NE0 + NE1 + NE2;
        auto E = [&](auto pn_, auto i_) {
            constexpr int PN = decltype(pn_)::value, i = decltype(i_)::value;
            if constexpr (i < NE0) {
                constexpr int q = i / 2, j = i % 2;
                am[PN][j][q] = bl[(q * H + (2 * w + j) * 16) / 4 + g];
                if constexpr (FIRST) al[PN][j][q] = splat4(0.f);
            } else if constexpr (i < NE0 + NE1) {
                constexpr int c = (i - NE0) / 2, hl = (i - NE0) % 2;
                xb[c][hl] = as_f16x8(xsb[xs_off[PN] + (c * 2 + hl) * 64]);
            } else {
                constexpr int k = (i - NE0 - NE1) / 4, jh = (i - NE0 - NE1) % 4;
                wtmp[k][jh >> 1][jh & 1] = as_f16x8(wul_w[(k * 4 + jh) * 64]);
            }
        };
        auto xbegin = [&](auto pn_, int slot) {   // un-woven form (prologue)
            constexpr int PN = decltype(pn_)::value;
            xs_off[PN] = slot * (KX * 2 * 64) + lane;
            static_for<0, NE>([&](auto i_) { E(pn_, i_); });
        };
        auto X = [&](auto pn_, auto i_) {
            constexpr int PN = decltype(pn_)::value, i = decltype(i_)::value;
            constexpr int c = i / 18, r = i % 18, q = r / 6, sweep = (r % 6) / 2, j = r % 2;
            constexpr int place = x_place<KX, FIRST>(q, c);
            constexpr int k = lds_group<KX, FIRST>(q, c);
            const f16x8 B = xb[c & 1][sweep == 2 ? 1 : 0];
            f16x8 W;
            if constexpr (place == kInLds) W = wtmp[k & 1][j][sweep == 1 ? 1 : 0];
            else W = wx[j][q][c][sweep == 1 ? 1 : 0];
            // the first layer's x-part is the mel frame, whose lo piece (and the lo piece of its weights) carries the 2^11 scale:
            // those cross terms have accumulators of their own; everything else goes into ONE accumulator
            if constexpr (sweep == 0 || !FIRST) am[PN][j][q] = mfma_f16(W, B, am[PN][j][q]);
            else al[PN][j][q] = mfma_f16(W, B, al[PN][j][q]);
            if constexpr (place == kInLds && r % 6 == 5 && k + 2 < NG) {       // this group's register set is free again
#pragma unroll
                for (int jh = 0; jh < 4; ++jh) wtmp[k & 1][jh >> 1][jh & 1] = as_f16x8(wul_w[((k + 2) * 4 + jh) * 64]);
            }
            if constexpr (r == 17 && c + 2 < KX) {
                xb[c & 1][0] = as_f16x8(xsb[xs_off[PN] + ((c + 2) * 2 + 0) * 64]);
                xb[c & 1][1] = as_f16x8(xsb[xs_off[PN] + ((c + 2) * 2 + 1) * 64]);
            }
        };
        // G: recurrent part of r and u, 12 MFMAs per chunk of h(t-1); Cm: of the candidate, 6 per chunk of r (.) h.
        f16x8 hB[2][2];
        auto hread = [&](const u32x4* src, int m, int buf) {
            hB[buf][0] = as_f16x8(src[(m * 2 + 0) * 64 + lane]);
            hB[buf][1] = as_f16x8(src[(m * 2 + 1) * 64 + lane]);
        };
        auto G = [&](auto pc_, auto i_) {
            constexpr int PC = decltype(pc_)::value, i = decltype(i_)::value;
            constexpr int m = i / 12, r = i % 12, sweep = r / 4, j = (r % 4) / 2, q = r % 2;
            const f16x8 B = hB[m & 1][sweep == 2 ? 1 : 0];
            const f16x8 W = wh[j][q][m][sweep == 1 ? 1 : 0];
            am[PC][j][q] = mfma_f16(W, B, am[PC][j][q]);
            if constexpr (r == 11 && m + 2 < 4) hread(hb, m + 2, m & 1);
        };
        auto Cm = [&](auto pc_, auto i_) {
            constexpr int PC = decltype(pc_)::value, i = decltype(i_)::value;
            constexpr int m = i / 6, r = i % 6, sweep = r / 2, j = r % 2;
            const f16x8 B = hB[m & 1][sweep == 2 ? 1 : 0];
            const f16x8 W = wh[j][2][m][sweep == 1 ? 1 : 0];
            am[PC][j][2] = mfma_f16(W, B, am[PC][j][2]);
            if constexpr (r == 5 && m + 2 < 4) hread(rhb, m + 2, m & 1);
        };
        // R, U, Cc: the activation arithmetic, ONE scalar VALU instruction per element (packed fp32 instructions beside MFMAs
        // are slower than the two scalar ones they replace), the lane's eight values (tile j, row e) in flight side by side so
        // that no element waits for the one before it: stage-major order.  The fp16 pack stages have four elements.
        // Split of a value v with hi already packed:  lo16 = fp16(fma(hi16 as f32, -2^11, v * 2^11))  -- v_fma_mixlo/hi_f16 reads
        // the fp16 half directly and writes the rounded fp16 half in place (bit-identical to the subtract-scale-convert form).
        float va[8], vb[8], uu[8];      // (vb: the candidate path's h - c)
        unsigned phi[4], plo[4] = {0u, 0u, 0u, 0u};
        auto mix_lo = [&](auto un_, float m) {          // lo16 = fp16(m - hi16): the value's own hi half, times the inline constant -1.0, plus the value
            constexpr int un = decltype(un_)::value;
            const unsigned hi = phi[un >> 1];
            unsigned d = plo[un >> 1];
            if constexpr ((un & 1) == 0) asm("v_fma_mixlo_f16 %0, %1, -1.0, %2 op_sel:[0,0,0] op_sel_hi:[1,0,0]" : "=v"(d) : "v"(hi), "v"(m));
            else asm("v_fma_mixhi_f16 %0, %1, -1.0, %2 op_sel:[1,0,0] op_sel_hi:[1,0,0]" : "+v"(d) : "v"(hi), "v"(m));
            plo[un >> 1] = d;
        };
        constexpr int S0 = FIRST ? 1 : 0;             // elementwise stages in front of the exp: the first layer folds its x-part's lo accumulators in
        auto R = [&](auto pc_, auto i_) {
            constexpr int PC = decltype(pc_)::value, i = decltype(i_)::value;
            // stages: [0 fma: first layer] 1 exp, 2 add, 3 rcp, 4 mul h | 5 pack hi (4) | 7 lo half
            constexpr int NE8 = (S0 + 4) * 8;
            constexpr int st = i < NE8 ? i / 8 + (1 - S0) : i < NE8 + 4 ? 5 : 7;
            constexpr int un = i < NE8 ? i % 8 : i < NE8 + 4 ? i - NE8 : (i - NE8 - 4) % 8;
            if constexpr (st == 5) phi[un] = __builtin_bit_cast(unsigned, __builtin_convertvector((f32x2){va[2 * un], va[2 * un + 1]}, f16x2));
            else {
                constexpr int j = un >> 2, e = un & 3;
                if constexpr (st == 0) va[un] = __builtin_fmaf(al[PC][j][0][e], cLoInv, am[PC][j][0][e]);
                else if constexpr (st == 1) va[un] = __builtin_amdgcn_exp2f(FIRST ? va[un] : am[PC][j][0][e]);
                else if constexpr (st == 2) va[un] = va[un] + 1.0f;
                else if constexpr (st == 3) va[un] = __builtin_amdgcn_rcpf(va[un]);
                else if constexpr (st == 4) va[un] = va[un] * hreg[j][e];
                else mix_lo(std::integral_constant<int, un>{}, va[un]);
            }
        };
        constexpr int NR = (S0 + 4) * 8 + 4 + 8;      // 44 (52 in the first layer)
        auto U = [&](auto pc_, auto i_) {
            constexpr int PC = decltype(pc_)::value, i = decltype(i_)::value, st = i / 8 + (1 - S0), un = i % 8, j = un >> 2, e = un & 3;
            if constexpr (st == 0) uu[un] = __builtin_fmaf(al[PC][j][1][e], cLoInv, am[PC][j][1][e]);
            else if constexpr (st == 1) uu[un] = __builtin_amdgcn_exp2f(FIRST ? uu[un] : am[PC][j][1][e]);
            else if constexpr (st == 2) uu[un] = uu[un] + 1.0f;
            else uu[un] = __builtin_amdgcn_rcpf(uu[un]);
        };
        constexpr int NU = (S0 + 3) * 8;              // 24 (32 in the first layer)
        unsigned live = 0u;
        constexpr int NCS = S0 + 6 + (MASKED ? 1 : 0);           // elementwise stages of the candidate path before the split
        auto Cc = [&](auto pc_, auto i_) {
            constexpr int PC = decltype(pc_)::value, i = decltype(i_)::value;
            // stages: [0 fma: first layer] 1 exp, 2 add, 3 rcp, 4 fma (tanh), 5 sub, 6 fma (update) [, 7 select: MASKED] | pack hi (4) | lo half
            constexpr int st = i < NCS * 8 ? i / 8 + (1 - S0) : i < NCS * 8 + 4 ? 100 : 102;
            constexpr int un = i < NCS * 8 ? i % 8 : i < NCS * 8 + 4 ? i - NCS * 8 : (i - NCS * 8 - 4) % 8;
            if constexpr (st == 100) phi[un] = __builtin_bit_cast(unsigned, __builtin_convertvector((f32x2){hreg[un >> 1][2 * (un & 1)], hreg[un >> 1][2 * (un & 1) + 1]}, f16x2));
            else {
                constexpr int j = un >> 2, e = un & 3;
                if constexpr (st == 0) va[un] = __builtin_fmaf(al[PC][j][2][e], cLoInv, am[PC][j][2][e]);
                else if constexpr (st == 1) va[un] = __builtin_amdgcn_exp2f(FIRST ? va[un] : am[PC][j][2][e]);
                else if constexpr (st == 2) va[un] = va[un] + 1.0f;
                else if constexpr (st == 3) va[un] = __builtin_amdgcn_rcpf(va[un]);
                else if constexpr (st == 4) va[un] = __builtin_fmaf(va[un], cNegTwo, 1.0f);                     // tanh
                else if constexpr (st == 5) vb[un] = hreg[j][e] - va[un];
                else if constexpr (st == 6) {
                    if constexpr (MASKED) va[un] = __builtin_fmaf(uu[un], vb[un], va[un]);                      // c + u (h - c)
                    else hreg[j][e] = __builtin_fmaf(uu[un], vb[un], va[un]);
                } else if constexpr (st == 7) hreg[j][e] = bitsel(live, va[un], hreg[j][e]);                  // copy-through past seq_len
                else mix_lo(std::integral_constant<int, un>{}, hreg[j][e]);
            }
        };
        constexpr int NC = NCS * 8 + 4 + 8;           // 60 (68 in the first layer; + 8 with the mask)
        // FIRST: the next-but-one mel frame (one dwordx4 per lane, streams 4w..4w+3) -> scaled, clamped, split, scattered into
        // its xsb slot; 24 scalar VALU elements that ride under the G MFMAs, then the two LDS stores
        float mv[4], mr[4];
        unsigned mh[2], ml[2];
        auto Mq = [&](auto i_) {
            constexpr int i = decltype(i_)::value;
            // stages: 0 scale (4), 1 clamp (4), 2 pack hi (2), 3 unpack (4), 4 sub (4), 5 mul (4), 6 pack lo (2)
            constexpr int st = i < 8 ? i / 4 : i < 10 ? 2 : i < 22 ? 3 + (i - 10) / 4 : 6;
            constexpr int un = i < 8 ? i % 4 : i < 10 ? i - 8 : i < 22 ? (i - 10) % 4 : i - 22;
            if constexpr (st == 0) mv[un] = fl.mel[un] * kMelScale;
            else if constexpr (st == 1) mv[un] = __builtin_amdgcn_fmed3f(mv[un], -kHalfMax, kHalfMax);
            else if constexpr (st == 2) mh[un] = __builtin_bit_cast(unsigned, __builtin_convertvector((f32x2){mv[2 * un], mv[2 * un + 1]}, f16x2));
            else if constexpr (st == 3) mr[un] = (float)__builtin_bit_cast(f16x2, mh[un >> 1])[un & 1];
            else if constexpr (st == 4) mr[un] = mv[un] - mr[un];
            else if constexpr (st == 5) mr[un] = mr[un] * kLoScale;
            else ml[un] = __builtin_bit_cast(unsigned, __builtin_convertvector((f32x2){mr[2 * un], mr[2 * un + 1]}, f16x2));
        };
        constexpr int NM = FIRST ? 24 : 0;
        // how many of the NX x-part MFMAs go where: beside the r path; behind the r (.) h store while it lands (before barrier
        // 1); into the LDS round trip behind barrier 1; beside the candidate path; behind the h store (before barrier 2); and
        // (the tail) into the round trip behind barrier 2 at the top of the next frame
        constexpr int XR = NX * 20 / 72, XW1 = NX * 6 / 72, XB1 = NX * 8 / 72, XC = NX * 26 / 72, XW2 = NX * 6 / 72,
                      XB2 = NX - XR - XW1 - XB1 - XC - XW2;

        // prologue: frame 0's x-part up to its tail
        {
            constexpr auto P0 = std::integral_constant<int, 0>{};
            xbegin(P0, 0);
            run<NX - XB2, 0>([&](auto i_) { X(P0, i_); });
        }
#ifdef KWS_F16_TIMING
        long long tsum[16] = {0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0};
        long long tlast = __builtin_readcyclecounter();
#endif
        int slot1 = 1 % NS, slot2 = 2 % NS;      // xsb slots of frames t+1 and t+2

        auto frame = [&](auto pc_, int t) {
            constexpr int PC = decltype(pc_)::value, PN = PC ^ 1;
            constexpr auto pc = std::integral_constant<int, PC>{};
            constexpr auto pn = std::integral_constant<int, PN>{};
            live = t < len_s ? 0xffffffffu : 0u;
#ifdef KWS_F16_TIMING
            tlast = __builtin_readcyclecounter();
#endif
            // ---- LAST: the previous frame's four partial logit vectors -> one row of the 16-frame ring.  Every wave folds its own
            // four streams (lane & 7 = (stream, half); the other lanes repeat the same reads and the same store), reads first,
            // sum behind the x tail: nobody falls a whole LDS round trip behind the others, as a single folding wave would
            f32x4 fr[4];
            const int fold_s = 4 * w + ((lane & 7) >> 1), fold_h = lane & 1;
            if constexpr (LAST) {
#pragma unroll
                for (int k = 0; k < 4; ++k) fr[k] = *reinterpret_cast<const f32x4*>(epi.pstage + (k * 16 + fold_s) * 8 + 4 * fold_h);
            }
            // ---- h(t-1) on its way; meanwhile the tail of this frame's own x-part ----
            hread(hb, 0, 0);
            hread(hb, 1, 1);
            pin();
            run<XB2, NX - XB2>([&](auto i_) { X(pc, i_); });
            if constexpr (LAST) {
                if (t > 0) *reinterpret_cast<f32x4*>(epi.lring + ((((t - 1) & (kRingFrames - 1)) * 16 + fold_s) * 8 + 4 * fold_h)) = ((fr[0] + fr[1]) + fr[2]) + fr[3];
                pin();
            }
            KWS_STAMP(0);
            if constexpr (LAST) {
                // the previous 16 frames' logits leave here, not at the end of frame t-1: between the x stream's tail and its next
                // start the fewest registers are live, and the flush (softmax, decode rule, stores) needs ~60 of its own
#ifndef KWS_ABL_NOFLUSH      // experiment builds only (tools/build_variant.sh): what the periodic flush costs
                if (t > 0 && (t & (kRingFrames - 1)) == 0) {
                    lds_barrier();
                    epilogue_flush<true>(p.epi, epi, group, t - kRingFrames, kRingFrames, w, lane, false);
                }
#endif
            }
            KWS_STAMP(9);
            // the next frame's x-part starts beside the r path: its accumulators (bias), first input chunks and operand groups
            // are requested now, a whole MFMA phase ahead
            xs_off[PN] = slot1 * (KX * 2 * 64) + lane;
            pin();
            // ---- recurrent part of r and u (48 MFMAs); in their shadow: the E reads, FIRST: x(t+2) is prepared ----
            if constexpr (FIRST) {
                zip<48, 0, NE + NM, 0>([&](auto i_) { G(pc, i_); }, [&](auto i_) {
                    constexpr int i = decltype(i_)::value;
                    if constexpr (i < NE) E(pn, i_);
                    else Mq(std::integral_constant<int, i - NE>{});
                });
                // every lane stores (the idle ones into a dump row behind the slots): a conditional store would let the compiler
                // sink the whole stream into the branch, out of the MFMAs' shadow
                *reinterpret_cast<uint2*>(xsb_dw + (xl_active ? slot2 * (KX * 2 * 64 * 4) + xs_lane : xs_dump)) = make_uint2(mh[0], mh[1]);
                *reinterpret_cast<uint2*>(xsb_dw + (xl_active ? slot2 * (KX * 2 * 64 * 4) + 64 * 4 + xs_lane : xs_dump)) = make_uint2(ml[0], ml[1]);
                fetch(fl, t + 3);
                pin();
            } else {
                zip<48, 0, NE, 0>([&](auto i_) { G(pc, i_); }, [&](auto i_) { E(pn, i_); });
            }
            KWS_STAMP(1);
            // ---- r, r (.) h, its split -> LDS; woven in: the head of the next frame's x-part ----
            zip<NR, 0, XR, 0>([&](auto i_) { R(pc, i_); }, [&](auto i_) { X(pn, i_); });
            KWS_STAMP(2);
            rhb[(w * 2 + 0) * 64 + lane] = (u32x4){phi[0], phi[1], phi[2], phi[3]};
            rhb[(w * 2 + 1) * 64 + lane] = (u32x4){plo[0], plo[1], plo[2], plo[3]};
            pin();
            run<XW1, XR>([&](auto i_) { X(pn, i_); });       // while the store lands
            lds_barrier();            // #1: r (.) h visible; hb fully consumed; the slot of x(t) fully consumed
            pin();
            KWS_STAMP(3);
            hread(rhb, 0, 0);
            hread(rhb, 1, 1);
            if constexpr (!FIRST) {
                // x(t+2) -> the slot this frame's x-part has just released; request x(t+3)
                commit(fl, slot2);
                fetch(fl, t + 3);
            }
            pin();
            run<XB1, XR + XW1>([&](auto i_) { X(pn, i_); });
            KWS_STAMP(4);
            // ---- candidate (24 MFMAs) with the u sigmoid in its shadow ----
            zip<24, 0, NU, 0>([&](auto i_) { Cm(pc, i_); }, [&](auto i_) { U(pc, i_); });
            KWS_STAMP(5);
            // ---- tanh, state update, split -> LDS, seam / projection; woven in: more of the next frame's x-part ----
            f16x8 wfc[2];
            if constexpr (LAST) {
                wfc[0] = as_f16x8(wfc_src[lane]);
                wfc[1] = as_f16x8(wfc_src[64 + lane]);
                pin();
            }
            zip<NC, 0, XC, XR + XW1 + XB1>([&](auto i_) { Cc(pc, i_); }, [&](auto i_) { X(pn, i_); });
            KWS_STAMP(6);
            u32x4 hhi = (u32x4){phi[0], phi[1], phi[2], phi[3]}, hlo = (u32x4){plo[0], plo[1], plo[2], plo[3]};
            hb[(w * 2 + 0) * 64 + lane] = hhi;
            hb[(w * 2 + 1) * 64 + lane] = hlo;
            pin();
            // the layer's OUTPUT row is zero past seq_len (dynamic_rnn), its state is copied through
            if constexpr (MASKED) {
#pragma unroll
                for (int e = 0; e < 4; ++e) { hhi[e] &= live; hlo[e] &= live; }
            }
            if constexpr (!LAST) {
                u32x4* dst = seam_out + (size_t)t * (4 * 2 * 64);
                dst[lane] = hhi;
                dst[64 + lane] = hlo;
            }
            if constexpr (LAST) {
                // dense: this wave's 32 units are exactly k-chunk w of Wfc^T
                f32x4 fm = w == 0 ? bl[(3 * H) / 4 + g] : splat4(0.f), fl2 = splat4(0.f);
                fm = mfma_f16(wfc[0], as_f16x8(hhi), fm);
                fl2 = mfma_f16(wfc[1], as_f16x8(hhi), fl2);
                fl2 = mfma_f16(wfc[0], as_f16x8(hlo), fl2);
                pin();
                run<XW2, XR + XW1 + XB1 + XC>([&](auto i_) { X(pn, i_); });     // while the h store lands and the projection drains
                const f32x4 accf = fm + fl2;
                if (g < 2) *reinterpret_cast<f32x4*>(epi.pstage + (w * 16 + s) * 8 + 4 * g) = accf;
            } else {
                run<XW2, XR + XW1 + XB1 + XC>([&](auto i_) { X(pn, i_); });
            }
            slot1 = slot2;
            slot2 = slot2 + 1 == NS ? 0 : slot2 + 1;
            pin();
            KWS_STAMP(7);
            lds_barrier();            // #2: h(t), x(t+2), the partial logits visible
            pin();
            KWS_STAMP(8);
            if constexpr (LAST) {
                if (t == T - 1) {          // the call's last frame and block (a full one included): no next frame to do it in
                    const int t0 = t & ~(kRingFrames - 1);
                    const int fs = 4 * w + ((lane & 7) >> 1), fh = lane & 1;
                    f32x4 v = *reinterpret_cast<const f32x4*>(epi.pstage + (0 * 16 + fs) * 8 + 4 * fh);
#pragma unroll
                    for (int k = 1; k < 4; ++k) v += *reinterpret_cast<const f32x4*>(epi.pstage + (k * 16 + fs) * 8 + 4 * fh);
                    *reinterpret_cast<f32x4*>(epi.lring + (((t & (kRingFrames - 1)) * 16 + fs) * 8 + 4 * fh)) = v;
                    lds_barrier();
                    epilogue_flush<true>(p.epi, epi, group, t0, t - t0 + 1, w, lane, true);
                }
            }
            KWS_STAMP(9);
        };
        for (int t = 0; t < T; t += 2) {
            frame(std::integral_constant<int, 0>{}, t);
            if (t + 1 < T) frame(std::integral_constant<int, 1>{}, t + 1);
        }
#ifdef KWS_F16_TIMING
        if (g_timing && lane == 0 && group < 4)
            for (int i = 0; i < 16; ++i) g_timing[((FIRST ? 0 : 16) + group * 4 + w) * 16 + i] = tsum[i];
#endif
        if (bvalid) {
#pragma unroll
            for (int j = 0; j < 2; ++j)
                *reinterpret_cast<f32x4*>(p.state_out + (size_t)b * H + (2 * w + j) * 16 + 4 * g) = hreg[j];
        }
        if constexpr (WINDOW) {
            // detector.py:195-209 for this group's 16 streams: the call's frame words wait in epi.cwords (final flush), the
            // scratch is hb | rhb, which nobody reads after the last frame
            __syncthreads();
            const WindowTail win = window_tail_params_from_kernarg(kWinOffset);
            WindowTailRegs<2> wreq;
            window_tail_request<2>(win, p.B, group * kStreamsPerGroup, tid, wreq);
            window_tail<2>(win, p.B, group * kStreamsPerGroup, T, epi.cwords, kWinTailWordsStride, win_dl, reinterpret_cast<char*>(hb), tid, wreq);
            __syncthreads();
        }
    }
}

bool gru_f16x3_vgpr_form() {
#ifdef KWS_F16X3_VGPR_FORM
    return true;
#else
    return false;
#endif
}

bool gru_f16x3_supported(int hidden, int n_mel) { return hidden == 128 && n_mel % 4 == 0 && n_mel >= 4 && n_mel <= 64; }

template <int KX, bool FIRST, bool LAST, bool MASKED, bool WINDOW = false>
static hipError_t launch_f16x3m(const GruF16Params& p, hipStream_t st) {
    const size_t lds = gru_f16x3_lds_bytes(KX, FIRST, LAST, WINDOW);
    static LdsGrant granted;
    {
        const hipError_t e = grant_dynamic_lds(gru_layer_f16x3<KX, FIRST, LAST, MASKED, WINDOW>, granted, lds);
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
#ifdef KWS_F16_TIMING
    static long long* tbuf = nullptr;
    if (!tbuf) {
        (void)hipMalloc(&tbuf, 32 * 16 * 8);
        (void)hipMemset(tbuf, 0, 32 * 16 * 8);
        (void)hipMemcpyToSymbol(HIP_SYMBOL(g_timing), &tbuf, sizeof(tbuf));
    }
#endif
    hipLaunchKernelGGL((gru_layer_f16x3<KX, FIRST, LAST, MASKED, WINDOW>), dim3(groups < cus ? groups : cus), dim3(256), lds, st, p);
#ifdef KWS_F16_TIMING
    if (getenv("KWS_F16_TIMING")) {
        (void)hipDeviceSynchronize();
        long long h[32 * 16];
        (void)hipMemcpy(h, tbuf, sizeof(h), hipMemcpyDeviceToHost);
        const int base = FIRST ? 0 : 16;
        static const char* nm[16] = {"top+xtail", "G", "R|X", "wr+bar1", "rd+X", "Cm|U", "Cc|X", "tailwork", "bar2", "epilogue", "-", "-", "-", "-", "-", "-"};
        fprintf(stderr, "f16x3 %s T=%d cycles/frame (group 0, waves 0..3):", FIRST ? "first" : "upper", p.T);
        for (int i = 0; i < 10; ++i)
            fprintf(stderr, " %s=%lld/%lld/%lld/%lld", nm[i], h[(base + 0) * 16 + i] / p.T, h[(base + 1) * 16 + i] / p.T, h[(base + 2) * 16 + i] / p.T,
                    h[(base + 3) * 16 + i] / p.T);
        fprintf(stderr, "\n");
    }
#endif
    return hipGetLastError();
}

// the copy-through past seq_len costs 16 VALU instructions per frame: its own instantiation, used only when lengths are given
template <int KX, bool FIRST, bool LAST>
static hipError_t launch_f16x3(const GruF16Params& p, hipStream_t st) {
    if constexpr (LAST) {
        // the window tail exists without the length mask only: the stream manager never passes seq_len (kws_api.hip checks)
        if (p.epi.win.tab != nullptr) return p.seq_len ? hipErrorInvalidValue : launch_f16x3m<KX, FIRST, LAST, false, true>(p, st);
    }
    return p.seq_len ? launch_f16x3m<KX, FIRST, LAST, true>(p, st) : launch_f16x3m<KX, FIRST, LAST, false>(p, st);
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

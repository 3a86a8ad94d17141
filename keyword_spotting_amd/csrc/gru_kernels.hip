// Streaming GRU layer kernels for gfx950 (MI355X).
//
// Replaces, per launch, one layer of the TF while_loop the reference builds at
// models/rnn_ctc.py:228-243 (GRUCell x L under MultiRNNCell + dynamic_rnn) and, in the last layer's
// epilogue, models/rnn_ctc.py:247-284 (inference2: dense), :165 (softmax) and
// utils/prediction.py:65-86 (ctc_decode2's per-frame rule).
//
// Mapping (see DESIGN.md "Kernels"):
//   * one workgroup = 4 waves = 16 streams; the MFMA is v_mfma_f32_16x16x4_f32 with
//     M = 16 output units, N = 16 streams, K = 4 input rows.  Orientation D[unit][stream]:
//     A = weights (lane (g,i): W[k(g)][unit i]), B = activations (lane (g,s): act[s][k(g)]).
//   * wave w owns units [32w, 32w+32) (two 16-unit tiles) of r, u, c and h'.
//   * the K index is permuted so that the C/D register image of a tile IS the B operand of four
//     k-chunks (kws_internal.h "xl" layout): h' feeds the next step with no transpose, only an
//     8 KiB LDS exchange so that every wave sees all 128 units.
//   * resident kernel: the layer's recurrent + candidate weights live in registers (AGPR side of
//     the unified file, 192 + 2*KCX fragments per wave), the gate x-part in LDS; nothing but the
//     mel / previous layer's h stream is read per step.
//   * x-part MFMAs of frame t+1 are issued behind frame t's two barriers (software pipeline), so
//     the LDS exchange latency overlaps independent matrix work.
#include <cstddef>

#include "gru_device.h"
#include "window_device.h"

namespace kws {

// ------------------------------------------------------------------------------------------------
// Resident kernel, H = 128.  KCX = x-part k-chunks (ceil(I/4) for the first layer, 32 above it).
// ------------------------------------------------------------------------------------------------
// WINDOW (instantiated for the upper last layer only): the decode-window step of the stream manager rides at the end of
// every group (window_device.h); every other instantiation compiles exactly as without the parameter.
template <int KCX, bool FIRST, bool LAST, bool WINDOW = false>
__global__ void __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(1, 1)))
gru_layer_resident(const GruLayerParams p) {
    static_assert(!WINDOW || (LAST && !FIRST), "the window tail belongs to the last layer of a stack");
    constexpr int H = 128, NT = 8, KCH = 32;
    const int tid = threadIdx.x;
    const int w = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int lane = tid & 63, g = lane >> 4, s = lane & 15;
    // Persistent over stream groups: the weights are staged ONCE per workgroup and launch; with more groups than the grid
    // (B > 16 x CUs) a workgroup takes groups blockIdx.x, blockIdx.x + gridDim.x, ... one after the other.  Everything
    // that depends on the group is (re)set at the top of the group loop below; the lambdas see it by reference.
    const int n_groups = (p.B + kStreamsPerGroup - 1) / kStreamsPerGroup;
    int group = blockIdx.x;
    int b_raw = group * kStreamsPerGroup + s;
    bool bvalid = b_raw < p.B;
    int b = bvalid ? b_raw : p.B - 1;
    const int T = p.T;
    const int n0 = 2 * w, n1 = 2 * w + 1;

    extern __shared__ __attribute__((aligned(16))) char smem[];
    f32x4* hbuf = reinterpret_cast<f32x4*>(smem);       // [NT][64]  h_{t-1}, xl layout
    f32x4* rhbuf = hbuf + NT * 64;                       // [NT][64]  r (.) h_{t-1}
    f32x4* wlds = rhbuf + NT * 64;                       // [4 waves][KCX][64] gate x-part: {r0,u0,r1,u1}
    EpilogueLds epi = epilogue_carve(reinterpret_cast<char*>(wlds + 4 * KCX * 64));   // LAST only
    if constexpr (WINDOW) epi.cwords = reinterpret_cast<int8_t*>(reinterpret_cast<char*>(wlds + 4 * KCX * 64) + kEpilogueLdsBytes);
    const uint8_t* win_dl = reinterpret_cast<const uint8_t*>(epi.cwords) + 16 * kWinTailWordsStride;     // WINDOW only: the label matcher
    constexpr size_t kWinOffset = offsetof(GruLayerParams, win);
    if constexpr (WINDOW) window_tail_prepare(window_tail_params_from_kernarg(kWinOffset), const_cast<uint8_t*>(win_dl), tid);   // (visible after the group loop's first barrier)
    // FIRST only: one frame of mel for the group, [16 streams x 4 lane groups][kXsStride] floats, row
    // (4s+g) holds x[s][4*kc+g] for kc = 0..KCX-1 -- each lane's B operands are contiguous
    constexpr int kXsStride = xs_stride(KCX);       // 4 * odd: rows 16 apart in one ds_read_b128 group spread over the banks
    float* xs = reinterpret_cast<float*>(reinterpret_cast<char*>(wlds + 4 * KCX * 64) + (LAST ? kEpilogueLdsBytes : 0));

    // ---- stage weights: registers (recurrent + candidate) and LDS (gate x-part) ------------------
    // p.wh is the group-of-4 layout [NT][3][KCH/4][64][4]: one dwordx4 per four fragments.  p.wx is the
    // same layout above the first layer and fragment-major [NT][3][KCX][64] (interleaved k map) in it.
    float wgh[2][2][KCH];   // [tile][r|u][k-chunk]  A fragments of Wg rows I..I+H
    float wch[2][KCH];      // candidate, h-part
    float wcx[2][KCX];      // candidate, x-part
    const f32x4* wh4 = reinterpret_cast<const f32x4*>(p.wh);
#pragma unroll
    for (int j = 0; j < 2; ++j) {
        const int n = 2 * w + j;
#pragma unroll
        for (int k4 = 0; k4 < KCH / 4; ++k4) {
            const f32x4 vr = wh4[((n * 3 + 0) * (KCH / 4) + k4) * 64 + lane];
            const f32x4 vu = wh4[((n * 3 + 1) * (KCH / 4) + k4) * 64 + lane];
            const f32x4 vc = wh4[((n * 3 + 2) * (KCH / 4) + k4) * 64 + lane];
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                wgh[j][0][4 * k4 + e] = vr[e];
                wgh[j][1][4 * k4 + e] = vu[e];
                wch[j][4 * k4 + e] = vc[e];
            }
        }
    }
    if constexpr (FIRST) {
#pragma unroll
        for (int j = 0; j < 2; ++j)
#pragma unroll
            for (int kc = 0; kc < KCX; ++kc) wcx[j][kc] = p.wx[(((2 * w + j) * 3 + 2) * KCX + kc) * 64 + lane];
        for (int kc = 0; kc < KCX; ++kc) {
            f32x4 v;
            v.x = p.wx[((n0 * 3 + 0) * KCX + kc) * 64 + lane];
            v.y = p.wx[((n0 * 3 + 1) * KCX + kc) * 64 + lane];
            v.z = p.wx[((n1 * 3 + 0) * KCX + kc) * 64 + lane];
            v.w = p.wx[((n1 * 3 + 1) * KCX + kc) * 64 + lane];
            wlds[(w * KCX + kc) * 64 + lane] = v;
        }
    } else {
        const f32x4* wx4 = reinterpret_cast<const f32x4*>(p.wx);
#pragma unroll
        for (int j = 0; j < 2; ++j)
#pragma unroll
            for (int k4 = 0; k4 < KCX / 4; ++k4) {
                const f32x4 vc = wx4[(((2 * w + j) * 3 + 2) * (KCX / 4) + k4) * 64 + lane];
#pragma unroll
                for (int e = 0; e < 4; ++e) wcx[j][4 * k4 + e] = vc[e];
            }
        for (int k4 = 0; k4 < KCX / 4; ++k4) {
            const f32x4 r0 = wx4[((n0 * 3 + 0) * (KCX / 4) + k4) * 64 + lane];
            const f32x4 u0 = wx4[((n0 * 3 + 1) * (KCX / 4) + k4) * 64 + lane];
            const f32x4 r1 = wx4[((n1 * 3 + 0) * (KCX / 4) + k4) * 64 + lane];
            const f32x4 u1 = wx4[((n1 * 3 + 1) * (KCX / 4) + k4) * 64 + lane];
#pragma unroll
            for (int e = 0; e < 4; ++e)
                wlds[(w * KCX + 4 * k4 + e) * 64 + lane] = (f32x4){r0[e], u0[e], r1[e], u1[e]};
        }
    }
    // park the recurrent fragments in AGPRs for the whole launch (192 of the 256)
#pragma unroll
    for (int j = 0; j < 2; ++j) {
#pragma unroll
        for (int kc = 0; kc < KCH; ++kc) {
            asm volatile("" : "+a"(wgh[j][0][kc]));
            asm volatile("" : "+a"(wgh[j][1][kc]));
            asm volatile("" : "+a"(wch[j][kc]));
        }
    }
    asm volatile("s_nop 7" ::: "memory");   // v_accvgpr_write -> MFMA SrcA distance
    f32x4 bias_r[2], bias_u[2], bias_c[2];
#pragma unroll
    for (int j = 0; j < 2; ++j) {
        const int n = 2 * w + j;
        bias_r[j] = ld4(p.bias + 0 * H + n * 16 + 4 * g);
        bias_u[j] = ld4(p.bias + 1 * H + n * 16 + 4 * g);
        bias_c[j] = ld4(p.bias + 2 * H + n * 16 + 4 * g);
    }
    float wfc[2][4];
    f32x4 bfc4 = splat4(0.f);
    if (LAST) {
#pragma unroll
        for (int j = 0; j < 2; ++j)
#pragma unroll
            for (int e = 0; e < 4; ++e) wfc[j][e] = p.wfc[((2 * w + j) * 4 + e) * 64 + lane];
        if (w == 0) bfc4 = ld4(p.bfc + 4 * g);
    }

    // ---- per-group state: set by enter_group() -----------------------------------------------------
    int len_s = T;
    f32x4 hreg[2];
    const float4* xl_src = nullptr;
    const float4* xprev = nullptr;
    // the x-stream descriptors below (xl_row, xl_q, xl_active) do not depend on the group
    constexpr int XQ = KCX;                          // float4 pieces per mel row (I == 4*KCX)
    const int xl_row = lane / XQ, xl_q = lane % XQ;  // this lane's (stream-in-quarter, piece)
    const bool xl_active = FIRST && lane < 4 * XQ;
    auto enter_group = [&]() {
        b_raw = group * kStreamsPerGroup + s;
        bvalid = b_raw < p.B;
        b = bvalid ? b_raw : p.B - 1;
        const bool do_reset = p.reset != nullptr && p.reset[b] != 0;
        len_s = p.seq_len ? p.seq_len[b] - p.t_base : T;
#pragma unroll
        for (int j = 0; j < 2; ++j) {
            const int n = 2 * w + j;
            hreg[j] = do_reset ? splat4(0.f) : ld4(p.state_in + (size_t)b * H + n * 16 + 4 * g);
            hbuf[n * 64 + lane] = hreg[j];
        }
        if (LAST && tid < 16) {
            const int bb = group * kStreamsPerGroup + tid;
            int pw = -1;
            if (bb < p.B && p.prev_word && !(p.reset && p.reset[bb])) pw = p.prev_word[bb];
            epi.carry[tid] = pw;          // block 0 reads carry[0][.]
        }
        if constexpr (FIRST) {
            const int xl_b = min(group * kStreamsPerGroup + 4 * w + (xl_active ? xl_row : 0), p.B - 1);
            xl_src = reinterpret_cast<const float4*>(p.x_mel + (size_t)xl_b * (p.t_stride ? p.t_stride : T) * p.I) + xl_q;
        } else {
            xprev = p.x_prev + (size_t)group * T * NT * 64 + lane;
        }
    };

    // ---- x stream --------------------------------------------------------------------------------
    // First layer: the four waves fetch the group's mel frame COOPERATIVELY -- wave w loads streams
    // 4w..4w+3 (one global_load_dwordx4, 4*I/4 active lanes) two frames ahead, scatters it into `xs`
    // late in the frame, and every wave reads its B operands back with three LDS reads.  A global load
    // costs ~30 cycles of MFMA time and an LDS read ~2.4 (tools/ubench/mfma_operands.hip); ten divergent
    // dword loads per wave per frame were 4.5 % of this kernel.
    // Upper layers: the previous layer's xl-layout block, one slice per MFMA group (a burst of loads from
    // four phase-locked waves backs up the address path and stalls the MFMAs queued behind it).
    float4 xl_inflight = make_float4(0.f, 0.f, 0.f, 0.f);
    auto coop_issue = [&](int t_req) {               // global -> register (in flight)
        const int t = t_req < T ? t_req : T - 1;
        if (xl_active) xl_inflight = xl_src[(size_t)t * XQ];
    };
    auto coop_commit = [&]() {                       // register -> xs
        if (xl_active) {
            float* dst = xs + (4 * (4 * w + xl_row)) * kXsStride + xl_q;
            dst[0 * kXsStride] = xl_inflight.x;
            dst[1 * kXsStride] = xl_inflight.y;
            dst[2 * kXsStride] = xl_inflight.z;
            dst[3 * kXsStride] = xl_inflight.w;
        }
    };
    float xbuf0[KCX];
    auto read_xs = [&](float (&dst)[KCX]) {          // xs -> this lane's B operands
        const float* row = xs + (4 * s + g) * kXsStride;
#pragma unroll
        for (int k4 = 0; k4 < KCX / 4; ++k4) {
            const f32x4 v = *reinterpret_cast<const f32x4*>(row + 4 * k4);
            dst[4 * k4 + 0] = v[0]; dst[4 * k4 + 1] = v[1]; dst[4 * k4 + 2] = v[2]; dst[4 * k4 + 3] = v[3];
        }
        if constexpr (KCX % 4 >= 2) {
            const f32x2 v = *reinterpret_cast<const f32x2*>(row + (KCX / 4) * 4);
            dst[(KCX / 4) * 4 + 0] = v[0]; dst[(KCX / 4) * 4 + 1] = v[1];
        }
        if constexpr (KCX % 2 == 1) dst[KCX - 1] = row[KCX - 1];
    };
    auto load_x_slice = [&](float (&dst)[KCX], int t_req, const int sl) {   // upper layers; sl: unrolled constant
        if constexpr (!FIRST) {
            const int t = t_req < T ? t_req : T - 1;
            const float4 v = xprev[((size_t)t * NT + sl) * 64];
            dst[4 * sl + 0] = v.x; dst[4 * sl + 1] = v.y; dst[4 * sl + 2] = v.z; dst[4 * sl + 3] = v.w;
        }
    };

    f32x4 acc_r[2], acc_u[2], acc_c[2];
    // gate x-part for k-chunks [K0, K1): A fragments stream from LDS through a 3-deep register ring
    // (two ds_read_b128 in flight behind the MFMAs that consume the third)
    auto gates_x_part = [&](const float (&xB)[KCX], auto k0_, auto k1_, auto pin_) {
        constexpr int K0 = decltype(k0_)::value, K1 = decltype(k1_)::value;
        constexpr bool PIN = decltype(pin_)::value;
        f32x4 ring[3];
        if (K0 < K1) ring[K0 % 3] = wlds[(w * KCX + K0) * 64 + lane];
        if (K0 + 1 < K1) ring[(K0 + 1) % 3] = wlds[(w * KCX + K0 + 1) * 64 + lane];
#pragma unroll
        for (int kc = K0; kc < K1; ++kc) {
            if (kc + 2 < K1) ring[(kc + 2) % 3] = wlds[(w * KCX + kc + 2) * 64 + lane];
            if (PIN) __builtin_amdgcn_sched_barrier(0);   // keep the read two groups ahead of its MFMAs
            const f32x4 a4 = ring[kc % 3];
            acc_r[0] = mfma4(a4.x, xB[kc], acc_r[0]);
            acc_u[0] = mfma4(a4.y, xB[kc], acc_u[0]);
            acc_r[1] = mfma4(a4.z, xB[kc], acc_r[1]);
            acc_u[1] = mfma4(a4.w, xB[kc], acc_u[1]);
            if (PIN) __builtin_amdgcn_sched_barrier(0);
        }
    };
    using pinned = std::true_type;
    constexpr int KSPLIT = KCX / 2;
    using k_lo = std::integral_constant<int, 0>;
    using k_mid = std::integral_constant<int, KSPLIT>;
    using k_hi = std::integral_constant<int, KCX>;
    auto cand_x = [&](const float (&xB)[KCX]) {
#pragma unroll
        for (int j = 0; j < 2; ++j) acc_c[j] = bias_c[j];
#pragma unroll
        for (int kc = 0; kc < KCX; ++kc) {
            acc_c[0] = mfma4(wcx[0][kc], xB[kc], acc_c[0]);
            acc_c[1] = mfma4(wcx[1][kc], xB[kc], acc_c[1]);
        }
    };

    f32x4 hb_a, hb_b;            // exchange-read pipeline registers (two float4 in flight)

    // One frame (xcur == xnxt == the single B-operand buffer: x(t+1) lands in it during this frame).
    auto frame = [&](int t, float (&xcur)[KCX], float (&xnxt)[KCX]) {
        // gates, h-part:  acc_{r,u} += Wg[I:,:]^T h_{t-1}   (hb_a/hb_b were fetched behind cand_x);
        // one slice of x(t+1) is requested per group
        mfma_prefence(acc_r[0], acc_u[0], acc_r[1], acc_u[1]);
#pragma unroll
        for (int nn = 0; nn < NT; ++nn) {
            const f32x4 hb = (nn & 1) ? hb_b : hb_a;
            if (nn + 2 < NT) {
                if (nn & 1) hb_b = hbuf[(nn + 2) * 64 + lane]; else hb_a = hbuf[(nn + 2) * 64 + lane];
            }
            if constexpr (FIRST) { if (nn == 0) coop_issue(t + 2); } else load_x_slice(xnxt, t + 1, nn);
            __builtin_amdgcn_sched_barrier(0);
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                const int kc = 4 * nn + e;
                const float hv = hb[e];
                KWS_MFMA_A(acc_r[0], wgh[0][0][kc], hv);
                KWS_MFMA_A(acc_u[0], wgh[0][1][kc], hv);
                KWS_MFMA_A(acc_r[1], wgh[1][0][kc], hv);
                KWS_MFMA_A(acc_u[1], wgh[1][1][kc], hv);
            }
        }
        mfma_fence(acc_r[0], acc_u[0], acc_r[1], acc_u[1]);
        if constexpr (FIRST) read_xs(xcur);          // x(t+1), committed to LDS during frame t-1
        // ---- region A: the 16 sigmoids as one VALU cluster (r first so r(.)h reaches LDS early), then the
        // first half of frame t+1's gate x-part as cover for the exchange
        f32x4 u[2];
#pragma unroll
        for (int j = 0; j < 2; ++j) {
            const f32x2 r_lo = sigmoid2((f32x2){acc_r[j][0], acc_r[j][1]});
            const f32x2 r_hi = sigmoid2((f32x2){acc_r[j][2], acc_r[j][3]});
            const f32x2 rh_lo = r_lo * (f32x2){hreg[j][0], hreg[j][1]};
            const f32x2 rh_hi = r_hi * (f32x2){hreg[j][2], hreg[j][3]};
            rhbuf[(2 * w + j) * 64 + lane] = (f32x4){rh_lo.x, rh_lo.y, rh_hi.x, rh_hi.y};
        }
#pragma unroll
        for (int j = 0; j < 2; ++j) {
            const f32x2 u_lo = sigmoid2((f32x2){acc_u[j][0], acc_u[j][1]});
            const f32x2 u_hi = sigmoid2((f32x2){acc_u[j][2], acc_u[j][3]});
            u[j] = (f32x4){u_lo.x, u_lo.y, u_hi.x, u_hi.y};
        }
#pragma unroll
        for (int j = 0; j < 2; ++j) { acc_r[j] = bias_r[j]; acc_u[j] = bias_u[j]; }
        __builtin_amdgcn_sched_barrier(0);
        gates_x_part(xcur, k_lo{}, k_mid{}, pinned{});
        lds_barrier();           // #1: r(.)h visible; every wave is done reading hbuf
        hb_a = rhbuf[0 * 64 + lane];
        hb_b = rhbuf[1 * 64 + lane];
        gates_x_part(xcur, k_mid{}, k_hi{}, pinned{});     // second half hides the rhbuf read latency

        // candidate, h-part:  acc_c += Wc[I:,:]^T (r (.) h_{t-1})
        mfma_prefence(acc_c[0], acc_c[1]);
#pragma unroll
        for (int nn = 0; nn < NT; ++nn) {
            const f32x4 rb = (nn & 1) ? hb_b : hb_a;
            if (nn + 2 < NT) {
                if (nn & 1) hb_b = rhbuf[(nn + 2) * 64 + lane]; else hb_a = rhbuf[(nn + 2) * 64 + lane];
            }
            __builtin_amdgcn_sched_barrier(0);
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                const int kc = 4 * nn + e;
                const float rv = rb[e];
                KWS_MFMA_A(acc_c[0], wch[0][kc], rv);
                KWS_MFMA_A(acc_c[1], wch[1][kc], rv);
            }
        }
        mfma_fence(acc_c[0], acc_c[1]);
        // ---- region B: tanh + state update as one VALU cluster
        const unsigned live = t < len_s ? 0xffffffffu : 0u;   // dynamic_rnn copy-through past seq_len
        f32x4 hout[2];
#pragma unroll
        for (int j = 0; j < 2; ++j) {
#pragma unroll
            for (int h2 = 0; h2 < 2; ++h2) {
                const f32x2 c = tanh2((f32x2){acc_c[j][2 * h2], acc_c[j][2 * h2 + 1]});
                const f32x2 uu = {u[j][2 * h2], u[j][2 * h2 + 1]};
                const f32x2 hh = {hreg[j][2 * h2], hreg[j][2 * h2 + 1]};
                const f32x2 hn = (1.0f - uu) * c + uu * hh;       // u*h + (1-u)*c
                hreg[j][2 * h2] = bitsel(live, hn.x, hh.x);
                hreg[j][2 * h2 + 1] = bitsel(live, hn.y, hh.y);
                if (LAST) {
                    hout[j][2 * h2] = bitsel(live, hn.x, 0.f);
                    hout[j][2 * h2 + 1] = bitsel(live, hn.y, 0.f);
                }
            }
        }
#pragma unroll
        for (int j = 0; j < 2; ++j) acc_c[j] = bias_c[j];
        constexpr int NCX = 2 * KCX;                         // candidate x-part MFMAs of frame t+1
        constexpr int NPOST = NCX >= 32 ? 16 : 8;            // kept for after barrier #2 (covers the hbuf read)
        constexpr int NPRE = NCX - NPOST;
        auto cand_x_mfma = [&](auto mc) {
            constexpr int m = decltype(mc)::value, kc = m / 2;
            if constexpr (m % 2 == 0) acc_c[0] = mfma4(wcx[0][kc], xcur[kc], acc_c[0]);
            else acc_c[1] = mfma4(wcx[1][kc], xcur[kc], acc_c[1]);
        };
        if constexpr (FIRST) coop_commit();          // x(t+2): visible after barrier #2, read in frame t+1
#pragma unroll
        for (int j = 0; j < 2; ++j) {
            hbuf[(2 * w + j) * 64 + lane] = hreg[j];
            if (!LAST) {
                const f32x4 o = hreg[j];
                p.h_out[((size_t)group * T + t) * NT * 64 + (2 * w + j) * 64 + lane] =
                    make_float4(o[0], o[1], o[2], o[3]);
            }
        }
        if (LAST) {
            // partial logits over this wave's 32 units: Wfc^T[:, units] h'[units]
            f32x4 accf = bfc4;
#pragma unroll
            for (int j = 0; j < 2; ++j)
#pragma unroll
                for (int e = 0; e < 4; ++e) accf = mfma4(wfc[j][e], hout[j][e], accf);
            if (g < 2) *reinterpret_cast<f32x4*>(epi.pstage + (w * 16 + s) * 8 + 4 * g) = accf;
        }
        __builtin_amdgcn_sched_barrier(0);
        static_for<0, NPRE>(cand_x_mfma);     // most of frame t+1's candidate x-part covers the LDS write
        __builtin_amdgcn_sched_barrier(0);
        lds_barrier();           // #2: h_t visible; every wave is done reading rhbuf
        hb_a = hbuf[0 * 64 + lane];
        hb_b = hbuf[1 * 64 + lane];
        FoldRegs fold;
        const bool folder = LAST && w == (t & 3);
        if (folder) epilogue_fold_load(epi, lane, fold);      // LDS reads in flight behind the MFMAs below
        __builtin_amdgcn_sched_barrier(0);
        static_for<NPRE, NCX>(cand_x_mfma);   // the rest of frame t+1's candidate x-part hides the hbuf read
        __builtin_amdgcn_sched_barrier(0);
        if (LAST) {
            if (folder) epilogue_fold_store(epi, t, lane, fold);
            if (((t + 1) & (kRingFrames - 1)) == 0 || t == T - 1) {
                const int t0 = t & ~(kRingFrames - 1);
                lds_barrier();                       // the fold of frame t is visible to every wave
                epilogue_flush(p, epi, group, t0, t - t0 + 1, w, lane, t == T - 1);
            }
        }
    };

    for (; group < n_groups; group += gridDim.x) {
    enter_group();
    __syncthreads();             // staged weights (first group) / this group's state and carry are in LDS
    if (T > 0) {
        if constexpr (FIRST) {
            coop_issue(0);
            coop_commit();
            __syncthreads();
            read_xs(xbuf0);                          // x(0)
            coop_issue(1);
        } else {
#pragma unroll
            for (int sl = 0; sl < NT; ++sl) load_x_slice(xbuf0, 0, sl);
        }
#pragma unroll
        for (int j = 0; j < 2; ++j) { acc_r[j] = bias_r[j]; acc_u[j] = bias_u[j]; }
        gates_x_part(xbuf0, k_lo{}, k_hi{}, pinned{});
        hb_a = hbuf[0 * 64 + lane];
        hb_b = hbuf[1 * 64 + lane];
        cand_x(xbuf0);
        if constexpr (FIRST) {
            __syncthreads();                         // every wave has read x(0) out of xs
            coop_commit();                           // x(1)
            __syncthreads();
        }
    }
    for (int t = 0; t < T; ++t) frame(t, xbuf0, xbuf0);

    if (bvalid) {
#pragma unroll
        for (int j = 0; j < 2; ++j)
            *reinterpret_cast<f32x4*>(p.state_out + (size_t)b * H + (2 * w + j) * 16 + 4 * g) = hreg[j];
    }
    __syncthreads();             // every wave is done with this group's LDS state before the next group overwrites it
    if constexpr (WINDOW) {
        // detector.py:195-209 for this group's 16 streams: the call's frame words wait in epi.cwords, the scratch is hbuf | rhbuf
        const WindowTail win = window_tail_params_from_kernarg(kWinOffset);
        WindowTailRegs<2> wreq;
        window_tail_request<2>(win, p.B, group * kStreamsPerGroup, tid, wreq);
        window_tail<2>(win, p.B, group * kStreamsPerGroup, T, epi.cwords, kWinTailWordsStride, win_dl, reinterpret_cast<char*>(hbuf), tid, wreq);
        __syncthreads();
    }
    }
}

// ------------------------------------------------------------------------------------------------
// Generic kernel: H = 64*TPW, weights streamed from L2 every frame (group-of-4 fragment layout).
// Same orientation, exchange layout and epilogue; no software pipeline.
// ------------------------------------------------------------------------------------------------
template <int TPW, bool FIRST, bool LAST, bool PIPE>
__device__ __forceinline__ void gru_layer_generic_body(const GruLayerParams& p, const int group) {
    constexpr int NT = 4 * TPW, H = 64 * TPW;
    const int tid = threadIdx.x;
    const int w = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int lane = tid & 63, g = lane >> 4, s = lane & 15;
    const int b_raw = group * kStreamsPerGroup + s;
    const bool bvalid = b_raw < p.B;
    const int b = bvalid ? b_raw : p.B - 1;
    const int T = p.T, I = p.I;
    const int KCX4 = p.KCX / 4;

    extern __shared__ __attribute__((aligned(16))) char smem[];
    f32x4* hbuf = reinterpret_cast<f32x4*>(smem);
    f32x4* rhbuf = hbuf + NT * 64;
    f32x4* xstage = rhbuf + NT * 64;                                                     // pipelined launch only
    f32x4* biasl = xstage + NT * 64;                                                      // [3][NT][4 g] bias fragments
    const EpilogueLds epi = epilogue_carve(reinterpret_cast<char*>(biasl + 3 * NT * 4));  // LAST only

    // The weight stream (and the seam of a layer-by-layer launch) goes through BUFFER loads: lane offset in one VGPR that
    // never changes, everything else in the scalar offset.  With flat global loads every fragment fetched per frame cost
    // two or three VALU instructions of 64-bit address arithmetic -- about one per MFMA, beside an f32 MFMA stream that
    // shares the FP32 pipe with them (profiles/r2_configC_pmc.json: 2.0 VALU instructions per MFMA).
    const __amdgpu_buffer_rsrc_t wx_rsrc = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(p.wx), (short)0, 0x7fffffff, 0x00020000);   // [NT][3][KCX4][64] float4
    const __amdgpu_buffer_rsrc_t wh_rsrc = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(p.wh), (short)0, 0x7fffffff, 0x00020000);   // [NT][3][NT][64] float4
    const int lane16 = lane * 16;
    auto bload = [&](const __amdgpu_buffer_rsrc_t& r, int frag) -> f32x4 {      // frag: wave-uniform fragment index (1 KiB each)
        return __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(r, lane16, frag * 1024, 0));
    };

    // biases live in LDS (the accumulators are re-initialised from there every frame): at TPW = 4 the 48 registers
    // they would pin are the difference between fitting the 512-register file and spilling
    f32x4 hreg[TPW];
    const bool do_reset = p.reset != nullptr && p.reset[b] != 0;
    const int len_s = p.seq_len ? p.seq_len[b] - p.t_base : T;
    for (int i = tid; i < 3 * NT * 4; i += 256) biasl[i] = ld4(p.bias + 4 * i);   // [gate][tile][g] = bias[gate*H + 16*tile + 4g ..]
#pragma unroll
    for (int j = 0; j < TPW; ++j) {
        const int n = TPW * w + j;
        hreg[j] = do_reset ? splat4(0.f) : ld4(p.state_in + (size_t)b * H + n * 16 + 4 * g);
        hbuf[n * 64 + lane] = hreg[j];
    }
    f32x4 bfc4 = splat4(0.f);
    if (LAST) {
        if (w == 0) bfc4 = ld4(p.bfc + 4 * g);
        if (tid < 16) {
            const int bb = group * kStreamsPerGroup + tid;
            int pw = -1;
            if (bb < p.B && p.prev_word && !(p.reset && p.reset[bb])) pw = p.prev_word[bb];
            epi.carry[tid] = pw;
        }
    }
    const float* xrow = FIRST ? p.x_mel + (size_t)b * (p.t_stride ? p.t_stride : T) * I : nullptr;
    const __amdgpu_buffer_rsrc_t xp_rsrc = __builtin_amdgcn_make_buffer_rsrc(
        FIRST ? const_cast<float*>(p.wh) : const_cast<float*>(reinterpret_cast<const float*>(p.x_prev + (size_t)group * T * NT * 64)),
        (short)0, 0x7fffffff, 0x00020000);       // this group's [T][NT][64] float4 block of the seam
    const __amdgpu_buffer_rsrc_t ho_rsrc = __builtin_amdgcn_make_buffer_rsrc(
        LAST ? const_cast<float*>(p.wh) : reinterpret_cast<float*>(p.h_out + (size_t)group * T * NT * 64), (short)0, 0x7fffffff, 0x00020000);
    constexpr int kSysScope = 1 | 16;            // cache-policy bits sc0 | sc1: system scope, past the non-coherent cache levels
    const bool vec_ok = (I & 3) == 0;
    __syncthreads();

    // Weight fragments stream from L2 every frame.  One "row" = the fragments of all TPW tiles for one k-group;
    // rows ping-pong between two register sets, the next row's loads pinned ahead of the current row's MFMAs
    // (12 TPW MFMAs = 1.5k cycles at TPW = 4, more than an L2 round trip).  Left to hipcc, every tile's three loads
    // were waited for right before their MFMAs and the kernel ran latency-bound at half the MFMA rate.
    auto x_operand = [&](int t, int k4) -> f32x4 {
        f32x4 xb;
        if (FIRST) {
            const int k = 16 * k4 + 4 * g;
            const float* src = xrow + (size_t)t * I + k;
            if (vec_ok && k + 3 < I) {
                xb = ld4(src);
            } else {
#pragma unroll
                for (int e = 0; e < 4; ++e) xb[e] = (k + e < I) ? src[e] : 0.f;
            }
        } else if (PIPE) {
            xb = xstage[k4 * 64 + lane];          // staged at the top of the frame (below)
        } else {
            xb = bload(xp_rsrc, t * NT + k4);
        }
        return xb;
    };
    // a row covers RT tiles of one k-group: all of the wave's tiles (half rows, RT = 2 at TPW = 4, were measured
    // 10-15 % slower).  The streaming loops must stay rolled (#pragma nounroll): unrolled, hipcc materialises a
    // 64-bit address pair per load and the TPW = 4 kernels spill.
    constexpr int RT = TPW, NP = TPW / RT;
    struct RowX { f32x4 a[RT][3]; f32x4 xb; };
    struct RowH { f32x4 a[RT][2]; };
    struct RowC { f32x4 a[RT]; };
    auto load_x = [&](RowX& r, int t, int q) {
        const int qq = q < KCX4 * NP ? q : KCX4 * NP - 1, kk = qq / NP, part = qq - kk * NP;
#pragma unroll
        for (int j = 0; j < RT; ++j)
#pragma unroll
            for (int c = 0; c < 3; ++c) r.a[j][c] = bload(wx_rsrc, ((TPW * w + part * RT + j) * 3 + c) * KCX4 + kk);
        r.xb = x_operand(t, kk);
        __builtin_amdgcn_sched_barrier(0);
    };
    auto load_h = [&](RowH& r, int q) {
        const int qq = q < NT * NP ? q : NT * NP - 1, kk = qq / NP, part = qq - kk * NP;
#pragma unroll
        for (int j = 0; j < RT; ++j)
#pragma unroll
            for (int c = 0; c < 2; ++c) r.a[j][c] = bload(wh_rsrc, ((TPW * w + part * RT + j) * 3 + c) * NT + kk);
        __builtin_amdgcn_sched_barrier(0);
    };
    auto load_c = [&](RowC& r, int q) {
        const int qq = q < NT * NP ? q : NT * NP - 1, kk = qq / NP, part = qq - kk * NP;
#pragma unroll
        for (int j = 0; j < RT; ++j) r.a[j] = bload(wh_rsrc, ((TPW * w + part * RT + j) * 3 + 2) * NT + kk);
        __builtin_amdgcn_sched_barrier(0);
    };
    for (int t = 0; t < T; ++t) {
        f32x4 acc_r[TPW], acc_u[TPW], acc_c[TPW];
#pragma unroll
        for (int j = 0; j < TPW; ++j) {
            const int n = TPW * w + j;
            acc_r[j] = biasl[(0 * NT + n) * 4 + g]; acc_u[j] = biasl[(1 * NT + n) * 4 + g]; acc_c[j] = biasl[(2 * NT + n) * 4 + g];
        }
        if (PIPE && !FIRST) {
            // layer-pipelined launch: frame t of the layer below must have landed (its workgroup runs concurrently
            // on another CU).  Every wave polls for itself; the bound turns a protocol bug into a wrong answer plus
            // an error flag instead of a hung GPU.
            // Seams and counters live in FINE-GRAINED device memory (uncached in L2, coherent across XCDs), so no
            // cache-wide acquire is needed -- an agent-scope acquire invalidates the L2 and with it the weight
            // stream of every workgroup on the XCD, once per frame (measured: slower than the sequential launches).
            int spins = 0;
            while (__hip_atomic_load(p.ready_in + group, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) <= t) {
                __builtin_amdgcn_s_sleep(2);
                if (++spins > (1 << 24)) { if (lane == 0) *reinterpret_cast<volatile int*>(p.pipe_error) = 1; break; }
            }
            asm volatile("" ::: "memory");
            // The frame's input block was written by a workgroup on another CU / XCD while this kernel runs:
            // system-scope (sc0 sc1) loads go past the non-coherent cache levels, dword by dword, at memory
            // latency -- so the whole block is fetched at once (all loads in flight) and parked in LDS.
            // (16-byte sc0 sc1 buffer loads: dword-granular system-scope accesses cost ~6x the fabric time per byte;
            // tearing is no concern, the rows are ordered by the frame counter)
            f32x4 xv[NT / 4];
#pragma unroll
            for (int i = 0; i < NT / 4; ++i)
                xv[i] = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(xp_rsrc, lane16, (t * NT + (w + 4 * i)) * 1024, kSysScope));
#pragma unroll
            for (int i = 0; i < NT / 4; ++i) xstage[(w + 4 * i) * 64 + lane] = xv[i];
            __syncthreads();
        }
        // x-part (gates and candidate): rows q = k4 * NP + part, ping-pong, unrolled by two (NP == 2 keeps the parity
        // of q equal to the part, NP == 1 has a single part)
        {
#define KWS_MMA_X(P_)                                                                                 \
            _Pragma("unroll") for (int j = 0; j < RT; ++j)                                             \
                _Pragma("unroll") for (int e = 0; e < 4; ++e) {                                        \
                    acc_r[P_ * RT + j] = mfma4(rr.a[j][0][e], rr.xb[e], acc_r[P_ * RT + j]);           \
                    acc_u[P_ * RT + j] = mfma4(rr.a[j][1][e], rr.xb[e], acc_u[P_ * RT + j]);           \
                    acc_c[P_ * RT + j] = mfma4(rr.a[j][2][e], rr.xb[e], acc_c[P_ * RT + j]);           \
                }
            RowX ra, rb;
            const int NQ = KCX4 * NP;
            load_x(ra, t, 0);
#pragma nounroll
            for (int q = 0; q < NQ; q += 2) {
                load_x(rb, t, q + 1);
                { const RowX& rr = ra; KWS_MMA_X(0); __builtin_amdgcn_sched_barrier(0); }
                if (q + 1 < NQ) {
                    load_x(ra, t, q + 2);
                    { const RowX& rr = rb; if (NP == 2) { KWS_MMA_X((NP - 1)); } else { KWS_MMA_X(0); } __builtin_amdgcn_sched_barrier(0); }
                }
            }
#undef KWS_MMA_X
        }
        // gates, h-part
        {
#define KWS_MMA_H(P_)                                                                                 \
            _Pragma("unroll") for (int j = 0; j < RT; ++j)                                             \
                _Pragma("unroll") for (int e = 0; e < 4; ++e) {                                        \
                    acc_r[P_ * RT + j] = mfma4(rr.a[j][0][e], hb[e], acc_r[P_ * RT + j]);              \
                    acc_u[P_ * RT + j] = mfma4(rr.a[j][1][e], hb[e], acc_u[P_ * RT + j]);              \
                }
            RowH ra, rb;
            constexpr int NQ = NT * NP;                   // even
            load_h(ra, 0);
#pragma nounroll
            for (int q = 0; q < NQ; q += 2) {
                load_h(rb, q + 1);
                { const RowH& rr = ra; const f32x4 hb = hbuf[(q / NP) * 64 + lane]; KWS_MMA_H(0); __builtin_amdgcn_sched_barrier(0); }
                load_h(ra, q + 2);
                { const RowH& rr = rb; const f32x4 hb = hbuf[((q + 1) / NP) * 64 + lane];
                  if (NP == 2) { KWS_MMA_H((NP - 1)); } else { KWS_MMA_H(0); } __builtin_amdgcn_sched_barrier(0); }
            }
#undef KWS_MMA_H
        }
        f32x4 u[TPW];
#pragma unroll
        for (int j = 0; j < TPW; ++j) {
            f32x4 rh;
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                rh[e] = sigmoid_f(acc_r[j][e]) * hreg[j][e];
                u[j][e] = sigmoid_f(acc_u[j][e]);
            }
            rhbuf[(TPW * w + j) * 64 + lane] = rh;
        }
        RowC ca, cb;
        if (PIPE && !LAST) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");   // last frame's rows (half a frame old): written through
        load_c(ca, 0);                 // does not depend on the exchange: issued ahead of the barrier
        __syncthreads();
        if (PIPE && !LAST && t > 0 && tid == 0)
            __hip_atomic_store(p.ready_out + group, t, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        {
#define KWS_MMA_C(P_)                                                                                 \
            _Pragma("unroll") for (int j = 0; j < RT; ++j)                                             \
                _Pragma("unroll") for (int e = 0; e < 4; ++e)                                          \
                    acc_c[P_ * RT + j] = mfma4(rr.a[j][e], rb[e], acc_c[P_ * RT + j]);
            constexpr int NQ = NT * NP;
#pragma nounroll
            for (int q = 0; q < NQ; q += 2) {
                load_c(cb, q + 1);
                { const RowC& rr = ca; const f32x4 rb = rhbuf[(q / NP) * 64 + lane]; KWS_MMA_C(0); __builtin_amdgcn_sched_barrier(0); }
                load_c(ca, q + 2);
                { const RowC& rr = cb; const f32x4 rb = rhbuf[((q + 1) / NP) * 64 + lane];
                  if (NP == 2) { KWS_MMA_C((NP - 1)); } else { KWS_MMA_C(0); } __builtin_amdgcn_sched_barrier(0); }
            }
#undef KWS_MMA_C
        }
        const unsigned live = t < len_s ? 0xffffffffu : 0u;
        f32x4 accf = bfc4;
#pragma unroll
        for (int j = 0; j < TPW; ++j) {
            const int n = TPW * w + j;
            f32x4 hout;
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                const float c = tanh_f(acc_c[j][e]);
                // explicit fma: which of the two products -ffp-contract fuses must not depend on the instantiation
                // (sequential and pipelined launches have to agree bit for bit)
                const float hn = fmaf(u[j][e], hreg[j][e], (1.0f - u[j][e]) * c);
                hreg[j][e] = bitsel(live, hn, hreg[j][e]);
                hout[e] = bitsel(live, hn, 0.f);
            }
            hbuf[n * 64 + lane] = hreg[j];
            if (!LAST) {
                const f32x4 o = hreg[j];
                float4* dst = p.h_out + ((size_t)group * T + t) * NT * 64 + n * 64 + lane;
                if (PIPE) {
                    typedef int i32x4 __attribute__((ext_vector_type(4)));
                    __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(i32x4, o), ho_rsrc, lane16, (t * NT + n) * 1024, kSysScope);
                } else {
                    *dst = make_float4(o[0], o[1], o[2], o[3]);
                }
            } else {
#pragma unroll
                for (int e = 0; e < 4; ++e) accf = mfma4(p.wfc[(n * 4 + e) * 64 + lane], hout[e], accf);
            }
        }
        if (LAST && g < 2) *reinterpret_cast<f32x4*>(epi.pstage + (w * 16 + s) * 8 + 4 * g) = accf;
        __syncthreads();
        // Pipelined producer: the counter of frame t moves at the NEXT frame's mid barrier (below), after every wave has
        // drained its stores there -- __syncthreads() itself does not wait for global stores (workgroup scope), and a
        // consumer on another CU must not see the counter before the rows (found by tools/stress_determinism.py: only
        // the first call after create showed it, later calls re-read the identical rows of the previous call).
        if (LAST) {
            if (w == (t & 3)) epilogue_fold(epi, t, lane);
            if (((t + 1) & (kRingFrames - 1)) == 0 || t == T - 1) {
                const int t0 = t & ~(kRingFrames - 1);
                __syncthreads();
                epilogue_flush(p, epi, group, t0, t - t0 + 1, w, lane, t == T - 1);
            }
        }
    }
    if (PIPE && !LAST) {                         // the last frame
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __syncthreads();
        if (tid == 0) __hip_atomic_store(p.ready_out + group, T, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    }
    if (bvalid) {
#pragma unroll
        for (int j = 0; j < TPW; ++j)
            *reinterpret_cast<f32x4*>(p.state_out + (size_t)b * H + (TPW * w + j) * 16 + 4 * g) = hreg[j];
    }
}

template <int TPW, bool FIRST, bool LAST>
__global__ void __launch_bounds__(256) gru_layer_generic(const GruLayerParams p) {
    gru_layer_generic_body<TPW, FIRST, LAST, false>(p, blockIdx.x);
}

// Layer-pipelined launch: ONE grid of L x G workgroups, layer-major, so that every layer of every 16-stream group
// runs concurrently on its own CU and layer l consumes frame t of layer l-1 as soon as it is published (a per-group
// frame counter in global memory: release after the h_out stores, acquire before the x loads).  Wall time becomes
// (T + L - 1) frame times instead of L x T.  It pays when L x G workgroups fit the chip at once -- BASELINE
// configs[4] (L = 4, B = 1024: 256 workgroups on 256 CUs) -- and cannot deadlock in any case: a workgroup only
// waits for one with a smaller index, and workgroups are dispatched in index order.
//
// XCD affinity: workgroups are dealt round-robin to the 8 XCDs (block i -> XCD i % 8), each with its own 4 MB L2.
// Layer-major order would put every layer's weight stream (1.5 MB per layer at H = 256, 6 MB in all) through every
// L2; with L dividing 8 the mapping below gives XCD x the layer x % L only (measured at configs[4]: 9.32 vs 9.53 ms),
// and a workgroup still waits only for block i - 1.
template <int TPW>
__global__ void __launch_bounds__(256) gru_stack_generic_pipelined(const GruStackParams sp) {
    int layer, group;
    if (sp.xcd_affine) {
        const int xcd = blockIdx.x & 7, slot = blockIdx.x >> 3, per = 8 / sp.L;
        layer = xcd % sp.L;
        group = slot * per + xcd / sp.L;
        if (group >= sp.G) return;              // grid padded to a multiple of 8; nobody waits for a padding block
    } else {
        layer = blockIdx.x / sp.G;
        group = blockIdx.x - layer * sp.G;
    }
    if (layer == 0) gru_layer_generic_body<TPW, true, false, true>(sp.layer[0], group);
    else if (layer == sp.L - 1) gru_layer_generic_body<TPW, false, true, true>(sp.layer[layer], group);
    else gru_layer_generic_body<TPW, false, false, true>(sp.layer[layer], group);
}

// ------------------------------------------------------------------------------------------------
// launchers
// ------------------------------------------------------------------------------------------------
static size_t resident_lds_bytes(int kcx, bool first, bool last) {
    size_t n = 2 * 8 * 64 * 16 + (size_t)4 * kcx * 64 * 16;
    if (last) n += kEpilogueLdsBytes;
    if (first) n += (size_t)64 * xs_stride(kcx) * 4;
    return n;
}
constexpr size_t kOneWorkgroupPerCuLds = 82 * 1024;      // > 160 KB / 2

static size_t generic_lds_bytes(int hidden, bool last) {
    size_t n = (size_t)3 * (hidden / 16) * 64 * 16 + (size_t)3 * hidden * 4;
    if (last) n += kEpilogueLdsBytes;
    return n;
}

int gru_resident_kcx(int in_dim, bool first) { return first ? (in_dim + 3) / 4 : 32; }

bool gru_resident_supported(int hidden, int in_dim, bool first) {
    if (hidden != 128) return false;
    if (!first) return in_dim == 128;
    // instantiated KCX = 8, 10, 12, 15, 16 (rows must be whole float4s, and the cooperative mel staging needs
    // 4 streams x KCX float4 pieces <= 64 lanes): the reference's 40 (README.md:17) and 60 (config/rnn_config.py:63),
    // plus the other common front-end widths up to 64
    return in_dim == 32 || in_dim == 40 || in_dim == 48 || in_dim == 60 || in_dim == 64;
}

template <typename K>
static hipError_t launch_with_lds(K kernel, const GruLayerParams& p, size_t lds, hipStream_t st) {
    static LdsGrant granted;             // per kernel instantiation (one static per template instance) and device
    {
        const hipError_t e = grant_dynamic_lds(kernel, granted, lds);
        if (e != hipSuccess) return e;
    }
    const int groups = (p.B + kStreamsPerGroup - 1) / kStreamsPerGroup;
    hipLaunchKernelGGL(kernel, dim3(groups), dim3(256), lds, st, p);
    return hipGetLastError();
}

// the resident kernels loop over stream groups themselves: one workgroup per CU at most (each fills a CU's register file),
// the weights staged once per workgroup however many groups it takes
static int device_cu_count() {
    static std::atomic<int> cached[kMaxDevices];
    int dev = 0;
    if (hipGetDevice(&dev) != hipSuccess || dev < 0 || dev >= kMaxDevices) return 256;
    int n = cached[dev].load(std::memory_order_relaxed);
    if (n <= 0) {
        hipDeviceProp_t prop;
        n = hipGetDeviceProperties(&prop, dev) == hipSuccess && prop.multiProcessorCount > 0 ? prop.multiProcessorCount : 256;
        cached[dev].store(n, std::memory_order_relaxed);
    }
    return n;
}
template <typename K>
static hipError_t launch_resident(K kernel, const GruLayerParams& p, size_t lds, hipStream_t st) {
    static LdsGrant granted;
    {
        const hipError_t e = grant_dynamic_lds(kernel, granted, lds);
        if (e != hipSuccess) return e;
    }
    const int groups = (p.B + kStreamsPerGroup - 1) / kStreamsPerGroup, cus = device_cu_count();
    hipLaunchKernelGGL(kernel, dim3(groups < cus ? groups : cus), dim3(256), lds, st, p);
    return hipGetLastError();
}

bool gru_resident_takes_window(bool first, bool last) { return last && !first; }

hipError_t launch_gru_layer_resident(const GruLayerParams& p, bool first, bool last, hipStream_t st) {
    const size_t lds = resident_lds_bytes(p.KCX, first, last);
    if (p.win.tab != nullptr) {               // with the window tail: the last layer of a stack only
        if (!gru_resident_takes_window(first, last) || p.seq_len) return hipErrorInvalidValue;
        return launch_resident(gru_layer_resident<32, false, true, true>, p, lds + kWinTailWordsBytes, st);
    }
#define KWS_RES(KCX_, F_, L_) return launch_resident(gru_layer_resident<KCX_, F_, L_>, p, lds, st)
    if (first) {
        if (p.KCX == 8) { if (last) KWS_RES(8, true, true); else KWS_RES(8, true, false); }
        if (p.KCX == 10) { if (last) KWS_RES(10, true, true); else KWS_RES(10, true, false); }
        if (p.KCX == 12) { if (last) KWS_RES(12, true, true); else KWS_RES(12, true, false); }
        if (p.KCX == 15) { if (last) KWS_RES(15, true, true); else KWS_RES(15, true, false); }
        if (p.KCX == 16) { if (last) KWS_RES(16, true, true); else KWS_RES(16, true, false); }
        return hipErrorInvalidValue;
    }
    if (last) KWS_RES(32, false, true); else KWS_RES(32, false, false);
#undef KWS_RES
}

template <int TPW>
static hipError_t launch_pipelined(const GruStackParams& sp, size_t lds, hipStream_t st) {
    static LdsGrant granted;             // per kernel instantiation (one static per template instance) and device
    {
        const hipError_t e = grant_dynamic_lds(gru_stack_generic_pipelined<TPW>, granted, lds);
        if (e != hipSuccess) return e;
    }
    const int per = sp.xcd_affine ? 8 / sp.L : 0;
    const int grid = sp.xcd_affine ? 8 * ((sp.G + per - 1) / per) : sp.G * sp.L;
    hipLaunchKernelGGL(gru_stack_generic_pipelined<TPW>, dim3(grid), dim3(256), lds, st, sp);
    return hipGetLastError();
}

hipError_t launch_gru_stack_generic_pipelined(const GruStackParams& sp, int hidden, hipStream_t st) {
    // ask for more than half a CU's LDS: one workgroup per CU, so that the L x G workgroups spread over L x G CUs
    // instead of doubling up on some of them
    size_t lds = generic_lds_bytes(hidden, true);
    if (lds < kOneWorkgroupPerCuLds) lds = kOneWorkgroupPerCuLds;
    if (hidden == 64) return launch_pipelined<1>(sp, lds, st);
    if (hidden == 128) return launch_pipelined<2>(sp, lds, st);
    if (hidden == 256) return launch_pipelined<4>(sp, lds, st);
    return hipErrorInvalidValue;
}

hipError_t launch_gru_layer_generic(const GruLayerParams& p, int hidden, bool first, bool last,
                                    hipStream_t st) {
    size_t lds = generic_lds_bytes(hidden, last);
    // a time block of an overlapped call (t_stride set): another layer's kernel runs beside this one on another stream;
    // keep one workgroup per CU or the dispatcher stacks both kernels onto the same CUs (measured: 2.2x slower each;
    // within ONE launch it spreads workgroups by itself, and padding then only costs occupancy when groups > CUs)
    if (p.t_stride != 0 && lds < kOneWorkgroupPerCuLds) lds = kOneWorkgroupPerCuLds;
#define KWS_GEN(TPW_) \
    do { \
        if (first && last) return launch_with_lds(gru_layer_generic<TPW_, true, true>, p, lds, st); \
        if (first) return launch_with_lds(gru_layer_generic<TPW_, true, false>, p, lds, st); \
        if (last) return launch_with_lds(gru_layer_generic<TPW_, false, true>, p, lds, st); \
        return launch_with_lds(gru_layer_generic<TPW_, false, false>, p, lds, st); \
    } while (0)
    if (hidden == 64) KWS_GEN(1);
    if (hidden == 128) KWS_GEN(2);
    if (hidden == 256) KWS_GEN(4);
#undef KWS_GEN
    return hipErrorInvalidValue;
}

}  // namespace kws

// Microbenchmark for VERDICT r4 item 1: would TWO 16-stream recurrence chains per wave, sharing one resident weight set, pay in
// gru_layer_f16x3?  The real per-frame multiset of the upper layer (KX = 4): per chain and frame 144 v_mfma_f32_16x16x32_f16
// (x-part 72 = gates 48 + candidate 24, recurrent gates 48, recurrent candidate 24), 168 single VALU instructions (r path 60, u
// sigmoid 32, candidate path 76), the LDS operand reads (4 per streamed weight group, 2 per input chunk), two 2 x 16 B LDS
// stores and two workgroup barriers with their write -> barrier -> read round trips.  Everything is issued through the
// compiler's builtins and real LDS pointers (it places the waitcnts and sees the hazards, as in the product kernel); the
// schedules are pinned with sched_barrier(0).
//
//   MODE 0  one chain, the product kernel's schedule: the next frame's x-part (second accumulator set) is the filler woven into the
//           two activation phases and around the two barriers                                          -> cycles per frame
//   MODE 1  two chains A and B half a frame apart, one accumulator set each, four phases per frame pair:
//             P1  MFMA  x_ru(A) rest, gates_h(A)        VALU  B: u sigmoid, candidate path -> h_B        barrier
//             P2  MFMA  x_ru(B), gates_h(B)             VALU  A: r path -> r(.)h_A, u sigmoid            barrier
//             P3  MFMA  x_c(A), x_c(B) half, cand_h(A)  VALU  B: r path -> r(.)h_B                       barrier
//             P4  MFMA  x_ru(A) head, x_c(B), cand_h(B) VALU  A: candidate path -> h_A                   barrier
//           every LDS round trip of one chain passes behind the other chain's matrix work             -> cycles per frame PAIR
//   MODE 2  MODE 1 without the LDS-streamed weight operands (as if all 96 operands per wave were register-resident)
//
// CAPACITY IS IGNORED HERE ON PURPOSE (that is the second half of the answer): the upper layer's 96 operands
// per wave are 64 in AGPRs + 32 that must sit in LDS / VGPRs; two chains need 2 x (hb + rhb + xsb) = 48 KB of LDS, which leaves
// room for 25 of those 32, and the rest does not fit 256 VGPRs beside two chains' state.  In this benchmark the streamed groups
// alias five LDS groups (80 KB) -- same instruction stream, fake data.
//
// build: hipcc --offload-arch=gfx950 -O3 -std=c++17 -mllvm -amdgpu-mfma-vgpr-form=1 -fno-slp-vectorize -o _bin/two_chains two_chains.hip
#include <hip/hip_runtime.h>
#include <cstdio>
#include <type_traits>
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef float f32x2 __attribute__((ext_vector_type(2)));
typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));
typedef _Float16 f16x2 __attribute__((ext_vector_type(2)));
typedef unsigned u32x4 __attribute__((ext_vector_type(4)));

template <int I0, int I1, class F>
__device__ __forceinline__ void static_for(F&& f) {
    if constexpr (I0 < I1) { f(std::integral_constant<int, I0>{}); static_for<I0 + 1, I1>(f); }
}
__device__ __forceinline__ void pin() { __builtin_amdgcn_sched_barrier(0); }
// NA elements of stream a and NB of stream b, evenly merged, each pinned in place
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
__device__ __forceinline__ f16x8 as_f16x8(u32x4 v) { return __builtin_bit_cast(f16x8, v); }
__device__ __forceinline__ f32x4 mfma(f16x8 a, f16x8 b, f32x4 c) { return __builtin_amdgcn_mfma_f32_16x16x32_f16(a, b, c, 0, 0, 0); }
__device__ __forceinline__ void lds_barrier() {
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();
    asm volatile("" ::: "memory");
}

template <int MODE>
__global__ void __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(1, 1)))
k(const u32x4* __restrict__ wsrc, float* __restrict__ dst, long long* cyc, int iters) {
    constexpr int NCH = MODE == 0 ? 1 : 2;
    constexpr bool STREAM = MODE != 2;
    const int tid = threadIdx.x, lane = tid & 63;
    const int w = __builtin_amdgcn_readfirstlane(tid >> 6);
    extern __shared__ __attribute__((aligned(16))) char smem[];
    u32x4* hb = reinterpret_cast<u32x4*>(smem);                  // [NCH][4 chunks][hi|lo][64]
    u32x4* rhb = hb + NCH * 4 * 2 * 64;
    u32x4* xsb = rhb + NCH * 4 * 2 * 64;                         // [NCH][4][2][64]
    u32x4* wul = xsb + NCH * 4 * 2 * 64;                         // [4 waves][5 groups][4][64]
    float* biasl = reinterpret_cast<float*>(wul + 4 * 5 * 4 * 64);

    // resident operands: recurrent 48 + candidate x-part 16 = all 256 AGPRs
    f16x8 wh[2][3][4][2], wxc[2][4][2];
#pragma unroll
    for (int j = 0; j < 2; ++j)
#pragma unroll
        for (int q = 0; q < 3; ++q)
#pragma unroll
            for (int m = 0; m < 4; ++m)
#pragma unroll
                for (int hl = 0; hl < 2; ++hl) wh[j][q][m][hl] = as_f16x8(wsrc[(((j * 3 + q) * 4 + m) * 2 + hl) * 64 + lane]);
#pragma unroll
    for (int j = 0; j < 2; ++j)
#pragma unroll
        for (int c = 0; c < 4; ++c)
#pragma unroll
            for (int hl = 0; hl < 2; ++hl) wxc[j][c][hl] = as_f16x8(wsrc[(48 + (j * 4 + c) * 2 + hl) * 64 + lane]);
#pragma unroll
    for (int j = 0; j < 2; ++j) {
#pragma unroll
        for (int q = 0; q < 3; ++q)
#pragma unroll
            for (int m = 0; m < 4; ++m)
#pragma unroll
                for (int hl = 0; hl < 2; ++hl) { f16x8& op = wh[j][q][m][hl]; asm volatile("" : "+a"(op)); }
#pragma unroll
        for (int c = 0; c < 4; ++c)
#pragma unroll
            for (int hl = 0; hl < 2; ++hl) { f16x8& op = wxc[j][c][hl]; asm volatile("" : "+a"(op)); }
    }
    for (int i = tid; i < 4 * 5 * 4 * 64; i += 256) wul[i] = wsrc[(64 + (i >> 6) % 20) * 64 + (i & 63)];
    for (int i = tid; i < NCH * 4 * 2 * 64; i += 256) { hb[i] = wsrc[i & 4095]; rhb[i] = wsrc[(i + 7) & 4095]; xsb[i] = wsrc[(i + 13) & 4095]; }
    for (int i = tid; i < 3 * 128; i += 256) biasl[i] = 0.01f * (i % 13);
    const f32x4* bl = reinterpret_cast<const f32x4*>(biasl);
    const u32x4* wul_w = wul + w * (5 * 4 * 64) + lane;
    const int g = lane >> 4;
    float cLoInv = 1.f / 2048.f, cLoScale = 2048.f, cNegLoScale = -2048.f, cNegTwo = -2.f;
    asm volatile("" : "+s"(cLoInv), "+s"(cLoScale), "+s"(cNegLoScale), "+s"(cNegTwo));
    __syncthreads();

    // ---- per-chain state -----------------------------------------------------------------------------------------------
    f32x4 hreg[NCH][2];
#pragma unroll
    for (int ch = 0; ch < NCH; ++ch) { hreg[ch][0] = bl[ch]; hreg[ch][1] = bl[ch + 2]; }
    // accumulators: MODE 0: two sets of [j][q] (frame parity); MODE 1/2: one set per chain
    constexpr int NSET = 2;
    f32x4 am[NSET][2][3], al[NSET][2][3];
    float va[8], vb[8], uu[NCH][8];
    unsigned phi[4], plo[4] = {0u, 0u, 0u, 0u};
    f16x8 xb[2][2], hB[2][2], wtmp[2][2][2];

    constexpr auto C0 = std::integral_constant<int, 0>{};
    constexpr auto C1 = std::integral_constant<int, 1>{};
    // operand requests a phase ahead of their first use: input chunk `c` (0 / 1) of chain CH -> xb[c]; the first two streamed groups
    auto pf_x = [&](auto ch_, auto c_) {
        constexpr int CH = decltype(ch_)::value, c = decltype(c_)::value;
        xb[c][0] = as_f16x8(xsb[(CH * 8 + c * 2 + 0) * 64 + lane]);
        xb[c][1] = as_f16x8(xsb[(CH * 8 + c * 2 + 1) * 64 + lane]);
    };
    auto pf_w = [&]() {
        if constexpr (STREAM) {
#pragma unroll
            for (int kk = 0; kk < 2; ++kk)
#pragma unroll
                for (int jh = 0; jh < 4; ++jh) wtmp[kk][jh >> 1][jh & 1] = as_f16x8(wul_w[((kk % 5) * 4 + jh) * 64]);
        }
    };
    // ---- element streams (SET = accumulator set: frame parity in MODE 0, chain in MODE 1/2; CH = chain for LDS buffers / state)
    auto acc_init = [&](auto set_, auto q0_, auto q1_) {        // 1 LDS read per accumulator (bias), lo = 0
        constexpr int SET = decltype(set_)::value;
        static_for<decltype(q0_)::value, decltype(q1_)::value>([&](auto q_) {
            constexpr int q = decltype(q_)::value;
#pragma unroll
            for (int j = 0; j < 2; ++j) { am[SET][j][q] = bl[(q * 128 + (2 * w + j) * 16) / 4 + g]; al[SET][j][q] = (f32x4){0.f, 0.f, 0.f, 0.f}; }
        });
    };
    // x-part of gates r, u: 48 MFMAs, chunk-major, operands streamed from LDS through two 4-operand register sets
    auto XRU = [&](auto set_, auto ch_, auto i_) {
        constexpr int SET = decltype(set_)::value, CH = decltype(ch_)::value, i = decltype(i_)::value;
        constexpr int c = i / 12, r = i % 12, q = r / 6, sweep = (r % 6) / 2, j = r % 2, k = c * 2 + q;
        if constexpr (i == 0 && MODE == 0) { pf_x(ch_, C0); pf_x(ch_, C1); pf_w(); }      // (two chains: requested a phase ahead, see the phase streams)
        const f16x8 B = xb[c & 1][sweep == 2 ? 1 : 0];
        f16x8 W;
        if constexpr (STREAM) W = wtmp[k & 1][j][sweep == 1 ? 1 : 0];
        else W = wh[j][q][c][sweep == 1 ? 1 : 0];
        if constexpr (sweep == 0) am[SET][j][q] = mfma(W, B, am[SET][j][q]);
        else al[SET][j][q] = mfma(W, B, al[SET][j][q]);
        if constexpr (STREAM && r % 6 == 5 && k + 2 < 8) {
#pragma unroll
            for (int jh = 0; jh < 4; ++jh) wtmp[k & 1][jh >> 1][jh & 1] = as_f16x8(wul_w[(((k + 2) % 5) * 4 + jh) * 64]);
        }
        if constexpr (r == 11 && c + 2 < 4) {
            xb[c & 1][0] = as_f16x8(xsb[(CH * 8 + (c + 2) * 2 + 0) * 64 + lane]);
            xb[c & 1][1] = as_f16x8(xsb[(CH * 8 + (c + 2) * 2 + 1) * 64 + lane]);
        }
    };
    // x-part of the candidate: 24 MFMAs, operands resident (AGPR)
    auto XC = [&](auto set_, auto ch_, auto i_) {
        constexpr int SET = decltype(set_)::value, CH = decltype(ch_)::value, i = decltype(i_)::value;
        constexpr int c = i / 6, r = i % 6, sweep = r / 2, j = r % 2;
        if constexpr (i == 0 && MODE == 0) { pf_x(ch_, C0); pf_x(ch_, C1); }
        const f16x8 B = xb[c & 1][sweep == 2 ? 1 : 0];
        const f16x8 W = wxc[j][c][sweep == 1 ? 1 : 0];
        if constexpr (sweep == 0) am[SET][j][2] = mfma(W, B, am[SET][j][2]);
        else al[SET][j][2] = mfma(W, B, al[SET][j][2]);
        if constexpr (r == 5 && c + 2 < 4) {
            xb[c & 1][0] = as_f16x8(xsb[(CH * 8 + (c + 2) * 2 + 0) * 64 + lane]);
            xb[c & 1][1] = as_f16x8(xsb[(CH * 8 + (c + 2) * 2 + 1) * 64 + lane]);
        }
    };
    auto hread = [&](const u32x4* src, int m, int buf) {
        hB[buf][0] = as_f16x8(src[(m * 2 + 0) * 64 + lane]);
        hB[buf][1] = as_f16x8(src[(m * 2 + 1) * 64 + lane]);
    };
    auto G = [&](auto set_, auto ch_, auto i_) {
        constexpr int SET = decltype(set_)::value, CH = decltype(ch_)::value, i = decltype(i_)::value;
        constexpr int m = i / 12, r = i % 12, sweep = r / 4, j = (r % 4) / 2, q = r % 2;
        const f16x8 B = hB[m & 1][sweep == 2 ? 1 : 0];
        const f16x8 W = wh[j][q][m][sweep == 1 ? 1 : 0];
        if constexpr (sweep == 0) am[SET][j][q] = mfma(W, B, am[SET][j][q]);
        else al[SET][j][q] = mfma(W, B, al[SET][j][q]);
        if constexpr (r == 11 && m + 2 < 4) hread(hb + CH * 512, m + 2, m & 1);
    };
    auto Cm = [&](auto set_, auto ch_, auto i_) {
        constexpr int SET = decltype(set_)::value, CH = decltype(ch_)::value, i = decltype(i_)::value;
        constexpr int m = i / 6, r = i % 6, sweep = r / 2, j = r % 2;
        const f16x8 B = hB[m & 1][sweep == 2 ? 1 : 0];
        const f16x8 W = wh[j][2][m][sweep == 1 ? 1 : 0];
        if constexpr (sweep == 0) am[SET][j][2] = mfma(W, B, am[SET][j][2]);
        else al[SET][j][2] = mfma(W, B, al[SET][j][2]);
        if constexpr (r == 5 && m + 2 < 4) hread(rhb + CH * 512, m + 2, m & 1);
    };
    auto mix_lo = [&](auto un_, float m) {
        constexpr int un = decltype(un_)::value;
        const unsigned hi = phi[un >> 1];
        const float nk = cNegLoScale;
        unsigned d = plo[un >> 1];
        if constexpr ((un & 1) == 0) asm("v_fma_mixlo_f16 %0, %1, %2, %3 op_sel:[0,0,0] op_sel_hi:[1,0,0]" : "=v"(d) : "v"(hi), "s"(nk), "v"(m));
        else asm("v_fma_mixhi_f16 %0, %1, %2, %3 op_sel:[1,0,0] op_sel_hi:[1,0,0]" : "+v"(d) : "v"(hi), "s"(nk), "v"(m));
        plo[un >> 1] = d;
    };
    auto R = [&](auto set_, auto ch_, auto i_) {        // 60 single VALU instructions
        constexpr int SET = decltype(set_)::value, CH = decltype(ch_)::value, i = decltype(i_)::value;
        constexpr int st = i < 40 ? i / 8 : i < 44 ? 5 : 6 + (i - 44) / 8;
        constexpr int un = i < 40 ? i % 8 : i < 44 ? i - 40 : (i - 44) % 8;
        if constexpr (st == 5) phi[un] = __builtin_bit_cast(unsigned, __builtin_convertvector((f32x2){va[2 * un], va[2 * un + 1]}, f16x2));
        else {
            constexpr int j = un >> 2, e = un & 3;
            if constexpr (st == 0) va[un] = __builtin_fmaf(al[SET][j][0][e], cLoInv, am[SET][j][0][e]);
            else if constexpr (st == 1) va[un] = __builtin_amdgcn_exp2f(va[un]);
            else if constexpr (st == 2) va[un] = va[un] + 1.0f;
            else if constexpr (st == 3) va[un] = __builtin_amdgcn_rcpf(va[un]);
            else if constexpr (st == 4) va[un] = va[un] * hreg[CH][j][e];
            else if constexpr (st == 6) vb[un] = va[un] * cLoScale;
            else mix_lo(std::integral_constant<int, un>{}, vb[un]);
        }
    };
    auto U = [&](auto set_, auto ch_, auto i_) {        // 32
        constexpr int SET = decltype(set_)::value, CH = decltype(ch_)::value, i = decltype(i_)::value, st = i / 8, un = i % 8, j = un >> 2, e = un & 3;
        if constexpr (st == 0) uu[CH][un] = __builtin_fmaf(al[SET][j][1][e], cLoInv, am[SET][j][1][e]);
        else if constexpr (st == 1) uu[CH][un] = __builtin_amdgcn_exp2f(uu[CH][un]);
        else if constexpr (st == 2) uu[CH][un] = uu[CH][un] + 1.0f;
        else uu[CH][un] = __builtin_amdgcn_rcpf(uu[CH][un]);
    };
    auto Cc = [&](auto set_, auto ch_, auto i_) {       // 76
        constexpr int SET = decltype(set_)::value, CH = decltype(ch_)::value, i = decltype(i_)::value;
        constexpr int st = i < 56 ? i / 8 : i < 60 ? 100 : 101 + (i - 60) / 8;
        constexpr int un = i < 56 ? i % 8 : i < 60 ? i - 56 : (i - 60) % 8;
        if constexpr (st == 100) phi[un] = __builtin_bit_cast(unsigned, __builtin_convertvector((f32x2){hreg[CH][un >> 1][2 * (un & 1)], hreg[CH][un >> 1][2 * (un & 1) + 1]}, f16x2));
        else {
            constexpr int j = un >> 2, e = un & 3;
            if constexpr (st == 0) va[un] = __builtin_fmaf(al[SET][j][2][e], cLoInv, am[SET][j][2][e]);
            else if constexpr (st == 1) va[un] = __builtin_amdgcn_exp2f(va[un]);
            else if constexpr (st == 2) va[un] = va[un] + 1.0f;
            else if constexpr (st == 3) va[un] = __builtin_amdgcn_rcpf(va[un]);
            else if constexpr (st == 4) va[un] = __builtin_fmaf(va[un], cNegTwo, 1.0f);
            else if constexpr (st == 5) vb[un] = hreg[CH][j][e] - va[un];
            else if constexpr (st == 6) hreg[CH][j][e] = __builtin_fmaf(uu[CH][un], vb[un], va[un]);
            else if constexpr (st == 101) vb[un] = hreg[CH][j][e] * cLoScale;
            else mix_lo(std::integral_constant<int, un>{}, vb[un]);
        }
    };
    auto store = [&](u32x4* dst, int ch) {
        dst[(ch * 8 + w * 2 + 0) * 64 + lane] = (u32x4){phi[0], phi[1], phi[2], phi[3]};
        dst[(ch * 8 + w * 2 + 1) * 64 + lane] = (u32x4){plo[0], plo[1], plo[2], plo[3]};
    };
    constexpr auto Q0 = std::integral_constant<int, 0>{};
    constexpr auto Q2 = std::integral_constant<int, 2>{};
    constexpr auto Q3 = std::integral_constant<int, 3>{};

    acc_init(C0, Q0, Q3);
    acc_init(C1, Q0, Q3);
    const long long t0 = __builtin_readcyclecounter();
    if constexpr (MODE == 0) {
        // ---- one chain, the product's schedule: x-part of frame t+1 (72 = x_ru 48 then x_c 24, accumulator set PN) woven in ----
        auto X = [&](auto pn_, auto i_) {
            constexpr int i = decltype(i_)::value;
            if constexpr (i < 48) XRU(pn_, C0, i_); else XC(pn_, C0, std::integral_constant<int, i - 48>{});
        };
        auto frame = [&](auto pc_) {
            constexpr int PC = decltype(pc_)::value;
            constexpr auto pc = std::integral_constant<int, PC>{};
            constexpr auto pn = std::integral_constant<int, PC ^ 1>{};
            hread(hb, 0, 0); hread(hb, 1, 1); pin();
            static_for<66, 72>([&](auto i_) { X(pc, i_); pin(); });                 // tail of this frame's own x-part
            zip<48, 0, 6, 0>([&](auto i_) { G(pc, C0, i_); }, [&](auto i_) {         // gates_h with the next x-part's setup reads
                constexpr int i = decltype(i_)::value;
                if constexpr (i == 0) acc_init(pn, Q0, Q3);
            });
            zip<60, 0, 20, 0>([&](auto i_) { R(pc, C0, i_); }, [&](auto i_) { X(pn, i_); });
            store(rhb, 0); pin();
            static_for<20, 26>([&](auto i_) { X(pn, i_); pin(); });
            lds_barrier(); pin();
            hread(rhb, 0, 0); hread(rhb, 1, 1); pin();
            static_for<26, 34>([&](auto i_) { X(pn, i_); pin(); });
            zip<24, 0, 32, 0>([&](auto i_) { Cm(pc, C0, i_); }, [&](auto i_) { U(pc, C0, i_); });
            zip<76, 0, 26, 34>([&](auto i_) { Cc(pc, C0, i_); }, [&](auto i_) { X(pn, i_); });
            store(hb, 0); pin();
            static_for<60, 66>([&](auto i_) { X(pn, i_); pin(); });
            lds_barrier(); pin();
        };
        for (int it = 0; it < iters; it += 2) { frame(C0); frame(C1); }
    } else {
        // ---- two chains: set / chain 0 = A, 1 = B ----
        // MFMA streams of the four phases (compile-time index -> element)
        // every stream's first operands are requested a phase (or half a phase) ahead, into registers the stream before has
        // just released: hB at the start of the phase whose second half reads it, xb / wtmp under the recurrent MFMAs before
        auto P1m = [&](auto i_) { constexpr int i = decltype(i_)::value;            // x_ru(A) 12..47, gates_h(A)
            if constexpr (i == 0) { hread(hb, 0, 0); hread(hb, 1, 1); }
            if constexpr (i == 40) { pf_x(C1, C0); pf_x(C1, C1); pf_w(); }          // x_ru(B) of P2
            if constexpr (i < 36) XRU(C0, C0, std::integral_constant<int, 12 + i>{}); else G(C0, C0, std::integral_constant<int, i - 36>{}); };
        auto P2m = [&](auto i_) { constexpr int i = decltype(i_)::value;            // x_ru(B), gates_h(B)
            if constexpr (i == 0) { hread(hb + 512, 0, 0); hread(hb + 512, 1, 1); }
            if constexpr (i == 52) { pf_x(C0, C0); pf_x(C0, C1); }                   // x_c(A) of P3
            if constexpr (i < 48) XRU(C1, C1, i_); else G(C1, C1, std::integral_constant<int, i - 48>{}); };
        auto P3m = [&](auto i_) { constexpr int i = decltype(i_)::value;            // x_c(A), x_c(B) 0..11, cand_h(A)
            if constexpr (i == 0) { hread(rhb, 0, 0); hread(rhb, 1, 1); }
            if constexpr (i == 18) pf_x(C1, C0);                                     // x_c(B): chunk registers as x_c(A) releases them
            if constexpr (i == 24) pf_x(C1, C1);
            if constexpr (i < 24) XC(C0, C0, i_); else if constexpr (i < 36) XC(C1, C1, std::integral_constant<int, i - 24>{});
            else Cm(C0, C0, std::integral_constant<int, i - 36>{}); };
        auto P4m = [&](auto i_) { constexpr int i = decltype(i_)::value;            // x_c(B) 12..23, x_ru(A) 0..11 (next frame), cand_h(B)
            if constexpr (i == 0) { hread(rhb + 512, 0, 0); hread(rhb + 512, 1, 1); pf_w(); }
            if constexpr (i == 6) pf_x(C0, C0);                                      // x_ru(A): chunk registers as x_c(B) releases them
            if constexpr (i == 12) pf_x(C0, C1);
            if constexpr (i < 12) XC(C1, C1, std::integral_constant<int, 12 + i>{}); else if constexpr (i < 24) XRU(C0, C0, std::integral_constant<int, i - 12>{});
            else Cm(C1, C1, std::integral_constant<int, i - 24>{}); };
        for (int it = 0; it < iters; ++it) {
            // P1: B's u sigmoid + candidate path under A's matrix work
            zip<84, 0, 108, 0>(P1m, [&](auto i_) { constexpr int i = decltype(i_)::value;
                if constexpr (i < 32) U(C1, C1, i_); else Cc(C1, C1, std::integral_constant<int, i - 32>{}); });
            store(hb, 1); acc_init(C1, Q0, Q3); pin();
            lds_barrier(); pin();
            // P2: A's r path + u sigmoid under B's matrix work
            zip<96, 0, 92, 0>(P2m, [&](auto i_) { constexpr int i = decltype(i_)::value;
                if constexpr (i < 60) R(C0, C0, i_); else U(C0, C0, std::integral_constant<int, i - 60>{}); });
            store(rhb, 0); acc_init(C0, Q0, Q2); pin();
            lds_barrier(); pin();
            // P3: B's r path under A's candidate
            zip<60, 0, 60, 0>(P3m, [&](auto i_) { R(C1, C1, i_); });
            store(rhb, 1); pin();
            lds_barrier(); pin();
            // P4: A's candidate path under B's candidate
            zip<48, 0, 76, 0>(P4m, [&](auto i_) { Cc(C0, C0, i_); });
            store(hb, 0); pin();
            {   // A's candidate accumulators restart (its gate accumulators restarted in P2, B's after P1)
#pragma unroll
                for (int j = 0; j < 2; ++j) { am[0][j][2] = bl[(2 * 128 + (2 * w + j) * 16) / 4 + g]; al[0][j][2] = (f32x4){0.f, 0.f, 0.f, 0.f}; }
            }
            lds_barrier(); pin();
        }
    }
    const long long t1 = __builtin_readcyclecounter();
    float r = 0;
#pragma unroll
    for (int ch = 0; ch < NCH; ++ch) r += hreg[ch][0][0] + hreg[ch][1][3];
#pragma unroll
    for (int s2 = 0; s2 < NSET; ++s2)
#pragma unroll
        for (int j = 0; j < 2; ++j)
#pragma unroll
            for (int q = 0; q < 3; ++q) r += am[s2][j][q][0] + al[s2][j][q][1];
    dst[blockIdx.x * 256 + tid] = r;
    if (tid == 0 && blockIdx.x == 0) *cyc = t1 - t0;
}

template <typename K>
static double run(const char* name, K kern, int nch, const u32x4* src, float* dst, long long* cyc, int per_iter_frames) {
    const size_t lds = (size_t)(3 * nch * 4 * 2 * 64 + 4 * 5 * 4 * 64) * 16 + 3 * 128 * 4;
    hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    const int iters = 600;
    hipEvent_t a, b;
    hipEventCreate(&a); hipEventCreate(&b);
    hipLaunchKernelGGL(kern, dim3(256), dim3(256), lds, 0, src, dst, cyc, 20);
    hipDeviceSynchronize();
    hipEventRecord(a);
    hipLaunchKernelGGL(kern, dim3(256), dim3(256), lds, 0, src, dst, cyc, iters);
    hipEventRecord(b);
    hipDeviceSynchronize();
    long long c = 0;
    float ms = 0;
    hipMemcpy(&c, cyc, 8, hipMemcpyDeviceToHost);
    hipEventElapsedTime(&ms, a, b);
    const double frames = (double)iters * per_iter_frames;
    printf("%-64s %8.0f counter ticks per stream-group frame, %7.3f us per stream-group frame (256 workgroups, whole kernel incl. prologue)\n",
           name, (double)c / frames, ms * 1e3 / frames);
    return ms * 1e3 / frames;
}

int main() {
    u32x4* src; float* dst; long long* cyc;
    hipMalloc(&src, 8192 * 16); hipMalloc(&dst, 256 * 256 * 4); hipMalloc(&cyc, 8);
    static unsigned hsrc[8192 * 4];
    for (int i = 0; i < 8192 * 4; ++i) {          // fp16 pairs of small magnitude
        const _Float16 a = (_Float16)(0.01f * ((i * 7) % 23 - 11)), b = (_Float16)(0.01f * ((i * 5) % 19 - 9));
        unsigned short ua, ub;
        __builtin_memcpy(&ua, &a, 2); __builtin_memcpy(&ub, &b, 2);
        hsrc[i] = (unsigned)ua | ((unsigned)ub << 16);
    }
    hipMemcpy(src, hsrc, sizeof(hsrc), hipMemcpyHostToDevice);
    const double one = run("MODE 0: one chain, the product's weave (per frame)", k<0>, 1, src, dst, cyc, 1);
    const double two = run("MODE 1: two chains half a frame apart (per frame of ONE chain)", k<1>, 2, src, dst, cyc, 2);
    const double reg = run("MODE 2: two chains, no LDS-streamed operands (per frame of one chain)", k<2>, 2, src, dst, cyc, 2);
    printf("two chains / one chain: %.3f of the time per stream-group frame (%.1f %% more frames per second); without streamed operands %.3f\n",
           two / one, 100.0 * (one / two - 1.0), reg / one);
    return 0;
}

// PCM -> mel front-end of the deploy graph: models/rnn_ctc.py:134-149 --
//   frames = tf_frame(x, 400, 160)            utils/stft.py:27-81  (no padding, NO window function)
//   linearspec = |rfft(frames, 400)|          models/rnn_ctc.py:137
//   melspec = linearspec @ mel_basis^T        models/rnn_ctc.py:139-149 (librosa.filters.mel, Slaney, area-normalised)
//
// One workgroup = kFT x 16 = 64 frames of the FLATTENED [B*T] frame index (a 22-frame chunk wastes nothing, and
// the cos/sin table, 346 KB that every workgroup streams from L2, is read once per 64 frames instead of once per
// 16: the first version was L2-bandwidth-bound at 8.7 TB/s).  The DFT is a dense fp32 contraction on
// v_mfma_f32_16x16x4_f32 with D[bin][frame]: A = cos / sin rows in group-of-4 fragment order, B = the frames.
// Real input: with e[n] = x[n] + x[N-n], o[n] = x[n] - x[N-n] (0 < n < N/2), e[0] = x[0], e[N/2] = x[N/2]:
//   Re X[k] = sum_{n<=N/2} e[n] cos(2 pi k n / N),   Im X[k] = -sum_{n<N/2} o[n] sin(2 pi k n / N)
// so both contractions run over N/2 (+1) samples instead of N -- half the MFMAs; the fold happens while the
// windows are staged into LDS (row stride odd: conflict-free column reads).  Work units are (bin tile, cos|sin):
// 26 for fft 400, dealt round-robin to the four waves (7/7/6/6; even waves only ever read e, odd waves only o).
// Re^2 and Im^2 meet in LDS in the xl layout (kws_internal.h), which is directly the B operand of the mel
// projection -- no transpose.  A radix-16x25 two-stage factorisation would cut the matrix work a further ~2.5x.
#include "gru_device.h"

namespace kws {

constexpr int kFT = 4;      // frame tiles (of 16) per workgroup
#ifndef KWS_FE_SF
#define KWS_FE_SF 8
#endif
constexpr int kSF = KWS_FE_SF;   // frames staged per round and wave
constexpr int kMaxUPW = 8;  // (bin tile, cos|sin) units per wave: 2 * nf_tiles <= 32, i.e. fft <= 496

// UPW = ceil(2 * nf_tiles / 4): the table is zero-padded to 4 * UPW units so that the unit loop carries no
// predicate (with one, hipcc keeps the accumulators in VGPRs, copies them through AGPRs around every unit and
// drains vmcnt(0) -- i.e. the table prefetch -- before each group of MFMAs)
template <int UPW>
__global__ void __launch_bounds__(256) mel_frontend_kernel(const FrontendParams p) {
    const int tid = threadIdx.x;
    const int w = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int lane = tid & 63, g = lane >> 4, f = lane & 15;
    const int N = p.fft, HOP = p.hop, NFT = p.nf_tiles;
    const int NH = N / 2;                    // folded length (cos part also uses sample NH)
    const int KC4 = p.kc4;                   // groups of 16 folded samples: ceil((NH+1)/16)
    const int stride = 16 * KC4 + 1;         // odd
    const long long total = (long long)p.B * p.T;
    const long long f0 = (long long)blockIdx.x * (16 * kFT);

    extern __shared__ __attribute__((aligned(16))) char smem[];
    float* xe = reinterpret_cast<float*>(smem);                  // [64][stride] even part
    float* xo = xe + 16 * kFT * stride;                          // [64][stride] odd part
    f32x4* sq = reinterpret_cast<f32x4*>(smem);                  // after the DFT: [NFT][cos|sin][kFT][64] squares

    // stage + fold: wave w takes frames 16w..16w+15, kSF frames per round with all 8*kSF loads of the round in
    // flight (a load -> fold -> store loop exposes the full memory latency 64 times per wave)
    const int NS = 16 * KC4;                 // <= 256 for fft <= 496: at most 4 samples per lane and frame
#ifdef KWS_FE_NOSTAGE
    for (int i0 = 0; i0 < 0; i0 += kSF) {
#else
    for (int i0 = 0; i0 < 16; i0 += kSF) {
#endif
        float xa[kSF][4], xc[kSF][4];
        bool ok[kSF];
#pragma unroll
        for (int i = 0; i < kSF; ++i) {
            const long long fidx = f0 + 16 * w + i0 + i;
            ok[i] = fidx < total;
            const long long b = ok[i] ? fidx / p.T : 0;
            const int t = ok[i] ? (int)(fidx - b * p.T) : 0;
            const float* x = p.pcm + (size_t)b * p.n_samples + (size_t)t * HOP;
#pragma unroll
            for (int q = 0; q < 4; ++q) {          // unconditional loads at clamped addresses: no branches, no waits
                const int n = lane + 64 * q;
                const int na = n <= NH ? n : NH;
                xa[i][q] = x[na];
                xc[i][q] = x[na == 0 ? 0 : N - na];
            }
        }
#pragma unroll
        for (int i = 0; i < kSF; ++i) {
            const int fr = 16 * w + i0 + i;
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                const int n = lane + 64 * q;
                if (n < NS) {
                    const bool in = ok[i] && n <= NH, edge = n == 0 || n == NH;
                    xe[fr * stride + n] = in ? (edge ? xa[i][q] : xa[i][q] + xc[i][q]) : 0.f;
                    xo[fr * stride + n] = (in && !edge) ? xa[i][q] - xc[i][q] : 0.f;
                }
            }
        }
    }
    __syncthreads();

    // DFT over the folded samples: units w, w+4, ... ; all of one wave's units are cos (even w) or sin (odd w)
    const int cs = w & 1;
    f32x4 acc[UPW][kFT];
#pragma unroll
    for (int j = 0; j < UPW; ++j)
#pragma unroll
        for (int ft = 0; ft < kFT; ++ft) acc[j][ft] = splat4(0.f);
    const f32x4* dft = reinterpret_cast<const f32x4*>(p.dft) + (size_t)w * KC4 * 64 + lane;   // [4*UPW units][KC4][64]
    const size_t ustride = (size_t)4 * KC4 * 64;                                                // unit u = w + 4j = 2*tile + cs
    const float* src = (cs ? xo : xe) + f * stride + g;
    // table fragments ping-pong between two register sets, one k4 group ahead (pinned: left alone, hipcc sinks the
    // loads to the end of the iteration and waits vmcnt(0) on them at once)
    f32x4 a0[UPW], a1[UPW];
    auto fetch = [&](f32x4 (&a)[UPW], int k4) {
        const int kk = k4 < KC4 ? k4 : KC4 - 1;
#pragma unroll
        for (int j = 0; j < UPW; ++j) a[j] = dft[j * ustride + (size_t)kk * 64];
        __builtin_amdgcn_sched_barrier(0);
    };
    auto contract = [&](const f32x4 (&a)[UPW], int k4) {
        f32x4 bv[kFT];
#pragma unroll
        for (int ft = 0; ft < kFT; ++ft)
#pragma unroll
            for (int e = 0; e < 4; ++e) bv[ft][e] = src[ft * 16 * stride + 16 * k4 + 4 * e];
#pragma unroll
        for (int j = 0; j < UPW; ++j)
#pragma unroll
            for (int e = 0; e < 4; ++e)
#pragma unroll
                for (int ft = 0; ft < kFT; ++ft)      // accumulators pinned in AGPRs (builtin MFMAs: ~200 accvgpr moves per group)
                    asm volatile("v_mfma_f32_16x16x4_f32 %0, %1, %2, %0" : "+a"(acc[j][ft]) : "v"(a[j][e]), "v"(bv[ft][e]));
        __builtin_amdgcn_sched_barrier(0);
    };
#ifndef KWS_FE_NODFT
    fetch(a0, 0);
#pragma unroll
    for (int j = 0; j < UPW; ++j) asm volatile("s_nop 3" : "+a"(acc[j][0]), "+a"(acc[j][1]), "+a"(acc[j][2]), "+a"(acc[j][3]));
    int k4 = 0;
    for (; k4 + 1 < KC4; k4 += 2) {
        fetch(a1, k4 + 1);
        contract(a0, k4);
        fetch(a0, k4 + 2);
        contract(a1, k4 + 1);
    }
    if (k4 < KC4) contract(a0, k4);
#endif
#pragma unroll
    for (int j = 0; j < UPW; ++j) asm volatile("s_nop 15" : "+a"(acc[j][0]), "+a"(acc[j][1]), "+a"(acc[j][2]), "+a"(acc[j][3]));
    __syncthreads();                       // every wave is done with the windows: the squares reuse their space
#pragma unroll
    for (int j = 0; j < UPW; ++j) {
        const int u = w + 4 * j;
        if (u < 2 * NFT) {
#pragma unroll
            for (int ft = 0; ft < kFT; ++ft) sq[((size_t)u * kFT + ft) * 64 + lane] = acc[j][ft] * acc[j][ft];
        }
    }
    __syncthreads();

    // magnitudes once, by all four waves, in place of the Re^2 slot (v_sqrt_f32: 1 ulp; an IEEE sqrtf sequence
    // per element, repeated by every mel-tile wave, cost more than the whole DFT)
    for (int i = w; i < NFT * kFT; i += 4) {
        const int nt = i / kFT, ft = i - nt * kFT;
        const f32x4 s2 = sq[((size_t)(2 * nt) * kFT + ft) * 64 + lane] + sq[((size_t)(2 * nt + 1) * kFT + ft) * 64 + lane];
        f32x4 m;
#pragma unroll
        for (int e = 0; e < 4; ++e) m[e] = __builtin_amdgcn_sqrtf(s2[e]);
        sq[((size_t)(2 * nt) * kFT + ft) * 64 + lane] = m;
    }
    __syncthreads();
    // mel projection: wave w = mel tile w over all frame tiles; B operand = the magnitudes in xl layout.  The
    // basis fragments of the tile are fetched in groups of 4 bin tiles ahead of their use (a load per chunk
    // inside the loop exposes the L2 latency 13 times)
#ifndef KWS_FE_NOMEL
    if (w < p.mel_tiles) {
        f32x4 o[kFT];
#pragma unroll
        for (int ft = 0; ft < kFT; ++ft) o[ft] = splat4(0.f);
        const float* melw = p.melw + (size_t)w * (4 * NFT) * 64 + lane;           // [mel tile][4*NFT chunks][64]
        float mw[16], mn[16];
        auto fetch_mel = [&](float (&m)[16], int nt0) {
#pragma unroll
            for (int c = 0; c < 16; ++c) {
                const int ch = 4 * nt0 + c;
                m[c] = melw[(size_t)(ch < 4 * NFT ? ch : 4 * NFT - 1) * 64];
            }
        };
        fetch_mel(mw, 0);
        for (int nt0 = 0; nt0 < NFT; nt0 += 4) {
            fetch_mel(mn, nt0 + 4);
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                const int nt = nt0 + q;
                if (nt < NFT) {
#pragma unroll
                    for (int ft = 0; ft < kFT; ++ft) {
                        const f32x4 m = sq[((size_t)(2 * nt) * kFT + ft) * 64 + lane];
#pragma unroll
                        for (int e = 0; e < 4; ++e) o[ft] = mfma4(mw[4 * q + e], m[e], o[ft]);
                    }
                }
            }
#pragma unroll
            for (int c = 0; c < 16; ++c) mw[c] = mn[c];
        }
#pragma unroll
        for (int ft = 0; ft < kFT; ++ft) {
            const long long fidx = f0 + 16 * ft + f;
            if (fidx < total) {
                float* out = p.mel + (size_t)fidx * p.n_mel + 16 * w + 4 * g;
#pragma unroll
                for (int e = 0; e < 4; ++e)
                    if (16 * w + 4 * g + e < p.n_mel) out[e] = o[ft][e];
            }
        }
    }
#endif
}

template <int UPW>
static hipError_t launch_upw(const FrontendParams& p, unsigned grid, size_t lds, hipStream_t st) {
    static size_t granted = 0;
    if (lds > granted) {
        hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(mel_frontend_kernel<UPW>),
                                           hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
        if (e != hipSuccess) return e;
        granted = lds;
    }
    hipLaunchKernelGGL(mel_frontend_kernel<UPW>, dim3(grid), dim3(256), lds, st, p);
    return hipGetLastError();
}

int frontend_units_per_wave(int nf_tiles) { return (2 * nf_tiles + 3) / 4; }

hipError_t launch_mel_frontend(const FrontendParams& p, int B, hipStream_t st) {
    const int stride = 16 * p.kc4 + 1;
    const size_t windows = (size_t)2 * 16 * kFT * stride * 4;
    const size_t squares = (size_t)p.nf_tiles * 2 * kFT * 64 * 16;
    const size_t lds = windows > squares ? windows : squares;
    const long long total = (long long)B * p.T;
    const unsigned grid = (unsigned)((total + 16 * kFT - 1) / (16 * kFT));
    switch (frontend_units_per_wave(p.nf_tiles)) {
        case 1: return launch_upw<1>(p, grid, lds, st);
        case 2: return launch_upw<2>(p, grid, lds, st);
        case 3: return launch_upw<3>(p, grid, lds, st);
        case 4: return launch_upw<4>(p, grid, lds, st);
        case 5: return launch_upw<5>(p, grid, lds, st);
        case 6: return launch_upw<6>(p, grid, lds, st);
        case 7: return launch_upw<7>(p, grid, lds, st);
        case 8: return launch_upw<8>(p, grid, lds, st);
        default: return hipErrorInvalidValue;
    }
}

}  // namespace kws

// PCM -> mel front-end of the deploy graph: models/rnn_ctc.py:134-149 --
//   frames = tf_frame(x, 400, 160)            utils/stft.py:27-81  (no padding, NO window function)
//   linearspec = |rfft(frames, 400)|          models/rnn_ctc.py:137
//   melspec = linearspec @ mel_basis^T        models/rnn_ctc.py:139-149 (librosa.filters.mel, Slaney, area-normalised)
//
// One workgroup = kFT x 16 = 64 frames of the FLATTENED [B*T] frame index (a 22-frame chunk wastes nothing, and
// the cos/sin table that every workgroup streams from L2 is read once per 64 frames: the first version, 16 frames
// per workgroup, was L2-bandwidth-bound at 8.7 TB/s).  The DFT is a dense fp32 contraction on
// v_mfma_f32_16x16x4_f32 with D[bin][frame]: A = cos / sin rows in group-of-4 fragment order, B = the frames.
// Two exact reductions of the matrix work, N = fft, NH = N/2, NQ = N/4:
//  * real input (fold):  e[n] = x[n] + x[N-n], o[n] = x[n] - x[N-n] (0 < n < NH), e[0] = x[0], e[NH] = x[NH]:
//        Re X[k] = sum_{n<=NH} e[n] cos(2 pi k n / N),   Im X[k] = -sum_{n<NH} o[n] sin(2 pi k n / N)
//  * bin mirror (one radix-2 step): cos(2 pi (NH-k) n / N) = (-1)^n cos(2 pi k n / N), and the sine likewise up to
//    sign, so with the sums split by the parity of n,  C0/C1 (cos, n even/odd) and S0/S1 (sin):
//        |X[k]|^2 = (C0+C1)^2 + (S0+S1)^2,     |X[NH-k]|^2 = (C0-C1)^2 + (S0-S1)^2        for k = 0..NQ
//    -- only bins 0..NQ are contracted, each over half the samples.
// Work units are (bin tile, cos|sin, parity): 4 per bin tile, so wave w owns ONE (cos|sin, parity) combination
// for every tile (7 units for fft 400) and reads one of the four folded arrays E0/E1/O0/O1 only.  The partial
// sums meet in LDS in the xl layout (kws_internal.h); the magnitudes of bins k and NH-k are written back in that
// same layout and are directly the B operands of the mel projection, whose basis fragments exist in a direct and a
// mirrored set -- no transpose anywhere.  A further radix step (k, NQ-k, ...) would halve the matrix work again;
// staging and the mel stage are now as expensive as the DFT itself.
#include "gru_device.h"

namespace kws {

#ifndef KWS_FE_FT
#define KWS_FE_FT 4
#endif
constexpr int kFT = KWS_FE_FT;      // frame tiles (of 16) per workgroup
#ifndef KWS_FE_SF
#define KWS_FE_SF 4
#endif
constexpr int kSF = KWS_FE_SF;   // frames staged per round and wave

// UPW = bin tiles over k = 0..NQ (= units per wave).  No predicate in the unit loop: with one, hipcc keeps the
// accumulators in VGPRs, copies them through AGPRs around every unit and drains vmcnt(0) before each MFMA group.
template <int UPW>
__global__ void __launch_bounds__(256) mel_frontend_kernel(const FrontendParams p) {
    const int tid = threadIdx.x;
    const int w = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int lane = tid & 63, g = lane >> 4, f = lane & 15;
    const int N = p.fft, HOP = p.hop;
    const int NH = N / 2, NQ = N / 4;
    const int KC4 = p.kc4;                   // groups of 16 samples of one parity class: ceil((NQ+1)/16) == UPW
    const int stride = 16 * KC4 + 1;         // odd
    const long long total = (long long)p.B * p.T;
    const long long f0 = (long long)blockIdx.x * (16 * kFT);

    extern __shared__ __attribute__((aligned(16))) char smem[];
    float* X = reinterpret_cast<float*>(smem);                   // [4: E0 E1 O0 O1][64 frames][stride]
    f32x4* P = reinterpret_cast<f32x4*>(smem);                   // after the DFT: [4*UPW units][kFT][64] partial sums
    const int asz = 16 * kFT * stride;

    // stage + fold: wave w takes frames 16w..16w+15, kSF frames per round with all 8*kSF loads of the round in
    // flight (a load -> fold -> store loop exposes the full memory latency once per element)
#ifndef KWS_FE_NOSTAGE
    // (stream, frame) of the wave's first frame by one division, then counted up
    long long sb = (f0 + 16 * w) / p.T;
    int st = (int)(f0 + 16 * w - sb * p.T);
    for (int i0 = 0; i0 < 16; i0 += kSF) {
        float xa[kSF][2][2], xc[kSF][2][2];
        bool ok[kSF];
#pragma unroll
        for (int i = 0; i < kSF; ++i) {
            ok[i] = f0 + 16 * w + i0 + i < total;
            // the signal of stream sb is carry[sb] (n_carry samples, may be 0) followed by pcm[sb] (detector.py:179)
            const long long row = ok[i] ? sb : 0;
            const float* xc_ = p.carry + (size_t)row * p.n_carry;
            const float* xp_ = p.pcm + (size_t)row * (p.n_samples - p.n_carry);
            const int s0 = ok[i] ? st * HOP : 0;
            if (++st == p.T) { st = 0; ++sb; }
            // only the first frames of a chunk straddle the seam; every other frame reads one array (uniform branch)
            const bool seam = s0 < p.n_carry && s0 + N > p.n_carry;
            if (!seam) {
                const float* x = s0 >= p.n_carry ? xp_ + (s0 - p.n_carry) : xc_ + s0;
#pragma unroll
                for (int q = 0; q < 2; ++q) {          // unconditional loads at clamped addresses: no branches, no waits
                    const int m = lane + 64 * q, mc = m <= NQ ? m : NQ;
#pragma unroll
                    for (int par = 0; par < 2; ++par) {
                        int n = 2 * mc + par;
                        n = n <= NH ? n : NH;
                        xa[i][q][par] = x[n];
                        xc[i][q][par] = x[n == 0 ? 0 : N - n];
                    }
                }
            } else {
#pragma unroll
                for (int q = 0; q < 2; ++q) {
                    const int m = lane + 64 * q, mc = m <= NQ ? m : NQ;
#pragma unroll
                    for (int par = 0; par < 2; ++par) {
                        int n = 2 * mc + par;
                        n = n <= NH ? n : NH;
                        const int ia = s0 + n, ic = s0 + (n == 0 ? 0 : N - n);
                        xa[i][q][par] = ia < p.n_carry ? xc_[ia] : xp_[ia - p.n_carry];
                        xc[i][q][par] = ic < p.n_carry ? xc_[ic] : xp_[ic - p.n_carry];
                    }
                }
            }
        }
#pragma unroll
        for (int i = 0; i < kSF; ++i) {
            const int fr = 16 * w + i0 + i;
#pragma unroll
            for (int q = 0; q < 2; ++q) {
                const int m = lane + 64 * q;
                if (m < 16 * KC4) {
#pragma unroll
                    for (int par = 0; par < 2; ++par) {
                        const int n = 2 * m + par;
                        const bool in = ok[i] && n <= NH, edge = n == 0 || n == NH;
                        const float a = xa[i][q][par], c = xc[i][q][par];
                        X[(0 + par) * asz + fr * stride + m] = in ? (edge ? a : a + c) : 0.f;
                        X[(2 + par) * asz + fr * stride + m] = (in && !edge) ? a - c : 0.f;
                    }
                }
            }
        }
    }
#endif
    __syncthreads();

    // DFT: wave w = (cos|sin = w >> 1, parity = w & 1), units u = 4 * tile + w
    f32x4 acc[UPW][kFT];
#pragma unroll
    for (int j = 0; j < UPW; ++j)
#pragma unroll
        for (int ft = 0; ft < kFT; ++ft) acc[j][ft] = splat4(0.f);
    const f32x4* dft = reinterpret_cast<const f32x4*>(p.dft) + (size_t)w * KC4 * 64 + lane;   // [4*UPW units][KC4][64]
    const size_t ustride = (size_t)4 * KC4 * 64;
    const float* src = X + w * asz + f * stride + g;
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
    for (int j = 0; j < UPW; ++j)
#pragma unroll
        for (int ft = 0; ft < kFT; ++ft) asm volatile("s_nop 1" : "+a"(acc[j][ft]));
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
    for (int j = 0; j < UPW; ++j)
#pragma unroll
        for (int ft = 0; ft < kFT; ++ft) asm volatile("s_nop 15" : "+a"(acc[j][ft]));
    __syncthreads();                       // every wave is done with the windows: the partial sums reuse their space
#pragma unroll
    for (int j = 0; j < UPW; ++j)
#pragma unroll
        for (int ft = 0; ft < kFT; ++ft) P[((size_t)(4 * j + w) * kFT + ft) * 64 + lane] = acc[j][ft];
    __syncthreads();

    // butterfly + magnitudes, once, by all four waves: slot 4t+0 <- |X[k]|, slot 4t+1 <- |X[NH-k]|
    // (v_sqrt_f32, 1 ulp: an IEEE sqrtf sequence per element cost more than the whole DFT)
    for (int i = w; i < UPW * kFT; i += 4) {
        const int t = i / kFT, ft = i - t * kFT;
        f32x4* q = P + ((size_t)(4 * t) * kFT + ft) * 64 + lane;
        const f32x4 c0 = q[0], c1 = q[kFT * 64], s0 = q[2 * kFT * 64], s1 = q[3 * kFT * 64];
        const f32x4 rp = c0 + c1, rm = c0 - c1, ip = s0 + s1, im = s0 - s1;
        const f32x4 d2 = rp * rp + ip * ip, m2 = rm * rm + im * im;
        f32x4 d, m;
#pragma unroll
        for (int e = 0; e < 4; ++e) { d[e] = __builtin_amdgcn_sqrtf(d2[e]); m[e] = __builtin_amdgcn_sqrtf(m2[e]); }
        q[0] = d;
        q[kFT * 64] = m;
    }
    __syncthreads();

    // mel projection: wave w = mel tile w over all frame tiles; B operands = the two magnitude blocks in xl layout,
    // A = the direct / mirrored basis fragments, fetched two bin tiles ahead of their use
#ifndef KWS_FE_NOMEL
    if (w < p.mel_tiles) {
        f32x4 o[kFT];
#pragma unroll
        for (int ft = 0; ft < kFT; ++ft) o[ft] = splat4(0.f);
        const float* melw = p.melw + (size_t)w * (8 * UPW) * 64 + lane;           // [mel tile][UPW][direct|mirror][4][64]
        float mw[16], mn[16];
        auto fetch_mel = [&](float (&mm)[16], int t0) {
#pragma unroll
            for (int c = 0; c < 16; ++c) {
                const int ch = 8 * t0 + c;
                mm[c] = melw[(size_t)(ch < 8 * UPW ? ch : 8 * UPW - 1) * 64];
            }
        };
        fetch_mel(mw, 0);
        for (int t0 = 0; t0 < UPW; t0 += 2) {
            fetch_mel(mn, t0 + 2);
#pragma unroll
            for (int q = 0; q < 2; ++q) {
                const int t = t0 + q;
                if (t < UPW) {
#pragma unroll
                    for (int ft = 0; ft < kFT; ++ft) {
                        const f32x4 d = P[((size_t)(4 * t) * kFT + ft) * 64 + lane];
                        const f32x4 m = P[((size_t)(4 * t + 1) * kFT + ft) * 64 + lane];
#pragma unroll
                        for (int e = 0; e < 4; ++e) o[ft] = mfma4(mw[8 * q + e], d[e], o[ft]);
#pragma unroll
                        for (int e = 0; e < 4; ++e) o[ft] = mfma4(mw[8 * q + 4 + e], m[e], o[ft]);
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

// next carry = the last n_next samples of [carry | chunk]   (detector.py:181-183)
__global__ void __launch_bounds__(256) carry_tail_kernel(const float* __restrict__ carry, int n_carry, const float* __restrict__ chunk,
                                                         int n_chunk, float* __restrict__ next, int n_next) {
    const int b = blockIdx.y;
    for (int j = blockIdx.x * 256 + threadIdx.x; j < n_next; j += gridDim.x * 256) {
        const int i = n_carry + n_chunk - n_next + j;
        next[(size_t)b * n_next + j] = i < n_carry ? carry[(size_t)b * n_carry + i] : chunk[(size_t)b * n_chunk + i - n_carry];
    }
}
hipError_t launch_carry_tail(const float* carry, int n_carry, const float* chunk, int n_chunk, float* next, int n_next, int B,
                             hipStream_t st) {
    hipLaunchKernelGGL(carry_tail_kernel, dim3((n_next + 255) / 256 > 0 ? (n_next + 255) / 256 : 1, B), dim3(256), 0, st, carry, n_carry,
                       chunk, n_chunk, next, n_next);
    return hipGetLastError();
}

template <int UPW>
static hipError_t launch_upw(const FrontendParams& p, unsigned grid, size_t lds, hipStream_t st) {
    static LdsGrant granted;             // per kernel instantiation (one static per template instance) and device
    {
        const hipError_t e = grant_dynamic_lds(mel_frontend_kernel<UPW>, granted, lds);
        if (e != hipSuccess) return e;
    }
    hipLaunchKernelGGL(mel_frontend_kernel<UPW>, dim3(grid), dim3(256), lds, st, p);
    return hipGetLastError();
}

hipError_t launch_mel_frontend(const FrontendParams& p, int B, hipStream_t st) {
    const int stride = 16 * p.kc4 + 1;
    const size_t windows = (size_t)4 * 16 * kFT * stride * 4;
    const size_t sums = (size_t)4 * p.nf_tiles * kFT * 64 * 16;
    const size_t lds = windows > sums ? windows : sums;
    const long long total = (long long)B * p.T;
    const unsigned grid = (unsigned)((total + 16 * kFT - 1) / (16 * kFT));
    switch (p.nf_tiles) {
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

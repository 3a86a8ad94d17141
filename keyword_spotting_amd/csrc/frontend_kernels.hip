// PCM -> mel front-end of the deploy graph: models/rnn_ctc.py:134-149 --
//   frames = tf_frame(x, 400, 160)            utils/stft.py:27-81  (no padding, NO window function)
//   linearspec = |rfft(frames, 400)|          models/rnn_ctc.py:137
//   melspec = linearspec @ mel_basis^T        models/rnn_ctc.py:139-149 (librosa.filters.mel, Slaney, area-normalised)
//
// One workgroup = 16 consecutive frames of one stream.  The DFT is a dense fp32 contraction on
// v_mfma_f32_16x16x4_f32 with D[bin][frame]: A = cos / sin rows streamed from L2 in group-of-4 fragment order,
// B = the frames.  Real input: with e[n] = x[n] + x[N-n], o[n] = x[n] - x[N-n] (0 < n < N/2), e[0] = x[0],
// e[N/2] = x[N/2]:   Re X[k] = sum_{n<=N/2} e[n] cos(2 pi k n / N),   Im X[k] = -sum_{n<N/2} o[n] sin(2 pi k n / N)
// so both contractions run over N/2 (+1) samples instead of N -- half the MFMAs; the fold happens while the
// window is staged into LDS (row stride odd: conflict-free column reads).  Magnitudes land in the xl layout
// (kws_internal.h), which is directly the B operand of the mel projection -- no transpose.
// A radix-16x25 two-stage factorisation would cut the matrix work a further ~2.5x.
#include "gru_device.h"

namespace kws {

__global__ void __launch_bounds__(256) mel_frontend_kernel(const FrontendParams p) {
    const int tid = threadIdx.x;
    const int w = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int lane = tid & 63, g = lane >> 4, f = lane & 15;
    const int t0 = blockIdx.x * 16, b = blockIdx.y;
    const int N = p.fft, HOP = p.hop, NFT = p.nf_tiles;
    const int NH = N / 2;                    // folded length (cos part also uses sample NH)
    const int KC4 = p.kc4;                   // groups of 16 folded samples: ceil((NH+1)/16)
    const int stride = 16 * KC4 + 1;         // odd

    extern __shared__ __attribute__((aligned(16))) char smem[];
    float* xe = reinterpret_cast<float*>(smem);                                   // [16][stride] even part
    float* xo = xe + 16 * stride;                                                  // [16][stride] odd part
    f32x4* magbuf = reinterpret_cast<f32x4*>(xo + ((16 * stride + 3) & ~3));      // [NFT][64]

    const float* pcm = p.pcm + (size_t)b * p.n_samples;
    for (int i = tid; i < 16 * 16 * KC4; i += 256) {
        const int fr = i / (16 * KC4), n = i - fr * (16 * KC4);
        const int t = t0 + fr;
        float e = 0.f, o = 0.f;
        if (t < p.T && n <= NH) {
            const float* x = pcm + (size_t)t * HOP;
            const float a = x[n];
            if (n == 0 || n == NH) { e = a; }
            else { const float c = x[N - n]; e = a + c; o = a - c; }
        }
        xe[fr * stride + n] = e;
        xo[fr * stride + n] = o;
    }
    __syncthreads();

    // DFT over the folded samples: tiles w, w+4, w+8, w+12
    f32x4 re[4], im[4];
#pragma unroll
    for (int j = 0; j < 4; ++j) { re[j] = splat4(0.f); im[j] = splat4(0.f); }
    const f32x4* dft = reinterpret_cast<const f32x4*>(p.dft);                      // [NFT][2][KC4][64]
    const float* erow = xe + f * stride + g;
    const float* orow = xo + f * stride + g;
    for (int k4 = 0; k4 < KC4; ++k4) {
        f32x4 eb, ob;
#pragma unroll
        for (int e = 0; e < 4; ++e) { eb[e] = erow[16 * k4 + 4 * e]; ob[e] = orow[16 * k4 + 4 * e]; }
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            const int tile = w + 4 * j;
            if (tile < NFT) {
                const f32x4 ac = dft[((size_t)(tile * 2 + 0) * KC4 + k4) * 64 + lane];
                const f32x4 as = dft[((size_t)(tile * 2 + 1) * KC4 + k4) * 64 + lane];
#pragma unroll
                for (int e = 0; e < 4; ++e) {
                    re[j] = mfma4(ac[e], eb[e], re[j]);
                    im[j] = mfma4(as[e], ob[e], im[j]);
                }
            }
        }
    }
#pragma unroll
    for (int j = 0; j < 4; ++j) {
        const int tile = w + 4 * j;
        if (tile < NFT) {
            f32x4 m;
#pragma unroll
            for (int e = 0; e < 4; ++e) m[e] = sqrtf(re[j][e] * re[j][e] + im[j][e] * im[j][e]);
            magbuf[tile * 64 + lane] = m;
        }
    }
    __syncthreads();

    // mel projection: wave w computes mel tile w (16 bins) over all bins
    if (w < p.mel_tiles) {
        f32x4 acc = splat4(0.f);
        const float* melw = p.melw + (size_t)w * (4 * NFT) * 64;                  // [mel tile][4*NFT chunks][64]
        for (int nt = 0; nt < NFT; ++nt) {
            const f32x4 mb = magbuf[nt * 64 + lane];
#pragma unroll
            for (int e = 0; e < 4; ++e) acc = mfma4(melw[(size_t)(4 * nt + e) * 64 + lane], mb[e], acc);
        }
        const int t = t0 + f;
        if (t < p.T) {
            float* out = p.mel + ((size_t)b * p.T + t) * p.n_mel + 16 * w + 4 * g;
#pragma unroll
            for (int e = 0; e < 4; ++e)
                if (16 * w + 4 * g + e < p.n_mel) out[e] = acc[e];
        }
    }
}

hipError_t launch_mel_frontend(const FrontendParams& p, int B, hipStream_t st) {
    const int stride = 16 * p.kc4 + 1;
    const size_t lds = (size_t)(16 * stride + ((16 * stride + 3) & ~3)) * 4 + (size_t)p.nf_tiles * 64 * 16;
    static size_t granted = 0;
    if (lds > granted) {
        hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(mel_frontend_kernel),
                                           hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
        if (e != hipSuccess) return e;
        granted = lds;
    }
    hipLaunchKernelGGL(mel_frontend_kernel, dim3((p.T + 15) / 16, B), dim3(256), lds, st, p);
    return hipGetLastError();
}

}  // namespace kws

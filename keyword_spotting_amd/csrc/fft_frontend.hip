// PCM -> mel front-end of the deploy graph for the reference's frame length, as a real mixed-radix FFT.
//   frames = tf_frame(x, 400, 160)            utils/stft.py:27-81  (no padding, NO window function)
//   linearspec = |rfft(frames, 400)|          models/rnn_ctc.py:137
//   melspec = linearspec @ mel_basis^T        models/rnn_ctc.py:139-149
// (frontend_kernels.hip keeps the dense-DFT kernel for every other frame length.)
//
// 400 = 16 x 25, n = 16 n1 + n2, k = k1 + 25 k2 (Cooley-Tukey):
//     X[k1 + 25 k2] = sum_{n2<16} W16^{n2 k2} * ( W400^{n2 k1} * Y_{n2}[k1] ),   Y_{n2}[k1] = sum_{n1<25} x[16 n1 + n2] W25^{n1 k1}
// Stage 1: the 25-point DFT of a REAL decimated sequence, one per lane (lane = (frame j of 4, n2)), as 5 x 5 with
//   n1 = 5a + b, k1 = c + 5d:  Z_b[c] = sum_a x[5a+b] W5^{ac} (real input: c = 0,1,2 suffice),
//   Y[c + 5d] = sum_b W5^{bd} (W25^{bc} Z_b[c]).  Y is Hermitian, so only k1 = 0..12 is kept, and those thirteen come
//   from c in {0,1,2} alone: Y3 = conj Y22, Y4 = conj Y21, Y8 = conj Y17, Y9 = conj Y16.
// Stage 2: for k1 = 0..12 a 16-point complex FFT over n2 (radix 4 x 4) gives X[k1 + 25 k2], k2 = 0..15.  Those 208 values
//   hold every bin 0..200 exactly once: k <= 200 directly, k > 200 as the mirror 400 - k (|X[400-k]| = |X[k]|; the bins of
//   residue 13..24 mod 25), and k1 = 0, k2 >= 9 are duplicates that the mel table zeroes.
// Between the stages the 13 x 16 complex values of a frame cross lanes through LDS (one transpose; a workgroup is 16
// frames).  In stage 2 lane = (g, frame f of 16) takes k1 = 4q + g in pass q (= wave q); the magnitudes return to LDS in BIN
// order, where the mel projection (v_mfma_f32_16x16x4_f32 over 4-bin groups) only touches the blocks of the basis that are
// not all zero.  ~9 kflop per frame on the VALU instead of the 100 kflop of the dense contraction.
#include <cstdlib>

#include "gru_device.h"
#include "vad_device.h"

#pragma clang fp contract(off)      // only the fmaf() written below fuse: every frame sees one fixed instruction sequence

namespace kws {

namespace {

struct c32 { float re, im; };
__device__ __forceinline__ c32 operator+(c32 a, c32 b) { return {a.re + b.re, a.im + b.im}; }
__device__ __forceinline__ c32 operator-(c32 a, c32 b) { return {a.re - b.re, a.im - b.im}; }
__device__ __forceinline__ c32 cmul(c32 a, float c, float s) {           // a * (c + i s)
    return {fmaf(-a.im, s, a.re * c), fmaf(a.re, s, a.im * c)};
}
__device__ __forceinline__ c32 conj(c32 a) { return {a.re, -a.im}; }

constexpr float kC1 = 0.30901699437494742f, kC2 = -0.80901699437494742f;   // cos(2 pi/5), cos(4 pi/5)
constexpr float kS1 = 0.95105651629515357f, kS2 = 0.58778525229247313f;    // sin(2 pi/5), sin(4 pi/5)

// 5-point DFT of a real sequence: z0 (real), z1, z2 (z3 = conj z2, z4 = conj z1)
__device__ __forceinline__ void rdft5(float x0, float x1, float x2, float x3, float x4, float& z0, c32& z1, c32& z2) {
    const float sa = x1 + x4, da = x1 - x4, sb = x2 + x3, db = x2 - x3;
    z0 = x0 + sa + sb;
    z1.re = fmaf(kC2, sb, fmaf(kC1, sa, x0));
    z2.re = fmaf(kC1, sb, fmaf(kC2, sa, x0));
    z1.im = fmaf(-kS2, db, -kS1 * da);
    z2.im = fmaf(kS1, db, -kS2 * da);
}
// 5-point DFT of a complex sequence
__device__ __forceinline__ void cdft5(c32 t0, c32 t1, c32 t2, c32 t3, c32 t4, c32& y0, c32& y1, c32& y2, c32& y3, c32& y4) {
    const c32 sa = t1 + t4, da = t1 - t4, sb = t2 + t3, db = t2 - t3;
    y0 = t0 + sa + sb;
    const c32 a1 = {fmaf(kC2, sb.re, fmaf(kC1, sa.re, t0.re)), fmaf(kC2, sb.im, fmaf(kC1, sa.im, t0.im))};
    const c32 a2 = {fmaf(kC1, sb.re, fmaf(kC2, sa.re, t0.re)), fmaf(kC1, sb.im, fmaf(kC2, sa.im, t0.im))};
    const c32 b1 = {fmaf(kS2, db.re, kS1 * da.re), fmaf(kS2, db.im, kS1 * da.im)};
    const c32 b2 = {fmaf(-kS1, db.re, kS2 * da.re), fmaf(-kS1, db.im, kS2 * da.im)};
    // y1 = a1 - i b1, y4 = a1 + i b1, y2 = a2 - i b2, y3 = a2 + i b2
    y1 = {a1.re + b1.im, a1.im - b1.re};
    y4 = {a1.re - b1.im, a1.im + b1.re};
    y2 = {a2.re + b2.im, a2.im - b2.re};
    y3 = {a2.re - b2.im, a2.im + b2.re};
}
// radix-4 butterfly, forward transform (W4 = -i)
__device__ __forceinline__ void bfly4(c32 v0, c32 v1, c32 v2, c32 v3, c32& r0, c32& r1, c32& r2, c32& r3) {
    const c32 e0 = v0 + v2, e1 = v0 - v2, o0 = v1 + v3, o1 = v1 - v3;
    r0 = e0 + o0;
    r2 = e0 - o0;
    r1 = {e1.re + o1.im, e1.im - o1.re};
    r3 = {e1.re - o1.im, e1.im + o1.re};
}

// W25^e = cos - i sin, e = b c for b = 1..4, c = 1,2
constexpr float kW25c[9] = {1.f, 0.96858316112863108f, 0.87630668004386358f, 0.72896862742141155f, 0.53582679497899666f,
                            0.f, 0.06279051952931353f, 0.f, -0.42577929156507272f};
constexpr float kW25s[9] = {0.f, 0.24868988716485479f, 0.48175367410171532f, 0.68454710592868873f, 0.84432792550201508f,
                            0.f, 0.99802672842827156f, 0.f, 0.90482705246601958f};
constexpr float kR2 = 0.70710678118654752f;                                     // 1/sqrt 2
constexpr float kW16c1 = 0.92387953251128674f, kW16s1 = 0.38268343236508977f;   // cos, sin (pi/8)

constexpr int kRowBytes = 128;                 // one (k1, frame) row: 16 complex values over n2
constexpr int kPlaneBytes = 16 * kRowBytes;    // one k1: 16 frames
constexpr int kFftLds = 13 * kPlaneBytes;      // 26,624 B per workgroup: six workgroups per CU
constexpr int kMelRegs = 24;                   // basis fragments (4-bin groups) of a mel tile that wait in registers
constexpr int kSpecRows = 208;                 // spectrum rows (bins 0..200 + zero padding) of 16 frames: 13,312 B of the same space

}  // namespace

// MT = mel tiles of 16 filters.  One workgroup = 4 waves = 16 frames of the flattened [B*T] frame index: wave w runs
// stage 1 for frames 4w..4w+3, after the barrier stage 2 for k1 = 4w..4w+3 of all sixteen frames, and the magnitudes go
// back to LDS in BIN order (S[bin][frame]) so that the mel projection only touches the 4-bin x 16-filter blocks whose
// weights are not all zero: filters are contiguous in frequency, so tile m needs one contiguous run of 4-bin groups --
// 52 MFMAs per 16 frames for 40 filters where the dense product takes 156.  Wave m = tile m.
// 26 KiB of LDS per workgroup (the spectrum reuses the transpose planes): six workgroups = 24 waves per CU.
// SampleT: float samples, or int16 PCM as the sound card delivers it (detector.py:40-43,74-79: scaled by 2^-15 on load --
// read once, in place, no widened copy).  GATE: the head of a stream-manager iteration rides along in the same launch: the
// workgroup that transforms a stream's FIRST frame also takes the vad sum of its new samples (the masks silent / reset) and writes
// its next sample carry (detector.py:168-183) -- see the block below.
#ifndef KWS_FE_OCC
#define KWS_FE_OCC 6          // workgroups per CU the register allocation aims at (tools/build_variant.sh -DKWS_FE_OCC=n for A/B)
#endif
template <int MT, typename SampleT, bool GATE>
__global__ void __launch_bounds__(256, KWS_FE_OCC) mel_fft400_kernel(const FrontendParams p) {
    __shared__ __attribute__((aligned(16))) char lds[kFftLds];
    const int tid = threadIdx.x;
    const int w = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int lane = tid & 63;
    const int hi = lane >> 4, lo = lane & 15;          // stage 1: (frame j of 4, n2); stage 2 / MFMA: (g, frame f of 16)
    const unsigned total = (unsigned)p.B * (unsigned)p.T;
    // XCD-aware block -> frame-block map.  Workgroups go round-robin over the 8 XCDs (blockIdx % 8), each with its own L2;
    // a stream's frames span two or three 16-frame blocks, every frame re-reads 240 samples of its predecessor, and with GATE
    // the block of a stream's first frame reads its whole row.  Giving XCD x the CONTIGUOUS run of blocks [x F, (x+1) F),
    // F = fft_blocks / 8, keeps all readers of a row behind one L2, so the PCM leaves HBM once (the grid is a multiple of 8).
    constexpr float kScale = sizeof(SampleT) == 2 ? 1.0f / 32768.0f : 1.0f;
    const SampleT* chunk_all = sizeof(SampleT) == 2 ? reinterpret_cast<const SampleT*>(p.pcm_i16) : reinterpret_cast<const SampleT*>(p.pcm);
    const int n_chunk = p.n_samples - p.n_carry;
    const unsigned xcd = blockIdx.x & 7u, seq = blockIdx.x >> 3;       // XCD, position in that XCD's sequence
    const unsigned F = (unsigned)p.fft_blocks >> 3;                   // transform blocks per XCD
    const unsigned local = seq;                                         // this XCD's transform block index
    const unsigned blk = xcd * F + local;
    const unsigned f0 = blk * 16u;
    if (f0 >= total) return;                            // padding block of the rounded-up grid (uniform: before any barrier)
#ifdef KWS_FE_TIMING       // tools/ubench/fe_phases.hip: s_memtime at the phase boundaries of every wave
#define KWS_FE_STAMP(i) do { if (lane == 0) p.timing[((size_t)blk * 4 + w) * 8 + (i)] = __builtin_readcyclecounter(); } while (0)
#else
#define KWS_FE_STAMP(i) do {} while (0)
#endif
    KWS_FE_STAMP(0);

    // GATE: the stream whose FIRST frame lies in this block gets its vad sum, masks and next carry here.  The row is requested
    // before the frames' samples and summed while those are in flight; the four wave totals cross at the transform's own first
    // barrier.  block_abs_sum's association term by term (thread t owns groups t, t + 256, ...; shuffle tree; the four wave
    // totals pairwise), so kws_vad, the stream manager's gate kernel and this one still take the same decision on the same
    // samples.  The transform reads the same row right behind: the PCM leaves HBM once (67 MB per 4096 x 3600-sample launch; gate
    // workgroups in front of the transforms pulled each XCD's 7 MB of rows through its 4 MB L2 first: 94-100 MB; round 3: 140).
    __shared__ float gate_part[4];
    unsigned gate_b = 0xffffffffu;                      // workgroup-uniform: the stream handled the fast way
    float gate_v[4][4];
    if constexpr (GATE) {
        const unsigned T = (unsigned)p.T;
        const unsigned f_end = f0 + 16u < total ? f0 + 16u : total;
        const unsigned sb0 = (f0 + T - 1u) / T;
        const bool fast = n_chunk <= 4096 && (n_chunk & 3) == 0 && ((size_t)n_chunk * sizeof(SampleT)) % (4 * sizeof(SampleT)) == 0 &&
                          (reinterpret_cast<uintptr_t>(chunk_all) & (4 * sizeof(SampleT) - 1)) == 0;
        for (unsigned sb = sb0; sb < (unsigned)p.B && sb * T < f_end; ++sb) {
            const SampleT* row = chunk_all + (size_t)sb * n_chunk;
            if (fast && sb == sb0) {
                gate_b = sb;
                const int full = n_chunk >> 2;
#pragma unroll
                for (int k = 0; k < 4; ++k) {
                    const int i = tid + 256 * k;
#pragma unroll
                    for (int e = 0; e < 4; ++e) gate_v[k][e] = 0.f;
                    if (i < full) {
                        if constexpr (sizeof(SampleT) == 2) {
                            const short4 q4 = reinterpret_cast<const short4*>(row)[i];
                            gate_v[k][0] = (float)q4.x * kScale; gate_v[k][1] = (float)q4.y * kScale; gate_v[k][2] = (float)q4.z * kScale; gate_v[k][3] = (float)q4.w * kScale;
                        } else {
                            const float4 q4 = reinterpret_cast<const float4*>(row)[i];
                            gate_v[k][0] = q4.x; gate_v[k][1] = q4.y; gate_v[k][2] = q4.z; gate_v[k][3] = q4.w;
                        }
                    }
                }
            } else {
                // a second stream starting in this block (chunks shorter than 16 frames), long or unaligned rows: one after the other
                const float tot = block_abs_sum<SampleT>(row, n_chunk, nullptr);
                if (tid == 0) vad_masks(tot, p.vad_thres, sb, p.restart, p.silent, p.reset);
                carry_tail<SampleT>(p.carry + (size_t)sb * p.n_carry, p.n_carry, row, n_chunk, p.next + (size_t)sb * p.n_next, p.n_next, tid, 256);
                __syncthreads();
            }
        }
    }

    // ---- stage 1: this wave's four frames ----
    {
        const int n2 = lo;
        const int f = 4 * w + hi;
        unsigned fidx = f0 + f;
        fidx = fidx < total ? fidx : total - 1;         // frames past the end redo the last one (finite values, never stored)
        const unsigned sb = fidx / (unsigned)p.T;
        const int st = (int)(fidx - sb * (unsigned)p.T);
        // the signal of stream sb is carry[sb] (n_carry samples, may be 0) followed by pcm[sb] (detector.py:179)
        const float* xc_ = p.carry + (size_t)sb * p.n_carry;
        const SampleT* xp_ = chunk_all + (size_t)sb * n_chunk;
        const int s0 = st * p.hop + n2;
        float x[25];
        const bool seam = s0 - n2 < p.n_carry && s0 - n2 + 400 > p.n_carry;
        const unsigned long long any_seam = __builtin_amdgcn_ballot_w64(seam);
        const unsigned long long any_carry = __builtin_amdgcn_ballot_w64(s0 - n2 < p.n_carry);
        if (!any_carry) {
            // every frame of this wave lies in the new samples (all but the first rounds of a chunk)
            const SampleT* src = xp_ + (s0 - p.n_carry);
#pragma unroll
            for (int n1 = 0; n1 < 25; ++n1) x[n1] = (float)src[16 * n1] * kScale;
        } else if (!any_seam && sizeof(SampleT) == 4) {
            // whole frames, some in the carried samples, some in the new ones; one float array each
            const float* src = s0 - n2 >= p.n_carry ? reinterpret_cast<const float*>(xp_) + (s0 - p.n_carry) : xc_ + s0;
#pragma unroll
            for (int n1 = 0; n1 < 25; ++n1) x[n1] = src[16 * n1];
        } else {
            // some frame of this wave straddles the seam (the first two or three frames of a chunk): per-sample select
#pragma unroll
            for (int n1 = 0; n1 < 25; ++n1) {
                const int idx = s0 + 16 * n1;
                x[n1] = idx < p.n_carry ? xc_[idx] : (float)xp_[idx - p.n_carry] * kScale;
            }
        }
        // W400^{n2 k1}, k1 = 1..12, for this lane's n2 (cos, sin): [12][16] float2
        float2 tw[12];          // table: [6 pairs of k1][16 n2] float4 = (cos, sin) of k1 = 2i+1 and 2i+2: six 16-byte loads
#pragma unroll
        for (int k = 0; k < 6; ++k) {
            const f32x4 t = reinterpret_cast<const f32x4*>(p.dft)[k * 16 + n2];
            tw[2 * k] = make_float2(t[0], t[1]);
            tw[2 * k + 1] = make_float2(t[2], t[3]);
        }
#ifdef KWS_FE_TIMING
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        KWS_FE_STAMP(1);
#endif
        if (GATE && gate_b != 0xffffffffu) {
            float acc = 0.f;
#pragma unroll
            for (int k = 0; k < 4; ++k)      // a thread without group k adds +0: the same bits as not adding
                acc += (fabsf(gate_v[k][0]) + fabsf(gate_v[k][1])) + (fabsf(gate_v[k][2]) + fabsf(gate_v[k][3]));
            acc = wave_sum_lane0(acc);
            if (lane == 0) gate_part[w] = acc;
        }
        // Z_b[c] = sum_a x[5a + b] W5^{ac}
        float z0[5];
        c32 z1[5], z2[5];
#pragma unroll
        for (int b = 0; b < 5; ++b) rdft5(x[b], x[5 + b], x[10 + b], x[15 + b], x[20 + b], z0[b], z1[b], z2[b]);
        c32 Y[13];
        {   // c = 0: real inputs again
            float y0;
            rdft5(z0[0], z0[1], z0[2], z0[3], z0[4], y0, Y[5], Y[10]);
            Y[0] = {y0, 0.f};
        }
        {   // c = 1: Y1, Y6, Y11, Y16, Y21
            c32 y16, y21;
            cdft5(z1[0], cmul(z1[1], kW25c[1], -kW25s[1]), cmul(z1[2], kW25c[2], -kW25s[2]), cmul(z1[3], kW25c[3], -kW25s[3]),
                  cmul(z1[4], kW25c[4], -kW25s[4]), Y[1], Y[6], Y[11], y16, y21);
            Y[9] = conj(y16);
            Y[4] = conj(y21);
        }
        {   // c = 2: Y2, Y7, Y12, Y17, Y22
            c32 y17, y22;
            cdft5(z2[0], cmul(z2[1], kW25c[2], -kW25s[2]), cmul(z2[2], kW25c[4], -kW25s[4]), cmul(z2[3], kW25c[6], -kW25s[6]),
                  cmul(z2[4], kW25c[8], -kW25s[8]), Y[2], Y[7], Y[12], y17, y22);
            Y[8] = conj(y17);
            Y[3] = conj(y22);
        }
        // twiddle W400^{n2 k1} and park the row: plane k1, row f, column n2 ^ (f & 14) (the swizzle that makes the
        // stage-2 ds_read_b128 of sixteen different rows conflict-free)
        char* dst = lds + f * kRowBytes + ((n2 ^ (f & 14)) << 3);
        *reinterpret_cast<float2*>(dst) = make_float2(Y[0].re, Y[0].im);
#pragma unroll
        for (int k1 = 1; k1 < 13; ++k1) {
            const c32 t = cmul(Y[k1], tw[k1 - 1].x, -tw[k1 - 1].y);
            *reinterpret_cast<float2*>(dst + k1 * kPlaneBytes) = make_float2(t.re, t.im);
        }
    }
    KWS_FE_STAMP(2);
    __syncthreads();
    KWS_FE_STAMP(3);
    if (GATE && gate_b != 0xffffffffu) {
        if (tid == 0) vad_masks((gate_part[0] + gate_part[1]) + (gate_part[2] + gate_part[3]), p.vad_thres, gate_b, p.restart, p.silent, p.reset);
        carry_tail<SampleT>(p.carry + (size_t)gate_b * p.n_carry, p.n_carry, chunk_all + (size_t)gate_b * n_chunk, n_chunk,
                            p.next + (size_t)gate_b * p.n_next, p.n_next, tid, 256);
    }

    // ---- stage 2: pass q = w, k1 = 4w + g ----
    const int g = hi, f = lo, q = w;
    float mag[16];
    {
        const int k1 = 4 * q + g < 12 ? 4 * q + g : 12;         // lanes past k1 = 12 recompute plane 12 and store nothing
        const int sf = (f >> 1) & 7;
        const char* row = lds + k1 * kPlaneBytes + f * kRowBytes;
        c32 z[16];
#pragma unroll
        for (int pr = 0; pr < 8; ++pr) {
            const f32x4 v = *reinterpret_cast<const f32x4*>(row + ((pr ^ sf) << 4));
            z[2 * pr] = {v[0], v[1]};
            z[2 * pr + 1] = {v[2], v[3]};
        }
        // 16-point FFT, n2 = 4a + b, k2 = c + 4d
        c32 u[4][4];
#pragma unroll
        for (int b = 0; b < 4; ++b) bfly4(z[b], z[4 + b], z[8 + b], z[12 + b], u[b][0], u[b][1], u[b][2], u[b][3]);
        c32 o[16];
        bfly4(u[0][0], u[1][0], u[2][0], u[3][0], o[0], o[4], o[8], o[12]);
        {   // c = 1: W16^1, W16^2, W16^3
            const c32 v1 = cmul(u[1][1], kW16c1, -kW16s1);
            const c32 v2 = {(u[2][1].re + u[2][1].im) * kR2, (u[2][1].im - u[2][1].re) * kR2};
            const c32 v3 = cmul(u[3][1], kW16s1, -kW16c1);
            bfly4(u[0][1], v1, v2, v3, o[1], o[5], o[9], o[13]);
        }
        {   // c = 2: W16^2, W16^4 = -i, W16^6
            const c32 v1 = {(u[1][2].re + u[1][2].im) * kR2, (u[1][2].im - u[1][2].re) * kR2};
            const c32 v2 = {u[2][2].im, -u[2][2].re};
            const c32 v3 = {(u[3][2].im - u[3][2].re) * kR2, -(u[3][2].re + u[3][2].im) * kR2};
            bfly4(u[0][2], v1, v2, v3, o[2], o[6], o[10], o[14]);
        }
        {   // c = 3: W16^3, W16^6, W16^9 = -W16^1
            const c32 v1 = cmul(u[1][3], kW16s1, -kW16c1);
            const c32 v2 = {(u[2][3].im - u[2][3].re) * kR2, -(u[2][3].re + u[2][3].im) * kR2};
            const c32 v3 = cmul(u[3][3], -kW16c1, kW16s1);
            bfly4(u[0][3], v1, v2, v3, o[3], o[7], o[11], o[15]);
        }
#pragma unroll
        for (int k2 = 0; k2 < 16; ++k2) mag[k2] = __builtin_amdgcn_sqrtf(fmaf(o[k2].re, o[k2].re, o[k2].im * o[k2].im));
    }
    // basis fragments of this wave's mel tile: the first kMelRegs groups of its run wait in registers (the loads fly while the
    // spectrum is written and the workgroup meets at the barrier); a longer run streams the rest
    const int lo4 = w < MT ? p.mel_lo[w] : 0, n = w < MT ? p.mel_cnt[w] : 0;   // first 4-bin group, number of groups (multiple of 4; zero-padded table)
    // table: [tile][group / 4][64 lanes] float4 (the fragments of four consecutive groups side by side): one 16-byte load each
    const f32x4* A = reinterpret_cast<const f32x4*>(p.melw) + ((size_t)(w < MT ? p.mel_off[w] : 0) / 4 * 64 + lane);
    float a[kMelRegs];
    if (n > 0) {         // every tile's table holds at least kMelRegs groups (zero padded): no clamping, one base + immediate offsets
#pragma unroll
        for (int e4 = 0; e4 < kMelRegs / 4; ++e4) {
            const f32x4 v = A[e4 * 64];
            a[4 * e4] = v[0]; a[4 * e4 + 1] = v[1]; a[4 * e4 + 2] = v[2]; a[4 * e4 + 3] = v[3];
        }
    }
    KWS_FE_STAMP(4);
    lds_barrier();             // every wave has read its planes: the spectrum takes their place
    KWS_FE_STAMP(5);
    // ---- |X| in bin order: S[bin][frame], 64 B rows.  bin = k1 + 25 k2 folded at 200 ----
    float* S = reinterpret_cast<float*>(lds);
    {
        const int k1 = 4 * q + g;
        if (k1 <= 12) {
#pragma unroll
            for (int k2 = 0; k2 < 8; ++k2) S[(k1 + 25 * k2) * 16 + f] = mag[k2];
            if (k1 == 0) {
                S[200 * 16 + f] = mag[8];
            } else {
#pragma unroll
                for (int k2 = 8; k2 < 16; ++k2) S[(400 - 25 * k2 - k1) * 16 + f] = mag[k2];
            }
        }
        if (tid < 16 * (kSpecRows - 201)) S[201 * 16 + tid] = 0.f;     // padding rows the last block may read
    }
    lds_barrier();
    KWS_FE_STAMP(6);

    // ---- mel projection: wave m = mel tile m over its contiguous run of 4-bin groups.  A = basis fragments
    // [tile][group][64 lanes], B = four spectrum rows (k = g) x 16 frames; two accumulators break the dependent chain ----
#ifdef KWS_ABL_NOMEL          // experiment builds only (tools/build_variant.sh nomel -DKWS_ABL_NOMEL): the kernel without its mel projection
    if (n > 0 && p.n_mel < 0) {
#else
    if (n > 0) {
#endif
        const float* Sg = S + lo4 * 64 + lane;
        f32x4 acc0 = splat4(0.f), acc1 = splat4(0.f);
        // runs are padded to multiples of four groups: four spectrum reads in flight, then four MFMAs
        static_for<0, kMelRegs / 4>([&](auto c) {
            constexpr int e0 = 4 * decltype(c)::value;
            if (e0 < n) {
                const float b0 = Sg[e0 * 64], b1 = Sg[(e0 + 1) * 64], b2 = Sg[(e0 + 2) * 64], b3 = Sg[(e0 + 3) * 64];
                acc0 = mfma4(a[e0], b0, acc0);
                acc1 = mfma4(a[e0 + 1], b1, acc1);
                acc0 = mfma4(a[e0 + 2], b2, acc0);
                acc1 = mfma4(a[e0 + 3], b3, acc1);
            }
        });
        for (int e = kMelRegs; e < n; e += 4) {
            const float b0 = Sg[e * 64], b1 = Sg[(e + 1) * 64], b2 = Sg[(e + 2) * 64], b3 = Sg[(e + 3) * 64];
            const f32x4 v = A[(size_t)(e / 4) * 64];
            acc0 = mfma4(v[0], b0, acc0);
            acc1 = mfma4(v[1], b1, acc1);
            acc0 = mfma4(v[2], b2, acc0);
            acc1 = mfma4(v[3], b3, acc1);
        }
        const f32x4 r = acc0 + acc1;
        // D[filter 16w + 4g + e][frame f]
        const unsigned fo = f0 + f;
        if (fo < total) {
            float* out = p.mel + (size_t)fo * p.n_mel;
            const int c0 = 16 * w + 4 * g;
            if ((p.n_mel & 3) == 0 && (reinterpret_cast<uintptr_t>(p.mel) & 15) == 0) {
                if (c0 < p.n_mel) *reinterpret_cast<f32x4*>(out + c0) = r;
            } else {
#pragma unroll
                for (int e = 0; e < 4; ++e)
                    if (c0 + e < p.n_mel) out[c0 + e] = r[e];
            }
        }
    }
    KWS_FE_STAMP(7);
}

template <typename SampleT, bool GATE>
static hipError_t launch_fft400_tiles(const FrontendParams& p, unsigned grid, hipStream_t st) {
    switch (p.mel_tiles) {
        case 1: hipLaunchKernelGGL((mel_fft400_kernel<1, SampleT, GATE>), dim3(grid), dim3(256), 0, st, p); break;
        case 2: hipLaunchKernelGGL((mel_fft400_kernel<2, SampleT, GATE>), dim3(grid), dim3(256), 0, st, p); break;
        case 3: hipLaunchKernelGGL((mel_fft400_kernel<3, SampleT, GATE>), dim3(grid), dim3(256), 0, st, p); break;
        case 4: hipLaunchKernelGGL((mel_fft400_kernel<4, SampleT, GATE>), dim3(grid), dim3(256), 0, st, p); break;
        default: return hipErrorInvalidValue;
    }
    return hipGetLastError();
}

hipError_t launch_mel_fft400(const FrontendParams& p, int B, hipStream_t st) {
    const long long total = (long long)B * p.T;        // < 2^31 (checked by the caller)
    const unsigned grid = (unsigned)(((total + 15) / 16 + 7) / 8 * 8);     // multiple of 8: the kernel's XCD-aware block map
    FrontendParams q = p;
    q.fft_blocks = (int)grid;
    if (!p.gate) return p.pcm_i16 ? launch_fft400_tiles<int16_t, false>(q, grid, st) : launch_fft400_tiles<float, false>(q, grid, st);
    return p.pcm_i16 ? launch_fft400_tiles<int16_t, true>(q, grid, st) : launch_fft400_tiles<float, true>(q, grid, st);
}

}  // namespace kws

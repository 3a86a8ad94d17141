// Greedy CTC collapse, keyword decision and VAD on the GPU.
//   ctc_decode / ctc_decode2 / ctc_decode_strict : utils/prediction.py:18-62, :65-86, :89-108
//   ctc_predict                                   : utils/prediction.py:111-118
//   vad                                           : utils/basic_vad.py:17-18
// HBM-bound byte work: one thread walks one stream's [T,C] window (the lockout / loose-mode rules
// are sequential in t); a warp-wide layout would buy nothing at 24 B per frame.
#include "kws_internal.h"
#include "vad_device.h"

namespace kws {

struct RowTop { float best; int arg; };

// first maximum over columns lo..hi-1 of one softmax row
__device__ __forceinline__ RowTop row_top(const float* row, int lo, int hi) {
    RowTop r{row[lo], 0};
    for (int c = lo + 1; c < hi; ++c)
        if (row[c] > r.best) { r.best = row[c]; r.arg = c - lo; }
    return r;
}

__global__ void ctc_decode_kernel(int kind, const float* __restrict__ softmax, const int32_t* __restrict__ lengths,
                                  int B, int T, int C, int lockout, float thres, float loose_thres,
                                  int32_t* __restrict__ words, int32_t* __restrict__ counts, int max_words) {
    const int b = blockIdx.x * blockDim.x + threadIdx.x;
    if (b >= B) return;
    int len = lengths ? lengths[b] : T;
    len = len < 0 ? 0 : (len > T ? T : len);
    const float* sm = softmax + (size_t)b * T * C;
    int32_t* out = words + (size_t)b * max_words;
    int n = 0;
    auto emit = [&](int wd) { if (n < max_words) out[n] = wd; ++n; };

    if (kind == KWS_DECODE2) {               // word changes only, utils/prediction.py:74-80
        int prev = -1;
        for (int t = 0; t < len; ++t) {
            const RowTop r = row_top(sm + (size_t)t * C, 1, C - 1);
            const int wd = r.best > thres ? r.arg : -1;
            if (wd >= 0 && wd != prev) emit(wd + 1);
            prev = wd;
        }
    } else if (kind == KWS_DECODE_STRICT) {  // threshold + lockout, :97-103
        int skip_until = 0;
        for (int t = 0; t < len; ++t) {
            if (t < skip_until) continue;
            const RowTop r = row_top(sm + (size_t)t * C, 1, C - 1);
            if (r.best > thres) { emit(r.arg + 1); skip_until = t + lockout; }
        }
    } else {                                 // ctc_decode with loose mode, :28-56 (columns 1:5)
        int skip_until = 0, last_t = 0;
        int h0 = 0, h1 = 0, h2 = 0;          // last three emitted words, h2 newest
        bool loose = false;
        for (int t = 0; t < len; ++t) {
            if (t < skip_until) continue;
            const float* row = sm + (size_t)t * C;
            const RowTop r = row_top(row, 1, 5);
            if (!loose) {
                if (r.best > thres) {
                    emit(r.arg + 1);
                    h0 = h1; h1 = h2; h2 = r.arg + 1; last_t = t;
                    skip_until = t + lockout;
                    loose = (h0 == 1 && h1 == 2 && h2 == 3);
                }
                continue;
            }
            if (r.best < loose_thres) {
                if (h2 != 3) { skip_until = t + lockout; loose = false; }
            } else if (row[3] > loose_thres) {       // le4 = column 2 of the 1:5 slice
                emit(3);
                h0 = h1; h1 = h2; h2 = 3; last_t = t;
                skip_until = t + lockout;
                loose = false;
            } else if (r.best > 0.6f && last_t + lockout < t) {
                emit(r.arg + 1);
                h0 = h1; h1 = h2; h2 = r.arg + 1; last_t = t;
            }
        }
    }
    counts[b] = n;
}

struct LabelDigits { int32_t d[16]; int n; };

__global__ void ctc_predict_kernel(const int32_t* __restrict__ words, const int32_t* __restrict__ counts, int B,
                                   int max_words, LabelDigits lab, int32_t* __restrict__ hit) {
    const int b = blockIdx.x * blockDim.x + threadIdx.x;
    if (b >= B) return;
    int n = counts[b];
    n = n > max_words ? max_words : n;
    const int32_t* w = words + (size_t)b * max_words;
    int found = lab.n == 0 ? 1 : 0;
    for (int i = 0; i + lab.n <= n && !found; ++i) {
        bool ok = true;
        for (int j = 0; j < lab.n; ++j) ok = ok && (w[i + j] == lab.d[j]);
        found = ok ? 1 : 0;
    }
    hit[b] = found;
}

__global__ void __launch_bounds__(256) vad_kernel(const float* __restrict__ pcm, int N, float thres,
                                                  uint8_t* __restrict__ speech, float* __restrict__ abs_sum) {
    const int b = blockIdx.x;
    const float total = block_abs_sum<float>(pcm + (size_t)b * N, N, nullptr);
    if (threadIdx.x == 0) {
        speech[b] = total > thres ? 1 : 0;
        if (abs_sum) abs_sum[b] = total;
    }
}

// The head of one loop iteration of detector.py:158-177 for every stream, in one pass over the new chunk: the samples
// as the ring buffer hands them over (int16 -> float by 2^-15, detector.py:40-43,74-79; float PCM is read as it is),
// vad(data, thres) (utils/basic_vad.py:17-18; block_abs_sum, the summation kws_vad uses: identical decisions), and
// the masks the rest of the iteration consumes: silent (-> clear the decode window before the chunk is added) and
// reset = silent | restart (-> the GRU starts this chunk from the zero state: clean_state()).
template <typename SampleT>
__global__ void __launch_bounds__(256) vad_gate_kernel(const SampleT* __restrict__ pcm, int N, float thres, float* __restrict__ pcm_f32,
                                                       const uint8_t* __restrict__ restart, uint8_t* __restrict__ silent,
                                                       uint8_t* __restrict__ reset, const float* __restrict__ carry, int n_carry,
                                                       float* __restrict__ next, int n_next) {
    const int b = blockIdx.x;
    const SampleT* row = pcm + (size_t)b * N;
    const float total = block_abs_sum<SampleT>(row, N, sizeof(SampleT) == 2 ? pcm_f32 + (size_t)b * N : nullptr);
    if (threadIdx.x == 0) vad_masks(total, thres, b, restart, silent, reset);
    // next carry, while the chunk is hot in the cache
    carry_tail<SampleT>(carry + (size_t)b * n_carry, n_carry, row, N, next + (size_t)b * n_next, n_next);
}

hipError_t launch_vad_gate(const void* pcm, int pcm_int16, int B, int N, float thres, float* pcm_f32, const uint8_t* restart,
                           uint8_t* silent, uint8_t* reset, const float* carry, int n_carry, float* next, int n_next, hipStream_t st) {
    if (pcm_int16)
        hipLaunchKernelGGL(vad_gate_kernel<int16_t>, dim3(B), dim3(256), 0, st, static_cast<const int16_t*>(pcm), N, thres, pcm_f32,
                           restart, silent, reset, carry, n_carry, next, n_next);
    else
        hipLaunchKernelGGL(vad_gate_kernel<float>, dim3(B), dim3(256), 0, st, static_cast<const float*>(pcm), N, thres, pcm_f32,
                           restart, silent, reset, carry, n_carry, next, n_next);
    return hipGetLastError();
}

// state_in and state_out may be the same buffer (kws_stream_feed works in place): no __restrict__ on them
__global__ void state_passthrough_kernel(const float* state_in, float* state_out, const uint8_t* __restrict__ reset,
                                         int32_t* __restrict__ prev_word, int L, int B, int H) {
    const size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= (size_t)L * B * H) return;
    const int b = (int)((i / H) % B);
    state_out[i] = reset[b] ? 0.f : state_in[i];
    // a reset stream starts its ctc_decode2 neighbour rule from pre_word = -1, exactly as the T > 0 kernels do
    if (prev_word && reset[b] && i < (size_t)B * H && i % H == 0) prev_word[b] = -1;
}
hipError_t launch_state_passthrough(const float* state_in, float* state_out, const uint8_t* reset, int32_t* prev_word, int L, int B, int H,
                                    hipStream_t st) {
    const size_t n = (size_t)L * B * H;
    hipLaunchKernelGGL(state_passthrough_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, st, state_in, state_out, reset, prev_word,
                       L, B, H);
    return hipGetLastError();
}

hipError_t launch_ctc_decode(int kind, const float* softmax, const int32_t* lengths, int B, int T, int C,
                             int lockout, float thres, float loose_thres, int32_t* words, int32_t* counts,
                             int max_words, hipStream_t st) {
    hipLaunchKernelGGL(ctc_decode_kernel, dim3((B + 63) / 64), dim3(64), 0, st, kind, softmax, lengths, B, T, C,
                       lockout, thres, loose_thres, words, counts, max_words);
    return hipGetLastError();
}

hipError_t launch_ctc_predict(const int32_t* words, const int32_t* counts, int B, int max_words,
                              const int32_t* label_digits, int label_len, int32_t* hit, hipStream_t st) {
    LabelDigits lab;
    for (int i = 0; i < 16; ++i) lab.d[i] = i < label_len ? label_digits[i] : 0;
    lab.n = label_len;
    hipLaunchKernelGGL(ctc_predict_kernel, dim3((B + 63) / 64), dim3(64), 0, st, words, counts, B, max_words, lab, hit);
    return hipGetLastError();
}

hipError_t launch_vad(const float* pcm, int B, int N, float thres, uint8_t* speech, float* abs_sum, hipStream_t st) {
    hipLaunchKernelGGL(vad_kernel, dim3(B), dim3(256), 0, st, pcm, N, thres, speech, abs_sum);
    return hipGetLastError();
}

}  // namespace kws

// Device-side stream manager: the per-stream bookkeeping of detector.py:168-209 for B streams at once.
//   * SimpleQueue(15) of softmax chunks (utils/queue.py:16-38)        -> ring [B][NQ][TMAX][C] + per-slot lengths
//   * clear on silence before the chunk is added (detector.py:171-177) -> clear_before[b]
//   * concatenate the window, ctc_decode2, ctc_predict(label) (detector.py:197-201) -> windowed re-scan,
//     exactly the reference's O(window) decode (not an incremental approximation)
//   * on trigger: clear the window and request a state reset (detector.py:202-208) -> restart[b] = 1
// One thread per stream: the decode is sequential in t and the data is ~8 KB per stream.
#include "kws_internal.h"

namespace kws {

__global__ void window_step_kernel(const WindowParams p) {
    const int b = blockIdx.x * blockDim.x + threadIdx.x;
    if (b >= p.B) return;
    const int NQ = p.nq, TM = p.tmax, C = p.C;
    int head = p.head[b], count = p.count[b];
    if (p.clear_before && p.clear_before[b]) { head = 0; count = 0; }
    // add(): drop the oldest chunk when full (utils/queue.py:26-32)
    int slot;
    if (count < NQ) { slot = (head + count) % NQ; ++count; }
    else { slot = head; head = (head + 1) % NQ; }
    float* dst = p.ring + ((size_t)b * NQ + slot) * TM * C;
    const float* src = p.softmax + (size_t)b * p.T * C;
    for (int i = 0; i < p.T * C; ++i) dst[i] = src[i];
    p.lens[b * NQ + slot] = p.T;
    // ctc_decode2 over the concatenated window + substring match of the label digits (KMP-free: the
    // automaton state is "how many label digits are matched at the tail", recomputed by fallback)
    int prev = -1, hit = 0;
    int hist[16];                    // last label_len emitted words (ring)
    int nh = 0;
    for (int q = 0; q < count && !hit; ++q) {
        const int sl = (head + q) % NQ;
        const float* chunk = p.ring + ((size_t)b * NQ + sl) * TM * C;
        const int len = p.lens[b * NQ + sl];
        for (int t = 0; t < len && !hit; ++t) {
            const float* row = chunk + t * C;
            float best = row[1];
            int arg = 0;
            for (int c = 2; c < C - 1; ++c)
                if (row[c] > best) { best = row[c]; arg = c - 1; }
            const int wd = best > p.thres ? arg : -1;
            if (wd >= 0 && wd != prev) {
                hist[nh & 15] = wd + 1;
                ++nh;
                if (nh >= p.label_len) {
                    bool ok = true;
                    for (int j = 0; j < p.label_len; ++j) ok = ok && hist[(nh - p.label_len + j) & 15] == p.label[j];
                    hit = ok ? 1 : 0;
                }
            }
            prev = wd;
        }
    }
    if (p.label_len == 0) hit = 1;
    if (hit) { head = 0; count = 0; }
    p.head[b] = head;
    p.count[b] = count;
    p.hit[b] = hit;
    if (p.restart) p.restart[b] = hit ? 1 : 0;
}

__global__ void window_reset_kernel(int B, int* head, int* count) {
    const int b = blockIdx.x * blockDim.x + threadIdx.x;
    if (b < B) { head[b] = 0; count[b] = 0; }
}

hipError_t launch_window_step(const WindowParams& p, hipStream_t st) {
    hipLaunchKernelGGL(window_step_kernel, dim3((p.B + 63) / 64), dim3(64), 0, st, p);
    return hipGetLastError();
}
hipError_t launch_window_reset(int B, int* head, int* count, hipStream_t st) {
    hipLaunchKernelGGL(window_reset_kernel, dim3((B + 63) / 64), dim3(64), 0, st, B, head, count);
    return hipGetLastError();
}

}  // namespace kws

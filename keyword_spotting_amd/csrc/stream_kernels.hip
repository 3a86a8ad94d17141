// Device-side stream manager: the per-stream bookkeeping of detector.py:168-209 for B streams at once.
//   * SimpleQueue(15) of softmax chunks (utils/queue.py:16-38)        -> ring [B][NQ][TMAX] of per-frame WORDS + lengths
//     (decode2's frame rule is a function of the frame alone; the threshold is fixed per window handle)
//   * clear on silence before the chunk is added (detector.py:171-177) -> clear_before[b]
//   * concatenate the window, ctc_decode2, ctc_predict(label) (detector.py:197-201) -> windowed re-scan,
//     exactly the reference's O(window) decode (not an incremental approximation)
//   * on trigger: clear the window and request a state reset (detector.py:202-208) -> restart[b] = 1
// One wave per stream (kws_window_step_kernel below).
#include "kws_internal.h"

namespace kws {

// One wave per stream: lanes = frames for the per-frame rule and for moving the ring through LDS with every load in
// flight; the FIFO walk + label match is inherently sequential but short (<= nq*tmax bytes) and runs on lane 0.
__global__ void __launch_bounds__(64) window_step_kernel(const WindowParams p) {
    const int b = blockIdx.x, lane = threadIdx.x;
    const int NQ = p.nq, TM = p.tmax, C = p.C;
    extern __shared__ __attribute__((aligned(16))) unsigned char ring[];       // [NQ][TM] words of this stream
    int head = p.head[b], count = p.count[b];
    if (p.clear_before && p.clear_before[b]) { head = 0; count = 0; }
    // add(): drop the oldest chunk when full (utils/queue.py:26-32)
    int slot;
    if (count < NQ) { slot = (head + count) % NQ; ++count; }
    else { slot = head; head = (head + 1) % NQ; }
    // ctc_decode2's frame rule (utils/prediction.py:67,74-75) depends on the frame alone, so the ring caches each
    // frame's word (-1 = below threshold) instead of its softmax row.  TM is a multiple of 16.
    int8_t* gring = p.words + (size_t)b * NQ * TM;
    for (int t = lane; t < p.T; t += 64) {
        const float* row = p.softmax + ((size_t)b * p.T + t) * C;
        float best = row[1];
        int arg = 0;
        for (int c = 2; c < C - 1; ++c)
            if (row[c] > best) { best = row[c]; arg = c - 1; }
        const int8_t wd = (int8_t)(best > p.thres ? arg : -1);
        gring[slot * TM + t] = wd;
    }
    // ring -> LDS, 16 bytes per lane per trip (the new chunk's bytes come from the stores above: same wave, so
    // wait for them and read back through the cache)
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");
    __syncthreads();
    const int nvec = NQ * TM / 16;
    for (int i = lane; i < nvec; i += 64)
        reinterpret_cast<uint4*>(ring)[i] = reinterpret_cast<const uint4*>(gring)[i];
    __syncthreads();
    if (lane == 0) {
        p.lens[b * NQ + slot] = p.T;
        // concatenate the window in FIFO order, emit on word changes, match the label at the tail of what was
        // emitted: the emitted words (1..9) are shifted into a 64-bit register, 4 bits each
        unsigned long long want = 0ull, mask = 0ull;
        for (int j = 0; j < p.label_len; ++j) { want = (want << 4) | (unsigned)p.label[j]; mask = (mask << 4) | 0xfull; }
        unsigned long long hist = 0ull;
        int prev = -1, hit = 0, nh = 0;
        for (int q = 0; q < count && !hit; ++q) {
            const int sl = (head + q) % NQ;
            const int len = p.lens[b * NQ + sl];
            const unsigned char* chunk = ring + sl * TM;
            for (int t = 0; t < len && !hit; ++t) {
                const int wd = (int)(signed char)chunk[t];
                if (wd >= 0 && wd != prev) {
                    hist = (hist << 4) | (unsigned)(wd + 1);
                    ++nh;
                    hit = (nh >= p.label_len && ((hist ^ want) & mask) == 0ull) ? 1 : 0;
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
}

__global__ void window_reset_kernel(int B, int* head, int* count) {
    const int b = blockIdx.x * blockDim.x + threadIdx.x;
    if (b < B) { head[b] = 0; count[b] = 0; }
}

hipError_t launch_window_step(const WindowParams& p, hipStream_t st) {
    hipLaunchKernelGGL(window_step_kernel, dim3(p.B), dim3(64), (size_t)p.nq * p.tmax, st, p);
    return hipGetLastError();
}
hipError_t launch_window_reset(int B, int* head, int* count, hipStream_t st) {
    hipLaunchKernelGGL(window_reset_kernel, dim3((B + 63) / 64), dim3(64), 0, st, B, head, count);
    return hipGetLastError();
}

}  // namespace kws

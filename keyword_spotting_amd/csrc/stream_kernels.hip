// Device-side stream manager: the per-stream bookkeeping of detector.py:168-209 for B streams at once.
//   * SimpleQueue(15) of softmax chunks (utils/queue.py:16-38)        -> ring [B][NQ][TMAX] of per-frame WORDS + lengths
//     (decode2's frame rule is a function of the frame alone; the threshold is fixed per window handle)
//   * clear on silence before the chunk is added (detector.py:171-177) -> clear_before[b]
//   * concatenate the window, ctc_decode2, ctc_predict(label) (detector.py:197-201) -> windowed re-scan,
//     exactly the reference's O(window) decode (not an incremental approximation)
//   * on trigger: clear the window and request a state reset (detector.py:202-208) -> restart[b] = 1
// One wave per stream (window_step_kernel below).
#include "kws_internal.h"
#include "window_device.h"

namespace kws {

// One wave per stream, all 64 lanes busy: the per-frame rule, the ring's trip through LDS, the walk over the window in
// FIFO order with the change-point compaction (ballot + popcount keeps the order) and the label match (every start
// position in parallel).  ctc_predict only asks whether the label occurs in the emitted words, so "first hit wins" in the
// reference's loop and "any hit" here are the same decision.  (A lane-0 replay of the window took 61 us for 4096
// streams: ~400 cycles per frame of single-lane, latency-bound code.)
__global__ void __launch_bounds__(64) window_step_kernel(const WindowParams p) {
    const int b = blockIdx.x, lane = threadIdx.x;
    const int NQ = p.nq, TM = p.tmax, C = p.C;
    extern __shared__ __attribute__((aligned(16))) unsigned char lds[];
    unsigned char* ring = lds;                    // [NQ][TM] words of this stream
    unsigned char* emit = ring + NQ * TM;         // emitted words (1-based), compacted in order
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
        gring[slot * TM + t] = (int8_t)(best > p.thres ? arg : -1);
    }
    // chunk lengths in FIFO order: lane q holds chunk q (NQ <= 64, checked by kws_window_create)
    int len = 0;
    if (lane < count) {
        const int sl = (head + lane) % NQ;
        len = sl == slot ? p.T : p.lens[b * NQ + sl];
    }
    // ring -> LDS, 16 bytes per lane per trip (the new chunk's bytes come from the stores above: same wave, so
    // wait for them and read back through the cache)
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");
    __syncthreads();
    const int nvec = NQ * TM / 16;
    for (int i = lane; i < nvec; i += 64)
        reinterpret_cast<uint4*>(ring)[i] = reinterpret_cast<const uint4*>(gring)[i];
    __syncthreads();
    // The window in FIFO order (detector.py:197) is walked over the PADDED grid j = q TM + t, 64 cells per trip, without copying it
    // together first: a cell is live if t < len_q; a live cell emits its word on a change against the live cell before it
    // (utils/prediction.py:76-80) -- the nearest live lane below (ballot + bit scan + one cross-lane read), or the last live
    // cell of the trips before (a wave-uniform carry).  Emitted words are compacted in order (ballot + popcount).  (A loop over
    // the queued chunks that first concatenated them in LDS took 0.5 us per chunk: 5 us of this kernel's 10 with a full window.)
    int n_emit = 0;
    int carry_word = -1;
    const float inv_tm = 1.0f / (float)TM;
    const int cells = count * TM;
    for (int base = 0; base < cells; base += 64) {
        const int j = base + lane;
        const int q = (int)(((float)j + 0.5f) * inv_tm), t = j - q * TM;        // exact: j < 64 * 4096
        const int lq = __shfl(len, q < count ? q : 0);
        const bool livec = j < cells && t < lq;
        int sl = head + q;
        sl -= sl >= NQ ? NQ : 0;
        const int wd = livec ? (int)(signed char)ring[sl * TM + t] : -1;
        const unsigned long long lv = __builtin_amdgcn_ballot_w64(livec);
        const unsigned long long below = lv & ((1ull << lane) - 1ull);
        const int src = below ? 63 - __builtin_clzll(below) : lane;
        const int pv = __shfl(wd, src);
        const int pw = below ? pv : carry_word;
        const bool flag = livec && wd >= 0 && wd != pw;
        const unsigned long long m = __builtin_amdgcn_ballot_w64(flag);
        if (flag) emit[n_emit + __builtin_popcountll(m & ((1ull << lane) - 1ull))] = (unsigned char)(wd + 1);
        n_emit += __builtin_popcountll(m);
        if (lv) carry_word = __shfl(wd, 63 - __builtin_clzll(lv));
    }
    __syncthreads();
    // ctc_predict (utils/prediction.py:111-118): is the label a substring of the emitted words?
    bool found = false;
    const int L = p.label_len;
    for (int base = 0; base + L <= n_emit && !found; base += 64) {
        const int i = base + lane;
        bool ok = i + L <= n_emit;
        for (int j = 0; j < L && ok; ++j) ok = (int)emit[i + j] == p.label[j];
        found = __builtin_amdgcn_ballot_w64(ok) != 0ull;
    }
    const int hit = (found || L == 0) ? 1 : 0;
    if (lane == 0) {
        p.lens[b * NQ + slot] = p.T;
        if (hit) { head = 0; count = 0; }        // detector.py:202-208
        p.head[b] = head;
        p.count[b] = count;
        p.hit[b] = hit;
        if (p.restart) p.restart[b] = hit ? 1 : 0;
    }
}

// The incremental window as a launch of its own (kws_window_step_incremental; kws_stream_feed where the last GRU layer's
// kernel has no window tail: generic / pipelined / int8 kernels, chunks of more than 64 frames, zero-frame chunks).  One
// workgroup = 16 streams, as the tail inside the GRU kernels: the frame rule of ctc_decode2 (utils/prediction.py:67,74-75)
// over the chunk's softmax rows, then window_tail.
__global__ void __launch_bounds__(256) window_inc_kernel(const WindowIncParams p) {
    extern __shared__ __attribute__((aligned(16))) char wlds[];
    const int stride = (p.T + 15) & ~15;
    int8_t* cw = reinterpret_cast<int8_t*>(wlds);                              // [16 streams][stride]
    uint8_t* dl = reinterpret_cast<uint8_t*>(wlds) + (size_t)16 * (stride > 0 ? stride : 16);
    char* scratch = reinterpret_cast<char*>(dl) + 256;
    const int tid = threadIdx.x, b0 = blockIdx.x * 16, C = p.C;
    dl[tid] = p.delta[tid];
    WindowTailRegs<4> req;
    window_tail_request<4>(p.win, p.B, b0, tid, req);       // in flight behind the frame rule below
    const int s = tid & 15, b = min(b0 + s, p.B - 1);
    for (int t = tid >> 4; t < p.T; t += 16) {
        const float* row = p.softmax + ((size_t)b * p.T + t) * C;
        float best = row[1];
        int arg = 0;
        for (int c = 2; c < C - 1; ++c)
            if (row[c] > best) { best = row[c]; arg = c - 1; }
        cw[s * stride + t] = (int8_t)(best > p.thres ? arg : -1);
    }
    __syncthreads();
    window_tail<4>(p.win, p.B, b0, p.T, cw, stride, dl, scratch, tid, req);
}

__global__ void window_reset_kernel(int B, int* head, int* count) {
    const int b = blockIdx.x * blockDim.x + threadIdx.x;
    if (b < B) { head[b] = 0; count[b] = 0; }
}

hipError_t launch_window_step(const WindowParams& p, hipStream_t st) {
    hipLaunchKernelGGL(window_step_kernel, dim3(p.B), dim3(64), (size_t)2 * p.nq * p.tmax, st, p);
    return hipGetLastError();
}
hipError_t launch_window_inc(const WindowIncParams& p, hipStream_t st) {
    const int stride = (p.T + 15) & ~15;
    const size_t lds = (size_t)16 * (stride > 0 ? stride : 16) + 256 + window_tail_scratch_bytes(p.win.nq);
    static LdsGrant granted;
    if (lds > 48 * 1024) {
        const hipError_t e = grant_dynamic_lds(window_inc_kernel, granted, lds);
        if (e != hipSuccess) return e;
    }
    hipLaunchKernelGGL(window_inc_kernel, dim3((p.B + 15) / 16), dim3(256), lds, st, p);
    return hipGetLastError();
}
hipError_t launch_window_reset(int B, int* head, int* count, hipStream_t st) {
    hipLaunchKernelGGL(window_reset_kernel, dim3((B + 63) / 64), dim3(64), 0, st, B, head, count);
    return hipGetLastError();
}

}  // namespace kws

// The decode window of the streaming loop (detector.py:195-209: prob_queue.add, concatenate, ctc_decode2, ctc_predict,
// clear on a hit; :171-177: clear on silence) in INCREMENTAL form: per queued chunk a summary instead of its frames, the
// window evaluated over the <= maxLen summaries -- O(chunks) per step, not O(frames in the window) -- and therefore cheap
// enough to ride at the end of the last GRU layer's launch.  Equivalence with the reference's re-scan, evictions included, is
// property-tested on the CPU (tests/test_window_incremental.py, the same algorithm in tests/window_model.py) and on the
// device against the re-scanning window_step_kernel and reference-generated traces (tests/test_gpu_window.py).
//
// Summary of a chunk of n frames with ctc_decode2 frame words w_0..w_{n-1} (utils/prediction.py:67,74-75; -1 = none):
//   meta   n > 0, first = w_0, last = w_{n-1}
//   tab[q] label-matcher state after the chunk's frames 1..n-1 when entered in state q.  Frame t >= 1 emits w_t iff
//          w_t >= 0 and w_t != w_{t-1} (:76-80) -- a function of the chunk alone.  The matcher is the KMP automaton of the
//          label over emitted words (ctc_predict asks whether the label occurs, :111-118): state = digits matched,
//          n_label = found (absorbing).
// Whether frame 0 emits depends on the chunk BEFORE it in the window (w_0 != that chunk's last word, or no such chunk) -- which
// is what an eviction changes -- so it is decided at evaluation time:
//   state = 0, prev = -1;  for each queued chunk, oldest first:  skip if empty;
//       if first >= 0 and first != prev: state = delta[state][first + 1];     state = tab[state];     prev = last
//   hit iff state reached n_label anywhere.
#pragma once
#include "kws_internal.h"

namespace kws {

// Called by EVERY thread of the workgroup (blockDim.x >= 256; threads >= 256 only keep the barriers).  Streams [b0, b0 + 16)
// of B; cw: LDS, [T][16] frame words of this chunk (complete and visible: the caller has synchronised); scratch: LDS,
// window_tail_scratch_bytes(W.nq) bytes the caller no longer needs.  Ends with every global result written; the caller
// synchronises before it reuses `scratch` or `cw`.
__device__ __forceinline__ void window_tail(const WindowTail& W, int B, int b0, int T, const int8_t* cw, char* scratch, int tid) {
    const int nq = W.nq, nl = W.n_label;
    uint8_t* dl = reinterpret_cast<uint8_t*>(scratch);               // [16][16] delta
    uint8_t* nt = dl + 256;                                          // [16 streams][16] this chunk's tab
    uint4* rtab = reinterpret_cast<uint4*>(nt + 256);                // [16 streams][nq] queued tabs
    uint32_t* rmeta = reinterpret_cast<uint32_t*>(rtab + 16 * nq);   // [16 streams][nq]
    const bool worker = tid < 256;
    // requests first: the label matcher, the 16 streams' rings, head / count / clear of "my" stream -- they are in flight
    // while the chunk's own table is built
    int head = 0, count = 0;
    bool clear = false;
    const int es = tid;                                              // evaluator threads: tid < 16, stream b0 + tid
    const bool evaluator = tid < 16 && b0 + es < B;
    if (evaluator) {
        head = W.head[b0 + es];
        count = W.count[b0 + es];
        clear = W.clear_before != nullptr && W.clear_before[b0 + es] != 0;
    }
    if (worker) {
        dl[tid] = W.delta[tid];
        for (int i = tid; i < 16 * nq; i += 256) {
            const int ss = i / nq, k = i - ss * nq;
            const int bb = min(b0 + ss, B - 1);
            rtab[i] = reinterpret_cast<const uint4*>(W.tab)[(size_t)bb * nq + k];
            rmeta[i] = W.meta[(size_t)bb * nq + k];
        }
    }
    __syncthreads();
    if (worker) {
        // thread (stream s, entry state q): walk the chunk's frames 1..T-1
        const int s = tid & 15, q = tid >> 4;
        int state = q;
        if (q < nl && T > 1) {
            int pre = cw[s];
            for (int t = 1; t < T; ++t) {
                const int w = cw[t * 16 + s];
                if (w >= 0 && w != pre && state < nl) state = w + 1 < 16 ? dl[state * 16 + w + 1] : 0;
                pre = w;
            }
        }
        nt[s * 16 + q] = (uint8_t)state;
    }
    __syncthreads();
    if (evaluator) {
        const int b = b0 + es;
        if (clear) { head = 0; count = 0; }                          // detector.py:171-177
        int slot;                                                    // add(): drop the oldest when full (utils/queue.py:26-32)
        if (count < nq) { slot = head + count; slot -= slot >= nq ? nq : 0; ++count; }
        else { slot = head; head = head + 1 == nq ? 0 : head + 1; }
        const uint4 row = *reinterpret_cast<const uint4*>(nt + es * 16);
        const int first = T > 0 ? (int)cw[es] : -1, last = T > 0 ? (int)cw[(T - 1) * 16 + es] : -1;
        const uint32_t m_new = T > 0 ? (0x10000u | (uint32_t)(first + 1) | ((uint32_t)(last + 1) << 8)) : 0u;
        rtab[es * nq + slot] = row;
        rmeta[es * nq + slot] = m_new;
        reinterpret_cast<uint4*>(W.tab)[(size_t)b * nq + slot] = row;
        W.meta[(size_t)b * nq + slot] = m_new;
        bool hit = nl == 0;                                          // '' occurs in anything (utils/prediction.py:118)
        int state = 0, prev = -1;
        for (int k = 0; k < count && !hit; ++k) {
            int sl = head + k;
            sl -= sl >= nq ? nq : 0;
            const uint32_t m = rmeta[es * nq + sl];
            if (!(m & 0x10000u)) continue;                           // an empty chunk holds a slot and nothing else
            const int f = (int)(m & 255u) - 1, l = (int)((m >> 8) & 255u) - 1;
            if (f >= 0 && f != prev) {
                state = f + 1 < 16 ? dl[state * 16 + f + 1] : 0;
                if (state == nl) { hit = true; break; }
            }
            state = reinterpret_cast<const uint8_t*>(rtab + es * nq + sl)[state];
            if (state == nl) { hit = true; break; }
            prev = l;
        }
        if (hit) { head = 0; count = 0; }                            // detector.py:202-208
        W.head[b] = head;
        W.count[b] = count;
        W.hit[b] = hit ? 1 : 0;
        if (W.restart) W.restart[b] = hit ? 1 : 0;
    }
}

}  // namespace kws

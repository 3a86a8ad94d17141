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

// Every slot also carries ftab: the chunk's whole table with its first frame already decided against the last word of its
// predecessor -- the nearest earlier non-empty chunk when it was queued.  Evictions are FIFO, so that predecessor is either
// still queued (ftab is right) or gone with everything before it (the chunk is then the first non-empty one and goes through
// tab).  Evaluation is one dependent LDS byte read per queued chunk.  States >= n_label map to themselves in both tables, so
// "found" (= n_label <= 15) is absorbing and the walk needs no exit.
//
// Work split: 16 lanes per stream (lane q = entry state q = ring slots q, q + 16, ...), four streams per wave; everything a
// stream needs stays inside its wave, so the tail has NO workgroup barrier of its own.

__device__ __forceinline__ void wave_lds_sync() {        // LDS traffic of a wave completes in order: later reads see earlier writes of any lane
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    __builtin_amdgcn_wave_barrier();
}
__device__ __forceinline__ unsigned group16_max(unsigned v) {
#pragma unroll
    for (int m = 8; m >= 1; m >>= 1) { const unsigned o = (unsigned)__shfl_xor((int)v, m, 16); v = o > v ? o : v; }
    return v;
}
__device__ __forceinline__ unsigned group16_min(unsigned v) {
#pragma unroll
    for (int m = 8; m >= 1; m >>= 1) { const unsigned o = (unsigned)__shfl_xor((int)v, m, 16); v = o < v ? o : v; }
    return v;
}

// The tail's ten pointers and two sizes, read from the kernel-argument segment WHERE THE TAIL RUNS.  Taken from the by-value
// kernel parameter like every other field they would sit in scalar registers for the whole launch -- and the frame loops of
// the register-resident kernels have none to spare (gru_layer_f16x3<4, false, true>: 23 -> 58 spilled vector registers, some
// of them reloaded inside the frame loop, +5 us per 22-frame call).  `offset`: of the WindowTail inside the kernel's single
// parameter struct; the pointer is laundered so that the loads stay inside the group loop.
__device__ __forceinline__ WindowTail window_tail_params_from_kernarg(size_t offset) {
    typedef const __attribute__((address_space(4))) char* karg_ptr;
    karg_ptr base = (karg_ptr)__builtin_amdgcn_kernarg_segment_ptr();
    asm volatile("" : "+s"(base));
    static_assert(sizeof(WindowTail) % 8 == 0, "copied in 8-byte words");
    union { WindowTail w; unsigned long long q[sizeof(WindowTail) / 8]; } u;
    const __attribute__((address_space(4))) unsigned long long* src = (const __attribute__((address_space(4))) unsigned long long*)(base + offset);
#pragma unroll
    for (size_t i = 0; i < sizeof(WindowTail) / 8; ++i) u.q[i] = src[i];
    return u.w;
}

// The label matcher goes to LDS once per workgroup and launch (256 bytes, `dl`); a barrier of the caller's separates this from
// the first window_tail.
__device__ __forceinline__ void window_tail_prepare(const WindowTail& W, uint8_t* dl, int tid) {
    if (tid < 256) dl[tid] = W.delta[tid];
}

// What a lane asks of global memory for its stream's window, in flight between window_tail_request and window_tail.  (Issuing
// the request before the kernels' final flush, to hide its latency there, was measured and made the chunk SLOWER -- bf16
// 97-98 -> 100-101 us: twenty more live registers across the flush cost more than the round trip they hid.)
template <int R>
struct WindowTailRegs {
    int head, count;
    bool clear;
    uint4 rt[R][2];
    uint32_t rm[R];
};
template <int R>
__device__ __forceinline__ void window_tail_request(const WindowTail& W, int B, int b0, int tid, WindowTailRegs<R>& g) {
    if (tid >= 256) return;
    const int nq = W.nq, s = tid >> 4, q = tid & 15;
    const int b = b0 + s < B ? b0 + s : B - 1;                       // (lanes past the batch repeat its last stream and write nothing)
    g.head = W.head[b];
    g.count = W.count[b];
    g.clear = W.clear_before != nullptr && W.clear_before[b] != 0;
#pragma unroll
    for (int r = 0; r < R; ++r) {
        const int slot = q + 16 * r;
        g.rm[r] = 0u;
        if (slot < nq) {
            const uint4* src = reinterpret_cast<const uint4*>(W.tab) + ((size_t)b * nq + slot) * 2;
            g.rt[r][0] = src[0];
            g.rt[r][1] = src[1];
            g.rm[r] = W.meta[(size_t)b * nq + slot];
        }
    }
}

// Called by the first 256 threads of the workgroup (more may call: they return at once), after window_tail_request.  Streams [b0, b0 + 16) of B; cw: LDS,
// [16 streams][cw_stride] frame words of this chunk, stream-major (complete and visible: the caller has synchronised); dl:
// window_tail_prepare's table; scratch: LDS, window_tail_scratch_bytes(W.nq) bytes the caller no longer needs; R: ring slots
// per lane (16 R >= W.nq).  The caller synchronises before it reuses `scratch` or `cw`.
template <int R>
__device__ __forceinline__ void window_tail(const WindowTail& W, int B, int b0, int T, const int8_t* cw, int cw_stride, const uint8_t* dl,
                                            char* scratch, int tid, const WindowTailRegs<R>& g) {
    if (tid >= 256) return;
    const int nq = W.nq, nl = W.n_label;
    const int s = tid >> 4, q = tid & 15;
    const bool valid = b0 + s < B;
    const int b = valid ? b0 + s : B - 1;
    uint8_t* ring = reinterpret_cast<uint8_t*>(scratch) + (size_t)s * (nq * 32 + 32);   // [nq][tab 16 | ftab 16] of this stream
    uint8_t* ntab = ring + nq * 32;                                  // [16] this chunk's tab
    int head = g.head, count = g.count;
    const bool clear = g.clear;
    const uint4 (&rt)[R][2] = g.rt;
    const uint32_t (&rm)[R] = g.rm;
    // ---- lane (stream s, entry state q): the chunk's frames 1..T-1 -- only an emission (a changed word) touches the matcher
    const int8_t* row = cw + s * cw_stride;
    int state = q;
    if (q < nl && T > 1) {
        int pre = row[0];
        for (int t = 1; t < T; ++t) {
            const int w = row[t];
            if (w >= 0 && w != pre && state < nl) state = w + 1 < 16 ? dl[state * 16 + w + 1] : 0;
            pre = w;
        }
    }
    ntab[q] = (uint8_t)state;
    const int first = T > 0 ? (int)row[0] : -1, last = T > 0 ? (int)row[T - 1] : -1;
    // ---- the queued chunks land in LDS; the predecessor = the newest non-empty chunk queued so far (detector.py:171-177 first)
    if (clear) { head = 0; count = 0; }
    unsigned pack = 0u;
#pragma unroll
    for (int r = 0; r < R; ++r) {
        const int slot = q + 16 * r;
        if (slot < nq) {
            *reinterpret_cast<uint4*>(ring + slot * 32) = rt[r][0];
            *reinterpret_cast<uint4*>(ring + slot * 32 + 16) = rt[r][1];
            int pos = slot - head;
            pos += pos < 0 ? nq : 0;
            if (pos < count && (rm[r] & 0x10000u)) {
                const unsigned p = ((unsigned)(pos + 1) << 8) | ((rm[r] >> 8) & 255u);
                pack = p > pack ? p : pack;
            }
        }
    }
    pack = group16_max(pack);
    const int pred_last = pack ? (int)(pack & 255u) - 1 : -1;
    int slot_new;                                                    // add(): drop the oldest when full (utils/queue.py:26-32)
    if (count < nq) { slot_new = head + count; slot_new -= slot_new >= nq ? nq : 0; ++count; }
    else { slot_new = head; head = head + 1 == nq ? 0 : head + 1; }
    wave_lds_sync();                                                 // ntab and the ring copy are in LDS
    int fq = q;
    if (q < nl && first >= 0 && first != pred_last) fq = first + 1 < 16 ? dl[q * 16 + first + 1] : 0;
    const uint8_t tab_q = (uint8_t)state, ftab_q = ntab[fq];         // (T == 0: state == q and fq == q: identity rows)
    const uint32_t m_new = T > 0 ? (0x10000u | (uint32_t)(first + 1) | ((uint32_t)(last + 1) << 8)) : 0u;
    ring[slot_new * 32 + q] = tab_q;
    ring[slot_new * 32 + 16 + q] = ftab_q;
    wave_lds_sync();
    if (valid && q < 2)
        reinterpret_cast<uint4*>(W.tab)[((size_t)b * nq + slot_new) * 2 + q] = *reinterpret_cast<const uint4*>(ring + slot_new * 32 + 16 * q);
    if (valid && q == 0) W.meta[(size_t)b * nq + slot_new] = m_new;
    // ---- the oldest non-empty chunk of the window as it is now: it alone is evaluated through tab
    unsigned oldest = 0xffffffffu;
#pragma unroll
    for (int r = 0; r < R; ++r) {
        const int slot = q + 16 * r;
        if (slot < nq) {
            const uint32_t m = slot == slot_new ? m_new : rm[r];
            int pos = slot - head;
            pos += pos < 0 ? nq : 0;
            if (pos < count && (m & 0x10000u)) {
                const unsigned p = ((unsigned)pos << 8) | (m & 255u);
                oldest = p < oldest ? p : oldest;
            }
        }
    }
    oldest = group16_min(oldest);
    bool hit = nl == 0;                                              // '' occurs in anything (utils/prediction.py:118)
    if (oldest != 0xffffffffu && nl > 0) {
        const int k0 = (int)(oldest >> 8), f0 = (int)(oldest & 255u) - 1;
        int sl = head + k0;
        sl -= sl >= nq ? nq : 0;
        int st = f0 >= 0 ? (f0 + 1 < 16 ? (int)dl[f0 + 1] : 0) : 0;   // state 0 before it: nothing has been emitted
        st = ring[sl * 32 + st];
        for (int k = k0 + 1; k < count; ++k) {                       // every lane of the stream walks the same chain: broadcast reads
            sl = sl + 1 == nq ? 0 : sl + 1;
            st = ring[sl * 32 + 16 + st];
        }
        hit = st == nl;
    }
    if (valid && q == 0) {
        W.head[b] = hit ? 0 : head;                                  // detector.py:202-208
        W.count[b] = hit ? 0 : count;
        W.hit[b] = hit ? 1 : 0;
        if (W.restart) W.restart[b] = hit ? 1 : 0;
    }
}

}  // namespace kws

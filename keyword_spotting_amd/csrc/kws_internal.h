// Internal declarations shared by the HIP translation units of libkws_amd.so (not part of the ABI).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

#include <atomic>

#include "kws_amd.h"

// Experiment / test switches compiled into the kernel sources (timing counters, ablations, the self-test's negative control).
// A product build defines none of them; tools/build_variant.sh -- the only place that defines KWS_VARIANT_BUILD -- passes them
// to every translation unit, so kws_version() of such a library names them (KWS_VARIANT_TAG) and a stray -D in a product
// build does not compile.
#if defined(KWS_FAULT_INJECT) || defined(KWS_ABL_NOFLUSH) || defined(KWS_ABL_NOMEL) || defined(KWS_F16_TIMING) || defined(KWS_FE_TIMING) || \
    defined(KWS_FE_OCC) || defined(KWS_FE_FT) || defined(KWS_FE_SF) || defined(KWS_FE_NOSTAGE) || defined(KWS_FE_NODFT) || defined(KWS_FE_NOMEL) || \
    defined(KWS_OABL_NODIV) || defined(KWS_OABL_NODOT) || defined(KWS_OVERLAP_MIN_T) || defined(KWS_EXP_F16_WLO_ZERO)
#ifndef KWS_VARIANT_BUILD
#error "an experiment switch (KWS_FAULT_INJECT / KWS_ABL_* / KWS_*_TIMING / KWS_FE_* / KWS_OABL_* / KWS_OVERLAP_MIN_T) is defined in a product build: use tools/build_variant.sh"
#endif
#endif
#ifdef KWS_VARIANT_BUILD
#define KWS_VARIANT_TAG_1(name) "; " #name
#ifdef KWS_FAULT_INJECT
#define KWS_TAG_FAULT KWS_VARIANT_TAG_1(FAULT_INJECT)
#else
#define KWS_TAG_FAULT ""
#endif
#ifdef KWS_ABL_NOFLUSH
#define KWS_TAG_NOFLUSH KWS_VARIANT_TAG_1(ABL_NOFLUSH)
#else
#define KWS_TAG_NOFLUSH ""
#endif
#ifdef KWS_ABL_NOMEL
#define KWS_TAG_NOMEL KWS_VARIANT_TAG_1(ABL_NOMEL)
#else
#define KWS_TAG_NOMEL ""
#endif
#if defined(KWS_F16_TIMING) || defined(KWS_FE_TIMING)
#define KWS_TAG_TIMING KWS_VARIANT_TAG_1(TIMING)
#else
#define KWS_TAG_TIMING ""
#endif
#ifdef KWS_EXP_F16_WLO_ZERO
#define KWS_TAG_WLO KWS_VARIANT_TAG_1(EXP_F16_WLO_ZERO)
#else
#define KWS_TAG_WLO ""
#endif
#define KWS_VARIANT_TAG "; VARIANT BUILD" KWS_TAG_FAULT KWS_TAG_NOFLUSH KWS_TAG_NOMEL KWS_TAG_TIMING KWS_TAG_WLO
#else
#define KWS_VARIANT_TAG ""
#endif

namespace kws {

constexpr int kStreamsPerGroup = 16;  // MFMA N dimension: one workgroup advances 16 streams
constexpr int kMaxClasses = 8;

// hipFuncAttributeMaxDynamicSharedMemorySize is a per-device property of a kernel: one process may drive several GPUs
// (one handle each) from several host threads, so the "already granted" cache is per (kernel instantiation, device)
// and atomic.  Setting the attribute twice is harmless; skipping it on a second device makes the launch fail there.
constexpr int kMaxDevices = 64;
struct LdsGrant { std::atomic<size_t> bytes[kMaxDevices]; };
template <typename K>
inline hipError_t grant_dynamic_lds(K kernel, LdsGrant& cache, size_t lds) {
    int dev = 0;
    hipError_t e = hipGetDevice(&dev);
    if (e != hipSuccess) return e;
    const bool cached = dev >= 0 && dev < kMaxDevices;
    if (cached && cache.bytes[dev].load(std::memory_order_acquire) >= lds) return hipSuccess;
    e = hipFuncSetAttribute(reinterpret_cast<const void*>(kernel), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    if (e != hipSuccess) return e;
    if (cached) {
        size_t seen = cache.bytes[dev].load(std::memory_order_relaxed);
        while (seen < lds && !cache.bytes[dev].compare_exchange_weak(seen, lds, std::memory_order_release)) {}
    }
    return hipSuccess;
}

// One launch = one GRU layer over all T frames of the call, for every 16-stream group.
//
// Exchange layout ("xl"): a [H x 16 streams] activation block is stored as [H/16][64 lanes] float4,
// lane = 16*g + s, component r  <->  unit 16*n + 4*g + r of stream s.  This is at once the MFMA
// 16x16x4 C/D register image of tile n and, read back as float4, the B operand of the four
// k-chunks 4n..4n+3 -- so hidden state never needs a cross-lane transpose.
// The incremental decode window of a stream manager (window_device.h), run as the tail of the last layer's launch or by
// window_inc_kernel.  tab == nullptr: none.
struct WindowTail {
    uint8_t* tab;               // [B][nq][32]  per queued chunk: tab[16] matcher state after its frames 1..n-1 when entered in state q; ftab[16] the same
                                //              with its first frame decided against its predecessor's last word (window_device.h)
    uint32_t* meta;             // [B][nq]      bit 16: the chunk has frames; bits 0-7: first frame's word + 1; bits 8-15: last frame's word + 1
    int* head;                  // [B]
    int* count;                 // [B]
    const uint8_t* delta;       // [16 states][16 words] label matcher (KMP automaton), device memory
    const uint8_t* clear_before;// [B] or null
    int32_t* hit;               // [B]
    uint8_t* restart;           // [B] or null
    int nq, n_label;
};
constexpr int kWinTailMaxFrames = 64;   // chunk lengths the fused tail takes (its per-frame words wait in LDS): longer -> window_inc_kernel
constexpr int kWinTailMaxChunks = 24;   // window lengths the fused tail takes (the rings of 16 streams are staged in 16 KiB of LDS)
constexpr int kWinTailWordsStride = kWinTailMaxFrames + 4;                    // bytes per stream row of the words (68: rows on different banks)
constexpr size_t kWinTailWordsBytes = (size_t)16 * kWinTailWordsStride + 256;   // the call's words [16 streams][stride] + the label matcher [16][16]
__host__ __device__ inline size_t window_tail_scratch_bytes(int nq) { return (size_t)16 * (nq * 32 + 32); }

struct GruLayerParams {
    // weights (device, packed by pack.cpp)
    const float* wx;        // x-part fragments  [NT][3][KCX/4][64][4]; resident first layer: [NT][3][KCX][64]
    const float* wh;        // h-part fragments  [NT][3][H/16][64][4]
    const float* bias;      // [3][H]  (r, u, c)
    const float* wfc;       // LAST: [H/4][64] fragments of Wfc^T padded to 16 rows
    const float* bfc;       // LAST: [16] padded
    // activations
    const float* x_mel;     // FIRST: mel [B,T,I]
    const float4* x_prev;   // !FIRST: previous layer's output, xl layout [G][T][NT][64]
    float4* h_out;          // !LAST: this layer's output, xl layout
    const float* state_in;  // [B,H] of this layer
    float* state_out;       // [B,H]
    const int32_t* seq_len; // [B] or null
    const uint8_t* reset;   // [B] or null
    // LAST-layer epilogue
    float* logits;          // [B,T,C] or null
    float* softmax;         // [B,T,C] or null
    int8_t* tokens;         // [B,T] or null
    int32_t* prev_word;     // [B] or null
    float decode_thres;
    float value_clip;
    int use_relu;
    int B, T, I, C;
    int t_stride;           // frames per stream row of x_mel / logits / softmax / tokens (0: T) -- a call on a time block
    int t_base;             // of a longer sequence passes pre-offset pointers, the full row stride, and its first frame
    int KCX;                // x-part k-chunks (generic: multiple of 4)
    // layer-pipelined launch (generic kernel): frames published by the layer below / by this layer, per group
    const int* ready_in;
    int* ready_out;
    int* pipe_error;
    // LAST, kernels instantiated with the window tail only (kws_stream_feed): the decode-window step of detector.py:195-209
    // for the group's 16 streams, after their last frame (kept at the end: every other field keeps its offset)
    WindowTail win;
};
struct GruStackParams {
    GruLayerParams layer[8];
    int L, G;
    int xcd_affine;   // L divides 8: block i -> (layer = (i % 8) % L, group = (i / 8) * (8 / L) + (i % 8) / L)
};
hipError_t launch_gru_stack_generic_pipelined(const GruStackParams& sp, int hidden, hipStream_t st);

// bf16 fused stack (gru_bf16.hip): every layer in one launch, no inter-layer scratch
struct GruBf16Params {
    const uint4* w[2];      // per layer: [8 tiles][3 gates][KC_l chunks][64 lanes] bf16x8 A operands (x chunks first)
    const float* bias[2];   // per layer [3][128] fp32
    const uint4* wfc;       // [4 chunks][64]  Wfc^T padded to 16 rows
    const float* bfc;       // [16]
    const float* x_mel;     // [B,T,I]
    const float* state_in;  // [L,B,128]
    float* state_out;
    const int32_t* seq_len;
    const uint8_t* reset;
    GruLayerParams epi;     // logits / softmax / tokens / prev_word / thresholds for the epilogue
    int B, T, I, L;
};
bool gru_bf16_supported(int hidden, int n_mel, int layers);
hipError_t launch_gru_stack_bf16(const GruBf16Params& p, int kx0, int nl, hipStream_t st);    // p.epi.win.tab != null: with the window tail
bool gru_stack_bf16_takes_window(int kx0, int nl);          // the launch above has a window-tail instantiation for this shape
const char* gru_stack_bf16_kernel_name(int kx0, int nl);   // the kernel launch_gru_stack_bf16 picks (KWS_BF16_WAVES aware)
bool gru_bf16_vgpr_form();                                  // built with -mllvm -amdgpu-mfma-vgpr-form=1 (csrc/Makefile)

// "f16x3": fp32-accuracy stack on the fp16 matrix pipe, operands split hi + 2^-11 lo (gru_f16x3.hip); one layer per launch
struct GruF16Params {
    const uint4* w;         // [8 tiles][3 gates][KX + 4 chunks][hi|lo][64 lanes] f16x8 A operands (x chunks first)
    const float* bias;      // [3][128] fp32
    const uint4* wfc;       // LAST: [4 chunks][hi|lo][64]  Wfc^T padded to 16 rows
    const float* bfc;       // [16]
    const float* x_mel;     // FIRST: [B,T,I]
    const uint4* x_prev;    // !FIRST: the layer below's output, split, B-operand order: [G][T][4 chunks][hi|lo][64 lanes]
    uint4* h_out;           // !LAST: this layer's output, same layout
    const float* state_in;  // [B,128] of this layer
    float* state_out;
    const int32_t* seq_len;
    const uint8_t* reset;
    GruLayerParams epi;     // LAST: logits / softmax / tokens / prev_word / thresholds for the epilogue
    int B, T, I;
};
// ... for the shapes the resident kernels do not cover (hidden = 256: BASELINE configs[4]), weights streamed from L2 every frame
// (gru_f16x3_generic.hip): one layer per launch, or all L x G workgroups in one layer-pipelined grid.  Tables as above with
// H/16 tiles, H/32 hidden chunks and the first layer's x chunks padded to an even count; the pipelined launch takes ready_in /
// ready_out / pipe_error from p.epi.
struct GruF16StackParams {
    GruF16Params layer[8];
    int L, G;
    int xcd_affine;
};
static_assert(sizeof(GruF16StackParams) <= 4096, "kernel arguments");
bool gru_f16x3_generic_supported(int hidden, int n_mel);
hipError_t launch_gru_layer_f16x3_generic(const GruF16Params& p, int hidden, bool first, bool last, hipStream_t st);
hipError_t launch_gru_stack_f16x3_pipelined(const GruF16StackParams& sp, int hidden, hipStream_t st);
bool gru_f16x3_supported(int hidden, int n_mel);
hipError_t launch_gru_layer_f16x3(const GruF16Params& p, bool first, bool last, hipStream_t st);    // last && p.epi.win.tab: with the window tail
bool gru_f16x3_vgpr_form();                                 // gru_f16x3.hip built with -mllvm -amdgpu-mfma-vgpr-form=1 (csrc/Makefile)

// int8 ("octbit") GRU layers and class projection (gru_octbit.hip)
struct GruOctbitParams {
    const uint32_t* wg;     // gates  [4 K-quarters][2 unit groups][2 units per lane][32 = 16 couples x (even,odd)][64 lanes]  int16 pairs
    const uint32_t* wc;     // cand.  [8 K-eighths][2 units per lane][16][64 lanes]
    const float* bias;      // [3][128] (r, u, c)
    const float* b127;      // [384]  127 * column sums of Wq: gates 256, candidate 128 (octbit_graph.py:202-204)
    float scale_g, scale_c; // octize_weight_int8_signed scales
    const float4* x_prev;   // previous layer's output, xl layout
    float4* h_out;          // this layer's output, xl layout
    const float* state_in;  // [B,128]
    float* state_out;
    const int32_t* seq_len;
    const uint8_t* reset;
    uint32_t* aq;           // activation exchange [G][2][16 streams][128 dwords]
    float2* range;          // top layer only: (min, max) of each stream's emitted rows, for the projection; else null
    int B, T;
};
struct OctbitFcParams {
    const uint32_t* wfc;    // [8 tiles][4 g][kMaxClasses][even,odd] int16 pairs
    const float* b127;      // [kMaxClasses]
    const float* bfc;       // [16] padded
    float scale_w;
    const float4* h_top;    // top layer output, xl layout
    float2* range;          // [G*16] (min, max) of each stream's [T,H] block
    int range_ready;        // the top int8 layer already produced it
    const int32_t* prev_in; // copy of prev_word taken before the launch (or null)
    float* logits; float* softmax; int8_t* tokens; int32_t* prev_word;
    float decode_thres, value_clip;
    int use_relu, B, T, C;
};
hipError_t launch_gru_layer_octbit(const GruOctbitParams& p, hipStream_t st);
hipError_t launch_octbit_fc(const OctbitFcParams& p, hipStream_t st);

// decode window of the stream manager (stream_kernels.hip)
struct WindowParams {
    int8_t* words;            // [B][nq][tmax] per-frame ctc_decode2 word (-1 none); tmax % 16 == 0
    int* lens;                // [B][nq]
    int* head;                // [B]
    int* count;               // [B]
    const float* softmax;     // [B][T][C] this chunk
    const uint8_t* clear_before;  // [B] or null
    int32_t* hit;             // [B]
    uint8_t* restart;         // [B] or null
    int32_t label[16];
    int label_len;
    float thres;
    int B, T, C, nq, tmax;
};
hipError_t launch_window_step(const WindowParams& p, hipStream_t st);
// the incremental form (summaries per queued chunk, window_device.h): softmax [B][T][C] -> per-frame words -> window tail
struct WindowIncParams {
    WindowTail win;
    uint8_t delta[256];       // by value: the standalone step takes the label per call
    const float* softmax;
    float thres;
    int B, T, C;
};
hipError_t launch_window_inc(const WindowIncParams& p, hipStream_t st);
hipError_t launch_window_reset(int B, int* head, int* count, hipStream_t st);

// PCM -> mel front-end (frontend_kernels.hip)
struct FrontendParams {
    const float* pcm;    // [B, n_samples - n_carry]  the new samples (float input)
    const int16_t* pcm_i16;   // fft_frontend.hip only: int16 PCM instead (scaled by 2^-15 as RingBuffer.get does), pcm unused
    // fft_frontend.hip only, the head of a stream-manager iteration fused into the same launch (gate != 0): vad over the new
    // samples -> silent / reset masks, and the next sample carry (the last n_next samples of [carry | chunk])
    int gate;
    int fft_blocks;      // set by launch_mel_fft400: transform blocks in the grid (multiple of 8)
    float vad_thres;
    const uint8_t* restart;
    uint8_t* silent;
    uint8_t* reset;
    float* next;
    int n_next;
    const float* carry;  // [B, n_carry] samples carried over from the previous chunk (n_carry may be 0)
    float* mel;          // [B, T, n_mel]
    const float* dft;    // [nf_tiles x (cos|sin) x parity][kc4][64][4]  A fragments: bins 0..fft/4 over the folded samples of one parity
    const float* melw;   // [mel_tiles][nf_tiles][direct|mirror][4][64]  A fragments of the mel basis, xl k map over k = 0..fft/4
    int n_samples, T, fft, hop, n_mel, nf_tiles, mel_tiles, kc4, B, n_carry;   // n_samples = n_carry + new samples
#ifdef KWS_FE_TIMING
    long long* timing;
#endif
    int mel_lo[4], mel_cnt[4], mel_off[4];   // fft_frontend.hip: per mel tile, first 4-bin group, number of groups (multiple of 4), offset of its fragments in melw (in groups)
};
hipError_t launch_mel_frontend(const FrontendParams& p, int B, hipStream_t st);
// fft 400 only (fft_frontend.hip): p.dft = twiddles [12][16] (cos, sin), p.melw = basis fragments [tile][group of its run][64]
hipError_t launch_mel_fft400(const FrontendParams& p, int B, hipStream_t st);      // honours p.pcm_i16 and p.gate
hipError_t launch_carry_tail(const float* carry, int n_carry, const float* chunk, int n_chunk, float* next, int n_next, int B,
                             hipStream_t st);

// launchers (gru_kernels.hip)
bool gru_resident_supported(int hidden, int in_dim, bool first);
int gru_resident_kcx(int in_dim, bool first);
hipError_t launch_gru_layer_resident(const GruLayerParams& p, bool first, bool last, hipStream_t st);    // p.win.tab != null: with the window tail
bool gru_resident_takes_window(bool first, bool last);
hipError_t launch_gru_layer_generic(const GruLayerParams& p, int hidden, bool first, bool last,
                                    hipStream_t st);

// decode_kernels.hip
hipError_t launch_ctc_decode(int kind, const float* softmax, const int32_t* lengths, int B, int T, int C,
                             int lockout, float thres, float loose_thres, int32_t* words,
                             int32_t* counts, int max_words, hipStream_t st);
hipError_t launch_ctc_predict(const int32_t* words, const int32_t* counts, int B, int max_words,
                              const int32_t* label_digits, int label_len, int32_t* hit, hipStream_t st);
hipError_t launch_vad(const float* pcm, int B, int N, float thres, uint8_t* speech, float* abs_sum,
                      hipStream_t st);

// state_out = reset[b] ? 0 : state_in for [L,B,H] (a kws_step over zero frames; in-place allowed)
hipError_t launch_state_passthrough(const float* state_in, float* state_out, const uint8_t* reset, int32_t* prev_word, int L, int B, int H,
                                    hipStream_t st);
// ... and, fused into the same pass, the next sample carry: next [B,n_next] = the last n_next samples of [carry | chunk]
// (n_next = 0: none)
hipError_t launch_vad_gate(const void* pcm, int pcm_int16, int B, int N, float thres, float* pcm_f32, const uint8_t* restart,
                           uint8_t* silent, uint8_t* reset, const float* carry, int n_carry, float* next, int n_next, hipStream_t st);

// octbit_kernels.hip
hipError_t launch_octbit_matmul(const float* x, const int8_t* Wq, float scale_w, const float* bias,
                                float* out, int A, int K, int N, int per_row, float* range_ws,
                                hipStream_t st);

}  // namespace kws

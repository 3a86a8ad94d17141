/* kws_amd.h -- C ABI of the MI355X-native streaming GRU keyword-spotting path.
 *
 * Drop-in boundary.  Every entry point replaces one interface of the reference
 * (paths relative to colinsongf/keyword_spotting):
 *
 *   kws_create / kws_destroy   tf.import_graph_def + tf.Session on the frozen DeployModel graph
 *                              (detector.py:134-146; export list main.py:339-342)
 *   kws_step                   sess.run(['model/softmax:0','model/logit:0','model/rnn_states:0'],
 *                              {'model/inputX:0', 'model/rnn_initial_states:0'})
 *                              (detector.py:190-193, :220-223, :280-283) on the mel-input variant
 *                              of the graph (models/rnn_ctc.py:150-153), batched over B streams;
 *                              arithmetic of models/rnn_ctc.py:155-165,202-284
 *   kws_ctc_decode             utils/prediction.py:18 ctc_decode, :65 ctc_decode2, :89 ctc_decode_strict
 *   kws_ctc_predict            utils/prediction.py:111 ctc_predict
 *   kws_vad                    utils/basic_vad.py:17 vad
 *   kws_stream_feed            one iteration of HotwordDetector.start's loop, detector.py:158-209, for B streams
 *   kws_octbit_matmul          REGISTER_OP("OctbitMatMul") octbit/octbit_ops_reg.cc:7-15,
 *                              OctbitMatMulOp::Compute octbit/octbit_mat_mul_op.cc:49-183
 *   kws_octbit_quantize        octize_weight_int8_signed octbit/octbit_graph.py:191-215
 *
 * Conventions
 *   - plain C types only; every tensor pointer is CALLER-OWNED DEVICE memory (hipMalloc / a PyTorch
 *     tensor's data_ptr) unless the parameter is documented as host memory.
 *   - `stream` is a hipStream_t passed as void* (NULL = the legacy default stream).  All work is
 *     enqueued asynchronously on it; no entry point synchronises the device except the create / destroy calls,
 *     kws_kernel_times, kws_reserve, the first kws_window_step of a window (allocates its frame ring) -- and kws_step only
 *     when the call needs more scratch than kws_reserve or any earlier call provided (it then grows the handle's scratch
 *     block, which waits for the device once; kws_scratch_stats counts those events).
 *   - return value: KWS_OK or a negative kws_status.  No exceptions, no abort.  The message for the
 *     last failure on the calling thread is kws_last_error().
 *   - a handle is immutable after kws_create except for its scratch buffer and profiling slots: ONE host thread at a
 *     time per handle.  The reference's OctbitMatMulOp::Compute is re-entrant (octbit/octbit_mat_mul_op.cc:49) because it
 *     allocates its temporaries per call; here the inter-layer seams belong to the handle, so the unit of concurrency is the
 *     handle (0.64 MB of weights + scratch each): one per host thread, each on its own HIP stream -- different handles are
 *     fully independent (tests/test_gpu_soak.py drives two threads on two handles).  Sharing one handle is detected, not
 *     undefined: a thread that enters kws_step / kws_reserve / kws_kernel_times while another is inside gets KWS_ERR_BUSY
 *     and nothing is launched; calls that are serialised by the caller but arrive on a different stream than the call before
 *     are ordered behind it ON THE DEVICE (every call records an event at its end, the next call's stream waits for it when it
 *     is another stream): no host wait, nothing that touches other handles' work, legal under stream capture.  (Streams are
 *     told apart by their handle value: do not destroy a stream with this handle's work still queued and expect a new stream
 *     that reuses the address to be ordered behind it.)  kws_stream_feed holds its MODEL handle for the whole iteration: stream
 *     managers that are fed concurrently need a model handle each; any number of them may share one handle when fed in turn.
 */
#ifndef KWS_AMD_H_
#define KWS_AMD_H_

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

typedef enum kws_status {
    KWS_OK = 0,
    KWS_ERR_INVALID_ARGUMENT = -1, /* the TF InvalidArgumentError cases: bad dims, null ptr, K%64 ... */
    KWS_ERR_UNSUPPORTED = -2,      /* shape outside what the kernels are built for                    */
    KWS_ERR_HIP = -3,              /* a HIP runtime call failed (message has hipGetErrorString)       */
    KWS_ERR_NO_DEVICE = -4,        /* no gfx950 device visible                                        */
    KWS_ERR_OUT_OF_MEMORY = -5,
    KWS_ERR_BUSY = -6              /* another host thread is inside kws_step / kws_reserve / kws_kernel_times / kws_stream_feed on this handle */
} kws_status;

/* Model shape: config/rnn_config.py:57-99 (n_mel :63, hidden_size :84, num_layers :76,
 * num_classes :88-91, use_relu :83, value_clip :80). */
typedef struct kws_config {
    int32_t n_mel;       /* I : mel bins per 10 ms frame                                   */
    int32_t hidden;      /* H : GRU units per layer; supported: 64, 128, 256                */
    int32_t num_layers;  /* L : 1..8                                                        */
    int32_t num_classes; /* C : 3..8 (space, words..., blank)                               */
    int32_t use_relu;    /* models/rnn_ctc.py:280                                           */
    float value_clip;    /* models/rnn_ctc.py:282 : >0 and use_relu -> clip logits to [0,20] */
    int32_t precision;   /* KWS_FP32 (reference arithmetic); KWS_BF16 (BASELINE configs[2]: bf16 weights and
                            matmul inputs, fp32 accumulate / state / activations; H=128, L<=2, n_mel%4==0, <=64);
                            KWS_INT8 (BASELINE configs[2], the graph octbit/octbit_graph.py:461-485 produces: the
                            gate/candidate MatMuls of cell_1.. and the class projection are OctbitMatMul calls --
                            name rule :218-225, weights quantised at kws_create by :191-215, op arithmetic
                            octbit_mat_mul_op.cc:90-181 incl. the int16 pair saturation, activation range per
                            stream as the batch-1 graph has it; layer 0, biases, activations fp32; H=128) */
} kws_config;
enum { KWS_FP32 = 0, KWS_BF16 = 1, KWS_INT8 = 2,
       /* fp32 results on the fp16 matrix pipe: every matmul operand split into two fp16 pieces (22 mantissa bits), three
          v_mfma_f32_16x16x32_f16 per product, fp32 accumulation, activations and state (csrc/gru_f16x3.hip).  Meets the
          fp32 path's tolerance (logits within 1e-4 of the reference semantics; observed ~3e-6) at ~2.5x its throughput;
          it is NOT bit-identical to KWS_FP32.  hidden = 128, n_mel % 4 == 0 and <= 64, any num_layers; |weights| < 64. */
       KWS_F16X3 = 3 };

typedef struct kws_model* kws_handle;

enum { KWS_KERNEL_AUTO = 0, KWS_KERNEL_GENERIC = 1, KWS_KERNEL_RESIDENT = 2 };
enum { KWS_DECODE = 0, KWS_DECODE2 = 1, KWS_DECODE_STRICT = 2 };

/* "kws_amd <ver> (gfx950; HIP x.y.z; <compiler version>; bf16 mfma-vgpr-form=<0|1>; f16x3 mfma-vgpr-form=<0|1>)": the compiler
 * is part of the version because the kernels depend on hand-placed MFMA hazard fences and, for the two files named, on an
 * internal LLVM option that csrc/Makefile applies only when this hipcc accepts it. */
const char* kws_version(void);
/* sizeof(kws_config) / sizeof(kws_frontend_config) as this library was compiled: a binding written in another
 * language (ctypes, cffi, JNI ...) compares it with its own struct declaration at load time, so a field added here
 * can never be read past the end of a caller's shorter struct. */
size_t kws_sizeof_config(void);
size_t kws_sizeof_frontend_config(void);
/* Message of the last error raised on this thread ("" if none). */
const char* kws_last_error(void);

/* Bytes of the canonical fp32 weight blob for `cfg`:
 *   per layer l (I_0 = n_mel, I_l = hidden):  Wg[I_l+H, 2H]  bg[2H]  Wc[I_l+H, H]  bc[H]
 *   then Wfc[H, C]  bfc[C]           -- TF variable layouts, row-major, gate order [r, u]. */
size_t kws_weights_nbytes(const kws_config* cfg);

/* Stages the weights on the current HIP device (re-tiled into MFMA fragment order) and returns a
 * handle.  `weights_blob` is HOST memory of kws_weights_nbytes(cfg) bytes. */
int kws_create(const kws_config* cfg, const void* weights_blob, size_t nbytes, kws_handle* out);
/* Releases the handle and always returns KWS_OK (the handle is gone afterwards, whatever it reports: never retry).  An
 * error a finished asynchronous step had raised and nobody collected is left in kws_last_error(); call kws_poll_error
 * first to get it as a status.  Stream handles created on it (kws_stream_create) fail cleanly afterwards. */
int kws_destroy(kws_handle h);

/* Kernel family used by kws_step (fp32): AUTO picks the register-resident kernels when the shape allows
 * (hidden == 128 and n_mel in {32, 40, 48, 60, 64}), else the generic ones (hidden 64/128/256, any n_mel).
 * RESIDENT on an unsupported shape -> KWS_ERR_UNSUPPORTED.  Ignored by the bf16 stack. */
int kws_set_kernel(kws_handle h, int kind);
/* Pre-sizes the handle's scratch for calls of B streams x up to T frames: afterwards kws_step with this B and T' <= T
 * never allocates or synchronises, whichever launch layout it picks for that shape (sequential layers, layers overlapped
 * on HIP streams, layer-pipelined launch).  A later call with a DIFFERENT B may select a layout this call did not size
 * (e.g. a smaller batch that becomes eligible for the layer-pipelined launch) and then grows the scratch once, which
 * synchronises: reserve every batch size you will use.  The scratch only grows. */
int kws_reserve(kws_handle h, int B, int T);
/* bytes_reserved: device scratch currently held for inter-layer seams; allocations: how many times it (or another
 * batch-sized side buffer) was (re)allocated -- each of those synchronised the device.  Either pointer may be NULL. */
int kws_scratch_stats(kws_handle h, size_t* bytes_reserved, int32_t* allocations);
/* KWS_OK, or the error a finished asynchronous step of this handle raised on the device (today: a layer-pipelined
 * launch whose wait for the layer below timed out -- the results of that step are invalid).  Does not synchronise: to
 * validate a given step, synchronise its stream first.  The same condition is also reported by the next kws_step and
 * by kws_kernel_times, whichever comes first; reporting clears it.  (kws_destroy only leaves it in kws_last_error().) */
int kws_poll_error(kws_handle h);

/* Proves the kernels `h` would launch (its shape and precision; for fp32 both the register-resident and the generic
 * family where the shape allows) against known answers the library carries itself: TensorFlow's published GRUCell /
 * MultiRNNCell unit-test constants (0.175991, 0.156736, 0.13248) embedded in the handle's shape, and 19 streams x 8
 * random frames against a plain double-precision host loop of the cell (models/rnn_ctc.py:179-185,228-243 semantics).
 * Uses temporary handles and buffers, synchronises the device, a few milliseconds.  KWS_OK, or KWS_ERR_HIP with a
 * message naming the kernel, the deviation and kws_version().  With KWS_SELFTEST=1 in the environment every
 * kws_create runs it and fails the same way.  (There is still no CPU fallback: a failed self-test is an error.) */
int kws_selftest(kws_handle h);
/* Name of the kernel(s) the last kws_step of this handle launched for profiling slot `slot` (0..L-1; "" when that layer
 * ran inside another slot's launch), e.g. "gru_layer_resident<32, false, true>".  buf: host memory of n bytes. */
int kws_last_launch(kws_handle h, int slot, char* buf, size_t n);

/* Advances B independent streams by T frames (one 10 ms hop each).
 *   mel        [B,T,I]  f32, 16-byte aligned         model/inputX:0 (mel variant), batch-major
 *   state_in   [L,B,H]  f32                          model/rnn_initial_states:0
 *   logits     [B,T,C]  f32   out, may be NULL       model/logit:0
 *   softmax    [B,T,C]  f32   out, may be NULL       model/softmax:0
 *   state_out  [L,B,H]  f32   out, may alias state_in  model/rnn_states:0
 *   seq_len    [B] i32, may be NULL (= T): rows with t >= seq_len[b] keep their state and emit the
 *              zero-output row (logits = bfc), as dynamic_rnn(sequence_length=...) does
 *   reset_mask [B] u8,  may be NULL: non-zero -> stream b starts this call from the zero state and
 *              prev_word = -1 (detector.py:313-316 clean_state + prob_queue.clear())
 *   tokens     [B,T] i8 out, may be NULL: fused ctc_decode2 -- 0, or the word (1..C-2) emitted at
 *              frame t (utils/prediction.py:74-80 with thres = decode2_thres)
 *   prev_word  [B] i32 in/out, required iff tokens != NULL: ctc_decode2's pre_word carried across
 *              calls (-1 = none)
 * T == 0 or B == 0 is a no-op (state_out = state_in). */
int kws_step(kws_handle h, const float* mel, const float* state_in, float* logits, float* softmax,
             float* state_out, const int32_t* seq_len, const uint8_t* reset_mask, int8_t* tokens,
             int32_t* prev_word, float decode2_thres, int B, int T, void* stream);

/* Per-kernel timing (bench.py roofline): when enabled, kws_step brackets each kernel launch with
 * hipEvents on `stream`.  kws_kernel_times synchronises those events and returns, per layer kernel
 * slot (0..L-1), the summed milliseconds and the launch count since the last reset. */
int kws_set_profiling(kws_handle h, int enable);
int kws_kernel_times(kws_handle h, float* ms_sum /*[L]*/, int32_t* launches /*[L]*/, int reset);

/* Greedy CTC collapse over whole windows, one GPU thread per stream.
 *   kind      KWS_DECODE (uses lockout, thres, loose_thres; columns 1:5 -> needs C >= 5),
 *             KWS_DECODE2 (thres), KWS_DECODE_STRICT (lockout, thres)
 *   softmax   [B,T,C] f32;  lengths [B] i32 or NULL (= T)
 *   words     [B,max_words] i32 out: emitted words in order (without the interleaved zeros of the
 *             reference's [0,w,0,...] format);  counts [B] i32 out (may exceed max_words: truncated) */
int kws_ctc_decode(int kind, const float* softmax, const int32_t* lengths, int B, int T, int C,
                   int lockout, float thres, float loose_thres, int32_t* words, int32_t* counts,
                   int max_words, void* stream);

/* ctc_predict: hit[b] = 1 iff the digits of label (host string of '1'..'9') occur contiguously in
 * words[b, :min(counts[b], max_words)]. */
int kws_ctc_predict(const int32_t* words, const int32_t* counts, int B, int max_words,
                    const char* label, int32_t* hit, void* stream);

/* vad: speech[b] = (sum_n |pcm[b,n]| > thres).  pcm [B,N] f32. */
int kws_vad(const float* pcm, int B, int N, float thres, uint8_t* speech, float* abs_sum_or_null,
            void* stream);

/* PCM -> mel front-end of the deploy graph (models/rnn_ctc.py:134-149, utils/stft.py:27-81): frames of
 * fft_size samples every hop_size (no padding, no window), |rfft|, projection on librosa.filters.mel(sr, n_fft,
 * n_mels, fmin, fmax) (Slaney scale, area-normalised).  pcm [B,N] f32 -> mel [B,T,n_mel], T = 1+(N-fft)/hop
 * (kws_frontend_frames).  fft_size must be a multiple of 16. */
typedef struct kws_frontend_config {
    int32_t samplerate, fft_size, hop_size, n_mel;
    float fmin, fmax;
} kws_frontend_config;
typedef struct kws_frontend* kws_frontend_handle;
int kws_frontend_create(const kws_frontend_config* cfg, kws_frontend_handle* out);
int kws_frontend_destroy(kws_frontend_handle h);
int kws_frontend_frames(const kws_frontend_config* cfg, int n_samples);
int kws_frontend_run(kws_frontend_handle h, const float* pcm, int B, int n_samples, float* mel, void* stream);
/* The streaming form (detector.py:179-183): the signal of stream b is carry[b] followed by chunk[b] --
 * np.concatenate((self.res, data)) -- read in place, never materialised; mel [B,T,n_mel] with
 * T = kws_frontend_frames(n_carry + n_chunk) (nothing is written when that is 0).  Also writes next_carry [B,n_next],
 * the last n_next samples of that signal (self.res = data[-res:]); it must not overlap carry or chunk. */
int kws_frontend_run_carry(kws_frontend_handle h, const float* carry, int n_carry, const float* chunk, int n_chunk,
                           int B, float* mel, float* next_carry, int n_next, void* stream);
/* Copies the fp32 mel basis [n_mel, fft/2+1] (host memory) the handle was built with -- for inspection/tests. */
int kws_frontend_mel_basis(kws_frontend_handle h, float* basis_host);

/* Device-side decode window of the streaming loop (detector.py:122,168-209; utils/queue.py): per stream a
 * bounded FIFO of up to `max_chunks` (1..64) softmax chunks (each <= max_frames frames; 2 * max_chunks *
 * round_up(max_frames, 16) bytes must fit 48 KiB, else KWS_ERR_UNSUPPORTED).  The frame ring behind kws_window_step is
 * allocated by its first call (a window that is only driven incrementally never holds one).  ctc_decode2's per-frame rule
 * (argmax over classes 1..C-2, strictly above `thres`) is a function of the frame alone, so the window stores each
 * frame's word rather than its softmax row; `thres` is therefore fixed per handle.  kws_window_step, per stream:
 *   clear_before[b] != 0 -> empty the window first (silence: detector.py:171-177);
 *   append softmax[b] (dropping the oldest chunk when full); ctc_decode2 over the concatenated window;
 *   hit[b] = label occurs in the decoded words (ctc_predict); on a hit the window is emptied and restart[b]=1
 *   (the caller passes it as the next kws_step reset_mask: detector.py:202-208). */
typedef struct kws_window* kws_window_handle;
int kws_window_create(int B, int max_chunks, int max_frames, int C, float thres, kws_window_handle* out);
int kws_window_destroy(kws_window_handle h);
int kws_window_step(kws_window_handle h, const float* softmax /*[B,T,C]*/, int T, const uint8_t* clear_before,
                    const char* label, int32_t* hit /*[B]*/, uint8_t* restart /*[B] or NULL*/, void* stream);

/* The same window step in incremental form: the window keeps a SUMMARY per queued chunk (first / last frame word and where the
 * label matcher ends up for each of its <= 16 entry states) instead of the chunk's frames, and evaluates the <= max_chunks
 * summaries, oldest first -- O(chunks) per step, not O(frames in the window).  Whether a chunk's first frame emits depends
 * on the chunk before it in the window, which is what an eviction changes: that one decision is taken at evaluation time, so
 * the result is exactly the re-scan's (property-tested against it, evictions / empty chunks / clears / triggers included:
 * tests/test_window_incremental.py, tests/test_gpu_window.py).  Same arguments and results as kws_window_step.  The summaries
 * are label-specific: the first call binds `label` to the window (uploads its matcher: synchronises once), later calls must
 * pass the same one (KWS_ERR_INVALID_ARGUMENT otherwise).  The incremental state is separate from kws_window_step's frame
 * ring: drive a window through ONE of the two entry points.  kws_stream_feed uses this form -- inside the last GRU layer's
 * launch where that kernel has the tail (fp32 / f16x3 / bf16 stacks at hidden = 128, chunks of <= 64 frames, windows of
 * <= 24 chunks) AND every workgroup takes one group of 16 streams (B <= 16 x the device's CUs: 4096 on 256 CUs); as a launch
 * of its own otherwise.  That launch stages 16 streams' frame words and rings in LDS: 16 * round_up(T, 16) + 256 +
 * 16 * (32 * max_chunks + 32) bytes must fit 160 KiB, else KWS_ERR_UNSUPPORTED (kws_window_create only sizes the re-scan). */
int kws_window_step_incremental(kws_window_handle h, const float* softmax /*[B,T,C]*/, int T, const uint8_t* clear_before,
                                const char* label, int32_t* hit /*[B]*/, uint8_t* restart /*[B] or NULL*/, void* stream);

/* The whole loop iteration of detector.py:158-209 for B streams as ONE call (device-side stream manager, native):
 *   data = ring_buffer.get()                      `pcm` [B,n]: float samples, or int16 PCM scaled by 2^-15 (:40-43,74-79)
 *   vad(data, vad_thres) false -> clean_state() + prob_queue.clear()                                   (:168-177)
 *   data = concatenate(res, data); res = data[-keep:]      carried samples, never materialised          (:179-183)
 *   softmax, state = sess.run(...)                 front-end + GRU stack on the carried state           (:190-196)
 *   prob_queue.add(softmax); ctc_decode2 over the window; ctc_predict(label)                            (:195-201)
 *   on a hit: window cleared, state reset requested for the next chunk                                  (:202-208)
 * i.e. kws_vad -> kws_frontend_run_carry -> kws_step -> kws_window_step_incremental with the mask logic fused into the first
 * kernel, the window step into the last, and no host work per stream: three launches per chunk (gate + front-end, two GRU
 * layers; two for the bf16 stack) when the window step rides in the last layer's launch (conditions above), one more
 * (window_inc_kernel) otherwise -- e.g. at B > 16 x CUs.  `label` is bound to `window` at kws_stream_create.  The handle
 * BORROWS the model, front-end and window handles (they should outlive it; a feed after one of them was destroyed fails with
 * KWS_ERR_INVALID_ARGUMENT) and the caller-owned device buffers `state` [L,B,H] (zero it to start) and `restart` [B] u8
 * (zero it).  It OWNS only the sample carry (2 x (fft_size - 1) floats per stream); one chunk's intermediates (mel, softmax,
 * masks) are carved out of a staging block of the MODEL handle that all its stream handles share, so M managers on one model
 * cost M x (state + carry + window summaries) -- about 4.8 KB per stream at the reference's shape -- not M x a chunk's
 * buffers.  An empty chunk (n == 0) is skipped as detector.py:164-166 does.  Fewer than
 * fft_size samples in total so far: the reference still runs its whole iteration on such a chunk, and so does this --
 * the VAD decision clears state and window, every sample is carried, the model runs over zero frames (state handed
 * back, or zeroed where the VAD said silence) and the empty softmax takes a slot of the window before the windowed
 * decode.  `label`: up to 15 digits '1'..'9' (the incremental window's matcher has 16 states). */
typedef struct kws_stream* kws_stream_handle;
int kws_stream_create(kws_handle model, kws_frontend_handle frontend, kws_window_handle window, int B, int max_chunk_samples,
                      float vad_thres, const char* label, float* state, uint8_t* restart, kws_stream_handle* out);
int kws_stream_destroy(kws_stream_handle h);
/* Forgets the carried samples (the model state, restart mask and window belong to the caller). */
int kws_stream_reset(kws_stream_handle h);
int kws_stream_feed(kws_stream_handle h, const void* pcm /*[B,n] device*/, int n, int pcm_int16, int32_t* hit /*[B] device*/,
                    void* stream);

/* OctbitMatMul: out[A,N] = (sum_k u8(x)[a,k] * Wq[n,k] - signed*bias[n]) * scale_w * s_x.
 *   x [A,K] f32, Wq [N,K] s8 (pre-transposed), bias [N] f32, out [A,N] f32.  K % 64 == 0, scale_w > 0.
 *   per_row_scale = 0: one dynamic activation range over the whole x (the reference op, whose A is
 *   1 in streaming);  1: one range per row a (what a batch of independent streams needs to
 *   reproduce the batch-1 result).  The u8*s8 pair sums saturate to int16 as _mm_maddubs_epi16 does. */
int kws_octbit_matmul(const float* x, const int8_t* Wq, float scale_w, const float* bias, float* out,
                      int A, int K, int N, int per_row_scale, void* stream);

/* Host-side quantiser: W [K,N] f32 (host) -> Wq [N,K] s8 (host), *scale, bias [N] f32 (host). */
int kws_octbit_quantize(const float* W, int K, int N, int8_t* Wq, float* scale, float* bias);

#ifdef __cplusplus
}
#endif
#endif /* KWS_AMD_H_ */

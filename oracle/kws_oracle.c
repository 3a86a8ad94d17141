/* TEST INFRASTRUCTURE ONLY -- plain-C CPU restatement of the reference's streaming GRU path and of
 * its int8 ("octbit") matmul.  Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg
 * may load the library built from this file (oracle/build.py -> oracle/_build/libkws_oracle.so).
 * The product library (keyword_spotting_amd/csrc) never links or calls it.
 *
 * GRU part: PARITY UNPINNED (TensorFlow 1.x GRUCell/dynamic_rnn is an absent, un-pinned third-party
 *   dependency; see oracle/gru_oracle.py header).  This file is validated against oracle/gru_oracle.py.
 *   Follows models/rnn_ctc.py:155-165 (DeployModel), :202-244 (inference1), :247-284 (inference2).
 * Octbit part: PINNED by the two known-answer tests of octbit/octbit_ops_test.py:24-34,41-53.
 *   Follows octbit/octbit_mat_mul_op.cc:90-181 step by step, with _mm_maddubs_epi16 written out as
 *   scalar arithmetic (u8*s8 adjacent-pair sum saturated to int16).
 */
#include <math.h>
#include <stdint.h>
#include <stdlib.h>
#include <string.h>

typedef struct {
    int n_mel, hidden, num_layers, num_classes, use_relu;
    float value_clip;
} oracle_cfg;

static inline float sigmoidf_(float x) { return 1.0f / (1.0f + expf(-x)); }

/* y[j] = b[j] + sum_k v[k] * W[k*n + j]   (row-major W[k_rows, n]) -- axpy form, vectorises */
static void affine(const float* v, int k_rows, const float* W, const float* b, int n, float* y) {
    for (int j = 0; j < n; ++j) y[j] = b[j];
    for (int k = 0; k < k_rows; ++k) {
        const float a = v[k];
        const float* w = W + (size_t)k * n;
        for (int j = 0; j < n; ++j) y[j] += a * w[j];
    }
}

/* One stream, T frames.  mel [T,I]; state [L][H] in/out (strided by state_stride between layers);
 * logits / softmax [T,C] (softmax may be NULL).  seq_len < T freezes the state and emits the
 * zero-output row (logits = bias) as dynamic_rnn does. */
static void gru_stream(const oracle_cfg* c, const float* blob, const float* mel, const float* s_in,
                       float* s_out, size_t state_stride, int T, int seq_len, float* logits,
                       float* softmax) {
    const int I = c->n_mel, H = c->hidden, L = c->num_layers, C = c->num_classes;
    float* h = (float*)malloc(sizeof(float) * (size_t)L * H);
    float* v = (float*)malloc(sizeof(float) * (size_t)(2 * H + (I > H ? I : H)));
    float* g = (float*)malloc(sizeof(float) * (size_t)3 * H);
    float* zero = (float*)calloc((size_t)H, sizeof(float));
    for (int l = 0; l < L; ++l) memcpy(h + (size_t)l * H, s_in + l * state_stride, sizeof(float) * H);
    for (int t = 0; t < T; ++t) {
        const float* top = zero;
        if (t < seq_len) {
            const float* x = mel + (size_t)t * I;
            int in = I;
            const float* p = blob;
            for (int l = 0; l < L; ++l) {
                const int K = in + H;
                const float *Wg = p, *bg = Wg + (size_t)K * 2 * H, *Wc = bg + 2 * H,
                            *bc = Wc + (size_t)K * H;
                float* hl = h + (size_t)l * H;
                memcpy(v, x, sizeof(float) * in);
                memcpy(v + in, hl, sizeof(float) * H);
                affine(v, K, Wg, bg, 2 * H, g);                       /* [x,h] Wg + bg */
                for (int j = 0; j < 2 * H; ++j) g[j] = sigmoidf_(g[j]);  /* r = g[0:H], u = g[H:2H] */
                for (int j = 0; j < H; ++j) v[in + j] = g[j] * hl[j];    /* [x, r*h] */
                affine(v, K, Wc, bc, H, g + 2 * H);
                for (int j = 0; j < H; ++j) {
                    const float cand = tanhf(g[2 * H + j]), u = g[H + j];
                    hl[j] = u * hl[j] + (1.0f - u) * cand;
                }
                x = hl;
                in = H;
                p = bc + H;
            }
            top = h + (size_t)(L - 1) * H;
        }
        {
            const float* p = blob;
            int in = I;
            for (int l = 0; l < L; ++l) { p += (size_t)(in + H) * 3 * H + 3 * H; in = H; }
            float* lg = logits + (size_t)t * C;
            affine(top, H, p, p + (size_t)H * C, C, lg);
            if (c->use_relu) {
                for (int j = 0; j < C; ++j) {
                    lg[j] = lg[j] > 0.f ? lg[j] : 0.f;
                    if (c->value_clip > 0.f && lg[j] > 20.f) lg[j] = 20.f;
                }
            }
            if (softmax) {
                float m = lg[0], s = 0.f;
                for (int j = 1; j < C; ++j) m = lg[j] > m ? lg[j] : m;
                for (int j = 0; j < C; ++j) { softmax[(size_t)t * C + j] = expf(lg[j] - m); s += softmax[(size_t)t * C + j]; }
                for (int j = 0; j < C; ++j) softmax[(size_t)t * C + j] /= s;
            }
        }
    }
    for (int l = 0; l < L; ++l) memcpy(s_out + l * state_stride, h + (size_t)l * H, sizeof(float) * H);
    free(h); free(v); free(g); free(zero);
}

/* mel [B,T,I], state [L,B,H], logits/softmax [B,T,C], seq_len [B] or NULL.  Streams are independent;
 * `threads` > 1 runs them under OpenMP (one stream per thread at a time). */
int oracle_gru_forward(const oracle_cfg* c, const float* blob, const float* mel, const float* state_in,
                       const int32_t* seq_len, float* logits, float* softmax, float* state_out, int B,
                       int T, int threads) {
    const int I = c->n_mel, H = c->hidden, C = c->num_classes;
    (void)threads;
#pragma omp parallel for num_threads(threads > 0 ? threads : 1) schedule(dynamic, 1)
    for (int b = 0; b < B; ++b) {
        gru_stream(c, blob, mel + (size_t)b * T * I, state_in + (size_t)b * H,
                   state_out + (size_t)b * H, (size_t)B * H, T, seq_len ? seq_len[b] : T,
                   logits + (size_t)b * T * C, softmax ? softmax + (size_t)b * T * C : NULL);
    }
    return 0;
}

/* utils/prediction.py:65-86 as a per-frame state machine; out holds the emitted words (not the
 * 0-interleaved form); returns the count. */
int oracle_ctc_decode2(const float* softmax, int T, int C, float thres, int32_t* out) {
    int prev = -1, n = 0;
    for (int t = 0; t < T; ++t) {
        const float* p = softmax + (size_t)t * C;
        int best = 1;
        for (int j = 2; j < C - 1; ++j) if (p[j] > p[best]) best = j;
        const int w = (C > 2 && p[best] > thres) ? best - 1 : -1;
        if (w >= 0 && w != prev) out[n++] = w + 1;
        prev = w;
    }
    return n;
}

/* ---- octbit/octbit_mat_mul_op.cc:90-181 -------------------------------------------------------- */
static inline int16_t sat16(int32_t v) { return v > 32767 ? 32767 : (v < -32768 ? -32768 : (int16_t)v); }

/* x [A,K] f32, Wq [N,K] s8 (already transposed, :41), bias [N], out [A,N].  Returns 0, or -1 when a
 * precondition of :46,:65-67 fails (scale <= 0, K % 64 != 0). */
int oracle_octbit_matmul(const float* x, const int8_t* Wq, float scale_w, const float* bias, float* out,
                         int A, int K, int N) {
    if (!(scale_w > 0.f) || K % 64 != 0) return -1;
    float mn = 3.402823466e+38f, mx = -3.402823466e+38f;                 /* :92-99 */
    for (int i = 0; i < A * K; ++i) { if (x[i] < mn) mn = x[i]; if (x[i] > mx) mx = x[i]; }
    const int is_signed = mn < 0.f;                                       /* :101 */
    uint8_t* q = (uint8_t*)malloc((size_t)A * K);
    float scale = scale_w;
    if (is_signed) {                                                      /* :105-114 */
        const float bscale = (-mn > mx ? -mn : mx) / 127;
        scale *= bscale;
        for (int i = 0; i < A * K; ++i) q[i] = (uint8_t)(round(x[i] / bscale) + 127);
    } else {                                                              /* :115-124 */
        const float bscale = mx / 254;
        scale *= bscale;
        for (int i = 0; i < A * K; ++i) q[i] = (uint8_t)round(x[i] / bscale);
    }
    for (int n = 0; n < N; ++n) {                                         /* :137-181 */
        for (int a = 0; a < A; ++a) {
            int32_t lane[4] = {0, 0, 0, 0};                              /* the __m128i sum, :141-145 */
            for (int k = 0; k < K; k += 2) {                              /* maddubs: pair, saturate */
                const int32_t pair = (int32_t)q[a * K + k] * Wq[(size_t)n * K + k] +
                                     (int32_t)q[a * K + k + 1] * Wq[(size_t)n * K + k + 1];
                lane[(k / 2) % 4] += sat16(pair);        /* lo/hi halves of each 8 x i16 fold, :152-154 */
            }
            float o = 0.f;
            for (int m = 0; m < 4; ++m) o += (float)lane[m];             /* :172-175 float adds */
            if (is_signed) o -= bias[n];
            out[(size_t)a * N + n] = o * scale;
        }
    }
    free(q);
    return 0;
}

// C ABI of libkws_amd.so (include/kws_amd.h): handle management, weight re-tiling, kws_step.
#include <algorithm>
#include <atomic>
#include <cmath>
#include <cstdarg>
#include <cstdio>
#include <cstdint>
#include <cstring>
#include <memory>
#include <mutex>
#include <new>
#include <string>
#include <unordered_map>
#include <vector>

#include "kws_internal.h"

namespace {

thread_local std::string g_last_error;
thread_local bool g_in_selftest = false;      // kws_selftest creates temporary handles: no recursion through KWS_SELFTEST=1

// Live handles (model / front-end / window): a stream handle borrows all three, and its owner may destroy one of them
// first.  kws_stream_feed checks its borrowed pointers here -- pointer AND the serial it saw at kws_stream_create, so a new
// handle that reuses a freed address does not pass -- and fails with KWS_ERR_INVALID_ARGUMENT instead of touching freed
// memory.
std::mutex g_live_mutex;
std::unordered_map<const void*, unsigned long long> g_live;
unsigned long long g_live_serial = 0;
unsigned long long live_register(const void* h) {
    std::lock_guard<std::mutex> lock(g_live_mutex);
    return g_live[h] = ++g_live_serial;
}
void live_unregister(const void* h) {
    std::lock_guard<std::mutex> lock(g_live_mutex);
    g_live.erase(h);
}
unsigned long long live_serial(const void* h) {      // 0: not a live handle
    std::lock_guard<std::mutex> lock(g_live_mutex);
    const auto it = g_live.find(h);
    return it == g_live.end() ? 0ull : it->second;
}

// kws_octbit_matmul's activation-range workspace, one per (device, stream) that has called it (never freed: a few KB each)
struct OctbitWorkspace { std::mutex mutex; float* p = nullptr; size_t floats = 0; };
struct PairHash { size_t operator()(const std::pair<int, const void*>& k) const { return std::hash<const void*>()(k.second) * 31u + (size_t)k.first; } };
std::mutex g_octbit_ws_mutex;
std::unordered_map<std::pair<int, const void*>, std::unique_ptr<OctbitWorkspace>, PairHash> g_octbit_ws;

int fail(int code, const char* fmt, ...) {
    char buf[1024];
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(buf, sizeof(buf), fmt, ap);
    va_end(ap);
    g_last_error = buf;
    return code;
}
int hip_fail(hipError_t e, const char* what) {
    return fail(e == hipErrorOutOfMemory ? KWS_ERR_OUT_OF_MEMORY : KWS_ERR_HIP, "%s: %s", what,
                hipGetErrorString(e));
}
// scope guard of kws_model::in_call
struct BusyGuard {
    std::atomic<int>* flag;
    bool owned;
    explicit BusyGuard(std::atomic<int>& f) : flag(&f), owned(f.exchange(1, std::memory_order_acquire) == 0) {}
    ~BusyGuard() { if (owned) flag->store(0, std::memory_order_release); }
    BusyGuard(const BusyGuard&) = delete;
    BusyGuard& operator=(const BusyGuard&) = delete;
};
#define KWS_HIP(call)                                         \
    do {                                                      \
        hipError_t e_ = (call);                               \
        if (e_ != hipSuccess) return hip_fail(e_, #call);     \
    } while (0)

struct LayerDev {
    int in_dim;
    // offsets (floats) into the single device allocation
    size_t wx_res, wx_gen, wh_gen, bias;   // wx_res: first-layer x-part, interleaved k map (resident kernel)
    int kcx_res, kcx_gen;
    bool resident_ok;
};

}  // namespace

struct kws_model {
    kws_config cfg;
    int device = 0;
    int kernel_kind = KWS_KERNEL_AUTO;
    std::vector<LayerDev> layers;
    size_t wfc_off = 0, bfc_off = 0;
    // bf16 stack: offsets (floats) of the packed bf16 A operands
    size_t bf_w[2] = {0, 0}, bf_wfc = 0;
    int bf_kx0 = 0;
    // f16x3 split stack: per layer the packed (hi, lo) fp16 A operands, and the projection's
    std::vector<size_t> f16_w;
    size_t f16_wfc = 0;
    int f16_kx0 = 0;
    bool f16_generic = false;        // hidden != 128: the L2-streaming kernels (gru_f16x3_generic.hip)
    // int8 ("octbit") variant: per quantised layer the packed int16 couples + 127*colsum, and the projection
    struct OctLayer { bool quantised = false; size_t wg = 0, wc = 0, b127 = 0; float scale_g = 0.f, scale_c = 0.f; };
    std::vector<OctLayer> oct;
    size_t oct_wfc = 0, oct_b127fc = 0;
    float oct_scale_fc = 0.f;
    uint32_t* oct_aq = nullptr;      // activation exchange [groups][2][16][128]
    float2* oct_range = nullptr;     // [groups*16]
    int32_t* oct_prev = nullptr;     // [B] copy of prev_word
    size_t oct_groups = 0;
    float* d_weights = nullptr;
    // Inter-layer seams.  ONE device allocation per memory kind that only ever grows (kws_reserve or the first call that
    // needs more); each kws_step carves the buffers of its launch layout out of it -- sequential (1-2 buffers of T
    // frames), layers overlapped on HIP streams (2(L-1) buffers of a time block), layer-pipelined (L-1 fine-grained
    // buffers) -- so alternating layouts never reallocates, and a step within the reserved size never synchronises.
    struct Arena { char* base = nullptr; size_t bytes = 0; };
    Arena arena, arena_fine;
    int scratch_allocs = 0;          // (re)allocations so far; each one synchronised the device (kws_scratch_stats)
    float4* scratch[8] = {nullptr, nullptr, nullptr, nullptr, nullptr, nullptr, nullptr, nullptr};   // this call's seams: l -> scratch[l % nscratch]
    int nscratch = 0;
    bool pipe_disabled = false;
    // time-blocked overlap of the layers on separate HIP streams (step_overlapped)
    hipStream_t lane_stream[8] = {nullptr, nullptr, nullptr, nullptr, nullptr, nullptr, nullptr, nullptr};
    std::vector<hipEvent_t> ovl_events;
    hipEvent_t ovl_tail[8] = {nullptr, nullptr, nullptr, nullptr, nullptr, nullptr, nullptr, nullptr};   // end of the last overlapped call, per lane
    bool ovl_tail_valid = false;      // fine-grained memory unavailable, or a pipelined launch timed out: sequential launches from then on
    // layer-pipelined launch of the generic kernel
    int num_cus = 0;
    int* pipe_ready = nullptr;       // [L][groups] frames published
    size_t pipe_groups = 0;
    int* pipe_error_host = nullptr;  // mapped pinned flag the kernel raises if a wait times out
    int* pipe_error_dev = nullptr;
    // One host thread at a time per handle (kws_amd.h): a second thread that enters kws_step / kws_reserve / kws_kernel_times
    // while another is inside gets KWS_ERR_BUSY instead of racing on the scratch arena and the profiling slots.
    std::atomic<int> in_call{0};
    // The seams (and the stream managers' staging below) are shared by all calls of the handle.  Every call records
    // `last_done` on its stream when its last launch is queued; a call that arrives on ANOTHER HIP stream than the one before
    // makes its stream wait for that event (device-side ordering: no host stall, nothing that touches other handles' work,
    // legal under stream capture).  The common path -- same stream as before -- pays one hipEventRecord.
    hipStream_t last_stream = nullptr;
    bool last_stream_valid = false;
    hipEvent_t last_done = nullptr;
    // Staging of the stream managers that borrow this handle (kws_stream_feed): widened PCM, mel, softmax and the two masks
    // of ONE chunk.  They live only inside a feed, feeds of one handle are ordered (one host thread at a time, stream
    // switches ordered by last_done), so every manager on the handle carves the same block: M managers cost M x their
    // per-stream state, not M x a chunk's intermediates.  Grows at kws_stream_create only.
    Arena stage;
    // profiling
    bool profiling = false;
    struct Pending { int slot; hipEvent_t a, b; };
    std::vector<Pending> pending;
    std::vector<hipEvent_t> event_pool;
    std::vector<float> ms_sum;
    std::vector<int32_t> launches;
    // kernel the last kws_step launched per profiling slot, as a small tag: the name is only formatted when somebody asks
    // (kws_last_launch, kws_selftest) -- not on the launch path, where a 22-frame call is ~100 us of device time
    enum LaunchFamily : uint8_t { kNone = 0, kBf16Stack, kF16x3, kPipelined, kOctbit, kOctbitFc, kResident, kGeneric, kF16x3Generic, kF16x3Pipelined };
    struct LaunchTag { uint8_t family = kNone, kx = 0, first = 0, last = 0, window = 0; };
    LaunchTag launch_tag[8];
    std::string launch_name(int slot) const {
        const LaunchTag& t = launch_tag[slot];
        char nm[96];
        nm[0] = 0;
        switch (t.family) {
            case kBf16Stack: break;
            case kF16x3: snprintf(nm, sizeof(nm), "gru_layer_f16x3<%d, %s, %s>", t.kx, t.first ? "true" : "false", t.last ? "true" : "false"); break;
            case kPipelined: snprintf(nm, sizeof(nm), "gru_stack_generic_pipelined<%d> (all %d layers, one launch)", t.kx, cfg.num_layers); break;
            case kF16x3Generic: snprintf(nm, sizeof(nm), "gru_layer_f16x3_generic<%d, %s, %s>", t.kx, t.first ? "true" : "false", t.last ? "true" : "false"); break;
            case kF16x3Pipelined: snprintf(nm, sizeof(nm), "gru_stack_f16x3_pipelined<%d> (all %d layers, one launch)", t.kx, cfg.num_layers); break;
            case kOctbit: snprintf(nm, sizeof(nm), "gru_layer_octbit_kernel"); break;
            case kOctbitFc: snprintf(nm, sizeof(nm), "gru_layer_octbit_kernel + octbit_fc_kernel"); break;
            case kResident: snprintf(nm, sizeof(nm), "gru_layer_resident<%d, %s, %s>", t.kx, t.first ? "true" : "false", t.last ? "true" : "false"); break;
            case kGeneric: snprintf(nm, sizeof(nm), "gru_layer_generic<%d, %s, %s>", t.kx, t.first ? "true" : "false", t.last ? "true" : "false"); break;
            default: break;
        }
        std::string out = t.family == kBf16Stack ? std::string(kws::gru_stack_bf16_kernel_name(bf_kx0, cfg.num_layers)) : std::string(nm);
        if (t.window) out += " + window tail";          // the stream manager's decode-window step rode in this launch
        return out;
    }
};

struct kws_window {
    int B = 0, nq = 0, tmax = 0, tmax_pad = 0, C = 0;
    float thres = 0.f;
    int8_t* words = nullptr;
    int *lens = nullptr, *head = nullptr, *count = nullptr;
    // the incremental form (kws_window_step_incremental, kws_stream_feed): a summary per queued chunk instead of its frames
    // (window_device.h); a state of its own -- a window is driven through one of the two entry points, not both
    uint8_t* inc_tab = nullptr;      // [B][nq][32]  tab | ftab
    uint32_t* inc_meta = nullptr;    // [B][nq]
    int *inc_head = nullptr, *inc_count = nullptr;
    uint8_t* inc_delta_dev = nullptr;   // [256] label matcher of the bound label
    uint8_t inc_delta[256] = {0};
    char inc_label[17] = {0};
    bool inc_bound = false;
};

struct kws_stream {
    kws_model* model = nullptr;
    kws_frontend* fe = nullptr;
    kws_window* win = nullptr;
    unsigned long long model_serial = 0, fe_serial = 0, win_serial = 0;   // live_serial() of the three at kws_stream_create
    int B = 0, max_chunk = 0, tmax = 0, n_carry = 0, cur = 0;
    float vad_thres = 0.f;
    char label[17] = {0};
    float* state = nullptr;          // caller-owned [L,B,H]
    uint8_t* restart = nullptr;      // caller-owned [B]
    float* carry[2] = {nullptr, nullptr};   // [B, fft - 1] each: the carried samples ping-pong (the only device memory a manager owns)
    // one chunk's intermediates, carved out of the MODEL handle's staging block (kws_model::stage) at every feed:
    size_t off_pcm_f32 = 0;          // [B, max_chunk]  int16 input widened here (front-ends other than the 400-point FFT, sub-frame chunks)
    size_t off_mel = 0;              // [B, tmax, n_mel]
    size_t off_softmax = 0;          // [B, tmax, C]
    size_t off_silent = 0;           // [B]
    size_t off_reset = 0;            // [B]
    size_t stage_bytes = 0;
    // the pointers of the current feed
    float* pcm_f32 = nullptr;
    float* mel = nullptr;
    float* softmax = nullptr;
    uint8_t* silent = nullptr;
    uint8_t* reset = nullptr;
};

struct kws_frontend {
    kws_frontend_config cfg;
    float* d_tables = nullptr;
    size_t dft_off = 0, melw_off = 0;
    size_t fft_tw_off = 0, fft_mel_off = 0;   // fft_frontend.hip tables (fft_size 400 only)
    int mel_lo[4] = {0, 0, 0, 0}, mel_cnt[4] = {0, 0, 0, 0}, mel_off[4] = {0, 0, 0, 0};
    bool use_fft = false;
    int nf_tiles = 0, mel_tiles = 0, kc4 = 0;
    std::vector<float> basis;      // [n_mel][fft/2+1]
};

namespace {

// librosa.filters.mel(sr, n_fft, n_mels, fmin, fmax, htk=False, norm=1) restated (the reference calls it at
// models/rnn_ctc.py:139-144; librosa itself is not available offline): Slaney mel scale -- linear below 1 kHz
// (200/3 Hz per mel), logarithmic above (step ln(6.4)/27) -- triangular filters, each scaled by 2/(f_hi - f_lo).
double hz_to_mel_slaney(double f) {
    const double f_sp = 200.0 / 3.0, min_log_hz = 1000.0, min_log_mel = min_log_hz / f_sp, logstep = std::log(6.4) / 27.0;
    return f >= min_log_hz ? min_log_mel + std::log(f / min_log_hz) / logstep : f / f_sp;
}
double mel_to_hz_slaney(double m) {
    const double f_sp = 200.0 / 3.0, min_log_hz = 1000.0, min_log_mel = min_log_hz / f_sp, logstep = std::log(6.4) / 27.0;
    return m >= min_log_mel ? min_log_hz * std::exp(logstep * (m - min_log_mel)) : f_sp * m;
}
std::vector<float> slaney_mel_basis(int sr, int n_fft, int n_mels, double fmin, double fmax) {
    const int nf = n_fft / 2 + 1;
    std::vector<double> mel_f(n_mels + 2);
    const double m_lo = hz_to_mel_slaney(fmin), m_hi = hz_to_mel_slaney(fmax);
    for (int i = 0; i < n_mels + 2; ++i) mel_f[i] = mel_to_hz_slaney(m_lo + (m_hi - m_lo) * i / (n_mels + 1));
    std::vector<float> w((size_t)n_mels * nf, 0.f);
    for (int i = 0; i < n_mels; ++i) {
        const double enorm = 2.0 / (mel_f[i + 2] - mel_f[i]);
        for (int k = 0; k < nf; ++k) {
            const double fk = (double)sr / 2.0 * k / (nf - 1);
            const double lower = (fk - mel_f[i]) / (mel_f[i + 1] - mel_f[i]);
            const double upper = (mel_f[i + 2] - fk) / (mel_f[i + 2] - mel_f[i + 1]);
            const double v = std::max(0.0, std::min(lower, upper));
            w[(size_t)i * nf + k] = (float)(v * enorm);
        }
    }
    return w;
}

bool config_ok(const kws_config* c, int* code) {
    if (!c) { *code = fail(KWS_ERR_INVALID_ARGUMENT, "config is null"); return false; }
    if (c->n_mel < 1 || c->n_mel > 1024) { *code = fail(KWS_ERR_INVALID_ARGUMENT, "n_mel=%d out of range [1,1024]", c->n_mel); return false; }
    if (c->num_layers < 1 || c->num_layers > 8) { *code = fail(KWS_ERR_INVALID_ARGUMENT, "num_layers=%d out of range [1,8]", c->num_layers); return false; }
    if (c->num_classes < 3 || c->num_classes > kws::kMaxClasses) { *code = fail(KWS_ERR_UNSUPPORTED, "num_classes=%d unsupported (3..8)", c->num_classes); return false; }
    if (c->hidden != 64 && c->hidden != 128 && c->hidden != 256) { *code = fail(KWS_ERR_UNSUPPORTED, "hidden=%d unsupported (64, 128, 256)", c->hidden); return false; }
    if (c->precision == KWS_INT8 && c->hidden != 128) {
        *code = fail(KWS_ERR_UNSUPPORTED, "int8 path needs hidden=128 (OctbitMatMul K=2*hidden must be a multiple of 64 and the kernel is built for 128); got %d", c->hidden);
        return false;
    }
    if (c->precision != KWS_FP32 && c->precision != KWS_BF16 && c->precision != KWS_INT8 && c->precision != KWS_F16X3) { *code = fail(KWS_ERR_INVALID_ARGUMENT, "unknown precision %d", c->precision); return false; }
    if (c->precision == KWS_F16X3 && !kws::gru_f16x3_supported(c->hidden, c->n_mel) && !kws::gru_f16x3_generic_supported(c->hidden, c->n_mel)) {
        *code = fail(KWS_ERR_UNSUPPORTED, "f16x3 path needs hidden=128 (register-resident kernels) or 256 (weights streamed from L2) and n_mel%%4==0, 4..64; "
                     "got hidden=%d n_mel=%d", c->hidden, c->n_mel);
        return false;
    }
    if (c->precision == KWS_BF16 && !kws::gru_bf16_supported(c->hidden, c->n_mel, c->num_layers)) {
        *code = fail(KWS_ERR_UNSUPPORTED, "bf16 path needs hidden=128, num_layers<=2, n_mel%%4==0 and <=64; got hidden=%d layers=%d n_mel=%d",
                     c->hidden, c->num_layers, c->n_mel);
        return false;
    }
    return true;
}

size_t weights_floats(const kws_config* c) {
    size_t n = 0;
    int in = c->n_mel;
    const size_t H = c->hidden;
    for (int l = 0; l < c->num_layers; ++l) {
        n += (size_t)(in + H) * 3 * H + 3 * H;
        in = c->hidden;
    }
    return n + H * c->num_classes + c->num_classes;
}

// K-index permutation of the "xl" layout: chunk kc, lane group g -> source row
inline int kmap_grouped(int kc, int g) { return 16 * (kc / 4) + 4 * g + (kc % 4); }
inline int kmap_interleaved(int kc, int g) { return 4 * kc + g; }

// gate q of canonical layer weights: q=0 r (Wg[:, :H]), q=1 u (Wg[:, H:]), q=2 c (Wc)
inline float wq(const float* Wg, const float* Wc, int H, int q, int row, int unit) {
    return q == 2 ? Wc[(size_t)row * H + unit] : Wg[(size_t)row * 2 * H + q * H + unit];
}

// fp32 -> bf16 bits, round to nearest even (what v_cvt_pk_bf16_f32 does)
inline uint16_t bf16_rne(float x) {
    uint32_t u;
    memcpy(&u, &x, 4);
    if ((u & 0x7fffffffu) > 0x7f800000u) return (uint16_t)((u >> 16) | 0x40);   // NaN
    u += 0x7fffu + ((u >> 16) & 1u);
    return (uint16_t)(u >> 16);
}
// fp32 -> fp16 bits, round to nearest even (what v_cvt_f16_f32 does); the host compiler is clang: _Float16 is native
inline uint16_t f16_rne(float x) {
    const _Float16 h = static_cast<_Float16>(x);
    uint16_t u;
    memcpy(&u, &h, 2);
    return u;
}
inline float f16_value(uint16_t u) {
    _Float16 h;
    memcpy(&h, &u, 2);
    return static_cast<float>(h);
}
// v = hi + 2^-11 lo (gru_f16x3.hip): the two fp16 pieces of a weight
inline void f16_split(float v, uint16_t* hi, uint16_t* lo) {
    *hi = f16_rne(v);
    *lo = f16_rne((v - f16_value(*hi)) * 2048.0f);
#ifdef KWS_EXP_F16_WLO_ZERO      // experiment builds only (tools/build_variant.sh wlo0 -DKWS_EXP_F16_WLO_ZERO): single-piece fp16 WEIGHTS in the
    *lo = 0;                     // f16x3 kernels -- the hardware check of the rounding model behind the "f16x1" decision
#endif
}
// ... and with the lo piece at its own magnitude: v = hi + lo.  Below 2^-14 the piece is an fp16 subnormal (absolute precision
// 2^-25): the value keeps max(2^-23 |v|, 2^-25) -- fp32's own rounding down to |v| = 1/4, a 3e-8 absolute floor below that
inline void f16_split_unscaled(float v, uint16_t* hi, uint16_t* lo) {
    *hi = f16_rne(v);
    *lo = f16_rne(v - f16_value(*hi));
#ifdef KWS_EXP_F16_WLO_ZERO
    *lo = 0;
#endif
}
// unit of a hidden vector addressed by (chunk m, lane group g, element j) in the bf16 exchange layout
inline int bf16_unit(int m, int g, int j) { return 32 * m + (j < 4 ? 4 * g + j : 16 + 4 * g + (j - 4)); }

}  // namespace

extern "C" {

// Compiler provenance is part of the version string: the fp32 resident kernels rely on hand-placed hazard fences around
// inline-asm MFMAs and gru_bf16.hip on an internal LLVM option (csrc/Makefile), so "which hipcc built this" is the first
// thing to know when kws_selftest fails on a deployment.
const char* kws_version(void) {
    static const std::string v = [] {
        char buf[384];
        snprintf(buf, sizeof(buf), "kws_amd 0.6 (gfx950; HIP %d.%d.%d; %s; bf16 mfma-vgpr-form=%d; f16x3 mfma-vgpr-form=%d" KWS_VARIANT_TAG ")", HIP_VERSION_MAJOR,
                 HIP_VERSION_MINOR, HIP_VERSION_PATCH, __VERSION__, kws::gru_bf16_vgpr_form() ? 1 : 0, kws::gru_f16x3_vgpr_form() ? 1 : 0);
        return std::string(buf);
    }();
    return v.c_str();
}
const char* kws_last_error(void) { return g_last_error.c_str(); }
size_t kws_sizeof_config(void) { return sizeof(kws_config); }
size_t kws_sizeof_frontend_config(void) { return sizeof(kws_frontend_config); }

size_t kws_weights_nbytes(const kws_config* cfg) {
    int code;
    if (!config_ok(cfg, &code)) return 0;
    return weights_floats(cfg) * sizeof(float);
}

int kws_create(const kws_config* cfg, const void* weights_blob, size_t nbytes, kws_handle* out) {
    int code;
    if (!out) return fail(KWS_ERR_INVALID_ARGUMENT, "out handle pointer is null");
    *out = nullptr;
    if (!config_ok(cfg, &code)) return code;
    if (!weights_blob) return fail(KWS_ERR_INVALID_ARGUMENT, "weights_blob is null");
    const size_t need = weights_floats(cfg) * sizeof(float);
    if (nbytes != need)
        return fail(KWS_ERR_INVALID_ARGUMENT, "weights_blob has %zu bytes, config needs %zu", nbytes, need);
    int ndev = 0;
    if (hipGetDeviceCount(&ndev) != hipSuccess || ndev == 0)
        return fail(KWS_ERR_NO_DEVICE, "no HIP device visible");

    kws_model* m = new (std::nothrow) kws_model();
    if (!m) return fail(KWS_ERR_OUT_OF_MEMORY, "host allocation failed");
    m->cfg = *cfg;
    {
        const hipError_t e = hipGetDevice(&m->device);
        if (e != hipSuccess) { delete m; return hip_fail(e, "hipGetDevice"); }
    }
    {
        hipDeviceProp_t prop;
        if (hipGetDeviceProperties(&prop, m->device) == hipSuccess) m->num_cus = prop.multiProcessorCount;
    }
    const int H = cfg->hidden, NT = H / 16, KCH = H / 4, C = cfg->num_classes;

    std::vector<float> host;
    auto reserve = [&](size_t n) { size_t off = host.size(); host.resize(off + ((n + 3) & ~size_t(3)), 0.f); return off; };
    const float* p = static_cast<const float*>(weights_blob);
    int in = cfg->n_mel;
    for (int l = 0; l < cfg->num_layers; ++l) {
        const bool first = l == 0;
        const float* Wg = p;
        const float* bg = Wg + (size_t)(in + H) * 2 * H;
        const float* Wc = bg + 2 * H;
        const float* bc = Wc + (size_t)(in + H) * H;
        LayerDev L;
        L.in_dim = in;
        L.resident_ok = kws::gru_resident_supported(H, in, first);
        L.kcx_res = kws::gru_resident_kcx(in, first);
        L.kcx_gen = 4 * ((in + 15) / 16);
        // biases [3][H]
        L.bias = reserve(3 * (size_t)H);
        for (int j = 0; j < 2 * H; ++j) host[L.bias + j] = bg[j];
        for (int j = 0; j < H; ++j) host[L.bias + 2 * H + j] = bc[j];
        // h-part, group-of-4 fragments [NT][3][NT][64][4] (one dwordx4 = four k-chunks; both kernel families)
        L.wh_gen = reserve((size_t)NT * 3 * KCH * 64);
        for (int n = 0; n < NT; ++n)
            for (int q = 0; q < 3; ++q)
                for (int kc = 0; kc < KCH; ++kc)
                    for (int lane = 0; lane < 64; ++lane) {
                        const int g = lane >> 4, i = lane & 15;
                        const float v = wq(Wg, Wc, H, q, in + kmap_grouped(kc, g), n * 16 + i);
                        host[L.wh_gen + ((((size_t)(n * 3 + q) * NT + kc / 4) * 64 + lane) * 4 + kc % 4)] = v;
                    }
        // x-part
        L.wx_res = reserve((size_t)NT * 3 * L.kcx_res * 64);
        L.wx_gen = reserve((size_t)NT * 3 * L.kcx_gen * 64);
        for (int n = 0; n < NT; ++n)
            for (int q = 0; q < 3; ++q)
                for (int lane = 0; lane < 64; ++lane) {
                    const int g = lane >> 4, i = lane & 15;
                    for (int kc = 0; kc < L.kcx_res; ++kc) {
                        const int row = first ? kmap_interleaved(kc, g) : kmap_grouped(kc, g);
                        host[L.wx_res + (((size_t)(n * 3 + q) * L.kcx_res + kc) * 64 + lane)] =
                            row < in ? wq(Wg, Wc, H, q, row, n * 16 + i) : 0.f;
                    }
                    for (int kc = 0; kc < L.kcx_gen; ++kc) {
                        const int row = kmap_grouped(kc, g);
                        host[L.wx_gen + ((((size_t)(n * 3 + q) * (L.kcx_gen / 4) + kc / 4) * 64 + lane) * 4 + kc % 4)] =
                            row < in ? wq(Wg, Wc, H, q, row, n * 16 + i) : 0.f;
                    }
                }
        m->layers.push_back(L);
        p = bc + H;
        in = H;
    }
    // dense: Wfc [H,C] -> A fragments of Wfc^T padded to 16 rows, [KCH][64]; bias padded to 16
    const float* Wfc = p;
    const float* bfc = Wfc + (size_t)H * C;
    m->wfc_off = reserve((size_t)KCH * 64);
    for (int kc = 0; kc < KCH; ++kc)
        for (int lane = 0; lane < 64; ++lane) {
            const int g = lane >> 4, i = lane & 15;
            host[m->wfc_off + (size_t)kc * 64 + lane] = i < C ? Wfc[(size_t)kmap_grouped(kc, g) * C + i] : 0.f;
        }
    m->bfc_off = reserve(16);
    for (int i = 0; i < C; ++i) host[m->bfc_off + i] = bfc[i];

    if (cfg->precision == KWS_BF16) {
        // A operands of v_mfma_f32_16x16x32_bf16: lane (g,i) holds 8 bf16 = W[row(c,g,j)][16n+i], j = 0..7
        const float* q = static_cast<const float*>(weights_blob);
        int in_l = cfg->n_mel;
        m->bf_kx0 = (cfg->n_mel + 31) / 32;
        for (int l = 0; l < cfg->num_layers; ++l) {
            const float* Wg = q;
            const float* Wc = Wg + (size_t)(in_l + H) * 2 * H + 2 * H;
            const int kx = l == 0 ? m->bf_kx0 : 4, kc = kx + 4;
            m->bf_w[l] = reserve((size_t)8 * 3 * kc * 64 * 4);
            uint16_t* dst = reinterpret_cast<uint16_t*>(&host[m->bf_w[l]]);
            for (int n = 0; n < 8; ++n)
                for (int gq = 0; gq < 3; ++gq)
                    for (int c = 0; c < kc; ++c)
                        for (int lane = 0; lane < 64; ++lane)
                            for (int j = 0; j < 8; ++j) {
                                const int g = lane >> 4, i = lane & 15;
                                int row;
                                bool ok = true;
                                if (c < kx) {
                                    if (l == 0) { row = 32 * c + 8 * g + j; ok = row < in_l; }
                                    else row = bf16_unit(c, g, j);
                                } else {
                                    row = in_l + bf16_unit(c - kx, g, j);
                                }
                                const float v = ok ? wq(Wg, Wc, H, gq, row, n * 16 + i) : 0.f;
                                dst[((((size_t)(n * 3 + gq) * kc + c) * 64 + lane) * 8) + j] = bf16_rne(v);
                            }
            q = Wc + (size_t)(in_l + H) * H + H;
            in_l = H;
        }
        m->bf_wfc = reserve((size_t)4 * 64 * 4);
        uint16_t* dst = reinterpret_cast<uint16_t*>(&host[m->bf_wfc]);
        for (int c = 0; c < 4; ++c)
            for (int lane = 0; lane < 64; ++lane)
                for (int j = 0; j < 8; ++j) {
                    const int g = lane >> 4, i = lane & 15;
                    dst[((size_t)c * 64 + lane) * 8 + j] = bf16_rne(i < C ? Wfc[(size_t)bf16_unit(c, g, j) * C + i] : 0.f);
                }
    }

    if (cfg->precision == KWS_F16X3) {
        // A operands of v_mfma_f32_16x16x32_f16, split: [8 tiles][3 gates][kc chunks][hi|lo][64 lanes] x 8 halves; lane (g,i)
        // holds W[row(c,g,j)][16n+i], j = 0..7, rows in gru_bf16's K permutation.  The first layer's x-part is scaled by 2^8
        // (the kernel feeds mel * 2^-8: both exact) so that mel magnitudes far beyond fp16's 65504 stay representable.
        const float* q = static_cast<const float*>(weights_blob);
        {
            // only the matrices become fp16 operands (x-part scaled by 256, candidate by 2 log2 e: 256 * 2.886 * 64 < 65504); the
            // biases and bfc stay fp32 in the kernels and may be of any size
            const float* base = q;
            const float* wq_ = q;
            int in_c = cfg->n_mel;
            auto range_ok = [&](const float* w, size_t n, const char* what, int layer) -> const float* {
                for (size_t i = 0; i < n; ++i)
                    if (!(std::fabs(w[i]) < 64.0f)) {
                        fail(KWS_ERR_UNSUPPORTED, "f16x3 path: %s weight of layer %d (blob index %zu) = %g is outside (-64, 64) (fp16 operands; "
                             "x-part scaled by 256, candidate by 2 log2 e)", what, layer, (size_t)(w + i - base), (double)w[i]);
                        return nullptr;
                    }
                return w + n;
            };
            for (int l = 0; l < cfg->num_layers && wq_; ++l) {
                wq_ = range_ok(wq_, (size_t)(in_c + H) * 2 * H, "gate", l);
                if (wq_) wq_ = range_ok(wq_ + 2 * H, (size_t)(in_c + H) * H, "candidate", l);
                if (wq_) wq_ += H;
                in_c = H;
            }
            if (wq_) wq_ = range_ok(wq_, (size_t)H * C, "projection", cfg->num_layers);
            if (!wq_) { delete m; return KWS_ERR_UNSUPPORTED; }
        }
        int in_l = cfg->n_mel;
        // hidden = 128: the register-resident kernels (gru_f16x3.hip); otherwise the streaming ones (gru_f16x3_generic.hip), whose
        // phases come in row pairs: the first layer's x chunks are padded to an even count (zero operands)
        m->f16_generic = !kws::gru_f16x3_supported(H, cfg->n_mel);
        m->f16_kx0 = m->f16_generic ? 2 * ((cfg->n_mel + 63) / 64) : (cfg->n_mel + 31) / 32;
        const int HCh = H / 32;                    // 32-wide chunks of a hidden vector
        for (int l = 0; l < cfg->num_layers; ++l) {
            const float* Wg = q;
            const float* Wc = Wg + (size_t)(in_l + H) * 2 * H + 2 * H;
            const int kx = l == 0 ? m->f16_kx0 : HCh, kc = kx + HCh;
            m->f16_w.push_back(reserve((size_t)NT * 3 * kc * 2 * 64 * 4));
            uint16_t* dst = reinterpret_cast<uint16_t*>(&host[m->f16_w[l]]);
            for (int n = 0; n < NT; ++n)
                for (int gq = 0; gq < 3; ++gq)
                    for (int c = 0; c < kc; ++c)
                        for (int lane = 0; lane < 64; ++lane)
                            for (int j = 0; j < 8; ++j) {
                                const int g = lane >> 4, i = lane & 15;
                                int row;
                                bool ok = true;
                                float scale = 1.f;
                                if (c < kx) {
                                    if (l == 0) { row = 32 * c + 8 * g + j; ok = row < in_l; scale = 256.f; }
                                    else row = bf16_unit(c, g, j);
                                } else {
                                    row = in_l + bf16_unit(c - kx, g, j);
                                }
                                // the exponent scale of the gate's activation rides in the weights: sigmoid(a) = 1 / (1 + exp2(-a log2 e)),
                                // tanh(a) = 1 - 2 / (1 + exp2(2 a log2 e)) -- the kernel applies exp2 to the pre-activation as it is
                                const float act = gq == 2 ? 2.0f * 1.4426950408889634f : -1.4426950408889634f;
                                const float v = ok ? scale * (act * wq(Wg, Wc, H, gq, row, n * 16 + i)) : 0.f;
                                uint16_t hi, lo;
                                // the register-resident kernels (gru_f16x3.hip) keep ONE accumulator per product: lo pieces unscaled,
                                // except the first layer's x-part, which meets the mel frame's 2^11-scaled lo piece; the streaming
                                // kernels (gru_f16x3_generic.hip) keep main / lo accumulators and scaled lo pieces throughout
                                if (m->f16_generic || (l == 0 && c < kx)) f16_split(v, &hi, &lo);
                                else f16_split_unscaled(v, &hi, &lo);
                                const size_t base = (((size_t)(n * 3 + gq) * kc + c) * 2) * 64;
                                dst[(base + lane) * 8 + j] = hi;
                                dst[(base + 64 + lane) * 8 + j] = lo;
                            }
            q = Wc + (size_t)(in_l + H) * H + H;
            in_l = H;
        }
        m->f16_wfc = reserve((size_t)HCh * 2 * 64 * 4);
        uint16_t* dst = reinterpret_cast<uint16_t*>(&host[m->f16_wfc]);
        for (int c = 0; c < HCh; ++c)
            for (int lane = 0; lane < 64; ++lane)
                for (int j = 0; j < 8; ++j) {
                    const int g = lane >> 4, i = lane & 15;
                    uint16_t hi, lo;
                    const float v = i < C ? Wfc[(size_t)bf16_unit(c, g, j) * C + i] : 0.f;
                    if (m->f16_generic) f16_split(v, &hi, &lo);
                    else f16_split_unscaled(v, &hi, &lo);
                    dst[(((size_t)c * 2 + 0) * 64 + lane) * 8 + j] = hi;
                    dst[(((size_t)c * 2 + 1) * 64 + lane) * 8 + j] = lo;
                }
    }

    if (cfg->precision == KWS_INT8) {
        // octbit/octbit_graph.py:218-225: every MatMul outside cell_0 is quantised -> layers >= 1 and the projection
        auto pack16 = [](int8_t lo, int8_t hi) { return (uint32_t)(uint16_t)(int16_t)lo | ((uint32_t)(uint16_t)(int16_t)hi << 16); };
        const float* q = static_cast<const float*>(weights_blob);
        int in_l = cfg->n_mel;
        m->oct.resize(cfg->num_layers);
        for (int l = 0; l < cfg->num_layers; ++l) {
            const int K = in_l + H;
            const float* Wg = q;
            const float* Wc = Wg + (size_t)K * 2 * H + 2 * H;
            if (l >= 1) {
                kws_model::OctLayer& O = m->oct[l];
                O.quantised = true;
                std::vector<int8_t> gq((size_t)2 * H * K), cq((size_t)H * K);
                std::vector<float> gb(2 * H), cb(H);
                int rc = kws_octbit_quantize(Wg, K, 2 * H, gq.data(), &O.scale_g, gb.data());
                if (rc == KWS_OK) rc = kws_octbit_quantize(Wc, K, H, cq.data(), &O.scale_c, cb.data());
                if (rc != KWS_OK) { delete m; return rc; }
                O.b127 = reserve(3 * (size_t)H);
                for (int j = 0; j < 2 * H; ++j) host[O.b127 + j] = gb[j];
                for (int j = 0; j < H; ++j) host[O.b127 + 2 * H + j] = cb[j];
                // gates: wave (kq, ug) -> [unit-in-lane 2][32 = 16 couples x (even, odd)][64 lanes]
                O.wg = reserve((size_t)4 * 2 * 64 * 64);
                {
                    uint32_t* dst = reinterpret_cast<uint32_t*>(&host[O.wg]);
                    for (int kq = 0; kq < 4; ++kq)
                        for (int ug = 0; ug < 2; ++ug)
                            for (int c = 0; c < 64; ++c)
                                for (int lane = 0; lane < 64; ++lane) {
                                    const int ul = c >> 5, c2 = c & 31;
                                    const int n = 128 * ug + 64 * ul + lane, k0 = 64 * kq + 4 * (c2 / 2) + (c2 & 1);
                                    dst[((size_t)(kq * 2 + ug) * 64 + c) * 64 + lane] =
                                        pack16(gq[(size_t)n * K + k0], gq[(size_t)n * K + k0 + 2]);
                                }
                }
                // candidate: wave k8 -> [unit-in-lane 2][16 = 8 couples x (even, odd)][64 lanes]
                O.wc = reserve((size_t)8 * 32 * 64);
                {
                    uint32_t* dst = reinterpret_cast<uint32_t*>(&host[O.wc]);
                    for (int k8 = 0; k8 < 8; ++k8)
                        for (int c = 0; c < 32; ++c)
                            for (int lane = 0; lane < 64; ++lane) {
                                const int ul = c >> 4, c2 = c & 15;
                                const int n = 64 * ul + lane, k0 = 32 * k8 + 4 * (c2 / 2) + (c2 & 1);
                                dst[((size_t)k8 * 32 + c) * 64 + lane] =
                                    pack16(cq[(size_t)n * K + k0], cq[(size_t)n * K + k0 + 2]);
                            }
                }
            }
            q = Wc + (size_t)K * H + H;
            in_l = H;
        }
        std::vector<int8_t> fq((size_t)C * H);
        std::vector<float> fb(C);
        int rc = kws_octbit_quantize(Wfc, H, C, fq.data(), &m->oct_scale_fc, fb.data());
        if (rc != KWS_OK) { delete m; return rc; }
        m->oct_b127fc = reserve(kws::kMaxClasses);
        for (int c = 0; c < C; ++c) host[m->oct_b127fc + c] = fb[c];
        m->oct_wfc = reserve((size_t)8 * 4 * 2 * kws::kMaxClasses);
        uint32_t* dst = reinterpret_cast<uint32_t*>(&host[m->oct_wfc]);
        for (int n = 0; n < 8; ++n)
            for (int g = 0; g < 4; ++g)
                for (int c = 0; c < kws::kMaxClasses; ++c)
                    for (int e = 0; e < 2; ++e) {
                        const int k0 = 16 * n + 4 * g + e;
                        dst[((size_t)(n * 4 + g) * kws::kMaxClasses + c) * 2 + e] =
                            c < C ? pack16(fq[(size_t)c * H + k0], fq[(size_t)c * H + k0 + 2]) : 0u;
                    }
    }

    hipError_t e = hipMalloc(reinterpret_cast<void**>(&m->d_weights), host.size() * sizeof(float));
    if (e != hipSuccess) { delete m; return hip_fail(e, "hipMalloc(weights)"); }
    e = hipMemcpy(m->d_weights, host.data(), host.size() * sizeof(float), hipMemcpyHostToDevice);
    if (e != hipSuccess) { hipFree(m->d_weights); delete m; return hip_fail(e, "hipMemcpy(weights)"); }
    // A pageable-host hipMemcpy may return once the data is staged; make sure the DMA has landed before
    // any stream can launch a kernel that reads the fragments (observed: the tail of the upload missing
    // in the first launch after create).
    e = hipDeviceSynchronize();
    if (e != hipSuccess) { hipFree(m->d_weights); delete m; return hip_fail(e, "hipDeviceSynchronize(weights)"); }
    e = hipEventCreateWithFlags(&m->last_done, hipEventDisableTiming);
    if (e != hipSuccess) { hipFree(m->d_weights); delete m; return hip_fail(e, "hipEventCreate"); }
    m->ms_sum.assign(cfg->num_layers, 0.f);
    m->launches.assign(cfg->num_layers, 0);
    live_register(m);
    // KWS_SELFTEST=1: every kws_create first proves the kernels this handle will use against the library's own known
    // answers (kws_selftest below) -- a few milliseconds; meant for deployments on a ROCm other than the validated one
    static const bool selftest_env = [] { const char* e = getenv("KWS_SELFTEST"); return e && e[0] == '1'; }();
    if (selftest_env && !g_in_selftest) {
        const int rc = kws_selftest(m);
        if (rc != KWS_OK) {
            const std::string keep = g_last_error;
            kws_destroy(m);
            g_last_error = keep;
            return rc;
        }
    }
    *out = m;
    return KWS_OK;
}

int kws_destroy(kws_handle h) {
    if (!h) return KWS_OK;
    live_unregister(h);
    hipDeviceSynchronize();
    for (auto& pd : h->pending) { hipEventDestroy(pd.a); hipEventDestroy(pd.b); }
    for (auto ev : h->event_pool) hipEventDestroy(ev);
    if (h->d_weights) hipFree(h->d_weights);
    if (h->arena.base) hipFree(h->arena.base);
    if (h->arena_fine.base) hipFree(h->arena_fine.base);
    if (h->stage.base) hipFree(h->stage.base);
    if (h->last_done) hipEventDestroy(h->last_done);
    if (h->pipe_ready) hipFree(h->pipe_ready);
    // a layer-pipelined step that timed out and was never followed by another call is still reported, once
    const bool pipe_failed = h->pipe_error_host && *reinterpret_cast<volatile int*>(h->pipe_error_host) != 0;
    if (h->pipe_error_host) hipHostFree(h->pipe_error_host);
    for (auto ev : h->ovl_events) hipEventDestroy(ev);
    for (auto ev : h->ovl_tail) if (ev) hipEventDestroy(ev);
    for (auto sx : h->lane_stream) if (sx) hipStreamDestroy(sx);
    if (h->oct_aq) hipFree(h->oct_aq);
    if (h->oct_range) hipFree(h->oct_range);
    if (h->oct_prev) hipFree(h->oct_prev);
    delete h;
    // The handle is gone whatever happened before: always KWS_OK (a caller that read a failure as "still alive" would free it
    // twice).  A pipelined step that timed out and was never followed by another call is left in kws_last_error().
    if (pipe_failed)
        fail(KWS_OK, "kws_destroy: the last layer-pipelined step of this handle had timed out waiting for the layer below; its results "
                     "were invalid (kws_poll_error before kws_destroy reports this as a status)");
    return KWS_OK;
}

int kws_set_kernel(kws_handle h, int kind) {
    if (!h) return fail(KWS_ERR_INVALID_ARGUMENT, "handle is null");
    if (kind != KWS_KERNEL_AUTO && kind != KWS_KERNEL_GENERIC && kind != KWS_KERNEL_RESIDENT)
        return fail(KWS_ERR_INVALID_ARGUMENT, "unknown kernel kind %d", kind);
    if (kind == KWS_KERNEL_RESIDENT)
        for (const auto& L : h->layers)
            if (!L.resident_ok)
                return fail(KWS_ERR_UNSUPPORTED, "resident kernels need hidden=128 and n_mel in {32, 40, 48, 60, 64}; got hidden=%d n_mel=%d",
                            h->cfg.hidden, h->cfg.n_mel);
    h->kernel_kind = kind;
    return KWS_OK;
}

// The layer-pipelined launch needs all L x groups workgroups resident at once (one per CU) and the streaming kernel
// on every layer; it pays when the layers would otherwise leave CUs idle.
static bool pipeline_eligible(kws_handle h, int B) {
    // AUTO keeps the resident kernels where they exist even when this launch would be faster (H=128, L=2: +10 % at
    // B <= 2048; L=4, B=1024: 2.2x; select it with KWS_KERNEL_GENERIC): the two kernel families round differently in
    // the last bit, and a stream's result must not depend on how many neighbours it is batched or sharded with.
    if (h->pipe_disabled) return false;
    const bool f16_streaming = h->cfg.precision == KWS_F16X3 && h->f16_generic;      // gru_stack_f16x3_pipelined
    if ((h->cfg.precision != KWS_FP32 && !f16_streaming) || h->cfg.num_layers < 2 || h->kernel_kind == KWS_KERNEL_RESIDENT) return false;
    for (const auto& L : h->layers)
        if (h->kernel_kind == KWS_KERNEL_AUTO && L.resident_ok) return false;
    const long long groups = (B + kws::kStreamsPerGroup - 1) / kws::kStreamsPerGroup;
    return h->num_cus > 0 && groups * h->cfg.num_layers <= h->num_cus;
}

// ---- scratch: what a call of shape (B, T) needs, and the arena it is carved from ------------------------------------
#ifndef KWS_OVERLAP_MIN_T
#define KWS_OVERLAP_MIN_T 64
#endif
static bool overlap_shape_ok(kws_handle h, int B, int T) {      // step_overlapped, leaving the profiling switch aside
    const kws_config& c = h->cfg;
    if (c.precision != KWS_FP32 || c.num_layers < 2 || c.num_layers > 5) return false;
    if (pipeline_eligible(h, B)) return false;            // the streaming kernel has its own in-kernel pipeline
    const long long groups = (B + kws::kStreamsPerGroup - 1) / kws::kStreamsPerGroup;
    return h->num_cus > 0 && groups * c.num_layers <= h->num_cus && T >= KWS_OVERLAP_MIN_T;
}
static bool overlap_eligible(kws_handle h, int B, int T) { return !h->profiling && overlap_shape_ok(h, B, T); }
static void overlap_blocks(int T, int* nb_out, int* tb_out) {
    int nb = T / 32 < 8 ? T / 32 : 8;              // more blocks: less fill/drain, more launch prologues (8: +6 % over 4)
    if (nb < 2) nb = 2;
    const int Tb = ((T + nb - 1) / nb + 15) & ~15;        // multiple of the epilogue ring
    *nb_out = (T + Tb - 1) / Tb;
    *tb_out = Tb;
}

struct SeamLayout { int nbuf; size_t bytes_each; bool fine; };
constexpr int kNoFineGrainedMemory = 1;      // carve_seams: internal, never returned through the ABI
enum { kLayoutSequential = 0, kLayoutOverlapped = 1 };
static SeamLayout seam_layout(kws_handle h, int B, int T, int which) {
    const kws_config& c = h->cfg;
    const size_t groups = (size_t)(B + kws::kStreamsPerGroup - 1) / kws::kStreamsPerGroup;
    const size_t frame_bytes = groups * (size_t)c.hidden * 16 * sizeof(float);
    SeamLayout s = {0, 0, false};
    if (c.precision == KWS_BF16 || T <= 0 || B <= 0) return s;                    // the bf16 stack has no seam
    const bool int8 = c.precision == KWS_INT8;
    if (c.num_layers < 2 && !int8) return s;
    if (which == kLayoutOverlapped) {
        int nb, Tb;
        overlap_blocks(T, &nb, &Tb);
        s.nbuf = 2 * (c.num_layers - 1); s.bytes_each = frame_bytes * Tb;
        return s;
    }
    if (pipeline_eligible(h, B)) { s.nbuf = c.num_layers - 1; s.bytes_each = frame_bytes * T; s.fine = true; return s; }
    s.nbuf = (c.num_layers > 2 || int8) ? 2 : 1;         // sequential launches ping-pong two buffers
    s.bytes_each = frame_bytes * T;
    return s;
}

// Grows the arena if this layout does not fit (the only place kws_step can synchronise: the old block may still be in
// use) and points h->scratch[] at the call's buffers.
static int carve_seams(kws_handle h, const SeamLayout& want) {
    SeamLayout s = want;
    s.bytes_each = (s.bytes_each + 255) & ~size_t(255);
    kws_model::Arena& A = s.fine ? h->arena_fine : h->arena;
    const size_t need = (size_t)s.nbuf * s.bytes_each;
    if (need > A.bytes) {
        KWS_HIP(hipDeviceSynchronize());
        if (A.base) { hipFree(A.base); A.base = nullptr; A.bytes = 0; }
        hipError_t e;
        // pipelined seams are read by another XCD while the kernel runs: fine-grained (uncached, coherent) memory
        if (s.fine) e = hipExtMallocWithFlags(reinterpret_cast<void**>(&A.base), need, hipDeviceMallocFinegrained);
        else e = hipMalloc(reinterpret_cast<void**>(&A.base), need);
        if (e != hipSuccess) {
            A.base = nullptr;
            (void)hipGetLastError();
            return s.fine ? kNoFineGrainedMemory : hip_fail(e, "hipMalloc(scratch)");
        }
        A.bytes = need;
        ++h->scratch_allocs;
    }
    for (int i = 0; i < 8; ++i)
        h->scratch[i] = i < s.nbuf ? reinterpret_cast<float4*>(A.base + (size_t)i * s.bytes_each) : nullptr;
    h->nscratch = s.nbuf;
    return KWS_OK;
}

// Everything besides the seams that depends on the batch size: int8 exchange buffers, the pipelined launch's counters.
static int ensure_side_buffers(kws_handle h, int B) {
    const size_t groups = (size_t)(B + kws::kStreamsPerGroup - 1) / kws::kStreamsPerGroup;
    if (h->cfg.precision == KWS_INT8 && groups > h->oct_groups) {
        KWS_HIP(hipDeviceSynchronize());
        if (h->oct_aq) hipFree(h->oct_aq);
        if (h->oct_range) hipFree(h->oct_range);
        if (h->oct_prev) hipFree(h->oct_prev);
        h->oct_aq = nullptr; h->oct_range = nullptr; h->oct_prev = nullptr; h->oct_groups = 0;
        KWS_HIP(hipMalloc(reinterpret_cast<void**>(&h->oct_aq), groups * 2 * 16 * 128 * sizeof(uint32_t)));
        KWS_HIP(hipMalloc(reinterpret_cast<void**>(&h->oct_range), groups * 16 * sizeof(float2)));
        KWS_HIP(hipMalloc(reinterpret_cast<void**>(&h->oct_prev), groups * 16 * sizeof(int32_t)));
        h->oct_groups = groups;
        ++h->scratch_allocs;
    }
    if (pipeline_eligible(h, B)) {
        if (!h->pipe_error_host) {
            KWS_HIP(hipHostMalloc(reinterpret_cast<void**>(&h->pipe_error_host), sizeof(int), hipHostMallocMapped));
            *h->pipe_error_host = 0;
            KWS_HIP(hipHostGetDevicePointer(reinterpret_cast<void**>(&h->pipe_error_dev), h->pipe_error_host, 0));
        }
        if (groups > h->pipe_groups) {
            KWS_HIP(hipDeviceSynchronize());
            if (h->pipe_ready) hipFree(h->pipe_ready);
            h->pipe_ready = nullptr; h->pipe_groups = 0;
            if (hipExtMallocWithFlags(reinterpret_cast<void**>(&h->pipe_ready), (size_t)h->cfg.num_layers * groups * sizeof(int),
                                      hipDeviceMallocFinegrained) != hipSuccess) {
                (void)hipGetLastError();
                h->pipe_ready = nullptr;
                h->pipe_disabled = true;         // no fine-grained device memory here: layer-by-layer launches from now on
                return KWS_OK;
            }
            h->pipe_groups = groups;
            ++h->scratch_allocs;
        }
    }
    return KWS_OK;
}

// Seams of the sequential / pipelined layout for a call of shape (B, T).
static int ensure_scratch(kws_handle h, int B, int T) {
    int rc = ensure_side_buffers(h, B);
    if (rc != KWS_OK) return rc;
    rc = carve_seams(h, seam_layout(h, B, T, kLayoutSequential));
    if (rc == kNoFineGrainedMemory) {    // fine-grained memory unavailable: give the pipelined launch up for this handle
        h->pipe_disabled = true;
        rc = carve_seams(h, seam_layout(h, B, T, kLayoutSequential));
    }
    return rc;
}

int kws_reserve(kws_handle h, int B, int T) {
    if (!h) return fail(KWS_ERR_INVALID_ARGUMENT, "handle is null");
    BusyGuard busy(h->in_call);
    if (!busy.owned) return fail(KWS_ERR_BUSY, "kws_reserve: another host thread is inside a call on this handle");
    if (B < 0 || T < 0) return fail(KWS_ERR_INVALID_ARGUMENT, "negative B=%d or T=%d", B, T);
    // whichever launch layout kws_step picks for (B, T) -- it depends on kws_set_profiling too -- fits afterwards
    int rc = ensure_scratch(h, B, T);
    if (rc != KWS_OK) return rc;
    if (overlap_shape_ok(h, B, T)) rc = carve_seams(h, seam_layout(h, B, T, kLayoutOverlapped));
    return rc;
}

int kws_scratch_stats(kws_handle h, size_t* bytes_reserved, int32_t* allocations) {
    if (!h) return fail(KWS_ERR_INVALID_ARGUMENT, "handle is null");
    if (bytes_reserved) *bytes_reserved = h->arena.bytes + h->arena_fine.bytes;
    if (allocations) *allocations = h->scratch_allocs;
    return KWS_OK;
}

// A layer-pipelined launch whose wait for the layer below timed out raises a mapped host flag; its results are invalid.
// The flag is looked at (without synchronising) on every kws_step, by kws_poll_error, after the event syncs of
// kws_kernel_times and in kws_destroy -- a step's own flag can only be seen once its kernel has run, so a caller that
// needs certainty synchronises the stream and asks kws_poll_error.
static int check_pipe_error(kws_handle h) {
    if (h->pipe_error_host && *reinterpret_cast<volatile int*>(h->pipe_error_host)) {
        *reinterpret_cast<volatile int*>(h->pipe_error_host) = 0;
        h->pipe_disabled = true;             // later steps launch layer by layer
        return fail(KWS_ERR_HIP, "a layer-pipelined launch timed out waiting for the layer below; the results of that step are invalid");
    }
    return KWS_OK;
}

int kws_poll_error(kws_handle h) {
    if (!h) return fail(KWS_ERR_INVALID_ARGUMENT, "handle is null");
    return check_pipe_error(h);
}

int kws_set_profiling(kws_handle h, int enable) {
    if (!h) return fail(KWS_ERR_INVALID_ARGUMENT, "handle is null");
    h->profiling = enable != 0;
    return KWS_OK;
}

int kws_kernel_times(kws_handle h, float* ms_sum, int32_t* launches, int reset) {
    if (!h) return fail(KWS_ERR_INVALID_ARGUMENT, "handle is null");
    BusyGuard busy(h->in_call);
    if (!busy.owned) return fail(KWS_ERR_BUSY, "kws_kernel_times: another host thread is inside a call on this handle");
    for (auto& pd : h->pending) {
        KWS_HIP(hipEventSynchronize(pd.b));
        float ms = 0.f;
        KWS_HIP(hipEventElapsedTime(&ms, pd.a, pd.b));
        h->ms_sum[pd.slot] += ms;
        h->launches[pd.slot] += 1;
        h->event_pool.push_back(pd.a);
        h->event_pool.push_back(pd.b);
    }
    h->pending.clear();
    for (int l = 0; l < h->cfg.num_layers; ++l) {
        if (ms_sum) ms_sum[l] = h->ms_sum[l];
        if (launches) launches[l] = h->launches[l];
    }
    if (reset) {
        std::fill(h->ms_sum.begin(), h->ms_sum.end(), 0.f);
        std::fill(h->launches.begin(), h->launches.end(), 0);
    }
    return check_pipe_error(h);
}

// Layers on separate HIP streams, time-blocked.  When L x groups workgroups fit the chip at once, the layers of a
// long call need not run one after another: the call is cut into time blocks, layer l works on block k while layer
// l-1 already works on block k+1 (its own stream, ordered by events; seam buffers double-buffered per block parity).
// The kernels are the ones a plain call uses -- a call on frames [t0, t1) with the state carried is bit-identical to
// the corresponding slice of one long call (tests/test_gpu_parity.py) -- so the result does not depend on whether
// this path was taken.  Wall time ~ (slowest layer) x (1 + 1/blocks) instead of the sum over layers.
static int step_overlapped(kws_handle h, const float* mel, const float* state_in, float* logits, float* softmax,
                           float* state_out, const int32_t* seq_len, const uint8_t* reset_mask, int8_t* tokens,
                           int32_t* prev_word, float decode2_thres, int B, int T, hipStream_t st) {
    const kws_config& c = h->cfg;
    const int H = c.hidden, L = c.num_layers, C = c.num_classes;
    int nb, Tb;
    overlap_blocks(T, &nb, &Tb);
    {
        const int rc = carve_seams(h, seam_layout(h, B, T, kLayoutOverlapped));
        if (rc != KWS_OK) return rc;
    }
    for (int l = 1; l < L; ++l)
        if (!h->lane_stream[l]) KWS_HIP(hipStreamCreateWithFlags(&h->lane_stream[l], hipStreamNonBlocking));
    while ((int)h->ovl_events.size() < L * nb + 1) {
        hipEvent_t ev;
        KWS_HIP(hipEventCreateWithFlags(&ev, hipEventDisableTiming));
        h->ovl_events.push_back(ev);
    }
    auto done = [&](int l, int k) { return h->ovl_events[1 + l * nb + k]; };
    KWS_HIP(hipEventRecord(h->ovl_events[0], st));                          // everything queued before this call
    for (int l = 1; l < L; ++l) KWS_HIP(hipStreamWaitEvent(h->lane_stream[l], h->ovl_events[0], 0));
    for (int k = 0; k < nb; ++k) {
        const int t0 = k * Tb, tk = (T - t0 < Tb) ? T - t0 : Tb;
        for (int l = 0; l < L; ++l) {
            const LayerDev& Ld = h->layers[l];
            const bool first = l == 0, last = l == L - 1;
            const bool resident = h->kernel_kind == KWS_KERNEL_RESIDENT || (h->kernel_kind == KWS_KERNEL_AUTO && Ld.resident_ok);
            hipStream_t sx = first ? st : h->lane_stream[l];
            if (!first) KWS_HIP(hipStreamWaitEvent(sx, done(l - 1, k), 0));                    // its input block
            if (!last && k >= 2) KWS_HIP(hipStreamWaitEvent(sx, done(l + 1, k - 2), 0));       // its output buffer is free again
            kws::GruLayerParams p;
            memset(&p, 0, sizeof(p));
            p.wx = h->d_weights + ((resident && first) ? Ld.wx_res : Ld.wx_gen);
            p.wh = h->d_weights + Ld.wh_gen;
            p.bias = h->d_weights + Ld.bias;
            p.wfc = h->d_weights + h->wfc_off;
            p.bfc = h->d_weights + h->bfc_off;
            p.x_mel = mel + (size_t)t0 * c.n_mel;
            p.x_prev = first ? nullptr : h->scratch[2 * (l - 1) + (k & 1)];
            p.h_out = last ? nullptr : h->scratch[2 * l + (k & 1)];
            p.state_in = (k == 0 ? state_in : state_out) + (size_t)l * B * H;
            p.state_out = state_out + (size_t)l * B * H;
            p.seq_len = seq_len;
            p.reset = k == 0 ? reset_mask : nullptr;
            p.logits = logits ? logits + (size_t)t0 * C : nullptr;
            p.softmax = softmax ? softmax + (size_t)t0 * C : nullptr;
            p.tokens = tokens ? tokens + t0 : nullptr;
            p.prev_word = prev_word;
            p.decode_thres = decode2_thres;
            p.value_clip = c.value_clip;
            p.use_relu = c.use_relu;
            p.B = B; p.T = tk; p.I = Ld.in_dim; p.C = C;
            p.t_stride = T; p.t_base = t0;
            p.KCX = resident ? Ld.kcx_res : Ld.kcx_gen;
            hipError_t e = resident ? kws::launch_gru_layer_resident(p, first, last, sx)
                                    : kws::launch_gru_layer_generic(p, H, first, last, sx);
            if (e != hipSuccess) return hip_fail(e, "launch (overlapped layers)");
            if (k == 0) h->launch_tag[l] = {(uint8_t)(resident ? kws_model::kResident : kws_model::kGeneric), (uint8_t)(resident ? p.KCX : H / 64), first, last, 0};
            KWS_HIP(hipEventRecord(done(l, k), sx));
        }
    }
    for (int l = 1; l < L; ++l) {
        KWS_HIP(hipStreamWaitEvent(st, done(l, nb - 1), 0));                                   // rejoin the caller's stream
        if (!h->ovl_tail[l]) KWS_HIP(hipEventCreateWithFlags(&h->ovl_tail[l], hipEventDisableTiming));
        KWS_HIP(hipEventRecord(h->ovl_tail[l], h->lane_stream[l]));
    }
    h->ovl_tail_valid = true;
    return KWS_OK;
}

// Can the last layer's launch of a (B, T) step on this handle take the window tail along?  (kws_stream_feed asks before it
// hands one to step_impl; the kernels without a tail instantiation -- generic, pipelined, overlapped, int8, single-layer fp32,
// 4-wave bf16 -- are followed by window_inc_kernel instead.)
static bool step_takes_window(kws_handle h, int B, int T, int window_chunks) {
    const kws_config& c = h->cfg;
    static const bool off = [] { const char* e = getenv("KWS_NO_WINDOW_TAIL"); return e && e[0] == '1'; }();     // A/B switch (tools/bench_e2e.py)
    if (off || T < 1 || T > kws::kWinTailMaxFrames || window_chunks > kws::kWinTailMaxChunks) return false;
    // One group per workgroup only: the tail is ~2 us of latency-bound work at the end of a group.  At the end of the launch
    // that replaces a ~4 us launch of its own; in a persistent workgroup it would sit between two groups, on the critical
    // path once per group (measured, bf16, 16384 streams: 0.349-0.359 ms per chunk with the tail against 0.338-0.340 with
    // window_inc_kernel behind the stack)
    if ((B + kws::kStreamsPerGroup - 1) / kws::kStreamsPerGroup > (h->num_cus > 0 ? h->num_cus : 256)) return false;
    if (c.precision == KWS_BF16) return kws::gru_stack_bf16_takes_window(h->bf_kx0, c.num_layers);
    if (c.precision == KWS_F16X3) return !h->f16_generic;
    if (c.precision != KWS_FP32 || pipeline_eligible(h, B) || overlap_eligible(h, B, T)) return false;
    const LayerDev& Ld = h->layers[c.num_layers - 1];
    const bool resident = h->kernel_kind == KWS_KERNEL_RESIDENT || (h->kernel_kind == KWS_KERNEL_AUTO && Ld.resident_ok);
    return resident && kws::gru_resident_takes_window(c.num_layers == 1, true);
}

static int step_impl(kws_handle h, const float* mel, const float* state_in, float* logits, float* softmax,
                     float* state_out, const int32_t* seq_len, const uint8_t* reset_mask, int8_t* tokens,
                     int32_t* prev_word, float decode2_thres, int B, int T, void* stream, const kws::WindowTail* wt, bool locked = false);

// Ordering of a call against the handle's previous one (kws_model::last_done): device-side, never a host wait.
static int call_enter(kws_handle h, hipStream_t st) {
    if (h->last_stream_valid && st != h->last_stream) KWS_HIP(hipStreamWaitEvent(st, h->last_done, 0));
    h->last_stream = st;
    h->last_stream_valid = true;
    return KWS_OK;
}
static int call_leave(kws_handle h, hipStream_t st) {
    KWS_HIP(hipEventRecord(h->last_done, st));
    return KWS_OK;
}

int kws_step(kws_handle h, const float* mel, const float* state_in, float* logits, float* softmax,
             float* state_out, const int32_t* seq_len, const uint8_t* reset_mask, int8_t* tokens,
             int32_t* prev_word, float decode2_thres, int B, int T, void* stream) {
    return step_impl(h, mel, state_in, logits, softmax, state_out, seq_len, reset_mask, tokens, prev_word, decode2_thres, B, T, stream, nullptr);
}

// wt: the stream manager's decode window, to ride at the end of the last layer's launch (step_takes_window said yes)
static int step_body(kws_handle h, const float* mel, const float* state_in, float* logits, float* softmax,
                     float* state_out, const int32_t* seq_len, const uint8_t* reset_mask, int8_t* tokens,
                     int32_t* prev_word, float decode2_thres, int B, int T, hipStream_t st, const kws::WindowTail* wt);

// locked: the caller (kws_stream_feed) already holds the handle and has ordered `stream` behind its previous call
static int step_impl(kws_handle h, const float* mel, const float* state_in, float* logits, float* softmax,
                     float* state_out, const int32_t* seq_len, const uint8_t* reset_mask, int8_t* tokens,
                     int32_t* prev_word, float decode2_thres, int B, int T, void* stream, const kws::WindowTail* wt, bool locked) {
    if (!h) return fail(KWS_ERR_INVALID_ARGUMENT, "handle is null");
    if (wt && (seq_len || !step_takes_window(h, B, T, wt->nq))) return fail(KWS_ERR_UNSUPPORTED, "internal: this step cannot take a window tail");
    if (B < 0 || T < 0) return fail(KWS_ERR_INVALID_ARGUMENT, "negative B=%d or T=%d", B, T);
    if (B == 0) return KWS_OK;   // nothing to advance (empty tensors have null data pointers)
    if (!state_in || !state_out) return fail(KWS_ERR_INVALID_ARGUMENT, "state_in/state_out must not be null");
    if (tokens && !prev_word) return fail(KWS_ERR_INVALID_ARGUMENT, "tokens requires prev_word");
    hipStream_t st = static_cast<hipStream_t>(stream);
    if (locked) return step_body(h, mel, state_in, logits, softmax, state_out, seq_len, reset_mask, tokens, prev_word, decode2_thres, B, T, st, wt);
    BusyGuard busy(h->in_call);
    if (!busy.owned)
        return fail(KWS_ERR_BUSY, "kws_step: another host thread is inside a call on this handle (one thread at a time per handle; "
                    "use one handle per thread)");
    int rc = call_enter(h, st);       // the previous call's kernels may still own the seams: this stream waits for them on the device
    if (rc != KWS_OK) return rc;
    rc = step_body(h, mel, state_in, logits, softmax, state_out, seq_len, reset_mask, tokens, prev_word, decode2_thres, B, T, st, wt);
    const int rl = call_leave(h, st);     // also after a failure: whatever was queued before it is what the next call must wait for
    return rc != KWS_OK ? rc : rl;
}

static int step_body(kws_handle h, const float* mel, const float* state_in, float* logits, float* softmax,
                     float* state_out, const int32_t* seq_len, const uint8_t* reset_mask, int8_t* tokens,
                     int32_t* prev_word, float decode2_thres, int B, int T, hipStream_t st, const kws::WindowTail* wt) {
    const kws_config& c = h->cfg;
    const int H = c.hidden, L = c.num_layers;
    {
        const int rc = check_pipe_error(h);      // raised by an earlier layer-pipelined step of this handle
        if (rc != KWS_OK) return rc;
        // the handle's weights and scratch live on the device that was current at kws_create
        int dev = -1;
        if (hipGetDevice(&dev) == hipSuccess && dev != h->device)
            return fail(KWS_ERR_INVALID_ARGUMENT, "handle was created on device %d, the current device is %d", h->device, dev);
    }
    if (h->ovl_tail_valid) {
        // the previous call ran its upper layers on the handle's own streams: whatever stream this call comes in on,
        // it must not touch the seam buffers (or reallocate them) before those kernels are done
        for (int l = 1; l < L; ++l)
            if (h->ovl_tail[l]) KWS_HIP(hipStreamWaitEvent(st, h->ovl_tail[l], 0));
        h->ovl_tail_valid = false;
    }
    if (T == 0) {
        // dynamic_rnn over zero frames hands the initial state back -- and clean_state() (detector.py:313-316) has already
        // zeroed it for the streams the mask names
        if (reset_mask) {
            hipError_t e = kws::launch_state_passthrough(state_in, state_out, reset_mask, prev_word, L, B, H, st);
            if (e != hipSuccess) return hip_fail(e, "launch state_passthrough");
        } else if (state_out != state_in) {
            KWS_HIP(hipMemcpyAsync(state_out, state_in, (size_t)L * B * H * sizeof(float), hipMemcpyDeviceToDevice, st));
        }
        return KWS_OK;
    }
    if (!mel) return fail(KWS_ERR_INVALID_ARGUMENT, "mel is null");
    if (c.precision != KWS_BF16 && (c.precision != KWS_F16X3 || h->f16_generic)) {
        // the streaming kernels address a group's seam (T x H/16 KiB) through buffer instructions with 32-bit offsets
        bool streaming = c.precision == KWS_F16X3;
        for (const auto& Ld : h->layers)
            streaming |= !(h->kernel_kind == KWS_KERNEL_RESIDENT || (h->kernel_kind == KWS_KERNEL_AUTO && Ld.resident_ok));
        if (streaming && (long long)T * (H / 16) >= (1LL << 21))
            return fail(KWS_ERR_UNSUPPORTED, "T=%d frames of hidden=%d exceed the 2 GiB a stream group's seam may span: split the call "
                        "(state carried across calls gives identical results)", T, H);
    }
    if (c.precision == KWS_BF16) {
        if ((reinterpret_cast<uintptr_t>(mel) & 15) != 0) return fail(KWS_ERR_INVALID_ARGUMENT, "mel must be 16-byte aligned");
        kws::GruBf16Params bp;
        memset(&bp, 0, sizeof(bp));
        for (int l = 0; l < L; ++l) {
            bp.w[l] = reinterpret_cast<const uint4*>(h->d_weights + h->bf_w[l]);
            bp.bias[l] = h->d_weights + h->layers[l].bias;
        }
        bp.wfc = reinterpret_cast<const uint4*>(h->d_weights + h->bf_wfc);
        bp.bfc = h->d_weights + h->bfc_off;
        bp.x_mel = mel; bp.state_in = state_in; bp.state_out = state_out;
        bp.seq_len = seq_len; bp.reset = reset_mask;
        bp.epi.logits = logits; bp.epi.softmax = softmax; bp.epi.tokens = tokens; bp.epi.prev_word = prev_word;
        bp.epi.decode_thres = decode2_thres; bp.epi.value_clip = c.value_clip; bp.epi.use_relu = c.use_relu;
        bp.epi.B = B; bp.epi.T = T; bp.epi.C = c.num_classes;
        if (wt) bp.epi.win = *wt;
        bp.B = B; bp.T = T; bp.I = c.n_mel; bp.L = L;
        hipEvent_t ea = nullptr, eb = nullptr;
        if (h->profiling) {
            for (hipEvent_t* ev : {&ea, &eb}) {
                if (!h->event_pool.empty()) { *ev = h->event_pool.back(); h->event_pool.pop_back(); }
                else KWS_HIP(hipEventCreate(ev));
            }
            KWS_HIP(hipEventRecord(ea, st));
        }
        hipError_t e = kws::launch_gru_stack_bf16(bp, h->bf_kx0, L, st);
        if (e != hipSuccess) return hip_fail(e, "launch gru_stack_bf16");
        h->launch_tag[0] = {kws_model::kBf16Stack, 0, 0, 0, (uint8_t)(wt != nullptr)};
        for (int l = 1; l < L; ++l) h->launch_tag[l] = {};
        if (h->profiling) {
            KWS_HIP(hipEventRecord(eb, st));
            h->pending.push_back({0, ea, eb});
        }
        return KWS_OK;
    }
    if ((reinterpret_cast<uintptr_t>(mel) & 15) != 0) return fail(KWS_ERR_INVALID_ARGUMENT, "mel must be 16-byte aligned");
    if (c.precision == KWS_F16X3) {
        // one launch per layer; the seams (same size as the fp32 ones) hold the layer outputs already split into fp16 pairs
        int rc = ensure_scratch(h, B, T);
        if (rc != KWS_OK) return rc;
        // hidden = 256: weights streamed from L2; all L x groups workgroups in ONE layer-pipelined grid when they fit the chip
        const bool f16_pipelined = h->f16_generic && pipeline_eligible(h, B);
        kws::GruF16StackParams fsp;
        if (f16_pipelined) {
            KWS_HIP(hipMemsetAsync(h->pipe_ready, 0, (size_t)L * h->pipe_groups * sizeof(int), st));
            memset(&fsp, 0, sizeof(fsp));
            fsp.L = L; fsp.G = (B + kws::kStreamsPerGroup - 1) / kws::kStreamsPerGroup; fsp.xcd_affine = (8 % L == 0) ? 1 : 0;
        }
        for (int l = 0; l < L; ++l) {
            const bool first = l == 0, last = l == L - 1;
            kws::GruF16Params fp;
            memset(&fp, 0, sizeof(fp));
            fp.w = reinterpret_cast<const uint4*>(h->d_weights + h->f16_w[l]);
            fp.bias = h->d_weights + h->layers[l].bias;
            fp.wfc = reinterpret_cast<const uint4*>(h->d_weights + h->f16_wfc);
            fp.bfc = h->d_weights + h->bfc_off;
            fp.x_mel = mel;
            fp.x_prev = first ? nullptr : reinterpret_cast<const uint4*>(h->scratch[(l - 1) % h->nscratch]);
            fp.h_out = last ? nullptr : reinterpret_cast<uint4*>(h->scratch[l % h->nscratch]);
            fp.state_in = state_in + (size_t)l * B * H;
            fp.state_out = state_out + (size_t)l * B * H;
            fp.seq_len = seq_len; fp.reset = reset_mask;
            fp.epi.logits = logits; fp.epi.softmax = softmax; fp.epi.tokens = tokens; fp.epi.prev_word = prev_word;
            fp.epi.decode_thres = decode2_thres; fp.epi.value_clip = c.value_clip; fp.epi.use_relu = c.use_relu;
            fp.epi.B = B; fp.epi.T = T; fp.epi.C = c.num_classes;
            if (wt && last) fp.epi.win = *wt;
            fp.B = B; fp.T = T; fp.I = h->layers[l].in_dim;
            if (f16_pipelined) {
                fp.epi.ready_in = first ? nullptr : h->pipe_ready + (size_t)(l - 1) * h->pipe_groups;
                fp.epi.ready_out = last ? nullptr : h->pipe_ready + (size_t)l * h->pipe_groups;
                fp.epi.pipe_error = h->pipe_error_dev;
                fsp.layer[l] = fp;
                h->launch_tag[l] = {};
                if (!last) continue;
            }
            hipEvent_t ea = nullptr, eb = nullptr;
            if (h->profiling) {
                for (hipEvent_t* ev : {&ea, &eb}) {
                    if (!h->event_pool.empty()) { *ev = h->event_pool.back(); h->event_pool.pop_back(); }
                    else KWS_HIP(hipEventCreate(ev));
                }
                KWS_HIP(hipEventRecord(ea, st));
            }
            hipError_t e;
            if (f16_pipelined) {
                e = kws::launch_gru_stack_f16x3_pipelined(fsp, H, st);        // timed as the last layer's slot
                if (e != hipSuccess) return hip_fail(e, "launch gru_stack_f16x3_pipelined");
                h->launch_tag[l] = {kws_model::kF16x3Pipelined, (uint8_t)(H / 64), 0, 0};
            } else if (h->f16_generic) {
                e = kws::launch_gru_layer_f16x3_generic(fp, H, first, last, st);
                if (e != hipSuccess) return hip_fail(e, "launch gru_layer_f16x3_generic");
                h->launch_tag[l] = {kws_model::kF16x3Generic, (uint8_t)(H / 64), first, last};
            } else {
                e = kws::launch_gru_layer_f16x3(fp, first, last, st);
                if (e != hipSuccess) return hip_fail(e, "launch gru_layer_f16x3");
                h->launch_tag[l] = {kws_model::kF16x3, (uint8_t)(first ? h->f16_kx0 : 4), first, last, (uint8_t)(wt != nullptr && last)};
            }
            if (h->profiling) {
                KWS_HIP(hipEventRecord(eb, st));
                h->pending.push_back({l, ea, eb});
            }
        }
        return KWS_OK;
    }
    if (overlap_eligible(h, B, T))
        return step_overlapped(h, mel, state_in, logits, softmax, state_out, seq_len, reset_mask, tokens, prev_word,
                               decode2_thres, B, T, st);
    int rc = ensure_scratch(h, B, T);
    if (rc != KWS_OK) return rc;

    const bool int8 = c.precision == KWS_INT8;
    if (int8 && prev_word)
        KWS_HIP(hipMemcpyAsync(h->oct_prev, prev_word, (size_t)B * sizeof(int32_t), hipMemcpyDeviceToDevice, st));
    // layer-pipelined launch (gru_stack_generic_pipelined): all layers of all groups in one grid
    const bool pipelined = pipeline_eligible(h, B);
    const int groups = (B + kws::kStreamsPerGroup - 1) / kws::kStreamsPerGroup;
    kws::GruStackParams sp;
    if (pipelined) {
        KWS_HIP(hipMemsetAsync(h->pipe_ready, 0, (size_t)L * h->pipe_groups * sizeof(int), st));
        memset(&sp, 0, sizeof(sp));
        sp.L = L; sp.G = groups; sp.xcd_affine = (8 % L == 0) ? 1 : 0;
    }
    for (int l = 0; l < L; ++l) {
        const LayerDev& Ld = h->layers[l];
        // int8: every GRU layer hands its output rows to the next stage through the xl scratch; the class
        // projection is its own OctbitMatMul call over the whole [T,H] block (launch_octbit_fc below)
        const bool first = l == 0, last = !int8 && l == L - 1;
        const bool resident = !pipelined && (h->kernel_kind == KWS_KERNEL_RESIDENT ||
                                             (h->kernel_kind == KWS_KERNEL_AUTO && Ld.resident_ok));
        kws::GruLayerParams p;
        memset(&p, 0, sizeof(p));
        // resident kernels read the group-of-4 layouts too (dwordx4 prologue), except the first layer's
        // x-part, whose k map is the interleaved one
        p.wx = h->d_weights + ((resident && first) ? Ld.wx_res : Ld.wx_gen);
        p.wh = h->d_weights + Ld.wh_gen;
        p.bias = h->d_weights + Ld.bias;
        p.wfc = h->d_weights + h->wfc_off;
        p.bfc = h->d_weights + h->bfc_off;
        p.x_mel = mel;
        p.x_prev = first ? nullptr : h->scratch[(l - 1) % h->nscratch];
        p.h_out = last ? nullptr : h->scratch[l % h->nscratch];
        p.state_in = state_in + (size_t)l * B * H;
        p.state_out = state_out + (size_t)l * B * H;
        p.seq_len = seq_len;
        p.reset = reset_mask;
        p.logits = logits;
        p.softmax = softmax;
        p.tokens = tokens;
        p.prev_word = prev_word;
        p.decode_thres = decode2_thres;
        p.value_clip = c.value_clip;
        p.use_relu = c.use_relu;
        p.B = B; p.T = T; p.I = Ld.in_dim; p.C = c.num_classes;
        p.KCX = resident ? Ld.kcx_res : Ld.kcx_gen;
        if (wt && last) p.win = *wt;

        if (pipelined) {
            p.ready_in = first ? nullptr : h->pipe_ready + (size_t)(l - 1) * h->pipe_groups;
            p.ready_out = last ? nullptr : h->pipe_ready + (size_t)l * h->pipe_groups;
            p.pipe_error = h->pipe_error_dev;
            sp.layer[l] = p;
            if (!last) continue;
        }
        hipEvent_t ea = nullptr, eb = nullptr;
        if (h->profiling) {
            for (hipEvent_t* ev : {&ea, &eb}) {
                if (!h->event_pool.empty()) { *ev = h->event_pool.back(); h->event_pool.pop_back(); }
                else KWS_HIP(hipEventCreate(ev));
            }
            KWS_HIP(hipEventRecord(ea, st));
        }
        hipError_t e;
        if (pipelined) {
            e = kws::launch_gru_stack_generic_pipelined(sp, H, st);       // timed as the last layer's slot
            if (e != hipSuccess) return hip_fail(e, "launch gru_stack_generic_pipelined");
        } else if (int8 && h->oct[l].quantised) {
            const kws_model::OctLayer& O = h->oct[l];
            kws::GruOctbitParams op;
            memset(&op, 0, sizeof(op));
            op.wg = reinterpret_cast<const uint32_t*>(h->d_weights + O.wg);
            op.wc = reinterpret_cast<const uint32_t*>(h->d_weights + O.wc);
            op.bias = p.bias; op.b127 = h->d_weights + O.b127;
            op.scale_g = O.scale_g; op.scale_c = O.scale_c;
            op.x_prev = p.x_prev; op.h_out = p.h_out;
            op.state_in = p.state_in; op.state_out = p.state_out;
            op.seq_len = seq_len; op.reset = reset_mask;
            op.aq = h->oct_aq; op.B = B; op.T = T;
            op.range = (l == L - 1) ? h->oct_range : nullptr;
            e = kws::launch_gru_layer_octbit(op, st);
            if (e != hipSuccess) return hip_fail(e, "launch gru_layer_octbit");
        } else {
            e = resident ? kws::launch_gru_layer_resident(p, first, last, st)
                         : kws::launch_gru_layer_generic(p, H, first, last, st);
            if (e != hipSuccess) return hip_fail(e, resident ? "launch gru_layer_resident" : "launch gru_layer_generic");
        }
        if (pipelined) {
            h->launch_tag[l] = {kws_model::kPipelined, (uint8_t)(H / 64), 0, 0};
            for (int k = 0; k < l; ++k) h->launch_tag[k] = {};
        } else if (int8 && h->oct[l].quantised) h->launch_tag[l] = {(uint8_t)(l == L - 1 ? kws_model::kOctbitFc : kws_model::kOctbit), 0, 0, 0};
        else if (resident) h->launch_tag[l] = {kws_model::kResident, (uint8_t)p.KCX, first, last, (uint8_t)(wt != nullptr && last)};
        else h->launch_tag[l] = {kws_model::kGeneric, (uint8_t)(H / 64), first, last};
        if (int8 && l == L - 1) {
            kws::OctbitFcParams fp;
            memset(&fp, 0, sizeof(fp));
            fp.wfc = reinterpret_cast<const uint32_t*>(h->d_weights + h->oct_wfc);
            fp.b127 = h->d_weights + h->oct_b127fc;
            fp.bfc = h->d_weights + h->bfc_off;
            fp.scale_w = h->oct_scale_fc;
            fp.h_top = h->scratch[l % h->nscratch];
            fp.range = h->oct_range;
            fp.range_ready = h->oct[l].quantised ? 1 : 0;
            fp.prev_in = prev_word ? h->oct_prev : nullptr;
            fp.logits = logits; fp.softmax = softmax; fp.tokens = tokens; fp.prev_word = prev_word;
            fp.decode_thres = decode2_thres; fp.value_clip = c.value_clip; fp.use_relu = c.use_relu;
            fp.B = B; fp.T = T; fp.C = c.num_classes;
            e = kws::launch_octbit_fc(fp, st);
            if (e != hipSuccess) return hip_fail(e, "launch octbit_fc");
        }
        if (h->profiling) {
            KWS_HIP(hipEventRecord(eb, st));
            h->pending.push_back({l, ea, eb});
        }
    }
    return KWS_OK;
}

int kws_window_create(int B, int max_chunks, int max_frames, int C, float thres, kws_window_handle* out) {
    if (!out) return fail(KWS_ERR_INVALID_ARGUMENT, "out handle pointer is null");
    *out = nullptr;
    if (B < 1 || max_chunks < 1 || max_frames < 1 || C < 3 || C > 64)
        return fail(KWS_ERR_INVALID_ARGUMENT, "bad window shape B=%d chunks=%d frames=%d C=%d", B, max_chunks, max_frames, C);
    // window_step_kernel: one lane per queued chunk and two byte images of the window (ring, emitted words) in LDS
    if (max_chunks > 64)
        return fail(KWS_ERR_UNSUPPORTED, "max_chunks=%d unsupported (1..64; the reference uses SimpleQueue(15), detector.py:122)", max_chunks);
    if ((size_t)2 * max_chunks * ((max_frames + 15) & ~15) > 48 * 1024)
        return fail(KWS_ERR_UNSUPPORTED, "window of %d chunks x %d frames exceeds the 48 KiB of LDS the kernel stages it in", max_chunks, max_frames);
    int ndev = 0;
    if (hipGetDeviceCount(&ndev) != hipSuccess || ndev == 0) return fail(KWS_ERR_NO_DEVICE, "no HIP device visible");
    kws_window* wnd = new (std::nothrow) kws_window();
    if (!wnd) return fail(KWS_ERR_OUT_OF_MEMORY, "host allocation failed");
    wnd->B = B; wnd->nq = max_chunks; wnd->tmax = max_frames; wnd->tmax_pad = (max_frames + 15) & ~15; wnd->C = C;
    wnd->thres = thres;
    // the summaries of the incremental form (what kws_stream_feed drives: 32 + 4 bytes per queued chunk and stream); the frame
    // ring of the re-scanning kws_window_step (tmax_pad + 4 bytes per queued chunk) is allocated by its first call
    hipError_t e = hipMalloc(reinterpret_cast<void**>(&wnd->inc_tab), (size_t)B * max_chunks * 32);
    if (e == hipSuccess) e = hipMalloc(reinterpret_cast<void**>(&wnd->inc_meta), (size_t)B * max_chunks * sizeof(uint32_t));
    if (e == hipSuccess) e = hipMalloc(reinterpret_cast<void**>(&wnd->inc_head), (size_t)B * sizeof(int));
    if (e == hipSuccess) e = hipMalloc(reinterpret_cast<void**>(&wnd->inc_count), (size_t)B * sizeof(int));
    if (e == hipSuccess) e = hipMalloc(reinterpret_cast<void**>(&wnd->inc_delta_dev), 256);
    if (e == hipSuccess) e = hipMemset(wnd->inc_tab, 0, (size_t)B * max_chunks * 32);
    if (e == hipSuccess) e = hipMemset(wnd->inc_meta, 0, (size_t)B * max_chunks * sizeof(uint32_t));
    if (e == hipSuccess) e = kws::launch_window_reset(B, wnd->inc_head, wnd->inc_count, nullptr);
    if (e == hipSuccess) e = hipDeviceSynchronize();
    if (e != hipSuccess) { kws_window_destroy(wnd); return hip_fail(e, "kws_window_create"); }
    live_register(wnd);
    *out = wnd;
    return KWS_OK;
}

int kws_window_destroy(kws_window_handle h) {
    if (!h) return KWS_OK;
    live_unregister(h);
    hipDeviceSynchronize();
    if (h->words) hipFree(h->words);
    if (h->lens) hipFree(h->lens);
    if (h->head) hipFree(h->head);
    if (h->count) hipFree(h->count);
    if (h->inc_tab) hipFree(h->inc_tab);
    if (h->inc_meta) hipFree(h->inc_meta);
    if (h->inc_head) hipFree(h->inc_head);
    if (h->inc_count) hipFree(h->inc_count);
    if (h->inc_delta_dev) hipFree(h->inc_delta_dev);
    delete h;
    return KWS_OK;
}

int kws_window_step(kws_window_handle h, const float* softmax, int T, const uint8_t* clear_before, const char* label,
                    int32_t* hit, uint8_t* restart, void* stream) {
    if (!h) return fail(KWS_ERR_INVALID_ARGUMENT, "handle is null");
    if (T < 0 || T > h->tmax) return fail(KWS_ERR_INVALID_ARGUMENT, "T=%d outside [0,%d]", T, h->tmax);
    if (!hit || (!softmax && T > 0) || !label) return fail(KWS_ERR_INVALID_ARGUMENT, "null pointer argument");
    const int n = (int)strlen(label);
    if (n > 16) return fail(KWS_ERR_INVALID_ARGUMENT, "label longer than 16 digits");
    kws::WindowParams p;
    memset(&p, 0, sizeof(p));
    for (int i = 0; i < n; ++i) {
        if (label[i] < '1' || label[i] > '9') return fail(KWS_ERR_INVALID_ARGUMENT, "label must be digits 1..9, got '%s'", label);
        p.label[i] = label[i] - '0';
    }
    p.label_len = n;
    if (!h->words) {          // first re-scanning step of this window: its frame ring (synchronises once)
        hipError_t e = hipMalloc(reinterpret_cast<void**>(&h->words), (size_t)h->B * h->nq * h->tmax_pad);
        if (e == hipSuccess) e = hipMalloc(reinterpret_cast<void**>(&h->lens), (size_t)h->B * h->nq * sizeof(int));
        if (e == hipSuccess) e = hipMalloc(reinterpret_cast<void**>(&h->head), (size_t)h->B * sizeof(int));
        if (e == hipSuccess) e = hipMalloc(reinterpret_cast<void**>(&h->count), (size_t)h->B * sizeof(int));
        if (e == hipSuccess) e = kws::launch_window_reset(h->B, h->head, h->count, nullptr);
        if (e == hipSuccess) e = hipDeviceSynchronize();
        if (e != hipSuccess) {
            for (void* q : {(void*)h->words, (void*)h->lens, (void*)h->head, (void*)h->count}) if (q) hipFree(q);
            h->words = nullptr; h->lens = nullptr; h->head = nullptr; h->count = nullptr;
            return hip_fail(e, "hipMalloc(window frame ring)");
        }
    }
    p.words = h->words; p.lens = h->lens; p.head = h->head; p.count = h->count;
    p.softmax = softmax; p.clear_before = clear_before; p.hit = hit; p.restart = restart;
    p.thres = h->thres; p.B = h->B; p.T = T; p.C = h->C; p.nq = h->nq; p.tmax = h->tmax_pad;
    hipError_t e = kws::launch_window_step(p, static_cast<hipStream_t>(stream));
    if (e != hipSuccess) return hip_fail(e, "launch window_step");
    return KWS_OK;
}

// The label matcher of the incremental window: KMP automaton over emitted words, delta[q * 16 + w] = digits of the label
// matched after reading word w (1..15) with q matched before (q < len); words the label does not contain lead to 0.
static void window_label_delta(const char* label, int n, uint8_t* delta) {
    memset(delta, 0, 256);
    for (int q = 0; q < n; ++q)
        for (int w = 1; w < 16; ++w) {
            int k = q + 1;                       // longest k with label[0..k) a suffix of label[0..q) + w
            for (; k > 0; --k) {
                if (label[k - 1] - '0' != w) continue;
                bool ok = true;
                for (int i = 0; i < k - 1 && ok; ++i) ok = label[i] == label[q - (k - 1) + i];
                if (ok) break;
            }
            delta[q * 16 + w] = (uint8_t)k;
        }
}

// LDS of window_inc_kernel for chunks of T frames (launch_window_inc, stream_kernels.hip): the 16 streams' frame words, the label
// matcher, the rings.  kws_window_create only sizes the re-scanning kernel; the incremental entry points check this one.
static size_t window_inc_lds_bytes(int T, int nq) {
    const int stride = (T + 15) & ~15;
    return (size_t)16 * (stride > 0 ? stride : 16) + 256 + kws::window_tail_scratch_bytes(nq);
}
constexpr size_t kWindowIncLdsMax = 160 * 1024;

// Binds `label` to the window's incremental state (the queued summaries are label-specific).  The first binding uploads the
// matcher (synchronises); the same label again is free; another label while chunks may be queued is refused.
static int window_bind_label(kws_window* w, const char* label) {
    const int n = (int)strlen(label);
    if (n > 15) return fail(KWS_ERR_INVALID_ARGUMENT, "the incremental window takes labels of up to 15 digits (its matcher has 16 states); "
                            "kws_window_step re-scans the frames for longer ones");
    for (int i = 0; i < n; ++i)
        if (label[i] < '1' || label[i] > '9') return fail(KWS_ERR_INVALID_ARGUMENT, "label must be digits 1..9, got '%s'", label);
    if (w->inc_bound) {
        if (strcmp(w->inc_label, label) == 0) return KWS_OK;
        return fail(KWS_ERR_INVALID_ARGUMENT, "the window's incremental state was built for label '%s'; it cannot continue with '%s' "
                    "(create another window, or use kws_window_step, which re-scans the frames)", w->inc_label, label);
    }
    window_label_delta(label, n, w->inc_delta);
    KWS_HIP(hipMemcpy(w->inc_delta_dev, w->inc_delta, 256, hipMemcpyHostToDevice));
    memcpy(w->inc_label, label, n + 1);
    w->inc_bound = true;
    return KWS_OK;
}

static kws::WindowTail window_tail_params(kws_window* w, const uint8_t* clear_before, int32_t* hit, uint8_t* restart) {
    kws::WindowTail t;
    memset(&t, 0, sizeof(t));
    t.tab = w->inc_tab; t.meta = w->inc_meta; t.head = w->inc_head; t.count = w->inc_count; t.delta = w->inc_delta_dev;
    t.clear_before = clear_before; t.hit = hit; t.restart = restart; t.nq = w->nq; t.n_label = (int)strlen(w->inc_label);
    return t;
}

int kws_window_step_incremental(kws_window_handle h, const float* softmax, int T, const uint8_t* clear_before, const char* label,
                                int32_t* hit, uint8_t* restart, void* stream) {
    if (!h) return fail(KWS_ERR_INVALID_ARGUMENT, "handle is null");
    if (T < 0 || T > h->tmax) return fail(KWS_ERR_INVALID_ARGUMENT, "T=%d outside [0,%d]", T, h->tmax);
    if (!hit || (!softmax && T > 0) || !label) return fail(KWS_ERR_INVALID_ARGUMENT, "null pointer argument");
    if (window_inc_lds_bytes(T, h->nq) > kWindowIncLdsMax)
        return fail(KWS_ERR_UNSUPPORTED, "the incremental window step stages 16 streams x %d frame words and their %d-chunk rings in LDS: %zu bytes "
                    "exceed the %zu a workgroup may hold (shorter chunks, or kws_window_step, which re-scans the frames)", T, h->nq,
                    window_inc_lds_bytes(T, h->nq), kWindowIncLdsMax);
    const int rc = window_bind_label(h, label);
    if (rc != KWS_OK) return rc;
    kws::WindowIncParams p;
    memset(&p, 0, sizeof(p));
    p.win = window_tail_params(h, clear_before, hit, restart);
    memcpy(p.delta, h->inc_delta, 256);
    p.softmax = softmax; p.thres = h->thres; p.B = h->B; p.T = T; p.C = h->C;
    hipError_t e = kws::launch_window_inc(p, static_cast<hipStream_t>(stream));
    if (e != hipSuccess) return hip_fail(e, "launch window_inc");
    return KWS_OK;
}

int kws_frontend_frames(const kws_frontend_config* cfg, int n_samples) {
    if (!cfg || cfg->fft_size <= 0 || cfg->hop_size <= 0 || n_samples < cfg->fft_size) return 0;
    return 1 + (n_samples - cfg->fft_size) / cfg->hop_size;
}

int kws_frontend_create(const kws_frontend_config* cfg, kws_frontend_handle* out) {
    if (!out) return fail(KWS_ERR_INVALID_ARGUMENT, "out handle pointer is null");
    *out = nullptr;
    if (!cfg) return fail(KWS_ERR_INVALID_ARGUMENT, "config is null");
    if (cfg->fft_size < 16 || cfg->fft_size > 496 || cfg->fft_size % 16 != 0)
        return fail(KWS_ERR_UNSUPPORTED, "fft_size=%d must be a multiple of 16 in [16,496] (the reference uses 400)", cfg->fft_size);
    if (cfg->hop_size < 1 || cfg->n_mel < 1 || cfg->n_mel > 64 || cfg->samplerate < 1)
        return fail(KWS_ERR_INVALID_ARGUMENT, "bad hop_size/n_mel/samplerate (%d/%d/%d)", cfg->hop_size, cfg->n_mel, cfg->samplerate);
    if (!(cfg->fmin >= 0.f) || !(cfg->fmax > cfg->fmin) || cfg->fmax > cfg->samplerate / 2.0f + 1e-3f)
        return fail(KWS_ERR_INVALID_ARGUMENT, "need 0 <= fmin < fmax <= sr/2");
    int ndev = 0;
    if (hipGetDeviceCount(&ndev) != hipSuccess || ndev == 0) return fail(KWS_ERR_NO_DEVICE, "no HIP device visible");
    kws_frontend* f = new (std::nothrow) kws_frontend();
    if (!f) return fail(KWS_ERR_OUT_OF_MEMORY, "host allocation failed");
    f->cfg = *cfg;
    // frontend_kernels.hip: bins k = 0..N/4 are contracted, each over the even and the odd folded samples
    const int N = cfg->fft_size, NF = N / 2 + 1, NH = N / 2, NQ = N / 4, TILES = (NQ + 1 + 15) / 16, KC4 = TILES;
    f->kc4 = KC4;
    f->nf_tiles = TILES;
    f->mel_tiles = (cfg->n_mel + 15) / 16;
    f->basis = slaney_mel_basis(cfg->samplerate, N, cfg->n_mel, cfg->fmin, cfg->fmax);
    std::vector<float> host;
    f->dft_off = 0;
    host.resize((size_t)4 * TILES * KC4 * 64 * 4, 0.f);
    const double two_pi = 6.283185307179586476925286766559;
    for (int tile = 0; tile < TILES; ++tile)
        for (int a = 0; a < 4; ++a)                      // a = 2 * (cos|sin) + parity of n
            for (int k4 = 0; k4 < KC4; ++k4)
                for (int lane = 0; lane < 64; ++lane)
                    for (int e = 0; e < 4; ++e) {
                        const int g = lane >> 4, i = lane & 15, cs = a >> 1, par = a & 1;
                        const int bin = 16 * tile + i, m = 4 * (4 * k4 + e) + g, n = 2 * m + par;
                        float v = 0.f;
                        // cos rows use folded samples 0..N/2, sin rows 1..N/2-1 (sin vanishes at 0 and N/2)
                        if (bin <= NQ && n <= NH && !(cs == 1 && (n == 0 || n == NH))) {
                            const double ang = two_pi * (double)(((long long)bin * n) % N) / N;
                            v = (float)(cs == 0 ? std::cos(ang) : std::sin(ang));
                        }
                        host[((((size_t)(4 * tile + a) * KC4 + k4) * 64 + lane) * 4) + e] = v;
                    }
    // mel basis fragments, xl k map over k = 0..N/4: direct set basis[m][k], mirrored set basis[m][N/2 - k] (k < N/4)
    f->melw_off = host.size();
    host.resize(host.size() + (size_t)f->mel_tiles * TILES * 8 * 64, 0.f);
    for (int mt = 0; mt < f->mel_tiles; ++mt)
        for (int t = 0; t < TILES; ++t)
            for (int mir = 0; mir < 2; ++mir)
                for (int e = 0; e < 4; ++e)
                    for (int lane = 0; lane < 64; ++lane) {
                        const int g = lane >> 4, i = lane & 15;
                        const int k = 16 * t + 4 * g + e, m = 16 * mt + i;
                        float v = 0.f;
                        if (m < cfg->n_mel) {
                            if (mir == 0 && k <= NQ) v = f->basis[(size_t)m * NF + k];
                            if (mir == 1 && k < NQ) v = f->basis[(size_t)m * NF + (NH - k)];
                        }
                        host[f->melw_off + ((((size_t)mt * TILES + t) * 2 + mir) * 4 + e) * 64 + lane] = v;
                    }
    if (N == 400) {
        // fft_frontend.hip: the 16 x 25 real FFT.  Twiddles W400^{n2 k1} as (cos, sin) [k1 = 1..12][n2 = 0..15].  Mel basis as MFMA
        // A fragments over 4-bin groups (k = g <-> bin 4 group + g): per tile of 16 filters only the contiguous run of groups that
        // carry a non-zero weight, padded to a multiple of four; bins > 200 are zero rows.
        f->fft_tw_off = host.size();                        // [6 pairs (k1 = 2i+1, 2i+2)][16 n2][cos, sin, cos, sin]
        host.resize(host.size() + 12 * 16 * 2, 0.f);
        for (int k1 = 1; k1 <= 12; ++k1)
            for (int n2 = 0; n2 < 16; ++n2) {
                const double ang = two_pi * (double)(n2 * k1) / 400.0;
                const size_t at = f->fft_tw_off + ((size_t)((k1 - 1) / 2) * 16 + n2) * 4 + 2 * ((k1 - 1) & 1);
                host[at + 0] = (float)std::cos(ang);
                host[at + 1] = (float)std::sin(ang);
            }
        f->fft_mel_off = host.size();
        int groups_total = 0;
        for (int mt = 0; mt < f->mel_tiles; ++mt) {
            int lo = 51, hi = -1;                       // 51 groups cover bins 0..203
            for (int grp = 0; grp < 51; ++grp)
                for (int b = 4 * grp; b < 4 * grp + 4 && b <= 200; ++b)
                    for (int m = 16 * mt; m < 16 * mt + 16 && m < cfg->n_mel; ++m)
                        if (f->basis[(size_t)m * NF + b] != 0.f) { lo = std::min(lo, grp); hi = std::max(hi, grp); }
            int cnt = hi >= lo ? hi - lo + 1 : 0;
            if (cnt == 0) lo = 0;
            cnt = (cnt + 3) & ~3;                       // the kernel works in fours: the extra groups carry zero weights and stay
            if (lo + cnt > 52) lo = 52 - cnt;           // inside the 52 groups (208 rows) of the spectrum block
            f->mel_lo[mt] = lo; f->mel_cnt[mt] = cnt; f->mel_off[mt] = groups_total;
            const int stored = std::max(cnt, 24);       // the kernel preloads 24 groups per tile unconditionally (kMelRegs): zero padded
            host.resize(host.size() + (size_t)stored * 64, 0.f);
            for (int e = 0; e < cnt; ++e)                 // [tile][e / 4][lane][e % 4]: four groups' fragments per 16-byte load
                for (int lane = 0; lane < 64; ++lane) {
                    const int g = lane >> 4, m = 16 * mt + (lane & 15), b = 4 * (lo + e) + g;
                    if (m < cfg->n_mel && b <= 200)
                        host[f->fft_mel_off + (((size_t)(groups_total + e) / 4 * 64) + lane) * 4 + (e & 3)] = f->basis[(size_t)m * NF + b];
                }
            groups_total += stored;
        }
        const char* dense = getenv("KWS_FRONTEND_DENSE");      // A/B switch: the dense-DFT kernel also handles 400
        f->use_fft = !(dense && dense[0] == '1');
    }
    hipError_t e = hipMalloc(reinterpret_cast<void**>(&f->d_tables), host.size() * sizeof(float));
    if (e != hipSuccess) { delete f; return hip_fail(e, "hipMalloc(frontend tables)"); }
    e = hipMemcpy(f->d_tables, host.data(), host.size() * sizeof(float), hipMemcpyHostToDevice);
    if (e == hipSuccess) e = hipDeviceSynchronize();
    if (e != hipSuccess) { hipFree(f->d_tables); delete f; return hip_fail(e, "hipMemcpy(frontend tables)"); }
    live_register(f);
    *out = f;
    return KWS_OK;
}

int kws_frontend_destroy(kws_frontend_handle h) {
    if (!h) return KWS_OK;
    live_unregister(h);
    hipDeviceSynchronize();
    if (h->d_tables) hipFree(h->d_tables);
    delete h;
    return KWS_OK;
}

int kws_frontend_mel_basis(kws_frontend_handle h, float* basis_host) {
    if (!h || !basis_host) return fail(KWS_ERR_INVALID_ARGUMENT, "null argument");
    memcpy(basis_host, h->basis.data(), h->basis.size() * sizeof(float));
    return KWS_OK;
}

// the head of a stream-manager iteration that the FFT front-end can take along in its own launch (kws_stream_feed)
struct FrontGate {
    const int16_t* pcm_i16;        // int16 input read in place (chunk is then ignored), or null
    float vad_thres;
    const uint8_t* restart;
    uint8_t *silent, *reset;
    float* next;
    int n_next;
};
static bool frontend_fuses_gate(kws_frontend_handle h, int B, int T) { return h->use_fft && T > 0 && (long long)B * T < (1LL << 31); }

static int frontend_run_impl(kws_frontend_handle h, const float* carry, int n_carry, const float* chunk, int n_chunk, int B,
                             float* mel, void* stream, const FrontGate* gate = nullptr) {
    const int n_samples = n_carry + n_chunk;
    const int T = kws_frontend_frames(&h->cfg, n_samples);
    if (B == 0 || T == 0) return KWS_OK;
    if ((!chunk && !(gate && gate->pcm_i16)) || !mel || (n_carry > 0 && !carry)) return fail(KWS_ERR_INVALID_ARGUMENT, "null pointer argument");
    if ((long long)B * T > (1LL << 36)) return fail(KWS_ERR_UNSUPPORTED, "B*T=%lld frames exceed the grid limit", (long long)B * T);
    kws::FrontendParams p = {};
    p.pcm = chunk; p.carry = n_carry > 0 ? carry : chunk; p.mel = mel;
    if (gate) {
        if (!frontend_fuses_gate(h, B, T)) return fail(KWS_ERR_UNSUPPORTED, "internal: the gate rides only on the FFT front-end");
        p.gate = 1; p.pcm_i16 = gate->pcm_i16; p.vad_thres = gate->vad_thres; p.restart = gate->restart;
        p.silent = gate->silent; p.reset = gate->reset; p.next = gate->next; p.n_next = gate->n_next;
        if (n_carry == 0) p.carry = gate->next;     // never dereferenced (n_carry == 0), only has to be a float pointer
    }
    p.dft = h->d_tables + h->dft_off; p.melw = h->d_tables + h->melw_off;
    p.n_samples = n_samples; p.n_carry = n_carry; p.T = T; p.fft = h->cfg.fft_size; p.hop = h->cfg.hop_size; p.n_mel = h->cfg.n_mel;
    p.nf_tiles = h->nf_tiles; p.mel_tiles = h->mel_tiles; p.kc4 = h->kc4; p.B = B;
    hipError_t e;
    if (h->use_fft && (long long)B * T < (1LL << 31)) {
        p.dft = h->d_tables + h->fft_tw_off; p.melw = h->d_tables + h->fft_mel_off;
        for (int m = 0; m < 4; ++m) { p.mel_lo[m] = h->mel_lo[m]; p.mel_cnt[m] = h->mel_cnt[m]; p.mel_off[m] = h->mel_off[m]; }
        e = kws::launch_mel_fft400(p, B, static_cast<hipStream_t>(stream));
    } else {
        e = kws::launch_mel_frontend(p, B, static_cast<hipStream_t>(stream));
    }
    if (e != hipSuccess) return hip_fail(e, "launch mel_frontend");
    return KWS_OK;
}

int kws_frontend_run(kws_frontend_handle h, const float* pcm, int B, int n_samples, float* mel, void* stream) {
    if (!h) return fail(KWS_ERR_INVALID_ARGUMENT, "handle is null");
    if (B < 0 || n_samples < 0) return fail(KWS_ERR_INVALID_ARGUMENT, "negative dimension");
    return frontend_run_impl(h, nullptr, 0, pcm, n_samples, B, mel, stream);
}

int kws_frontend_run_carry(kws_frontend_handle h, const float* carry, int n_carry, const float* chunk, int n_chunk, int B,
                           float* mel, float* next_carry, int n_next, void* stream) {
    if (!h) return fail(KWS_ERR_INVALID_ARGUMENT, "handle is null");
    if (B < 0 || n_carry < 0 || n_chunk < 0 || n_next < 0) return fail(KWS_ERR_INVALID_ARGUMENT, "negative dimension");
    if (n_next > n_carry + n_chunk) return fail(KWS_ERR_INVALID_ARGUMENT, "n_next=%d exceeds the %d available samples", n_next, n_carry + n_chunk);
    if (n_next > 0 && !next_carry) return fail(KWS_ERR_INVALID_ARGUMENT, "next_carry is null");
    if (B == 0) return KWS_OK;
    if (n_chunk > 0 && !chunk) return fail(KWS_ERR_INVALID_ARGUMENT, "chunk is null");
    if (n_carry > 0 && !carry) return fail(KWS_ERR_INVALID_ARGUMENT, "carry is null");
    if (n_carry + n_chunk >= h->cfg.fft_size) {
        if (!mel) return fail(KWS_ERR_INVALID_ARGUMENT, "mel is null");
        const int rc = frontend_run_impl(h, carry, n_carry, chunk, n_chunk, B, mel, stream);
        if (rc != KWS_OK) return rc;
    }
    if (n_next > 0) {
        hipError_t e = kws::launch_carry_tail(carry ? carry : chunk, n_carry, chunk ? chunk : carry, n_chunk, next_carry, n_next, B,
                                              static_cast<hipStream_t>(stream));
        if (e != hipSuccess) return hip_fail(e, "launch carry_tail");
    }
    return KWS_OK;
}

int kws_stream_create(kws_handle model, kws_frontend_handle frontend, kws_window_handle window, int B, int max_chunk_samples,
                      float vad_thres, const char* label, float* state, uint8_t* restart, kws_stream_handle* out) {
    if (!out) return fail(KWS_ERR_INVALID_ARGUMENT, "out handle pointer is null");
    *out = nullptr;
    if (!model || !frontend || !window || !state || !restart || !label) return fail(KWS_ERR_INVALID_ARGUMENT, "null argument");
    const unsigned long long ms = live_serial(model), fs = live_serial(frontend), ws = live_serial(window);
    if (!ms || !fs || !ws) return fail(KWS_ERR_INVALID_ARGUMENT, "model, front-end or window handle is not alive (destroyed, or not a handle)");
    if (B < 1 || max_chunk_samples < 1) return fail(KWS_ERR_INVALID_ARGUMENT, "bad stream shape B=%d max_chunk_samples=%d", B, max_chunk_samples);
    const int n = (int)strlen(label);
    if (n > 15) return fail(KWS_ERR_INVALID_ARGUMENT, "label longer than 15 digits (the incremental window's matcher has 16 states)");
    for (int i = 0; i < n; ++i)
        if (label[i] < '1' || label[i] > '9') return fail(KWS_ERR_INVALID_ARGUMENT, "label must be digits 1..9, got '%s'", label);
    if (frontend->cfg.n_mel != model->cfg.n_mel)
        return fail(KWS_ERR_INVALID_ARGUMENT, "front-end produces %d mel bins, the model takes %d", frontend->cfg.n_mel, model->cfg.n_mel);
    if (window->B != B || window->C != model->cfg.num_classes)
        return fail(KWS_ERR_INVALID_ARGUMENT, "window was created for B=%d C=%d, stream needs B=%d C=%d", window->B, window->C, B,
                    model->cfg.num_classes);
    const int fft = frontend->cfg.fft_size;
    const int tmax = kws_frontend_frames(&frontend->cfg, max_chunk_samples + fft - 1);
    if (tmax > window->tmax)
        return fail(KWS_ERR_INVALID_ARGUMENT, "chunks of %d samples give up to %d frames, the window holds %d per chunk", max_chunk_samples,
                    tmax, window->tmax);
    if (window_inc_lds_bytes(tmax, window->nq) > kWindowIncLdsMax)
        return fail(KWS_ERR_UNSUPPORTED, "chunks of up to %d frames with a %d-chunk window need %zu bytes of LDS in the incremental window step "
                    "(limit %zu): use shorter chunks", tmax, window->nq, window_inc_lds_bytes(tmax, window->nq), kWindowIncLdsMax);
    kws_stream* s = new (std::nothrow) kws_stream();
    if (!s) return fail(KWS_ERR_OUT_OF_MEMORY, "host allocation failed");
    s->model_serial = ms; s->fe_serial = fs; s->win_serial = ws;
    s->model = model; s->fe = frontend; s->win = window; s->B = B; s->max_chunk = max_chunk_samples; s->tmax = tmax;
    s->vad_thres = vad_thres; s->state = state; s->restart = restart;
    memcpy(s->label, label, n);
    const size_t carry_bytes = (size_t)B * (fft - 1) * sizeof(float);
    hipError_t e = hipMalloc(reinterpret_cast<void**>(&s->carry[0]), carry_bytes);
    if (e == hipSuccess) e = hipMalloc(reinterpret_cast<void**>(&s->carry[1]), carry_bytes);
    if (e != hipSuccess) { kws_stream_destroy(s); return hip_fail(e, "hipMalloc(stream buffers)"); }
    // One chunk's intermediates come out of the model handle's staging block: sized here, so a feed never allocates.  The
    // widened copy of int16 PCM is read by the dense-DFT front-end only (the 400-point FFT reads int16 in place); with the
    // FFT front-end it is written just by the gate of a chunk that completes no frame (< fft samples in total).
    {
        auto take = [&](size_t bytes) { const size_t at = s->stage_bytes; s->stage_bytes += (bytes + 255) & ~size_t(255); return at; };
        const size_t tm = (size_t)(tmax > 0 ? tmax : 1);
        const bool fused_gate = frontend->use_fft && (long long)B * tm < (1LL << 31);        // frontend_fuses_gate for every chunk with a frame
        s->off_pcm_f32 = take((size_t)B * (fused_gate ? std::min(max_chunk_samples, fft - 1) : max_chunk_samples) * sizeof(float));
        s->off_mel = take((size_t)B * tm * model->cfg.n_mel * sizeof(float));
        s->off_softmax = take((size_t)B * tm * model->cfg.num_classes * sizeof(float));
        s->off_silent = take((size_t)B);
        s->off_reset = take((size_t)B);
    }
    int rc = KWS_OK;
    {
        BusyGuard busy(model->in_call);
        if (!busy.owned) rc = fail(KWS_ERR_BUSY, "kws_stream_create: another host thread is inside a call on the model handle");
        else if (s->stage_bytes > model->stage.bytes) {
            // grows only here; the old block may still be read by a feed in flight
            hipError_t es = hipDeviceSynchronize();
            if (es == hipSuccess && model->stage.base) { hipFree(model->stage.base); model->stage.base = nullptr; model->stage.bytes = 0; }
            if (es == hipSuccess) es = hipMalloc(reinterpret_cast<void**>(&model->stage.base), s->stage_bytes);
            if (es != hipSuccess) { model->stage.base = nullptr; rc = hip_fail(es, "hipMalloc(stream staging)"); }
            else { model->stage.bytes = s->stage_bytes; ++model->scratch_allocs; }
        }
    }
    if (rc == KWS_OK) rc = kws_reserve(model, B, tmax);            // the GRU step of a chunk never allocates afterwards
    if (rc == KWS_OK) rc = window_bind_label(window, s->label);      // the window's summaries are built for this label
    if (rc != KWS_OK) { kws_stream_destroy(s); return rc; }
    *out = s;
    return KWS_OK;
}

int kws_stream_destroy(kws_stream_handle h) {
    if (!h) return KWS_OK;
    hipDeviceSynchronize();
    for (float* p : {h->carry[0], h->carry[1]}) if (p) hipFree(p);
    delete h;
    return KWS_OK;
}

int kws_stream_reset(kws_stream_handle h) {
    if (!h) return fail(KWS_ERR_INVALID_ARGUMENT, "handle is null");
    h->n_carry = 0;
    return KWS_OK;
}

static int stream_feed_locked(kws_stream_handle h, const void* pcm, int n, int pcm_int16, int32_t* hit, hipStream_t st);

int kws_stream_feed(kws_stream_handle h, const void* pcm, int n, int pcm_int16, int32_t* hit, void* stream) {
    if (!h) return fail(KWS_ERR_INVALID_ARGUMENT, "handle is null");
    if (n < 0 || n > h->max_chunk) return fail(KWS_ERR_INVALID_ARGUMENT, "chunk of %d samples outside [0,%d]", n, h->max_chunk);
    if (!hit || (!pcm && n > 0)) return fail(KWS_ERR_INVALID_ARGUMENT, "null pointer argument");
    if (live_serial(h->model) != h->model_serial || live_serial(h->fe) != h->fe_serial || live_serial(h->win) != h->win_serial)
        return fail(KWS_ERR_INVALID_ARGUMENT, "the model, front-end or window this stream was created on has been destroyed");
    hipStream_t st = static_cast<hipStream_t>(stream);
    if (n == 0) {            // detector.py:164-166: an empty read is skipped before anything else happens
        KWS_HIP(hipMemsetAsync(hit, 0, (size_t)h->B * sizeof(int32_t), st));
        return KWS_OK;
    }
    // The feed holds the model handle from its first launch to its last (the staging and the seams are the handle's), and
    // orders `stream` behind the handle's previous call when that ran on another stream.
    kws_model* model = h->model;
    BusyGuard busy(model->in_call);
    if (!busy.owned)
        return fail(KWS_ERR_BUSY, "kws_stream_feed: another host thread is inside a call on the model handle (one thread at a time per "
                    "handle; stream managers that run concurrently need a model handle each)");
    if (h->stage_bytes > model->stage.bytes) return fail(KWS_ERR_INVALID_ARGUMENT, "internal: the model handle's staging block is smaller than this stream's");
    int rc = call_enter(model, st);
    if (rc != KWS_OK) return rc;
    h->pcm_f32 = reinterpret_cast<float*>(model->stage.base + h->off_pcm_f32);
    h->mel = reinterpret_cast<float*>(model->stage.base + h->off_mel);
    h->softmax = reinterpret_cast<float*>(model->stage.base + h->off_softmax);
    h->silent = reinterpret_cast<uint8_t*>(model->stage.base + h->off_silent);
    h->reset = reinterpret_cast<uint8_t*>(model->stage.base + h->off_reset);
    rc = stream_feed_locked(h, pcm, n, pcm_int16, hit, st);
    const int rl = call_leave(model, st);
    return rc != KWS_OK ? rc : rl;
}

// one iteration with the model handle held and `st` ordered (kws_stream_feed above)
static int stream_feed_locked(kws_stream_handle h, const void* pcm, int n, int pcm_int16, int32_t* hit, hipStream_t st) {
    const kws_frontend_config& fc = h->fe->cfg;
    const int fft = fc.fft_size, hop = fc.hop_size, B = h->B;
    const int total = h->n_carry + n;
    const float* chunk = pcm_int16 ? h->pcm_f32 : static_cast<const float*>(pcm);
    const float* carry = h->carry[h->cur];
    float* next = h->carry[h->cur ^ 1];
    if (total < fft) {
        // Not a full frame yet.  The reference still runs the whole iteration on such a chunk (detector.py:168-209): vad ->
        // clean_state() + prob_queue.clear() when silent, the samples are carried (:179-183 keeps all of them), sess.run over
        // zero frames returns the state unchanged and an empty softmax, which takes a slot of the window (:195) before the
        // windowed decode (:197-201).
        hipError_t e = kws::launch_vad_gate(pcm, pcm_int16, B, n, h->vad_thres, h->pcm_f32, h->restart, h->silent, h->reset,
                                            h->n_carry ? carry : nullptr, h->n_carry, next, total, st);
        if (e != hipSuccess) return hip_fail(e, "launch vad_gate");
        int rc = step_impl(h->model, nullptr, h->state, nullptr, nullptr, h->state, nullptr, h->reset, nullptr, nullptr, 0.f, B, 0, st, nullptr, true);
        if (rc != KWS_OK) return rc;
        rc = kws_window_step_incremental(h->win, nullptr, 0, h->silent, h->label, hit, h->restart, st);
        if (rc != KWS_OK) return rc;
        h->n_carry = total; h->cur ^= 1;
        return KWS_OK;
    }
    const int keep = (total - fft) % hop + (fft - hop);                                  // detector.py:181-182
    const int T = kws_frontend_frames(&fc, total);
    int rc;
    if (frontend_fuses_gate(h->fe, B, T)) {
        // ONE launch: vad + masks and the next carry ride on the FFT front-end, which reads the PCM -- int16 as it is -- in place
        FrontGate gate = {pcm_int16 ? static_cast<const int16_t*>(pcm) : nullptr, h->vad_thres, h->restart, h->silent, h->reset, next, keep};
        rc = frontend_run_impl(h->fe, h->n_carry ? carry : nullptr, h->n_carry, pcm_int16 ? nullptr : chunk, n, B, h->mel, st, &gate);
        if (rc != KWS_OK) return rc;
    } else {
        // other frame lengths: one pass over the new chunk (int16 -> float, vad + masks, next carry), then the dense-DFT kernel
        hipError_t e = kws::launch_vad_gate(pcm, pcm_int16, B, n, h->vad_thres, h->pcm_f32, h->restart, h->silent, h->reset,
                                            h->n_carry ? carry : nullptr, h->n_carry, next, keep, st);
        if (e != hipSuccess) return hip_fail(e, "launch vad_gate");
        rc = kws_frontend_run_carry(h->fe, h->n_carry ? carry : nullptr, h->n_carry, chunk, n, B, h->mel, nullptr, 0, st);
        if (rc != KWS_OK) return rc;
    }
    rc = window_bind_label(h->win, h->label);          // (bound at kws_stream_create; refuses a window that went on with another label)
    if (rc != KWS_OK) return rc;
    if (step_takes_window(h->model, B, T, h->win->nq)) {
        // THREE launches per chunk (two for the bf16 stack): the decode-window step (prob_queue.add, ctc_decode2 over the
        // window, ctc_predict, clear + restart on a hit: detector.py:195-209) rides at the end of the last layer's launch, on
        // the frame words its flush has just produced -- no softmax round trip, no fourth launch.  The window's threshold
        // is the fused decoder's (ctc_decode2's frame rule, utils/prediction.py:74)
        const kws::WindowTail wt = window_tail_params(h->win, h->silent, hit, h->restart);
        rc = step_impl(h->model, h->mel, h->state, nullptr, nullptr, h->state, nullptr, h->reset, nullptr, nullptr, h->win->thres, B, T, st, &wt, true);
        if (rc != KWS_OK) return rc;
    } else {
        rc = step_impl(h->model, h->mel, h->state, nullptr, h->softmax, h->state, nullptr, h->reset, nullptr, nullptr, 0.f, B, T, st, nullptr, true);
        if (rc != KWS_OK) return rc;
        rc = kws_window_step_incremental(h->win, h->softmax, T, h->silent, h->label, hit, h->restart, st);
        if (rc != KWS_OK) return rc;
    }
    h->n_carry = keep; h->cur ^= 1;
    return KWS_OK;
}

int kws_ctc_decode(int kind, const float* softmax, const int32_t* lengths, int B, int T, int C, int lockout,
                   float thres, float loose_thres, int32_t* words, int32_t* counts, int max_words, void* stream) {
    if (kind != KWS_DECODE && kind != KWS_DECODE2 && kind != KWS_DECODE_STRICT)
        return fail(KWS_ERR_INVALID_ARGUMENT, "unknown decode kind %d", kind);
    if (B < 0 || T < 0 || max_words < 0) return fail(KWS_ERR_INVALID_ARGUMENT, "negative dimension");
    if (C < 3 || C > 64) return fail(KWS_ERR_INVALID_ARGUMENT, "classnum=%d out of range [3,64]", C);
    if (kind == KWS_DECODE && C < 5) return fail(KWS_ERR_INVALID_ARGUMENT, "ctc_decode slices columns 1:5 and needs classnum >= 5, got %d", C);
    if (lockout < 1 && kind != KWS_DECODE2) return fail(KWS_ERR_INVALID_ARGUMENT, "lockout must be >= 1, got %d", lockout);
    if (B == 0) return KWS_OK;
    if (!counts || (!words && max_words > 0) || (!softmax && T > 0))
        return fail(KWS_ERR_INVALID_ARGUMENT, "null pointer argument");
    hipError_t e = kws::launch_ctc_decode(kind, softmax, lengths, B, T, C, lockout, thres, loose_thres, words,
                                          counts, max_words, static_cast<hipStream_t>(stream));
    if (e != hipSuccess) return hip_fail(e, "launch ctc_decode");
    return KWS_OK;
}

int kws_ctc_predict(const int32_t* words, const int32_t* counts, int B, int max_words, const char* label,
                    int32_t* hit, void* stream) {
    if (!label) return fail(KWS_ERR_INVALID_ARGUMENT, "label is null");
    const int n = (int)strlen(label);
    if (n > 16) return fail(KWS_ERR_INVALID_ARGUMENT, "label longer than 16 digits");
    int32_t digits[16] = {0};
    for (int i = 0; i < n; ++i) {
        if (label[i] < '1' || label[i] > '9') return fail(KWS_ERR_INVALID_ARGUMENT, "label must be digits 1..9, got '%s'", label);
        digits[i] = label[i] - '0';
    }
    if (B < 0 || max_words < 0) return fail(KWS_ERR_INVALID_ARGUMENT, "negative dimension");
    if (B == 0) return KWS_OK;
    if (!counts || !hit || (!words && max_words > 0)) return fail(KWS_ERR_INVALID_ARGUMENT, "null pointer argument");
    hipStream_t st = static_cast<hipStream_t>(stream);
    // digits travel by value inside the launcher (kernel argument), no device allocation
    hipError_t e = kws::launch_ctc_predict(words, counts, B, max_words, digits, n, hit, st);
    if (e != hipSuccess) return hip_fail(e, "launch ctc_predict");
    return KWS_OK;
}

int kws_vad(const float* pcm, int B, int N, float thres, uint8_t* speech, float* abs_sum, void* stream) {
    if (B < 0 || N < 0) return fail(KWS_ERR_INVALID_ARGUMENT, "negative dimension");
    if (B == 0) return KWS_OK;
    if (!speech || (!pcm && N > 0)) return fail(KWS_ERR_INVALID_ARGUMENT, "null pointer argument");
    hipError_t e = kws::launch_vad(pcm, B, N, thres, speech, abs_sum, static_cast<hipStream_t>(stream));
    if (e != hipSuccess) return hip_fail(e, "launch vad");
    return KWS_OK;
}

int kws_octbit_matmul(const float* x, const int8_t* Wq, float scale_w, const float* bias, float* out, int A,
                      int K, int N, int per_row_scale, void* stream) {
    // preconditions of octbit/octbit_mat_mul_op.cc:41-46,61-73 as error codes
    if (!(scale_w > 0.f)) return fail(KWS_ERR_INVALID_ARGUMENT, "scale has to be positive");
    if (A < 0 || K < 0 || N < 0) return fail(KWS_ERR_INVALID_ARGUMENT, "negative dimension");
    if (K % 64 != 0) return fail(KWS_ERR_INVALID_ARGUMENT, "K=%d must be a multiple of 64", K);
    if (A == 0 || N == 0) return KWS_OK;
    if (K == 0) return fail(KWS_ERR_INVALID_ARGUMENT, "K must be positive");
    if (!x || !Wq || !bias || !out) return fail(KWS_ERR_INVALID_ARGUMENT, "null pointer argument");
    hipStream_t st = static_cast<hipStream_t>(stream);
    // The 2A+2 floats of activation ranges come from a block cached per (device, stream): the reference's Compute allocates its
    // temporaries per call (octbit_mat_mul_op.cc:49, re-entrant); here a call neither allocates nor frees once its stream has
    // seen a call of this size.  Calls on one stream are ordered by the stream; two host threads that share a stream are
    // serialised on the block's own mutex for the duration of the two launches.
    int dev = 0;
    KWS_HIP(hipGetDevice(&dev));
    OctbitWorkspace* w = nullptr;
    {
        std::lock_guard<std::mutex> lock(g_octbit_ws_mutex);
        std::unique_ptr<OctbitWorkspace>& slot = g_octbit_ws[std::make_pair(dev, (const void*)st)];
        if (!slot) slot.reset(new OctbitWorkspace());
        w = slot.get();
    }
    std::lock_guard<std::mutex> use(w->mutex);
    const size_t need = (size_t)(2 * A + 2);
    if (need > w->floats) {
        // stream-ordered: the old block is released behind the launches that still read it
        float* grown = nullptr;
        KWS_HIP(hipMallocAsync(reinterpret_cast<void**>(&grown), need * sizeof(float), st));
        if (w->p) (void)hipFreeAsync(w->p, st);
        w->p = grown;
        w->floats = need;
    }
    hipError_t e = kws::launch_octbit_matmul(x, Wq, scale_w, bias, out, A, K, N, per_row_scale, w->p, st);
    if (e != hipSuccess) return hip_fail(e, "launch octbit_matmul");
    return KWS_OK;
}

int kws_octbit_quantize(const float* W, int K, int N, int8_t* Wq, float* scale, float* bias) {
    if (!W || !Wq || !scale || !bias) return fail(KWS_ERR_INVALID_ARGUMENT, "null pointer argument");
    if (K <= 0 || N <= 0) return fail(KWS_ERR_INVALID_ARGUMENT, "K and N must be positive");
    // octbit/octbit_graph.py:196-204: scale = max|W|/127 in double, np.round = half-to-even
    float mx = W[0], mn = W[0];
    for (size_t i = 0; i < (size_t)K * N; ++i) { mx = std::max(mx, W[i]); mn = std::min(mn, W[i]); }
    const double nmax = std::max(std::fabs((double)mx), std::fabs((double)mn));
    if (!(nmax > 0.0)) return fail(KWS_ERR_INVALID_ARGUMENT, "weight matrix is all zero: scale would be 0");
    // numpy: float32 array / python float -> float32 array divided by a float32-cast scalar
    const float sc32 = (float)(nmax / 127.0);
    for (int j = 0; j < N; ++j) bias[j] = 0.f;
    std::vector<double> b(N, 0.0);
    for (int i = 0; i < K; ++i)
        for (int j = 0; j < N; ++j) {
            const float qf = std::nearbyintf(W[(size_t)i * N + j] / sc32);
            Wq[(size_t)j * K + i] = (int8_t)qf;
            b[j] += (double)qf * 127.0;
        }
    for (int j = 0; j < N; ++j) bias[j] = (float)b[j];
    *scale = (float)(nmax / 127.0);
    return KWS_OK;
}


int kws_last_launch(kws_handle h, int slot, char* buf, size_t n) {
    if (!h || !buf || n == 0) return fail(KWS_ERR_INVALID_ARGUMENT, "null handle / buffer");
    if (slot < 0 || slot >= h->cfg.num_layers) return fail(KWS_ERR_INVALID_ARGUMENT, "slot %d out of range [0,%d)", slot, h->cfg.num_layers);
    snprintf(buf, n, "%s", h->launch_name(slot).c_str());
    return KWS_OK;
}

}  // extern "C"

// ---- kws_selftest ---------------------------------------------------------------------------------------------------
// The kernels depend on things the compiler does not check: hand-placed wait states around inline-asm MFMAs (gru_device.h),
// an internal LLVM option for gru_bf16.hip (csrc/Makefile).  A build by another ROCm can therefore be silently wrong; the
// GPU test-suite catches that, a deployment has no test-suite.  So the library carries its own known answers: a plain
// double-precision host loop of the cell (below; the published TF-1.x GRUCell, models/rnn_ctc.py:179-185,228-243) and
// TensorFlow's own unit-test constants for it (rnn_cell_test.py testGRUCell / testMultiRNNCell: all kernels 0.5, gate bias
// 1, candidate bias 0, x = 1, h = 0.1 -> 0.175991, 0.156736 for three inputs, 0.13248 from the second stacked cell).
namespace {


// (mel [B,T,I], state [L,B,H]) -> (logits [B,T,C], state'), canonical blob layout of kws_weights_nbytes; double throughout
void host_forward(const kws_config& c, const float* blob, const float* mel, const float* st0, int B, int T,
                  std::vector<double>& logits, std::vector<double>& state) {
    const int H = c.hidden, L = c.num_layers, C = c.num_classes;
    state.assign(st0, st0 + (size_t)L * B * H);
    logits.assign((size_t)B * T * C, 0.0);
    std::vector<double> x, g(2 * H), cand(H), hn(H);
    for (int b = 0; b < B; ++b)
        for (int t = 0; t < T; ++t) {
            x.assign(mel + ((size_t)b * T + t) * c.n_mel, mel + ((size_t)b * T + t + 1) * c.n_mel);
            const float* p = blob;
            for (int l = 0; l < L; ++l) {
                const int I = (int)x.size();
                const float *Wg = p, *bg = Wg + (size_t)(I + H) * 2 * H, *Wc = bg + 2 * H, *bc = Wc + (size_t)(I + H) * H;
                double* h = &state[((size_t)l * B + b) * H];
                for (int j = 0; j < 2 * H; ++j) {
                    double a = bg[j];
                    for (int k = 0; k < I; ++k) a += x[k] * Wg[(size_t)k * 2 * H + j];
                    for (int k = 0; k < H; ++k) a += h[k] * Wg[(size_t)(I + k) * 2 * H + j];
                    g[j] = 1.0 / (1.0 + std::exp(-a));                          // [r | u]
                }
                for (int j = 0; j < H; ++j) {
                    double a = bc[j];
                    for (int k = 0; k < I; ++k) a += x[k] * Wc[(size_t)k * H + j];
                    for (int k = 0; k < H; ++k) a += g[k] * h[k] * Wc[(size_t)(I + k) * H + j];   // r (.) h before the matmul
                    cand[j] = std::tanh(a);
                }
                for (int j = 0; j < H; ++j) hn[j] = g[H + j] * h[j] + (1.0 - g[H + j]) * cand[j];
                std::copy(hn.begin(), hn.end(), h);
                x = hn;
                p = bc + H;
            }
            const float *Wfc = p, *bfc = Wfc + (size_t)H * C;
            for (int k = 0; k < C; ++k) {
                double a = bfc[k];
                for (int j = 0; j < H; ++j) a += x[j] * Wfc[(size_t)j * C + k];
                if (c.use_relu) { a = std::max(a, 0.0); if (c.value_clip > 0) a = std::min(a, 20.0); }
                logits[((size_t)b * T + t) * C + k] = a;
            }
        }
}

struct DevBuf {
    void* p = nullptr;
    ~DevBuf() { if (p) hipFree(p); }
    hipError_t alloc(size_t n) { return hipMalloc(&p, n ? n : 4); }
};

// one case through kws_step on a temporary handle; *err_state / *err_logit = max abs deviation from the host loop
int selftest_case(const kws_config& cfg, int kernel_kind, const std::vector<float>& blob, const std::vector<float>& mel,
                  const std::vector<float>& st0, int B, int T, double* err_logit, double* err_state, std::vector<float>* state_out,
                  std::string* kernels) {
    const int H = cfg.hidden, L = cfg.num_layers, C = cfg.num_classes;
    kws_handle m = nullptr;
    int rc = kws_create(&cfg, blob.data(), blob.size() * sizeof(float), &m);
    if (rc != KWS_OK) return rc;
    rc = kws_set_kernel(m, kernel_kind);
    DevBuf d_mel, d_st, d_lg;
    std::vector<float> lg((size_t)B * T * C), st((size_t)L * B * H);
    auto hip = [&](hipError_t e, const char* what) { if (e != hipSuccess && rc == KWS_OK) rc = hip_fail(e, what); };
    if (rc == KWS_OK) {
        hip(d_mel.alloc(mel.size() * 4), "selftest hipMalloc");
        hip(d_st.alloc(st.size() * 4), "selftest hipMalloc");
        hip(d_lg.alloc(lg.size() * 4), "selftest hipMalloc");
    }
    if (rc == KWS_OK) {
        hip(hipMemcpy(d_mel.p, mel.data(), mel.size() * 4, hipMemcpyHostToDevice), "selftest upload");
        hip(hipMemcpy(d_st.p, st0.data(), st0.size() * 4, hipMemcpyHostToDevice), "selftest upload");
        hip(hipDeviceSynchronize(), "selftest sync");
    }
    if (rc == KWS_OK)
        rc = kws_step(m, static_cast<const float*>(d_mel.p), static_cast<const float*>(d_st.p), static_cast<float*>(d_lg.p), nullptr,
                      static_cast<float*>(d_st.p), nullptr, nullptr, nullptr, nullptr, 0.4f, B, T, nullptr);
    if (rc == KWS_OK) {
        hip(hipDeviceSynchronize(), "selftest kernels");
        hip(hipMemcpy(lg.data(), d_lg.p, lg.size() * 4, hipMemcpyDeviceToHost), "selftest download");
        hip(hipMemcpy(st.data(), d_st.p, st.size() * 4, hipMemcpyDeviceToHost), "selftest download");
    }
    if (rc == KWS_OK) rc = kws_poll_error(m);
    if (rc == KWS_OK && kernels) {
        kernels->clear();
        for (int l = 0; l < L; ++l)
            if (m->launch_tag[l].family != kws_model::kNone) *kernels += (kernels->empty() ? "" : " + ") + m->launch_name(l);
    }
    const std::string keep = g_last_error;
    kws_destroy(m);
    if (rc != KWS_OK) { g_last_error = keep; return rc; }
    std::vector<double> want_l, want_s;
    host_forward(cfg, blob.data(), mel.data(), st0.data(), B, T, want_l, want_s);
    double el = 0.0, es = 0.0;
    for (size_t i = 0; i < lg.size(); ++i) { const double d = std::fabs(lg[i] - want_l[i]); el = (d > el || d != d) ? d : el; }
    for (size_t i = 0; i < st.size(); ++i) { const double d = std::fabs(st[i] - want_s[i]); es = (d > es || d != d) ? d : es; }
    *err_logit = el; *err_state = es;
    if (state_out) *state_out = st;
    return KWS_OK;
}

// deterministic pseudo-random floats in [-1, 1) (no <random>: identical on every libstdc++)
struct Lcg {
    uint64_t s;
    float next() { s = s * 6364136223846793005ull + 1442695040888963407ull; return (float)((double)(s >> 11) / 9007199254740992.0 * 2.0 - 1.0); }
};

}  // namespace

extern "C" int kws_selftest(kws_handle h) {
    if (!h) return fail(KWS_ERR_INVALID_ARGUMENT, "handle is null");
    if (g_in_selftest) return KWS_OK;
    struct Guard { Guard() { g_in_selftest = true; } ~Guard() { g_in_selftest = false; } } guard;
    const kws_config cfg = h->cfg;
    const int H = cfg.hidden, L = cfg.num_layers, C = cfg.num_classes, I0 = cfg.n_mel;
    const size_t nfl = weights_floats(&cfg);
    // tolerances: what the arithmetic of each precision leaves on these two cases (fp32: observed <= 4e-6 / 1e-7)
    double tol_rand_logit, tol_rand_state, tol_kat;
    switch (cfg.precision) {
        case KWS_BF16: tol_rand_logit = 6e-2; tol_rand_state = 2e-2; tol_kat = 2e-3; break;
        case KWS_INT8: tol_rand_logit = -1.0; tol_rand_state = -1.0; tol_kat = 2e-2; break;   // int8: known answers only
        default:       tol_rand_logit = 5e-5; tol_rand_state = 2e-5; tol_kat = 1e-6; break;   // fp32 and the f16x3 split
    }
    std::vector<int> kinds;
    if (cfg.precision == KWS_FP32) {
        bool res_ok = true;
        for (const auto& Ld : h->layers) res_ok &= Ld.resident_ok;
        if (res_ok) kinds.push_back(KWS_KERNEL_RESIDENT);
        kinds.push_back(KWS_KERNEL_GENERIC);
    } else {
        kinds.push_back(KWS_KERNEL_AUTO);
    }
    auto layer_off = [&](int l) { size_t o = 0; int in = I0; for (int k = 0; k < l; ++k) { o += (size_t)(in + H) * 3 * H + 3 * H; in = H; } return o; };
    for (int kind : kinds) {
        std::string kernels;
        // (1) TensorFlow's published constants, the 2-unit test cell embedded in this shape: units 0,1 and inputs 0..n_in-1 live
        for (int n_in = 2; n_in <= 3 && n_in <= I0; ++n_in) {
            std::vector<float> blob(nfl, 0.f);
            for (int l = 0; l < L; ++l) {
                const int in = l == 0 ? I0 : H, live = l == 0 ? n_in : 2;
                float* Wg = blob.data() + layer_off(l);
                float* bg = Wg + (size_t)(in + H) * 2 * H;
                float* Wc = bg + 2 * H;
                for (int j = 0; j < 2 * H; ++j) bg[j] = 1.f;
                for (int r = 0; r < live + 2; ++r) {
                    const int row = r < live ? r : in + (r - live);
                    for (int u = 0; u < 2; ++u) {
                        Wg[(size_t)row * 2 * H + u] = 0.5f; Wg[(size_t)row * 2 * H + H + u] = 0.5f; Wc[(size_t)row * H + u] = 0.5f;
                    }
                }
            }
            float* Wfc = blob.data() + layer_off(L);
            Wfc[0 * C + 0] = 1.f; Wfc[1 * C + 1] = 1.f;
            const int B = 19, T = 1;
            std::vector<float> mel((size_t)B * T * I0, 0.f), st0((size_t)L * B * H, 0.f), st;
            for (int b = 0; b < B; ++b) {
                for (int k = 0; k < n_in; ++k) mel[(size_t)b * I0 + k] = 1.f;
                for (int l = 0; l < L; ++l) st0[((size_t)l * B + b) * H + 0] = st0[((size_t)l * B + b) * H + 1] = 0.1f;
            }
            double el, es;
            const int rc = selftest_case(cfg, kind, blob, mel, st0, B, T, &el, &es, &st, &kernels);
            if (rc != KWS_OK) return rc;
            const double first = n_in == 2 ? 0.175991 : 0.156736;
            double worst = 0.0;
            for (int b = 0; b < B; ++b) {
                for (int u = 0; u < 2; ++u) worst = std::max(worst, std::fabs(st[(size_t)b * H + u] - first));
                if (L >= 2 && n_in == 2) for (int u = 0; u < 2; ++u) worst = std::max(worst, std::fabs(st[((size_t)B + b) * H + u] - 0.13248));
                if (cfg.precision != KWS_INT8)
                    for (int j = 2; j < H; ++j) if (st[(size_t)b * H + j] != 0.f) worst = 1.0;     // dead units stay exactly 0
            }
            if (!(worst <= tol_kat + 1e-6))
                return fail(KWS_ERR_HIP, "kws_selftest: %s returns TensorFlow's published GRUCell constant (%g) with error %.3g (tolerance %.1g). "
                            "This build (%s) computes wrong results on this device: rebuild with the ROCm release it was validated on, "
                            "or run the repository's GPU tests.", kernels.c_str(), first, worst, tol_kat, kws_version());
        }
        // (2) 8 random frames of 19 streams against the host double-precision loop
        if (tol_rand_logit > 0) {
            Lcg rng{0x9e3779b97f4a7c15ull + (uint64_t)kind};
            std::vector<float> blob(nfl);
            for (int l = 0; l < L; ++l) {
                const int in = l == 0 ? I0 : H;
                float* Wg = blob.data() + layer_off(l);
                float* bg = Wg + (size_t)(in + H) * 2 * H;
                float* Wc = bg + 2 * H;
                float* bc = Wc + (size_t)(in + H) * H;
                const float ag = std::sqrt(6.f / (in + H + 2 * H)), ac = std::sqrt(6.f / (in + H + H));
                for (size_t i = 0; i < (size_t)(in + H) * 2 * H; ++i) Wg[i] = ag * rng.next();
                for (int j = 0; j < 2 * H; ++j) bg[j] = 1.f + 0.3f * rng.next();
                for (size_t i = 0; i < (size_t)(in + H) * H; ++i) Wc[i] = ac * rng.next();
                for (int j = 0; j < H; ++j) bc[j] = 0.3f * rng.next();
            }
            float* Wfc = blob.data() + layer_off(L);
            for (int i = 0; i < H * C + C; ++i) Wfc[i] = rng.next();
            const int B = 19, T = 8;
            std::vector<float> mel((size_t)B * T * I0), st0((size_t)L * B * H);
            for (auto& v : mel) v = 2.f * std::fabs(rng.next());
            for (auto& v : st0) v = 0.5f * rng.next();
            double el, es;
            const int rc = selftest_case(cfg, kind, blob, mel, st0, B, T, &el, &es, nullptr, &kernels);
            if (rc != KWS_OK) return rc;
            if (!(el <= tol_rand_logit && es <= tol_rand_state))
                return fail(KWS_ERR_HIP, "kws_selftest: %s differs from the host double-precision loop on 19 streams x 8 frames: max |dlogit| "
                            "%.3g (tolerance %.1g), max |dstate| %.3g (tolerance %.1g). This build (%s) computes wrong results on this "
                            "device: rebuild with the ROCm release it was validated on, or run the repository's GPU tests.",
                            kernels.c_str(), el, tol_rand_logit, es, tol_rand_state, kws_version());
        }
    }
    return KWS_OK;
}

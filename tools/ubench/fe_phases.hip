// Phase timing of mel_fft400_kernel (s_memtime stamps per wave): where does a wave's lifetime go?
// build: hipcc --offload-arch=gfx950 -O3 -std=c++17 -fno-slp-vectorize -DKWS_FE_TIMING -I include -I keyword_spotting_amd/csrc tools/ubench/fe_phases.hip
#define KWS_FE_TIMING 1
#include "../../keyword_spotting_amd/csrc/fft_frontend.hip"
#include <algorithm>
#include <cmath>
#include <cstdio>
#include <vector>
int main() {
    const int B = 4096, T = 22, N = 3840, n_mel = 40;
    std::vector<float> pcm((size_t)B * N), tw(12 * 16 * 2), melw(52 * 64);
    for (size_t i = 0; i < pcm.size(); ++i) pcm[i] = (float)((i * 2654435761u >> 8) & 0xffff) / 65536.f - 0.5f;
    for (int k1 = 1; k1 <= 12; ++k1)
        for (int n2 = 0; n2 < 16; ++n2) {
            tw[((k1 - 1) * 16 + n2) * 2] = (float)cos(6.283185307179586 * n2 * k1 / 400.0);
            tw[((k1 - 1) * 16 + n2) * 2 + 1] = (float)sin(6.283185307179586 * n2 * k1 / 400.0);
        }
    for (auto& v : melw) v = 0.01f;
    float *d_pcm, *d_tw, *d_melw, *d_mel; long long* d_t;
    const size_t nw = (size_t)(B * T / 16) * 4;
    hipMalloc(&d_pcm, pcm.size() * 4); hipMalloc(&d_tw, tw.size() * 4); hipMalloc(&d_melw, melw.size() * 4);
    hipMalloc(&d_mel, (size_t)B * T * n_mel * 4); hipMalloc(&d_t, nw * 8 * 8);
    hipMemcpy(d_pcm, pcm.data(), pcm.size() * 4, hipMemcpyHostToDevice);
    hipMemcpy(d_tw, tw.data(), tw.size() * 4, hipMemcpyHostToDevice);
    hipMemcpy(d_melw, melw.data(), melw.size() * 4, hipMemcpyHostToDevice);
    kws::FrontendParams p = {};
    p.pcm = d_pcm; p.carry = d_pcm; p.mel = d_mel; p.dft = d_tw; p.melw = d_melw; p.timing = d_t;
    p.n_samples = N; p.n_carry = 0; p.T = T; p.fft = 400; p.hop = 160; p.n_mel = n_mel; p.mel_tiles = 3; p.B = B;
    const int lo[3] = {2, 9, 27}, cnt[3] = {8, 20, 24}, off[3] = {0, 8, 28};   // the runs of the 40-filter basis
    for (int m = 0; m < 3; ++m) { p.mel_lo[m] = lo[m]; p.mel_cnt[m] = cnt[m]; p.mel_off[m] = off[m]; }
    for (int it = 0; it < 3; ++it) kws::launch_mel_fft400(p, B, nullptr);
    hipDeviceSynchronize();
    std::vector<long long> t(nw * 8);
    hipMemcpy(t.data(), d_t, t.size() * 8, hipMemcpyDeviceToHost);
    const char* names[7] = {"loads land", "stage 1 math + LDS store", "barrier 1", "stage 2 (LDS read, FFT16, |X|, A loads issued)", "barrier 2",
                            "spectrum store + barrier 3", "mel MFMAs + store"};
    long long tmin = t[0], tmax = 0;
    for (size_t i = 0; i < nw; ++i) { tmin = std::min(tmin, t[i * 8]); tmax = std::max(tmax, t[i * 8 + 7]); }
    printf("kernel span %lld ticks, %zu waves\n", tmax - tmin, nw);
    for (int wv = 0; wv < 4; ++wv) {
        printf("wave %d:", wv);
        double life = 0;
        for (int ph = 0; ph < 7; ++ph) {
            double s = 0; size_t c = 0;
            for (size_t i = wv; i < nw; i += 4) { s += (double)(t[i * 8 + ph + 1] - t[i * 8 + ph]); ++c; }
            printf(" %s %.0f |", names[ph], s / c);
            life += s / c;
        }
        printf(" life %.0f\n", life);
    }
    return 0;
}

#!/usr/bin/env python3
"""Headline benchmark: mel-frames/s of the streaming GRU keyword-spotting path on N MI355X GPUs.

  python bench.py --gpus N --steps K --warmup W
  python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 \
         --master-port P bench.py --gpus N --steps K --warmup W

One "step" = one kws_step call: B=4096 concurrent streams per GPU advanced by T=300 mel frames
(BASELINE.json configs[1]: 2xGRU h=128, n_mel=40, 6 classes, fp32), state carried on the device from
step to step, logits + softmax + fused ctc_decode2 token events written.  Inputs are resident in HBM
before the timed region.  Weak scaling: every rank owns its own 4096 streams, no data-path collective.
Prints ONE JSON line on rank 0.
"""
import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

FLOP_PER_FRAME = {"total": 327168, "layer": [2 * 64512, 2 * (98304 + 768)]}   # SURVEY 8d / BASELINE.md 4
BYTES_PER_FRAME = 160 + 24 + 24 + 1                                          # mel in, logits, softmax, token out
PEAK_FP32_TFLOPS = 157.3                                                      # MI355X_MICROARCH.md
PEAK_HBM_GBS = 8000.0


def csrc_hash():
    """sha1 over the kernel sources this build was made from (comments and whitespace stripped); tools/prof.sh stores it next to every profile set
    (profiles/<tag>_source_hash.txt) so that figures read back from a committed profile are dropped when the kernels
    have changed since."""
    import glob
    import hashlib
    import re
    h = hashlib.sha1()
    d = os.path.join(ROOT, "keyword_spotting_amd", "csrc")
    for f in sorted(glob.glob(os.path.join(d, "*.hip")) + glob.glob(os.path.join(d, "*.h")) + [os.path.join(d, "Makefile")]):
        text = open(f, "r", errors="replace").read()
        if not f.endswith("Makefile"):          # comments and layout do not change the kernels: code tokens only
            text = re.sub(r"/\*.*?\*/", " ", text, flags=re.S)
            text = re.sub(r"//[^\n]*", " ", text)
        else:
            text = re.sub(r"#[^\n]*", " ", text)
        h.update(os.path.basename(f).encode())
        h.update(" ".join(text.split()).encode())
    return h.hexdigest()[:16]


def profile_is_current(path):
    """True iff the profile file `path` (relative to the repo root) was taken from the kernel sources of this build."""
    if not path:
        return False
    import re
    m = re.match(r"^(r\d+(?:_[A-Za-z0-9]+?)?)_(?:kernel_stats\.csv|pmc\.json)$", os.path.basename(path))
    if not m:
        return False
    stamp = os.path.join(ROOT, "profiles", m.group(1) + "_source_hash.txt")
    try:
        return open(stamp).read().split()[0] == csrc_hash()
    except Exception:
        return False


def rocprof_kernel_avg_ms(pattern, tag=None):
    """Average duration (ms) of the first kernel whose name contains `pattern` in the latest committed
    profiles/r<round>[_<tag>]_kernel_stats.csv (rocprofv3 --kernel-trace --stats of this command), and the file."""
    import csv
    import glob
    import re
    rx = re.compile(r"^r(\d+)_kernel_stats\.csv$" if tag is None else r"^r(\d+)_%s_kernel_stats\.csv$" % re.escape(tag))
    files = [f for f in glob.glob(os.path.join(ROOT, "profiles", "r*_kernel_stats.csv")) if rx.match(os.path.basename(f))]
    if not files:
        return None, None
    latest = max(files, key=lambda f: int(rx.match(os.path.basename(f)).group(1)))
    if not profile_is_current(latest):              # kernels changed since that trace: do not quote it
        return None, os.path.relpath(latest, ROOT) + " (stale: taken from other kernel sources)"
    for row in csv.DictReader(open(latest)):
        if pattern in row.get("Name", ""):
            return float(row["AverageNs"]) * 1e-6, os.path.relpath(latest, ROOT)
    return None, os.path.relpath(latest, ROOT)


def pmc_bytes(pattern, tag):
    """HBM bytes per launch (FETCH_SIZE x2 gfx950 correction + WRITE_SIZE) of a kernel in profiles/r<round>_<tag>_pmc.json."""
    import glob
    import re
    rx = re.compile(r"^r(\d+)_%s_pmc\.json$" % re.escape(tag))
    files = [f for f in glob.glob(os.path.join(ROOT, "profiles", "r*_pmc.json")) if rx.match(os.path.basename(f))]
    if not files:
        return None, None
    latest = max(files, key=lambda f: int(rx.match(os.path.basename(f)).group(1)))
    if not profile_is_current(latest):
        return None, os.path.relpath(latest, ROOT) + " (stale: taken from other kernel sources)"
    for k, c in json.load(open(latest)).items():
        if pattern in k and "hbm_bytes_per_launch" in c:
            return c["hbm_bytes_per_launch"], os.path.relpath(latest, ROOT)
    return None, os.path.relpath(latest, ROOT)


def pmc_field(pattern, tag, field):
    """One derived figure of a kernel from the latest committed profiles/r<round>_<tag>_pmc.json (None when stale or absent)."""
    import glob
    import re
    rx = re.compile(r"^r(\d+)_%s_pmc\.json$" % re.escape(tag))
    files = [f for f in glob.glob(os.path.join(ROOT, "profiles", "r*_pmc.json")) if rx.match(os.path.basename(f))]
    if not files:
        return None
    latest = max(files, key=lambda f: int(rx.match(os.path.basename(f)).group(1)))
    if not profile_is_current(latest):
        return None
    for k, c in json.load(open(latest)).items():
        if pattern in k:
            if field in c:
                return c[field]
            if field == "valu_busy_frac_of_simd_cycles" and c.get("GRBM_GUI_ACTIVE") and "SQ_ACTIVE_INST_VALU" in c:
                return 4.0 * c["SQ_ACTIVE_INST_VALU"] / (1024.0 * c["GRBM_GUI_ACTIVE"] / 8.0)
    return None


def secondary_lines(device):
    """Informational figures for the other BASELINE configs, measured after the timed region on rank 0 (never part of
    `value`): the bf16 and int8 variants of the headline workload, the end-to-end PCM -> trigger streaming loop (fp32 and
    bf16 stacks, with a roofline per kernel of the loop) and configs[4].  Every entry carries its own roofline figures so
    that the driver-run line alone holds them.  A few seconds in total."""
    import torch
    from keyword_spotting_amd import get_config, weights
    from keyword_spotting_amd.detector import StreamManager
    from keyword_spotting_amd.frontend import MelFrontend
    from keyword_spotting_amd.rnn_ctc import DeployModel

    def timed(fn, n):
        fn()
        fn()
        torch.cuda.synchronize(device)
        t0 = time.perf_counter()
        for _ in range(n):
            fn()
        torch.cuda.synchronize(device)
        return (time.perf_counter() - t0) / n

    def events(fn, n):
        """ms per call by HIP events on the stream the kernels are launched on (torch's current stream)."""
        fn()
        a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        a.record()
        for _ in range(n):
            fn()
        b.record()
        torch.cuda.synchronize(device)
        return a.elapsed_time(b) / n

    out = {}
    B, T = 4096, 300
    valu_pk_i16_peak = 256 * 64 * 2.4e9 * (8.0 / 3.0) / 1e12          # exact maddubs emulation: 8 int ops per 3 lane-instructions
    # max|dlogit| and token agreement of the reduced-precision variants against the fp32 path on the
    # same random-init weights and mel (64 streams x 300 frames)
    acc_mel = (torch.randn(64, T, 40, device=device).abs() * 2).contiguous()
    cfg32 = get_config()
    m32 = DeployModel(cfg32, weights.init_weights(cfg32, seed=0), device=device)
    ref = m32.forward(acc_mel, m32.zero_state(64), prev_word=m32.fresh_prev_word(64))
    m32.close()
    for prec, steps in (("f16x3", 20), ("bf16", 5), ("int8", 3)):
        cfg = get_config(precision=prec)
        m = DeployModel(cfg, weights.init_weights(cfg, seed=0), device=device)
        got = m.forward(acc_mel, m.zero_state(64), prev_word=m.fresh_prev_word(64))
        dl = (got["logits"] - ref["logits"]).abs()
        accuracy = {"vs": "fp32 path, same weights and mel, 64 streams x %d frames" % T,
                    "max_abs_dlogit": float(dl.max()), "mean_abs_dlogit": float(dl.mean()),
                    "frames_with_identical_decode2_token": float((got["tokens"] == ref["tokens"]).float().mean()),
                    "streams_with_identical_token_sequence": float((got["tokens"] == ref["tokens"]).all(1).float().mean())}
        mel = (torch.randn(B, T, cfg.n_mel, device=device).abs() * 2).contiguous()
        st, pw = m.zero_state(B), m.fresh_prev_word(B)
        m.set_profiling(True)
        dt = timed(lambda: m.forward(mel, st, prev_word=pw, state_out=st), steps)
        kt = m.kernel_times()
        entry = {"mel_frames_per_s": B * T / dt, "ms_per_step": dt * 1e3, "accuracy": accuracy}
        if prec == "f16x3":
            # fp32 results on the fp16 matrix pipe: every product is three MFMAs on split operands (csrc/gru_f16x3.hip); the
            # roofline counts the ISSUED matrix flops (3x the algorithmic ones) against the dense fp16 MFMA peak
            per = [k_[0] / max(k_[1], 1) for k_ in kt]
            dom = max(range(len(per)), key=lambda l: per[l])
            issued = 3 * FLOP_PER_FRAME["layer"][dom] * B * T / (per[dom] * 1e-3) / 1e12
            entry["dtype"] = "f16x3 split: fp16 MFMA on (hi, lo) operand pairs, fp32 accumulate/activations/state -- fp32 tolerance, NOT fp32 arithmetic"
            entry["speedup_vs_fp32_path"] = None          # filled by the caller, which knows the headline value
            entry["roofline"] = {"bound": "mfma (fp16, dense)", "kernel": m.kernel_names()[dom], "kernel_ms": per[dom],
                                 "achieved": issued, "peak": 2500.0, "unit": "TFLOP/s issued (3 MFMAs per product)", "frac": issued / 2500.0,
                                 "algorithmic_tflops_all_layers": FLOP_PER_FRAME["total"] * B * T / (sum(per) * 1e-3) / 1e12,
                                 "per_layer_ms": per, "kernels": m.kernel_names()}
        elif prec == "bf16":
            ms = kt[0][0] / max(kt[0][1], 1)
            tf = FLOP_PER_FRAME["total"] * B * T / (ms * 1e-3) / 1e12
            entry["roofline"] = {"bound": "mfma", "kernel": m.kernel_names()[0], "kernel_ms": ms, "achieved": tf,
                                 "peak": 2500.0, "unit": "TFLOP/s", "frac": tf / 2500.0,
                                 "hbm_algorithmic_GBps": BYTES_PER_FRAME * B * T / (ms * 1e-3) / 1e9}
        else:
            ms = kt[1][0] / max(kt[1][1], 1)
            tops = FLOP_PER_FRAME["layer"][1] * B * T / (ms * 1e-3) / 1e12
            entry["roofline"] = {"bound": "valu-pk-i16", "kernel": "gru_layer_octbit layer 1 + projection", "kernel_ms": ms, "achieved": tops,
                                 "peak": valu_pk_i16_peak, "unit": "TOP/s", "frac": tops / valu_pk_i16_peak,
                                 "per_layer_ms": [k[0] / max(k[1], 1) for k in kt]}
        out[("configs[1] at fp32 tolerance on the fp16 matrix pipe (%s), %d streams x %d frames" if prec == "f16x3" else
             "configs[2] %s, %d streams x %d frames") % (prec, B, T)] = entry
        m.close()
    # SURVEY 8(d) config 2, second clause: the streaming hop itself -- T = 22 frames per call (detector.py:119 feeds 3600-sample
    # chunks), 64 consecutive kws_step calls with the state carried on the device, mel already resident; per-layer kernel time
    # by HIP events, each layer against the peak of the pipe it runs on, next to the T = 300 figures above.  The T = 1 run
    # gives the per-launch fixed cost (dispatch, weight staging, un-woven first frame, drain): what a 22-frame call cannot amortise
    hop_T, hop_calls = 22, 64
    hop_mel = [(torch.randn(B, hop_T, 40, device=device).abs() * 2).contiguous() for _ in range(4)]
    for prec in ("fp32", "f16x3", "bf16"):
        cfg = get_config(precision=prec)
        m = DeployModel(cfg, weights.init_weights(cfg, seed=0), device=device)
        st, pw = m.zero_state(B), m.fresh_prev_word(B)
        hop_out = {"logits": torch.empty(B, hop_T, cfg.num_classes, device=device), "softmax": torch.empty(B, hop_T, cfg.num_classes, device=device),
                   "tokens": torch.empty(B, hop_T, dtype=torch.int8, device=device)}
        m.reserve(B, hop_T)
        hk = [0]
        def hop():
            m.forward(hop_mel[hk[0] % 4], st, prev_word=pw, state_out=st, out=hop_out)
            hk[0] += 1
        for _ in range(8):
            hop()
        torch.cuda.synchronize(device)
        m.set_profiling(True)
        m.kernel_times()
        t0 = time.perf_counter()
        for _ in range(hop_calls):
            hop()
        torch.cuda.synchronize(device)
        dt = (time.perf_counter() - t0) / hop_calls
        kt = m.kernel_times()
        names = m.kernel_names()
        # T = 1: the same handle, 64 calls of one frame
        one = hop_mel[0][:, :1].contiguous()
        one_out = {k_: v_[:, :1].contiguous() for k_, v_ in hop_out.items()}
        for _ in range(4):
            m.forward(one, st, prev_word=pw, state_out=st, out=one_out)
        m.kernel_times()
        for _ in range(hop_calls):
            m.forward(one, st, prev_word=pw, state_out=st, out=one_out)
        k1 = m.kernel_times()
        m.set_profiling(False)
        nslots = 1 if prec == "bf16" else len(kt)
        per = [kt[l][0] / max(kt[l][1], 1) for l in range(nslots)]
        per1 = [k1[l][0] / max(k1[l][1], 1) for l in range(nslots)]
        peak = PEAK_FP32_TFLOPS if prec == "fp32" else 2500.0
        mult = 3 if prec == "f16x3" else 1
        fl = [FLOP_PER_FRAME["total"]] if prec == "bf16" else FLOP_PER_FRAME["layer"]
        layers_ = []
        for l in range(nslots):
            tf = mult * fl[l] * B * hop_T / (per[l] * 1e-3) / 1e12
            layers_.append({"kernel": names[l], "kernel_ms": per[l], "tflops" + ("_issued" if mult == 3 else ""): tf, "frac": tf / peak,
                            "kernel_ms_at_T1": per1[l],
                            "fixed_cost_share_of_the_call": per1[l] / per[l] if per[l] > 0 else None})
        out["configs[1] T=%d, %d consecutive kws_step calls, state carried, %s, %d streams" % (hop_T, hop_calls, prec, B)] = {
            "mel_frames_per_s": B * hop_T / dt, "realtime_streams_gru_only": B * hop_T / dt / 100.0, "ms_per_call": dt * 1e3,
            "kernel_ms_sum": sum(per), "host_and_launch_gap_ms": dt * 1e3 - sum(per),
            "peak_tflops": peak, "layers": layers_,
            "what": "mel-fed kws_step only (no front-end, no window): SURVEY 8(d) config 2's streaming-hop entry; `frac` per layer is against the "
                    "fp32 MFMA peak (fp32), the dense fp16 peak on 3x the flops (f16x3), the dense bf16 peak (bf16)"}
        m.close()
    del hop_mel
    # the loop the reference ships: PCM chunks of 225 ms in, trigger decisions out (kws_stream_feed)
    pcm = [(torch.randn(B, 3600, device=device) * 0.1).contiguous() for _ in range(4)]
    fe_ms = None
    # ... and the same loop with four times the streams per call: a GPU at full load serves hundreds of 4096-stream batches
    # per 225 ms period, so the caller may hand over bigger ones; the resident kernels loop over stream groups inside one
    # launch (weights staged once per CU) and the launch gaps are paid once
    big = 16384
    pcm_big = [(torch.randn(big, 3600, device=device) * 0.1).contiguous() for _ in range(2)]
    for prec in ("fp32", "f16x3", "bf16"):
        cfg = get_config(precision=prec)
        m = DeployModel(cfg, weights.init_weights(cfg, seed=0), device=device)
        fe, mgr = MelFrontend(cfg), StreamManager(m, big)
        kb = [0]
        def chunk_big():
            mgr.feed_pcm(pcm_big[kb[0] % 2], fe)
            kb[0] += 1
        for _ in range(3):
            chunk_big()
        dt = timed(chunk_big, 60)
        out["detector.py loop, PCM in -> trigger out, %s, %d streams x 225 ms chunks per call" % (prec, big)] = {
            "realtime_streams": big * 0.225 / dt, "ms_per_chunk": dt * 1e3}
        mgr.close()
        m.close()
    del pcm_big
    for prec in ("fp32", "f16x3", "bf16"):
        cfg = get_config(precision=prec)
        m = DeployModel(cfg, weights.init_weights(cfg, seed=0), device=device)
        fe, mgr = MelFrontend(cfg), StreamManager(m, B)
        k = [0]
        def chunk():
            mgr.feed_pcm(pcm[k[0] % 4], fe)
            k[0] += 1
        for _ in range(3):
            chunk()
        m.set_profiling(True)
        m.kernel_times()
        dt = timed(chunk, 200)
        kt = m.kernel_times()
        m.set_profiling(False)
        frames = B * 22.5                                   # 3600-sample hops: 22 and 23 frames alternate
        names = m.kernel_names()
        rides = any("window tail" in nm for nm in names)
        gru_launches = sum(1 for nm in names if nm)
        entry = {"realtime_streams": B * 0.225 / dt, "ms_per_chunk": dt * 1e3,
                 "kernel_launches_per_chunk": 1 + gru_launches + (0 if rides else 1),
                 "window_step": ("incremental, inside the last GRU layer's launch" if rides else "incremental, window_inc_kernel behind the stack")}
        if prec == "fp32":
            per = [k_[0] / max(k_[1], 1) for k_ in kt]
            entry["gru_kernels"] = [{"kernel": "gru_layer_resident layer %d" % l, "kernel_ms": per[l],
                                     "tflops": FLOP_PER_FRAME["layer"][l] * frames / (per[l] * 1e-3) / 1e12,
                                     "frac": FLOP_PER_FRAME["layer"][l] * frames / (per[l] * 1e-3) / 1e12 / PEAK_FP32_TFLOPS}
                                    for l in range(len(per))]
            # (rocprof names carry the kernels' fourth template parameter: the loop's last layer is the instantiation with the window tail)
            for l, pat in enumerate(("gru_layer_resident<10, true, false, false>", "gru_layer_resident<32, false, true, %s>" % ("true" if rides else "false"))):
                rp, rp_src = rocprof_kernel_avg_ms(pat, "e2e")           # the committed trace of tools/bench_e2e.py (same loop)
                if rp:
                    entry["gru_kernels"][l].update({"kernel_ms_rocprof": rp, "rocprof_source": rp_src,
                                                    "frac_rocprof": FLOP_PER_FRAME["layer"][l] * frames / (rp * 1e-3) / 1e12 / PEAK_FP32_TFLOPS})
            # the front-end kernel alone on one chunk with its carried samples (22 frames per stream): HBM-bound,
            # algorithmic bytes = PCM in (3840 samples) + mel out
            carry = torch.zeros(B, 240, device=device)
            fe_ms = events(lambda: fe.forward_carry(carry, pcm[0], 240), 20)
            alg = B * (3840 * 4 + 22 * cfg.n_mel * 4)
            traffic, src = pmc_bytes("mel_fft400_kernel", "fe")
            rp_ms, rp_src = rocprof_kernel_avg_ms("mel_fft400_kernel", "fe")
            # The kernel moves exactly its algorithmic bytes (traffic / algorithmic ~ 1.0) at about a third of the HBM peak, so HBM
            # is not what binds it: it is bound by VALU ISSUE (the 16x25 butterflies; ~580 VALU instructions per wave).  `frac`
            # is therefore the share of all SIMD-cycles of the launch with the VALU busy (PMC: SQ_ACTIVE_INST_VALU against
            # GRBM_GUI_ACTIVE, profiles/r*_fe_pmc.json); the HBM fraction is kept beside it as the secondary lens.
            valu_busy = pmc_field("mel_fft400_kernel", "fe", "valu_busy_frac_of_simd_cycles")
            entry["frontend"] = {"kernel": "mel_fft400_kernel (16x25 real FFT + mel MFMA) + carry_tail", "kernel_ms": fe_ms,
                                 "mel_frames_per_s": B * 22 / (fe_ms * 1e-3),
                                 "roofline": {"bound": "valu-issue", "frac": valu_busy, "unit": "share of SIMD-cycles with the VALU busy (PMC)",
                                              "achieved": valu_busy, "peak": 1.0,
                                              "hbm": {"achieved": alg / (fe_ms * 1e-3) / 1e9, "peak": PEAK_HBM_GBS, "unit": "GB/s",
                                                      "frac": alg / (fe_ms * 1e-3) / 1e9 / PEAK_HBM_GBS, "algorithmic_bytes_per_launch": alg,
                                                      "traffic": traffic, "traffic_source": src},
                                              "kernel_ms_rocprof": rp_ms, "rocprof_source": rp_src}}
        elif prec == "f16x3":
            per = [k_[0] / max(k_[1], 1) for k_ in kt]
            entry["gru_kernels"] = [{"kernel": m.kernel_names()[l], "kernel_ms": per[l],
                                     "tflops_issued": 3 * FLOP_PER_FRAME["layer"][l] * frames / (per[l] * 1e-3) / 1e12,
                                     "frac": 3 * FLOP_PER_FRAME["layer"][l] * frames / (per[l] * 1e-3) / 1e12 / 2500.0}
                                    for l in range(len(per))]
        else:
            ms = kt[0][0] / max(kt[0][1], 1)
            entry["gru_kernels"] = [{"kernel": m.kernel_names()[0], "kernel_ms": ms,
                                     "tflops": FLOP_PER_FRAME["total"] * frames / (ms * 1e-3) / 1e12,
                                     "frac": FLOP_PER_FRAME["total"] * frames / (ms * 1e-3) / 1e12 / 2500.0}]
        out["detector.py loop, PCM in -> trigger out, %s, %d streams x 225 ms chunks (VAD, front-end, GRU, window)" % (prec, B)] = entry
        mgr.close()
        m.close()
    # the int8 ("octbit") graph through the same loop (no window tail in its last layer: window_inc_kernel follows the stack)
    cfg = get_config(precision="int8")
    m = DeployModel(cfg, weights.init_weights(cfg, seed=0), device=device)
    fe, mgr = MelFrontend(cfg), StreamManager(m, B)
    k8 = [0]
    def chunk8():
        mgr.feed_pcm(pcm[k8[0] % 4], fe)
        k8[0] += 1
    for _ in range(3):
        chunk8()
    dt = timed(chunk8, 100)
    names = m.kernel_names()
    out["detector.py loop, PCM in -> trigger out, int8, %d streams x 225 ms chunks (VAD, front-end, GRU, window)" % B] = {
        "realtime_streams": B * 0.225 / dt, "ms_per_chunk": dt * 1e3,
        "kernel_launches_per_chunk": 5,       # gate + front-end | fp32 layer 0 | int8 layer 1 (leaves the projection's activation ranges) | int8 projection | window step
        "kernels": names, "window_step": "incremental, window_inc_kernel behind the stack"}
    mgr.close()
    m.close()
    del pcm
    # the reference's SHIPPED default front-end width: n_mel = 60 (config/rnn_config.py:63; BASELINE's 40 is README.md:17) --
    # resident first layer with 15 x-part k-chunks; 4096 x 300 and the 22-frame streaming hop, fp32 and f16x3
    flop60 = [2 * (60 + 128) * 3 * 128, 2 * ((128 + 128) * 3 * 128 + 128 * 6)]
    for prec in ("fp32", "f16x3"):
        cfg = get_config(precision=prec, n_mel=60)
        m = DeployModel(cfg, weights.init_weights(cfg, seed=0), device=device)
        entry = {}
        for t_ in (T, 22):
            mel = (torch.randn(B, t_, 60, device=device).abs() * 2).contiguous()
            st, pw = m.zero_state(B), m.fresh_prev_word(B)
            o_ = {"logits": torch.empty(B, t_, cfg.num_classes, device=device), "softmax": torch.empty(B, t_, cfg.num_classes, device=device),
                  "tokens": torch.empty(B, t_, dtype=torch.int8, device=device)}
            m.reserve(B, t_)
            for _ in range(3):
                m.forward(mel, st, prev_word=pw, state_out=st, out=o_)
            m.set_profiling(True)
            m.kernel_times()
            dt = timed(lambda: m.forward(mel, st, prev_word=pw, state_out=st, out=o_), 10 if t_ == T else 64)
            kt = m.kernel_times()
            m.set_profiling(False)
            per = [k_[0] / max(k_[1], 1) for k_ in kt]
            mult, peak = (3, 2500.0) if prec == "f16x3" else (1, PEAK_FP32_TFLOPS)
            entry["%d frames per call" % t_] = {
                "mel_frames_per_s": B * t_ / dt, "ms_per_call": dt * 1e3, "kernels": m.kernel_names(), "per_layer_ms": per,
                "frac_per_layer": [mult * flop60[l] * B * t_ / (per[l] * 1e-3) / 1e12 / peak for l in range(2)],
                "peak_tflops": peak, "flops_counted": "issued (3 MFMAs per product)" if mult == 3 else "algorithmic"}
        out["the reference's shipped default n_mel=60 (config/rnn_config.py:63), 2xGRU h=128, %s, %d streams" % (prec, B)] = entry
        m.close()
    cfg = get_config(n_mel=60, hidden_size=256, num_layers=4)
    m = DeployModel(cfg, weights.init_weights(cfg, seed=0), device=device)
    mel = (torch.randn(1024, T, 60, device=device).abs() * 2).contiguous()
    st = m.zero_state(1024)
    dt = timed(lambda: m.forward(mel, st, state_out=st), 5)
    macs = sum(((60 if l == 0 else 256) + 256) * 3 * 256 for l in range(4)) + 256 * 6
    tf = 2 * macs * 1024 * T / dt / 1e12
    dt_fp32_c4 = dt
    out["configs[4] 4xGRU h=256 n_mel=60, 1024 streams x %d frames, fp32 (layer-pipelined launch)" % T] = {
        "mel_frames_per_s": 1024 * T / dt, "ms_per_step": dt * 1e3, "tflops": tf, "frac": tf / PEAK_FP32_TFLOPS,
        "roofline": {"bound": "mfma", "kernel": "gru_stack_generic_pipelined<4>", "achieved": tf, "peak": PEAK_FP32_TFLOPS,
                     "unit": "TFLOP/s", "frac": tf / PEAK_FP32_TFLOPS}}
    m.close()
    # ... and at fp32 tolerance on the fp16 matrix pipe: every layer's (hi, lo) fp16 operand pairs (1.5 MiB at h = 256, the size of the
    # fp32 weights) streamed from L2 each frame, three MFMAs per pair.  What bounds it is the L2 -> CU stream, not the matrix pipe:
    # bytes = table bytes of all layers x groups x frames (tools/ubench/l2_stream_f16x3.hip: 12.7-13.8 us per frame at best)
    cfg = get_config(n_mel=60, hidden_size=256, num_layers=4, precision="f16x3")
    m = DeployModel(cfg, weights.init_weights(cfg, seed=0), device=device)
    acc_ref = DeployModel(get_config(n_mel=60, hidden_size=256, num_layers=4), weights.init_weights(cfg, seed=0), device=device)
    am_ = (torch.randn(64, T, 60, device=device).abs() * 2).contiguous()
    ra_ = acc_ref.forward(am_, acc_ref.zero_state(64), prev_word=acc_ref.fresh_prev_word(64))
    rf_ = m.forward(am_, m.zero_state(64), prev_word=m.fresh_prev_word(64))
    acc_ref.close()
    from keyword_spotting_amd import sharding as _sh
    clk = _sh.ClockSampler(device, 0.02)             # shader clock while the steps below run (the L2 -> CU path is clocked with it)
    clk.start()
    dt = timed(lambda: m.forward(mel, st, state_out=st), 8)
    clk_mhz = clk.result()
    tf = 2 * macs * 1024 * T / dt / 1e12
    l2_bytes = (16 * 3 * (2 + 8) + 3 * 16 * 3 * 16) * 2 * 1024 * 64 * T          # operands of 2 x 1 KiB: layer 0 has 2 + 8 chunks, layers 1-3 8 + 8
    out["configs[4] 4xGRU h=256 n_mel=60, 1024 streams x %d frames, f16x3 (fp32 tolerance; weights streamed from L2, layer-pipelined launch)" % T] = {
        "mel_frames_per_s": 1024 * T / dt, "ms_per_step": dt * 1e3, "speedup_vs_the_fp32_kernels": dt_fp32_c4 / dt,
        "algorithmic_tflops": tf, "algorithmic_tflops_over_fp32_mfma_peak": tf / PEAK_FP32_TFLOPS,
        "accuracy": {"vs": "fp32 kernels, same weights and mel, 64 streams x %d frames" % T,
                     "max_abs_dlogit": float((rf_["logits"] - ra_["logits"]).abs().max()),
                     "streams_with_identical_token_sequence": float((rf_["tokens"] == ra_["tokens"]).all(1).float().mean())},
        "roofline": {"bound": "l2-stream", "kernel": m.kernel_names()[-1], "achieved": l2_bytes / dt / 1e12, "peak": 34.5, "unit": "TB/s out of the eight L2s",
                     "frac": l2_bytes / dt / 1e12 / 34.5, "bytes_per_step": l2_bytes,
                     "mfma": {"issued_tflops": 3 * tf, "peak": 2500.0, "frac": 3 * tf / 2500.0},
                     "floor_us_per_frame_microbenchmark": "12.7-13.8 (tools/ubench/l2_stream_f16x3.hip, profiles/r6_l2_stream_ubench.txt)",
                     "us_per_frame": dt * 1e6 / (T + 3),
                     # the guide's 34.5 TB/s is 64 B/clk/CU at 2.1 GHz; under this kernel's load the chip clocks lower
                     "shader_clock_mhz_during_the_run": clk_mhz,
                     "upper_layers_B_per_clk_per_cu": (1536 * 1024 / (dt / (T + 3)) / (clk_mhz * 1e6)) if clk_mhz else None,
                     "frac_of_64_B_per_clk_per_cu": (1536 * 1024 / (dt / (T + 3)) / (clk_mhz * 1e6) / 64.0) if clk_mhz else None}}
    m.close()
    del mel, st
    torch.cuda.empty_cache()
    # BASELINE.json's metric as worded -- "real-time audio streams SUSTAINED": N = M x 16384 DISTINCT streams resident (state, sample
    # carry, decode window, and a 225 ms int16 chunk of its own per manager), every manager fed one chunk per 225 ms period for 40
    # periods (9 s) in real time, M native calls per period in turn on 2 HIP streams / model handles; the largest N with ZERO
    # deadline misses (tools/bench_serve.py; detector.py:119,158-209).  The host-fed variant starts every chunk in pinned host memory.
    try:
        sys.path.insert(0, os.path.join(ROOT, "tools"))
        import bench_serve
        for prec in ("fp32", "f16x3", "bf16"):
            out["sustained real-time, N distinct streams, %s" % prec] = bench_serve.sustained_streams(device, prec, periods=40, max_attempts=3)
        out["sustained real-time, N distinct streams, fp32, host-fed (pinned int16 over PCIe, copy streams overlapped)"] = \
            bench_serve.sustained_streams(device, "fp32", periods=40, max_attempts=3, host_fed=True)
    except Exception as exc:
        out["sustained real-time, N distinct streams"] = {"error": repr(exc)}
    return out


def visible_gpus():
    """GPUs a fresh child process sees (torch.cuda.device_count() there), or None when the probe itself fails.  The calling
    process never touches the GPU runtime: bench.py's launcher only starts children."""
    import subprocess
    try:
        r = subprocess.run([sys.executable, "-c", "import torch; print(torch.cuda.device_count())"], capture_output=True, text=True, timeout=300)
        return int(r.stdout.strip().splitlines()[-1]) if r.returncode == 0 else None
    except Exception:
        return None


def usable_cores():
    """(threads to use, logical CPUs visible, cgroup quota or None): floor of the cgroup v2/v1 CPU quota when there is
    one, else the size of the affinity mask."""
    try:
        visible = len(os.sched_getaffinity(0))
    except Exception:
        visible = os.cpu_count() or 1
    quota = None
    try:
        q = open("/sys/fs/cgroup/cpu.max").read().split()
        quota = None if q[0] == "max" else float(q[0]) / float(q[1])
    except Exception:
        try:
            cq = float(open("/sys/fs/cgroup/cpu/cpu.cfs_quota_us").read())
            cp = float(open("/sys/fs/cgroup/cpu/cpu.cfs_period_us").read())
            quota = cq / cp if cq > 0 else None
        except Exception:
            pass
    cores = visible if quota is None else max(1, min(visible, int(quota)))
    return cores, visible, quota


def cpu_baseline(cfg, w, seconds_budget=25.0):
    """The oracle timed on this host (reference TF-1.x is not executable here or on the GPU box):
    (ii) tight C restatement, 1 core and all cores; (i) torch-CPU eager op-by-op = 'TF-CPU stand-in'.
    Workload = BASELINE config 1: batch 1, 300 frames, state round trip, greedy decode."""
    import numpy as np
    import torch
    from oracle import build as obuild
    from oracle import gru_oracle as G
    from oracle import torch_eager as TE
    out = os.path.join(ROOT, "oracle", "_build", "libkws_oracle_native.so")
    try:
        orc = obuild.load(native=True, out=out)
    except Exception:
        orc = obuild.load()
    blob = G.weights_to_blob(w)
    ctuple = (cfg.n_mel, cfg.hidden_size, cfg.num_layers, cfg.num_classes, 0, -1.0)
    mel1 = G.synthetic_mel(1, 300, cfg.n_mel, seed=1)
    st1 = np.zeros((cfg.num_layers, 1, cfg.hidden_size), np.float32)

    def run_c(mel, st, threads):
        lg, sm, s2 = orc.gru_forward(ctuple, blob, mel, st, threads=threads)
        for b in range(mel.shape[0]):
            orc.ctc_decode2(sm[b], cfg.num_classes)
        return s2

    def rate(threads, streams, budget):
        mel = G.synthetic_mel(streams, 300, cfg.n_mel, seed=2)
        st = np.zeros((cfg.num_layers, streams, cfg.hidden_size), np.float32)
        run_c(mel, st, threads)
        n, t0 = 0, time.perf_counter()
        while time.perf_counter() - t0 < budget:
            run_c(mel, st, threads)
            n += 1
        return n * streams * 300 / (time.perf_counter() - t0)

    c1 = rate(1, 1, seconds_budget / 5)
    # the OpenMP team is pinned to what this job may actually use -- the cgroup CPU quota (the box shows more logical
    # CPUs than that), else the affinity mask -- so the figure reproduces from run to run
    cores, visible, quota = usable_cores()
    call = rate(cores, cores * 8, seconds_budget * 0.4) if cores > 1 else c1
    # torch eager, op by op, batch 1, 22-frame chunks with the state round-tripping through numpy
    torch.set_num_threads(1)
    tw = TE.to_torch(w)
    melt = torch.from_numpy(mel1)
    frames, t0 = 0, time.perf_counter()
    state = torch.zeros(cfg.num_layers, 1, cfg.hidden_size)
    while time.perf_counter() - t0 < seconds_budget / 5:
        for pos in range(0, 300, 22):
            lg, sm, state = TE.gru_forward(tw, melt[:, pos:pos + 22], state)
            state = torch.from_numpy(state.numpy().copy())
            frames += lg.shape[1]
    eager = frames / (time.perf_counter() - t0)
    # ... and on all usable cores at once, one independent batch-1 stream per pinned process (BASELINE.md section 3 asks
    # for each arm on 1 and on N cores): fresh interpreters (spawn), so nothing of this process's GPU state is inherited
    eager_all, eager_all_note = None, None
    if cores > 1:
        try:
            import multiprocessing as mp
            cpus = sorted(os.sched_getaffinity(0))
            pin = [cpus[i * len(cpus) // cores] for i in range(cores)] if len(cpus) >= cores else [None] * cores   # spread over the mask
            with mp.get_context("spawn").Pool(cores) as pool:
                rates = pool.map(TE.eager_stream_rate, [(c, seconds_budget / 5, cfg.n_mel, cfg.hidden_size, cfg.num_layers,
                                                        cfg.num_classes) for c in pin])
            eager_all = float(sum(rates))
            eager_all_note = "%d pinned single-thread processes, one stream each, ~%.0fs: min %.0f / max %.0f frames/s per process" % (
                cores, seconds_budget / 5, min(rates), max(rates))
        except Exception as exc:              # never lose the line over the baseline's baseline
            eager_all_note = "not measured: %r" % (exc,)
    return {"value": call, "unit": "mel-frames/s", "cores": cores, "kind": "port",
            "sample": "oracle/kws_oracle.c (restatement of reference semantics; TF-1.x not executable): "
                      "%d independent batch-1 streams x 300 frames + ctc_decode2 per pass, OpenMP team of %d "
                      "(= the job's cgroup CPU quota %s; %d logical CPUs visible), ~%.0fs"
                      % (cores * 8, cores, quota, visible, seconds_budget * 0.4),
            "single_core_value": c1,
            "eager_stand_in": {"value": eager, "cores": 1, "what": "torch-CPU op-by-op GRUCell loop, batch 1, 22-frame "
                               "chunks, state round trip (analogue of the reference's per-op TF dispatch)",
                               "all_cores_value": eager_all, "all_cores": cores, "all_cores_how": eager_all_note,
                               "extrapolated_x_cores": eager * cores}}


def sustained_run(step, device_sync, frames_per_step, seconds, window=100):
    """Runs `step` back to back for at least `seconds`, synchronising every `window` steps: mean mel-frames/s over the
    whole run, the slowest/fastest window, and last-window / first-window (clock droop under sustained load shows here)."""
    rates = []
    device_sync()
    t_start = time.perf_counter()
    while time.perf_counter() - t_start < seconds or len(rates) < 2:
        t0 = time.perf_counter()
        for _ in range(window):
            step()
        device_sync()
        rates.append(window * frames_per_step / (time.perf_counter() - t0))
    total = time.perf_counter() - t_start
    return {"seconds": total, "steps": window * len(rates), "window_steps": window, "windows": len(rates),
            "mel_frames_per_s": window * len(rates) * frames_per_step / total,
            "min_window": min(rates), "max_window": max(rates), "first_window": rates[0], "last_window": rates[-1],
            "last_over_first": rates[-1] / rates[0]}


def pmc_traffic_all():
    """{kernel name: HBM bytes per launch} for every kernel of the latest committed PMC pass of THIS command
    (profiles/r<round>_pmc.json, written by tools/prof.sh <tag> on bench.py's default workload; FETCH_SIZE x2 gfx950
    correction + WRITE_SIZE), and its path.  The other r<round>_<variant>_pmc.json files belong to other workloads."""
    import glob
    import re
    files = [f for f in glob.glob(os.path.join(ROOT, "profiles", "r*_pmc.json")) if re.match(r"^r\d+_pmc\.json$", os.path.basename(f))]
    if not files:
        return {}, None
    latest = max(files, key=lambda f: int(re.match(r"^r(\d+)_", os.path.basename(f)).group(1)))
    if not profile_is_current(latest):
        return {}, os.path.relpath(latest, ROOT) + " (stale: taken from other kernel sources)"
    data = json.load(open(latest))
    return {k: c["hbm_bytes_per_launch"] for k, c in data.items() if "hbm_bytes_per_launch" in c}, os.path.relpath(latest, ROOT)


def main(argv=None, model_factory=None):
    """model_factory(cfg, weights, device, kernel) -> object with DeployModel's surface; tests pass a stub that needs no
    GPU so that THIS function -- launcher, rank/device plumbing, barriers, reduction, the JSON line -- is what the
    world_size-2 gloo test executes (tests/test_dist_gloo.py).  The driver never passes one."""
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--warmup", type=int, default=5)
    ap.add_argument("--batch", type=int, default=4096, help="streams per GPU")
    ap.add_argument("--frames", type=int, default=300, help="mel frames per stream per step")
    ap.add_argument("--kernel", default="auto")
    ap.add_argument("--precision", default="fp32", choices=["fp32", "f16x3", "bf16", "int8"],
                    help="fp32 = the reference arithmetic (headline); bf16 / int8 = BASELINE configs[2] variants (secondary)")
    ap.add_argument("--no-cpu-baseline", action="store_true", help="skip the sustained run, the secondary lines and the CPU baseline")
    ap.add_argument("--sustain-seconds", type=float, default=10.0, help="length of the sustained-load run after the timed region")
    ap.add_argument("--dist-backend", default="nccl", help="nccl (= RCCL over xGMI); gloo only to exercise the "
                    "multi-process path on a box with fewer GPUs than ranks")
    ap.add_argument("--force-dist", action="store_true", help="initialise the process group even at one rank: runs the RCCL "
                    "init / barrier / reduction / gather code of the N > 1 path on a single-GPU box")
    args = ap.parse_args(argv)
    stub = model_factory is not None

    if args.gpus > 1 and int(os.environ.get("WORLD_SIZE", "1")) == 1:
        # not under torchrun: start one rank per GPU as CHILD processes (nothing here has touched the GPU yet)
        import socket
        import subprocess
        if args.dist_backend == "nccl" and not stub:
            # one rank per GPU over RCCL: fewer GPUs than ranks is decided HERE, before any rendezvous, by a child probe (the
            # launcher itself never initialises the GPU runtime: it only ever starts children)
            n_vis = visible_gpus()
            if n_vis is not None and n_vis < args.gpus:
                sys.stderr.write("bench.py --gpus %d: only %d GPU(s) visible on this box (one rank per GPU over RCCL); nothing was run\n"
                                 % (args.gpus, n_vis))
                raise SystemExit(2)
        sock = socket.socket()
        sock.bind(("127.0.0.1", 0))
        port = sock.getsockname()[1]
        sock.close()
        cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", str(args.gpus),
               "--master-addr", "127.0.0.1", "--master-port", str(port), os.path.abspath(sys.argv[0])] + \
              (sys.argv[1:] if argv is None else list(argv))
        raise SystemExit(subprocess.call(cmd))

    import torch
    from keyword_spotting_amd import get_config, sharding, weights

    rank, local_rank, world = sharding.env_rank_world()
    if world != args.gpus:
        raise SystemExit("bench.py: --gpus %d but WORLD_SIZE=%d" % (args.gpus, world))
    dist = None
    if not stub and args.dist_backend == "nccl" and world > 1 and torch.cuda.device_count() < world:
        # under an external torchrun on a box with fewer GPUs than ranks: say so once and leave before the rendezvous
        # (device_count() does not initialise the GPU runtime)
        if rank == 0:
            sys.stderr.write("bench.py --gpus %d: only %d GPU(s) visible on this box (one rank per GPU over RCCL); nothing was run\n"
                             % (world, torch.cuda.device_count()))
        raise SystemExit(2)
    if world > 1 or args.force_dist:
        import torch.distributed as dist
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29511")
        os.environ.setdefault("RANK", "0")
        os.environ.setdefault("WORLD_SIZE", "1")
        if args.dist_backend == "nccl":
            torch.cuda.set_device(local_rank)
            dist.init_process_group("nccl", device_id=torch.device("cuda", local_rank))
        else:
            dist.init_process_group(args.dist_backend)
    if stub:
        device = torch.device("cpu")
    else:
        if not torch.cuda.is_available():
            raise SystemExit("bench.py needs a GPU: the HIP path has no CPU fallback")
        device = torch.device("cuda", local_rank % torch.cuda.device_count())
        torch.cuda.set_device(device)
        from keyword_spotting_amd.rnn_ctc import DeployModel
        model_factory = lambda cfg_, w_, dev_, kern_: DeployModel(cfg_, w_, device=dev_, kernel=kern_)   # noqa: E731
    sync_device = device if args.dist_backend == "nccl" else torch.device("cpu")

    def device_sync():
        if device.type == "cuda":
            torch.cuda.synchronize(device)

    cfg = get_config(precision=args.precision)
    w = weights.init_weights(cfg, seed=0)
    model = model_factory(cfg, w, device, args.kernel)
    B, T = args.batch, args.frames
    model.reserve(B, T)
    gen = torch.Generator(device=device).manual_seed(sharding.shard_seed(1, rank))
    mel = (torch.randn(B, T, cfg.n_mel, generator=gen, device=device).abs() * 2).contiguous()
    state = model.zero_state(B)
    prev_word = model.fresh_prev_word(B)
    out = {"logits": torch.empty(B, T, cfg.num_classes, device=device),
           "softmax": torch.empty(B, T, cfg.num_classes, device=device),
           "tokens": torch.empty(B, T, dtype=torch.int8, device=device)}

    def step():
        model.forward(mel, state, prev_word=prev_word, state_out=state, out=out)

    t_w = time.perf_counter()
    for _ in range(args.warmup):
        step()
    device_sync()
    # shader clock while the timed steps run: sampled once by a helper thread half-way through (estimated from the warm-up),
    # its sysfs path resolved here -- nothing of it executes inside the timed interval on this thread
    est_run = (time.perf_counter() - t_w) / max(args.warmup, 1) * args.steps
    clock = sharding.ClockSampler(device, est_run * 0.5)
    model.set_profiling(True)
    model.kernel_times(reset=True)
    sharding.barrier(dist, sync_device)
    device_sync()
    clock.start()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        step()
    device_sync()
    elapsed_own = time.perf_counter() - t0          # this rank alone (per_rank); `value` uses the barrier-to-barrier time
    sharding.barrier(dist, sync_device)
    elapsed = time.perf_counter() - t0
    clock_mhz = clock.result()                      # None if unreadable as this user
    ktimes = model.kernel_times(reset=True)
    model.set_profiling(False)
    frames, seconds = sharding.reduce_throughput(dist, B * T * args.steps, elapsed, sync_device)
    ranks_seen = sharding.count_ranks(dist, sync_device)
    # who took part: host, device identity and own rate of every rank, so that an N-GPU line proves N distinct devices and
    # shows the slowest one (8 bytes of bookkeeping per rank after the timed region; not a data-path collective)
    me = sharding.rank_identity(rank, local_rank, device, B * T * args.steps / elapsed_own)
    me["kernel_ms"] = [k[0] / max(k[1], 1) for k in ktimes]       # this rank's own per-layer kernel time (HIP events): a slow rank explains itself
    me["clock_mhz_if_readable"] = clock_mhz
    per_rank = sharding.gather_rank_info(dist, me)

    if rank == 0:
        value = frames / seconds
        if args.precision == "bf16":     # one fused launch: slot 0 holds the whole stack
            ktimes = ktimes[:1]
        dom = max(range(len(ktimes)), key=lambda l: ktimes[l][0])
        dom_ms = ktimes[dom][0] / max(ktimes[dom][1], 1)
        dom_flops = FLOP_PER_FRAME["total"] if args.precision == "bf16" else FLOP_PER_FRAME["layer"][dom]
        if args.precision == "f16x3":
            dom_flops *= 3              # three MFMAs on split operands per product: the issued matrix flops
        achieved = dom_flops * B * T / (dom_ms * 1e-3) / 1e12
        peak = 2500.0 if args.precision in ("bf16", "f16x3") else PEAK_FP32_TFLOPS
        bound = "mfma"
        if args.precision == "int8" and dom >= 1:
            # exact emulation of the reference's int16-saturating pair sums runs on the packed-int16 VALU
            # (3 instructions per 2 pairs = 8 int ops / 3 lane-instructions)
            peak, bound = 256 * 64 * 2.4e9 * (8.0 / 3.0) / 1e12, "valu-pk-i16 (secondary line)"
        all_ms = sum(k[0] / max(k[1], 1) for k in ktimes)
        dom_name = "gru_layer_resident<%s, false>" % ("10, true, false" if dom == 0 else "32, false, true")   # as rocprofv3 prints it (4th parameter: no window tail)
        pmc, traffic_src = pmc_traffic_all() if getattr(model, "kernel", "") != "generic" and args.precision == "fp32" else ({}, None)
        traffic = next((v for k, v in pmc.items() if dom_name in k), None)
        pmc_step = sum(v for k, v in pmc.items() if "gru_layer_resident" in k) if pmc else None
        path_bytes = BYTES_PER_FRAME * B * T                  # 160 mel in + 24 logits (+24 softmax +1 token) out
        # the dominant kernel's own share of those boundary bytes; the fp32 inter-layer seam (512 B/frame written by
        # layer 0 and read back by layer 1) is INTERNAL traffic, not algorithmic work -- it is what `traffic` exceeds by
        dom_alg = ((160 if dom == 0 else 49) if args.precision != "bf16" else BYTES_PER_FRAME) * B * T
        rp_ms, rp_src = (rocprof_kernel_avg_ms(dom_name) if B == 4096 and T == 300 and pmc else (None, None))
        line = {
            "metric": "mel-frames/s (real-time 10 ms-hop audio streams sustained = value/100)",
            "value": value, "unit": "mel-frames/s", "n_gpus": world, "ranks_seen": ranks_seen, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": seconds / args.steps * 1e3, "higher_is_better": True, "scaling": "weak",
            "vs_baseline": None, "dtype": {"fp32": "f32", "f16x3": "f16x3 split, fp32 tolerance (secondary line; headline is f32)", "bf16": "bf16 (secondary line; headline is f32)",
                                           "int8": "u8 x s8 -> sat i16 -> i32, layer 0 f32 (secondary line; headline is f32)"}[args.precision],
            "data": "synthetic",
            "config": {"workload": "configs[1]: 2xGRU h=128 n_mel=40 6-class, %d concurrent streams/GPU x %d frames "
                                   "per step, %s, state carried on device, logits+softmax+fused ctc_decode2"
                                   % (B, T, {"fp32": "fp32", "f16x3": "fp32-tolerance f16x3 split variant", "bf16": "configs[2] bf16 variant", "int8": "configs[2] octbit int8 variant"}[args.precision]),
                       "streams_per_gpu": B, "frames_per_step": T, "parallelism": "utterance-dp%d" % world,
                       "kernel": getattr(model, "kernel", "stub")},
            "realtime_streams": value / 100.0,
            "per_rank": per_rank,
            "distinct_devices": len({(r.get("host"), r.get("device_id")) for r in per_rank}),
            "roofline": {"bound": bound, "achieved": achieved, "peak": peak, "unit": "TFLOP/s",
                         "frac": achieved / peak,
                         # the same fraction from the committed rocprofv3 kernel-trace average of this command (tracer attached)
                         "frac_rocprof": (dom_flops * B * T / (rp_ms * 1e-3) / 1e12 / peak) if rp_ms else None,
                         "kernel_ms_rocprof": rp_ms, "rocprof_source": rp_src,
                         "traffic": traffic, "traffic_source": traffic_src,
                         "algorithmic_bytes_per_launch": dom_alg,
                         "seam_bytes_per_launch": 0 if args.precision == "bf16" else 512 * B * T,
                         "path_algorithmic_bytes_per_step": path_bytes,
                         "pmc_bytes_per_step": pmc_step,
                         "pmc_over_algorithmic": (pmc_step / path_bytes) if pmc_step else None,
                         "kernel": (model.kernel_names()[dom] if hasattr(model, "kernel_names") and model.kernel_names()[dom] else
                                    "gru_layer_%s layer %d" % ("resident" if getattr(model, "kernel", "") != "generic" else "generic", dom)),
                         "kernel_ms": dom_ms, "launches": ktimes[dom][1],
                         "all_layers_tflops": FLOP_PER_FRAME["total"] * B * T / (all_ms * 1e-3) / 1e12,
                         "per_layer_ms": [k[0] / max(k[1], 1) for k in ktimes],
                         "hbm_algorithmic_GBps": BYTES_PER_FRAME * B * T / (all_ms * 1e-3) / 1e9,
                         "hbm_frac_of_peak": BYTES_PER_FRAME * B * T / (all_ms * 1e-3) / 1e9 / PEAK_HBM_GBS},
        }
        if world == 1 and not args.no_cpu_baseline and not stub:
            # the headline above is K steps (tens of milliseconds): the same step() for >= 10 s tells whether it holds under
            # sustained power/thermal load (reported beside it, never as `value`)
            line["sustained"] = sustained_run(step, device_sync, B * T, args.sustain_seconds)
            line["value_sustained_10s"] = line["sustained"]["mel_frames_per_s"]         # the same step for >= 10 s, next to the K-step `value`
            try:
                line["secondary"] = secondary_lines(device)
                for k_, e_ in line["secondary"].items():
                    if isinstance(e_, dict) and "speedup_vs_fp32_path" in e_:
                        e_["speedup_vs_fp32_path"] = e_["mel_frames_per_s"] / value
                # BASELINE's metric as worded, lifted beside the extrapolated `realtime_streams` = value / 100: distinct streams, each
                # fed its own 225 ms chunk every 225 ms for 9 s with zero deadline misses (secondary: "sustained real-time, ...")
                line["realtime_streams_sustained_distinct"] = {
                    p_: (line["secondary"].get("sustained real-time, N distinct streams, %s" % p_) or {}).get("sustained_streams")
                    for p_ in ("fp32", "f16x3", "bf16")}
            except Exception as exc:          # informational only: never lose the headline line over it
                line["secondary"] = {"error": repr(exc)}
            line["cpu_baseline"] = cpu_baseline(cfg, w)
            line["speedup_vs_cpu_baseline"] = value / line["cpu_baseline"]["value"]
            es = line["cpu_baseline"]["eager_stand_in"]
            line["speedup_vs_eager_stand_in_x_cores"] = value / (es["value"] * line["cpu_baseline"]["cores"])
            if es.get("all_cores_value"):
                line["speedup_vs_eager_stand_in_all_cores_measured"] = value / es["all_cores_value"]
        print(json.dumps(line), flush=True)
    if dist is not None:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()

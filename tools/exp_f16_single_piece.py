#!/usr/bin/env python3
"""Hardware check of the rounding model behind the "f16x1" decision (tests/test_oracle_gru.py::
test_what_single_piece_fp16_operands_would_cost): the f16x3 kernels with the LO PIECE OF EVERY WEIGHT ZEROED -- i.e. fp16
weights, full-precision inputs, the model's "w16" row -- against the fp32 kernels on the same weights and mel, measured exactly
as bench.py's `accuracy` entries (streams whose ctc_decode2 token sequence over 300 frames is identical).

  tools/build_variant.sh wlo0 -DKWS_EXP_F16_WLO_ZERO
  python tools/exp_f16_single_piece.py                  # runs itself twice: product library, then variants/libkws_wlo0.so
"""
import os, subprocess, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def measure():
    import torch
    from keyword_spotting_amd import _lib, get_config, weights
    from keyword_spotting_amd.rnn_ctc import DeployModel
    b, t = 256, 300
    mel = (torch.randn(b, t, 40, generator=torch.Generator().manual_seed(5)).abs() * 2).cuda()
    out = {}
    for prec in ("fp32", "f16x3", "bf16"):
        cfg = get_config(precision=prec)
        m = DeployModel(cfg, weights.init_weights(cfg, seed=0))
        out[prec] = m.forward(mel, m.zero_state(b), prev_word=m.fresh_prev_word(b))
        m.close()
    ver = _lib.load().kws_version().decode()
    for prec in ("f16x3", "bf16"):
        same = (out[prec]["tokens"] == out["fp32"]["tokens"])
        print("%-5s vs fp32, %d streams x %d frames: max |dlogit| %.2e, frames identical %.5f, streams with the identical token sequence %.3f   [%s]"
              % (prec, b, t, float((out[prec]["logits"] - out["fp32"]["logits"]).abs().max()), float(same.float().mean()), float(same.all(1).float().mean()), ver), flush=True)


if __name__ == "__main__":
    if os.environ.get("KWS_EXP_CHILD"):
        measure()
    else:
        for lib in (None, os.path.join(ROOT, "variants", "libkws_wlo0.so")):
            env = dict(os.environ, KWS_EXP_CHILD="1")
            if lib:
                if not os.path.exists(lib):
                    print("missing %s: tools/build_variant.sh wlo0 -DKWS_EXP_F16_WLO_ZERO" % lib)
                    continue
                env["KWS_AMD_LIB"] = lib
            subprocess.check_call([sys.executable, os.path.abspath(__file__)], env=env)

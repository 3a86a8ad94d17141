"""Per-phase cycle counts of the f16x3 frame loop (needs a timing build: KWS_AMD_LIB=variants/libkws_timing.so)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from keyword_spotting_amd import get_config, weights
from keyword_spotting_amd.rnn_ctc import DeployModel
cfg = get_config(precision="f16x3")
m = DeployModel(cfg, weights.init_weights(cfg, seed=0))
B, T = 4096, int(os.environ.get("KWS_T", "300"))
mel = (torch.randn(B, T, cfg.n_mel, device="cuda").abs() * 2).contiguous()
st = m.zero_state(B)
for _ in range(2):
    m.forward(mel, st, state_out=st)
torch.cuda.synchronize()
os.environ["KWS_F16_TIMING"] = "1"
m.forward(mel, st, state_out=st)
torch.cuda.synchronize()

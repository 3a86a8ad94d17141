import os, subprocess, sys
ROOT = os.environ.get("GRAFT_REPO_ROOT", "/root/repo")
snippet = r"""
import sys
sys.path.insert(0, %r)
from keyword_spotting_amd import _lib, get_config, weights
from keyword_spotting_amd.rnn_ctc import DeployModel
for kw in (dict(), dict(precision="bf16"), dict(n_mel=60, hidden_size=256, num_layers=4), dict(n_mel=60, num_layers=1)):
    cfg = get_config(**kw)
    for rep in range(3):
        try:
            m = DeployModel(cfg, weights.init_weights(cfg, seed=0))
            m.selftest()
            print("PASS", kw)
        except _lib.KwsError as e:
            print("FAIL", kw, str(e)[:200])
""" % ROOT
for v in ("nofence", "noprefence", "nofences", "vgprform"):
    env = dict(os.environ, KWS_AMD_LIB=os.path.join(ROOT, "variants", "libkws_%s.so" % v))
    r = subprocess.run([sys.executable, "-c", snippet], env=env, capture_output=True, text=True)
    print("=====", v, r.returncode)
    print(r.stdout[-3000:])
    print(r.stderr[-500:])

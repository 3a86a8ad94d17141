"""A/B of libkws variants on the f16x3 headline shape: python tools/ab_f16x3.py variants/libkws_a.so variants/libkws_b.so ..."""
import os, subprocess, sys
root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for so in ["default"] + sys.argv[1:]:
    env = dict(os.environ)
    if so != "default":
        env["KWS_AMD_LIB"] = os.path.join(root, so)
    r = subprocess.run([sys.executable, os.path.join(root, "tools", "bench_precision.py"), "f16x3"], env=env, capture_output=True, text=True)
    print(os.path.basename(so), [l for l in r.stdout.splitlines() if "ms/step" in l] or r.stderr[-300:], flush=True)

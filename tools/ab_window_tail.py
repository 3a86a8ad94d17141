#!/usr/bin/env python3
"""A/B of the decode-window step inside the last GRU layer's launch against window_inc_kernel behind it (KWS_NO_WINDOW_TAIL=1):
tools/bench_e2e.py alternately, `--rounds` times each, 300 chunks per run; prints every run and the medians (us per chunk)."""
import argparse, os, re, statistics, subprocess, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
ap = argparse.ArgumentParser()
ap.add_argument("--rounds", type=int, default=5)
ap.add_argument("--batch", type=int, default=4096)
ap.add_argument("--precisions", default="fp32,f16x3,bf16")
a = ap.parse_args()
for prec in a.precisions.split(","):
    res = {"tail": [], "own launch": []}
    for r in range(a.rounds):
        for name, env in (("tail", {}), ("own launch", {"KWS_NO_WINDOW_TAIL": "1"})):
            out = subprocess.run([sys.executable, os.path.join(ROOT, "tools", "bench_e2e.py"), "--precision", prec, "--chunks", "300", "--batch", str(a.batch)],
                                 env=dict(os.environ, **env), capture_output=True, text=True).stdout
            m = re.search(r"precision=\S+: ([0-9.]+) ms per 225 ms chunk", out)
            if m:
                res[name].append(float(m.group(1)) * 1e3)
    print("%-6s B=%d  window in the last layer's launch: median %.1f us (%s)   as a launch of its own: median %.1f us (%s)" % (
        prec, a.batch, statistics.median(res["tail"]), " ".join("%.1f" % v for v in res["tail"]),
        statistics.median(res["own launch"]), " ".join("%.1f" % v for v in res["own launch"])), flush=True)

#!/usr/bin/env python3
"""A/B of library builds on one GPU box: runs bench.py once per library (interleaved, `--rounds` times) and prints the
per-layer kernel times.  usage: tools/ab_variants.py [--rounds 2] [--args "--precision int8"] name=path.so ...
(`base` = the in-tree library)."""
import argparse, json, os, subprocess, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
ap = argparse.ArgumentParser()
ap.add_argument("--rounds", type=int, default=2)
ap.add_argument("--args", default="")
ap.add_argument("libs", nargs="+")
a = ap.parse_args()
libs = []
for spec in a.libs:
    name, _, path = spec.partition("=")
    libs.append((name, os.path.abspath(path) if path else None))
res = {n: [] for n, _ in libs}
for r in range(a.rounds):
    for name, path in libs:
        env = dict(os.environ)
        if path:
            env["KWS_AMD_LIB"] = path
        out = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--steps", "10", "--warmup", "3", "--no-cpu-baseline"] + a.args.split(),
                             env=env, capture_output=True, text=True)
        lines = [l for l in out.stdout.splitlines() if l.startswith("{")]
        if not lines:
            print(name, "FAILED", out.stderr[-500:]); continue
        d = json.loads(lines[-1])
        res[name].append((d["value"], d["roofline"]["per_layer_ms"]))
for name, _ in libs:
    for v, pl in res[name]:
        print("%-16s %8.1f M frames/s   per-layer ms %s" % (name, v / 1e6, " ".join("%.4f" % x for x in pl)))

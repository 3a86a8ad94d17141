#!/usr/bin/env python3
"""Weight converter: an .npz of TF variables of the reference graph (either naming generation) or this
package's own .npz  ->  the flat fp32 blob `kws_create` takes (+ the canonical .npz).  No TensorFlow
needed; dump a checkpoint with `np.savez(path, **{v.name: sess.run(v) for v in tf.global_variables()})`
on a machine that has it.  (Replaces the freeze/export half of main.py:316-371 for this path.)

    python tools/convert_weights.py vars.npz --n-mel 40 --out model          # model.blob + model.npz
"""
import argparse, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from keyword_spotting_amd import get_config, weights

ap = argparse.ArgumentParser()
ap.add_argument("src")
ap.add_argument("--out", required=True)
ap.add_argument("--n-mel", type=int, default=40)
ap.add_argument("--hidden", type=int, default=128)
ap.add_argument("--layers", type=int, default=2)
a = ap.parse_args()
cfg = get_config(n_mel=a.n_mel, hidden_size=a.hidden, num_layers=a.layers)
z = np.load(a.src)
if any("gru_cell" in k for k in z.files):
    w = weights.from_tf_variables(cfg, {k: z[k] for k in z.files})
else:
    w = weights.load_npz(a.src)
    weights.check_shapes(cfg, w)
blob = weights.to_blob(cfg, w)
blob.tofile(a.out + ".blob")
weights.save_npz(a.out + ".npz", w)
print("%s: %d floats (%d bytes) -> %s.blob, %s.npz" % (a.src, blob.size, blob.nbytes, a.out, a.out))

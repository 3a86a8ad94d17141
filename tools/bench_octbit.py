import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from keyword_spotting_amd.octbit_ops import octbit_mat_mul
rng = np.random.default_rng(0)
for (a, k, n) in ((4096, 256, 256), (4096, 256, 128), (4096, 128, 6), (1, 256, 256)):
    x = torch.randn(a, k, device="cuda")
    wq = torch.from_numpy(rng.integers(-127, 128, (n, k)).astype(np.int8)).cuda()
    bias = (127.0 * wq.float().sum(1)).contiguous()
    for _ in range(3): octbit_mat_mul(x, wq, scale=0.01, bias=bias, per_row_scale=True)
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(20): octbit_mat_mul(x, wq, scale=0.01, bias=bias, per_row_scale=True)
    torch.cuda.synchronize(); dt = (time.perf_counter() - t0) / 20
    print("octbit [%d,%d]x[%d,%d]^T per-row: %.1f us  %.2f TMAC/s" % (a, k, n, k, dt * 1e6, a * k * n / dt / 1e12))

import sys, os, time, threading
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
import torch
from keyword_spotting_amd import get_config, weights, sharding
from keyword_spotting_amd.rnn_ctc import DeployModel
dev = torch.device("cuda", 0)
path = sharding.find_sclk_path(dev)
def sample(fn, label, n):
    fn(); torch.cuda.synchronize()
    vals = []
    stop = threading.Event()
    def poll():
        while not stop.is_set():
            v = sharding.read_sclk_file(path)
            if v: vals.append(v)
            time.sleep(0.01)
    th = threading.Thread(target=poll); th.start()
    t0 = time.perf_counter()
    for _ in range(n): fn()
    torch.cuda.synchronize()
    dt = (time.perf_counter() - t0) / n
    stop.set(); th.join()
    print("%-40s %.3f ms  clock MHz: median %s min %s max %s (%d samples)" % (label, dt * 1e3, sorted(vals)[len(vals)//2] if vals else None, min(vals) if vals else None, max(vals) if vals else None, len(vals)), flush=True)
for prec in ("f16x3", "fp32"):
    cfg = get_config(n_mel=60, hidden_size=256, num_layers=4, precision=prec)
    m = DeployModel(cfg, weights.init_weights(cfg, seed=0))
    mel = (torch.randn(1024, 300, 60, device=dev).abs() * 2).contiguous()
    st = m.zero_state(1024)
    sample(lambda: m.forward(mel, st, state_out=st), "configs[4] %s 1024x300" % prec, 60)
    m.close()

import sys, os, time
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
import torch
from keyword_spotting_amd import get_config, weights
from keyword_spotting_amd.rnn_ctc import DeployModel
for prec in ("fp32", "f16x3", "bf16"):
    cfg = get_config(precision=prec)
    m = DeployModel(cfg, weights.init_weights(cfg, seed=0))
    B, T = 4096, 22
    mel = (torch.randn(B, T, 40, device="cuda").abs() * 2).contiguous()
    st = m.zero_state(B); pw = m.fresh_prev_word(B)
    out = {"logits": torch.empty(B, T, 6, device="cuda"), "softmax": torch.empty(B, T, 6, device="cuda"), "tokens": torch.empty(B, T, dtype=torch.int8, device="cuda")}
    m.reserve(B, T)
    # eager reference: 8 calls
    for _ in range(8): m.forward(mel, st, prev_word=pw, state_out=st, out=out)
    torch.cuda.synchronize()
    ref_state, ref_logits = st.clone(), out["logits"].clone()
    st.zero_(); pw.fill_(-1)
    s = torch.cuda.Stream()
    g = torch.cuda.CUDAGraph()
    with torch.cuda.stream(s):
        m.forward(mel, st, prev_word=pw, state_out=st, out=out)       # warm-up on the capture stream
        s.synchronize()
        st.zero_(); pw.fill_(-1)
        s.synchronize()
        with torch.cuda.graph(g, stream=s):
            m.forward(mel, st, prev_word=pw, state_out=st, out=out)
    for _ in range(8): g.replay()
    torch.cuda.synchronize()
    ok = torch.equal(st, ref_state) and torch.equal(out["logits"], ref_logits)
    # timing: 64 eager calls vs 64 replays
    def timeit(fn, n=64):
        fn(); torch.cuda.synchronize(); t0 = time.perf_counter()
        for _ in range(n): fn()
        torch.cuda.synchronize(); return (time.perf_counter() - t0) / n * 1e3
    te = timeit(lambda: m.forward(mel, st, prev_word=pw, state_out=st, out=out))
    tg = timeit(lambda: g.replay())
    print("%s: graph-captured kws_step replays bit-identical: %s; eager %.4f ms/call, graph replay %.4f ms/call" % (prec, ok, te, tg), flush=True)
    m.close()

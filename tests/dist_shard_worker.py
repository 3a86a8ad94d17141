"""One rank of the partitioning test on the real library (started by tests/test_gpu_dist.py through torch.distributed.run,
gloo rendezvous on 127.0.0.1): computes its contiguous shard of a seeded global batch with libkws_amd.so and saves the
results.  Not a test module itself."""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np  # noqa: E402
import torch  # noqa: E402
import torch.distributed as dist  # noqa: E402

from keyword_spotting_amd import get_config, sharding, weights  # noqa: E402
from keyword_spotting_amd.rnn_ctc import DeployModel  # noqa: E402


def main():
    out_dir, total, frames, precision = sys.argv[1], int(sys.argv[2]), int(sys.argv[3]), sys.argv[4]
    rank, local_rank, world = sharding.env_rank_world()
    os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
    dist.init_process_group("gloo")
    try:
        device = torch.device("cuda", local_rank % torch.cuda.device_count())
        torch.cuda.set_device(device)
        cfg = get_config(precision=precision)
        model = DeployModel(cfg, weights.init_weights(cfg, seed=0), device=device)
        gen = torch.Generator().manual_seed(1234)                       # the GLOBAL batch, identical on every rank
        mel = (torch.randn(total, frames, cfg.n_mel, generator=gen).abs() * 2)
        state = 0.1 * torch.randn(cfg.num_layers, total, cfg.hidden_size, generator=gen)
        lo, hi = sharding.shard_bounds(total, rank, world)
        sharding.barrier(dist, torch.device("cpu"))
        r = model.forward(mel[lo:hi].contiguous(), state[:, lo:hi].contiguous(), prev_word=model.fresh_prev_word(hi - lo))
        np.savez(os.path.join(out_dir, "shard_%d.npz" % rank), lo=lo, hi=hi, logits=r["logits"].cpu().numpy(),
                 softmax=r["softmax"].cpu().numpy(), state=r["state"].cpu().numpy(), tokens=r["tokens"].cpu().numpy())
        frames_all, _ = sharding.reduce_throughput(dist, (hi - lo) * frames, 1.0, torch.device("cpu"))
        assert frames_all == total * frames
        sharding.barrier(dist, torch.device("cpu"))
    finally:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()

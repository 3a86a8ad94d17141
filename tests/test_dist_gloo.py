"""N>1 path on CPU: two gloo ranks shard a stream batch with no data-path collective; the
concatenated shard results equal the unsharded result and the throughput reduction aggregates
frames (SUM) and time (MAX).  The per-shard compute stands in via the oracle (no GPU here)."""
import os
import socket

import numpy as np
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

from oracle import gru_oracle as G


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    return port


def _worker(rank, world, port, total, tmpdir):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world),
                      LOCAL_RANK=str(rank))
    from keyword_spotting_amd import sharding
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        r, lr, w = sharding.env_rank_world()
        assert (r, lr, w) == (rank, rank, world)
        lo, hi = sharding.shard_bounds(total, rank, world)
        weights = G.init_weights()
        mel = G.synthetic_mel(total, 12, 40, seed=9)          # the global batch, identical on every rank
        sharding.barrier(dist, torch.device("cpu"))
        logits, state = G.gru_forward(weights, mel[lo:hi])
        np.save(os.path.join(tmpdir, "logits_%d.npy" % rank), logits)
        np.save(os.path.join(tmpdir, "state_%d.npy" % rank), state)
        frames, seconds = sharding.reduce_throughput(dist, (hi - lo) * 12, 1.0 + rank, torch.device("cpu"))
        assert frames == total * 12 and seconds == float(world)
        sharding.barrier(dist, torch.device("cpu"))
    finally:
        dist.destroy_process_group()


def test_two_rank_sharding_equals_unsharded(tmp_path):
    world, total = 2, 11                                       # ragged: 6 + 5 streams
    mp.spawn(_worker, args=(world, _free_port(), total, str(tmp_path)), nprocs=world, join=True)
    weights = G.init_weights()
    mel = G.synthetic_mel(total, 12, 40, seed=9)
    want_l, want_s = G.gru_forward(weights, mel)
    got_l = np.concatenate([np.load(str(tmp_path / ("logits_%d.npy" % r))) for r in range(world)], 0)
    got_s = np.concatenate([np.load(str(tmp_path / ("state_%d.npy" % r))) for r in range(world)], 1)
    np.testing.assert_allclose(got_l, want_l, atol=1e-5, rtol=0)     # BLAS blocking varies with shard size
    np.testing.assert_allclose(got_s, want_s, atol=1e-6, rtol=0)


def test_single_process_reduce_is_identity():
    from keyword_spotting_amd import sharding
    assert sharding.reduce_throughput(None, 1200, 0.5, torch.device("cpu")) == (1200, 0.5)

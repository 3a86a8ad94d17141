"""The N>1 path of bench.py on real HIP: two ranks (one process each, started by bench.py's own launcher through
torch.distributed.run, rendezvous on 127.0.0.1) drive two instances of libkws_amd.so.  The box has one GPU, so both ranks
land on cuda:0 (local_rank % device_count) and the process group is gloo -- what is exercised is the launcher, the
rank/device plumbing, two library instances side by side, the barriers around the timed region and the SUM/MAX
reduction, i.e. everything of BASELINE configs[3] except RCCL itself and the other seven GPUs.  RCCL itself is initialised and
used by the last test: one rank under torch.distributed.run with the nccl backend (--force-dist), so that the opening and
closing barriers, the SUM/MAX reductions and the per-rank gather of the N > 1 path have run over RCCL on this image at
least once before an 8-GPU node sees them."""
import json
import os
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _bench(*args):
    env = dict(os.environ)
    env.pop("RANK", None), env.pop("WORLD_SIZE", None), env.pop("LOCAL_RANK", None)
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    # a child process started before it touches the GPU (never an exec from this, GPU-initialised, process image)
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py")] + list(args), cwd=ROOT, env=env, capture_output=True,
                       text=True, timeout=900)
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-4000:]
    lines = [ln for ln in r.stdout.splitlines() if ln.startswith("{")]
    assert len(lines) == 1, r.stdout[-2000:]                    # rank 0 alone prints, once
    return json.loads(lines[0])


def test_two_ranks_on_one_gpu_drive_the_hip_library_and_aggregate():
    common = ["--steps", "3", "--warmup", "1", "--no-cpu-baseline"]
    two = _bench("--gpus", "2", "--dist-backend", "gloo", "--batch", "1024", *common)
    assert two["n_gpus"] == 2 and two["ranks_seen"] == 2 and two["scaling"] == "weak"
    assert two["config"]["kernel"] != "stub" and two["config"]["streams_per_gpu"] == 1024
    assert two["roofline"]["launches"] == 3 and two["roofline"]["kernel_ms"] > 0
    # the same total work (2 x 1024 streams) in one process on the same GPU: the two ranks share the chip, so the aggregate
    # must land within 15 % of the single-process figure -- far off would mean a rank did not run, ran twice, or the reduction is wrong
    one = _bench("--gpus", "1", "--batch", "2048", *common)
    assert one["n_gpus"] == 1 and one["ranks_seen"] == 1
    ratio = two["value"] / one["value"]
    print("two ranks on one GPU: %.1f M frames/s aggregate, one process on the same total batch: %.1f M (ratio %.2f)"
          % (two["value"] / 1e6, one["value"] / 1e6, ratio))
    assert 0.85 < ratio < 1.15, (two["value"], one["value"])          # measured 0.98: the two processes' kernels share the chip
    assert two["ms_per_step"] > 0 and abs(two["value"] - 2 * 1024 * 300 / (two["ms_per_step"] * 1e-3)) < 1e-6 * two["value"]


def test_rccl_process_group_path_at_one_rank():
    import socket
    sock = socket.socket()
    sock.bind(("127.0.0.1", 0))
    port = sock.getsockname()[1]
    sock.close()
    env = dict(os.environ)
    env.pop("RANK", None), env.pop("WORLD_SIZE", None), env.pop("LOCAL_RANK", None)
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "1", "--master-addr", "127.0.0.1",
           "--master-port", str(port), os.path.join(ROOT, "bench.py"), "--gpus", "1", "--steps", "3", "--warmup", "1",
           "--no-cpu-baseline", "--sustain-seconds", "0", "--force-dist"]
    r = subprocess.run(cmd, cwd=ROOT, env=env, capture_output=True, text=True, timeout=900)
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-4000:]
    lines = [ln for ln in r.stdout.splitlines() if ln.startswith("{")]
    assert len(lines) == 1, r.stdout[-2000:]
    line = json.loads(lines[0])
    assert line["n_gpus"] == 1 and line["ranks_seen"] == 1 and len(line["per_rank"]) == 1 and line["distinct_devices"] == 1
    assert line["per_rank"][0]["device"] == "cuda:0" and line["value"] > 1e8


@pytest.mark.parametrize("world,precision", [(2, "fp32"), (3, "fp32"), (2, "bf16")])
def test_sharded_results_equal_the_unsharded_run_bitwise(world, precision, tmp_path):
    """the same seeded global batch partitioned over `world` ranks (ragged: 37 streams) gives, shard by
    shard, exactly the bytes of the one-process run -- streams are independent and a stream's result does not depend on
    the batch it sits in, so partitioning is invisible.  Real library, one process per rank, all on cuda:0."""
    import socket
    import numpy as np
    import torch
    from keyword_spotting_amd import get_config, weights
    from keyword_spotting_amd.rnn_ctc import DeployModel
    total, frames = 37, 23
    sock = socket.socket()
    sock.bind(("127.0.0.1", 0))
    port = sock.getsockname()[1]
    sock.close()
    env = dict(os.environ)
    env.pop("RANK", None), env.pop("WORLD_SIZE", None), env.pop("LOCAL_RANK", None)
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", str(world), "--master-addr", "127.0.0.1",
           "--master-port", str(port), os.path.join(ROOT, "tests", "dist_shard_worker.py"), str(tmp_path), str(total), str(frames), precision]
    r = subprocess.run(cmd, cwd=ROOT, env=env, capture_output=True, text=True, timeout=900)
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-4000:]
    cfg = get_config(precision=precision)
    model = DeployModel(cfg, weights.init_weights(cfg, seed=0))
    gen = torch.Generator().manual_seed(1234)
    mel = (torch.randn(total, frames, cfg.n_mel, generator=gen).abs() * 2)
    state = 0.1 * torch.randn(cfg.num_layers, total, cfg.hidden_size, generator=gen)
    want = model.forward(mel, state, prev_word=model.fresh_prev_word(total))
    covered = 0
    for rank in range(world):
        z = np.load(str(tmp_path / ("shard_%d.npz" % rank)))
        lo, hi = int(z["lo"]), int(z["hi"])
        assert lo == covered and hi > lo
        covered = hi
        np.testing.assert_array_equal(z["logits"], want["logits"][lo:hi].cpu().numpy())
        np.testing.assert_array_equal(z["softmax"], want["softmax"][lo:hi].cpu().numpy())
        np.testing.assert_array_equal(z["state"], want["state"][:, lo:hi].cpu().numpy())
        np.testing.assert_array_equal(z["tokens"], want["tokens"][lo:hi].cpu().numpy())
    assert covered == total

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


def _run_stub_bench(args, env_extra=None):
    import json
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = dict(os.environ)
    for k in ("RANK", "WORLD_SIZE", "LOCAL_RANK", "MASTER_PORT"):
        env.pop(k, None)
    env.update(env_extra or {})
    r = subprocess.run([sys.executable] + args, cwd=root, env=env, capture_output=True, text=True, timeout=300)
    assert r.returncode == 0, r.stderr[-2000:]
    lines = [ln for ln in r.stdout.splitlines() if ln.startswith("{")]
    assert len(lines) == 1, r.stdout                           # ONE JSON line, from rank 0 only
    return json.loads(lines[0])


def test_bench_main_two_ranks_over_gloo_prints_one_line():
    """bench.py's own launcher + N>1 path (what the driver runs with --gpus N on an 8-GPU node), model stubbed at the
    DeployModel boundary: self-launch through torch.distributed.run, both ranks counted, frames summed over ranks,
    time = the slower rank's."""
    stub = os.path.join("tests", "bench_stub_main.py")
    line = _run_stub_bench([stub, "--gpus", "2", "--steps", "6", "--warmup", "2", "--batch", "64", "--frames", "10",
                            "--dist-backend", "gloo"])
    assert line["n_gpus"] == 2 and line["ranks_seen"] == 2 and line["scaling"] == "weak"
    assert line["steps"] == 6 and line["warmup"] == 2 and line["unit"] == "mel-frames/s"
    assert line["config"]["parallelism"] == "utterance-dp2"
    # rank 1 sleeps 20 ms per step: 6 steps >= 120 ms; both ranks' frames are counted
    assert line["ms_per_step"] >= 19.0
    assert abs(line["value"] - 2 * 64 * 10 * 6 / (line["ms_per_step"] * 6e-3)) < 1e-6 * line["value"]
    assert "cpu_baseline" not in line                           # rank 0 at N=1 only
    # every rank identifies itself: `world` entries, distinct (host, device) pairs, the slower rank visible
    assert [r["rank"] for r in line["per_rank"]] == [0, 1] and line["distinct_devices"] == 2
    assert all(r["host"] and r["device_id"] for r in line["per_rank"])
    assert line["per_rank"][1]["mel_frames_per_s"] < line["per_rank"][0]["mel_frames_per_s"]
    # weak scaling, time = the slowest rank's: value ~ world x the slowest rank's own rate
    assert abs(2 * min(r["mel_frames_per_s"] for r in line["per_rank"]) - line["value"]) < 0.2 * line["value"]


def test_bench_main_under_an_external_torchrun_and_single_rank():
    """The driver's own form: python -m torch.distributed.run ... bench.py --gpus N (RANK/WORLD_SIZE from the env)."""
    import sys
    stub = os.path.join("tests", "bench_stub_main.py")
    line = _run_stub_bench(["-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2", "--master-addr",
                            "127.0.0.1", "--master-port", str(_free_port()), stub, "--gpus", "2", "--steps", "3",
                            "--warmup", "1", "--batch", "32", "--frames", "8", "--dist-backend", "gloo"])
    assert line["n_gpus"] == 2 and line["ranks_seen"] == 2
    one = _run_stub_bench([stub, "--steps", "3", "--warmup", "1", "--batch", "32", "--frames", "8", "--dist-backend", "gloo"])
    assert one["n_gpus"] == 1 and one["ranks_seen"] == 1 and len(one["per_rank"]) == 1
    assert len(line["per_rank"]) == 2 and line["distinct_devices"] == 2
    assert abs(one["value"] - 32 * 8 * 3 / (one["ms_per_step"] * 3e-3)) < 1e-6 * one["value"]


def test_bench_main_eight_stub_ranks():
    """configs[3]'s launch shape (8 ranks on one node) through bench.py's own launcher, model stubbed: eight ranks counted,
    eight distinct (host, device) pairs, value = 8 x the slowest rank's own rate (weak scaling, MAX-over-ranks time), and
    every rank reports its per-layer kernel time and a clock field."""
    stub = os.path.join("tests", "bench_stub_main.py")
    line = _run_stub_bench([stub, "--gpus", "8", "--steps", "4", "--warmup", "1", "--batch", "16", "--frames", "5",
                            "--dist-backend", "gloo"], {"KWS_STUB_SLOW_RANK": "5"})
    assert line["n_gpus"] == 8 and line["ranks_seen"] == 8 and line["distinct_devices"] == 8
    assert [r["rank"] for r in line["per_rank"]] == list(range(8))
    assert line["config"]["parallelism"] == "utterance-dp8" and line["scaling"] == "weak"
    slowest = min(line["per_rank"], key=lambda r: r["mel_frames_per_s"])
    assert slowest["rank"] == 5
    assert abs(8 * slowest["mel_frames_per_s"] - line["value"]) < 0.25 * line["value"]
    assert abs(line["value"] - 8 * 16 * 5 * 4 / (line["ms_per_step"] * 4e-3)) < 1e-6 * line["value"]
    for r in line["per_rank"]:
        assert len(r["kernel_ms"]) == 2 and "clock_mhz_if_readable" in r
    assert slowest["kernel_ms"][0] > line["per_rank"][0]["kernel_ms"][0]       # the slow rank explains itself


def test_bench_refuses_more_ranks_than_gpus_before_any_rendezvous():
    """--gpus N on a box with fewer than N GPUs (here: none): ONE clear line, non-zero exit, decided by a child probe before
    the launcher starts any rank."""
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    if torch.cuda.device_count() >= 8:
        import pytest
        pytest.skip("this box has 8 GPUs")
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "WORLD_SIZE", "LOCAL_RANK", "MASTER_PORT")}
    r = subprocess.run([sys.executable, "bench.py", "--gpus", "8", "--steps", "1", "--warmup", "0"], cwd=root, env=env,
                       capture_output=True, text=True, timeout=600)
    assert r.returncode == 2, (r.returncode, r.stderr[-500:])
    msgs = [ln for ln in r.stderr.splitlines() if "bench.py --gpus 8" in ln]
    assert len(msgs) == 1 and "GPU(s) visible" in msgs[0] and "nothing was run" in msgs[0]
    assert not [ln for ln in r.stdout.splitlines() if ln.startswith("{")]

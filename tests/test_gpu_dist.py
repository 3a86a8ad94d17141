"""The N>1 path of bench.py on real HIP: two ranks (one process each, started by bench.py's own launcher through
torch.distributed.run, rendezvous on 127.0.0.1) drive two instances of libkws_amd.so.  The box has one GPU, so both ranks
land on cuda:0 (local_rank % device_count) and the process group is gloo -- what is exercised is the launcher, the
rank/device plumbing, two library instances side by side, the barriers around the timed region and the SUM/MAX
reduction, i.e. everything of BASELINE configs[3] except RCCL itself and the other seven GPUs."""
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

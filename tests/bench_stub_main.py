"""Runs bench.py's own main() -- launcher, rank plumbing, barriers, reduction, JSON line -- with the model stubbed at
the DeployModel boundary, so the N>1 path the driver executes on an 8-GPU node is exercised on CPU (gloo).
Not a test module itself: tests/test_dist_gloo.py starts it as a subprocess."""
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench  # noqa: E402


class StubModel(object):
    """DeployModel's surface as bench.py uses it; a step costs a fixed sleep per rank (rank 1 slower, so that the
    MAX-over-ranks time is observable in the reported value)."""
    kernel = "stub"

    def __init__(self, cfg, weights, device, kernel):
        import torch
        self.cfg, self.device, self.torch = cfg, device, torch
        self.rank = int(os.environ.get("RANK", "0"))
        self.calls, self.profiling, self.timed = 0, False, 0

    def reserve(self, b, t):
        self.reserved = (b, t)

    def zero_state(self, b):
        return self.torch.zeros(self.cfg.num_layers, b, self.cfg.hidden_size)

    def fresh_prev_word(self, b):
        return self.torch.full((b,), -1, dtype=self.torch.int32)

    def forward(self, mel, state, prev_word=None, state_out=None, out=None):
        assert tuple(mel.shape[:2]) == self.reserved
        slow = os.environ.get("KWS_STUB_SLOW_RANK")          # default: rank r sleeps (1 + r) x 10 ms; with the variable, only that rank is slow
        time.sleep(0.01 * (1 + self.rank) if slow is None else (0.04 if self.rank == int(slow) else 0.01))
        self.calls += 1
        if self.profiling:
            self.timed += 1

    def set_profiling(self, on):
        self.profiling = bool(on)

    def kernel_times(self, reset=True):
        k = 1.0 + self.rank if os.environ.get("KWS_STUB_SLOW_RANK") is None else (4.0 if self.rank == int(os.environ["KWS_STUB_SLOW_RANK"]) else 1.0)
        r = [(4.0 * k * self.timed, self.timed), (6.0 * k * self.timed, self.timed)]
        if reset:
            self.timed = 0
        return r


if __name__ == "__main__":
    bench.main(model_factory=StubModel)

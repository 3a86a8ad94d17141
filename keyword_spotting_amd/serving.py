"""StreamServer -- HotwordDetector.start's loop (detector.py:148-212) for a whole GPU's worth of microphones.

The reference serves ONE stream: every 225 ms its ring buffer hands over 3600 samples (detector.py:119: sleep_time 0.225 s
of 16 kHz audio) and one loop iteration (:158-209) must finish before the next chunk is due.  Here N = M x S distinct streams
are resident on one device -- M StreamManagers of S streams each (state, sample carry and decode window of every stream stay
in HBM between its chunks) -- and one *period* feeds every manager one chunk: M kws_stream_feed calls, issued in turn on a
few HIP streams, each with a model handle of its own (the unit of concurrency of the C ABI, include/kws_amd.h) that its
managers share (weights, inter-layer seams and one chunk's staging once per handle, not per manager).  A period meets its
deadline when the last manager's trigger decisions exist before the next chunks are due.

Host code only: every per-stream decision is on the device (kws_stream_feed); there is no CPU fallback.
"""
import time

import torch

from . import weights as _weights
from .detector import StreamManager
from .frontend import MelFrontend
from .rnn_ctc import DeployModel


class StreamServer(object):
    def __init__(self, config, weights=None, device="cuda:0", streams_per_manager=16384, handles=2, label=None,
                 window_chunks=15, max_frames=32, vad_thres=30, decode_thres=0.4):
        self.config, self.device = config, torch.device(device)
        self.streams_per_manager = int(streams_per_manager)
        w = weights if weights is not None else _weights.init_weights(config, seed=0)
        self.models = [DeployModel(config, w, device=self.device) for _ in range(int(handles))]
        self.streams = [torch.cuda.Stream(device=self.device) for _ in self.models]
        self.frontend = MelFrontend(config, device=self.device)       # immutable tables: shared by every handle
        self._mgr_args = dict(window_chunks=window_chunks, max_frames=max_frames, vad_thres=vad_thres, label=label,
                              decode_thres=decode_thres)
        self.managers = []

    # -- population ----------------------------------------------------------------------------------------------------
    def resize(self, n_managers):
        """Grows or shrinks the population to n_managers x streams_per_manager resident streams."""
        while len(self.managers) > n_managers:
            self.managers.pop().close()
        with torch.cuda.device(self.device):
            while len(self.managers) < n_managers:
                k = len(self.managers)
                self.managers.append(StreamManager(self.models[k % len(self.models)], self.streams_per_manager, **self._mgr_args))
        # the managers' zeroed state was queued on the creating thread's current stream; the feeds run on the server's own
        torch.cuda.current_stream(self.device).synchronize()
        return self

    @property
    def n_streams(self):
        return len(self.managers) * self.streams_per_manager

    def handle_of(self, k):
        return k % len(self.models)

    # -- one period ----------------------------------------------------------------------------------------------------
    def feed_period(self, chunk_of):
        """Issues one chunk for every manager: chunk_of(k) -> [S, n] PCM (int16 or float, device resident) of manager k, called
        with the manager's HIP stream current (a host-fed caller queues its upload and the wait for it there).  Asynchronous.
        Returns the number of native calls issued."""
        for k, mgr in enumerate(self.managers):
            with torch.cuda.stream(self.streams[k % len(self.models)]):
                mgr.feed_pcm(chunk_of(k), self.frontend)
        return len(self.managers)

    def wait(self):
        for s in self.streams:
            s.synchronize()

    def hits(self):
        """[M, S] int32 trigger decisions of the last period (synchronises)."""
        self.wait()
        return torch.stack([m.hit for m in self.managers]) if self.managers else torch.zeros(0, self.streams_per_manager, dtype=torch.int32)

    def launches_per_chunk(self):
        """Kernel launches of one kws_stream_feed on this configuration: gate + front-end, the GRU launches, and the window step
        when it does not ride in the last layer's launch."""
        names = self.models[0].kernel_names()
        rides = any("window tail" in nm for nm in names)
        return 1 + sum(1 for nm in names if nm) + (0 if rides else 1)

    def close(self):
        for m in self.managers:
            m.close()
        self.managers = []
        self.frontend.close()
        for m in self.models:
            m.close()
        self.models = []


def run_paced(server, chunk_of, periods=40, period_s=0.225):
    """Feeds `periods` periods in real time: period p is issued at t0 + p x period_s (never earlier: the audio does not exist
    yet) and must be complete -- every manager's decisions on the device -- by t0 + (p + 1) x period_s.  chunk_of(p, k) is manager
    k's chunk of period p.  -> dict with the per-period compute times (issue start to completion), the deadline misses and the
    lateness of the issue itself (a period that starts late because the one before overran)."""
    compute, late_start, misses = [], [], 0
    server.wait()
    t0 = time.perf_counter() + 0.01
    for p in range(periods):
        due = t0 + p * period_s
        now = time.perf_counter()
        if now < due:
            time.sleep(due - now)
        t_issue = time.perf_counter()
        server.feed_period(lambda k: chunk_of(p, k))
        server.wait()
        t_done = time.perf_counter()
        compute.append(t_done - t_issue)
        late_start.append(max(0.0, t_issue - due))
        if t_done > due + period_s:
            misses += 1
    ordered = sorted(compute)
    pick = lambda q: ordered[min(len(ordered) - 1, int(q * len(ordered)))]       # noqa: E731
    return {"periods": periods, "period_ms": period_s * 1e3, "deadline_misses": misses,
            "compute_ms_p50": pick(0.50) * 1e3, "compute_ms_p99": pick(0.99) * 1e3, "compute_ms_max": ordered[-1] * 1e3,
            "compute_ms_min": ordered[0] * 1e3, "late_start_ms_max": max(late_start) * 1e3,
            "load": pick(0.50) / period_s}

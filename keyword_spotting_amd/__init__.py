"""MI355X-native streaming GRU keyword-spotting inference path.

Host-side mirror of the reference's interface for this path (colinsongf/keyword_spotting):
  rnn_ctc.DeployModel        <- models/rnn_ctc.py:113-166 (mel-input variant :150-153)
  prediction.ctc_decode*     <- utils/prediction.py
  queue.SimpleQueue          <- utils/queue.py
  basic_vad.vad              <- utils/basic_vad.py
  detector.HotwordDetector   <- detector.py:104-316 (state-carry loop, without audio I/O)
  octbit_ops.octbit_mat_mul  <- octbit/octbit_ops.py:17-26
  octbit_graph.*             <- octbit/octbit_graph.py:191-225
All arithmetic runs in hand-written HIP kernels behind the C ABI in include/kws_amd.h
(libkws_amd.so, built by `make -C keyword_spotting_amd/csrc`).  There is no CPU fallback: importing
a compute entry point without the library raises.
"""
from .config import Config, get_config  # noqa: F401

__all__ = ["Config", "get_config"]

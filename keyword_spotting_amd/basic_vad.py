"""vad -- utils/basic_vad.py:17-18 on the GPU (kws_vad): sum(|sig|) > thres."""
import torch

from . import _lib


def vad(sig, thres=40, return_sum=False):
    """sig: [N] (one chunk -> bool) or [B,N] (-> uint8 device tensor)."""
    lib = _lib.load()
    x = torch.as_tensor(sig)
    if x.dtype != torch.float32:
        x = x.to(torch.float32)
    single = x.dim() == 1
    if single:
        x = x.unsqueeze(0)
    if not x.is_cuda:
        x = x.to("cuda:0")
    x = x.contiguous()
    b, n = int(x.shape[0]), int(x.shape[1])
    speech = torch.zeros(b, dtype=torch.uint8, device=x.device)
    sums = torch.zeros(b, dtype=torch.float32, device=x.device)
    with torch.cuda.device(x.device):
        _lib.check(lib.kws_vad(_lib.ptr(x), b, n, float(thres), _lib.ptr(speech), _lib.ptr(sums),
                               _lib.current_stream_ptr()))
    if single:
        return (bool(speech[0].item()), float(sums[0].item())) if return_sum else bool(speech[0].item())
    return (speech, sums) if return_sum else speech

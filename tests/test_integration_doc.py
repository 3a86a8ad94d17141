"""INTEGRATION.md's reference-side binding is executed verbatim, so it cannot drift from include/kws_amd.h.

CPU: the block loads the library, its KwsConfig has the C struct's size, kws_weights_nbytes answers, and the 7th field
(`precision`) really reaches kws_create (an out-of-range value is rejected by name; a valid one gets as far as the
device check).  GPU: one `run` through the block's own functions against the fp64 oracle."""
import ctypes
import os
import re
import types

import numpy as np
import pytest

from conftest import ROOT, have_gpu


def _binding():
    text = open(os.path.join(ROOT, "INTEGRATION.md")).read()
    blocks = re.findall(r"```python\n(.*?)```", text, flags=re.S)
    assert blocks, "INTEGRATION.md lost its binding block"
    src = blocks[0]
    assert 'ctypes.CDLL("libkws_amd.so")' in src
    import torch  # noqa: F401  -- the process's HIP runtime first, as the block's comment says
    src = src.replace('ctypes.CDLL("libkws_amd.so")',
                      'ctypes.CDLL(%r)' % os.path.join(ROOT, "keyword_spotting_amd", "libkws_amd.so"))
    mod = types.ModuleType("kws_amd_binding")
    exec(compile(src, "INTEGRATION.md:binding", "exec"), mod.__dict__)
    return mod


def _header_config_fields():
    text = open(os.path.join(ROOT, "include", "kws_amd.h")).read()
    body = re.search(r"typedef struct kws_config \{(.*?)\} kws_config;", text, flags=re.S).group(1)
    body = re.sub(r"/\*.*?\*/", "", body, flags=re.S)
    return re.findall(r"\b(?:int32_t|float)\s+([a-z_]+)\s*;", body)


def test_doc_binding_struct_matches_the_header():
    b = _binding()
    assert [f[0] for f in b.KwsConfig._fields_] == _header_config_fields()
    assert ctypes.sizeof(b.KwsConfig) == b.lib.kws_sizeof_config() == 4 * len(_header_config_fields())
    from keyword_spotting_amd import _lib
    assert [f[0] for f in _lib.KwsConfig._fields_] == _header_config_fields()


def test_doc_binding_runs_on_cpu_up_to_the_device_check():
    from keyword_spotting_amd import get_config
    b = _binding()
    cfg = get_config()
    assert b.lib.kws_weights_nbytes(ctypes.byref(b.make_config(cfg))) == 657432
    blob = np.zeros(657432 // 4, np.float32)
    with pytest.raises(RuntimeError, match="unknown precision 7"):       # the field is read, not 4 bytes of garbage
        b.create(cfg, blob, precision=7)
    with pytest.raises(RuntimeError, match="config needs 657432"):
        b.create(cfg, blob[:-1])
    if not have_gpu():
        for precision in (0, 1, 2):
            with pytest.raises(RuntimeError, match="no HIP device"):
                b.create(cfg, blob, precision=precision)


@pytest.mark.gpu
def test_doc_binding_one_step_against_the_oracle():
    import torch
    from keyword_spotting_amd import get_config
    from oracle import gru_oracle as G
    b = _binding()
    cfg = get_config()
    w = G.init_weights()
    h = b.create(cfg, G.weights_to_blob(w))
    mel = G.synthetic_mel(5, 22)
    mel_d = torch.from_numpy(mel).cuda()
    state = torch.zeros(2, 5, 128, device="cuda")
    logits, softmax, state_out = torch.empty(5, 22, 6, device="cuda"), torch.empty(5, 22, 6, device="cuda"), torch.empty_like(state)
    b.run(h, mel_d.data_ptr(), state.data_ptr(), logits.data_ptr(), softmax.data_ptr(), state_out.data_ptr(), 5, 22,
          torch.cuda.current_stream().cuda_stream)
    torch.cuda.synchronize()
    want_l, want_s = G.gru_forward(w, mel, dtype=np.float64)
    assert np.abs(logits.cpu().numpy() - want_l).max() < 1e-4
    assert np.abs(state_out.cpu().numpy() - want_s).max() < 1e-4
    assert np.abs(softmax.cpu().numpy() - G.softmax(want_l)).max() < 2e-5
    assert b.lib.kws_destroy(h) == 0

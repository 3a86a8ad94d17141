import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")
    # the C-ABI library is git-ignored: on a checkout where __graft_entry__.build() has not run yet, build it once
    # (hipcc cross-compiles for gfx950 without a GPU) so that the ABI tests test the library, not its absence
    so = os.path.join(ROOT, "keyword_spotting_amd", "libkws_amd.so")
    if not os.path.exists(so) and os.path.exists("/opt/rocm/bin/hipcc"):
        import subprocess
        subprocess.call(["make", "-s", "-C", os.path.join(ROOT, "keyword_spotting_amd", "csrc")],
                        stdout=subprocess.DEVNULL, stderr=subprocess.DEVNULL)


@pytest.fixture(scope="session")
def oracle_c():
    """The C restatement (oracle/kws_oracle.c), built on demand."""
    from oracle import build
    return build.load()


@pytest.fixture(scope="session")
def golden():
    import numpy as np
    return np.load(os.path.join(ROOT, "tests", "golden", "decode_golden.npz"))


def have_gpu():
    try:
        import torch
        return torch.cuda.is_available()
    except Exception:
        return False

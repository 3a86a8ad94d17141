"""ctypes binding of libkws_amd.so (include/kws_amd.h).  Fails loudly when the library is missing."""
import ctypes
import os

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.environ.get("KWS_AMD_LIB", os.path.join(_HERE, "libkws_amd.so"))   # override: timing experiments

KWS_OK = 0
KWS_ERR_INVALID_ARGUMENT = -1
KWS_ERR_UNSUPPORTED = -2
KWS_ERR_HIP = -3
KWS_ERR_NO_DEVICE = -4
KWS_ERR_OUT_OF_MEMORY = -5
KWS_ERR_BUSY = -6
KERNEL_AUTO, KERNEL_GENERIC, KERNEL_RESIDENT = 0, 1, 2
DECODE, DECODE2, DECODE_STRICT = 0, 1, 2
FP32, BF16, INT8, F16X3 = 0, 1, 2, 3


class KwsConfig(ctypes.Structure):
    _fields_ = [("n_mel", ctypes.c_int32), ("hidden", ctypes.c_int32), ("num_layers", ctypes.c_int32),
                ("num_classes", ctypes.c_int32), ("use_relu", ctypes.c_int32),
                ("value_clip", ctypes.c_float), ("precision", ctypes.c_int32)]


class KwsFrontendConfig(ctypes.Structure):
    _fields_ = [("samplerate", ctypes.c_int32), ("fft_size", ctypes.c_int32), ("hop_size", ctypes.c_int32),
                ("n_mel", ctypes.c_int32), ("fmin", ctypes.c_float), ("fmax", ctypes.c_float)]


class KwsError(RuntimeError):
    """Base of the errors the C ABI reports."""

    def __init__(self, code, msg):
        RuntimeError.__init__(self, "kws_amd error %d: %s" % (code, msg))
        self.code = code


class InvalidArgumentError(KwsError, ValueError):
    """Mirror of tf.errors.InvalidArgumentError on the reference's boundary."""


class UnsupportedError(KwsError, NotImplementedError):
    pass


class BusyError(KwsError):
    """Another host thread is inside a call on the same handle (one thread at a time per handle)."""


_vp, _i, _f = ctypes.c_void_p, ctypes.c_int, ctypes.c_float
_SIGNATURES = {
    "kws_version": (ctypes.c_char_p, []),
    "kws_last_error": (ctypes.c_char_p, []),
    "kws_sizeof_config": (ctypes.c_size_t, []),
    "kws_sizeof_frontend_config": (ctypes.c_size_t, []),
    "kws_weights_nbytes": (ctypes.c_size_t, [ctypes.POINTER(KwsConfig)]),
    "kws_create": (_i, [ctypes.POINTER(KwsConfig), _vp, ctypes.c_size_t, ctypes.POINTER(_vp)]),
    "kws_destroy": (_i, [_vp]),
    "kws_set_kernel": (_i, [_vp, _i]),
    "kws_reserve": (_i, [_vp, _i, _i]),
    "kws_scratch_stats": (_i, [_vp, ctypes.POINTER(ctypes.c_size_t), ctypes.POINTER(ctypes.c_int32)]),
    "kws_poll_error": (_i, [_vp]),
    "kws_selftest": (_i, [_vp]),
    "kws_last_launch": (_i, [_vp, _i, ctypes.c_char_p, ctypes.c_size_t]),
    "kws_step": (_i, [_vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _f, _i, _i, _vp]),
    "kws_set_profiling": (_i, [_vp, _i]),
    "kws_kernel_times": (_i, [_vp, _vp, _vp, _i]),
    "kws_ctc_decode": (_i, [_i, _vp, _vp, _i, _i, _i, _i, _f, _f, _vp, _vp, _i, _vp]),
    "kws_ctc_predict": (_i, [_vp, _vp, _i, _i, ctypes.c_char_p, _vp, _vp]),
    "kws_vad": (_i, [_vp, _i, _i, _f, _vp, _vp, _vp]),
    "kws_frontend_create": (_i, [ctypes.POINTER(KwsFrontendConfig), ctypes.POINTER(_vp)]),
    "kws_frontend_destroy": (_i, [_vp]),
    "kws_frontend_frames": (_i, [ctypes.POINTER(KwsFrontendConfig), _i]),
    "kws_frontend_run": (_i, [_vp, _vp, _i, _i, _vp, _vp]),
    "kws_frontend_run_carry": (_i, [_vp, _vp, _i, _vp, _i, _i, _vp, _vp, _i, _vp]),
    "kws_frontend_mel_basis": (_i, [_vp, _vp]),
    "kws_window_create": (_i, [_i, _i, _i, _i, _f, ctypes.POINTER(_vp)]),
    "kws_window_destroy": (_i, [_vp]),
    "kws_window_step": (_i, [_vp, _vp, _i, _vp, ctypes.c_char_p, _vp, _vp, _vp]),
    "kws_window_step_incremental": (_i, [_vp, _vp, _i, _vp, ctypes.c_char_p, _vp, _vp, _vp]),
    "kws_stream_create": (_i, [_vp, _vp, _vp, _i, _i, _f, ctypes.c_char_p, _vp, _vp, ctypes.POINTER(_vp)]),
    "kws_stream_destroy": (_i, [_vp]),
    "kws_stream_reset": (_i, [_vp]),
    "kws_stream_feed": (_i, [_vp, _vp, _i, _i, _vp, _vp]),
    "kws_octbit_matmul": (_i, [_vp, _vp, _f, _vp, _vp, _i, _i, _i, _i, _vp]),
    "kws_octbit_quantize": (_i, [_vp, _i, _i, _vp, _vp, _vp]),
}
EXPORTED_SYMBOLS = tuple(sorted(_SIGNATURES))

_lib = None


def load():
    """Returns the loaded library; raises ImportError if it has not been built."""
    global _lib
    if _lib is None:
        if not os.path.exists(LIB_PATH):
            raise ImportError("%s is missing -- build it with `make -C keyword_spotting_amd/csrc` "
                              "(or __graft_entry__.build()); there is no CPU fallback" % LIB_PATH)
        # PyTorch-ROCm bundles its own libamdhip64; it must be the HIP runtime of this process BEFORE
        # libkws_amd.so resolves its libamdhip64 dependency, or the two would not share devices,
        # streams and allocations (a second runtime instance sees "no HIP device").
        import torch  # noqa: F401
        lib = ctypes.CDLL(LIB_PATH)
        for name, (res, args) in _SIGNATURES.items():
            fn = getattr(lib, name)
            fn.restype, fn.argtypes = res, args
        if lib.kws_sizeof_config() != ctypes.sizeof(KwsConfig) or \
                lib.kws_sizeof_frontend_config() != ctypes.sizeof(KwsFrontendConfig):
            raise ImportError("%s was built from a different include/kws_amd.h than this binding (struct sizes differ); "
                              "rebuild it" % LIB_PATH)
        _lib = lib
    return _lib


def check(rc):
    if rc == KWS_OK:
        return
    msg = load().kws_last_error().decode("utf-8", "replace")
    if rc == KWS_ERR_INVALID_ARGUMENT:
        raise InvalidArgumentError(rc, msg)
    if rc == KWS_ERR_UNSUPPORTED:
        raise UnsupportedError(rc, msg)
    if rc == KWS_ERR_BUSY:
        raise BusyError(rc, msg)
    raise KwsError(rc, msg)


def current_stream_ptr():
    import torch
    return ctypes.c_void_p(torch.cuda.current_stream().cuda_stream)


def ptr(t):
    """Device pointer of a torch tensor (None -> NULL)."""
    return None if t is None else ctypes.c_void_p(t.data_ptr())

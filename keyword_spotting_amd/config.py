"""Model/stream configuration -- the subset of config/rnn_config.py the inference path reads."""


class Config(object):
    """Attribute bag like the reference's (config/rnn_config.py:20-99); defaults are the
    BASELINE north-star shape (n_mel=40 from README.md:17; the repo file ships n_mel=60, :63)."""

    def __init__(self, **overrides):
        self.label_dict = {"ni3": 1, "hao3": 2, "le4": 3}   # :26  0 space, 4 other, 5 ctc blank
        self.label_seqs = "1233"                            # :28
        self.fft_size = 400                                 # :57
        self.hop_size = 160                                 # :58
        self.samplerate = 16000                             # :59
        self.n_mel = 40                                     # :63 (60 in the repo file)
        self.fmin = 300                                     # :64
        self.fmax = 8000                                    # :65
        self.num_layers = 2                                 # :76
        self.value_clip = -1.0                              # :80
        self.use_relu = False                               # :83
        self.hidden_size = 128                              # :84
        self.precision = "fp32"                             # "fp32" (reference arithmetic) | "f16x3" (fp32 results on the fp16 matrix pipe, split operands) | "bf16" | "int8" (octbit graph) -- BASELINE configs[2]
        for k, v in overrides.items():
            if not hasattr(self, k):
                raise AttributeError("unknown config key %r" % k)
            setattr(self, k, type(getattr(self, k))(v) if not isinstance(getattr(self, k), dict) else v)

    @property
    def num_classes(self):                                  # :88-91
        return len(self.label_dict) + 3

    @property
    def freq_size(self):                                    # :97-99 (mfcc=False)
        return self.n_mel


def get_config(**overrides):
    return Config(**overrides)

"""mobilequant_amd: MI355X-native hot path of MobileQuant (simulated-quant forward + range calibration).

Layout: csrc/ (HIP kernels + C ABI), _lib.py (ctypes binding), ops.py (tensor wrappers),
quantization/ (the reference's qmodule API), calibration.py (generate_act_range counterpart).
"""
from . import _lib, ops  # noqa: F401
from .quantization import *  # noqa: F401,F403

__version__ = "0.1.0"

"""mobilequant_amd: MI355X-native hot path of MobileQuant (simulated-quant forward + range calibration).

Layout: csrc/ (HIP kernels + C ABI), _lib.py (ctypes binding), ops.py (tensor wrappers),
quantization/ (the reference's qmodule API), calibration.py (generate_act_range counterpart).
"""
from . import _lib, ops  # noqa: F401
from .quantization import *  # noqa: F401,F403

__version__ = "0.1.0"


def install_reference_alias():
    """Serve ``mobilellm.quantization.qmodule`` from this package, so the reference's callers
    (``ptq/mobilequant.py:20-21``, ``ptq/generate_qcfg.py:16``, ``eval/harness_eval.py:75``,
    ``device/debug.py:181``) import the MI355X path without an edit.  The rest of ``mobilellm`` (model zoo,
    utilities) keeps coming from the reference checkout when it is importable; empty parent packages are
    registered when it is not.  Call before the first ``from mobilellm.quantization.qmodule import ...``."""
    import importlib
    import sys
    import types
    from .quantization import qmodule
    for name in ("mobilellm", "mobilellm.quantization"):
        if name not in sys.modules:
            try:
                importlib.import_module(name)
            except Exception:                       # not installed (or its own imports fail): a bare namespace is enough
                pkg = types.ModuleType(name)
                pkg.__path__ = []
                sys.modules[name] = pkg
    sys.modules["mobilellm.quantization.qmodule"] = qmodule
    setattr(sys.modules["mobilellm.quantization"], "qmodule", qmodule)
    setattr(sys.modules["mobilellm"], "quantization", sys.modules["mobilellm.quantization"])
    return qmodule

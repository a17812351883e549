"""ctypes binding of libmobilequant_amd.so (the C ABI in include/mobilequant_amd.h).

This is the binding a MobileQuant maintainer would add (INTEGRATION.md shows it in isolation).  There is
deliberately NO fallback: if the library is missing or a call fails, a RuntimeError is raised.
`import torch` must precede loading so the library binds to the HIP runtime torch already loaded
(both carry the soname libamdhip64.so.7), which makes torch's streams and device pointers valid here.
"""
from __future__ import annotations

import ctypes
import os
from ctypes import POINTER, c_char_p, c_float, c_int, c_int64, c_size_t, c_void_p

import torch  # noqa: F401  (must be imported before the shared library is loaded)

_HERE = os.path.dirname(os.path.abspath(__file__))
# MQ_LIB_PATH: another build of the SAME library (A/B timing of kernel changes); never a fallback implementation
LIB_PATH = os.environ.get("MQ_LIB_PATH") or os.path.join(_HERE, "lib", "libmobilequant_amd.so")

# dtype codes of enum mq_dtype
MQ_F32, MQ_F16, MQ_I8, MQ_U8, MQ_I16, MQ_U16, MQ_I32 = range(7)
MQ_OK = 0

_P = c_void_p


class MqGrid(ctypes.Structure):
    _fields_ = [("scale", c_void_p), ("offset", c_void_p), ("qmin", c_float), ("qmax", c_float)]


class MqDecodeGemvArgs(ctypes.Structure):
    _fields_ = [("x", c_void_p), ("xq", c_void_p), ("K", c_int), ("N", c_int), ("norm_w", c_void_p), ("norm_bias", c_void_p), ("layernorm", c_int), ("norm_in", MqGrid),
                ("eps", c_float), ("a_grid", MqGrid), ("w", c_void_p), ("alpha", c_void_p), ("w_zp", c_void_p),
                ("col_term", c_void_p), ("bias", c_void_p), ("seg_end", c_int * 2), ("out_grid", MqGrid * 3),
                ("resid", c_void_p), ("y", c_void_p), ("gate_act", c_int), ("gate_mid", MqGrid), ("gate_actout", MqGrid),
                ("gate_out", MqGrid), ("gate_q", c_void_p), ("w4", c_int), ("consts", c_void_p),
                # round 6 (MQ_VERSION 300): clears o_proj's accumulators; o_proj's epilogue as a prologue
                ("zero_acc", c_void_p), ("zero_n", c_int), ("o_acc", c_void_p), ("o_alpha", c_void_p), ("o_ct", c_void_p),
                ("o_bias", c_void_p), ("o_out", MqGrid), ("x_mid", c_void_p)]


class MqDecodeAttentionOprojArgs(ctypes.Structure):
    _fields_ = [("qkv", c_void_p), ("k_cache", c_void_p), ("v_cache", c_void_p), ("rope_row", c_void_p), ("pos", c_void_p),
                ("heads", c_int), ("kv_heads", c_int), ("head_dim", c_int), ("cache_len", c_int), ("rot_dim", c_int),
                ("qk_a", MqGrid), ("qk_b", MqGrid), ("qk_out", MqGrid), ("pv_a", MqGrid),
                ("pv_b", MqGrid), ("pv_out", MqGrid), ("o_in", MqGrid), ("consts", c_void_p), ("o_w", c_void_p), ("o_wzp", c_void_p),
                ("o_acc", c_void_p), ("N", c_int), ("slices", c_int), ("tpr", c_int), ("lg_slices", c_int), ("lg_group", c_int), ("lg_kv", c_int),
                ("out_q", c_void_p), ("prefetch", c_void_p),
                ("prefetch_bytes_per_wg", c_int64), ("prefetch_stride", c_int64), ("prefetch_total", c_int64), ("prefetch_wgs", c_int),
                ("prefetch_delay", c_int), ("threads", c_int)]


class MqDecodeAttentionArgs(ctypes.Structure):
    _fields_ = [("qkv", c_void_p), ("k_cache", c_void_p), ("v_cache", c_void_p), ("cos", c_void_p), ("sin", c_void_p),
                ("pos", c_void_p), ("heads", c_int), ("kv_heads", c_int), ("head_dim", c_int), ("cache_len", c_int),
                ("rot_dim", c_int), ("nsplit", c_int), ("qk_a", MqGrid), ("qk_b", MqGrid), ("qk_out", MqGrid), ("pv_a", MqGrid),
                ("pv_b", MqGrid), ("pv_out", MqGrid), ("o_in", MqGrid), ("consts", c_void_p), ("out", c_void_p), ("out_q", c_void_p),
                ("part", c_void_p), ("ticket", c_void_p), ("prefetch", c_void_p), ("prefetch_bytes_per_wg", c_int64),
                ("prefetch_stride", c_int64), ("prefetch_total", c_int64), ("prefetch_wgs", c_int), ("prefetch_delay", c_int)]


class MqAttentionArgs(ctypes.Structure):
    _fields_ = [("q", c_void_p), ("k", c_void_p), ("v", c_void_p), ("cos", c_void_p), ("sin", c_void_p), ("seq", c_int),
                ("heads", c_int), ("kv_heads", c_int), ("head_dim", c_int), ("inv_sqrt_d", c_float), ("qk_a", MqGrid),
                ("qk_b", MqGrid), ("qk_out", MqGrid), ("pv_a", MqGrid), ("pv_b", MqGrid), ("pv_out", MqGrid), ("out", c_void_p),
                ("q_i8", c_void_p), ("k_i8", c_void_p), ("vt_i8", c_void_p), ("q_rowsum", c_void_p), ("k_rowsum", c_void_p),
                ("out_i8", c_void_p), ("out_rowsum", c_void_p), ("out_row0", c_int64), ("seq_real", c_int),
                ("out_shift", c_int), ("out_i8_tiled", c_int), ("qkv_idx", c_void_p), ("q_in", MqGrid), ("k_in", MqGrid), ("v_in", MqGrid),
                ("rot_dim", c_int), ("v_prefix", c_void_p), ("pos0", c_int), ("cache_seq", c_int), ("q_f16", c_void_p), ("k_f16", c_void_p), ("batch", c_int)]


_SIGNATURES = {
    # name: (restype, argtypes)
    "mq_version": (c_int, []),
    "mq_last_error": (c_char_p, []),
    "mq_device_info": (c_int, [POINTER(c_int), POINTER(c_int), ctypes.c_char_p, c_size_t]),
    "mq_scale_offset_from_minmax": (c_int, [_P, _P, c_int64, c_int, c_int, _P, _P, _P]),
    "mq_minmax_init": (c_int, [_P, _P, c_int64, _P]),
    "mq_minmax_tensor": (c_int, [_P, c_int, c_int64, _P, _P, _P]),
    "mq_minmax_tensor_fresh": (c_int, [_P, c_int, c_int64, _P, _P, _P, c_int64, _P]),
    "mq_minmax_rows": (c_int, [_P, c_int, c_int64, c_int64, _P, _P, _P]),
    "mq_minmax_cols": (c_int, [_P, c_int, c_int64, c_int64, _P, _P, _P]),
    "mq_fake_quant": (c_int, [_P, _P, c_int, c_int64, c_int64, _P, _P, c_int64, c_float, c_float, _P]),
    "mq_fake_quant_backward": (c_int, [_P, _P, c_int64, c_int64, _P, _P, c_int64, c_float, c_float, _P, _P, _P, _P]),
    "mq_lwc_fake_quant": (c_int, [_P, c_int64, c_int64, _P, _P, c_int, c_int, _P, _P, _P, _P, _P, _P]),
    "mq_lwc_fake_quant_backward": (c_int, [_P, _P, c_int64, c_int64, _P, _P, _P, _P, c_int, c_int, _P, _P, _P, _P]),
    "mq_attention_probs_train": (c_int, [_P, c_int64, c_int64, _P, c_int64, _P, _P, c_float, c_float, _P, _P, c_float, c_float, c_float, _P, _P]),
    "mq_attention_probs_train_backward": (c_int, [_P, _P, c_int64, c_int64, _P, c_int64, _P, _P, c_float, c_float, _P, _P, c_float, c_float,
                                                  c_float, _P, _P, _P]),
    "mq_w8a8_linear_grouped": (c_int, [_P, _P, c_int64, c_int64, c_int64, c_int64, _P, _P, _P, _P, _P, _P, _P]),
    "mq_quantize": (c_int, [_P, c_int, c_int64, c_int64, _P, _P, c_int64, c_float, c_float, c_int, _P, _P, c_int, _P, _P]),
    "mq_linear_epilogue_prepare": (c_int, [_P, _P, c_int, _P, _P, c_int64, c_int, _P, c_int64, c_int64, _P, _P, _P, _P]),
    "mq_w8a8_linear": (c_int, [_P, _P, c_int64, c_int64, c_int64, _P, _P, _P, _P, _P, _P, _P, c_float, c_float, _P, c_int, _P]),
    "mq_w8a8_linear_residual": (c_int, [_P, _P, c_int64, c_int64, c_int64, _P, _P, _P, _P, _P, _P, _P, c_float, c_float, _P, _P, _P]),
    "mq_w8a8_linear_segmented": (c_int, [_P, _P, c_int64, c_int64, c_int64, _P, _P, _P, _P, _P, c_int, POINTER(c_int64), POINTER(MqGrid), _P, _P]),
    "mq_gemm_tiled_w4_supported": (c_int, [c_int64, c_int64, c_int64]),
    "mq_w4a8_linear_tiled": (c_int, [_P, _P, c_int64, c_int64, c_int64, _P, _P, _P, _P, _P, c_int, POINTER(c_int64), POINTER(MqGrid), _P, c_int, _P]),
    "mq_w8a8_linear_tiled_segmented": (c_int, [_P, _P, c_int64, c_int64, c_int64, _P, _P, _P, _P, _P, c_int, POINTER(c_int64), POINTER(MqGrid), _P, _P]),
    "mq_w8a8_linear_tiled_residual": (c_int, [_P, _P, c_int64, c_int64, c_int64, _P, _P, _P, _P, _P, _P, _P, c_float, c_float, _P, _P, _P]),
    "mq_w4a8_linear_tiled_residual": (c_int, [_P, _P, c_int64, c_int64, c_int64, _P, _P, _P, _P, _P, _P, _P, c_float, c_float, _P, _P, _P]),
    "mq_gemm_tiled128_supported": (c_int, [c_int64, c_int64, c_int64]),
    "mq_gemm_set_residual_tile": (c_int, [c_int]),
    "mq_gemm_set_segmented_tile": (c_int, [c_int]),
    "mq_quantize_tiled_set_staged": (c_int, [c_int]),
    "mq_attention_set_cache": (c_int, [c_int]),
    "mq_norm_tiled_set_rows": (c_int, [c_int]),
    "mq_quantize_tiled_set_rows": (c_int, [c_int]),
    "mq_attention_set_fused_q": (c_int, [c_int]),
    "mq_attention_set_f16": (c_int, [c_int]),
    "mq_attention_set_pair": (c_int, [c_int]),
    "mq_w4a8_linear_segmented": (c_int, [_P, _P, c_int64, c_int64, c_int64, _P, _P, _P, _P, _P, c_int, POINTER(c_int64), POINTER(MqGrid), _P, _P]),
    "mq_gated_table": (c_int, [c_int, _P, _P, _P, _P, _P, _P, c_float, c_float, _P, _P, c_float, c_float, _P, _P, c_float, c_float, c_int, _P, _P]),
    "mq_gated_lookup": (c_int, [_P, _P, c_int64, c_int64, _P, _P, _P, _P]),
    "mq_gated_lookup_tiled": (c_int, [_P, _P, c_int64, c_int64, _P, _P, _P, _P]),
    "mq_gemm_tiled_supported": (c_int, [c_int64, c_int64, c_int64]),
    "mq_quantize_tiled": (c_int, [_P, c_int, c_int64, c_int64, _P, _P, c_float, c_float, c_int, _P, _P, _P, _P]),
    "mq_w8a8_linear_tiled": (c_int, [_P, _P, c_int64, c_int64, c_int64, _P, _P, _P, _P, _P, _P, _P, c_float, c_float, _P, c_int, _P]),
    "mq_w8a8_linear_tiled_pair": (c_int, [_P, c_int64, c_int64, c_int64, _P] + [_P] * 16 + [c_int, _P]),
    "mq_w8a8_linear_tiled_gated": (c_int, [_P, c_int64, c_int64, c_int64, _P] + [_P] * 14 + [_P, _P, _P, _P, _P]),
    "mq_w4a8_linear_tiled_gated": (c_int, [_P, c_int64, c_int64, c_int64, _P] + [_P] * 14 + [_P, _P, _P, _P, _P]),
    "mq_w8a8_linear_f32in": (c_int, [_P, _P, _P, c_float, c_float, c_int, _P, c_int64, c_int64, c_int64, _P, _P, _P, _P, _P, _P,
                                     c_float, c_float, _P, c_int, _P]),
    "mq_w4a8_linear_f32in": (c_int, [_P, _P, _P, c_float, c_float, c_int, _P, c_int64, c_int64, c_int64, _P, _P, _P, _P, _P, _P,
                                     c_float, c_float, _P, c_int, _P]),
    "mq_pack_w4": (c_int, [_P, c_int64, c_int64, _P, _P]),
    "mq_act_quant": (c_int, [_P, c_int64, c_int, _P, _P, c_float, c_float, _P, _P, c_float, c_float, _P, _P, c_float, c_float, _P, _P]),
    "mq_gated_act_quant": (c_int, [_P, _P, c_int, c_int64, c_int64, c_int, _P, _P, _P, _P, _P, _P, c_float, c_float, _P, _P, c_float, c_float,
                                   _P, _P, c_float, c_float, c_int, _P, _P, _P, _P]),
    "mq_rmsnorm_quant": (c_int, [_P, c_int64, c_int64, _P, _P, c_float, _P, _P, c_float, c_float, _P, _P, c_float, c_float, _P,
                                 _P, _P, c_int, _P, _P]),
    "mq_layernorm_quant": (c_int, [_P, c_int64, c_int64, _P, _P, c_float, _P, _P, c_float, c_float, _P, _P, c_float, c_float, _P,
                                   _P, _P, c_int, _P, _P]),
    "mq_w4a8_linear": (c_int, [_P, _P, c_int64, c_int64, c_int64, _P, _P, _P, _P, _P, _P, _P, c_float, c_float, _P, c_int, _P]),
    "mq_decode_pack_grids": (c_int, [POINTER(MqGrid), c_int, _P, _P]),
    "mq_decode_gemv": (c_int, [POINTER(MqDecodeGemvArgs), _P]),
    "mq_decode_gemv_geometry": (c_int, [POINTER(MqDecodeGemvArgs), POINTER(c_int64), POINTER(c_int64), POINTER(c_int64)]),
    "mq_decode_attention": (c_int, [POINTER(MqDecodeAttentionArgs), _P]),
    "mq_decode_attention_oproj": (c_int, [POINTER(MqDecodeAttentionOprojArgs), _P]),
    "mq_decode_embed": (c_int, [_P, _P, c_int64, c_int64, _P, _P, _P, c_int, c_int, _P, _P, _P]),
    "mq_decode_head": (c_int, [_P, _P, _P, c_int, c_float, _P, _P, c_int64, c_int64, _P, _P]),
    "mq_attention_quant": (c_int, [POINTER(MqAttentionArgs), _P]),
    "mq_calib_attention_probs": (c_int, [_P, _P, c_int64, c_int64, _P, c_int64, ctypes.c_double, _P, _P, _P, _P, _P]),
    "mq_calib_attention_probs_causal": (c_int, [_P, _P, c_int64, c_int64, ctypes.c_double, c_int, _P, _P, _P, _P, _P]),
    "mq_calib_norm": (c_int, [_P, _P, _P, _P, c_int64, c_int64, _P, _P, ctypes.c_float, c_int, _P, _P, _P, _P, _P, _P, _P]),
    "mq_calib_gated": (c_int, [_P, _P, _P, c_int64, c_int, _P, _P]),
    "mq_calib_rope": (c_int, [_P, _P, _P, _P, c_int64, c_int64, c_int, c_int, c_int, c_int, _P, _P, _P, _P]),
    "mq_calib_rope_qkv": (c_int, [_P, _P, _P, _P, _P, _P, c_int64, c_int64, c_int, c_int, c_int, c_int, _P, _P, _P, _P]),
    "mq_qmatmul": (c_int, [_P, _P, _P, c_int64, c_int64, c_int64, c_int64, c_int, POINTER(MqGrid), POINTER(MqGrid), POINTER(MqGrid), _P]),
    "mq_gemm_set_variant": (c_int, [c_int]),
    "mq_gemm_variant_name": (c_char_p, [c_int]),
    "mq_gemm_set_debug": (c_int, [c_int]),
    "mq_gemm_set_clock_probe": (c_int, [_P]),
    "mq_gemm_set_w4_mode": (c_int, [c_int]),
    "mq_gemm_set_group_m": (c_int, [c_int]),
    "mq_gemm_set_pair_mode": (c_int, [c_int]),
}
EXPORTED_SYMBOLS = tuple(_SIGNATURES)
HEADER_MAJOR = 3          # MQ_VERSION / 100 of the header the ctypes structs in this file mirror

_lib = None


class MobileQuantLibraryError(RuntimeError):
    pass


def load() -> ctypes.CDLL:
    """Load (once) and return the library.  Raises loudly if it has not been built."""
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(LIB_PATH):
        raise MobileQuantLibraryError(
            f"{LIB_PATH} is missing: build it with `python -m mobilequant_amd.build` (hipcc, gfx950). "
            "mobilequant_amd has no CPU or PyTorch fallback for its kernels.")
    lib = ctypes.CDLL(LIB_PATH)
    for name, (res, args) in _SIGNATURES.items():
        fn = getattr(lib, name)          # AttributeError if the header and the library disagree
        fn.restype, fn.argtypes = res, args
    # the argument structs above mirror include/mobilequant_amd.h of ONE major version (MQ_VERSION / 100): structs grow at their tails
    # between majors, so a library of another major would read past (or short of) what this package passes
    if lib.mq_version() // 100 != HEADER_MAJOR:
        raise MobileQuantLibraryError(f"libmobilequant_amd.so reports version {lib.mq_version()}, this Python package is built against "
                                      f"major {HEADER_MAJOR} (include/mobilequant_amd.h MQ_VERSION): rebuild with `python -m mobilequant_amd.build --force`")
    _lib = lib
    return lib


def call(name: str, *args) -> None:
    """Invoke an entry point that returns mq_status; raise with the library's message on error."""
    lib = load()
    rc = getattr(lib, name)(*args)
    if rc != MQ_OK:
        msg = lib.mq_last_error().decode("utf-8", "replace")
        raise MobileQuantLibraryError(f"{name} failed (status {rc}): {msg}")


def device_info():
    lib = load()
    cu, khz = c_int(0), c_int(0)
    buf = ctypes.create_string_buffer(64)
    rc = lib.mq_device_info(ctypes.byref(cu), ctypes.byref(khz), buf, 64)
    if rc != MQ_OK:
        raise MobileQuantLibraryError("mq_device_info: " + lib.mq_last_error().decode())
    return {"cu_count": cu.value, "max_clock_khz": khz.value, "arch": buf.value.decode()}

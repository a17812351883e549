"""``torch.ops.mobilequant_amd.*``: the C ABI's core entry points registered with ``torch.library`` ("Native
boundary").

The product path calls the C ABI through ``ops.py`` (ctypes; no dispatcher hop on the decode step's launch path).  This module
puts the same kernels behind PyTorch's operator registry so that a caller can reach them as ``torch.ops.mobilequant_amd.<op>``,
trace / export graphs that contain them (every op has a FakeTensor / meta kernel: shapes and dtypes without a device), and see
them in profiler traces under their own names.  Device implementations are registered for "cuda" (= ROCm) ONLY: a CPU tensor
has no kernel to dispatch to and raises ``NotImplementedError`` from the dispatcher -- there is no CPU path here either.

    import mobilequant_amd.torch_ops            # registers the operators
    y = torch.ops.mobilequant_amd.fake_quant(x, scale, offset, 0.0, 255.0)

Operators (reference code each one replaces):
    fake_quant(x, scale, offset, qmin, qmax) -> Tensor                              Quantizer.forward, qmodule.py:286-295
    quantize(x, scale, offset, qmin, qmax, shift, chan_scale?) -> (int8, int32)      the index half of it (+ row sums), n1
    minmax(x, per_channel) -> Tensor[2] | Tensor[2, C]                              compute_min_max_from_tensor, qmodule.py:26-34
    minmax_update_(running[2, n], x) -> ()                                         update_act_range, generate_act_range.py:49-76
    w8a8_linear(xq, row_sum, wq, alpha, w_zp, col_term, bias?, out_scale?, out_offset?, out_qmin, out_qmax) -> Tensor
                                                                                    QLinear.forward, qmodule.py:341-358
    pack_w4(nibbles) -> uint8                                                       the 4-bit image of a quantised weight
    w4a8_linear(xq, row_sum, w_packed, ...) -> Tensor                               the same linear on packed 4-bit weights
"""
from __future__ import annotations

from typing import Optional, Tuple

import torch

from . import ops
from ._lib import MQ_F32, MQ_I8

_NS = "mobilequant_amd"


@torch.library.custom_op(f"{_NS}::fake_quant", mutates_args=(), device_types="cuda")
def fake_quant(x: torch.Tensor, scale: torch.Tensor, offset: torch.Tensor, qmin: float, qmax: float) -> torch.Tensor:
    return ops.fake_quant(x, scale, offset, qmin, qmax)


@fake_quant.register_fake
def _(x, scale, offset, qmin, qmax):
    return torch.empty_like(x, memory_format=torch.contiguous_format)


@torch.library.custom_op(f"{_NS}::quantize", mutates_args=(), device_types="cuda")
def quantize(x: torch.Tensor, scale: torch.Tensor, offset: torch.Tensor, qmin: float, qmax: float, shift: int,
             chan_scale: Optional[torch.Tensor] = None) -> Tuple[torch.Tensor, torch.Tensor]:
    rows = scale.numel() if scale.numel() > 1 else x.numel() // x.shape[-1]
    q, rs = ops.quantize(x, scale, offset, qmin, qmax, q_dtype=MQ_I8, shift=shift, rows=rows, want_row_sum=True, chan_scale=chan_scale)
    return q, rs


@quantize.register_fake
def _(x, scale, offset, qmin, qmax, shift, chan_scale=None):
    rows = scale.numel() if scale.numel() > 1 else x.numel() // x.shape[-1]
    return x.new_empty(x.shape, dtype=torch.int8), x.new_empty((rows,), dtype=torch.int32)


@torch.library.custom_op(f"{_NS}::minmax", mutates_args=(), device_types="cuda")
def minmax(x: torch.Tensor, per_channel: bool) -> torch.Tensor:
    if per_channel:
        mn, mx = ops.minmax_cols(x.reshape(-1, x.shape[-1]).contiguous())
        return torch.stack((mn, mx))
    mn, mx = ops.minmax_tensor(x)
    return torch.cat((mn.reshape(1), mx.reshape(1)))


@minmax.register_fake
def _(x, per_channel):
    return x.new_empty((2, x.shape[-1]) if per_channel else (2,), dtype=torch.float32)


@torch.library.custom_op(f"{_NS}::minmax_update_", mutates_args=("running",), device_types="cuda")
def minmax_update_(running: torch.Tensor, x: torch.Tensor) -> None:
    """running [2, n] (row 0 = min, row 1 = max): n == 1 -> per-tensor statistic of x, n == x.shape[-1] -> per channel."""
    if running.shape[1] == 1:
        ops.minmax_tensor_(x, running[0], running[1])
    else:
        ops.minmax_cols_(x.reshape(-1, x.shape[-1]).contiguous(), running[0], running[1])


@minmax_update_.register_fake
def _(running, x):
    return None


def _linear(xq, row_sum, wq, alpha, w_zp, col_term, bias, out_scale, out_offset, out_qmin, out_qmax, w4):
    return ops.int8_linear(xq, wq, row_sum, alpha, w_zp, col_term, bias, out_scale=out_scale, out_offset=out_offset, out_qmin=out_qmin,
                           out_qmax=out_qmax, out_dtype=MQ_F32, w4=w4)


@torch.library.custom_op(f"{_NS}::w8a8_linear", mutates_args=(), device_types="cuda")
def w8a8_linear(xq: torch.Tensor, row_sum: torch.Tensor, wq: torch.Tensor, alpha: torch.Tensor, w_zp: torch.Tensor, col_term: torch.Tensor,
                bias: Optional[torch.Tensor] = None, out_scale: Optional[torch.Tensor] = None, out_offset: Optional[torch.Tensor] = None,
                out_qmin: float = 0.0, out_qmax: float = 255.0) -> torch.Tensor:
    return _linear(xq, row_sum, wq, alpha, w_zp, col_term, bias, out_scale, out_offset, out_qmin, out_qmax, False)


@w8a8_linear.register_fake
def _(xq, row_sum, wq, alpha, w_zp, col_term, bias=None, out_scale=None, out_offset=None, out_qmin=0.0, out_qmax=255.0):
    return xq.new_empty((xq.shape[0], wq.shape[0]), dtype=torch.float32)


@torch.library.custom_op(f"{_NS}::pack_w4", mutates_args=(), device_types="cuda")
def pack_w4(nibbles: torch.Tensor) -> torch.Tensor:
    return ops.pack_w4(nibbles)


@pack_w4.register_fake
def _(nibbles):
    return nibbles.new_empty((nibbles.shape[0], nibbles.shape[1] // 2), dtype=torch.uint8)


@torch.library.custom_op(f"{_NS}::w4a8_linear", mutates_args=(), device_types="cuda")
def w4a8_linear(xq: torch.Tensor, row_sum: torch.Tensor, w_packed: torch.Tensor, alpha: torch.Tensor, w_zp: torch.Tensor,
                col_term: torch.Tensor, bias: Optional[torch.Tensor] = None, out_scale: Optional[torch.Tensor] = None,
                out_offset: Optional[torch.Tensor] = None, out_qmin: float = 0.0, out_qmax: float = 255.0) -> torch.Tensor:
    return _linear(xq, row_sum, w_packed, alpha, w_zp, col_term, bias, out_scale, out_offset, out_qmin, out_qmax, True)


@w4a8_linear.register_fake
def _(xq, row_sum, w_packed, alpha, w_zp, col_term, bias=None, out_scale=None, out_offset=None, out_qmin=0.0, out_qmax=255.0):
    return xq.new_empty((xq.shape[0], w_packed.shape[0]), dtype=torch.float32)


OPS = ("fake_quant", "quantize", "minmax", "minmax_update_", "w8a8_linear", "pack_w4", "w4a8_linear")

"""Tensor-level wrappers over the C ABI (include/mobilequant_amd.h).

PyTorch is plumbing here: it owns device memory and the current HIP stream; every computation is a
hand-written HIP kernel in libmobilequant_amd.so.  All functions require ROCm device tensors and raise
otherwise -- there is no CPU or eager-PyTorch fallback.
"""
from __future__ import annotations

import ctypes
from typing import Optional, Tuple

import torch

from . import _lib
from ._lib import MQ_F16, MQ_F32, MQ_I8, MQ_I16, MQ_I32, MQ_U8, MQ_U16

_DT = {torch.float32: MQ_F32, torch.float16: MQ_F16}
_QDT = {MQ_I8: torch.int8, MQ_U8: torch.uint8, MQ_I16: torch.int16, MQ_U16: torch.uint16, MQ_I32: torch.int32}


def _dev(t: torch.Tensor, what: str) -> torch.Tensor:
    if not isinstance(t, torch.Tensor) or not t.is_cuda:
        raise RuntimeError(f"mobilequant_amd: {what} must be a ROCm device tensor (got "
                           f"{getattr(t, 'device', type(t))}); there is no CPU path")
    return t


def _stream() -> int:
    return torch.cuda.current_stream().cuda_stream


class _on:
    """Device guard: every tensor handed to one C-ABI call must live on the same ROCm device, and the call is made with
    that device current (so the stream is that device's current stream and the kernel launches there) -- a model placed on
    cuda:1 while cuda:0 is current must not race or fault.  No-op when the device is already current."""

    __slots__ = ("idx", "prev")

    def __init__(self, first: torch.Tensor, *rest):
        dev = first.device
        for t in rest:
            if t is not None and t.device != dev:
                raise RuntimeError(f"mobilequant_amd: tensors on different devices in one call ({dev} vs {t.device}); move the "
                                   "quantizer grids / statistics to the activation's device")
        self.idx = dev.index if dev.type == "cuda" else None
        self.prev = None

    def __enter__(self):
        if self.idx is not None and torch.cuda.current_device() != self.idx:
            self.prev = torch.cuda.current_device()
            torch.cuda.set_device(self.idx)
        return self

    def __exit__(self, *exc):
        if self.prev is not None:
            torch.cuda.set_device(self.prev)
        return False


def _fdt(t: torch.Tensor) -> int:
    try:
        return _DT[t.dtype]
    except KeyError:
        raise RuntimeError(f"mobilequant_amd: dtype {t.dtype} not supported (float32, float16)") from None


def _f32(t: torch.Tensor, what: str) -> torch.Tensor:
    _dev(t, what)
    if t.dtype != torch.float32 or not t.is_contiguous():
        t = t.to(torch.float32).contiguous()
    return t


# ---- a1 --------------------------------------------------------------------------------------------
def scale_offset_from_minmax(min_val: torch.Tensor, max_val: torch.Tensor, bitwidth: int, is_symmetric: bool):
    """Device version of compute_scale_offset_from_min_max (qmodule.py:40-61); shapes follow min_val."""
    mn, mx = _f32(min_val, "min_val"), _f32(max_val, "max_val")
    scale, offset = torch.empty_like(mn), torch.empty_like(mn)
    with _on(mn, mx):
        _lib.call("mq_scale_offset_from_minmax", mn.data_ptr(), mx.data_ptr(), mn.numel(), int(bitwidth),
                  int(bool(is_symmetric)), scale.data_ptr(), offset.data_ptr(), _stream())
    return scale, offset


# ---- a3 / a12 --------------------------------------------------------------------------------------
def minmax_new(n: int, device) -> Tuple[torch.Tensor, torch.Tensor]:
    """Fresh running statistics (min = +inf, max = -inf)."""
    mn = torch.empty(n, dtype=torch.float32, device=device)
    mx = torch.empty(n, dtype=torch.float32, device=device)
    with _on(mn, mx):
        _lib.call("mq_minmax_init", mn.data_ptr(), mx.data_ptr(), n, _stream())
    return mn, mx


def minmax_tensor_(x: torch.Tensor, mn: torch.Tensor, mx: torch.Tensor) -> None:
    """Running per-tensor update: mn[0] = min(mn[0], x.min()), mx[0] = max(mx[0], x.max())."""
    x = _dev(x, "x").contiguous()
    with _on(x, mn, mx):
        _lib.call("mq_minmax_tensor", x.data_ptr(), _fdt(x), x.numel(), mn.data_ptr(), mx.data_ptr(), _stream())


def minmax_rows_(x2d: torch.Tensor, mn: torch.Tensor, mx: torch.Tensor) -> None:
    x2d = _dev(x2d, "x").contiguous()
    rows, cols = x2d.shape
    with _on(x2d, mn, mx):
        _lib.call("mq_minmax_rows", x2d.data_ptr(), _fdt(x2d), rows, cols, mn.data_ptr(), mx.data_ptr(), _stream())


def minmax_cols_(x2d: torch.Tensor, mn: torch.Tensor, mx: torch.Tensor) -> None:
    x2d = _dev(x2d, "x").contiguous()
    rows, cols = x2d.shape
    with _on(x2d, mn, mx):
        _lib.call("mq_minmax_cols", x2d.data_ptr(), _fdt(x2d), rows, cols, mn.data_ptr(), mx.data_ptr(), _stream())


def calib_attention_probs_(raw: torch.Tensor, mask: Optional[torch.Tensor], sqrt_d: float, raw_min: torch.Tensor, raw_max: torch.Tensor,
                           probs_min: torch.Tensor, probs_max: torch.Tensor) -> torch.Tensor:
    """Calibration-mode score chain (mq_calib_attention_probs): raw [..., S, T] fp32 scores are replaced IN PLACE by
    softmax(raw / sqrt_d + mask, -1) while the running [min, max] of the raw scores and of the probabilities (1-element fp32 device
    tensors, ActRangeCollector's slots) are updated -- hf_model.py:513-530 between generate_act_range.py's two hooks.  mask: None or an
    additive fp32 [S, T] tensor.  Returns `raw` (now the probabilities)."""
    raw = _dev(raw, "raw")
    if raw.dtype != torch.float32 or not raw.is_contiguous():
        raise RuntimeError("mobilequant_amd: calib_attention_probs_ takes a contiguous float32 score tensor")
    cols = raw.shape[-1]
    rows = raw.numel() // max(cols, 1)
    mrows = 0
    if mask is not None:
        mask = _dev(mask, "mask")
        if mask.dtype != torch.float32 or not mask.is_contiguous() or mask.dim() != 2 or mask.shape[1] != cols or raw.shape[-2] != mask.shape[0]:
            raise RuntimeError("mobilequant_amd: calib_attention_probs_ mask must be a contiguous float32 [S, T] tensor")
        mrows = mask.shape[0]
    with _on(raw, mask, raw_min, raw_max, probs_min, probs_max):
        _lib.call("mq_calib_attention_probs", raw.data_ptr(), raw.data_ptr(), rows, cols, mask.data_ptr() if mask is not None else None, mrows,
                  float(sqrt_d), raw_min.data_ptr(), raw_max.data_ptr(), probs_min.data_ptr(), probs_max.data_ptr(), _stream())
    return raw


def calib_norm_(x: torch.Tensor, delta: Optional[torch.Tensor], weight: torch.Tensor, bias: Optional[torch.Tensor], eps: float, layernorm: bool,
                in_min: torch.Tensor, in_max: torch.Tensor, out_min: torch.Tensor, out_max: torch.Tensor,
                delta_min: Optional[torch.Tensor] = None, delta_max: Optional[torch.Tensor] = None):
    """Calibration-mode norm (mq_calib_norm): h = x (+ delta), y = RMSNorm / LayerNorm(h) with the running [min, max] of h (the module's
    input hook) and of y (its output hook) taken in the same pass; delta_min / delta_max: also that of delta itself.  Returns (h, y); h is x
    itself without a delta."""
    x = _dev(x, "x")
    if x.dtype != torch.float32 or not x.is_contiguous() or (delta is not None and (delta.dtype != torch.float32 or not delta.is_contiguous()
                                                                                        or delta.shape != x.shape)):
        raise RuntimeError("mobilequant_amd: calib_norm_ takes contiguous float32 tensors of one shape")
    cols = x.shape[-1]
    rows = x.numel() // max(cols, 1)
    y = torch.empty_like(x)
    h = torch.empty_like(x) if delta is not None else None
    w = _f32(weight, "weight").contiguous()
    b = _f32(bias, "bias").contiguous() if bias is not None else None
    with _on(x, delta, w, b, y, h, in_min, in_max, out_min, out_max, delta_min, delta_max):
        _lib.call("mq_calib_norm", x.data_ptr(), delta.data_ptr() if delta is not None else None, h.data_ptr() if h is not None else None,
                  y.data_ptr(), rows, cols, w.data_ptr(), b.data_ptr() if b is not None else None, float(eps), int(bool(layernorm)),
                  in_min.data_ptr(), in_max.data_ptr(), out_min.data_ptr(), out_max.data_ptr(),
                  delta_min.data_ptr() if delta_min is not None else None, delta_max.data_ptr() if delta_max is not None else None, _stream())
    return (h if h is not None else x), y


def calib_gated_(a: torch.Tensor, b: torch.Tensor, act: str, stats) -> torch.Tensor:
    """Calibration-mode act(a) * b (mq_calib_gated) with the running [min, max] of a, act(a), b and the product: stats = eight 1-element
    fp32 device tensors (min, max) x 4 in that order.  act: "silu" | "gelu" (erf).  Returns the product."""
    a, b = _dev(a, "a"), _dev(b, "b")
    if a.dtype != torch.float32 or b.dtype != torch.float32 or a.shape != b.shape or not a.is_contiguous() or not b.is_contiguous():
        raise RuntimeError("mobilequant_amd: calib_gated_ takes two contiguous float32 tensors of one shape")
    out = torch.empty_like(a)
    ptrs = (ctypes.c_void_p * 8)(*[t.data_ptr() for t in stats])
    with _on(a, b, out, *stats):
        _lib.call("mq_calib_gated", a.data_ptr(), b.data_ptr(), out.data_ptr(), a.numel(), {"silu": 0, "gelu": 1}[act], ptrs, _stream())
    return out


def calib_rope_(q_lin: torch.Tensor, k_lin: torch.Tensor, heads: int, kv_heads: int, head_dim: int, cos: torch.Tensor, sin: torch.Tensor, stats):
    """Calibration-mode RoPE (mq_calib_rope): q_lin [B, S, heads * D], k_lin [B, S, kv_heads * D] (the projections' outputs) -> q [B, heads,
    S, D], k [B, kv_heads, S, D] contiguous, rotated with cos / sin [S, rot]; stats = eight 1-element fp32 device tensors: (min, max) of
    q_lin, q, k_lin, k.  The bits of llama.apply_rope."""
    q_lin, k_lin = _dev(q_lin, "q"), _dev(k_lin, "k")
    B, S = q_lin.shape[0], q_lin.shape[1]
    cos, sin = _f32(cos, "cos").contiguous(), _f32(sin, "sin").contiguous()
    if (q_lin.dtype != torch.float32 or k_lin.dtype != torch.float32 or not q_lin.is_contiguous() or not k_lin.is_contiguous()
            or q_lin.shape[-1] != heads * head_dim or k_lin.shape[-1] != kv_heads * head_dim or cos.shape != sin.shape or cos.shape[0] != S):
        raise RuntimeError("mobilequant_amd: calib_rope_ takes the contiguous float32 outputs of q_proj / k_proj and cos / sin [S, rot]")
    q = torch.empty((B, heads, S, head_dim), dtype=torch.float32, device=q_lin.device)
    k = torch.empty((B, kv_heads, S, head_dim), dtype=torch.float32, device=q_lin.device)
    ptrs = (ctypes.c_void_p * 8)(*[t.data_ptr() for t in stats])
    with _on(q_lin, k_lin, q, k, cos, sin, *stats):
        _lib.call("mq_calib_rope", q_lin.data_ptr(), k_lin.data_ptr(), q.data_ptr(), k.data_ptr(), B, S, heads, kv_heads, head_dim, cos.shape[1],
                  cos.data_ptr(), sin.data_ptr(), ptrs, _stream())
    return q, k


def calib_rope_qkv_(q_lin: torch.Tensor, k_lin: torch.Tensor, v_lin: torch.Tensor, heads: int, kv_heads: int, head_dim: int, cos: torch.Tensor,
                    sin: torch.Tensor, stats):
    """calib_rope_ with v carried along and repeat_kv inside (mq_calib_rope_qkv): returns q [B, heads, S, D] and k, v [B, heads, S, D]
    with every kv head repeated heads / kv_heads times; stats = ten 1-element fp32 device tensors: (min, max) of q_lin, q, k_lin, k, v_lin."""
    q_lin, k_lin, v_lin = _dev(q_lin, "q"), _dev(k_lin, "k"), _dev(v_lin, "v")
    B, S = q_lin.shape[0], q_lin.shape[1]
    cos, sin = _f32(cos, "cos").contiguous(), _f32(sin, "sin").contiguous()
    if (any(t.dtype != torch.float32 or not t.is_contiguous() for t in (q_lin, k_lin, v_lin)) or q_lin.shape[-1] != heads * head_dim
            or k_lin.shape[-1] != kv_heads * head_dim or v_lin.shape != k_lin.shape or cos.shape != sin.shape or cos.shape[0] != S):
        raise RuntimeError("mobilequant_amd: calib_rope_qkv_ takes the contiguous float32 outputs of q_proj / k_proj / v_proj and cos / sin [S, rot]")
    q, k, v = (torch.empty((B, heads, S, head_dim), dtype=torch.float32, device=q_lin.device) for _ in range(3))
    ptrs = (ctypes.c_void_p * 12)(*([t.data_ptr() for t in stats] + [None, None]))
    with _on(q_lin, k_lin, v_lin, q, k, v, cos, sin, *stats):
        _lib.call("mq_calib_rope_qkv", q_lin.data_ptr(), k_lin.data_ptr(), v_lin.data_ptr(), q.data_ptr(), k.data_ptr(), v.data_ptr(), B, S, heads,
                  kv_heads, head_dim, cos.shape[1], cos.data_ptr(), sin.data_ptr(), ptrs, _stream())
    return q, k, v


def calib_attention_probs_causal_(raw: torch.Tensor, out: torch.Tensor, sqrt_d: float, store_masked: bool, raw_min: torch.Tensor, raw_max: torch.Tensor,
                                  probs_min: torch.Tensor, probs_max: torch.Tensor) -> torch.Tensor:
    """calib_attention_probs_ under the causal mask of square [S, S] blocks (mq_calib_attention_probs_causal): no mask tensor is read;
    store_masked = False leaves the quads above the diagonal of `out` (a buffer of raw's shape, not raw itself) untouched -- it must
    already hold zeros there.  Returns `out`."""
    raw, out = _dev(raw, "raw"), _dev(out, "out")
    S = raw.shape[-1]
    if (raw.dtype != torch.float32 or out.dtype != torch.float32 or not raw.is_contiguous() or not out.is_contiguous() or raw.shape != out.shape
            or raw.dim() < 2 or raw.shape[-2] != S):
        raise RuntimeError("mobilequant_amd: calib_attention_probs_causal_ takes contiguous float32 [..., S, S] tensors")
    with _on(raw, out, raw_min, raw_max, probs_min, probs_max):
        _lib.call("mq_calib_attention_probs_causal", raw.data_ptr(), out.data_ptr(), raw.numel() // max(S, 1), S, float(sqrt_d), int(bool(store_masked)),
                  raw_min.data_ptr(), raw_max.data_ptr(), probs_min.data_ptr(), probs_max.data_ptr(), _stream())
    return out


def minmax_tensor(x: torch.Tensor):
    """(min, max) of one tensor as 1-element device tensors: partials + fold, no atomics, no init launch."""
    x = _dev(x, "x").contiguous()
    buf = torch.empty(2 + 1024, dtype=torch.float32, device=x.device)     # [min | max | scratch]
    mn, mx, scratch = buf[0:1], buf[1:2], buf[2:]
    with _on(x):
        _lib.call("mq_minmax_tensor_fresh", x.data_ptr(), _fdt(x), x.numel(), mn.data_ptr(), mx.data_ptr(), scratch.data_ptr(),
                  scratch.numel(), _stream())
    return mn, mx


def minmax_rows(x2d: torch.Tensor):
    mn, mx = minmax_new(x2d.shape[0], x2d.device)
    minmax_rows_(x2d, mn, mx)
    return mn, mx


def minmax_cols(x2d: torch.Tensor):
    mn, mx = minmax_new(x2d.shape[1], x2d.device)
    minmax_cols_(x2d, mn, mx)
    return mn, mx


# ---- a5 --------------------------------------------------------------------------------------------
def _rows_cols(x: torch.Tensor, n_scale: int) -> Tuple[int, int]:
    if n_scale == 1:
        return 1, x.numel()
    if x.numel() % n_scale:
        raise RuntimeError(f"mobilequant_amd: {n_scale} scales do not divide a tensor of {x.numel()} elements")
    return n_scale, x.numel() // n_scale


def fake_quant(x: torch.Tensor, scale: torch.Tensor, offset: torch.Tensor, qmin: float, qmax: float,
               out: Optional[torch.Tensor] = None) -> torch.Tensor:
    """Quantizer.forward arithmetic (qmodule.py:286-295).  `scale`/`offset` hold 1 element (per-tensor)
    or one per row of x viewed as [n_scale, -1] (per-channel / per-group)."""
    x = _dev(x, "x").contiguous()
    s, o = _f32(scale, "scale"), _f32(offset, "offset")
    rows, cols = _rows_cols(x, s.numel())
    y = torch.empty_like(x) if out is None else out
    with _on(x, s, o, y):
        _lib.call("mq_fake_quant", x.data_ptr(), y.data_ptr(), _fdt(x), rows, cols, s.data_ptr(), o.data_ptr(),
                  s.numel(), float(qmin), float(qmax), _stream())
    return y


def fake_quant_backward(x: torch.Tensor, grad_y: torch.Tensor, scale: torch.Tensor, offset: torch.Tensor, qmin: float,
                        qmax: float):
    """Straight-through gradients of fake_quant: (grad_x, grad_scale, grad_offset), fp32."""
    x = _f32(x, "x")
    g = _f32(grad_y, "grad_y")
    s, o = _f32(scale, "scale"), _f32(offset, "offset")
    rows, cols = _rows_cols(x, s.numel())
    gx = torch.empty_like(x)
    acc = torch.zeros(2 * s.numel(), dtype=torch.float32, device=x.device)      # one fill for both accumulators
    gs, go = acc[:s.numel()].view(s.shape), acc[s.numel():].view(o.shape)
    with _on(x, g, s, o):
        _lib.call("mq_fake_quant_backward", x.data_ptr(), g.data_ptr(), rows, cols, s.data_ptr(), o.data_ptr(), s.numel(),
                  float(qmin), float(qmax), gx.data_ptr(), gs.data_ptr(), go.data_ptr(), _stream())
    return gx, gs, go


def _chan(chan_scale, cols: int, x: torch.Tensor):
    if chan_scale is None:
        return None
    cs = _f32(chan_scale, "chan_scale").reshape(-1)
    if cs.numel() != cols or x.dtype != torch.float32:
        raise RuntimeError(f"mobilequant_amd: chan_scale needs {cols} fp32 entries (got {cs.numel()}) and float32 activations")
    return cs


def lwc_fake_quant_supported(w: torch.Tensor) -> bool:
    return (w.is_cuda and w.dtype == torch.float32 and w.dim() == 2 and w.shape[1] % 4 == 0 and 0 < w.shape[1] <= 16384
            and w.is_contiguous() and w.data_ptr() % 16 == 0)


def lwc_fake_quant(w: torch.Tensor, sig_lo: torch.Tensor, sig_hi: torch.Tensor, bitwidth: int, is_symmetric: bool):
    """Per-row range -> sigmoid(bound) * range -> grid -> fake-quant of a weight [N, K] in one pass (mq_lwc_fake_quant).
    Returns (w_q, row_min, row_max, scale, offset), the last four [N]."""
    w = _f32(_dev(w, "w"), "w")
    N, K = w.shape
    lo, hi = _f32(sig_lo, "sig_lo").reshape(-1), _f32(sig_hi, "sig_hi").reshape(-1)
    if lo.numel() != N or hi.numel() != N:
        raise RuntimeError("mobilequant_amd: lwc_fake_quant needs one bound factor per output row")
    out = torch.empty_like(w)
    mn, mx, sc, of = (torch.empty(N, dtype=torch.float32, device=w.device) for _ in range(4))
    with _on(w, lo, hi):
        _lib.call("mq_lwc_fake_quant", w.data_ptr(), N, K, lo.data_ptr(), hi.data_ptr(), int(bitwidth), int(bool(is_symmetric)),
                  out.data_ptr(), mn.data_ptr(), mx.data_ptr(), sc.data_ptr(), of.data_ptr(), _stream())
    return out, mn, mx, sc, of


def lwc_fake_quant_backward(w, grad_out, sig_lo, sig_hi, row_min, row_max, bitwidth: int, is_symmetric: bool):
    """(grad_w, grad_sig_lo, grad_sig_hi) of lwc_fake_quant (mq_lwc_fake_quant_backward)."""
    w, g = _f32(_dev(w, "w"), "w"), _f32(_dev(grad_out, "grad_out"), "grad_out")
    N, K = w.shape
    lo, hi = _f32(sig_lo, "sig_lo").reshape(-1), _f32(sig_hi, "sig_hi").reshape(-1)
    gw = torch.empty_like(w)
    glo, ghi = torch.empty(N, dtype=torch.float32, device=w.device), torch.empty(N, dtype=torch.float32, device=w.device)
    with _on(w, g, lo, hi, row_min, row_max):
        _lib.call("mq_lwc_fake_quant_backward", w.data_ptr(), g.data_ptr(), N, K, lo.data_ptr(), hi.data_ptr(), row_min.data_ptr(),
                  row_max.data_ptr(), int(bitwidth), int(bool(is_symmetric)), gw.data_ptr(), glo.data_ptr(), ghi.data_ptr(), _stream())
    return gw, glo, ghi


def attention_probs_train_supported(raw: torch.Tensor, mask: Optional[torch.Tensor]) -> bool:
    if not (raw.is_cuda and raw.dtype == torch.float32 and raw.dim() >= 2 and raw.is_contiguous() and raw.shape[-1] % 4 == 0
            and 0 < raw.shape[-1] <= 4096 and raw.data_ptr() % 16 == 0):
        return False
    if mask is None:
        return True
    return (mask.is_cuda and mask.dtype == torch.float32 and mask.dim() >= 2 and all(d == 1 for d in mask.shape[:-2])
            and mask.shape[-1] == raw.shape[-1] and mask.shape[-2] == raw.shape[-2])


def _attn_probs_args(raw, mask, g1, g2):
    cols = raw.shape[-1]
    rows = raw.numel() // cols
    m = None if mask is None else _f32(mask, "mask").reshape(-1, cols)
    s1, o1, s2, o2 = (_f32(t, "grid") for t in (g1[0], g1[1], g2[0], g2[1]))
    return rows, cols, m, (s1, o1, s2, o2)


def attention_probs_train(raw, mask, grid1, grid2, sqrt_d: float):
    """Q2(softmax(Q1(raw) / sqrt_d + mask)) over the last dimension (mq_attention_probs_train); grid = (scale, offset, qmin, qmax)."""
    raw = _f32(_dev(raw, "raw"), "raw")
    rows, cols, m, (s1, o1, s2, o2) = _attn_probs_args(raw, mask, grid1, grid2)
    out = torch.empty_like(raw)
    with _on(raw, m, s1, o1, s2, o2):
        _lib.call("mq_attention_probs_train", raw.data_ptr(), rows, cols, m.data_ptr() if m is not None else None, m.shape[0] if m is not None else 1,
                  s1.data_ptr(), o1.data_ptr(), float(grid1[2]), float(grid1[3]), s2.data_ptr(), o2.data_ptr(), float(grid2[2]), float(grid2[3]),
                  float(sqrt_d), out.data_ptr(), _stream())
    return out


def attention_probs_train_backward(raw, grad_out, mask, grid1, grid2, sqrt_d: float):
    """(grad_raw, [d s1, d o1, d s2, d o2]) of attention_probs_train."""
    raw, g = _f32(_dev(raw, "raw"), "raw"), _f32(_dev(grad_out, "grad_out"), "grad_out")
    rows, cols, m, (s1, o1, s2, o2) = _attn_probs_args(raw, mask, grid1, grid2)
    graw = torch.empty_like(raw)
    gg = torch.zeros(4, dtype=torch.float32, device=raw.device)
    with _on(raw, g, m, s1, o1, s2, o2):
        _lib.call("mq_attention_probs_train_backward", raw.data_ptr(), g.data_ptr(), rows, cols, m.data_ptr() if m is not None else None,
                  m.shape[0] if m is not None else 1, s1.data_ptr(), o1.data_ptr(), float(grid1[2]), float(grid1[3]), s2.data_ptr(), o2.data_ptr(),
                  float(grid2[2]), float(grid2[3]), float(sqrt_d), graw.data_ptr(), gg.data_ptr(), _stream())
    return graw, gg


def quantize(x: torch.Tensor, scale: torch.Tensor, offset: torch.Tensor, qmin: float, qmax: float, *,
             q_dtype: int = MQ_I8, shift: int = 0, rows: Optional[int] = None, want_row_sum: bool = False,
             chan_scale: Optional[torch.Tensor] = None):
    """Integer indices (qmodule.py:286-287) as integers; optional per-row sums of the stored values.
    x is viewed as [rows, -1]; rows defaults to the number of scales (per-row) or all leading dims.
    chan_scale [cols]: SmoothQuant per-channel scale fused in front (index of x / chan_scale; per-tensor grids, fp32)."""
    x = _dev(x, "x").contiguous()
    s, o = _f32(scale, "scale"), _f32(offset, "offset")
    if rows is None:
        rows = s.numel() if s.numel() > 1 else (x.numel() // x.shape[-1] if x.dim() > 0 else 1)
    cols = x.numel() // max(rows, 1)
    q = torch.empty(x.shape, dtype=_QDT[q_dtype], device=x.device)
    rs = torch.empty(rows, dtype=torch.int32, device=x.device) if want_row_sum else None
    cs = _chan(chan_scale, cols, x)
    with _on(x, s, o, cs):
        _lib.call("mq_quantize", x.data_ptr(), _fdt(x), rows, cols, s.data_ptr(), o.data_ptr(), s.numel(), float(qmin),
                  float(qmax), int(shift), cs.data_ptr() if cs is not None else None, q.data_ptr(), q_dtype,
                  rs.data_ptr() if rs is not None else None, _stream())
    return (q, rs) if want_row_sum else q


# ---- a8 --------------------------------------------------------------------------------------------
def linear_epilogue_prepare(a_scale, a_offset, a_shift: int, w_scale, w_offset, w_shift: int, w_colsum, K: int):
    sa, oa = _f32(a_scale, "a_scale"), _f32(a_offset, "a_offset")
    sw, ow = _f32(w_scale, "w_scale"), _f32(w_offset, "w_offset")
    N = w_colsum.numel()
    dev = w_colsum.device
    alpha = torch.empty(N, dtype=torch.float32, device=dev)
    w_zp = torch.empty(N, dtype=torch.int32, device=dev)
    col_term = torch.empty(N, dtype=torch.int32, device=dev)
    with _on(w_colsum, sa, oa, sw, ow):
        _lib.call("mq_linear_epilogue_prepare", sa.data_ptr(), oa.data_ptr(), int(a_shift), sw.data_ptr(), ow.data_ptr(),
                  sw.numel(), int(w_shift), w_colsum.data_ptr(), N, int(K), alpha.data_ptr(), w_zp.data_ptr(),
                  col_term.data_ptr(), _stream())
    return alpha, w_zp, col_term


_OUT_TORCH = {MQ_F32: torch.float32, MQ_F16: torch.float16, MQ_U8: torch.uint8, MQ_I8: torch.int8,
              MQ_U16: torch.uint16, MQ_I16: torch.int16}


def int8_linear(a_q: torch.Tensor, w_q: torch.Tensor, a_rowsum: Optional[torch.Tensor], alpha: torch.Tensor,
                w_zp: torch.Tensor, col_term: torch.Tensor, bias: Optional[torch.Tensor] = None, *,
                out_scale: Optional[torch.Tensor] = None, out_offset: Optional[torch.Tensor] = None,
                out_qmin: float = 0.0, out_qmax: float = 255.0, out_dtype: int = MQ_F32, w4: bool = False,
                out: Optional[torch.Tensor] = None, a_tiled_rows: Optional[int] = None,
                resid: Optional[torch.Tensor] = None) -> torch.Tensor:
    """QLinear as an int8 MFMA GEMM with fused dequant (+ output quantizer).  a_q [M,K] int8, w_q [N,K]
    int8 (or [N,K/2] packed nibbles when w4).  a_tiled_rows = M: a_q is the fragment-blocked buffer of
    quantize_tiled ([ceil16(M), K] bytes) and the generated-ISA GEMM path runs (gemm_tiled_supported shapes).
    resid [M, N] fp32 (row-major int8 weights and activations, fp32 output, M > 8): out = resid + Qout(linear), the add fused into
    the GEMM's store (residual_supported())."""
    _dev(a_q, "a_q"); _dev(w_q, "w_q")
    M, K = a_q.shape
    if a_tiled_rows is not None:
        M = int(a_tiled_rows)
    N = w_q.shape[0]
    if out is None:
        out = torch.empty((M, N), dtype=_OUT_TORCH[out_dtype], device=a_q.device)
    b = _f32(bias, "bias") if bias is not None else None
    os_ = _f32(out_scale, "out_scale") if out_scale is not None else None
    oo_ = _f32(out_offset, "out_offset") if out_offset is not None else None
    if resid is not None:
        if w4 and a_tiled_rows is None:
            raise RuntimeError("mobilequant_amd: int8_linear(resid=..., w4=True) needs the fragment-blocked activation image (a_tiled_rows)")
        if out_dtype != MQ_F32 or M <= 8:
            raise RuntimeError("mobilequant_amd: int8_linear(resid=...) needs fp32 output and M > 8")
        if a_tiled_rows is not None and not (os_ is not None and out_qmax - out_qmin > 255.0 and gemm_tiled128_supported(M, N, K)):
            raise RuntimeError("mobilequant_amd: int8_linear(resid=..., a_tiled_rows=...) needs a 16-bit output grid and a "
                               "gemm_tiled128_supported shape")
        resid = _f32(_dev(resid, "resid"), "resid")
        if resid.numel() != M * N or resid.data_ptr() == out.data_ptr():
            raise RuntimeError("mobilequant_amd: resid must be [M, N] and must not alias the output")
        with _on(a_q, w_q, a_rowsum, alpha, w_zp, col_term, b, os_, oo_, out, resid):
            _lib.call(("mq_w4a8_linear_tiled_residual" if w4 else "mq_w8a8_linear_tiled_residual") if a_tiled_rows is not None else "mq_w8a8_linear_residual",
                      a_q.data_ptr(), w_q.data_ptr(), M, N, K,
                      a_rowsum.data_ptr() if a_rowsum is not None else None, alpha.data_ptr(), w_zp.data_ptr(), col_term.data_ptr(),
                      b.data_ptr() if b is not None else None, os_.data_ptr() if os_ is not None else None,
                      oo_.data_ptr() if oo_ is not None else None, float(out_qmin), float(out_qmax), resid.data_ptr(), out.data_ptr(),
                      _stream())
        return out
    fn = "mq_w8a8_linear_tiled" if a_tiled_rows is not None else ("mq_w4a8_linear" if w4 else "mq_w8a8_linear")
    with _on(a_q, w_q, a_rowsum, alpha, w_zp, col_term, b, os_, oo_, out):
        _lib.call(fn, a_q.data_ptr(), w_q.data_ptr(), M, N, K,
                  a_rowsum.data_ptr() if a_rowsum is not None else None, alpha.data_ptr(), w_zp.data_ptr(),
                  col_term.data_ptr(), b.data_ptr() if b is not None else None,
                  os_.data_ptr() if os_ is not None else None, oo_.data_ptr() if oo_ is not None else None,
                  float(out_qmin), float(out_qmax), out.data_ptr(), out_dtype, _stream())
    return out


def int8_linear_grouped(a_q: torch.Tensor, w_q: torch.Tensor, group_size: int, a_gsum: torch.Tensor, alpha: torch.Tensor, cw: torch.Tensor,
                        t: torch.Tensor, bias: Optional[torch.Tensor] = None) -> torch.Tensor:
    """QLinear with per-group weight grids as an int8 MFMA GEMM (mq_w8a8_linear_grouped): a_q [M, K] / w_q [N, K] stored int8 values,
    a_gsum [G, M] int32, alpha [G, N] fp32, cw / t [G, N] int32 (include/mobilequant_amd.h); fp32 [M, N] out."""
    _dev(a_q, "a_q"); _dev(w_q, "w_q")
    M, K = a_q.shape
    N = w_q.shape[0]
    out = torch.empty((M, N), dtype=torch.float32, device=a_q.device)
    b = _f32(bias, "bias") if bias is not None else None
    al = _f32(alpha, "alpha")
    with _on(a_q, w_q, a_gsum, al, cw, t, b):
        _lib.call("mq_w8a8_linear_grouped", a_q.data_ptr(), w_q.data_ptr(), M, N, K, int(group_size), a_gsum.data_ptr(), al.data_ptr(),
                  cw.data_ptr(), t.data_ptr(), b.data_ptr() if b is not None else None, out.data_ptr(), _stream())
    return out


def int8_linear_segmented(a_q: torch.Tensor, w_q: torch.Tensor, a_rowsum: torch.Tensor, alpha: torch.Tensor, w_zp: torch.Tensor,
                          col_term: torch.Tensor, bias: Optional[torch.Tensor], seg_ends, grids, w4: bool = False,
                          a_tiled_rows: Optional[int] = None) -> torch.Tensor:
    """1..3 linears reading one row-major int8 activation as ONE GEMM (mq_w8a8_linear_segmented): w_q / alpha / w_zp / col_term /
    bias concatenated along N, seg_ends = cumulative column ends, grids[i] = (scale, offset) of segment i's 8-bit unsigned output
    grid.  Returns the uint8 output indices [M, N].  w4: w_q holds packed nibbles [N, K/2] (pack_w4).  a_tiled_rows = M: a_q is the
    fragment-blocked image (gemm_tiled128_supported shapes, int8 weights)."""
    _dev(a_q, "a_q"); _dev(w_q, "w_q")
    M, K = a_q.shape
    fn = "mq_w4a8_linear_segmented" if w4 else "mq_w8a8_linear_segmented"
    if a_tiled_rows is not None:
        if w4:
            raise RuntimeError("mobilequant_amd: the fragment-blocked segmented GEMM takes int8 weights")
        M, fn = int(a_tiled_rows), "mq_w8a8_linear_tiled_segmented"
    N = w_q.shape[0]
    n = len(seg_ends)
    out = torch.empty((M, N), dtype=torch.uint8, device=a_q.device)
    b = _f32(bias, "bias") if bias is not None else None
    ends = (ctypes.c_int64 * n)(*[int(e) for e in seg_ends])
    keep, gs = [], (_lib.MqGrid * n)()
    for i, g in enumerate(grids):
        sc, of = _f32(g[0], "scale"), _f32(g[1], "offset")
        keep += [sc, of]
        gs[i] = _lib.MqGrid(sc.data_ptr(), of.data_ptr(), 0.0, 255.0)
    with _on(a_q, w_q, a_rowsum, alpha, w_zp, col_term, b, out, *keep):
        _lib.call(fn, a_q.data_ptr(), w_q.data_ptr(), M, N, K, a_rowsum.data_ptr() if a_rowsum is not None else None,
                  alpha.data_ptr(), w_zp.data_ptr(), col_term.data_ptr(), b.data_ptr() if b is not None else None, n, ends, gs,
                  out.data_ptr(), _stream())
    return out


def gemm_tiled_w4_supported(M: int, N: int, K: int) -> bool:
    """Shapes served by the packed-4-bit generated kernels on fragment-blocked activations (mq_w4a8_linear_tiled)."""
    return bool(_lib.load().mq_gemm_tiled_w4_supported(int(M), int(N), int(K)))


def w4a8_linear_tiled(a_tiled: torch.Tensor, M: int, w_packed: torch.Tensor, a_rowsum: Optional[torch.Tensor], alpha: torch.Tensor,
                      w_zp: torch.Tensor, col_term: torch.Tensor, bias: Optional[torch.Tensor], grids, seg_ends=None,
                      out_dtype: int = MQ_U8, out: Optional[torch.Tensor] = None) -> torch.Tensor:
    """Packed 4-bit weights [N, K/2] (pack_w4) x fragment-blocked int8 activations (quantize_tiled, M rows) on the generated-ISA kernels
    (mq_w4a8_linear_tiled): 8-bit unsigned output grid(s) -> uint8 indices (or int8 = index - 128) [M, N].  grids: [(scale, offset)] per
    column segment; seg_ends: cumulative column ends when there is more than one."""
    _dev(a_tiled, "a_tiled"); _dev(w_packed, "w_packed")
    N, K = int(w_packed.shape[0]), 2 * int(w_packed.shape[1])
    n = len(grids)
    if out is None:
        out = torch.empty((int(M), N), dtype=_OUT_TORCH[out_dtype], device=a_tiled.device)
    b = _f32(bias, "bias") if bias is not None else None
    ends = (ctypes.c_int64 * n)(*[int(e) for e in (seg_ends if seg_ends is not None else [N])])
    keep, gs = [], (_lib.MqGrid * n)()
    for i, g in enumerate(grids):
        sc, of = _f32(g[0], "scale"), _f32(g[1], "offset")
        keep += [sc, of]
        gs[i] = _lib.MqGrid(sc.data_ptr(), of.data_ptr(), 0.0, 255.0)
    with _on(a_tiled, w_packed, a_rowsum, alpha, w_zp, col_term, b, out, *keep):
        _lib.call("mq_w4a8_linear_tiled", a_tiled.data_ptr(), w_packed.data_ptr(), int(M), N, K,
                  a_rowsum.data_ptr() if a_rowsum is not None else None, alpha.data_ptr(), w_zp.data_ptr(), col_term.data_ptr(),
                  b.data_ptr() if b is not None else None, n, ends, gs, out.data_ptr(), out_dtype, _stream())
    return out


def gemm_tiled128_supported(M: int, N: int, K: int) -> bool:
    """Shapes served by the 128-column generated kernels on fragment-blocked activations (o_proj / w2 with the residual, q | k | v)."""
    return bool(_lib.load().mq_gemm_tiled128_supported(int(M), int(N), int(K)))


def gemm_tiled_supported(M: int, N: int, K: int) -> bool:
    """Shapes served by the fragment-blocked activation layout + generated-ISA GEMM loop (TinyLlama / StableLM FFN)."""
    return bool(_lib.load().mq_gemm_tiled_supported(int(M), int(N), int(K)))


def quantize_tiled(x2d: torch.Tensor, scale: torch.Tensor, offset: torch.Tensor, qmin: float, qmax: float, shift: int,
                   want_row_sum: bool = True, chan_scale: Optional[torch.Tensor] = None):
    """[M,K] activations -> fragment-blocked int8 ([ceil16(M), K] buffer, layout: include/mobilequant_amd.h) + row sums."""
    x2d = _dev(x2d, "x").contiguous()
    M, K = x2d.shape
    q = torch.empty(((M + 15) // 16 * 16, K), dtype=torch.int8, device=x2d.device)
    rs = torch.empty(M, dtype=torch.int32, device=x2d.device) if want_row_sum else None
    s, o = _f32(scale, "scale"), _f32(offset, "offset")
    cs = _chan(chan_scale, K, x2d)
    with _on(x2d, s, o, cs):
        _lib.call("mq_quantize_tiled", x2d.data_ptr(), _fdt(x2d), M, K, s.data_ptr(), o.data_ptr(), float(qmin), float(qmax),
                  int(shift), cs.data_ptr() if cs is not None else None, q.data_ptr(), rs.data_ptr() if rs is not None else None,
                  _stream())
    return (q, rs) if want_row_sum else q


def decode_shape(M: int, K: int) -> bool:
    """Shapes mq_w8a8_linear_f32in accepts (activation quantize fused into the weight-streaming GEMV)."""
    return M <= 8 and M * K <= 64 * 1024 - 64 and K % 256 == 0


def int8_linear_f32in(x2d: torch.Tensor, a_scale, a_offset, a_qmin: float, a_qmax: float, a_shift: int, w_q: torch.Tensor,
                      alpha, w_zp, col_term, bias=None, *, out_scale=None, out_offset=None, out_qmin: float = 0.0,
                      out_qmax: float = 255.0, out_dtype: int = MQ_F32, out: Optional[torch.Tensor] = None, w4: bool = False):
    """Decode-shape QLinear in ONE kernel: fp32 activations are quantised inside the GEMV (M <= 8).
    w4: w_q holds packed nibbles ([N, K/2], pack_w4)."""
    x2d = _f32(_dev(x2d, "x"), "x")
    M, K = x2d.shape
    N = w_q.shape[0]
    if out is None:
        out = torch.empty((M, N), dtype=_OUT_TORCH[out_dtype], device=x2d.device)
    sa, oa = _f32(a_scale, "a_scale"), _f32(a_offset, "a_offset")
    b = _f32(bias, "bias") if bias is not None else None
    os_ = _f32(out_scale, "out_scale") if out_scale is not None else None
    oo_ = _f32(out_offset, "out_offset") if out_offset is not None else None
    with _on(x2d, sa, oa, w_q, alpha, w_zp, col_term, b, os_, oo_, out):
        _lib.call("mq_w4a8_linear_f32in" if w4 else "mq_w8a8_linear_f32in", x2d.data_ptr(), sa.data_ptr(), oa.data_ptr(), float(a_qmin), float(a_qmax), int(a_shift),
                  w_q.data_ptr(), M, N, K, alpha.data_ptr(), w_zp.data_ptr(), col_term.data_ptr(),
                  b.data_ptr() if b is not None else None, os_.data_ptr() if os_ is not None else None,
                  oo_.data_ptr() if oo_ is not None else None, float(out_qmin), float(out_qmax), out.data_ptr(), out_dtype,
                  _stream())
    return out


def rmsnorm_quant(x: torch.Tensor, weight: torch.Tensor, bias, eps: float, in_grid=None, out_grid=None, emit_int8: bool = False,
                  layernorm: bool = False, emit_tiled: bool = False, want_y: bool = True, emit_rowmajor: bool = True):
    """QRMSNorm.forward (layernorm=True: QLayerNorm.forward) in one launch.  in_grid / out_grid: None or
    (scale, offset, qmin, qmax) per-tensor.  Returns y, or (y, q_int8, row_sum, shift, q_tiled) with emit_int8
    (8-bit output grids only; q_tiled = the fragment-blocked copy when emit_tiled, else None).  want_y=False (emit_int8 only): the
    fp32 result is not written (y = None); emit_rowmajor=False: only the fragment-blocked image (q_int8 = None)."""
    x = _f32(_dev(x, "x"), "x").contiguous()
    cols = x.shape[-1]
    rows = x.numel() // cols
    w = _f32(weight, "weight").contiguous()
    b = _f32(bias, "bias").contiguous() if bias is not None else None
    if not want_y and not emit_int8:
        raise RuntimeError("mobilequant_amd: rmsnorm_quant(want_y=False) needs emit_int8")
    y = torch.empty_like(x) if want_y else None
    si = oi = so = oo = None
    iqmin = iqmax = oqmin = oqmax = 0.0
    if in_grid is not None:
        si, oi, iqmin, iqmax = _f32(in_grid[0], "in_scale"), _f32(in_grid[1], "in_offset"), float(in_grid[2]), float(in_grid[3])
    if out_grid is not None:
        so, oo, oqmin, oqmax = _f32(out_grid[0], "out_scale"), _f32(out_grid[1], "out_offset"), float(out_grid[2]), float(out_grid[3])
    q = rs = qt = None
    shift = 0
    if emit_int8:
        shift = 128 if oqmax > 127 else 0
        if emit_rowmajor or not emit_tiled:
            q = torch.empty((rows, cols), dtype=torch.int8, device=x.device)
        rs = torch.empty(rows, dtype=torch.int32, device=x.device)
        if emit_tiled:
            qt = torch.empty(((rows + 15) // 16 * 16, cols), dtype=torch.int8, device=x.device)
    with _on(x, w, b, si, oi, so, oo):
        _lib.call("mq_layernorm_quant" if layernorm else "mq_rmsnorm_quant", x.data_ptr(), rows, cols, w.data_ptr(),
                  b.data_ptr() if b is not None else None, float(eps),
                  si.data_ptr() if si is not None else None, oi.data_ptr() if oi is not None else None, iqmin, iqmax,
                  so.data_ptr() if so is not None else None, oo.data_ptr() if oo is not None else None, oqmin, oqmax,
                  y.data_ptr() if y is not None else None, q.data_ptr() if q is not None else None,
                  qt.data_ptr() if qt is not None else None, shift,
                  rs.data_ptr() if rs is not None else None, _stream())
    return (y, q, rs, shift, qt) if emit_int8 else y


def act_quant(x: torch.Tensor, act: str, in_grid=None, mid_grid=None, out_grid=None):
    """QSiLU ("silu") / QGELU ("gelu") forward in one launch; grids: None or (scale, offset, qmin, qmax) per-tensor."""
    x = _f32(_dev(x, "x"), "x").contiguous()
    y = torch.empty_like(x)
    ptrs = []
    for g in (in_grid, mid_grid, out_grid):
        if g is None:
            ptrs += [None, None, 0.0, 0.0]
        else:
            ptrs += [_f32(g[0], "scale").data_ptr(), _f32(g[1], "offset").data_ptr(), float(g[2]), float(g[3])]
    with _on(x):
        _lib.call("mq_act_quant", x.data_ptr(), x.numel(), {"silu": 0, "gelu": 1}[act], *ptrs, y.data_ptr(), _stream())
    return y


def pack_w4(nibbles: torch.Tensor) -> torch.Tensor:
    """[N,K] uint8 nibbles (0..15) -> [N,K/2] packed (layout: include/mobilequant_amd.h, mq_pack_w4)."""
    nibbles = _dev(nibbles, "nibbles").contiguous()
    N, K = nibbles.shape
    packed = torch.empty((N, K // 2), dtype=torch.uint8, device=nibbles.device)
    with _on(nibbles):
        _lib.call("mq_pack_w4", nibbles.data_ptr(), N, K, packed.data_ptr(), _stream())
    return packed


# ---- f1: integer chaining inside the gated FFN ---------------------------------------------------------------------------------
def int8_linear_pair(a_tiled: torch.Tensor, rows: int, a_rowsum: Optional[torch.Tensor], first: dict, second: dict,
                     out_dtype: int = MQ_U8):
    """Two QLinears over the SAME fragment-blocked activation in one launch (w1 / w3 of a gated FFN).  `first` / `second`:
    dicts with w [N,K] int8, alpha, w_zp, col_term, bias (or None), out_scale, out_offset (8-bit unsigned output grids).
    Returns the two [rows, N] index tensors (u8, or i8 = index - 128)."""
    M, K = int(rows), a_tiled.shape[1]
    N = first["w"].shape[0]
    outs = [torch.empty((M, N), dtype=_OUT_TORCH[out_dtype], device=a_tiled.device) for _ in range(2)]
    ptrs, keep = [], []
    for p, o in zip((first, second), outs):
        b = _f32(p["bias"], "bias") if p.get("bias") is not None else None
        os_, oo_ = _f32(p["out_scale"], "out_scale"), _f32(p["out_offset"], "out_offset")
        keep += [b, os_, oo_]
        ptrs += [p["w"].data_ptr(), p["alpha"].data_ptr(), p["w_zp"].data_ptr(), p["col_term"].data_ptr(),
                 b.data_ptr() if b is not None else None, os_.data_ptr(), oo_.data_ptr(), o.data_ptr()]
    with _on(a_tiled, a_rowsum, first["w"], second["w"], first["alpha"], second["alpha"], *[k for k in keep if k is not None]):
        _lib.call("mq_w8a8_linear_tiled_pair", a_tiled.data_ptr(), M, N, K, a_rowsum.data_ptr() if a_rowsum is not None else None,
                  *ptrs, out_dtype, _stream())
    return outs[0], outs[1]


def int8_linear_gated(a_tiled: torch.Tensor, rows: int, a_rowsum: Optional[torch.Tensor], first: dict, second: dict, table: torch.Tensor,
                      w4: bool = False):
    """w1, w3 and the gate of a gated FFN in TWO launches (mq_w8a8_linear_tiled_gated): `first` (w1) writes its 8-bit indices, `second`
    (w3) looks (w1 index, w3 index) up in `table` (gated_table) inside its epilogue.  Returns (w2's int8 input image, fragment-blocked
    [ceil16(rows), N]; its row sums [rows]) -- the bits of int8_linear_pair + gated_lookup(tiled=True)."""
    # w4: both weights are mq_pack_w4 images ([N, K / 2] bytes) and the packed kernels run (mq_w4a8_linear_tiled_gated: identical bytes)
    M, K = int(rows), a_tiled.shape[1]
    N = first["w"].shape[0]
    dev = a_tiled.device
    if table.dtype != torch.int8 or table.numel() != 65536:
        raise RuntimeError("mobilequant_amd: int8_linear_gated needs an int8 [65536] table (gated_table)")
    idx = torch.empty((M, N), dtype=torch.uint8, device=dev)
    q = torch.empty(((M + 15) // 16 * 16, N), dtype=torch.int8, device=dev)
    rs = torch.empty(M, dtype=torch.int32, device=dev)
    ptrs, keep = [], []
    for p in (first, second):
        b = _f32(p["bias"], "bias") if p.get("bias") is not None else None
        os_, oo_ = _f32(p["out_scale"], "out_scale"), _f32(p["out_offset"], "out_offset")
        keep += [b, os_, oo_]
        ptrs += [p["w"].data_ptr(), p["alpha"].data_ptr(), p["w_zp"].data_ptr(), p["col_term"].data_ptr(),
                 b.data_ptr() if b is not None else None, os_.data_ptr(), oo_.data_ptr()]
    with _on(a_tiled, a_rowsum, first["w"], second["w"], first["alpha"], second["alpha"], table, *[k for k in keep if k is not None]):
        _lib.call("mq_w4a8_linear_tiled_gated" if w4 else "mq_w8a8_linear_tiled_gated", a_tiled.data_ptr(), M, N, K,
                  a_rowsum.data_ptr() if a_rowsum is not None else None,
                  *ptrs, table.data_ptr(), idx.data_ptr(), q.data_ptr(), rs.data_ptr(), _stream())
    return q, rs


def gated_act_quant(a: torch.Tensor, b: torch.Tensor, act: str, out_grid, *, a_grid=None, b_grid=None, mid_grid=None, act_grid=None,
                    q_shift: int = 128, want_y: bool = False):
    """act(a) * b -> int8 image on `out_grid` (+ row sums, + the fp32 product when want_y) in one launch: the gated FFN between
    the w1 / w3 GEMMs and w2.  a / b: fp32 values [rows, cols], or uint8 output indices of the GEMMs with their grids
    (a_grid / b_grid).  Grids: (scale, offset, qmin, qmax) per-tensor or None."""
    a, b = _dev(a, "a").contiguous(), _dev(b, "b").contiguous()
    if a.dtype != b.dtype or a.shape != b.shape or a.dtype not in (torch.float32, torch.uint8):
        raise RuntimeError("mobilequant_amd: gated_act_quant needs two float32 or two uint8 tensors of one shape")
    cols = a.shape[-1]
    rows = a.numel() // max(cols, 1)
    q = torch.empty(a.shape, dtype=torch.int8, device=a.device)
    rs = torch.empty(rows, dtype=torch.int32, device=a.device)
    y = torch.empty(a.shape, dtype=torch.float32, device=a.device) if want_y else None
    ptrs, keep = [], []
    for g, with_limits in ((a_grid, False), (b_grid, False), (mid_grid, True), (act_grid, True), (out_grid, True)):
        if g is None:
            ptrs += [None, None] + ([0.0, 0.0] if with_limits else [])
        else:
            s, o = _f32(g[0], "scale"), _f32(g[1], "offset")
            keep += [s, o]
            ptrs += [s.data_ptr(), o.data_ptr()] + ([float(g[2]), float(g[3])] if with_limits else [])
    with _on(a, b, *keep):
        _lib.call("mq_gated_act_quant", a.data_ptr(), b.data_ptr(), MQ_U8 if a.dtype == torch.uint8 else MQ_F32, rows, cols,
                  {"silu": 0, "gelu": 1}[act], *ptrs, int(q_shift), q.data_ptr(), rs.data_ptr(),
                  y.data_ptr() if y is not None else None, _stream())
    return (q, rs, y) if want_y else (q, rs)


def gated_table(act: str, out_grid, a_grid, b_grid, mid_grid=None, act_grid=None, q_shift: int = 128) -> torch.Tensor:
    """The 256 x 256 table of gated_act_quant on index inputs (mq_gated_table): int8 [65536], entry [ia * 256 + ib]."""
    dev = out_grid[0].device
    table = torch.empty(65536, dtype=torch.int8, device=dev)
    ptrs, keep = [], []
    for g, with_limits in ((a_grid, False), (b_grid, False), (mid_grid, True), (act_grid, True), (out_grid, True)):
        if g is None:
            ptrs += [None, None] + ([0.0, 0.0] if with_limits else [])
        else:
            s, o = _f32(g[0], "scale"), _f32(g[1], "offset")
            keep += [s, o]
            ptrs += [s.data_ptr(), o.data_ptr()] + ([float(g[2]), float(g[3])] if with_limits else [])
    with _on(table, *keep):
        _lib.call("mq_gated_table", {"silu": 0, "gelu": 1}[act], *ptrs, int(q_shift), table.data_ptr(), _stream())
    return table


def gated_lookup(a: torch.Tensor, b: torch.Tensor, table: torch.Tensor, tiled: bool = False):
    """uint8 index tensors a, b [rows, cols] -> (int8 image, row sums) through a gated_table (mq_gated_lookup).  tiled: the image
    in the fragment-blocked layout of quantize_tiled ([ceil16(rows), cols], cols % 64 == 0)."""
    a, b = _dev(a, "a").contiguous(), _dev(b, "b").contiguous()
    if a.dtype != torch.uint8 or b.dtype != torch.uint8 or a.shape != b.shape or table.dtype != torch.int8 or table.numel() != 65536:
        raise RuntimeError("mobilequant_amd: gated_lookup needs two uint8 tensors of one shape and an int8 [65536] table")
    cols = a.shape[-1]
    rows = a.numel() // max(cols, 1)
    q = torch.empty(((rows + 15) // 16 * 16, cols) if tiled else a.shape, dtype=torch.int8, device=a.device)
    rs = torch.empty(rows, dtype=torch.int32, device=a.device)
    with _on(a, b, table):
        _lib.call("mq_gated_lookup_tiled" if tiled else "mq_gated_lookup", a.data_ptr(), b.data_ptr(), rows, cols, table.data_ptr(), q.data_ptr(), rs.data_ptr(), _stream())
    return q, rs


def qmatmul(x1: torch.Tensor, x2: torch.Tensor, grid1, grid2, grid_out=None) -> torch.Tensor:
    """QMatMul.forward (qmodule.py:453-466) in one launch (include/mobilequant_amd.h: mq_qmatmul): x1 [..., M, K] @ x2 [..., K, N] with
    both input quantizers and the output quantizer fused around an exact int8 contraction.  Grids are (scale, offset, qmin, qmax) with
    1-element device tensors (static per-tensor), grid_out may be None.  x2 is read in place when it is a dense [..., K, N] tensor with
    N % 4 == 0 or a transposed view of a dense [..., N, K] one (hf_model.py:513 passes k.transpose(2, 3)); anything else is copied
    K-contiguous first.
    Leading dims must match (no broadcasting).  Returns fp32 [..., M, N]."""
    _dev(x1, "x1"), _dev(x2, "x2")
    if x1.dtype != torch.float32 or x2.dtype != torch.float32:
        raise RuntimeError("mobilequant_amd: qmatmul takes float32 operands")
    if x1.dim() < 2 or x2.dim() != x1.dim() or x1.shape[:-2] != x2.shape[:-2] or x1.shape[-1] != x2.shape[-2]:
        raise RuntimeError(f"mobilequant_amd: qmatmul shapes {tuple(x1.shape)} @ {tuple(x2.shape)} (equal leading dims, no broadcasting)")
    lead = tuple(x1.shape[:-2])
    M, K, N = x1.shape[-2], x1.shape[-1], x2.shape[-1]
    batch = 1
    for d in lead:
        batch *= d
    a = x1 if x1.is_contiguous() else x1.contiguous()
    if x2.is_contiguous() and N % 4 == 0 and x2.data_ptr() % 16 == 0:
        b, kc = x2, 0
    else:                       # a transposed view of a dense [..., N, K] tensor is read in place; anything else is copied into that order
        t = x2.transpose(-1, -2)
        b, kc = (t if t.is_contiguous() else t.contiguous()), 1
    out = torch.empty(lead + (M, N), dtype=torch.float32, device=x1.device)
    keep = [a, b]

    def struct(g):
        if g is None:
            return None
        sc, of = _f32(g[0], "grid scale").reshape(-1), _f32(g[1], "grid offset").reshape(-1)
        if sc.numel() != 1 or of.numel() != 1:
            raise RuntimeError("mobilequant_amd: qmatmul takes per-tensor grids")
        keep.extend((sc, of))
        return _lib.MqGrid(sc.data_ptr(), of.data_ptr(), float(g[2]), float(g[3]))
    g1, g2, go = struct(grid1), struct(grid2), struct(grid_out)
    with _on(a, b, *keep):
        _lib.call("mq_qmatmul", a.data_ptr(), b.data_ptr(), out.data_ptr(), batch, M, N, K, kc, ctypes.byref(g1), ctypes.byref(g2),
                  ctypes.byref(go) if go is not None else None, _stream())
    return out


def qmatmul_supported(x1: torch.Tensor, x2: torch.Tensor) -> bool:
    """Shapes mq_qmatmul serves (the grids are the caller's to check): fp32 device tensors with equal leading dims (no broadcasting);
    any M / N / K."""
    if not (x1.is_cuda and x2.is_cuda and x1.dtype == torch.float32 and x2.dtype == torch.float32):
        return False
    if x1.dim() < 2 or x2.dim() != x1.dim() or x1.shape[:-2] != x2.shape[:-2] or x1.shape[-1] != x2.shape[-2] or x1.numel() == 0 or x2.numel() == 0:
        return False
    return x1.shape[-1] <= 131071            # the int32 MFMA accumulators hold K x 128 x 128


def attention_image_cache(kv_heads: int, head_dim: int, max_len: int, device) -> dict:
    """Caller-owned K / vT image caches for attention_quant(cache=..., pos0=...): one per attention block and sequence."""
    rows = (int(max_len) + 63) // 64 * 64
    dev = torch.device(device)
    return {"rows": rows, "kv_heads": int(kv_heads), "head_dim": int(head_dim),
            "k_i8": torch.zeros(kv_heads * rows * head_dim, dtype=torch.int8, device=dev),
            "vt_i8": torch.zeros(kv_heads * rows * head_dim, dtype=torch.int8, device=dev),
            "k_rs": torch.zeros(kv_heads * rows, dtype=torch.int32, device=dev),
            "k_f16": torch.zeros(kv_heads * rows * head_dim, dtype=torch.float16, device=dev) if head_dim == 64 else None,
            "v_pre": torch.zeros(kv_heads * (rows // 64) * head_dim, dtype=torch.int32, device=dev) if head_dim != 64 else None}


def attention_quant(q: Optional[torch.Tensor], k: Optional[torch.Tensor], v: Optional[torch.Tensor], cos: torch.Tensor, sin: torch.Tensor,
                    heads: int, kv_heads: int, grids: dict, image=None, want_out: bool = True, qkv_idx=None, head_dim: int = 64,
                    cache=None, pos0: int = 0):
    """Quantized causal prefill attention of ONE sequence (mq_attention_quant; head_dim 64 or 256 -- "64" below reads head_dim): q [S, heads*64], k / v [S, kv_heads*64] fp32
    projection outputs before RoPE, cos / sin [S, 64]; grids: qk_a, qk_b, qk_out, pv_a, pv_b, pv_out -> (scale, offset, qmin, qmax)
    per tensor or None (qk_out / pv_out only).  Returns pv_bmm's output [S, heads*64] fp32 (o_proj's input layout).
    image = (q_i8, row_sum [rows] int32, row0, shift, tiled): additionally (want_out=False: only) write the pv_out indices of this
    sequence as rows row0 .. row0+S-1 of the consumer linear's int8 input image: row-major [rows, heads*64], or (tiled) the
    fragment-blocked [ceil16(rows), heads*64] layout of quantize_tiled.
    Batch: q / k / v [B, S, ...] (or qkv_idx [B, S, ...]) run as ONE launch pair (mq_attention_args.batch; not with a cache); the
    result is [B, S, heads*64] and sequence b owns rows row0 + b * S of the image.
    cache = attention_image_cache(...) + pos0: cache continuation (chunked prefill).  The cache already holds positions 0 .. pos0 - 1
    from earlier calls (same grids); q / k / v / cos / sin describe positions pos0 .. pos0 + S - 1, which are appended and attend to
    everything before them.  pos0 % 64 == 0 (every chunk but the last is a multiple of 64 long)."""
    cos, sin = _f32(cos, "cos"), _f32(sin, "sin")
    D = int(head_dim)
    if D not in (64, 128, 256):
        raise RuntimeError("mobilequant_amd: attention_quant serves head_dim 64, 128 and 256")
    rot = cos.shape[-1]                  # partial rotary: cos / sin [S, rot_dim] with rot_dim < 64 (hf_model.py:489-500)
    if rot > D or rot % 2 or sin.shape != cos.shape:
        raise RuntimeError("mobilequant_amd: attention_quant cos / sin must be [S, rot_dim], rot_dim even and <= 64")
    idx = None
    if qkv_idx is not None:       # (uint8 [S, (heads + 2 kv_heads) * 64] of int8_linear_segmented, ((scale, offset) x 3))
        idx, in_grids = qkv_idx
        idx = _dev(idx, "qkv_idx").contiguous()
        B = idx.shape[0] if idx.dim() == 3 else 0                    # 0: one sequence, 2-D tensors
        S = idx.shape[-2]
        if idx.dtype != torch.uint8 or idx.shape[-2:] != (S, (heads + 2 * kv_heads) * D) or idx.dim() not in (2, 3) or cos.shape != (S, rot):
            raise RuntimeError("mobilequant_amd: attention_quant qkv_idx must be uint8 [(B,) S, (H + 2 KV) * 64], cos / sin [S, 64]")
        q = k = v = None
    else:
        q, k, v = (_dev(t, n).contiguous() for t, n in ((q, "q"), (k, "k"), (v, "v")))
        B = q.shape[0] if q.dim() == 3 else 0
        S = q.shape[-2]
        lead = (B,) if B else ()
        if (q.dtype != torch.float32 or k.dtype != torch.float32 or v.dtype != torch.float32 or q.shape != lead + (S, heads * D)
                or k.shape != lead + (S, kv_heads * D) or v.shape != k.shape or cos.shape != (S, rot)):
            raise RuntimeError("mobilequant_amd: attention_quant needs fp32 q [(B,) S, H*64], k / v [(B,) S, KV*64], cos / sin [S, 64]")
    if B and cache is not None:
        raise RuntimeError("mobilequant_amd: attention_quant serves a batch in one launch only without an image cache (one call per sequence there)")
    nb = max(B, 1)
    S_real = S
    if S % 64:                    # pad the sequence: under the causal mask a padded key is only ever seen by padded queries
        pad = 64 - S % 64
        cos, sin = (torch.nn.functional.pad(t, (0, 0, 0, pad)) for t in (cos, sin))
        if idx is not None:
            idx = torch.nn.functional.pad(idx, (0, 0, 0, pad))
        else:
            q, k, v = (torch.nn.functional.pad(t, (0, 0, 0, pad)) for t in (q, k, v))
        S += pad
    a = _lib.MqAttentionArgs()
    keep = []
    for name in ("qk_a", "qk_b", "qk_out", "pv_a", "pv_b", "pv_out"):
        g = grids.get(name)
        if g is None:
            setattr(a, name, _lib.MqGrid(None, None, 0.0, 0.0))
        else:
            s, o = _f32(g[0], "scale"), _f32(g[1], "offset")
            keep += [s, o]
            setattr(a, name, _lib.MqGrid(s.data_ptr(), o.data_ptr(), float(g[2]), float(g[3])))
    first = idx if idx is not None else q
    dev = first.device
    if idx is not None:
        a.qkv_idx = idx.data_ptr()
        for name, g in zip(("q_in", "k_in", "v_in"), in_grids):
            sc, of = _f32(g[0], "scale"), _f32(g[1], "offset")
            keep += [sc, of]
            setattr(a, name, _lib.MqGrid(sc.data_ptr(), of.data_ptr(), 0.0, 255.0))
    out = torch.empty(((B,) if B else ()) + (S, heads * D), dtype=torch.float32, device=dev) if want_out or image is None else None
    q_i8 = torch.empty(nb * heads * S * D, dtype=torch.int8, device=dev)
    q_rs = torch.empty(nb * heads * S, dtype=torch.int32, device=dev)
    a.batch = B
    if cache is not None:
        if (cache["kv_heads"], cache["head_dim"]) != (kv_heads, D) or cache["k_i8"].device != dev:
            raise RuntimeError("mobilequant_amd: attention_quant cache was built for another shape / device")
        if pos0 % 64 or pos0 < 0 or pos0 + S > cache["rows"]:
            raise RuntimeError(f"mobilequant_amd: attention_quant cache continuation needs pos0 % 64 == 0 and pos0 + padded S <= {cache['rows']} "
                               f"(pos0={pos0}, S={S})")
        k_i8, vt_i8, k_rs, v_pre = cache["k_i8"], cache["vt_i8"], cache["k_rs"], cache["v_pre"]
        k_f16 = cache.get("k_f16")
        a.pos0, a.cache_seq = int(pos0), int(cache["rows"])
    else:
        if pos0:
            raise RuntimeError("mobilequant_amd: attention_quant(pos0 > 0) needs a cache (attention_image_cache)")
        k_i8 = torch.empty(nb * kv_heads * S * D, dtype=torch.int8, device=dev)
        vt_i8 = torch.empty(nb * kv_heads * S * D, dtype=torch.int8, device=dev)
        k_rs = torch.empty(nb * kv_heads * S, dtype=torch.int32, device=dev)
        v_pre = torch.empty(nb * kv_heads * (S // 64) * D, dtype=torch.int32, device=dev) if D != 64 else None
        k_f16 = torch.empty(nb * kv_heads * S * D, dtype=torch.float16, device=dev) if D == 64 else None
    if k_f16 is not None and grids.get("qk_out") is not None:
        # head_dim 64 with a score grid: fp16 images of the centred q / k indices -> the f16 score contraction (mq_attention_args.q_f16)
        q_f16 = torch.empty(nb * heads * S * D, dtype=torch.float16, device=dev)
        keep += [q_f16, k_f16]
        a.q_f16, a.k_f16 = q_f16.data_ptr(), k_f16.data_ptr()
    if v_pre is not None:
        keep.append(v_pre)
        a.v_prefix = v_pre.data_ptr()
    if idx is None:
        a.q, a.k, a.v = q.data_ptr(), k.data_ptr(), v.data_ptr()
    a.cos, a.sin = cos.data_ptr(), sin.data_ptr()
    a.seq, a.heads, a.kv_heads, a.head_dim, a.inv_sqrt_d = S, heads, kv_heads, D, 1.0 / (D ** 0.5)
    a.rot_dim = rot
    a.out, a.q_i8, a.k_i8, a.vt_i8 = out.data_ptr() if out is not None else None, q_i8.data_ptr(), k_i8.data_ptr(), vt_i8.data_ptr()
    a.q_rowsum, a.k_rowsum = q_rs.data_ptr(), k_rs.data_ptr()
    a.seq_real = S_real
    if image is not None:
        q_t, rs_t, row0, shift, tiled = image
        need = (row0 + nb * S_real + 15) // 16 * 16 if tiled else row0 + nb * S_real
        if (q_t.dtype != torch.int8 or rs_t.dtype != torch.int32 or not q_t.is_contiguous() or q_t.shape[-1] != heads * D
                or row0 < 0 or row0 + nb * S_real > rs_t.numel() or need > q_t.shape[0]):
            raise RuntimeError("mobilequant_amd: attention_quant image must be int8 [rows (tiled: ceil16), heads*64] + int32 row sums [rows]")
        keep += [q_t, rs_t]
        a.out_i8, a.out_rowsum, a.out_row0, a.out_shift, a.out_i8_tiled = q_t.data_ptr(), rs_t.data_ptr(), int(row0), int(shift), int(bool(tiled))
    with _on(first, k, v, cos, sin, *keep):
        _lib.call("mq_attention_quant", ctypes.byref(a), _stream())
    if out is None:
        return None
    return out if S_real == S else out[..., :S_real, :].contiguous()

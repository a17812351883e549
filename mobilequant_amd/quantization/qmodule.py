"""MI355X-native mirror of MobileQuant's ``mobilellm.quantization.qmodule`` API.

Same public names, constructor signatures, attribute names, ``state_dict`` keys and JSON formats as
the reference, so ``ptq/mobilequant.py`` / ``eval/harness_eval.py`` style
callers work against this module unchanged -- but every tensor-sized computation is a hand-written
HIP kernel reached through the C ABI (``mobilequant_amd.ops``):

  * ``Quantizer.forward``  -> fused fake-quant kernel (reference: ~8 torch elementwise ops,
    qmodule.py:286-295); first-forward / dynamic ranges -> single-pass min/max kernels + a
    scale/offset kernel, no host sync (reference: qmodule.py:262-277).
  * ``QLinear.forward``    -> real int8 path when the configuration allows it: quantize-to-int8 (+row
    sums) -> ``v_mfma_i32_16x16x64_i8`` GEMM -> zero-point-corrected dequant + fused output quantizer
    (reference: fp32 simulation, qmodule.py:341-358).  Otherwise the simulated path with HIP fake-quant
    kernels around the library fp GEMM.

Tensors must live on a ROCm device; CPU tensors raise (no fallback).  Host-side scalar set-up
(ranges read from ``act_dict.json`` -> scale/offset) is plain fp32 arithmetic with the reference's
operation order.
"""
from __future__ import annotations

import math
import threading
import weakref
from copy import deepcopy
from dataclasses import dataclass
from typing import Optional

import torch
import torch.nn as nn
import torch.nn.functional as F

from .. import ops
from .._lib import MQ_F16, MQ_F32, MQ_I8, MQ_U8
from .fp_ops import FMatMul, HFRMSNorm

CLIPMIN = 1e-5
CLIPMAX = 1e6


# ------------------------------------------------------------------------------------------------
# scalar / small-tensor range math (reference: qmodule.py:17-76)
# ------------------------------------------------------------------------------------------------
def round_ste(x: torch.Tensor) -> torch.Tensor:
    """Round half to even with an identity gradient (reference: qmodule.py:17-21)."""
    return x + (torch.round(x) - x).detach()


def _grid_limits(bitwidth: int, is_symmetric: bool):
    if is_symmetric:
        half = 2 ** (bitwidth - 1)
        return -half, half - 1
    return 0, 2 ** bitwidth - 1


def _as_f32(v, device=None) -> torch.Tensor:
    if torch.is_tensor(v):
        return v
    return torch.tensor(v, dtype=torch.float32, device=device)


def _needs_grad(*tensors) -> bool:
    return torch.is_grad_enabled() and any(t is not None and t.requires_grad for t in tensors)


def _ver(t: torch.Tensor) -> int:
    """Version counter for cache keys.  Inference tensors (created under torch.inference_mode()) have none -- reading
    ``_version`` raises -- and cannot be modified outside inference mode, so they key as version 0."""
    return 0 if t.is_inference() else t._version


def _tag_grid(y: torch.Tensor, q: "Quantizer") -> torch.Tensor:
    """Remember on the tensor OBJECT which quantizer's grid its values sit on.  A QLinear without an input quantizer
    (q/k/v/o/w1/w3, qmodule.py:848-850) reads the tag to find its producer's LIVE grid -- trained scales, a changed
    bitwidth or a 16-bit producer are seen as they are, never assumed from a calibration file."""
    try:
        y._mq_grid = weakref.ref(q)
    except (AttributeError, TypeError):
        pass
    return y


def _producer_grid(x: torch.Tensor):
    ref = getattr(x, "_mq_grid", None)
    return ref() if ref is not None else None


def compute_min_max_from_tensor(x: torch.Tensor, is_per_channel: bool = False, group_size: int = -1):
    """amin/amax per tensor, per last-dim row (keepdim) or per group (reference: qmodule.py:26-34).
    Runs the single-pass HIP min/max kernels; results stay on the device."""
    if is_per_channel:
        if group_size != -1:
            x = x.reshape(-1, group_size)
        lead = x.shape[:-1]
        mn, mx = ops.minmax_rows(x.reshape(-1, x.shape[-1]))
        return mn.reshape(*lead, 1), mx.reshape(*lead, 1)
    mn, mx = ops.minmax_tensor(x)
    return mn.reshape(()), mx.reshape(())


def compute_scale_offset_from_min_max(min_val, max_val, bitwidth: int, is_symmetric: bool):
    """(scale, offset, alpha, beta, q_min, q_max) exactly as the reference returns them
    (qmodule.py:40-61).  Device tensors use the HIP kernel; host scalars / CPU tensors are tiny
    configuration values and use the same fp32 expression on the host.  Ranges that carry a gradient
    (learnable weight clipping: ``sigmoid(bound_factor) * max``, qmodule.py:271-273) are [N,1]-sized and keep
    the differentiable torch expression so the factors train."""
    q_min, q_max = _grid_limits(bitwidth, is_symmetric)
    mn, mx = _as_f32(min_val), _as_f32(max_val)
    if mn.is_cuda and not _needs_grad(mn, mx):
        scale, offset = ops.scale_offset_from_minmax(mn, mx.to(mn.device), bitwidth, is_symmetric)
        alpha = torch.maximum(mn.abs(), mx.abs()) if is_symmetric else mx - mn
        beta = 0 if is_symmetric else mn
        return scale, offset, alpha, beta, q_min, q_max
    if is_symmetric:
        alpha, beta = torch.maximum(mn.abs(), mx.abs()), 0
    else:
        alpha, beta = mx - mn, mn
    scale = (alpha / q_max).clamp(min=CLIPMIN, max=CLIPMAX)
    offset = -(beta / scale).round()
    return scale, offset, alpha, beta, q_min, q_max


def compute_min_max_from_scale_offset(scale, offset, bitwidth: int, is_symmetric: bool):
    """Inverse of the above (reference: qmodule.py:66-76); used by export_act_range."""
    _, q_max = _grid_limits(bitwidth, is_symmetric)
    s = scale.clamp(min=CLIPMIN, max=CLIPMAX)
    hi = s * q_max + (-offset) * s
    lo = -hi if is_symmetric else (-offset) * s
    return lo, hi


# ------------------------------------------------------------------------------------------------
@dataclass
class QuantConfig:
    """Five-field quantizer description, serialised with string values (reference: qmodule.py:82-107)."""
    bitwidth: int = 32
    group_size: int = -1
    is_symmetric: bool = False
    is_per_channel: bool = False
    is_dynamic: bool = False

    @classmethod
    def from_dict(cls, cfg: dict) -> "QuantConfig":
        flag = lambda k: cfg[k] in ("True", "true")   # noqa: E731
        return cls(bitwidth=int(cfg["bitwidth"]), group_size=int(cfg["group_size"]), is_symmetric=flag("is_symmetric"),
                   is_per_channel=flag("is_per_channel"), is_dynamic=flag("is_dynamic"))

    def to_dict(self) -> dict:
        return {k: str(getattr(self, k)) for k in ("bitwidth", "group_size", "is_symmetric", "is_per_channel", "is_dynamic")}


# ------------------------------------------------------------------------------------------------
class _FakeQuantFn(torch.autograd.Function):
    """HIP fake-quant with the reference's straight-through backward.

    The reference trains through ``Quantizer.forward`` (algorithm.py:381 / :587): ``round_ste`` passes the
    gradient through the rounding, the clamp masks it, and ``scale`` / ``offset`` are learnable.  Both
    directions are HIP kernels (``mq_fake_quant`` / ``mq_fake_quant_backward``); gradients match torch
    autograd of qmodule.py:286-290 (tests: golden ``quantizer_grads.npz``).  fp32 only.
    """

    @staticmethod
    def forward(ctx, x, scale, offset, qmin, qmax):
        ctx.save_for_backward(x, scale, offset)
        ctx.limits = (qmin, qmax)
        return ops.fake_quant(x, scale.detach(), offset.detach(), qmin, qmax)

    @staticmethod
    def backward(ctx, grad_out):
        x, scale, offset = ctx.saved_tensors
        if x.dtype != torch.float32:
            raise NotImplementedError("mobilequant_amd: fake-quant backward is implemented for float32 tensors")
        gx, gs, go = ops.fake_quant_backward(x, grad_out, scale.detach(), offset.detach(), *ctx.limits)
        gs = gs.reshape(scale.shape) if ctx.needs_input_grad[1] else None
        go = go.reshape(offset.shape) if ctx.needs_input_grad[2] else None
        return (gx if ctx.needs_input_grad[0] else None), gs, go, None, None


class _LwcFakeQuantFn(torch.autograd.Function):
    """A weight quantizer in LWC mode, per output row, as ONE HIP pass per direction (mq_lwc_fake_quant / _backward): the row
    ranges (qmodule.py:263-268), sigmoid(bound factor) * range (:271-273), the grid (:40-61) and the fake-quant (:286-290)
    forward; the straight-through gradient plus the range gradients scattered into the extreme elements backward -- what
    ``_RangeFn`` + the [rows, 1]-sized torch chain + ``_FakeQuantFn`` compute in ~40 launches and eight weight-sized passes
    (the e2equant inner step re-derives seven such grids per layer and step, algorithm.py:187-233, :742-747)."""

    @staticmethod
    def forward(ctx, w, sig_lo, sig_hi, bitwidth, is_symmetric):
        y, mn, mx, scale, offset = ops.lwc_fake_quant(w.detach(), sig_lo.detach(), sig_hi.detach(), bitwidth, is_symmetric)
        ctx.save_for_backward(w, sig_lo, sig_hi, mn, mx)
        ctx.cfg = (bitwidth, is_symmetric)
        ctx.mark_non_differentiable(scale, offset)
        ctx.set_materialize_grads(False)          # no zero tensors for the two grid outputs on every backward
        return y, scale, offset

    @staticmethod
    def backward(ctx, grad_y, _gs, _go):
        w, sig_lo, sig_hi, mn, mx = ctx.saved_tensors
        if grad_y is None:
            return None, None, None, None, None
        gw, glo, ghi = ops.lwc_fake_quant_backward(w.detach(), grad_y.contiguous(), sig_lo.detach(), sig_hi.detach(), mn, mx, *ctx.cfg)
        return (gw if ctx.needs_input_grad[0] else None, glo.reshape(sig_lo.shape) if ctx.needs_input_grad[1] else None,
                ghi.reshape(sig_hi.shape) if ctx.needs_input_grad[2] else None, None, None)


class _AttnProbsFn(torch.autograd.Function):
    """qk_bmm's output quantizer -> / sqrt(d) -> + mask -> softmax -> pv_bmm's input quantizer over the [.., S, keys] scores as ONE
    HIP pass per direction (mq_attention_probs_train / _backward; hf_model.py:511-520 under the PTQ training loops): the module chain
    moves the score tensor through HBM ten times per step and keeps four copies for autograd; here the raw scores are the only
    saved tensor.  Gradients: the raw scores and the four learnable grid parameters (--lrl)."""

    @staticmethod
    def forward(ctx, raw, s1, o1, s2, o2, mask, lim1, lim2, sqrt_d):
        ctx.save_for_backward(raw, s1, o1, s2, o2, mask)
        ctx.cfg = (lim1, lim2, sqrt_d)
        return ops.attention_probs_train(raw.detach(), mask, (s1.detach(), o1.detach(), *lim1), (s2.detach(), o2.detach(), *lim2), sqrt_d)

    @staticmethod
    def backward(ctx, grad_out):
        raw, s1, o1, s2, o2, mask = ctx.saved_tensors
        lim1, lim2, sqrt_d = ctx.cfg
        graw, gg = ops.attention_probs_train_backward(raw.detach(), grad_out.contiguous(), mask, (s1.detach(), o1.detach(), *lim1),
                                                      (s2.detach(), o2.detach(), *lim2), sqrt_d)
        need = ctx.needs_input_grad
        return (graw if need[0] else None, gg[0].reshape(s1.shape) if need[1] else None, gg[1].reshape(o1.shape) if need[2] else None,
                gg[2].reshape(s2.shape) if need[3] else None, gg[3].reshape(o2.shape) if need[4] else None, None, None, None, None)


def attention_probs_for_training(qk: "QMatMul", pv: "QMatMul", q, k_t, mask, sqrt_d: float):
    """The probabilities pv_bmm multiplies with v, on ITS input grid -- Q_pv_in(softmax(qk_bmm(q, k^T) / sqrt_d + mask)) -- through
    the fused pass above.  Returns ``(probabilities, True)``; ``None`` when the block is not in that situation at all (no gradient
    wanted, a grid that is not a static per-tensor one, another dtype): the caller runs the module chain; or ``(qk_bmm's output,
    False)`` when the scores were already formed but their shape / the mask's is outside the kernel: the caller continues the chain
    from there.  ``train_fused = False`` on either QMatMul opts out."""
    if not (isinstance(qk, QMatMul) and isinstance(pv, QMatMul)) or not (getattr(qk, "train_fused", True) and getattr(pv, "train_fused", True)):
        return None
    oq, iq = qk.output_quantizer, pv.input_quantizer
    if not (_static_per_tensor(oq, 16) and _static_per_tensor(iq, 16)) or q.dtype != torch.float32:
        return None
    if not _needs_grad(q, k_t, oq.scale, oq.offset, iq.scale, iq.offset):
        return None
    raw = torch.matmul(_apply(qk.input_quantizer, q), _apply(qk.input2_quantizer, k_t))
    if not raw.is_contiguous():
        raw = raw.contiguous()
    if mask is not None and (mask.requires_grad or not ops.attention_probs_train_supported(raw, mask)) or not ops.attention_probs_train_supported(raw, None):
        return _apply(oq, raw), False                      # the caller continues the chain from qk_bmm's output
    for qz in (oq, iq):
        if qz.scale.device != raw.device:
            qz.scale.data, qz.offset.data = qz.scale.to(raw.device), qz.offset.to(raw.device)
    p = _AttnProbsFn.apply(raw, oq.scale, oq.offset, iq.scale, iq.offset, mask, (float(oq.qmin), float(oq.qmax)),
                           (float(iq.qmin), float(iq.qmax)), float(sqrt_d))
    return _tag_grid(p, iq), True


class _RangeFn(torch.autograd.Function):
    """min / max of a weight as the reference's ``torch.amin`` / ``torch.amax`` (qmodule.py:263-268) with their gradient:
    the forward is the single-pass HIP reduction; the backward sends the gradient of a row's (or the tensor's) min / max to
    its extreme element(s), split evenly between ties -- what torch autograd does.  Only reached when learnable weight
    clipping trains through the range (algorithm.py:381 / :587); weight-sized elementwise torch ops, training time only."""

    @staticmethod
    def forward(ctx, x2, per_channel):
        if per_channel:
            lead = x2.shape[:-1]
            mn, mx = ops.minmax_rows(x2.reshape(-1, x2.shape[-1]))
            mn, mx = mn.reshape(*lead, 1), mx.reshape(*lead, 1)
        else:
            mn, mx = ops.minmax_tensor(x2)
            mn, mx = mn.reshape(()), mx.reshape(())
        ctx.save_for_backward(x2, mn, mx)
        ctx.per_channel = per_channel
        return mn.to(x2.dtype), mx.to(x2.dtype)

    @staticmethod
    def backward(ctx, g_mn, g_mx):
        x2, mn, mx = ctx.saved_tensors
        is_mn, is_mx = (x2 == mn.to(x2.dtype)), (x2 == mx.to(x2.dtype))
        if ctx.per_channel:
            n_mn, n_mx = is_mn.sum(-1, keepdim=True), is_mx.sum(-1, keepdim=True)
        else:
            n_mn, n_mx = is_mn.sum(), is_mx.sum()
        gx = is_mn * (g_mn / n_mn) + is_mx * (g_mx / n_mx)
        return gx.to(x2.dtype), None


class Quantizer(nn.Module):
    """Fake-quantizer with cached or on-the-fly (scale, offset) (reference: qmodule.py:112-295)."""

    def __init__(self, qcfg):
        super().__init__()
        self.qcfg = deepcopy(qcfg)
        self.lwc = False
        self.lwc_fused = True         # LWC forward / backward as one HIP pass each where the shape allows (False: the module chain, for A/B)
        self.enable = True
        self._gen = 0                 # bumped whenever (scale, offset) are re-created: cache keys never rely on addresses

    # -- configuration ---------------------------------------------------------------------------
    def update_qcfg(self, qcfg):
        if not isinstance(qcfg, QuantConfig):
            assert isinstance(qcfg, dict)
            qcfg = QuantConfig.from_dict(qcfg)
        self.qcfg = deepcopy(qcfg)
        self._gen = getattr(self, "_gen", 0) + 1
        for name in ("scale", "offset"):          # a new config invalidates the cached grid
            if hasattr(self, name):
                delattr(self, name)

    def export_qcfg(self):
        return self.qcfg.to_dict()

    def _has_grid(self) -> bool:
        return hasattr(self, "scale") and hasattr(self, "offset")

    # -- learnable weight clipping (training-time; reference: qmodule.py:133-185) ------------------
    def enable_lwc(self, w):
        self.lwc = True
        if self.qcfg.group_size != -1:
            rows = int(w.shape[0] * math.ceil(w.shape[1] / self.qcfg.group_size))
        else:
            rows = w.shape[0]
        shape = (rows, 1) if self.qcfg.is_per_channel else (1,)
        init = torch.full(shape, 4.0, dtype=w.dtype, device=w.device)
        self.upbound_factor = nn.Parameter(init.clone())
        self.lowbound_factor = nn.Parameter(init.clone())

    def disable_lwc(self):
        self.lwc = False
        for name in ("upbound_factor", "lowbound_factor"):
            if hasattr(self, name):
                delattr(self, name)

    def _tensor_range(self, x2):
        """min/max of the (already group-reshaped) tensor, with the LWC factors applied if enabled."""
        if self.lwc and _needs_grad(x2):          # the reference's amin / amax back-propagate into the extreme elements
            mn, mx = _RangeFn.apply(x2, self.qcfg.is_per_channel)
            return torch.sigmoid(self.lowbound_factor) * mn, torch.sigmoid(self.upbound_factor) * mx
        if self.qcfg.is_per_channel:
            lead = x2.shape[:-1]
            mn, mx = ops.minmax_rows(x2.reshape(-1, x2.shape[-1]))
            mn, mx = mn.reshape(*lead, 1), mx.reshape(*lead, 1)
        else:
            mn, mx = ops.minmax_tensor(x2)
            mn, mx = mn.reshape(()), mx.reshape(())
        if self.lwc:
            mx = torch.sigmoid(self.upbound_factor) * mx
            mn = torch.sigmoid(self.lowbound_factor) * mn
        return mn, mx

    def run_lwc(self, input_):
        """Clamp a weight to its (learned) clipping range and drop the LWC state (qmodule.py:159-185)."""
        grouped = self.qcfg.is_per_channel and self.qcfg.group_size != -1
        x = input_.reshape(-1, self.qcfg.group_size) if grouped else input_
        mn, mx = self._tensor_range(x)
        if self.lwc:
            for name in ("scale", "offset"):
                if hasattr(self, name):
                    delattr(self, name)
            self.disable_lwc()
        x = torch.maximum(torch.minimum(x, mx.to(x.dtype)), mn.to(x.dtype))
        return x.reshape(input_.shape).type(input_.dtype)

    # -- grid -----------------------------------------------------------------------------------
    def set_scale_offset_from_minmax(self, min_val, max_val, cache_mode=None, device=None):
        scale, offset, _, _, q_min, q_max = compute_scale_offset_from_min_max(
            min_val, max_val, self.qcfg.bitwidth, self.qcfg.is_symmetric)
        self.qmin, self.qmax = q_min, q_max
        self._gen = getattr(self, "_gen", 0) + 1
        self._auto_grid = False                # (_prepare sets it again when the range came from the tensor being quantised)
        scale, offset = scale.to(device), offset.to(device)
        for name in ("scale", "offset"):
            if hasattr(self, name):
                delattr(self, name)
        if cache_mode == "parameter":
            self.register_parameter("scale", nn.Parameter(scale))
            self.register_parameter("offset", nn.Parameter(offset))
        elif cache_mode == "buffer":
            self.register_buffer("scale", scale)
            self.register_buffer("offset", offset)
        else:
            self.scale, self.offset = scale, offset
        # Host-side identity of a grid that was set from Python numbers (calibrated ranges out of act_dict.json):
        # two quantizers built from the same (min, max, bitwidth, symmetry) ARE the same grid even though their
        # scale tensors are different objects -- what lets a producer hand its integer output to a consumer
        # (grid_token).  Dropped as soon as scale / offset are touched again.
        if isinstance(min_val, (int, float)) and isinstance(max_val, (int, float)):
            self._host_range = (float(min_val), float(max_val), int(self.qcfg.bitwidth), bool(self.qcfg.is_symmetric),
                                self._gen, _ver(self.scale), _ver(self.offset))
        else:
            self._host_range = None

    def grid_token(self):
        """Hashable identity of the current per-tensor grid: value based when the range came from host numbers and
        the tensors were not modified since, storage based otherwise."""
        hr = getattr(self, "_host_range", None)
        if hr is not None and self._has_grid() and hr[4:] == (self._gen, _ver(self.scale), _ver(self.offset)):
            return ("host",) + hr[:4]
        return ("dev", id(self), self._gen, _ver(self.scale), _ver(self.offset), self.qmin, self.qmax)

    def set_scale_offset_from_tensor(self, x, cache_mode=None):
        mn, mx = compute_min_max_from_tensor(x, self.qcfg.is_per_channel, self.qcfg.group_size)
        self.set_scale_offset_from_minmax(mn, mx, cache_mode, x.device)

    def _prepare(self, x2, use_scale_offset_as):
        """Make sure (scale, offset) exist for this call and sit on x's device (qmodule.py:262-283)."""
        if self.qcfg.is_dynamic or self.lwc or not self._has_grid():
            mn, mx = self._tensor_range(x2)
            transient = self.qcfg.is_dynamic or self.lwc
            self.set_scale_offset_from_minmax(mn, mx, None if transient else use_scale_offset_as, x2.device)
            # derived from the tensor itself (first forward), not loaded / calibrated / trained.  The stamp lets a later reader tell an
            # untouched auto grid from one the optimizer stepped or load_state_dict copied into IN PLACE (same Parameter objects, new
            # versions): only the former may be re-derived (auto_grid_untouched)
            self._auto_grid = (self._gen, _ver(self.scale), _ver(self.offset))
        if self.scale.device != x2.device:
            self.scale.data = self.scale.to(x2.device)
        if self.offset.device != x2.device:
            self.offset.data = self.offset.to(x2.device)

    def _set_transient_grid(self, scale, offset):
        """What _prepare leaves behind in LWC mode (set_scale_offset_from_minmax with cache_mode None): plain tensors, a new generation."""
        self.qmin, self.qmax = _grid_limits(self.qcfg.bitwidth, self.qcfg.is_symmetric)
        self._gen = getattr(self, "_gen", 0) + 1
        for name in ("scale", "offset"):
            if hasattr(self, name):
                delattr(self, name)
        self.scale, self.offset = scale, offset
        self._host_range = None
        self._auto_grid = (self._gen, _ver(self.scale), _ver(self.offset))

    def auto_grid_untouched(self) -> bool:
        """True while (scale, offset) are exactly what the first forward derived from the tensor it quantised: not set from ranges, not
        loaded (load_state_dict copies into the existing tensors and bumps their versions), not stepped by an optimizer."""
        st = getattr(self, "_auto_grid", False)
        return bool(st) and self._has_grid() and st == (self._gen, _ver(self.scale), _ver(self.offset))

    def _load_from_state_dict(self, state_dict, prefix, *args, **kwargs):
        # a grid that arrives from a checkpoint is the caller's, whatever produced the tensors it lands in
        if any(k in state_dict for k in (prefix + "scale", prefix + "offset")):
            self._auto_grid = False
        return super()._load_from_state_dict(state_dict, prefix, *args, **kwargs)

    def bypassed(self) -> bool:
        return (not self.enable) or self.qcfg.bitwidth > 16

    # -- forward ---------------------------------------------------------------------------------
    def forward(self, input_, use_scale_offset_as="parameter"):
        if self.bypassed():
            return input_
        grouped = self.qcfg.is_per_channel and self.qcfg.group_size != -1
        x = input_.reshape(-1, self.qcfg.group_size) if grouped else input_
        if (self.lwc and getattr(self, "lwc_fused", True) and self.qcfg.is_per_channel and x.dim() == 2 and x.shape[1] >= 64
                and self.upbound_factor.numel() == x.shape[0] and ops.lwc_fake_quant_supported(x)):
            # one pass per direction instead of range reduction + ~12 [rows, 1]-sized launches + fake-quant (and ~35 launches backward)
            y, scale, offset = _LwcFakeQuantFn.apply(x, torch.sigmoid(self.lowbound_factor), torch.sigmoid(self.upbound_factor),
                                                     int(self.qcfg.bitwidth), bool(self.qcfg.is_symmetric))
            self._set_transient_grid(scale.view(-1, 1), offset.view(-1, 1))
            return y.reshape(input_.shape) if grouped else _tag_grid(y, self)
        self._prepare(x, use_scale_offset_as)
        if _needs_grad(x, self.scale, self.offset):
            y = _FakeQuantFn.apply(x, self.scale, self.offset, self.qmin, self.qmax)
        else:
            y = ops.fake_quant(x, self.scale.detach(), self.offset.detach(), self.qmin, self.qmax)
        return y.reshape(input_.shape) if grouped else _tag_grid(y, self)

    def quantize_to_int(self, x, q_dtype=MQ_I8, want_row_sum=False, rows=None, chan_scale=None):
        """Integer indices of x on this quantizer's grid (static per-tensor or per-row grids).
        Returns (q, row_sum or None, shift): int8 storage subtracts shift = 128 from unsigned grids."""
        shift = 128 if (q_dtype == MQ_I8 and self.qmax > 127) else 0
        out = ops.quantize(x, self.scale.detach(), self.offset.detach(), self.qmin, self.qmax, q_dtype=q_dtype,
                           shift=shift, rows=rows, want_row_sum=want_row_sum, chan_scale=chan_scale)
        q, rs = out if want_row_sum else (out, None)
        return q, rs, shift


# ------------------------------------------------------------------------------------------------
class _QuantizedOp:
    """Shared config/range plumbing of the Q* modules.  ``_slots`` lists (role, attribute) pairs in
    the order the reference's update_qcfg signatures take them."""

    _slots = ()
    _optional_update = ()      # roles whose config may be None in update_qcfg (QSiLU / QGELU input)

    def _init_quantizers(self, **cfgs):
        for role, attr in self._slots:
            cfg = cfgs.get(role)
            setattr(self, attr, Quantizer(cfg) if cfg is not None else None)

    def update_qcfg(self, *cfgs):
        for (role, attr), cfg in zip(self._slots, cfgs):
            q = getattr(self, attr)
            if q is None or (cfg is None and role in self._optional_update):
                continue
            q.update_qcfg(cfg)

    def export_qcfg(self):
        return {role: getattr(self, attr).export_qcfg() for role, attr in self._slots if getattr(self, attr) is not None}

    def _range_device(self):
        w = getattr(self, "weight", None)
        return w.device if torch.is_tensor(w) else None

    def set_scale_offset(self, act_scale, use_scale_offset_as="parameter"):
        # weights get no preset range: their statistics are computed on the first forward
        self._act_range = {k: act_scale[k] for k in ("input", "input2", "output") if k in act_scale}
        for role, attr in self._slots:
            q = getattr(self, attr)
            if q is None or role == "weight":
                continue
            if role == "input2" and "input2" not in act_scale and isinstance(self, QSiLU):
                lo, hi = 0.0, 1.0                      # sigmoid range (reference: qmodule.py:731-734)
            else:
                lo, hi = act_scale[role][0], act_scale[role][1]
            q.set_scale_offset_from_minmax(lo, hi, use_scale_offset_as, self._range_device())


def _apply(q: Optional[Quantizer], x):
    return x if q is None else q(x)


def _static_per_tensor(q: Optional[Quantizer], max_bits: int) -> bool:
    return (q is not None and not q.bypassed() and q.qcfg.bitwidth <= max_bits and not q.qcfg.is_dynamic
            and not q.lwc and not q.qcfg.is_per_channel and q._has_grid() and q.scale.numel() == 1)


def _dynamic_per_tensor(q: Optional[Quantizer], max_bits: int) -> bool:
    """A per-tensor grid that is re-derived from the tensor on every call (qmodule.py:262-277, `is_dynamic`): min / max, scale and offset
    stay on the device (mq_minmax_tensor -> mq_scale_offset_from_minmax), so the integer kernels take it by pointer like a static one."""
    return (q is not None and not q.bypassed() and q.qcfg.bitwidth <= max_bits and q.qcfg.is_dynamic and not q.lwc
            and not q.qcfg.is_per_channel)


class _SharedActivation:
    """Memo of the integer images of the LAST activation tensor quantised for an int8 linear.  q_proj / k_proj / v_proj
    (and w1 / w3) are called with the SAME tensor object and the same input grid, so the reference's three (two)
    identical fake-quant passes collapse into one quantize launch.  Keyed on the tensor OBJECT (weak reference) and its
    version counter -- a different tensor that happens to reuse the address can never hit -- and, per entry, on the
    grid's identity (Quantizer.grid_token) and the layout tag (a_shift for row-major, ("tiled", a_shift) for the
    fragment-blocked layout).  The fused norm kernels file their int8 outputs here too, so their consumers find them."""

    def __init__(self):
        self.clear()

    def clear(self):
        self._ref = None
        self._base = None
        self._vals = {}

    @staticmethod
    def _base_key(x):
        stream = torch.cuda.current_stream(x.device).cuda_stream if x.is_cuda else 0      # images are ordered on ONE stream
        return (_ver(x), x.data_ptr(), tuple(x.shape), x.dtype, stream)

    def _same(self, x):
        return self._ref is not None and self._ref() is x and self._base == self._base_key(x)

    def get(self, x, grid, tag):
        return self._vals.get((grid.grid_token(), tag)) if self._same(x) else None

    def put(self, x, grid, tag, val):
        if not self._same(x):
            try:
                ref = weakref.ref(x)
            except TypeError:
                self.clear()
                return
            self._ref, self._base, self._vals = ref, self._base_key(x), {}
        self._vals[(grid.grid_token(), tag)] = val


class _PerThread(threading.local):
    def __init__(self):
        self.memo = _SharedActivation()


class _SharedActivationProxy:
    """One memo per Python thread (and, through the key, per HIP stream): two threads driving two models never see each
    other's activation images."""

    _tls = _PerThread()

    def __getattr__(self, name):
        return getattr(self._tls.memo, name)


_shared_activation = _SharedActivationProxy()


class QLinear(nn.Linear, _QuantizedOp):
    """``nn.Linear`` with weight / input / output quantizers (reference: qmodule.py:298-405)."""

    _slots = (("input", "input_quantizer"), ("weight", "weight_quantizer"), ("output", "output_quantizer"))

    def __init__(self, kargs, input_quant_cfg, weight_quant_cfg, output_quant_cfg):
        super().__init__(**kargs)
        self.use_temporary_parameter = False
        self._init_quantizers(input=input_quant_cfg, weight=weight_quant_cfg, output=output_quant_cfg)
        self.int8_mode = "auto"        # "auto" | "off": real-int8 MFMA path when the config allows it
        self._input_grid = None        # producer's output grid for linears without an input quantizer
        self._plan = None
        self.input_chan_scale = None   # SmoothQuant per-input-channel scale applied at RUN time (set_input_channel_scale)
        self._scaled_weight = None

    # -- integer path ------------------------------------------------------------------------------
    def set_input_grid(self, min_val, max_val, bitwidth=8, is_symmetric=False):
        """Declare the grid the incoming activation already sits on (the producer's output quantizer).
        q/k/v/o/w1/w3 have no input quantizer in the reference (qmodule.py:848-850): their inputs are
        fake-quantised by the previous module, so re-quantising on that same grid is the identity."""
        q = Quantizer(QuantConfig(bitwidth=bitwidth, is_symmetric=is_symmetric))
        q.set_scale_offset_from_minmax(min_val, max_val, None, self.weight.device)
        self._input_grid = q

    # -- SmoothQuant at run time ---------------------------------------------------------------------
    def set_input_channel_scale(self, scales: Optional[torch.Tensor]):
        """Apply a SmoothQuant per-input-channel scale s at RUN time: ``out = Qout(linear(Qin(x / s), Qw(W * s)))``.
        The reference folds s offline into the producer (``ln.weight /= s``) and the consumer (``fc.weight *= s``;
        ptq/smoothquant.py:64-69, algorithm.py:47-68); this is the same transformation for a producer that cannot absorb
        1/s: the division is fused into the activation quantize kernel (``mq_quantize(..., chan_scale)``), the integer
        weights are formed from ``W * s``.  Needs this linear's own input quantizer (its grid is that of x / s).
        ``None`` removes the scale."""
        if scales is None:
            self.input_chan_scale = None
        else:
            s = scales.detach().to(device=self.weight.device, dtype=torch.float32).reshape(-1).contiguous()
            assert s.numel() == self.in_features, (s.numel(), self.in_features)
            assert self.input_quantizer is not None, "a run-time channel scale needs the linear's own input quantizer"
            self.input_chan_scale = s
        self._scaled_weight = None
        self._plan = None
        # The weight quantizer's cached grid (first-forward range of W, qmodule.py:262-277) belongs to the OLD effective weight: the
        # reference folds first and quantises after (smoothquant.py:64-69), so the grid of W * s must come from W * s.  A grid that
        # was loaded, calibrated or trained is the caller's: left alone.
        wq = self.weight_quantizer
        if wq is not None and wq._has_grid() and wq.auto_grid_untouched():
            wq._gen = getattr(wq, "_gen", 0) + 1
            for name in ("scale", "offset"):
                delattr(wq, name)
        return self

    def _effective_weight(self, weight):
        """W * s (per input channel), cached until W or s change."""
        cs = self.input_chan_scale
        if cs is None:
            return weight
        key = (weight.data_ptr(), _ver(weight), cs.data_ptr(), _ver(cs))
        hit = self._scaled_weight
        if hit is not None and hit[0] == key and hit[1]() is weight and not _needs_grad(weight):
            return hit[2]
        w = weight * cs.view(1, -1).to(weight.dtype)
        if not _needs_grad(weight):
            self._scaled_weight = (key, weakref.ref(weight), w)
        return w

    def _activation_grid(self, x=None, refresh=False) -> Optional[Quantizer]:
        """The grid the int8 image of x is formed on: the own input quantizer; else the LIVE output quantizer of the module
        that produced x (tag left on the tensor object); else the grid declared by wire_integer_inputs (for inputs whose tag
        is lost on the way, e.g. o_proj behind a transpose / reshape).  None -> simulated path (e.g. a 16-bit producer).
        A DYNAMIC own input quantizer (round 4) is served too: refresh=True re-derives its grid from x on the device (once per
        forward, by _input_image); a dynamic producer's grid is the one its forward has just set."""
        iq = self.input_quantizer
        if iq is not None and not iq.bypassed():
            if _static_per_tensor(iq, 8):
                return iq
            if _dynamic_per_tensor(iq, 8) and x is not None:
                if refresh:
                    xin = _materialize(x)
                    if self.input_chan_scale is not None:
                        xin = xin / self.input_chan_scale.to(xin.dtype)          # the grid of x / s comes from x / s
                    iq._prepare(xin.reshape(-1, xin.shape[-1]), None)
                return iq
            return None
        prod = _producer_grid(x) if x is not None else None
        if prod is not None:
            ok = _static_per_tensor(prod, 8) or (_dynamic_per_tensor(prod, 8) and prod._has_grid() and prod.scale.numel() == 1)
            return prod if ok else None
        return self._input_grid

    # 4-bit weights at prefill (M > 8): "image" = one byte per nibble on the int8 kernels (+ the packed image, built on first decode use);
    # "packed" = the packed image only, for every kernel (include/mobilequant_amd.h: mq_w4a8_linear_tiled / mq_w4a8_linear)
    w4_prefill = "image"

    def _int8_reason(self, x, weight) -> Optional[str]:
        """None when this call can run on the integer kernels, else WHY it takes the simulated path (HIP fake-quant around the
        fp32 library GEMM: the reference's semantics, ~10x slower) -- recorded per module, read by int8_coverage()."""
        if self.int8_mode == "off":
            return "int8_mode off"
        if not x.is_cuda:
            return "not a device tensor"
        if x.dtype not in (torch.float32, torch.float16):
            return f"activation dtype {x.dtype}"
        if x.numel() == 0:
            return "empty batch"              # (every op of the simulated path handles empties)
        wq = self.weight_quantizer
        if wq is None or wq.bypassed():
            return "no weight quantizer"
        if wq.qcfg.bitwidth > 8:
            return f"{wq.qcfg.bitwidth}-bit weights"
        if wq.qcfg.is_dynamic:
            return "dynamic weight quantizer"
        if wq.lwc:
            return "learnable weight clipping active"
        if wq.qcfg.is_per_channel and wq.qcfg.group_size != -1:
            return "per-group weight grid"
        K, N = weight.shape[1], weight.shape[0]
        M = x.numel() // max(K, 1)
        if K % 128 or N % 4 or K > 65536 or M * K >= 2 ** 31 or N * K >= 2 ** 31 or M * N >= 2 ** 40:
            return f"shape M={M} K={K} N={N} outside mq_w8a8_linear's limits"
        if (x.data_ptr() % 16 and x.is_contiguous()) or (self.bias is not None and self.bias.data_ptr() % 16):
            return "operand not 16-byte aligned"
        if self._activation_grid(x) is None:
            iq = self.input_quantizer
            if iq is not None and not iq.bypassed():
                return (f"input quantizer: {iq.qcfg.bitwidth}-bit" + (" per-channel" if iq.qcfg.is_per_channel else "")
                        + (" lwc" if iq.lwc else "") + (" without a range" if not iq._has_grid() and not iq.qcfg.is_dynamic else ""))
            return "no 8-bit per-tensor grid on the input (16-bit / per-channel / untagged producer)"
        params = [x, weight, self.bias, getattr(wq, "scale", None)]
        if _needs_grad(*params):
            return "gradient required"
        return None

    def _int8_ready(self, x, weight) -> bool:
        return self._int8_reason(x, weight) is None

    # -- per-group weight grids (Quantizer.group_size != -1; qmodule.py:259-260, :292-293) on the integer path -----------------------
    def _grouped_reason(self, x, weight) -> Optional[str]:
        """None when mq_w8a8_linear_grouped serves this call, else why the per-group recipe stays simulated."""
        wq = self.weight_quantizer
        gs, (N, K) = int(wq.qcfg.group_size), weight.shape
        M = x.numel() // max(K, 1)
        if gs % 64 or K % gs or N % 128:
            return f"per-group weight grid: group_size {gs} / K {K} / N {N} outside mq_w8a8_linear_grouped (group_size % 64, K % group_size, N % 128)"
        if M * K >= 2 ** 31 or N * K >= 2 ** 31 or M * N >= 2 ** 31:
            return "per-group weight grid: operand too large"
        if self._activation_grid(x) is None:
            return "per-group weight grid: no 8-bit per-tensor grid on the input"
        if _needs_grad(x, weight, self.bias, getattr(wq, "scale", None)):
            return "gradient required"
        return None

    def _grouped_plan(self, weight):
        """stored int8 weights [N, K], per-group sums, scales, offsets [N, G]; cached like _weight_plan"""
        wq = self.weight_quantizer
        if not wq._has_grid():
            wq._prepare(weight.reshape(-1, wq.qcfg.group_size), "parameter")
        key = ("grouped", weight.data_ptr(), _ver(weight), tuple(weight.shape), wq.grid_token(), wq.qcfg.bitwidth, wq.qcfg.is_symmetric,
               wq.qcfg.group_size)
        plan = getattr(self, "_gplan", None)
        if plan is not None and plan["key"] == key and plan["wref"]() is weight:
            return plan
        N, K = weight.shape
        gs = int(wq.qcfg.group_size)
        G = K // gs
        shift = 128 if wq.qmax > 127 else 0
        w32 = weight.detach().to(torch.float32).reshape(N * G, gs)
        q, gsum = ops.quantize(w32, wq.scale.detach().reshape(-1), wq.offset.detach().reshape(-1), wq.qmin, wq.qmax, q_dtype=MQ_I8,
                               shift=shift, rows=N * G, want_row_sum=True)
        cw = (shift - wq.offset.detach().reshape(N, G)).round().to(torch.int32)
        # the kernel folds acc + mul24(cw, a_gsum) + t in int32 (mq_gemm_grouped.hip): |cw| must fit 24 bits and the bracket 31.  A
        # narrow group far from zero (scale at CLIPMIN, offset = -round(min / scale)) breaks that: ONE host read per weight version
        # decides, the forward falls back to the simulated path (the activation side of the bound is checked on the device, below)
        cw_max = int(cw.abs().max())
        plan = {"key": key, "wref": weakref.ref(weight), "w": q.view(N, K), "wsum_t": gsum.view(N, G).t().contiguous(),
                "cw_t": cw.t().contiguous(), "sw_t": wq.scale.detach().reshape(N, G).t().contiguous().float(), "G": G, "gs": gs,
                "cw_max": cw_max, "fold_ok": cw_max < 2 ** 23 and (cw_max + 128) * 128 * gs < 2 ** 30}
        self._gplan = plan
        return plan

    def _forward_int8_grouped(self, x, weight, bias):
        """x -> int8 image on the input grid (+ its per-group row sums) -> mq_w8a8_linear_grouped -> fp32 -> output quantizer.  The
        group vectors follow the activation grid by device arithmetic (no host read-back: dynamic input grids work the same way)."""
        grid = self._activation_grid(x, refresh=True)
        plan = self._grouped_plan(weight)
        N, K = weight.shape
        if grid.scale.device != x.device:
            grid.scale.data, grid.offset.data = grid.scale.to(x.device), grid.offset.to(x.device)
        x2d = _materialize(x).reshape(-1, K)
        M = x2d.shape[0]
        a_q, _, a_shift = grid.quantize_to_int(x2d, MQ_I8, want_row_sum=False, chan_scale=self.input_chan_scale)
        a_gsum = a_q.view(M, plan["G"], plan["gs"]).sum(-1, dtype=torch.int32).t().contiguous()
        epi_key = (grid.grid_token(), a_shift)
        if plan.get("epi_key") != epi_key:               # static grids: once; dynamic grids: per call (a handful of [G, N]-sized launches)
            c_a = (a_shift - grid.offset.detach().reshape(())).round().to(torch.int64)
            t64 = c_a * plan["wsum_t"].to(torch.int64) + (plan["gs"] * c_a) * plan["cw_t"].to(torch.int64)   # (a 0-dim int64 does not promote)
            # int32 bracket of the kernel: |P_g| <= 2^14 gs, |cw A_g| <= cw_max 128 gs, |T|.  An activation grid far from zero (c_a huge)
            # would wrap it: decided on the device (dynamic grids have no host copy) -- alpha becomes NaN, the output is loudly wrong
            fits = (t64.abs().max() + (plan["cw_max"] + 128) * 128 * plan["gs"]) < 2 ** 31
            plan["t"] = t64.to(torch.int32).contiguous()
            alpha = grid.scale.detach().reshape(()).float() * plan["sw_t"]
            plan["alpha"] = torch.where(fits, alpha, torch.full_like(alpha, float("nan"))).contiguous()
            plan["epi_key"] = epi_key
        y = ops.int8_linear_grouped(a_q, plan["w"], plan["gs"], a_gsum, plan["alpha"], plan["cw_t"], plan["t"],
                                    None if bias is None else bias.detach().float())
        y = y.reshape(*x.shape[:-1], N).to(x.dtype)
        return _apply(self.output_quantizer, y)

    def _weight_plan(self, weight):
        """Integer weights + column sums, cached until the weight or its quantizer changes (SURVEY 8a' item 5)."""
        wq = self.weight_quantizer
        if not wq._has_grid():
            wq._prepare(weight, "parameter")           # first forward: range from the weight itself
        packed_only = wq.qcfg.bitwidth == 4 and QLinear.w4_prefill == "packed" and weight.shape[1] % 32 == 0
        key = (weight.data_ptr(), _ver(weight), tuple(weight.shape), wq.grid_token(), wq.qcfg.bitwidth, wq.qcfg.is_symmetric,
               wq.qcfg.is_per_channel, packed_only)
        plan = self._plan
        if plan is not None and plan["key"] == key and plan["wref"]() is weight:
            return plan
        bits4 = wq.qcfg.bitwidth == 4
        w32 = weight.detach().to(torch.float32)
        if bits4:
            # 4-bit weights: unsigned nibble values (index - qmin) in 0 .. 15.  Two images of the SAME numbers, with the same epilogue
            # vectors: one byte per value for M > 8 -- a prefill GEMM is compute-bound, so it runs the int8 MFMA kernels (generated-ISA
            # pair / tiled kernels, fused residual, segmented q|k|v) at their int8 speed instead of unpacking nibbles in its inner loop
            # -- and the packed two-per-byte image (built on first use) for the weight-streaming decode kernels, which are
            # bandwidth-bound and unpack in registers (mq_gemv.hip, mq_decode.hip).  mq_w4a8_linear (in-LDS unpack in front of the MFMAs)
            # stays in the C ABI for callers that hold only packed weights.
            q, colsum = ops.quantize(w32, wq.scale.detach(), wq.offset.detach(), wq.qmin, wq.qmax, q_dtype=MQ_U8,
                                     shift=wq.qmin, rows=weight.shape[0], want_row_sum=True)
            wint, shift = q.view(torch.int8), wq.qmin
            if packed_only:
                # QLinear.w4_prefill = "packed" (round 4): ONE image, two nibbles per byte (0.5 B / weight), for prefill and decode alike --
                # mq_w4a8_linear_tiled (generated ISA: the pieces are expanded once per workgroup into the int8 W ring) where a fused block
                # asks for 8-bit indices, mq_w4a8_linear elsewhere.  Measured 5-10 % behind the int8 image on the GEMMs it serves, so the
                # image stays the default.
                wint = ops.pack_w4(q)
        else:
            wint, colsum, shift = wq.quantize_to_int(w32, MQ_I8, want_row_sum=True, rows=weight.shape[0])
        plan = {"key": key, "wref": weakref.ref(weight), "w": wint, "colsum": colsum, "shift": shift, "w4": packed_only, "bits4": bits4,
                "packed": wint if packed_only else None, "epi_key": None}
        self._plan = plan
        return plan

    @staticmethod
    def _decode_weights(plan):
        """(weight image, w4 flag) for the M <= 8 weight-streaming kernels: packed nibbles for 4-bit weights."""
        if not plan["bits4"]:
            return plan["w"], False
        if plan["packed"] is None:
            plan["packed"] = ops.pack_w4(plan["w"].view(torch.uint8))
        return plan["packed"], True

    def _input_image(self, x, weight):
        """int8 image of the activation on this linear's input grid: (grid, a_q, a_rs, a_shift, tiled_rows, decode).  An image
        left by a producer (fused norm) or a sibling linear wins: no quantize launch at all; the large FFN shapes want the
        fragment-blocked layout of the generated-ISA GEMM kernels.  decode (M <= 8): no image, the GEMV quantises itself."""
        grid = self._activation_grid(x, refresh=True)
        plan = self._weight_plan(weight)
        K, N = weight.shape[1], weight.shape[0]
        if grid.scale.device != x.device:
            grid.scale.data, grid.offset.data = grid.scale.to(x.device), grid.offset.to(x.device)
        x2d = x.reshape(-1, K)
        decode = x.dtype == torch.float32 and ops.decode_shape(x2d.shape[0], K)
        a_shift = 128 if grid.qmax > 127 else 0
        tiled_rows = None
        cs = self.input_chan_scale
        cs_tag = None if cs is None else ("cs", cs.data_ptr(), _ver(cs))      # images of x / s are not images of x
        if decode and cs is not None:
            decode = False                  # the fused decode GEMV quantises x itself: take quantize + GEMV instead
        if decode:
            return grid, None, None, a_shift, None, True
        eligible = (not plan["w4"]) and ops.gemm_tiled_supported(x2d.shape[0], N, K)
        hit = _shared_activation.get(x, grid, ("tiled", a_shift, cs_tag)) if eligible else None
        if hit is not None:
            tiled_rows = x2d.shape[0]
        else:
            hit = _shared_activation.get(x, grid, (a_shift, cs_tag))
        if hit is None and getattr(x, "_mq_image_only", False):
            x2d = _materialize(x).reshape(-1, K)        # the producer left its image in another layout: rebuild the values
        if hit is None and eligible:
            q_t, rs_t = ops.quantize_tiled(x2d, grid.scale.detach(), grid.offset.detach(), grid.qmin, grid.qmax, a_shift,
                                           chan_scale=cs)
            hit = (q_t, rs_t, a_shift)
            _shared_activation.put(x, grid, ("tiled", a_shift, cs_tag), hit)
            tiled_rows = x2d.shape[0]
        elif hit is None:
            hit = grid.quantize_to_int(x2d, MQ_I8, want_row_sum=True, chan_scale=cs)
            _shared_activation.put(x, grid, (a_shift, cs_tag), hit)
        a_q, a_rs, a_shift = hit
        return grid, a_q, a_rs, a_shift, tiled_rows, False

    def _tiled_residual_ok(self, M, N, K):
        """x + Q16(linear) from a fragment-blocked activation image (mq_w8a8_linear_tiled_residual): o_proj / w2 of the fused layer."""
        oq = self.output_quantizer
        return (oq is not None and not oq.bypassed() and _static_per_tensor(oq, 16) and oq.qmax - oq.qmin > 255
                and ops.gemm_tiled128_supported(M, N, K))

    def _forward_int8(self, x, weight, bias):
        grid, a_q, a_rs, a_shift, tiled_rows, decode = self._input_image(x, weight)
        return self._int8_from_image(x, weight, bias, grid, a_q, a_rs, a_shift, tiled_rows, decode=decode)

    def _epilogue_vectors(self, plan, grid, a_shift, K):
        """per-n dequantisation vectors of (activation grid, weight grid); every integer GEMM of this module -- its own forward or a
        fused block's -- passes through here: counted for int8_coverage()"""
        self._count_path(None)
        wq = self.weight_quantizer
        epi_key = (grid.grid_token(), a_shift)
        if plan["epi_key"] != epi_key:
            plan["alpha"], plan["w_zp"], plan["col_term"] = ops.linear_epilogue_prepare(
                grid.scale.detach(), grid.offset.detach(), a_shift, wq.scale.detach(), wq.offset.detach(),
                plan["shift"], plan["colsum"], K)
            plan["epi_key"] = epi_key
        return plan

    def _int8_from_image(self, x, weight, bias, grid, a_q, a_rs, a_shift, tiled_rows, decode=False, lead_shape=None, resid=None,
                         skip_oq=False):
        """The GEMM half of the integer path: a ready int8 image of the activation (row-major, or fragment-blocked with
        tiled_rows) on `grid` -> this linear's output.  x only supplies dtype / leading shape (and the fp32 values for the
        fused decode GEMV); it may be None when lead_shape is given.  resid (fp32, the output's shape): returns
        resid + output -- fused into the GEMM's store where the kernel allows (no launch, no pass), a plain add elsewhere."""
        oq = None if skip_oq else self.output_quantizer
        K, N = weight.shape[1], weight.shape[0]
        active = oq is not None and not oq.bypassed()
        fused = active and _static_per_tensor(oq, 16)
        if active and not fused:
            # an output grid the GEMM epilogue cannot take by pointer BEFORE it has the values (dynamic: its range is the output's own
            # min / max, qmodule.py:262-277; per-channel; learnable clipping): integer GEMM -> fp values -> the HIP Quantizer, as the
            # reference orders it (qmodule.py:353-357)
            out = self._int8_from_image(x, weight, bias, grid, a_q, a_rs, a_shift, tiled_rows, decode=decode, lead_shape=lead_shape,
                                        skip_oq=True)
            out = oq(out)
            return out if resid is None else resid + out
        plan = self._epilogue_vectors(self._weight_plan(weight), grid, a_shift, K)
        dev = a_q.device if a_q is not None else x.device
        if fused and oq.scale.device != dev:
            oq.scale.data, oq.offset.data = oq.scale.to(dev), oq.offset.to(dev)
        lead = tuple(lead_shape) if lead_shape is not None else tuple(x.shape[:-1])
        f16 = x is not None and x.dtype == torch.float16
        if decode:   # M <= 8: activation quantize fused into the weight-streaming GEMV (one launch)
            x2d = x.reshape(-1, K)
            w_dec, w4_dec = self._decode_weights(plan)
            out = ops.int8_linear_f32in(
                x2d, grid.scale.detach(), grid.offset.detach(), grid.qmin, grid.qmax, a_shift, w_dec, plan["alpha"],
                plan["w_zp"], plan["col_term"], bias, out_scale=oq.scale.detach() if fused else None,
                out_offset=oq.offset.detach() if fused else None, out_qmin=oq.qmin if fused else 0.0,
                out_qmax=oq.qmax if fused else 0.0, out_dtype=MQ_F32, w4=w4_dec)
            out = out.reshape(*lead, N)
            if resid is not None:
                return resid + out
            return _tag_grid(out, oq) if fused else out
        if resid is not None:
            rows = a_q.shape[0] if tiled_rows is None else tiled_rows
            # packed-only 4-bit weights: the generated residual kernel expands them from the packed image (mq_w4a8_linear_tiled_residual)
            if ((not plan["w4"] or tiled_rows is not None) and not f16 and rows > 8 and resid.dtype == torch.float32 and resid.is_contiguous()
                    and (tiled_rows is None or self._tiled_residual_ok(rows, N, K))):
                out = ops.int8_linear(
                    a_q, plan["w"], a_rs, plan["alpha"], plan["w_zp"], plan["col_term"], bias,
                    out_scale=oq.scale.detach() if fused else None, out_offset=oq.offset.detach() if fused else None,
                    out_qmin=oq.qmin if fused else 0.0, out_qmax=oq.qmax if fused else 0.0, out_dtype=MQ_F32, resid=resid,
                    a_tiled_rows=tiled_rows, w4=plan["w4"])
                return out.reshape(*lead, N)
            return resid + self._int8_from_image(x, weight, bias, grid, a_q, a_rs, a_shift, tiled_rows, decode, lead_shape)
        out = ops.int8_linear(
            a_q, plan["w"], a_rs, plan["alpha"], plan["w_zp"], plan["col_term"], bias,
            out_scale=oq.scale.detach() if fused else None, out_offset=oq.offset.detach() if fused else None,
            out_qmin=oq.qmin if fused else 0.0, out_qmax=oq.qmax if fused else 0.0,
            out_dtype=MQ_F16 if f16 else MQ_F32, w4=plan["w4"], a_tiled_rows=tiled_rows)
        out = out.reshape(*lead, N)
        return _tag_grid(out, oq) if fused else out

    def _count_path(self, reason):
        c = self.__dict__.setdefault("_path_counts", {})
        key = "int8" if reason is None else "simulated: " + reason
        c[key] = c.get(key, 0) + 1

    # -- forward -----------------------------------------------------------------------------------
    def forward(self, input_):
        weight = self.temp_weight if self.use_temporary_parameter else self.weight
        bias = self.temp_bias if self.use_temporary_parameter else self.bias
        weight = self._effective_weight(weight)
        reason = self._int8_reason(input_, weight)
        if reason == "per-group weight grid":          # its own integer kernel (mq_w8a8_linear_grouped); the fused blocks stay away
            reason = self._grouped_reason(input_, weight)
            if reason is None and not self._grouped_plan(weight)["fold_ok"]:
                reason = "per-group weight grid: a group's offset exceeds the kernel's 24-bit fold (narrow range far from zero)"
            if reason is None:
                self._count_path(None)
                return self._forward_int8_grouped(input_, weight, bias)
        if reason is not None:
            self._count_path(reason)
        if reason is None:
            return self._forward_int8(input_, weight, bias)
        input_ = _materialize(input_)
        # simulated path: HIP fake-quant kernels around the library GEMM
        weight = _apply(self.weight_quantizer, weight)
        if self.input_chan_scale is not None:
            input_ = input_ / self.input_chan_scale.to(input_.dtype)
        input_ = _apply(self.input_quantizer, input_)
        out = F.linear(input_, weight, bias=bias)
        return _apply(self.output_quantizer, out)

    @staticmethod
    def from_float(module, input_quant_cfg, weight_quant_cfg, output_quant_cfg):
        kargs = dict(in_features=module.in_features, out_features=module.out_features, bias=module.bias is not None,
                     device=module.weight.device, dtype=module.weight.dtype)
        out = QLinear(kargs, input_quant_cfg, weight_quant_cfg, output_quant_cfg)
        with torch.no_grad():
            out.weight.copy_(module.weight)
            if out.bias is not None:
                out.bias.copy_(module.bias)
        return out

    @staticmethod
    def to_float(module):
        out = nn.Linear(module.in_features, module.out_features, bias=module.bias is not None,
                        device=module.weight.device, dtype=module.weight.dtype)
        with torch.no_grad():
            out.weight.copy_(module.weight)
            if module.bias is not None:
                out.bias.copy_(module.bias)
        return out


class QMatMul(nn.Module, _QuantizedOp):
    """Quantized ``torch.matmul`` of two activations (reference: qmodule.py:408-466)."""

    _slots = (("input", "input_quantizer"), ("input2", "input2_quantizer"), ("output", "output_quantizer"))

    def __init__(self, input_quant_cfg, input2_quant_cfg, output_quant_cfg):
        super().__init__()
        self._init_quantizers(input=input_quant_cfg, input2=input2_quant_cfg, output=output_quant_cfg)

    int8_mode = "auto"          # "off": always the simulated path (HIP fake-quant kernels around the library matmul)

    def _int8_reason(self, x1, x2) -> Optional[str]:
        """None when mq_qmatmul serves this call (one launch, exact integer contraction), else why the simulated path runs."""
        if self.int8_mode == "off":
            return "int8_mode off"
        q1, q2, qo = self.input_quantizer, self.input2_quantizer, self.output_quantizer
        if not _static_per_tensor(q1, 16) or not _static_per_tensor(q2, 8):
            return "input grids: static per-tensor, at most 16 (input) / 8 (input2) bits"
        if qo is not None and not qo.bypassed() and not _static_per_tensor(qo, 16):
            return "output grid: static per-tensor, at most 16 bits (or none)"
        if not (torch.is_tensor(x1) and torch.is_tensor(x2)) or not ops.qmatmul_supported(x1, x2):
            return "operands outside mq_qmatmul (fp32 device tensors with equal leading dims)"
        if _needs_grad(x1, x2, q1.scale, q2.scale, None if qo is None or qo.bypassed() else qo.scale):
            return "gradient required"
        return None

    def forward(self, x1, x2):
        reason = self._int8_reason(x1, x2)
        c = self.__dict__.setdefault("_path_counts", {})
        key = "int8" if reason is None else "simulated: " + reason
        c[key] = c.get(key, 0) + 1
        if reason is None:
            q1, q2, qo = self.input_quantizer, self.input2_quantizer, self.output_quantizer
            if qo is not None and qo.bypassed():
                qo = None
            for q in (q1, q2, qo):
                if q is not None and q.scale.device != x1.device:
                    q.scale.data, q.offset.data = q.scale.to(x1.device), q.offset.to(x1.device)
            grid = lambda q: None if q is None else (q.scale.detach(), q.offset.detach(), q.qmin, q.qmax)      # noqa: E731
            out = ops.qmatmul(_materialize(x1), _materialize(x2), grid(q1), grid(q2), grid(qo))
            return out if qo is None else _tag_grid(out, qo)
        out = torch.matmul(_apply(self.input_quantizer, _materialize(x1)), _apply(self.input2_quantizer, _materialize(x2)))
        return _apply(self.output_quantizer, out)


def _image_only(shape, device, quantizer):
    """Stand-in for an activation that exists only as its int8 image in the shared-activation memo: a 1-element tensor expanded
    to `shape`, tagged with the producer grid.  Integer consumers look the image up by this object and never touch its memory;
    anything else calls _materialize() first."""
    t = _tag_grid(torch.empty(1, dtype=torch.float32, device=device).expand(*shape), quantizer)
    t._mq_image_only = True
    return t


def _materialize(x):
    """fp32 values of an image-only activation: (index - offset) * scale from the memo image -- the arithmetic of the fake-quant
    output itself, so the result is what the producer would have written."""
    if not getattr(x, "_mq_image_only", False):
        return x
    grid = _producer_grid(x)
    K = x.shape[-1]
    M = x.numel() // K
    for tag in ((128, None), (0, None)):
        hit = _shared_activation.get(x, grid, tag)
        if hit is not None:
            q = hit[0].reshape(M, K)
            break
    else:
        for shift in (128, 0):
            hit = _shared_activation.get(x, grid, ("tiled", shift, None))
            if hit is not None:
                break
        if hit is None:
            raise RuntimeError("mobilequant_amd: image-only activation without an image (consumed on another thread / stream?)")
        q = hit[0].view(-1, K // 64, 4, 16, 16).permute(0, 3, 1, 2, 4).reshape(-1, K)[:M]
    y = (q.to(torch.float32) + float(hit[2]) - grid.offset.detach()) * grid.scale.detach()
    return _tag_grid(y.reshape(x.shape), grid)


def _fused_norm(self, input_, weight, bias, layernorm, images=None):
    """QRMSNorm / QLayerNorm.forward as ONE launch (mq_rmsnorm_quant / mq_layernorm_quant) instead of six; with an
    8-bit output grid the int8 indices + row sums are handed to the consumer linears through the shared-activation
    memo, so q/k/v (w1/w3) launch no quantize.  None -> the caller runs the composite ops.
    images = "rowmajor" | "tiled" (the decoder-layer pass, whose consumers are all integer): ONLY that int8 image is written -- no
    fp32 result (16 of the 24 MB the kernel writes at [2048, 2048]); the return value is an _image_only() stand-in."""
    if (self.fused_mode == "off" or weight is None or not input_.is_cuda or input_.dtype != torch.float32
            or weight.dtype != torch.float32 or input_.shape[-1] % 4 or input_.numel() == 0
            or _needs_grad(input_, weight, bias)):
        return None
    gi, go = QRMSNorm._grid_or_none(self.input_quantizer), QRMSNorm._grid_or_none(self.output_quantizer)
    if gi is False or go is False:
        return None
    for g in (gi, go):
        if g is not None and g[0].device != input_.device:
            return None                      # first call after a device move: the composite path migrates the grids
    wq = self.weight_quantizer
    if wq is not None and _needs_grad(getattr(wq, "scale", None), getattr(wq, "offset", None)):
        return None
    # the fake-quantised [dim] weight vector is cached until the weight or its grid changes (the first call
    # also fixes the weight range from the weight itself, qmodule.py:262-277)
    key = (weight.data_ptr(), _ver(weight), None if wq is None or not wq._has_grid() else wq.grid_token(),
           None if wq is None else (wq.enable, wq.lwc, wq.qcfg.bitwidth, wq.qcfg.is_dynamic))
    cached = getattr(self, "_wfq", None)
    if cached is not None and cached[0] == key and key[2] is not None:
        wfq = cached[1]
    else:
        with torch.no_grad():
            wfq = _apply(wq, weight)
        if wq is not None and wq._has_grid() and not wq.lwc and not wq.qcfg.is_dynamic:
            key = (key[0], key[1], wq.grid_token(), key[3])
            self._wfq = (key, wfq)
    emit = go is not None and self.output_quantizer.qcfg.bitwidth <= 8
    # the fragment-blocked copy (for w1 / w3, which the generated-ISA GEMM loop serves) costs 1/4 of the fp32 write;
    # int8_tiled = None decides by shape: prefill-sized inputs whose width fits the layout
    rows = input_.numel() // input_.shape[-1]
    tiled = getattr(self, "int8_tiled", None)
    tiled = emit and (input_.shape[-1] % 128 == 0) and (rows >= 1536 if tiled is None else bool(tiled))
    if images is not None and emit and rows > 8 and (images == "rowmajor" or input_.shape[-1] % 64 == 0):   # (M <= 8: the decode GEMV reads fp32)
        _, q, rs, shift, qt = ops.rmsnorm_quant(input_, wfq, bias, self.eps, gi, go, emit_int8=True, layernorm=layernorm,
                                                emit_tiled=images == "tiled", want_y=False, emit_rowmajor=images == "rowmajor")
        y = _image_only(input_.shape, input_.device, self.output_quantizer)
        if q is not None:
            _shared_activation.put(y, self.output_quantizer, (shift, None), (q, rs, shift))
        if qt is not None:
            _shared_activation.put(y, self.output_quantizer, ("tiled", shift, None), (qt, rs, shift))
        return y
    res = ops.rmsnorm_quant(input_, wfq, bias, self.eps, gi, go, emit_int8=emit, layernorm=layernorm, emit_tiled=tiled)
    if not emit:
        return _tag_grid(res, self.output_quantizer) if go is not None else res
    y, q, rs, shift, qt = res
    _tag_grid(y, self.output_quantizer)
    _shared_activation.put(y, self.output_quantizer, (shift, None), (q, rs, shift))
    if qt is not None:
        _shared_activation.put(y, self.output_quantizer, ("tiled", shift, None), (qt, rs, shift))
    return y


class QRMSNorm(HFRMSNorm, _QuantizedOp):
    """RMSNorm with quantized weight / input / output (reference: qmodule.py:469-576)."""

    _slots = (("input", "input_quantizer"), ("weight", "weight_quantizer"), ("output", "output_quantizer"))

    def __init__(self, kargs, input_quant_cfg, weight_quant_cfg, output_quant_cfg):
        super().__init__(**kargs)
        self.use_temporary_parameter = False
        self._init_quantizers(input=input_quant_cfg, weight=weight_quant_cfg, output=output_quant_cfg)

    fused_mode = "auto"        # "off": always the composite torch ops around the HIP quantizers

    @staticmethod
    def _grid_or_none(q):
        """(scale, offset, qmin, qmax) of a static per-tensor quantizer, None for an absent / bypassed one,
        False when the quantizer cannot be folded into the fused kernel (dynamic, per-channel, LWC, no range yet)."""
        if q is None or q.bypassed():
            return None
        if not _static_per_tensor(q, 16):
            return False
        return (q.scale.detach(), q.offset.detach(), q.qmin, q.qmax)

    def _forward_fused(self, input_, weight, images=None):
        if self.l2norm_as_rmsnorm:
            return None
        return _fused_norm(self, input_, weight, self.bias, layernorm=False, images=images)

    def forward_images(self, input_, layout: str):
        """forward() for callers whose consumers are all integer linears (llama.fuse_decoder_layer): only the int8 image in
        `layout` ("rowmajor" | "tiled") is produced where the fused kernel applies; otherwise the ordinary forward."""
        weight = self.temp_weight if self.use_temporary_parameter else self.weight
        out = self._forward_fused(input_, weight, images=layout)
        return out if out is not None else self.forward(input_)

    def forward(self, input_):
        weight = self.temp_weight if self.use_temporary_parameter else self.weight
        out = self._forward_fused(input_, weight)
        if out is not None:
            return out
        weight = _apply(self.weight_quantizer, weight)
        out = self.forward_impl(_apply(self.input_quantizer, input_), weight, self.bias)
        return _apply(self.output_quantizer, out)

    @staticmethod
    def _kargs(module):
        return dict(dim=len(module.weight), eps=module.eps, device=module.weight.device, dtype=module.weight.dtype,
                    l2norm_as_rmsnorm=module.l2norm_as_rmsnorm)

    @staticmethod
    def from_float(module, input_quant_cfg, weight_quant_cfg, output_quant_cfg):
        out = QRMSNorm(QRMSNorm._kargs(module), input_quant_cfg, weight_quant_cfg, output_quant_cfg)
        with torch.no_grad():
            out.weight.copy_(module.weight)
        return out

    @staticmethod
    def to_float(module):
        out = HFRMSNorm(**QRMSNorm._kargs(module))
        with torch.no_grad():
            out.weight.copy_(module.weight)
        return out


class QLayerNorm(nn.LayerNorm, _QuantizedOp):
    """LayerNorm with quantized weight / input / output (reference: qmodule.py:579-688)."""

    fused_mode = "auto"

    _slots = (("input", "input_quantizer"), ("weight", "weight_quantizer"), ("output", "output_quantizer"))

    def __init__(self, kargs, input_quant_cfg, weight_quant_cfg, output_quant_cfg):
        super().__init__(**kargs)
        self.use_temporary_parameter = False
        self._init_quantizers(input=input_quant_cfg, weight=weight_quant_cfg, output=output_quant_cfg)

    def forward_images(self, input_, layout: str):
        """QRMSNorm.forward_images for the LayerNorm families (StableLM): only the int8 image in `layout` where the fused kernel applies."""
        weight = self.temp_weight if self.use_temporary_parameter else self.weight
        bias = self.temp_bias if self.use_temporary_parameter else self.bias
        out = _fused_norm(self, input_, weight, bias, layernorm=True, images=layout)
        return out if out is not None else self.forward(input_)

    def forward(self, input_):
        weight = self.temp_weight if self.use_temporary_parameter else self.weight
        bias = self.temp_bias if self.use_temporary_parameter else self.bias
        out = _fused_norm(self, input_, weight, bias, layernorm=True)
        if out is not None:
            return out
        weight = _apply(self.weight_quantizer, weight)
        input_ = _apply(self.input_quantizer, input_)
        out = F.layer_norm(input_, input_.shape[-1:], weight=weight, bias=bias, eps=self.eps)
        return _apply(self.output_quantizer, out)

    @staticmethod
    def from_float(module, input_quant_cfg, weight_quant_cfg, output_quant_cfg):
        kargs = dict(normalized_shape=len(module.weight), eps=module.eps, elementwise_affine=module.elementwise_affine,
                     device=module.weight.device, dtype=module.weight.dtype)
        out = QLayerNorm(kargs, input_quant_cfg, weight_quant_cfg, output_quant_cfg)
        with torch.no_grad():
            out.weight.copy_(module.weight)
            if out.bias is not None:
                out.bias.copy_(module.bias)
        return out

    @staticmethod
    def to_float(module):
        out = nn.LayerNorm(len(module.weight), eps=module.eps, elementwise_affine=module.elementwise_affine,
                           bias=module.bias is not None, device=module.weight.device, dtype=module.weight.dtype)
        with torch.no_grad():
            out.weight.copy_(module.weight)
            if module.bias is not None:
                out.bias.copy_(module.bias)
        return out


def _fused_activation(module, x, act, quantizers):
    """QSiLU / QGELU.forward as ONE launch (mq_act_quant) when every quantizer involved is absent or a static
    per-tensor grid already on x's device; None -> the caller runs the composite ops."""
    if module.fused_mode == "off" or not x.is_cuda or x.dtype != torch.float32 or x.numel() == 0 or _needs_grad(x):
        return None
    grids = []
    for q in quantizers:
        g = QRMSNorm._grid_or_none(q)
        if g is False or (g is not None and (g[0].device != x.device or _needs_grad(q.scale, q.offset))):
            return None
        grids.append(g)
    y = ops.act_quant(x, act, *grids)
    return _tag_grid(y, quantizers[-1]) if grids[-1] is not None else y


class QSiLU(nn.Module, _QuantizedOp):
    """x * quant(sigmoid(x)) with quantized output (reference: qmodule.py:691-753)."""

    _slots = (("input", "input_quantizer"), ("input2", "input2_quantizer"), ("output", "output_quantizer"))
    _optional_update = ("input",)

    def __init__(self, input_quant_cfg, input2_quant_cfg, output_quant_cfg):
        super().__init__()
        self._init_quantizers(input=input_quant_cfg, input2=input2_quant_cfg, output=output_quant_cfg)

    fused_mode = "auto"        # "off": composite torch ops around the HIP quantizers

    def forward(self, x):
        y = _fused_activation(self, x, "silu", (self.input_quantizer, self.input2_quantizer, self.output_quantizer))
        if y is not None:
            return y
        x = _apply(self.input_quantizer, x)
        gate = _apply(self.input2_quantizer, torch.sigmoid(x))
        return _apply(self.output_quantizer, x * gate)


class QGELU(nn.Module, _QuantizedOp):
    """GELU with quantized input / output (reference: qmodule.py:756-798)."""

    _slots = (("input", "input_quantizer"), ("output", "output_quantizer"))
    _optional_update = ("input",)

    def __init__(self, input_quant_cfg, output_quant_cfg):
        super().__init__()
        self._init_quantizers(input=input_quant_cfg, output=output_quant_cfg)

    fused_mode = "auto"

    def forward(self, x):
        y = _fused_activation(self, x, "gelu", (self.input_quantizer, None, self.output_quantizer))
        if y is not None:
            return y
        return _apply(self.output_quantizer, F.gelu(_apply(self.input_quantizer, x)))


QuantLinear = QLinear   # the north star's name for the same class


# ------------------------------------------------------------------------------------------------
# model surgery and artefact IO (reference: qmodule.py:835-970)
# ------------------------------------------------------------------------------------------------
_Q_TYPES = (QLinear, QRMSNorm, QLayerNorm, QMatMul, QSiLU, QGELU)
_NO_INPUT_QUANT = ("q_proj", "k_proj", "v_proj", "o_proj", "w1", "w3")


def _type_named(module, *names) -> bool:
    """isinstance by class name too, so the reference's own model classes are recognised when they are
    importable (mobilellm.model.ops.FMatMul, hf_model.HFRMSNorm, transformers' GELU variants)."""
    return any(c.__name__ in names for c in type(module).__mro__)


def create_sim_qmodel(model, default_weight_qcfg=None, default_act_qcfg=None):
    """Swap float leaves for Q-modules by the reference's name rules (qmodule.py:835-865)."""
    wcfg = default_weight_qcfg if default_weight_qcfg is not None else QuantConfig()
    acfg = default_act_qcfg if default_act_qcfg is not None else QuantConfig()
    for name, module in reversed(list(model._modules.items())):
        if "lm_head" in name or ("norm" in name and "layernorm" not in name):
            continue                                   # final norm and predictor stay in floating point
        if isinstance(module, nn.Linear):
            q = QLinear.from_float(module, acfg, wcfg, acfg)
            if any(tag in name for tag in _NO_INPUT_QUANT):
                q.input_quantizer = None               # already quantized by the producing module
            model._modules[name] = q
        elif isinstance(module, FMatMul) or _type_named(module, "FMatMul"):
            model._modules[name] = QMatMul(acfg, acfg, acfg)
        elif isinstance(module, nn.SiLU):
            q = QSiLU(acfg, acfg, acfg)
            q.input_quantizer = None
            model._modules[name] = q
        elif isinstance(module, nn.GELU) or _type_named(module, "GELUActivation", "PytorchGELUTanh"):
            q = QGELU(acfg, acfg)
            q.input_quantizer = None
            model._modules[name] = q
        elif isinstance(module, HFRMSNorm) or _type_named(module, "HFRMSNorm"):
            model._modules[name] = QRMSNorm.from_float(module, acfg, wcfg, acfg)
        elif isinstance(module, nn.LayerNorm):
            model._modules[name] = QLayerNorm.from_float(module, acfg, wcfg, acfg)
        elif len(list(module.children())) > 0:
            create_sim_qmodel(module, wcfg, acfg)
    return model


def create_fp_model(model):
    """Inverse surgery (reference: qmodule.py:889-905)."""
    for name, module in reversed(list(model._modules.items())):
        if isinstance(module, (QLinear, QRMSNorm, QLayerNorm)):
            model._modules[name] = type(module).to_float(module)
        elif isinstance(module, QMatMul):
            model._modules[name] = FMatMul()
        elif isinstance(module, QSiLU):
            model._modules[name] = nn.SiLU()
        elif isinstance(module, QGELU):
            model._modules[name] = nn.GELU()
        elif len(list(module.children())) > 1:
            create_fp_model(module)
    return model


def create_weight_only_qmodel(model, w_qcfg=None):
    """W4A16 packing is the third-party auto_gptq path of the reference (qmodule.py:803-886), which is
    outside the W8A8 / W4A8 hot path this package implements."""
    raise NotImplementedError("weight-only (W4A16) packing relies on auto_gptq in the reference and is out of scope "
                              "here; use the W4A8 path (QLinear with a 4-bit weight quantizer)")


def _minmax_entry(q: Quantizer):
    lo, hi = compute_min_max_from_scale_offset(q.scale, q.offset, q.qcfg.bitwidth, q.qcfg.is_symmetric)
    return [lo.item(), hi.item()]


def export_act_range(model):
    """Ranges implied by the current activation grids (reference: qmodule.py:908-937)."""
    act_dict = {}
    for name, m in model.named_modules():
        if not isinstance(m, _Q_TYPES):
            continue
        roles = ("input", "input2", "output") if isinstance(m, (QMatMul, QSiLU)) else ("input", "output")
        entry = act_dict.get(name, {})
        for role in roles:
            q = getattr(m, role + "_quantizer", None)
            if q is not None:
                entry[role] = _minmax_entry(q)
        act_dict[name] = entry
    return act_dict


def update_qcfg(model, override_qcfg):
    for name, module in model.named_modules():
        if not isinstance(module, _Q_TYPES):
            continue
        assert name in override_qcfg
        cfg = override_qcfg[name]
        if isinstance(module, (QLinear, QRMSNorm, QLayerNorm)):
            module.update_qcfg(cfg.get("input", None), cfg["weight"], cfg["output"])
        elif isinstance(module, QMatMul):
            module.update_qcfg(cfg["input"], cfg["input2"], cfg["output"])
        elif isinstance(module, QSiLU):
            module.update_qcfg(cfg.get("input", None), cfg["input2"], cfg["output"])
        else:
            module.update_qcfg(cfg.get("input", None), cfg["output"])
    return model


def export_qcfg(model):
    return {name: m.export_qcfg() for name, m in model.named_modules() if isinstance(m, _Q_TYPES)}


def set_scale_and_offset(model, act_dict, use_scale_offset_as="buffer"):
    for name, module in model.named_modules():
        if isinstance(module, _Q_TYPES):
            assert name in act_dict
            module.set_scale_offset(act_dict[name], use_scale_offset_as)
    return model


def _u8_grid(q: Optional[Quantizer]) -> bool:
    return _static_per_tensor(q, 8) and q.qmin == 0 and q.qmax == 255


def _gated_mlp_forward(self, x, resid=None):
    """``w2(act_fn(w1(x)) * w3(x))`` (hf_model.py:1057) on the integer chain, in four launches and without a single fp32
    intermediate: [int8 image of x -- usually left by the norm] -> ONE pair GEMM writing the 8-bit output indices of w1 and
    w3 -> ONE gated-activation kernel (dequantise the indices, QSiLU / QGELU, product, w2's input quantizer -> int8 + row sums)
    -> w2's GEMM.  Bit-identical to the chain of modules on their integer paths; anything the chain cannot serve falls back to
    the chain itself.  resid (optional, the output's shape): returns resid + mlp(x) with the add inside w2's GEMM store (used by
    llama.fuse_decoder_layer)."""
    w1, w2, w3, act = self.w1, self.w2, self.w3, self.act_fn
    plain0 = self._mq_plain_forward

    def plain(t):
        out = plain0(_materialize(t))
        return out if resid is None else resid + out

    def weight_of(m):
        return m._effective_weight(m.temp_weight if m.use_temporary_parameter else m.weight)
    if (getattr(self, "fused_mode", "auto") == "off" or not x.is_cuda or x.dtype != torch.float32
            or any(m.use_temporary_parameter or m.input_chan_scale is not None for m in (w1, w2, w3))):
        return plain(x)
    wt1, wt3, wt2 = weight_of(w1), weight_of(w3), weight_of(w2)
    if not (w1._int8_ready(x, wt1) and w3._int8_ready(x, wt3) and wt1.shape == wt3.shape):
        return plain(x)
    M, (N, K) = x.numel() // x.shape[-1], wt1.shape
    if not (_u8_grid(w1.output_quantizer) and _u8_grid(w3.output_quantizer) and _u8_grid(w2.input_quantizer) and N % 16 == 0 and M > 8):
        return plain(x)
    # ONE pair launch of the free-running kernel where it applies (int8 weights, N % 176 == 0, >= 192 tiles); every other shape or a
    # 4-bit weight (Gemma's FFN, small batches, W4A8 recipes) runs w1 and w3 as two GEMMs writing indices -- the rest of the chain is
    # the same
    pair = ops.gemm_tiled_supported(M, N, K) and K >= 768 and K % 256 == 0
    silu = isinstance(act, QSiLU)
    act_in = act.input_quantizer
    if act_in is not None and not act_in.bypassed():
        return plain(x)                     # (the surgery leaves QSiLU / QGELU without an input quantizer, qmodule.py:855,858)
    mid = QRMSNorm._grid_or_none(act.input2_quantizer) if silu else None
    aout = QRMSNorm._grid_or_none(act.output_quantizer)
    if mid is False or aout is False or act.fused_mode == "off":
        return plain(x)
    g1, g3 = w1._activation_grid(x), w3._activation_grid(x)
    # a DYNAMIC own input quantizer has no grid before its first refresh and a stale one afterwards: the module chain re-derives it
    # per forward (_input_image), the fused block does not -- it stays on the chain
    if (any(_dynamic_per_tensor(m.input_quantizer, 8) for m in (w1, w3)) or g1 is None or g3 is None
            or g1.grid_token() != g3.grid_token()):
        return plain(x)
    # w2: what _int8_ready would check on the product tensor, which never exists here
    wq2, oq2 = w2.weight_quantizer, w2.output_quantizer
    if (w2.int8_mode == "off" or wq2 is None or wq2.bypassed() or wq2.qcfg.bitwidth > 8 or wq2.qcfg.is_dynamic or wq2.lwc
            or (wq2.qcfg.is_per_channel and wq2.qcfg.group_size != -1) or wt2.shape[1] % 128 or wt2.shape[0] % 4
            or (oq2 is not None and not oq2.bypassed() and not _static_per_tensor(oq2, 16))
            or _needs_grad(wt2, w2.bias, getattr(wq2, "scale", None))):
        return plain(x)
    any_w4 = w1._weight_plan(wt1)["w4"] or w3._weight_plan(wt3)["w4"]
    both_w4 = w1._weight_plan(wt1)["w4"] and w3._weight_plan(wt3)["w4"] and ops.gemm_tiled_w4_supported(M, N, K)
    if any_w4:
        pair = False
    t_hit = None
    if (not pair and not any_w4 and M > 8 and ops.gemm_tiled128_supported(M, N, K)) or both_w4:
        # N does not tile by 176 (Gemma: 16384): w1 / w3 run one by one on the 128-column generated kernel, reading the
        # fragment-blocked image the norm left (index outputs only -- which is all this block needs)
        t_hit = _shared_activation.get(x, g1, ("tiled", 128 if g1.qmax > 127 else 0, None))
    if t_hit is not None:
        grid, (a_q, a_rs, a_shift), tiled_rows, decode = g1, t_hit, M, False
    else:
        grid, a_q, a_rs, a_shift, tiled_rows, decode = w1._input_image(x, wt1)
    if decode:
        return plain(x)
    if any_w4 and tiled_rows is not None and not both_w4:
        return plain(x)                     # (a 4-bit sibling of an int8 linear that chose the fragment-blocked image: not a real recipe)
    pair = pair and tiled_rows is not None and not any_w4
    halves = []
    for m, wt in ((w1, wt1), (w3, wt3)):
        plan = m._epilogue_vectors(m._weight_plan(wt), grid, a_shift, K)
        oq = m.output_quantizer
        if oq.scale.device != x.device:
            oq.scale.data, oq.offset.data = oq.scale.to(x.device), oq.offset.to(x.device)
        halves.append(dict(w=plan["w"], alpha=plan["alpha"], w_zp=plan["w_zp"], col_term=plan["col_term"],
                           bias=m.temp_bias if m.use_temporary_parameter else m.bias, out_scale=oq.scale.detach(),
                           out_offset=oq.offset.detach()))
    iq2 = w2.input_quantizer
    # everything static and the pair shape: w1, then w3 with the gate in its epilogue (table lookup -> w2's fragment-blocked image):
    # no index tensor of w3, no lookup launch
    narrow = (not pair) and t_hit is not None and ((M + 255) // 256) * (N // 128) >= 192     # 256 x 128 tiles (Gemma's FFN width)
    gate_fused = ((pair or narrow) and N % 64 == 0 and getattr(self, "gated_table", True) and getattr(self, "gated_epilogue", True)
                  and resid is not None and resid.dtype == torch.float32 and resid.is_contiguous()
                  and (both_w4 or not any_w4) and w2._tiled_residual_ok(M, wt2.shape[0], N))
    if gate_fused:              # (packed-only 4-bit weights: the packed forms of both kernels, mq_w4a8_linear_tiled_gated)
        table = _gated_table_of(self, act, silu, w1.output_quantizer, w3.output_quantizer, iq2, x.device)
        p_q, p_rs = ops.int8_linear_gated(a_q, M, a_rs, halves[0], halves[1], table, w4=both_w4)
        return w2._int8_from_image(None, wt2, w2.temp_bias if w2.use_temporary_parameter else w2.bias, iq2, p_q, p_rs, 128, M,
                                   lead_shape=x.shape[:-1], resid=resid)
    if pair:
        a_idx, b_idx = ops.int8_linear_pair(a_q, M, a_rs, halves[0], halves[1], out_dtype=MQ_U8)
    elif both_w4 and tiled_rows is not None:        # packed weights on the generated kernels (mq_w4a8_linear_tiled), indices out
        a_idx, b_idx = (ops.w4a8_linear_tiled(a_q, M, h["w"], a_rs, h["alpha"], h["w_zp"], h["col_term"], h["bias"],
                                              [(h["out_scale"], h["out_offset"])]) for h in halves)
    else:
        a_idx, b_idx = (ops.int8_linear(a_q, h["w"], a_rs, h["alpha"], h["w_zp"], h["col_term"], h["bias"], out_scale=h["out_scale"],
                                        out_offset=h["out_offset"], out_qmin=0.0, out_qmax=255.0, out_dtype=MQ_U8, w4=m._weight_plan(wt)["w4"],
                                        a_tiled_rows=tiled_rows)
                        for h, (m, wt) in zip(halves, ((w1, wt1), (w3, wt3))))
    for q in (iq2, act.output_quantizer, act.input2_quantizer if silu else None):
        if q is not None and not q.bypassed() and q.scale.device != x.device:
            q.scale.data, q.offset.data = q.scale.to(x.device), q.offset.to(x.device)
    o1, o3 = w1.output_quantizer, w3.output_quantizer
    if N % 8 == 0 and getattr(self, "gated_table", True):
        # static grids: act(a) * b -> index is a function of the two 8-bit indices -- a 64 KiB table built once per set of grids
        cached = (None, _gated_table_of(self, act, silu, o1, o3, iq2, x.device))
        # w2 with the residual: the fragment-blocked image its generated kernel reads
        w2_tiled = (resid is not None and resid.dtype == torch.float32 and resid.is_contiguous() and N % 64 == 0
                    and w2._tiled_residual_ok(M, wt2.shape[0], N))
        p_q, p_rs = ops.gated_lookup(a_idx, b_idx, cached[1], tiled=w2_tiled)
        if w2_tiled:
            return w2._int8_from_image(None, wt2, w2.temp_bias if w2.use_temporary_parameter else w2.bias, iq2, p_q, p_rs, 128, M,
                                       lead_shape=x.shape[:-1], resid=resid)
    else:
        p_q, p_rs = ops.gated_act_quant(a_idx, b_idx, "silu" if silu else "gelu",
                                        (iq2.scale.detach(), iq2.offset.detach(), iq2.qmin, iq2.qmax),
                                        a_grid=(o1.scale.detach(), o1.offset.detach()), b_grid=(o3.scale.detach(), o3.offset.detach()),
                                        mid_grid=QRMSNorm._grid_or_none(act.input2_quantizer) if silu else None,
                                        act_grid=QRMSNorm._grid_or_none(act.output_quantizer), q_shift=128)
    return w2._int8_from_image(None, wt2, w2.temp_bias if w2.use_temporary_parameter else w2.bias, iq2, p_q.view(M, N), p_rs, 128,
                               None, lead_shape=x.shape[:-1], resid=resid)


def _gated_table_of(block, act, silu, o1, o3, iq2, device):
    """The 256 x 256 table (ia, ib) -> w2 input index of this block's grids (ops.gated_table), cached on the block."""
    for q in (iq2, act.output_quantizer, act.input2_quantizer if silu else None):
        if q is not None and not q.bypassed() and q.scale.device != device:
            q.scale.data, q.offset.data = q.scale.to(device), q.offset.to(device)
    mid_q = act.input2_quantizer if silu else None
    key = ("silu" if silu else "gelu",) + tuple(None if q is None or q.bypassed() else q.grid_token()
                                                for q in (o1, o3, mid_q, act.output_quantizer, iq2))
    cached = getattr(block, "_gated_lut", None)
    if cached is None or cached[0] != key or cached[1].device != device:
        table = ops.gated_table(key[0], (iq2.scale.detach(), iq2.offset.detach(), iq2.qmin, iq2.qmax),
                                (o1.scale.detach(), o1.offset.detach()), (o3.scale.detach(), o3.offset.detach()),
                                mid_grid=QRMSNorm._grid_or_none(mid_q) if silu else None,
                                act_grid=QRMSNorm._grid_or_none(act.output_quantizer), q_shift=128)
        cached = block._gated_lut = (key, table)
    return cached[1]


def int8_coverage(model, reset=False):
    """Which QLinear / QMatMul modules of `model` ran on the integer kernels, and which took the simulated path (HIP fake-quant around the fp32
    library GEMM / bmm) and why -- since the last reset.  Returns {"modules": {name: {path: calls}}, "int8_calls", "simulated_calls",
    "simulated_modules": [names], "summary": str}.  The simulated path is correct by the reference's semantics but ~10x slower; an
    evaluation that silently lands on it (a 16-bit producer in front of q/k/v, a per-group recipe, grad mode) shows up here.
    Modules executed inside a fused block (fuse_gated_mlp / fuse_attention / fuse_decoder_layer) count as "int8 (fused block)"."""
    mods, n_int, n_sim, sim_names = {}, 0, 0, []
    for name, mod in model.named_modules():
        if isinstance(mod, (QLinear, QMatMul)):
            c = dict(mod.__dict__.get("_path_counts", {}))
            mods[name] = c
            i = sum(v for k, v in c.items() if k.startswith("int8"))
            sm = sum(v for k, v in c.items() if k.startswith("simulated"))
            n_int, n_sim = n_int + i, n_sim + sm
            if sm:
                sim_names.append(name)
            if reset:
                mod.__dict__["_path_counts"] = {}
    why = {}
    for name in sim_names:
        for k, v in mods[name].items():
            if k.startswith("simulated"):
                why[k] = why.get(k, 0) + v
    summary = f"{n_int} integer-path calls, {n_sim} simulated-path calls in {len(sim_names)} of {len(mods)} QLinear / QMatMul modules"
    if why:
        summary += "; " + "; ".join(f"{v} x {k}" for k, v in sorted(why.items(), key=lambda kv: -kv[1]))
    return {"modules": mods, "int8_calls": n_int, "simulated_calls": n_sim, "simulated_modules": sim_names, "summary": summary}


def fuse_gated_mlp(model) -> int:
    """Graph pass (like wire_integer_inputs, no counterpart in the reference): every block with QLinear children w1 / w2 / w3
    and a QSiLU / QGELU child act_fn -- the reference's HFMLP after create_sim_qmodel (hf_model.py:1042-1062) -- gets the
    integer-chain forward above.  The block keeps its children and parameters; `block.fused_mode = "off"` restores the chain.
    Returns the number of blocks fused."""
    import types
    n = 0
    for _, m in model.named_modules():
        kids = dict(m.named_children())
        if (all(isinstance(kids.get(k), QLinear) for k in ("w1", "w2", "w3")) and isinstance(kids.get("act_fn"), (QSiLU, QGELU))
                and not hasattr(m, "_mq_plain_forward")):
            m._mq_plain_forward = m.forward
            m.forward = types.MethodType(_gated_mlp_forward, m)
            n += 1
    return n


def wire_integer_inputs(model, act_bitwidth=8, act_is_symmetric=False):
    """Graph pass the integer path needs and the reference never did (hard parts): every QLinear
    without an input quantizer is told the grid its input sits on.  At run time the LIVE grid of the producing quantizer
    (the tag Quantizer.forward / the fused kernels leave on their output tensor) always wins -- trained scales, changed
    configs and 16-bit producers are seen as they are.  This pass only supplies the DECLARED fallback for inputs whose tag
    is lost between producer and consumer (o_proj: pv_bmm -> transpose -> reshape): the calibrated range of the linear's
    own input (``act_dict[name]['input']``, kept by set_scale_offset) on the activation bitwidth.  A declared grid is a
    promise by the caller; a model whose producers were re-trained must be re-wired (or left to the tags / the simulated
    path).  Returns the number of linears wired."""
    n = 0
    for _, m in model.named_modules():
        if isinstance(m, QLinear) and m.input_quantizer is None and "input" in getattr(m, "_act_range", {}):
            lo, hi = m._act_range["input"]
            m.set_input_grid(lo, hi, act_bitwidth, act_is_symmetric)
            n += 1
    return n

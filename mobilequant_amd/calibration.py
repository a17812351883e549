"""Activation-range calibration: MI355X-native counterpart of ``ptq/generate_act_range.py`` and of the
absmax half of ``ptq/generate_act_scale_shift.py``.

The reference (generate_act_range.py:49-122) runs the model sample by sample and, in a forward hook
on every Linear / Norm / SiLU / GELU / Softmax / MatMul leaf, computes ``x.min().item()`` and
``x.max().item()``: two reduction passes and two host syncs per tensor, ~576 tensors per sample,
one process.  Here:

  * each hooked tensor is reduced by ONE single-pass HIP kernel (``mq_minmax_tensor`` /
    ``mq_minmax_cols``) straight into a device-resident running statistic -- no host sync in the loop;
  * samples are sharded round-robin over the ranks of a ``torch.distributed`` group (one process per
    GPU); min/max are exact and order independent, so any sharding gives bit-identical statistics;
  * at the end every rank packs ``[-min..., max...]`` of all tensors into one flat fp32 buffer and
    the group performs ONE all-reduce(MAX) (RCCL over xGMI on GPUs; payload is KBs for per-tensor
    mode, ~10 MB for per-channel mode: latency bound, so one collective, not one per tensor);
  * artefacts keep the reference formats: ``act_dict.json`` (indent 4, sorted keys;
    mobilellm/utils/io.py:34-36), ``act_dict_per_channel.pth``, ``act_scales.pth`` keys
    ``"<module>_<input|output>"``.
"""
from __future__ import annotations

import json
from contextlib import nullcontext as _nullcontext
from collections import OrderedDict
from typing import Dict, Iterable, List, Optional, Sequence

import torch
import torch.distributed as dist
import torch.nn as nn

from . import ops
from .quantization.fp_ops import FMatMul, HFRMSNorm

_HOOKED_TYPE_NAMES = ("GELUActivation", "NewGELUActivation", "PytorchGELUTanh", "HFRMSNorm", "FMatMul")


def is_calibrated_leaf(name: str, m: nn.Module) -> bool:
    """The reference's hook set (generate_act_range.py:93-95)."""
    if isinstance(m, (nn.Linear, nn.SiLU, nn.GELU, nn.Softmax, nn.LayerNorm, HFRMSNorm, FMatMul)):
        return True                 # (nn.GELU: this package's graphs carry it where the reference's carry transformers' GELUActivation)
    if any(c.__name__ in _HOOKED_TYPE_NAMES for c in type(m).__mro__):
        return True
    return "attn_quantizer" in name or "softmax_quantizer" in name


def _is_matmul(m: nn.Module) -> bool:
    return isinstance(m, FMatMul) or any(c.__name__ == "FMatMul" for c in type(m).__mro__)


class ActRangeCollector:
    """Running min/max of the input / output (/ input2) of every calibrated leaf of ``model``.

    Slots are laid out from ``model.named_modules()`` order, not from the order tensors happen to be
    seen, so every rank has the same layout even if it processed no sample.
    """

    def __init__(self, model: nn.Module, per_channel: bool = False, device=None):
        self.model = model
        self.per_channel = per_channel
        self.device = torch.device(device) if device is not None else next(model.parameters()).device
        self.slots: "OrderedDict[tuple, int]" = OrderedDict()
        for name, m in model.named_modules():
            if is_calibrated_leaf(name, m):
                for field in (("input", "output", "input2") if _is_matmul(m) else ("input", "output")):
                    self.slots[(name, field)] = len(self.slots)
        n = len(self.slots)
        if per_channel:
            self._pc: List[Optional[tuple]] = [None] * n      # lazily sized (min[C], max[C]) per slot
        else:
            self._mn = torch.full((n,), float("inf"), dtype=torch.float32, device=self.device)
            self._mx = torch.full((n,), float("-inf"), dtype=torch.float32, device=self.device)
        self._hooks = []
        self._aware = []               # attention blocks that take their score-chain statistics from attention_probs()
        # One tensor hooked under several names (round 6; per-tensor mode): in a decoder layer the norm's output IS the input of q_proj,
        # k_proj and v_proj, w1's output IS the activation's input, pv_bmm's output holds the values of o_proj's input, ... -- the
        # reference reduces each once per hook (generate_act_range.py:76-89).  A graph that knows these identities declares them
        # (model.calibration_alias_groups(): lists of (module name, field), llama.LlamaForCausalLM); the FIRST forward pass still reduces
        # every slot and the declared groups are compared (one host read at its end): where all members agree bit for bit the later
        # passes reduce the first member only and the others mirror it -- the same numbers, 7 of ~25 reductions per layer fewer.
        self.mirror_declared_aliases = True
        self._mirror: Dict[int, int] = {}              # slot -> the slot whose statistic it mirrors (verified groups)
        self._mirror_pending: List[List[int]] = []     # declared groups, not yet compared
        groups = getattr(model, "calibration_alias_groups", None)
        if callable(groups):                                        # (per-channel mode: a group survives the check only if its members are one tensor)
            for grp in groups():
                idx = [self.slots[k] for k in grp if k in self.slots]
                if len(idx) == len(grp) and len(idx) > 1:
                    self._mirror_pending.append(idx)
        self._probs_buf: Dict[tuple, torch.Tensor] = {}        # attention_probs(): the probabilities' buffer whose upper triangles stay zero
        self._forced = set()           # mirrors set by a fused pass itself (the module whose hook would fill the slot is not run)
        self._layers = []              # decoder layers / MLPs that take their glue statistics from norm_pass / gated_pass
        self.bytes_aliased = 0         # hooked bytes NOT read again: their slot mirrors another one
        self.bytes_seen = 0            # bytes of hooked tensors reduced so far (host-side bookkeeping for the benchmark)
        self.bytes_fused = 0           # bytes of tensors whose statistics were folded into the pass that produced them
        self.n_collectives = 0

    # -- hooks -------------------------------------------------------------------------------------
    def _update(self, name: str, field: str, t: torch.Tensor) -> None:
        i = self.slots[(name, field)]
        t = t.detach()
        if i in self._mirror:
            if i not in self._forced:
                self.bytes_aliased += t.numel() * t.element_size()
                return
            # a fused pass used to vouch for this slot and skip its hook; this time the hook ran: the slot takes over the mirrored history
            # (equal to its own so far) and goes on alone
            j = self._mirror.pop(i)
            self._forced.discard(i)
            if not self.per_channel:
                self._mn[i:i + 1].copy_(self._mn[j:j + 1])
                self._mx[i:i + 1].copy_(self._mx[j:j + 1])
        self.bytes_seen += t.numel() * t.element_size()
        if t.device != self.device:
            raise RuntimeError(f"calibration of {name}.{field}: tensor on {t.device}, statistics on {self.device} -- run one "
                               "process per GPU (the data-parallel path) instead of one model sharded over devices")
        if self.per_channel:
            t2 = t.reshape(-1, t.shape[-1])
            if self._pc[i] is None:
                self._pc[i] = ops.minmax_new(t2.shape[1], t.device)
            mn, mx = self._pc[i]
            if mn.numel() != t2.shape[1]:
                raise RuntimeError(f"per-channel calibration of {name}.{field}: channel count changed from "
                                   f"{mn.numel()} to {t2.shape[1]} (the reference has the same restriction: "
                                   "use a fixed sequence length)")
            ops.minmax_cols_(t2, mn, mx)
        else:
            # layout does not matter for a per-tensor statistic: reduce the dense storage in place
            if not t.is_contiguous():
                p = t.permute(sorted(range(t.dim()), key=lambda d: -t.stride(d)))
                t = p if p.is_contiguous() else t.contiguous()      # transposed views (k^T in qk_bmm): no copy
            ops.minmax_tensor_(t, self._mn[i:i + 1], self._mx[i:i + 1])

    def _hook(self, name: str, matmul: bool):
        def fn(m, xx, yy):
            x = xx[0] if isinstance(xx, tuple) else xx
            y = yy[0] if isinstance(yy, tuple) else yy
            owned = m.__dict__.get("_mq_calib_skip")             # (collector, fields) whose statistic the fused score chain takes (below)
            skip = owned[1] if owned is not None and owned[0] is self else ()
            if "input" not in skip:
                self._update(name, "input", x)
            if "output" not in skip:
                self._update(name, "output", y)
            if matmul and "input2" not in skip:
                self._update(name, "input2", xx[1])
        return fn

    # statistics where the tensors are PRODUCED (round 5): an attention block of this package's leaf graph (llama.Attention) asks the
    # collector for its score chain -- raw scores -> / sqrt(d) -> + mask -> softmax -- and gets the probabilities back with
    # qk_bmm.output's and pv_bmm.input's statistics already folded in (ops.calib_attention_probs_: one pass over [heads, S, S] instead
    # of two hook reductions and five elementwise passes).  Per-tensor mode only; any other graph keeps the hooks.
    fuse_attention_statistics = True

    # a causal mask (llama.LlamaForCausalLM tags the one it builds): the probabilities above the diagonal are zeros -- they go to ONE buffer
    # per shape that keeps them from call to call, so the pass stores the lower triangles only (round 6: 200 -> ~150 us at [32, 2048, 2048])
    keep_causal_zeros = True

    def attention_probs(self, qk_name: str, pv_name: str, raw: torch.Tensor, mask, sqrt_d: float) -> torch.Tensor:
        i, j = self.slots[(qk_name, "output")], self.slots[(pv_name, "input")]
        self.bytes_fused += 2 * raw.numel() * raw.element_size()
        st = (self._mn[i:i + 1], self._mx[i:i + 1], self._mn[j:j + 1], self._mx[j:j + 1])
        if (self.keep_causal_zeros and mask is not None and getattr(mask, "_mq_causal", False) and raw.shape[-1] == raw.shape[-2]
                and tuple(mask.shape) == tuple(raw.shape[-2:])):
            key = (tuple(raw.shape), raw.device)
            buf = self._probs_buf.get(key)
            if buf is None:
                self._probs_buf.clear()                             # one shape at a time (a calibration set of mixed lengths re-zeroes)
                buf = self._probs_buf[key] = torch.zeros_like(raw)
            return ops.calib_attention_probs_causal_(raw, buf, sqrt_d, False, *st)
        return ops.calib_attention_probs_(raw, mask, sqrt_d, *st)

    # ... and the glue between the linears (round 6): a norm with both of its statistics in one pass (optionally with the residual add
    # in front of it), and act(w1(x)) * w3(x) with the four statistics around it (ops.calib_norm_ / calib_gated_).  llama.DecoderLayer /
    # MLP ask for them while ONE collector is attached (per-tensor mode); the hooks of the modules involved skip what the pass took.
    fuse_layer_statistics = True

    def can_fuse_layer(self, x: torch.Tensor) -> bool:
        return (self.fuse_layer_statistics and not self.per_channel and x.dtype == torch.float32 and x.device == self.device
                and x.device.type == "cuda" and x.is_contiguous() and x.shape[-1] % 4 == 0 and x.shape[-1] <= 8192)

    def norm_pass(self, name: str, module: nn.Module, x: torch.Tensor, delta: Optional[torch.Tensor] = None, delta_slot: Optional[tuple] = None):
        """(h, y): h = x (+ delta), y = module(h); module's input / output statistics are taken here (its hooks are not run).
        delta_slot = (module name, field): the branch's own statistic is taken in the same pass (the hook that would have read it is
        told to skip the field by the caller)."""
        i, j = self.slots[(name, "input")], self.slots[(name, "output")]
        nbytes = x.numel() * x.element_size()
        self.bytes_fused += 2 * nbytes
        dm = dx = None
        if delta is not None and delta_slot is not None:
            k = self.slots[delta_slot]
            dm, dx = self._mn[k:k + 1], self._mx[k:k + 1]
            self.bytes_fused += nbytes
        return ops.calib_norm_(x, delta, module.weight, getattr(module, "bias", None), module.eps, type(module) is nn.LayerNorm,
                               self._mn[i:i + 1], self._mx[i:i + 1], self._mn[j:j + 1], self._mx[j:j + 1], dm, dx)

    @staticmethod
    def norm_is_plain(module: nn.Module) -> bool:
        """A norm the one-pass kernel reproduces: nn.LayerNorm over the last dim with affine parameters, or HFRMSNorm in its rsqrt form
        without a bias."""
        if type(module) is nn.LayerNorm:                             # (exact types: QLayerNorm / QRMSNorm carry quantizers of their own)
            return module.elementwise_affine and module.weight is not None and len(module.normalized_shape) == 1
        return type(module) is HFRMSNorm and not module.l2norm_as_rmsnorm and module.bias is None

    def gated_pass(self, w1_name: str, w3_name: str, act_name: str, w2_name: str, a: torch.Tensor, b: torch.Tensor, act: str) -> torch.Tensor:
        """act(a) * b with the statistics of w1.output (= act.input), act.output, w3.output and w2.input."""
        sl = [self.slots[(w1_name, "output")], self.slots[(act_name, "output")], self.slots[(w3_name, "output")], self.slots[(w2_name, "input")]]
        self.bytes_fused += 4 * a.numel() * a.element_size()
        k = self.slots[(act_name, "input")]
        self._mirror[k] = sl[0]                                        # the activation module is not run: its input IS w1's output
        self._forced.add(k)
        stats = []
        for k in sl:
            stats += [self._mn[k:k + 1], self._mx[k:k + 1]]
        return ops.calib_gated_(a, b, act, stats)

    def rope_pass(self, q_name: str, k_name: str, qk_name: str, q_lin: torch.Tensor, k_lin: torch.Tensor, heads: int, kv_heads: int, head_dim: int,
                  cos: torch.Tensor, sin: torch.Tensor):
        """(q, k) rotated, [B, heads, S, D] / [B, kv_heads, S, D], with the statistics of q_proj.output, qk_bmm.input, k_proj.output and
        qk_bmm.input2 (the repeated k^T holds k's values) taken in the same pass (ops.calib_rope_)."""
        sl = [self.slots[(q_name, "output")], self.slots[(qk_name, "input")], self.slots[(k_name, "output")], self.slots[(qk_name, "input2")]]
        self.bytes_fused += 2 * (q_lin.numel() + k_lin.numel()) * q_lin.element_size()
        stats = []
        for k in sl:
            stats += [self._mn[k:k + 1], self._mx[k:k + 1]]
        return ops.calib_rope_(q_lin, k_lin, heads, kv_heads, head_dim, cos, sin, stats)

    def rope_qkv_pass(self, q_name: str, k_name: str, v_name: str, qk_name: str, q_lin, k_lin, v_lin, heads: int, kv_heads: int, head_dim: int, cos, sin):
        """rope_pass with v carried along and repeat_kv inside: (q, k, v) all [B, heads, S, D]; + the statistic of v_proj.output."""
        sl = [self.slots[(q_name, "output")], self.slots[(qk_name, "input")], self.slots[(k_name, "output")], self.slots[(qk_name, "input2")],
              self.slots[(v_name, "output")]]
        self.bytes_fused += (2 * q_lin.numel() + 2 * k_lin.numel() + v_lin.numel()) * q_lin.element_size()
        stats = []
        for k in sl:
            stats += [self._mn[k:k + 1], self._mx[k:k + 1]]
        return ops.calib_rope_qkv_(q_lin, k_lin, v_lin, heads, kv_heads, head_dim, cos, sin, stats)

    def mirror_values(self, name: str, field: str, src_name: str, src_field: str) -> bool:
        """The caller vouches that (name, field)'s tensor holds exactly the values of (src_name, src_field)'s in this pass and skips the
        hook: the slot mirrors the source from now on (per-tensor mode).  False: keep the hook."""
        if self.per_channel or (name, field) not in self.slots or (src_name, src_field) not in self.slots:
            return False
        k = self.slots[(name, field)]
        self._mirror[k] = self.slots[(src_name, src_field)]
        self._forced.add(k)
        return True

    def can_fuse_attention(self, raw_shape, dtype, device, mask) -> bool:
        return (self.fuse_attention_statistics and not self.per_channel and dtype == torch.float32 and device == self.device
                and device.type == "cuda" and raw_shape[-1] % 4 == 0 and raw_shape[-1] <= 4096
                and (mask is None or (mask.dim() == 2 and mask.dtype == torch.float32 and mask.is_contiguous()
                                      and tuple(mask.shape) == tuple(raw_shape[-2:]))))

    def _pass_end(self, *_):
        """End of a forward pass of the whole model: compare the declared alias groups once (first pass), then mirror the verified ones."""
        if not self._mirror_pending:
            return
        pending, self._mirror_pending = self._mirror_pending, []
        if not self.mirror_declared_aliases:
            return
        if self.per_channel:
            for idx in pending:
                r = self._pc[idx[0]]
                if r is not None and all(self._pc[i] is not None and self._pc[i][0].shape == r[0].shape and torch.equal(self._pc[i][0], r[0])
                                         and torch.equal(self._pc[i][1], r[1]) for i in idx[1:]):
                    for i in idx[1:]:
                        self._mirror[i] = idx[0]
            return
        mn, mx = self._mn.tolist(), self._mx.tolist()
        for idx in pending:
            root = idx[0]
            if mn[root] <= mx[root] and all(mn[i] == mn[root] and mx[i] == mx[root] for i in idx[1:]):
                for i in idx[1:]:
                    self._mirror[i] = root

    def _resolve(self) -> None:
        """Mirrored slots read the statistic of the slot that took their reductions."""
        if not self._mirror:
            return
        if self.per_channel:
            for i, j in self._mirror.items():
                if self._pc[j] is not None:
                    self._pc[i] = (self._pc[j][0].clone(), self._pc[j][1].clone())
            return
        ii = torch.tensor(list(self._mirror.keys()), device=self._mn.device)
        jj = torch.tensor(list(self._mirror.values()), device=self._mn.device)
        self._mn[ii] = self._mn[jj]
        self._mx[ii] = self._mx[jj]

    def attach(self) -> "ActRangeCollector":
        names = {id(m): name for name, m in self.model.named_modules()}
        if self._mirror_pending:
            self._hooks.append(self.model.register_forward_hook(self._pass_end))
        for name, m in self.model.named_modules():
            if is_calibrated_leaf(name, m):
                self._hooks.append(m.register_forward_hook(self._hook(name, _is_matmul(m))))
            qk, pv = getattr(m, "qk_bmm", None), getattr(m, "pv_bmm", None)
            if (getattr(m, "_mq_calibration_aware", False) and qk is not None and pv is not None
                    and (names.get(id(qk)), "output") in self.slots and (names.get(id(pv)), "input") in self.slots):
                if m.__dict__.get("_mq_calib") is not None and m.__dict__["_mq_calib"][0] is not self:
                    # ONE slot per module: a second collector would take the fused statistics away from the first (whose
                    # hooks skip these fields) -- it keeps the plain hooks instead (the statistics agree within a few ulp: the fused
                    # softmax is not torch's bit for bit)
                    continue
                m._mq_calib = (self, names[id(qk)], names[id(pv)])
                self._aware.append(m)
            want = getattr(m, "_mq_calibration_layer_parts", None)     # llama.DecoderLayer / MLP: the submodules a fused pass stands for
            if want is not None and not self.per_channel and self.fuse_layer_statistics:
                parts = []
                for a in want:                                     # ("w1", ...) or dotted ("mlp.w2")
                    q = m
                    for piece in a.split("."):
                        q = getattr(q, piece, None) if q is not None else None
                    parts.append(q)
                keys = [names.get(id(q)) for q in parts]
                if all(k is not None and (k, "input") in self.slots and (k, "output") in self.slots for k in keys):
                    owner = m.__dict__.get("_mq_calib_layer")
                    if owner is not None and owner[0] is not self:
                        owner[0]._drop_layer_fusion()                  # two collectors on one model: modules a fused pass skips would never
                        self.fuse_layer_statistics = False             # reach the other one's hooks -- both keep the plain hooks
                        continue
                    m._mq_calib_layer = (self, tuple(keys))
                    self._layers.append(m)
        return self

    def _drop_layer_fusion(self) -> None:
        self.fuse_layer_statistics = False
        for m in self._layers:
            if m.__dict__.get("_mq_calib_layer", (None,))[0] is self:
                m.__dict__.pop("_mq_calib_layer", None)
        self._layers = []
        if self._forced:
            self._resolve()
            for k in self._forced:
                self._mirror.pop(k, None)
            self._forced = set()

    def detach(self) -> None:
        for h in self._hooks:
            h.remove()
        self._hooks = []
        for m in self._aware:
            if m.__dict__.get("_mq_calib", (None,))[0] is self:
                m.__dict__.pop("_mq_calib", None)
        self._aware = []
        for m in self._layers:
            if m.__dict__.get("_mq_calib_layer", (None,))[0] is self:
                m.__dict__.pop("_mq_calib_layer", None)
        self._layers = []
        self._probs_buf.clear()

    # -- merge -------------------------------------------------------------------------------------
    def _layout(self) -> Dict[int, int]:
        """Per-channel mode: {slot: channels} of what THIS rank observed."""
        self._resolve()
        return {i: int(s[0].numel()) for i, s in enumerate(self._pc) if s is not None}

    @staticmethod
    def _layout_checksum(layout: Dict[int, int]) -> float:
        """An integer below 2^23 (exact in fp32, and so is its negation) that depends on which slots are packed and how wide."""
        acc = 17
        for i in sorted(layout):
            acc = (acc * 8191 + (i + 1) * 131 + layout[i] * 7919) % 8388593
        return float(acc)

    def _packed(self, layout: Optional[Dict[int, int]] = None) -> torch.Tensor:
        if not self.per_channel:
            self._resolve()
            return torch.cat((-self._mn, self._mx))
        layout = self._layout() if layout is None else layout
        cs = self._layout_checksum(layout)
        parts = [torch.tensor([cs, -cs], dtype=torch.float32, device=self.device)]     # MAX keeps (cs, -cs) iff every rank packed this layout
        for i in sorted(layout):
            s = self._pc[i]
            if s is None:                           # observed elsewhere only: the neutral element of MAX over [-min, max]
                parts.append(torch.full((2 * layout[i],), float("-inf"), dtype=torch.float32, device=self.device))
            else:
                parts += [-s[0].to(self.device), s[1].to(self.device)]
        return torch.cat(parts)

    def _unpack(self, buf: torch.Tensor, layout: Optional[Dict[int, int]] = None) -> None:
        if not self.per_channel:
            n = len(self.slots)
            self._mn, self._mx = -buf[:n], buf[n:]
            return
        layout = self._layout() if layout is None else layout
        if float(buf[0]) != -float(buf[1]):
            raise RuntimeError("per-channel calibration merge: the ranks packed different slot layouts (a hooked module ran on some "
                               "ranks only); call all_reduce(verify_layout=True) / get_act_range(..., verify_layout=True)")
        off = 2
        for i in sorted(layout):
            c = layout[i]
            self._pc[i] = (-buf[off:off + c], buf[off + c:off + 2 * c])
            off += 2 * c

    def all_reduce(self, group=None, force: bool = False, verify_layout: bool = False) -> None:
        """ONE all-reduce(MAX) over the packed [-min, max] buffer of every tensor -- per-tensor and per-channel mode alike.
        Per-tensor: the layout is the model's named_modules(), identical on every rank.  Per-channel: slot widths come from each
        rank's own observations; get_act_range makes every rank run at least one sample (a duplicate if it owns none; min / max are
        idempotent) and the reference's per-channel mode already requires equal shapes across samples
        (generate_act_range.py:57-63), so on a dense graph the packed layout is identical on all ranks without a second collective.
        The buffer carries a checksum of the layout (+c, -c under MAX): equal-length but different layouts raise instead of
        mis-merging.  Different LENGTHS (a hooked module that ran on some ranks only: data-dependent paths, MoE) would abort inside
        the collective, so callers with such graphs pass verify_layout=True: a 4-float guard all-reduce first and, on a mismatch,
        one all_gather_object of the slot widths to agree on the union layout (ranks fill what they did not see with the neutral
        element).  force: also run the pack / reduce / unpack path in a 1-rank group (tests)."""
        if not (dist.is_available() and dist.is_initialized()):
            return
        if dist.get_world_size(group) == 1 and not force:
            return
        layout = None
        if self.per_channel and verify_layout:
            mine = self._layout()
            n, cs = float(2 + 2 * sum(mine.values())), self._layout_checksum(mine)
            guard = torch.tensor([n, -n, cs, -cs], dtype=torch.float32, device=self.device)
            dist.all_reduce(guard, op=dist.ReduceOp.MAX, group=group)
            if float(guard[0]) != -float(guard[1]) or float(guard[2]) != -float(guard[3]):
                every = [None] * dist.get_world_size(group)
                dist.all_gather_object(every, mine, group=group)
                layout = {}
                for other in every:
                    for i, c in other.items():
                        if layout.setdefault(i, c) != c:
                            name = [k for k, v in self.slots.items() if v == i][0]
                            raise RuntimeError(f"per-channel calibration merge: slot {name} has {layout[i]} channels on one rank and {c} on another")
        buf = self._packed(layout)
        if buf.numel():
            dist.all_reduce(buf, op=dist.ReduceOp.MAX, group=group)
            self.n_collectives += 1
        self._unpack(buf, layout)

    # -- results -----------------------------------------------------------------------------------
    def act_dict(self) -> Dict[str, dict]:
        """``{module: {field: [min, max]}}`` (per-tensor, Python floats -- the only host readback of the
        whole run) or ``{module: {field: Tensor[2, C]}}`` on the CPU (per-channel)."""
        out: Dict[str, dict] = {}
        if self.per_channel:
            self._resolve()
            for (name, field), i in self.slots.items():
                if self._pc[i] is not None:
                    out.setdefault(name, {})[field] = torch.stack(self._pc[i], dim=0).cpu()
            return out
        self._resolve()
        mn, mx = self._mn.tolist(), self._mx.tolist()
        for (name, field), i in self.slots.items():
            if mn[i] <= mx[i]:
                out.setdefault(name, {})[field] = [mn[i], mx[i]]
        return out

    def act_scales(self) -> Dict[str, torch.Tensor]:
        """SmoothQuant statistics: per-channel absmax keyed ``"<module>_<field>"`` for Linear / Norm
        leaves (generate_act_scale_shift.py:47-71); absmax = max(|min|, |max|) of the same running stats."""
        assert self.per_channel, "act_scales needs per-channel statistics"
        self._resolve()
        out = {}
        mods = dict(self.model.named_modules())
        for (name, field), i in self.slots.items():
            m = mods[name]
            is_norm = isinstance(m, (nn.LayerNorm, HFRMSNorm)) or any(c.__name__ == "HFRMSNorm" for c in type(m).__mro__)
            if field in ("input", "output") and (isinstance(m, nn.Linear) or is_norm) and self._pc[i] is not None:
                mn, mx = self._pc[i]
                out[f"{name}_{field}"] = torch.maximum(mn.abs(), mx.abs()).float().cpu()
        return out


@torch.no_grad()
def get_act_range(model: nn.Module, samples: Sequence[torch.Tensor], per_channel: bool = False, group=None,
                  forward=None, force_collective: bool = False, verify_layout: bool = False) -> Dict[str, dict]:
    """Data-parallel counterpart of ``get_act_range`` (generate_act_range.py:49-122).

    ``samples``: the full list of calibration inputs (token-id tensors), identical on every rank; rank r
    processes samples r, r + world, ...  Returns the merged act_dict on every rank.
    """
    model.eval()
    rank = dist.get_rank(group) if dist.is_available() and dist.is_initialized() else 0
    world = dist.get_world_size(group) if dist.is_available() and dist.is_initialized() else 1
    col = ActRangeCollector(model, per_channel).attach()
    dev = col.device
    run = forward if forward is not None else (lambda s: model(s))
    mine = list(range(rank, len(samples), world))
    if not mine and len(samples):
        mine = [rank % len(samples)]          # a duplicate: min / max are idempotent, and the rank learns every slot's shape
    try:
        for i in mine:
            with torch.cuda.device(dev) if dev.type == "cuda" else _nullcontext():
                run(samples[i].to(dev))
    finally:
        col.detach()
    col.all_reduce(group, force=force_collective, verify_layout=verify_layout)
    return col.act_dict()


def save_act_dict(path: str, act_dict: Dict[str, dict]) -> None:
    """Same bytes as the reference's json_save (io.py:34-36)."""
    with open(path, "w") as f:
        json.dump(act_dict, f, indent=4, sort_keys=True)

"""Multi-threaded CPU restatement of the reference's simulated-quant forward, in torch CPU ops.  TEST INFRASTRUCTURE ONLY.

Same role and rules as ``oracle/mq_oracle.py`` (only ``tests/``, ``__graft_entry__.smoke()`` and ``bench.py``'s ``cpu_baseline``
import it).  The numpy oracle is the parity checker; this file exists for the CPU BASELINE of BASELINE.md section 3 / SURVEY
section 8d: the reference's own op sequence (``mobilellm/quantization/qmodule.py:251-358``: ~8 elementwise torch ops per
``Quantizer.forward``, the weight re-quantised on every ``QLinear.forward``, an fp32 ``F.linear``) executed by torch's
multi-threaded CPU kernels with ``torch.set_num_threads(physical cores)`` -- i.e. what running the reference on the GPU box's host
cores costs.  ``tests/test_oracle_golden.py`` checks it bit-exactly against the numpy oracle (elementwise parts) so the baseline
times the right arithmetic.  Each function cites the reference lines it follows.
"""
from __future__ import annotations

import math

import torch
import torch.nn.functional as F

CLIPMIN, CLIPMAX = 1e-5, 1e6      # qmodule.py:11-12


def scale_offset(mn: torch.Tensor, mx: torch.Tensor, bitwidth: int, symmetric: bool):
    """qmodule.py:40-61."""
    if symmetric:
        qmin, qmax = -(2 ** (bitwidth - 1)), 2 ** (bitwidth - 1) - 1
        alpha, beta = torch.maximum(mn.abs(), mx.abs()), torch.zeros_like(mn)
    else:
        qmin, qmax = 0, 2 ** bitwidth - 1
        alpha, beta = mx - mn, mn
    scale = (alpha / qmax).clamp(min=CLIPMIN, max=CLIPMAX)
    offset = -(beta / scale).round()
    return scale, offset, qmin, qmax


class Quantizer:
    """Static or first-forward per-tensor / per-row quantizer (qmodule.py:112-295), fp32."""

    def __init__(self, bitwidth=8, symmetric=False, per_channel=False):
        self.bitwidth, self.symmetric, self.per_channel = bitwidth, symmetric, per_channel
        self.scale = self.offset = None

    def set_range(self, lo: float, hi: float):
        self.scale, self.offset, self.qmin, self.qmax = scale_offset(torch.tensor(lo), torch.tensor(hi), self.bitwidth,
                                                                     self.symmetric)
        return self

    def __call__(self, x: torch.Tensor) -> torch.Tensor:
        if self.bitwidth > 16:                                   # qmodule.py:252
            return x
        if self.scale is None:                                   # first forward: range from the tensor (qmodule.py:262-277)
            if self.per_channel:
                mn, mx = x.amin(dim=-1, keepdim=True), x.amax(dim=-1, keepdim=True)
            else:
                mn, mx = x.amin(), x.amax()
            self.scale, self.offset, self.qmin, self.qmax = scale_offset(mn, mx, self.bitwidth, self.symmetric)
        t = x / self.scale                                       # qmodule.py:286-290, op for op
        q = (t + (t.round() - t)) + self.offset                  # round_ste (qmodule.py:17-21)
        q = q.clamp(self.qmin, self.qmax)
        return (q - self.offset) * self.scale


def qlinear(x, w, bias, wq: Quantizer, iq, oq):
    """QLinear.forward (qmodule.py:341-358): the weight is fake-quantised on EVERY call (its cached grid is reused)."""
    wf = wq(w)
    if iq is not None:
        x = iq(x)
    y = F.linear(x, wf, bias)
    return oq(y) if oq is not None else y


def qrmsnorm(x, weight, eps, iq, wq, oq):
    """QRMSNorm.forward (qmodule.py:515-531 around hf_model.py:162-198)."""
    w = wq(weight)
    x = iq(x)
    y = x * torch.rsqrt(x.pow(2).mean(-1, keepdim=True) + eps)
    return oq(w * y)


def qsilu(x, mq: Quantizer, oq: Quantizer):
    """QSiLU.forward (qmodule.py:739-753): x * Q(sigmoid(x)), sigmoid grid [0, 1]."""
    return oq(x * mq(torch.sigmoid(x)))


def qmatmul(a, b, q1, q2, oq):
    """QMatMul.forward (qmodule.py:453-466)."""
    return oq(torch.matmul(q1(a), q2(b)))


class SimLayer:
    """One TinyLlama-shaped decoder layer under the W8A8 recipe (module graph: hf_model.py:426-534, :1057, :1208-1260; mixed
    precision: ptq/mobilequant.py:175-201 -- 16-bit norm inputs / weights, 16-bit o_proj / w2 outputs, per-channel w2, 16-bit
    qk_bmm output and pv_bmm input), random weights, static activation ranges."""

    def __init__(self, hidden=2048, heads=32, kv_heads=4, head_dim=64, ffn=5632, seed=0):
        g = torch.Generator().manual_seed(seed)
        r = lambda *s: torch.randn(*s, generator=g) * 0.02       # noqa: E731
        self.h, self.H, self.KV, self.D, self.F = hidden, heads, kv_heads, head_dim, ffn
        self.w = {"q": r(heads * head_dim, hidden), "k": r(kv_heads * head_dim, hidden), "v": r(kv_heads * head_dim, hidden),
                  "o": r(hidden, heads * head_dim), "w1": r(ffn, hidden), "w3": r(ffn, hidden), "w2": r(hidden, ffn)}
        self.n1, self.n2 = torch.ones(hidden), torch.ones(hidden)
        Q = Quantizer
        self.wq = {k: Q(8, per_channel=(k == "w2")) for k in self.w}
        a8 = lambda lo, hi: Q(8).set_range(lo, hi)                # noqa: E731
        a16 = lambda lo, hi: Q(16).set_range(lo, hi)              # noqa: E731
        self.q = dict(n1_in=a16(-6, 6), n1_w=Q(16), n1_out=a8(-4, 4), n2_in=a16(-6, 6), n2_w=Q(16), n2_out=a8(-4, 4),
                      q_out=a8(-3, 3), k_out=a8(-3, 3), v_out=a8(-3, 3), qk_a=a8(-3, 3), qk_b=a8(-3, 3), qk_out=a16(-8, 8),
                      pv_a=a16(0, 1), pv_b=a8(-3, 3), pv_out=a8(-3, 3), o_in=a8(-3, 3), o_out=a16(-3, 3),
                      w1_out=a8(-3, 3), w3_out=a8(-3, 3), silu_mid=a8(0, 1), silu_out=a8(-1, 3), w2_in=a8(-3, 3),
                      w2_out=a16(-3, 3))

    def forward(self, x):
        q, w, wq = self.q, self.w, self.wq
        S = x.shape[0]
        h = qrmsnorm(x, self.n1, 1e-5, q["n1_in"], q["n1_w"], q["n1_out"])
        qs = qlinear(h, w["q"], None, wq["q"], None, q["q_out"]).view(S, self.H, self.D).transpose(0, 1)
        ks = qlinear(h, w["k"], None, wq["k"], None, q["k_out"]).view(S, self.KV, self.D).transpose(0, 1)
        vs = qlinear(h, w["v"], None, wq["v"], None, q["v_out"]).view(S, self.KV, self.D).transpose(0, 1)
        rep = self.H // self.KV
        ks, vs = ks.repeat_interleave(rep, dim=0), vs.repeat_interleave(rep, dim=0)
        att = qmatmul(qs, ks.transpose(1, 2), q["qk_a"], q["qk_b"], q["qk_out"]) / math.sqrt(self.D)
        att = att + torch.full((S, S), float("-inf")).triu(1)
        att = torch.softmax(att, dim=-1)
        o = qmatmul(att, vs, q["pv_a"], q["pv_b"], q["pv_out"]).transpose(0, 1).reshape(S, self.H * self.D)
        x = x + qlinear(o, w["o"], None, wq["o"], None, q["o_out"])
        h = qrmsnorm(x, self.n2, 1e-5, q["n2_in"], q["n2_w"], q["n2_out"])
        a = qsilu(qlinear(h, w["w1"], None, wq["w1"], None, q["w1_out"]), q["silu_mid"], q["silu_out"])
        b = qlinear(h, w["w3"], None, wq["w3"], None, q["w3_out"])
        return x + qlinear(a * b, w["w2"], None, wq["w2"], q["w2_in"], q["w2_out"])


def physical_cores() -> int:
    """Physical cores of this host (measured, not assumed): psutil if present, else /proc/cpuinfo, else os.cpu_count()."""
    try:
        import psutil
        n = psutil.cpu_count(logical=False)
        if n:
            return int(n)
    except Exception:
        pass
    try:
        seen = set()
        phys = core = None
        for line in open("/proc/cpuinfo"):
            if line.startswith("physical id"):
                phys = line.split(":")[1].strip()
            elif line.startswith("core id"):
                core = line.split(":")[1].strip()
            elif not line.strip() and phys is not None and core is not None:
                seen.add((phys, core))
                phys = core = None
        if seen:
            return len(seen)
    except OSError:
        pass
    import os
    return os.cpu_count() or 1

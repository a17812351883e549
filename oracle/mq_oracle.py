"""CPU oracle for the MobileQuant simulated-quant hot path.  TEST INFRASTRUCTURE ONLY.

This file is a numpy restatement of the reference's algorithm for the one hot path this
repository accelerates: the fake-quant arithmetic of
``mobilellm/quantization/qmodule.py`` and the running min/max statistics of
``ptq/generate_act_range.py`` / ``ptq/generate_act_scale_shift.py``.  Every function cites
the reference ``file:line`` it follows (paths relative to the reference checkout).

Rules (see DESIGN.md "Oracle"):
  * Only ``tests/``, ``__graft_entry__.smoke()`` and the ``cpu_baseline`` leg of
    ``bench.py`` may import this module.  The product package ``mobilequant_amd`` never does;
    it fails loudly when the HIP library is missing.
  * Parity is PINNED: ``tests/test_oracle_golden.py`` checks every function here bit-exactly
    against ``tests/golden/*.npz``, which ``oracle/gen_golden.py`` produced by importing the
    real reference (torch CPU) in the build container.
  * All arithmetic is IEEE fp32 (numpy float32), same operation order as the reference, so the
    integer indices it yields are the reference's, bit for bit.  The only operation that is NOT
    bit-reproducible is the fp32 matmul inside ``qlinear_sim`` (BLAS summation order); tests
    state a tolerance there and use ``qlinear_int_exact`` for the exact integer contraction.
"""
from __future__ import annotations

import numpy as np

F32 = np.float32
CLIPMIN = F32(1e-5)   # qmodule.py:11
CLIPMAX = F32(1e6)    # qmodule.py:12


# ----------------------------------------------------------------------------------------------
# a1  compute_scale_offset_from_min_max            qmodule.py:40-61
# ----------------------------------------------------------------------------------------------
def qrange(bitwidth: int, is_symmetric: bool):
    """(qmin, qmax) integer grid limits.  qmodule.py:48-54."""
    if is_symmetric:
        return -(2 ** (bitwidth - 1)), 2 ** (bitwidth - 1) - 1
    return 0, 2 ** bitwidth - 1


def scale_offset_from_min_max(min_val, max_val, bitwidth: int, is_symmetric: bool):
    """Returns (scale, offset, qmin, qmax); scale/offset are fp32 arrays shaped like min_val.

    qmodule.py:40-61: asymmetric ``alpha = max-min, beta = min``; symmetric
    ``alpha = max(|min|,|max|), beta = 0``; ``scale = clamp(alpha/qmax, 1e-5, 1e6)``;
    ``offset = -round(beta/scale)`` (round half to even; symmetric gives -0.0).
    Python floats enter through ``torch.tensor(v)`` = fp32 (qmodule.py:41-44).
    """
    mn = np.asarray(min_val, dtype=F32)
    mx = np.asarray(max_val, dtype=F32)
    qmin, qmax = qrange(bitwidth, is_symmetric)
    if is_symmetric:
        alpha = np.maximum(np.abs(mn), np.abs(mx))
        beta = np.zeros_like(alpha)
    else:
        alpha = mx - mn
        beta = mn
    scale = (alpha / F32(qmax)).astype(F32)
    scale = np.clip(scale, CLIPMIN, CLIPMAX).astype(F32)
    offset = (-np.rint((beta / scale).astype(F32))).astype(F32)
    return scale, offset, qmin, qmax


# ----------------------------------------------------------------------------------------------
# a2  compute_min_max_from_scale_offset            qmodule.py:66-76
# ----------------------------------------------------------------------------------------------
def min_max_from_scale_offset(scale, offset, bitwidth: int, is_symmetric: bool):
    """Inverse map used by export_act_range (qmodule.py:908-937)."""
    _, qmax = qrange(bitwidth, is_symmetric)
    s = np.clip(np.asarray(scale, dtype=F32), CLIPMIN, CLIPMAX).astype(F32)
    o = np.asarray(offset, dtype=F32)
    alpha = (s * F32(qmax)).astype(F32)
    beta = ((-o) * s).astype(F32)
    max_val = (alpha + beta).astype(F32)
    min_val = (-max_val) if is_symmetric else beta
    return min_val, max_val


# ----------------------------------------------------------------------------------------------
# a3  compute_min_max_from_tensor                  qmodule.py:26-34
# ----------------------------------------------------------------------------------------------
def min_max_from_tensor(x, is_per_channel: bool = False, group_size: int = -1):
    """Per-tensor amin/amax of the flattened tensor, or per-row (last dim, keepdim) /
    per-group (reshape(-1, g) first).  qmodule.py:26-34 and :259-268."""
    x = np.asarray(x)
    if is_per_channel:
        if group_size != -1:
            x = x.reshape(-1, group_size)
        return x.min(axis=-1, keepdims=True), x.max(axis=-1, keepdims=True)
    flat = x.reshape(-1)
    return flat.min(), flat.max()


# ----------------------------------------------------------------------------------------------
# a4/a5  round_ste + Quantizer.forward arithmetic   qmodule.py:17-21, :286-295
# ----------------------------------------------------------------------------------------------
def quantize_index(x, scale, offset, qmin: int, qmax: int):
    """Integer grid index as an fp32 array: ``clamp(round(x/scale) + offset, qmin, qmax)``.

    qmodule.py:286-287.  True IEEE division, round-half-even, fp32 add, clamp.  round_ste
    (qmodule.py:17-21) evaluates ``(round(t) - t) + t``: exact for finite t (the difference is exactly
    representable), NaN for t = +-inf (inf - inf) -- so an infinite input comes out NaN like a NaN one
    (frozen in tests/golden/nonfinite_cases.npz).  np.clip, like torch.clamp, keeps NaN.
    """
    x = np.asarray(x, dtype=F32)
    s = np.asarray(scale, dtype=F32)
    o = np.asarray(offset, dtype=F32)
    with np.errstate(invalid="ignore"):
        t = (x / s).astype(F32)
        r = ((np.rint(t) - t).astype(F32) + t).astype(F32)
        q = (r + o).astype(F32)
    return np.clip(q, F32(qmin), F32(qmax)).astype(F32)


def dequantize_index(q, scale, offset):
    """``(q - offset) * scale`` in fp32.  qmodule.py:290."""
    q = np.asarray(q, dtype=F32)
    s = np.asarray(scale, dtype=F32)
    o = np.asarray(offset, dtype=F32)
    return ((q - o).astype(F32) * s).astype(F32)


def fake_quant(x, scale, offset, qmin: int, qmax: int):
    """quantize -> dequantize, fp32.  qmodule.py:286-290."""
    return dequantize_index(quantize_index(x, scale, offset, qmin, qmax), scale, offset)


def fake_quant_backward(x, grad_y, scale, offset, qmin: int, qmax: int):
    """Gradients of ``y = (clamp(round_ste(x/s) + o, qmin, qmax) - o) * s`` (qmodule.py:17-21, :286-290) as
    torch autograd derives them: round is a straight-through identity, clamp passes the gradient where
    qmin <= q <= qmax.  With t = x/s, r = round(t), q = r + o:
        dL/dx = (g*s)/s      inside | 0                 clamped
        dL/ds = g * (r - t)  inside | g * (clamp(q)-o)  clamped
        dL/do = 0            inside | -g * s            clamped
    scale/offset gradients are summed over the elements that share them (all, or one row).
    Returns (grad_x, grad_scale, grad_offset) with the shapes of x / scale / offset."""
    x = np.asarray(x, dtype=F32)
    g = np.asarray(grad_y, dtype=F32)
    s = np.asarray(scale, dtype=F32)
    o = np.asarray(offset, dtype=F32)
    with np.errstate(invalid="ignore"):
        t = (x / s).astype(F32)
        r = ((np.rint(t) - t).astype(F32) + t).astype(F32)       # round_ste, see quantize_index
        q = (r + o).astype(F32)
    inside = (q >= F32(qmin)) & (q <= F32(qmax))
    qc = np.clip(q, F32(qmin), F32(qmax))
    # autograd chain for x: (g * s) through the clamp mask, then the division's grad / s  -- reproduced
    # op for op so grad_x is bit-identical to torch's (it is g only up to one rounding)
    gx = (np.where(inside, (g * s).astype(F32), F32(0)) / s).astype(F32)
    gs_e = np.where(inside, g * (r - t), g * (qc - o)).astype(np.float64)
    go_e = np.where(inside, 0.0, -g * s).astype(np.float64)
    if s.size == 1:
        return gx, np.asarray(gs_e.sum(), dtype=F32).reshape(s.shape), np.asarray(go_e.sum(), dtype=F32).reshape(o.shape)
    axes = tuple(range(1, x.ndim))
    return gx, gs_e.sum(axis=axes).astype(F32).reshape(s.shape), go_e.sum(axis=axes).astype(F32).reshape(o.shape)


def fake_quant_f16_per_tensor(x, scale, offset, qmin: int, qmax: int):
    """fp16 input with 0-dim fp32 scale/offset: the RESULT dtype stays fp16 (SURVEY 8a' item 4).

    Measured against torch CPU (and frozen in quantizer_cases.npz): each op of qmodule.py:286-290
    is evaluated in float with the half operand widened and the 0-dim fp32 scale/offset used at
    full fp32 precision, and the result is rounded to half once per op.
    """
    H = np.float16
    x = np.asarray(x, dtype=H)
    s = F32(scale)
    o = F32(offset)
    t = (x.astype(F32) / s).astype(H)
    with np.errstate(invalid="ignore"):
        r0 = np.rint(t.astype(F32)).astype(H)
        r = ((r0.astype(F32) - t.astype(F32)).astype(H).astype(F32) + t.astype(F32)).astype(H)   # round_ste in half
    q = (r.astype(F32) + o).astype(H)
    q = np.clip(q, H(qmin), H(qmax)).astype(H)
    d = (q.astype(F32) - o).astype(H)
    return (d.astype(F32) * s).astype(H), q


class QuantizerOracle:
    """State machine of ``Quantizer`` (qmodule.py:112-295) without autograd.

    ``forward`` reproduces qmodule.py:251-295: bypass when disabled or bitwidth > 16; optional
    group reshape; (re)compute scale/offset when dynamic / LWC / not cached; fake-quantise.
    """

    def __init__(self, bitwidth=32, group_size=-1, is_symmetric=False, is_per_channel=False,
                 is_dynamic=False):
        self.bitwidth, self.group_size = bitwidth, group_size
        self.is_symmetric, self.is_per_channel, self.is_dynamic = is_symmetric, is_per_channel, is_dynamic
        self.enable = True
        self.scale = self.offset = None
        self.qmin = self.qmax = None

    def set_from_minmax(self, mn, mx):                       # qmodule.py:216-245
        self.scale, self.offset, self.qmin, self.qmax = scale_offset_from_min_max(
            mn, mx, self.bitwidth, self.is_symmetric)

    def forward(self, x, return_index=False):
        if (not self.enable) or self.bitwidth > 16:          # qmodule.py:252-253
            return (x, None) if return_index else x
        x = np.asarray(x, dtype=F32)
        shape = x.shape
        if self.is_per_channel and self.group_size != -1:    # qmodule.py:259-260
            x = x.reshape(-1, self.group_size)
        if self.is_dynamic or self.scale is None:            # qmodule.py:262-277
            mn, mx = min_max_from_tensor(x, self.is_per_channel, -1)
            self.set_from_minmax(mn, mx)
        q = quantize_index(x, self.scale, self.offset, self.qmin, self.qmax)
        y = dequantize_index(q, self.scale, self.offset).reshape(shape)
        return (y, q.reshape(shape)) if return_index else y


# ----------------------------------------------------------------------------------------------
# a8  QLinear.forward                               qmodule.py:341-358
# ----------------------------------------------------------------------------------------------
def qlinear_sim(x, weight, bias, w_q: QuantizerOracle | None, in_q: QuantizerOracle | None,
                out_q: QuantizerOracle | None):
    """The reference's simulated path: fake-quant W, (opt) fake-quant x, fp32 linear, fake-quant out.

    qmodule.py:341-358.  The fp32 matmul's summation order is BLAS-defined, so the pre-output-
    quant values agree with torch only to fp32 round-off (tests state the tolerance).
    """
    w = np.asarray(weight, dtype=F32)
    if w_q is not None:
        w = w_q.forward(w)
    x = np.asarray(x, dtype=F32)
    if in_q is not None:
        x = in_q.forward(x)
    out = x.reshape(-1, w.shape[1]) @ w.T
    if bias is not None:
        out = out + np.asarray(bias, dtype=F32)
    out = out.astype(F32).reshape(*x.shape[:-1], w.shape[0])
    if out_q is not None:
        out = out_q.forward(out)
    return out


def qlinear_int_exact(qa, za, sa, qw, zw, sw, bias=None, blas=False):
    """Integer-GEMM equivalence of QLinear (SURVEY 8a' item 9), exact contraction.

    ``out[m,n] = sa*sw[n] * sum_k (qa[m,k]-za)(qw[n,k]-zw[n]) + bias[n]``.  The contraction is
    done in int64 (exact); the scaling mirrors the HIP epilogue: one int->fp32 conversion, one
    multiply by fp32(sa*sw[n]), one add.  Returns (acc_int64, out_fp32).
    blas=True: the same contraction as a float64 BLAS product -- still exact (every partial sum is an integer below
    K * 511 * 511 < 2^53 for K <= 65536), and fast enough to check EVERY output of the full-size BASELINE shapes.
    """
    qa = np.asarray(qa, dtype=np.int64)
    qw = np.asarray(qw, dtype=np.int64)
    za_i = np.asarray(za, dtype=np.int64)
    zw_i = np.asarray(zw, dtype=np.int64).reshape(-1, 1) if np.ndim(zw) else np.int64(zw)
    if blas:
        assert qa.shape[1] * 511 * 511 < 2 ** 53
        acc = np.rint((qa - za_i).astype(np.float64) @ (qw - zw_i).astype(np.float64).T).astype(np.int64)
    else:
        acc = (qa - za_i) @ (qw - zw_i).T
    alpha = (F32(sa) * np.asarray(sw, dtype=F32).reshape(-1)).astype(F32)
    out = (acc.astype(F32) * alpha).astype(F32)
    if bias is not None:
        out = (out + np.asarray(bias, dtype=F32)).astype(F32)
    return acc, out


# ----------------------------------------------------------------------------------------------
# a12  update_act_range                             ptq/generate_act_range.py:55-69
# ----------------------------------------------------------------------------------------------
class ActRangeOracle:
    """Running min/max per (module name, field) over a stream of tensors.

    per-tensor: Python ``min``/``max`` of ``.min().item()/.max().item()`` (generate_act_range.py:65-69);
    per-channel: ``reshape(-1, C)``, min/max over dim 0, running ``minimum``/``maximum`` kept as a
    ``[2, C]`` tensor (generate_act_range.py:57-63).
    """

    def __init__(self, per_channel: bool = False):
        self.per_channel = per_channel
        self.act_dict: dict = {}

    def update(self, name: str, field: str, t):
        t = np.asarray(t, dtype=F32)
        entry = self.act_dict.setdefault(name, {})
        if self.per_channel:
            t2 = t.reshape(-1, t.shape[-1])
            cur = np.stack((t2.min(axis=0), t2.max(axis=0)), axis=0)
            if field in entry:
                cur[0] = np.minimum(entry[field][0], cur[0])
                cur[1] = np.maximum(entry[field][1], cur[1])
            entry[field] = cur
        else:
            mn, mx = float(t.min()), float(t.max())
            if field in entry:
                mn, mx = min(entry[field][0], mn), max(entry[field][1], mx)
            entry[field] = [mn, mx]

    @staticmethod
    def merge(dicts, per_channel: bool):
        """Min/max merge of several shards' act_dicts: what the all-reduce computes."""
        out: dict = {}
        for d in dicts:
            for name, fields in d.items():
                for field, v in fields.items():
                    e = out.setdefault(name, {})
                    if field not in e:
                        e[field] = np.array(v, dtype=F32).copy() if per_channel else list(v)
                    elif per_channel:
                        e[field][0] = np.minimum(e[field][0], v[0])
                        e[field][1] = np.maximum(e[field][1], v[1])
                    else:
                        e[field] = [min(e[field][0], v[0]), max(e[field][1], v[1])]
        return out


# ----------------------------------------------------------------------------------------------
# a13  stat_tensor (SmoothQuant absmax)             ptq/generate_act_scale_shift.py:47-55
# ----------------------------------------------------------------------------------------------
class ActScaleOracle:
    """Running per-channel absmax keyed ``"<module>_<field>"`` (generate_act_scale_shift.py:47-55)."""

    def __init__(self):
        self.act_scales: dict = {}

    def update(self, name: str, field: str, t):
        t = np.asarray(t, dtype=F32)
        cur = np.abs(t.reshape(-1, t.shape[-1])).max(axis=0)
        key = f"{name}_{field}"
        self.act_scales[key] = np.maximum(self.act_scales[key], cur) if key in self.act_scales else cur


# ----------------------------------------------------------------------------------------------
# Storage formats the integer path adds (no reference counterpart; defined in DESIGN.md).
# The values they carry are the reference's indices from quantize_index().
# ----------------------------------------------------------------------------------------------
def qrmsnorm(x, weight, bias, eps, in_q, w_q, out_q):
    """QRMSNorm.forward (qmodule.py:518-530 around HFRMSNorm.forward_impl, hf_model.py:184-195), fp32:
    ``Qout( Qw(weight) * (xi * rsqrt(mean(xi^2, -1) + eps)) (+ bias) )`` with ``xi = Qin(x)``; any quantizer may be
    None.  rsqrt is 1/sqrt with both operations correctly rounded (what torch's CPU kernel computes); the mean is
    numpy's pairwise sum / n, which is not torch's summation order: agreement with the frozen reference outputs
    is within one output LSB on a vanishing fraction of elements (tests state the bound)."""
    x = np.asarray(x, dtype=F32)
    w = np.asarray(weight, dtype=F32)
    xi = in_q.forward(x) if in_q is not None else x
    wq = w_q.forward(w) if w_q is not None else w
    ms = (xi * xi).astype(F32).mean(axis=-1, keepdims=True, dtype=F32)
    r = (F32(1.0) / np.sqrt((ms + F32(eps)).astype(F32)).astype(F32)).astype(F32)
    y = (wq * (xi * r).astype(F32)).astype(F32)
    if bias is not None:
        y = (y + np.asarray(bias, dtype=F32)).astype(F32)
    return out_q.forward(y) if out_q is not None else y


def qlayernorm(x, weight, bias, eps, in_q, w_q, out_q):
    """QLayerNorm.forward (qmodule.py:624-640 around F.layer_norm), fp32: mean and biased variance over the last dim,
    ``y = (xi * rstd + (-rstd * mean)) * Qw(weight) + bias`` (the expression of torch's CPU kernel; bias is not
    quantised), then the output quantizer.  Same summation-order caveat as qrmsnorm."""
    x = np.asarray(x, dtype=F32)
    xi = in_q.forward(x) if in_q is not None else x
    wq = w_q.forward(np.asarray(weight, dtype=F32)) if w_q is not None else np.asarray(weight, dtype=F32)
    mu = xi.mean(axis=-1, keepdims=True, dtype=F32)
    d = (xi - mu).astype(F32)
    var = (d * d).astype(F32).mean(axis=-1, keepdims=True, dtype=F32)
    rstd = (F32(1.0) / np.sqrt((var + F32(eps)).astype(F32)).astype(F32)).astype(F32)
    y = (((xi * rstd).astype(F32) + (-rstd * mu).astype(F32)).astype(F32) * wq).astype(F32)
    if bias is not None:
        y = (y + np.asarray(bias, dtype=F32)).astype(F32)
    return out_q.forward(y) if out_q is not None else y


def qsilu(x, in_q, mid_q, out_q):
    """QSiLU.forward (qmodule.py:739-754): ``Qout( xi * Qmid(sigmoid(xi)) )``, ``xi = Qin(x)``; fp32, sigmoid =
    1 / (1 + exp(-x)).  exp is the platform's (<= 1 ulp), so agreement with the frozen reference output is up to
    one output LSB on a vanishing fraction of elements."""
    x = np.asarray(x, dtype=F32)
    xi = in_q.forward(x) if in_q is not None else x
    with np.errstate(over="ignore"):
        g = (F32(1.0) / (F32(1.0) + np.exp(-xi).astype(F32)).astype(F32)).astype(F32)
    g = mid_q.forward(g) if mid_q is not None else g
    y = (xi * g).astype(F32)
    return out_q.forward(y) if out_q is not None else y


def qgelu(x, in_q, out_q):
    """QGELU.forward (qmodule.py:790-798): ``Qout( 0.5 * xi * (1 + erf(xi / sqrt 2)) )``, fp32."""
    from scipy.special import erf
    x = np.asarray(x, dtype=F32)
    xi = in_q.forward(x) if in_q is not None else x
    e = erf((xi * F32(0.7071067811865476)).astype(F32).astype(np.float64)).astype(F32)
    y = ((F32(0.5) * xi).astype(F32) * (F32(1.0) + e).astype(F32)).astype(F32)
    return out_q.forward(y) if out_q is not None else y


def index_to_i8(q, qmin: int):
    """Signed-byte storage of an 8-bit index: unsigned grids [0,255] are stored as q-128
    (MFMA i8 is signed), signed grids [-128,127] as is.  Returns (int8 array, shift)."""
    shift = 128 if qmin == 0 else 0
    return (np.asarray(q, dtype=np.int32) - shift).astype(np.int8), shift


def pack_w4(q, qmin: int):
    """Pack 4-bit weight indices [N,K] two per byte, K-interleaved in blocks of 32:
    byte j of a 16-byte group holds element j (low nibble) and element j+16 (high nibble) of the
    32-element K block.  Nibbles are stored unsigned (q - qmin).  K must be a multiple of 32."""
    q = np.asarray(q, dtype=np.int32) - qmin
    n, k = q.shape
    assert k % 32 == 0 and q.min() >= 0 and q.max() <= 15
    blk = q.reshape(n, k // 32, 2, 16)
    return (blk[:, :, 0, :] | (blk[:, :, 1, :] << 4)).astype(np.uint8).reshape(n, k // 2)


def unpack_w4(packed, qmin: int):
    p = np.asarray(packed, dtype=np.uint8)
    n, kh = p.shape
    blk = p.reshape(n, kh // 16, 16).astype(np.int32)
    out = np.stack((blk & 15, blk >> 4), axis=2).reshape(n, kh * 2)
    return out + qmin


# ----------------------------------------------------------------------------------------------
# a7  learnable weight clipping                      qmodule.py:133-185, :262-277
# ----------------------------------------------------------------------------------------------
def sigmoid_f32(x):
    """torch.sigmoid in fp32 (vectorised CPU kernels differ from this expression by at most one ulp; tests state it)."""
    x = np.asarray(x, dtype=F32)
    return (F32(1.0) / (F32(1.0) + np.exp(-x).astype(F32))).astype(F32)


def lwc_range(w, up_factor, low_factor, is_per_channel: bool):
    """Clipped range of a weight under LWC: ``max * sigmoid(upbound_factor)``, ``min * sigmoid(lowbound_factor)``
    (qmodule.py:271-273; the same lines in run_lwc :172-174)."""
    mn, mx = min_max_from_tensor(w, is_per_channel, -1)
    return (sigmoid_f32(low_factor) * mn).astype(F32), (sigmoid_f32(up_factor) * mx).astype(F32)


def lwc_forward(w, up_factor, low_factor, bitwidth: int, is_symmetric: bool, is_per_channel: bool):
    """Quantizer.forward with LWC enabled (qmodule.py:262-290): grid from the clipped range, then fake-quant.
    Returns (y, scale, offset)."""
    mn, mx = lwc_range(w, up_factor, low_factor, is_per_channel)
    scale, offset, qmin, qmax = scale_offset_from_min_max(mn, mx, bitwidth, is_symmetric)
    return fake_quant(w, scale, offset, qmin, qmax), scale, offset


def run_lwc(w, up_factor, low_factor, is_per_channel: bool):
    """Quantizer.run_lwc (qmodule.py:159-185): clamp the weight to its clipped range."""
    mn, mx = lwc_range(w, up_factor, low_factor, is_per_channel)
    return np.clip(np.asarray(w, dtype=F32), mn, mx).astype(F32)


def lwc_backward(w, up_factor, low_factor, grad_y, bitwidth: int, is_symmetric: bool, is_per_channel: bool):
    """Gradients of ``(lwc_forward(w) * grad_y).sum()`` w.r.t. upbound_factor, lowbound_factor and w, as torch autograd derives
    them from qmodule.py:262-290 + :40-61: through fake_quant to (scale, offset) (fake_quant_backward), through
    ``scale = clamp(alpha / qmax)``, ``offset = -round(beta / scale)`` (round: zero gradient) to the clipped (min, max), through
    the sigmoid to the factors, and through amin / amax (gradient to the extreme element, split evenly between ties) to w."""
    w = np.asarray(w, dtype=F32)
    mn0, mx0 = min_max_from_tensor(w, is_per_channel, -1)
    su, sl = sigmoid_f32(up_factor), sigmoid_f32(low_factor)
    mn, mx = (sl * mn0).astype(F32), (su * mx0).astype(F32)
    scale, offset, qmin, qmax = scale_offset_from_min_max(mn, mx, bitwidth, is_symmetric)
    gx, gs, go = fake_quant_backward(w, grad_y, scale, offset, qmin, qmax)
    # offset = -round(beta / scale): torch.round has zero gradient -> nothing flows through the offset
    raw = (mx - mn) / F32(qmax) if not is_symmetric else np.maximum(np.abs(mn), np.abs(mx)) / F32(qmax)
    inside = ((raw >= CLIPMIN) & (raw <= CLIPMAX)).astype(F32)            # clamp(min=1e-5, max=1e6)
    g_alpha = gs.reshape(np.shape(scale)) * inside / F32(qmax)
    if is_symmetric:
        take_max = (np.abs(mx) >= np.abs(mn)).astype(F32)                 # torch.maximum: gradient to the larger (ties: split)
        tie = (np.abs(mx) == np.abs(mn)).astype(F32)
        wmx = take_max - 0.5 * tie
        wmn = (1 - take_max) + 0.5 * tie
        g_mx = g_alpha * wmx * np.sign(mx)
        g_mn = g_alpha * wmn * np.sign(mn)
    else:
        g_mx, g_mn = g_alpha, -g_alpha
    g_up = (g_mx * mx0 * su * (1 - su)).astype(F32)
    g_lo = (g_mn * mn0 * sl * (1 - sl)).astype(F32)
    # amax / amin backward: the extreme element(s) of each reduction group
    is_mx = (w == mx0).astype(F32)
    is_mn = (w == mn0).astype(F32)
    ax = -1 if is_per_channel else None
    g_w = gx + (g_mx * su) * is_mx / is_mx.sum(axis=ax, keepdims=is_per_channel) \
             + (g_mn * sl) * is_mn / is_mn.sum(axis=ax, keepdims=is_per_channel)
    return g_up, g_lo, g_w.astype(F32)


# ----------------------------------------------------------------------------------------------
# a10  QMatMul.forward                               qmodule.py:453-466
# ----------------------------------------------------------------------------------------------
def qmatmul_sim(a, b, q1: QuantizerOracle | None, q2: QuantizerOracle | None, out_q: QuantizerOracle | None):
    """fake-quant both operands, fp32 matmul (summation order BLAS-defined), fake-quant the product."""
    a = np.asarray(a, dtype=F32)
    b = np.asarray(b, dtype=F32)
    if q1 is not None:
        a = q1.forward(a)
    if q2 is not None:
        b = q2.forward(b)
    out = np.matmul(a, b).astype(F32)
    return out_q.forward(out) if out_q is not None else out


def qmatmul_int_exact(qa, za, sa, qb, zb, sb):
    """Integer equivalent of the product inside QMatMul: ``sa*sb * sum_k (qa-za)(qb-zb)`` with an exact contraction (float64
    BLAS over integers), scaled once: one int -> fp32 conversion, one multiply by fp32(sa*sb).  qa [..., M, K], qb [..., K, N]."""
    acc = np.rint(np.matmul((np.asarray(qa, np.float64) - np.float64(za)), (np.asarray(qb, np.float64) - np.float64(zb)))).astype(np.int64)
    return acc, (acc.astype(F32) * (F32(sa) * F32(sb))).astype(F32)


# ----------------------------------------------------------------------------------------------
# a10 (in context)  attention core around the two QMatMuls      hf_model.py:486-534
# ----------------------------------------------------------------------------------------------
def rope_rotate_half(x, cos, sin):
    """x [..., S, D], cos / sin [S, D]:  x * cos + rotate_half(x) * sin  (hf_model.py:486-488 -> apply_rotary_pos_emb)."""
    x = np.asarray(x, dtype=F32)
    h = x.shape[-1] // 2
    rot = np.concatenate((-x[..., h:], x[..., :h]), axis=-1)
    return (x * cos.astype(F32)).astype(F32) + (rot * sin.astype(F32)).astype(F32)


def rope_partial(x, cos, sin):
    """Partial rotary embedding (hf_model.py:489-500): cos / sin [S, rot] with rot < D rotate the first rot dims of x [..., S, D],
    the rest passes through; rot == D is the plain rotate-half form (hf_model.py:487)."""
    rot = cos.shape[-1]
    x = np.asarray(x, dtype=F32)
    if rot == x.shape[-1]:
        return rope_rotate_half(x, cos, sin)
    return np.concatenate((rope_rotate_half(x[..., :rot], cos, sin), x[..., rot:]), axis=-1)


def _qmatmul_exact(a, b, q1: QuantizerOracle, q2: QuantizerOracle, out_q: QuantizerOracle | None, double_scale: bool):
    """QMatMul with its contraction carried out EXACTLY over the quantizer indices (what the MFMA / dot4 kernels compute), where the
    reference's fp32 matmul rounds after every product: s_a s_b sum_k (ia - za)(ib - zb), one rounding of fl32(s_a * s_b), then either
    one fp32 multiply of the (exact in fp32: |sum| < 2^24) integer (double_scale False: the q.k^T product) or one rounding of the
    double product (True: the p.v product, whose integer sums exceed 2^24 -- mq_attention.hip / mq_decode.hip)."""
    _, ia = q1.forward(np.asarray(a, F32), return_index=True)
    _, ib = q2.forward(np.asarray(b, F32), return_index=True)
    acc = np.rint(np.matmul(ia.astype(np.float64) - np.float64(q1.offset), ib.astype(np.float64) - np.float64(q2.offset))).astype(np.int64)
    alpha = F32(F32(q1.scale) * F32(q2.scale))
    out = (acc.astype(np.float64) * np.float64(alpha)).astype(F32) if double_scale else (acc.astype(F32) * alpha).astype(F32)
    return out_q.forward(out) if out_q is not None else out


def qmatmul_exact(a, b, q1: QuantizerOracle, q2: QuantizerOracle, out_q: QuantizerOracle | None):
    """What mq_qmatmul (the standalone integer QMatMul, qmodule.py:453-466) computes: the exact contraction over the indices, one
    rounding -- single-precision scale for operands of at most 8 bits, the rounded double product when the first operand is wider
    (its integer sums exceed 2^24) -- then the output quantizer."""
    return _qmatmul_exact(a, b, q1, q2, out_q, double_scale=q1.bitwidth > 8)


def attention_sim(q, k, v, cos, sin, heads, kv_heads, qk: tuple, pv: tuple, exact_int: bool = False):
    """Causal prefill attention of one sequence as the reference computes it: q [S, heads*D], k / v [S, kv_heads*D] projection
    outputs; RoPE (cos / sin [S, rot_dim]: full or partial, hf_model.py:486-500); repeat_kv (hf_model.py:509-510); qk_bmm (a QMatMul:
    qk = (input, input2, output) QuantizerOracles) / sqrt(D); + causal mask; fp32 softmax; pv_bmm (pv = its three quantizers).
    Returns [S, heads*D] (the layout o_proj reads).
    exact_int: both contractions exact over the indices (_qmatmul_exact) instead of the reference's fp32 matmuls -- the arithmetic
    of the integer kernels WITHOUT their fast quantizer / exponential forms: what separates "the integer path differs from an fp32
    matmul by that matmul's own rounding" from "the kernel's approximations flipped an index" in the tests."""
    S = q.shape[0]
    D = q.shape[1] // heads
    qh = rope_partial(np.asarray(q, F32).reshape(S, heads, D).transpose(1, 0, 2), cos, sin)
    kh = rope_partial(np.asarray(k, F32).reshape(S, kv_heads, D).transpose(1, 0, 2), cos, sin)
    vh = np.asarray(v, F32).reshape(S, kv_heads, D).transpose(1, 0, 2)
    rep = heads // kv_heads
    kh, vh = np.repeat(kh, rep, axis=0), np.repeat(vh, rep, axis=0)
    if exact_int:
        att = _qmatmul_exact(qh, kh.transpose(0, 2, 1), *qk, double_scale=False) / F32(np.sqrt(F32(D)))
    else:
        att = qmatmul_sim(qh, kh.transpose(0, 2, 1), *qk) / F32(np.sqrt(F32(D)))
    mask = np.triu(np.full((S, S), -np.inf, dtype=F32), 1)
    att = (att + mask).astype(F32)
    att = att - att.max(axis=-1, keepdims=True)
    e = np.exp(att, dtype=F32)
    p = (e / e.sum(axis=-1, keepdims=True, dtype=F32)).astype(F32)
    out = _qmatmul_exact(p, vh, *pv, double_scale=True) if exact_int else qmatmul_sim(p, vh, *pv)
    return out.transpose(1, 0, 2).reshape(S, heads * D)

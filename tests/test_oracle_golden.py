"""Pins oracle/mq_oracle.py against the golden vectors produced by the real reference
(oracle/gen_golden.py).  Bit-exact unless a tolerance is written next to the assert."""
import hashlib
import json

import numpy as np
import pytest

from conftest import load_json, load_meta, load_npz
from oracle import mq_oracle as O

F32 = np.float32


def test_scale_offset_grid_scalar():
    g = load_npz("scale_offset_grid.npz")
    for mn, mx, bits, sym, s, o, qmin, qmax, imn, imx in g["grid"]:
        sc, of, a, b = O.scale_offset_from_min_max(mn, mx, int(bits), bool(sym))
        assert F32(sc) == F32(s) and F32(of) == F32(o), (mn, mx, bits, sym)
        assert np.signbit(of) == np.signbit(F32(o))          # symmetric offset is -0.0
        assert (a, b) == (int(qmin), int(qmax))
        rmn, rmx = O.min_max_from_scale_offset(sc, of, int(bits), bool(sym))
        assert F32(rmn) == F32(imn) and F32(rmx) == F32(imx)


def test_scale_offset_grid_tensor():
    g = load_npz("scale_offset_grid.npz")
    for bits in (4, 8, 16):
        for sym in (0, 1):
            s, o, _, _ = O.scale_offset_from_min_max(g["tmin"], g["tmax"], bits, bool(sym))
            assert np.array_equal(s, g[f"t_scale_b{bits}_s{sym}"])
            assert np.array_equal(o, g[f"t_offset_b{bits}_s{sym}"])


def _quantizer_from_meta(m, npz):
    qz = O.QuantizerOracle(m["bitwidth"], m["group_size"], m["is_symmetric"], m["is_per_channel"], m["is_dynamic"])
    if m["rng"] == "tensor":
        qz.set_from_minmax(npz[m["id"] + "_rmin"], npz[m["id"] + "_rmax"])
    elif m["rng"] is not None:
        qz.set_from_minmax(*m["rng"])
    return qz


def test_quantizer_cases_fp32_bit_exact():
    z = load_npz("quantizer_cases.npz")
    n = 0
    for m in load_meta(z):
        if m["dtype"] != "float32":
            continue
        k = m["id"]
        qz = _quantizer_from_meta(m, z)
        y, q = qz.forward(z[k + "_x"], return_index=True)
        assert (qz.qmin, qz.qmax) == (m["qmin"], m["qmax"])
        assert np.array_equal(np.asarray(qz.scale, F32).reshape(z[k + "_scale"].shape), z[k + "_scale"]), m["tag"]
        assert np.array_equal(np.asarray(qz.offset, F32).reshape(z[k + "_offset"].shape), z[k + "_offset"]), m["tag"]
        assert np.array_equal(q, z[k + "_q"]), m["tag"]
        assert np.array_equal(y.view(np.uint32), z[k + "_y"].view(np.uint32)), m["tag"]   # incl. sign of zero
        n += 1
    assert n == 39


def test_quantizer_cases_fp16():
    z = load_npz("quantizer_cases.npz")
    seen = set()
    for m in load_meta(z):
        if m["dtype"] != "float16":
            continue
        k = m["id"]
        x = z[k + "_x"]
        if m["is_per_channel"]:
            # [N,1] fp32 scale promotes the math to fp32; result cast back to half (qmodule.py:295)
            qz = _quantizer_from_meta(m, z)
            y = qz.forward(x.astype(F32)).astype(np.float16)
        else:
            s, o, qmin, qmax = O.scale_offset_from_min_max(*m["rng"], m["bitwidth"], m["is_symmetric"])
            y, q = O.fake_quant_f16_per_tensor(x, s, o, qmin, qmax)
            assert np.array_equal(q.astype(F32), z[k + "_q"]), m["tag"]
        assert np.array_equal(y.view(np.uint16), z[k + "_y"].view(np.uint16)), m["tag"]
        seen.add(m["tag"])
    assert len(seen) == 3


def test_quantizer_backward_matches_reference_autograd():
    """fp32 sums in a different order than torch: tolerance 1e-5 relative on the scale/offset gradients;
    grad_x is a masked copy and must be bit-exact."""
    z = load_npz("quantizer_grads.npz")
    for m in load_meta(z):
        k = m["id"]
        gx, gs, go = O.fake_quant_backward(z[k + "_x"], z[k + "_gy"], z[k + "_scale"], z[k + "_offset"], m["qmin"], m["qmax"])
        assert np.array_equal(gx, z[k + "_gx"]), k
        assert np.allclose(gs, z[k + "_gscale"], rtol=1e-5, atol=1e-4), (k, np.abs(gs - z[k + "_gscale"]).max())
        assert np.allclose(go, z[k + "_goffset"], rtol=1e-5, atol=1e-5), k


def _qlinear_oracle(m, z):
    k = m["id"]
    wq = O.QuantizerOracle(m["wbits"], -1, m["wsym"], m["wpc"])
    iq = None
    if m["in_cfg"] is not None:
        iq = O.QuantizerOracle(m["in_cfg"]["bitwidth"], -1, m["in_cfg"].get("is_symmetric", False))
        iq.set_from_minmax(*m["act"]["input"])
    oq = O.QuantizerOracle(m["out_bits"])
    oq.set_from_minmax(*m["act"]["output"])
    b = z[k + "_b"] if m["bias"] else None
    return wq, iq, oq, b


def test_qlinear_sim_cases():
    """fp32 matmul order differs between BLAS builds: the pre-output-quant value agrees to fp32
    round-off, so a post-output-quant element may sit one grid step away on rare near-tie elements.
    Tolerance: |diff| <= 1 output LSB everywhere; > 99.5 % of elements bit-identical for 8-bit
    outputs and > 90 % for 16-bit outputs (whose LSB is only ~50x the fp32 accumulation noise)."""
    z = load_npz("qlinear_cases.npz")
    for m in load_meta(z):
        k = m["id"]
        wq, iq, oq, b = _qlinear_oracle(m, z)
        y = O.qlinear_sim(z[k + "_x"], z[k + "_w"], b, wq, iq, oq)
        assert np.array_equal(np.asarray(wq.scale).reshape(z[k + "_wscale"].shape), z[k + "_wscale"]), m["tag"]
        assert np.array_equal(np.asarray(wq.offset).reshape(z[k + "_woffset"].shape), z[k + "_woffset"]), m["tag"]
        lsb = float(oq.scale)
        d = np.abs(y - z[k + "_y"])
        assert d.max() <= lsb * 1.01, (m["tag"], d.max(), lsb)     # 1 LSB (+ fp32 rounding of (q-o)*s)
        assert (d == 0).mean() > (0.995 if m["out_bits"] == 8 else 0.90), (m["tag"], (d == 0).mean())


def test_qlinear_dynamic_cases_oracle():
    """Dynamic activation quantizers (qmodule.py:262-277): the oracle re-derives the range from the tensor on every call, as the
    reference does; same 1-LSB / > 99.5 % (8-bit) / > 90 % (16-bit) bars as the static cases (BLAS summation order)."""
    z = load_npz("qlinear_dynamic_cases.npz")
    for m in load_meta(z):
        k = m["id"]
        wq = O.QuantizerOracle(8, -1, False, m["wpc"])
        iq = O.QuantizerOracle(8, is_dynamic=m["in_dyn"])
        oq = O.QuantizerOracle(m["out_bits"], is_dynamic=m["out_dyn"])
        if not m["in_dyn"]:
            iq.set_from_minmax(*m["act"]["input"])
        if not m["out_dyn"]:
            oq.set_from_minmax(*m["act"]["output"])
        y = O.qlinear_sim(z[k + "_x"], z[k + "_w"], z[k + "_b"] if m["bias"] else None, wq, iq, oq)
        lsb = float(z[k + "_oscale"])
        d = np.abs(y - z[k + "_y"])
        assert d.max() <= lsb * 1.01, (m["tag"], d.max(), lsb)
        assert (d == 0).mean() > (0.995 if m["out_bits"] == 8 else 0.90), (m["tag"], (d == 0).mean())


def test_qlinear_grouped_cases_oracle():
    """Per-group weight grids (qmodule.py:259-260, :292-293): the oracle's weight quantizer reshapes to [-1, group_size] as the reference
    does and reproduces its per-group scale / offset bit for bit; outputs within one LSB, > 99.5 % identical (BLAS summation order)."""
    z = load_npz("qlinear_grouped_cases.npz")
    for m in load_meta(z):
        k = m["id"]
        wq = O.QuantizerOracle(m["wbits"], m["gs"], m["sym"], True)
        iq, oq = O.QuantizerOracle(8), O.QuantizerOracle(8)
        iq.set_from_minmax(*m["act"]["input"])
        oq.set_from_minmax(*m["act"]["output"])
        y = O.qlinear_sim(z[k + "_x"], z[k + "_w"], z[k + "_b"] if m["bias"] else None, wq, iq, oq)
        assert np.array_equal(np.asarray(wq.scale, dtype=np.float32).reshape(-1), z[k + "_wscale"].reshape(-1)), m["tag"]
        assert np.array_equal(np.asarray(wq.offset, dtype=np.float32).reshape(-1) + 0.0, z[k + "_woffset"].reshape(-1) + 0.0), m["tag"]
        lsb = float(z[k + "_oscale"])
        d = np.abs(y - z[k + "_y"])
        assert d.max() <= lsb * 1.01, (m["tag"], d.max(), lsb)
        assert (d == 0).mean() > 0.995, (m["tag"], (d == 0).mean())


def test_qlinear_int_equivalence():
    """SURVEY 8a' item 9: the integer contraction reproduces the simulated path to fp32 round-off."""
    z = load_npz("qlinear_cases.npz")
    for m in load_meta(z):
        k = m["id"]
        wq, iq, oq, b = _qlinear_oracle(m, z)
        x = z[k + "_x"]
        if iq is not None:
            _, qa = iq.forward(x, return_index=True)
            sa, za = iq.scale, iq.offset
        else:
            sa, za = z[k + "_xscale"], z[k + "_xoffset"]
            qa = np.rint(x / sa) + za                          # x is already on the producer's grid
            assert np.array_equal(O.dequantize_index(qa, sa, za), x)
        _, qw = wq.forward(z[k + "_w"], return_index=True)
        acc, out = O.qlinear_int_exact(qa.reshape(-1, x.shape[-1]), za, sa, qw, np.asarray(wq.offset).reshape(-1),
                                       np.broadcast_to(np.asarray(wq.scale).reshape(-1), (qw.shape[0],)), b)
        y = oq.forward(out).reshape(z[k + "_y"].shape)
        lsb = float(oq.scale)
        d = np.abs(y - z[k + "_y"])
        assert d.max() <= lsb * 1.01, (m["tag"], d.max(), lsb)     # 1 LSB (+ fp32 rounding of (q-o)*s)
        assert (d == 0).mean() > (0.995 if m["out_bits"] == 8 else 0.90), (m["tag"], (d == 0).mean())


def _stream(z, prefix):
    items = {}
    for key in z.files:
        if key.startswith(prefix + "|"):
            _, name, field, idx = key.split("|")
            items.setdefault(int(idx), []).append((name, field, z[key]))
    return [items[i] for i in sorted(items)]


def test_act_range_per_tensor_and_sharded_merge():
    z = load_npz("calib_stream.npz")
    samples = _stream(z, "stream_pt")
    full = O.ActRangeOracle(False)
    for s in samples:
        for name, field, t in s:
            full.update(name, field, t)
    expected = {k: z[k] for k in z.files if k.startswith("pt|")}
    assert expected
    for k, v in expected.items():
        _, name, field = k.split("|")
        assert full.act_dict[name][field] == [float(v[0]), float(v[1])], k
    # shard the samples 2/3/6 ways round-robin, merge with min/max == unsharded
    for ws in (2, 3, 6):
        shards = []
        for r in range(ws):
            o = O.ActRangeOracle(False)
            for s in samples[r::ws]:
                for name, field, t in s:
                    o.update(name, field, t)
            shards.append(o.act_dict)
        assert O.ActRangeOracle.merge(shards, False) == full.act_dict


def test_act_range_per_channel():
    z = load_npz("calib_stream.npz")
    samples = _stream(z, "stream_pc")
    full = O.ActRangeOracle(True)
    for s in samples:
        for name, field, t in s:
            full.update(name, field, t)
    n = 0
    for k in z.files:
        if k.startswith("pc|"):
            _, name, field = k.split("|")
            assert np.array_equal(full.act_dict[name][field], z[k]), k
            n += 1
    assert n >= 8
    shards = []
    for r in range(2):
        o = O.ActRangeOracle(True)
        for s in samples[r::2]:
            for name, field, t in s:
                o.update(name, field, t)
        shards.append(o.act_dict)
    merged = O.ActRangeOracle.merge(shards, True)
    for name, fields in full.act_dict.items():
        for f, v in fields.items():
            assert np.array_equal(merged[name][f], v)


def test_absmax_stat_tensor():
    z = load_npz("calib_stream.npz")
    o = O.ActScaleOracle()
    for s in _stream(z, "stream_pt"):
        for name, field, t in s:
            if field in ("input", "output") and name in ("fc1", "fc2", "ln"):
                o.update(name, field, t)
    keys = [k for k in z.files if k.startswith("absmax|")]
    assert len(keys) == 6
    for k in keys:
        assert np.array_equal(o.act_scales[k.split("|")[1]], z[k]), k


def _sha(a):
    return hashlib.sha256(np.ascontiguousarray(a).tobytes()).hexdigest()


def test_full_size_checksums():
    """The oracle at BASELINE.json's full sizes, pinned by sha256 of the reference's indices."""
    cs = load_json("checksums.json")
    x = np.random.default_rng(1337).standard_normal((2048, 2048), dtype=F32)
    for bits, sym in ((8, False), (8, True), (16, False)):
        e = cs[f"act_2048x2048_b{bits}_s{int(sym)}"]
        s, o, qmin, qmax = O.scale_offset_from_min_max(e["rng"][0], e["rng"][1], bits, sym)
        q = O.quantize_index(x, s, o, qmin, qmax)
        assert _sha(q.astype(np.int32)) == e["q_sha256"]
        assert _sha(O.dequantize_index(q, s, o)) == e["y_sha256"]
    w = (np.random.default_rng(4242).standard_normal((5632, 2048), dtype=F32) * F32(0.02)).astype(F32)
    for bits, sym, pc in ((8, False, False), (8, False, True), (4, True, True), (4, False, True)):
        e = cs[f"w_5632x2048_b{bits}_s{int(sym)}_pc{int(pc)}"]
        qz = O.QuantizerOracle(bits, -1, sym, pc)
        y, q = qz.forward(w, return_index=True)
        assert _sha(q.astype(np.int32)) == e["q_sha256"]
        assert _sha(y) == e["y_sha256"]
        assert _sha(np.asarray(qz.scale, F32)) == e["scale_sha256"]
        assert _sha(np.asarray(qz.offset, F32)) == e["offset_sha256"]


def test_w4_pack_roundtrip():
    rng = np.random.default_rng(0)
    for qmin in (0, -8):
        q = rng.integers(qmin, qmin + 16, size=(24, 96))
        p = O.pack_w4(q, qmin)
        assert p.shape == (24, 48) and p.dtype == np.uint8
        assert np.array_equal(O.unpack_w4(p, qmin), q)


def eq_nan(a, b):
    """Equal where finite / infinite (bit-exact), NaN exactly where the reference has NaN."""
    a, b = np.asarray(a, F32), np.asarray(b, F32)
    na, nb = np.isnan(a), np.isnan(b)
    return a.shape == b.shape and np.array_equal(na, nb) and np.array_equal(a[~na], b[~nb])


def test_nonfinite_inputs_follow_reference():
    """NaN propagates through clamp / amin / amax; +-inf becomes NaN inside round_ste (torch semantics)."""
    z = load_npz("nonfinite_cases.npz")
    x = z["x"]
    assert np.isnan(x).sum() == 3 and np.isinf(x).sum() == 3
    for m in load_meta(z):
        qz = O.QuantizerOracle(m["bitwidth"], -1, m["is_symmetric"], m["is_per_channel"], False)
        if m["is_per_channel"]:
            qz.set_from_minmax(z[m["id"] + "_rmin"], z[m["id"] + "_rmax"])
        else:
            qz.set_from_minmax(*m["rng"])
        y = qz.forward(x)
        assert eq_nan(y, z[m["id"] + "_y"]), m
        assert np.array_equal(np.isnan(y), ~np.isfinite(x))      # round_ste turns +-inf into NaN (inf - inf)
    xs, xi = z["xs"], z["xi"]
    mn, mx = O.min_max_from_tensor(xs)
    assert eq_nan(mn, z["xs_t_min"]) and eq_nan(mx, z["xs_t_max"]) and np.isnan(mn) and np.isnan(mx)
    mn, mx = O.min_max_from_tensor(xs, True)
    assert eq_nan(mn, z["xs_r_min"]) and eq_nan(mx, z["xs_r_max"])
    assert eq_nan(xs.min(0), z["xs_c_min"]) and eq_nan(xs.max(0), z["xs_c_max"])
    mn, mx = O.min_max_from_tensor(xi)
    assert mn == z["xi_t_min"] == -np.inf and mx == z["xi_t_max"] == np.inf
    for sym in (0, 1):
        s, o, _, _ = O.scale_offset_from_min_max(z["so_min"], z["so_max"], 8, bool(sym))
        assert eq_nan(s, z[f"so_scale_s{sym}"]) and eq_nan(o, z[f"so_offset_s{sym}"]), sym


def _norm_quantizers(m, z):
    k = m["id"]
    in_q = out_q = None
    if m["in_bits"]:
        in_q = O.QuantizerOracle(m["in_bits"]); in_q.set_from_minmax(*m["act"]["input"])
    if m["out_bits"]:
        out_q = O.QuantizerOracle(m["out_bits"]); out_q.set_from_minmax(*m["act"]["output"])
    w_q = O.QuantizerOracle(16)           # first forward: range from the weight itself (qmodule.py:262-277)
    return in_q, w_q, out_q


def norm_close(got, want, m):
    """Bound for QRMSNorm against the reference's frozen output: identical up to the summation order of mean(x^2).
    Quantised output: at most 1 LSB apart, > 99.9 % identical; float output: 2e-6 relative."""
    got, want = np.asarray(got, F32), np.asarray(want, F32)
    if m["out_bits"]:
        lo, hi = m["act"]["output"]
        lsb = F32((hi - lo) / (2 ** m["out_bits"] - 1))
        d = np.abs(got - want)
        return d.max() <= lsb * F32(1.01) and (d == 0).mean() > 0.999
    return np.allclose(got, want, rtol=2e-6, atol=1e-7)


def test_qrmsnorm_cases():
    z = load_npz("qrmsnorm_cases.npz")
    for m in load_meta(z):
        k = m["id"]
        in_q, w_q, out_q = _norm_quantizers(m, z)
        if m.get("layernorm"):
            y = O.qlayernorm(z[k + "_x"], z[k + "_w"], z[k + "_b"], m["eps"], in_q, w_q, out_q)
        else:
            y = O.qrmsnorm(z[k + "_x"], z[k + "_w"], None, m["eps"], in_q, w_q, out_q)
        assert np.array_equal(np.asarray(w_q.scale, F32).reshape(z[k + "_wscale"].shape), z[k + "_wscale"])
        assert y.shape == z[k + "_y"].shape and norm_close(y, z[k + "_y"], m), m


def act_close(got, want, m):
    """QSiLU / QGELU bound: exp / erf differ by an ulp or two between math libraries; after the output quantizer the
    results are at most 1 LSB apart (> 99.9 % identical on an 8-bit grid, > 99 % on a 16-bit one); 3e-6 relative as floats."""
    got, want = np.asarray(got, F32), np.asarray(want, F32)
    if m["out_bits"]:
        lo, hi = m["act"]["output"]
        lsb = F32((hi - lo) / (2 ** m["out_bits"] - 1))
        d = np.abs(got - want)
        return d.max() <= lsb * F32(1.01) and (d == 0).mean() > (0.999 if m["out_bits"] <= 8 else 0.99)
    return np.allclose(got, want, rtol=3e-6, atol=2e-6)      # 1 + erf cancels in the negative tail: absolute, not relative


def _act_quantizers(m):
    mk = lambda bits, rng: None if not bits else (lambda q: (q.set_from_minmax(*rng), q)[1])(O.QuantizerOracle(bits))   # noqa: E731
    in_q = mk(m["in_bits"], m["act"]["input"])
    out_q = mk(m["out_bits"], m["act"]["output"])
    mid_q = mk(8, m["act"].get("input2", [0.0, 1.0]))       # qmodule.py:731-734: sigmoid grid defaults to [0, 1]
    return in_q, mid_q, out_q


def test_qsilu_qgelu_cases():
    z = load_npz("qact_cases.npz")
    n = 0
    for m in load_meta(z):
        in_q, mid_q, out_q = _act_quantizers(m)
        y = O.qsilu(z["x"], in_q, mid_q, out_q) if m["kind"] == "silu" else O.qgelu(z["x"], in_q, out_q)
        assert act_close(y, z[m["id"] + "_y"], m), m
        n += 1
    assert n == 8


# ---- round 2: LWC, QMatMul -----------------------------------------------------------------------------------------
def test_lwc_oracle_matches_reference_forward_and_autograd():
    """a7: forward values, run_lwc and the gradients to upbound_factor / lowbound_factor / the weight (incl. the amin / amax
    path) against the reference's autograd; torch's vectorised sigmoid may differ from exp-based fp32 by an ulp, hence 1e-6."""
    z = load_npz("lwc_cases.npz")
    for m in load_meta(z):
        t = m["id"]
        args = (m["bitwidth"], m["is_symmetric"], m["is_per_channel"])
        y, s, o = O.lwc_forward(z[t + "_w"], z[t + "_up"], z[t + "_lo"], *args)
        assert np.allclose(s.reshape(-1), z[t + "_scale"].reshape(-1), rtol=3e-7, atol=0)
        assert np.array_equal(o.reshape(-1), z[t + "_offset"].reshape(-1))
        assert np.abs(y - z[t + "_y"]).max() <= 4e-8 + 1e-6 * np.abs(z[t + "_y"]).max()
        assert np.abs(O.run_lwc(z[t + "_w"], z[t + "_up"], z[t + "_lo"], m["is_per_channel"]) - z[t + "_clamped"]).max() <= 4e-8
        gu, gl, gw = O.lwc_backward(z[t + "_w"], z[t + "_up"], z[t + "_lo"], z[t + "_gy"], *args)
        for got, want in ((gu, z[t + "_g_up"]), (gl, z[t + "_g_lo"]), (gw, z[t + "_g_w"])):
            assert np.abs(got.reshape(-1) - want.reshape(-1)).max() <= 2e-6 * np.abs(want).max(), t


def test_qmatmul_oracle_matches_reference():
    z = load_npz("qmatmul_cases.npz")
    for m in load_meta(z):
        t = m["id"]
        qs = []
        for bits, rng in zip(m["bits"], (m["act"]["input"], m["act"]["input2"], m["act"]["output"])):
            q = O.QuantizerOracle(bits)
            q.set_from_minmax(*rng)
            qs.append(q)
        y = O.qmatmul_sim(z[t + "_a"], z[t + "_b"], *qs)
        d = np.abs(y - z[t + "_y"])
        assert d.max() <= float(qs[2].scale) * 1.001 and (d == 0).mean() > 0.99, (t, d.max(), (d == 0).mean())


def test_exact_integer_qmatmul_oracle_sits_within_one_step_of_the_reference():
    """oracle.qmatmul_exact -- the arithmetic of the standalone integer QMatMul kernel (mq_qmatmul: exact contraction over the
    indices, ONE rounding) -- against the reference's QMatMul.forward outputs (fp32 matmul of the dequantised operands, rounding per
    product; qmodule.py:453-466): never more than one step of the output grid apart, identical on > 99 % of the outputs."""
    z = load_npz("qmatmul_cases.npz")
    for m in load_meta(z):
        t = m["id"]
        qs = []
        for bits, rng in zip(m["bits"], (m["act"]["input"], m["act"]["input2"], m["act"]["output"])):
            q = O.QuantizerOracle(bits)
            q.set_from_minmax(*rng)
            qs.append(q)
        y = O.qmatmul_exact(z[t + "_a"], z[t + "_b"], *qs)
        d = np.abs(y - z[t + "_y"])
        assert d.max() <= float(qs[2].scale) * 1.001 and (d == 0).mean() > 0.99, (t, d.max(), (d == 0).mean())


def test_attention_sim_without_quantizers_is_causal_sdpa():
    """oracle.attention_sim (hf_model.py:486-534 restated) with every quantizer absent == torch's causal attention on the RoPE'd heads."""
    import torch
    rng = np.random.default_rng(0)
    S, H, KV, D = 48, 4, 2, 32
    q, k, v = (rng.standard_normal((S, n * D), dtype=np.float32) for n in (H, KV, KV))
    inv = 1.0 / (10000.0 ** (np.arange(0, D, 2, dtype=np.float32) / D))
    ang = np.outer(np.arange(S, dtype=np.float32), inv).astype(np.float32)
    ang = np.concatenate((ang, ang), -1)
    cos, sin = np.cos(ang).astype(np.float32), np.sin(ang).astype(np.float32)
    got = O.attention_sim(q, k, v, cos, sin, H, KV, (None, None, None), (None, None, None))
    t = lambda a, n: torch.from_numpy(a).view(S, n, D).transpose(0, 1)      # noqa: E731
    rope = lambda x: x * torch.from_numpy(cos) + torch.cat((-x[..., D // 2:], x[..., :D // 2]), -1) * torch.from_numpy(sin)    # noqa: E731
    qh, kh, vh = rope(t(q, H)), rope(t(k, KV)).repeat_interleave(H // KV, 0), t(v, KV).repeat_interleave(H // KV, 0)
    want = torch.nn.functional.scaled_dot_product_attention(qh, kh, vh, is_causal=True).transpose(0, 1).reshape(S, H * D).numpy()
    assert np.allclose(got, want, atol=2e-6, rtol=1e-5)

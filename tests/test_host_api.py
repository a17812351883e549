"""Host-side logic of the qmodule mirror (runs without a GPU): config (de)serialisation, scalar range
math, model surgery by name rules, artefact formats -- pinned to fixtures frozen from the reference."""
import json

import numpy as np
import pytest
import torch
import torch.nn as nn

import mobilequant_amd.quantization.qmodule as Q
from conftest import load_json, load_npz
from toy_models import ToyLM, apply_mixed_precision


def test_quantconfig_roundtrip_uses_string_fields():
    c = Q.QuantConfig(bitwidth=4, group_size=32, is_symmetric=True, is_per_channel=True)
    d = c.to_dict()
    assert d == {"bitwidth": "4", "group_size": "32", "is_symmetric": "True", "is_per_channel": "True", "is_dynamic": "False"}
    assert Q.QuantConfig.from_dict(d) == c
    assert Q.QuantConfig.from_dict({**d, "is_symmetric": "true"}).is_symmetric
    assert Q.QuantConfig() == Q.QuantConfig(32, -1, False, False, False)


def test_host_scale_offset_matches_reference_grid():
    g = load_npz("scale_offset_grid.npz")
    for mn, mx, bits, sym, s, o, qmin, qmax, imn, imx in g["grid"]:
        sc, of, _, _, a, b = Q.compute_scale_offset_from_min_max(float(mn), float(mx), int(bits), bool(sym))
        assert sc.dtype == torch.float32 and np.float32(sc.item()) == np.float32(s) and np.float32(of.item()) == np.float32(o)
        assert (a, b) == (int(qmin), int(qmax))
        lo, hi = Q.compute_min_max_from_scale_offset(sc, of, int(bits), bool(sym))
        assert np.float32(lo.item()) == np.float32(imn) and np.float32(hi.item()) == np.float32(imx)
    for bits in (4, 8, 16):
        for sym in (0, 1):
            sc, of, *_ = Q.compute_scale_offset_from_min_max(torch.from_numpy(g["tmin"]), torch.from_numpy(g["tmax"]), bits, bool(sym))
            assert np.array_equal(sc.numpy(), g[f"t_scale_b{bits}_s{sym}"]) and np.array_equal(of.numpy(), g[f"t_offset_b{bits}_s{sym}"])


def test_quantizer_state_and_bypass():
    q = Q.Quantizer(Q.QuantConfig(bitwidth=8))
    q.set_scale_offset_from_minmax(-1.0, 2.0, "parameter")
    assert sorted(q.state_dict()) == ["offset", "scale"] and isinstance(q.scale, nn.Parameter)
    assert (q.qmin, q.qmax) == (0, 255) and q.offset.item() == 85.0
    q.update_qcfg({"bitwidth": "16", "group_size": "-1", "is_symmetric": "False", "is_per_channel": "False", "is_dynamic": "False"})
    assert not hasattr(q, "scale") and q.qcfg.bitwidth == 16          # new config drops the cached grid
    q.set_scale_offset_from_minmax(-1.0, 2.0, "buffer")
    assert sorted(q.state_dict()) == ["offset", "scale"] and not isinstance(q.scale, nn.Parameter)
    x = torch.randn(3, 3)
    for bypass in (Q.Quantizer(Q.QuantConfig(bitwidth=32)), Q.Quantizer(Q.QuantConfig(bitwidth=17))):
        assert bypass(x) is x
    q.enable = False
    assert q(x) is x
    q.enable = True
    with pytest.raises(RuntimeError, match="ROCm device tensor"):
        q(x)                                                           # no CPU path, loudly


def _sim_toy():
    m = ToyLM().eval()
    z = load_npz("toy_lm.npz")
    m.load_state_dict({k[3:]: torch.from_numpy(z[k]) for k in z.files if k.startswith("sd|")})
    Q.create_sim_qmodel(m, Q.QuantConfig(bitwidth=8), Q.QuantConfig(bitwidth=8))
    apply_mixed_precision(m, Q)
    return m, z


def test_surgery_matches_reference_module_map_and_qcfg():
    surf = load_json("api_surface.json")
    m, _ = _sim_toy()
    got = {n: type(mm).__name__ for n, mm in m.named_modules() if n}
    want = {k: v.lstrip("_") for k, v in surf["module_types"].items()}      # fixture container classes are _Block / _ToyLM
    # the reference's op shims appear as extra children of its HFRMSNorm (elementwisemul); ours match by name too
    assert got == want
    assert Q.export_qcfg(m) == surf["qcfg"]
    # q/k/v/o/w1/w3 and SiLU lost their input quantizer; lm_head and the final norm stay float
    assert m.layers[0].q_proj.input_quantizer is None and m.layers[0].w2.input_quantizer is not None
    assert type(m.lm_head) is nn.Linear and type(m.norm).__name__ == "HFRMSNorm"


def test_qcfg_json_roundtrip_and_act_range_export():
    surf = load_json("api_surface.json")
    m, _ = _sim_toy()
    qcfg = json.loads(json.dumps(Q.export_qcfg(m)))
    m2 = ToyLM().eval()
    Q.create_sim_qmodel(m2)                      # default placeholders (bitwidth 32)
    Q.update_qcfg(m2, qcfg)
    assert Q.export_qcfg(m2) == surf["qcfg"]
    Q.set_scale_and_offset(m2, surf["act_dict"], "buffer")
    exported = Q.export_act_range(m2)
    assert exported.keys() == surf["exported_act_range"].keys()
    for name, fields in surf["exported_act_range"].items():
        assert exported[name].keys() == fields.keys(), name
        for f, (lo, hi) in fields.items():
            assert exported[name][f] == [lo, hi], (name, f)
    keys = sorted(k for k in m2.state_dict().keys())
    want = [k for k in surf["state_dict_keys"] if "weight_quantizer" not in k]   # weight grids appear on first forward
    assert keys == want
    with pytest.raises(AssertionError):
        Q.set_scale_and_offset(m2, {}, "buffer")   # same `assert name in act_dict` behaviour


def test_create_fp_model_restores_float_leaves():
    m, _ = _sim_toy()
    Q.create_fp_model(m)
    kinds = {type(mm).__name__ for _, mm in m.named_modules()}
    assert not any(k.startswith("Q") for k in kinds)
    assert type(m.layers[0].w1) is nn.Linear and type(m.layers[0].act_fn) is nn.SiLU


def test_wire_integer_inputs_uses_calibrated_input_range():
    surf = load_json("api_surface.json")
    m, _ = _sim_toy()
    Q.set_scale_and_offset(m, surf["act_dict"], "buffer")
    n = Q.wire_integer_inputs(m, 8, False)
    assert n == 2 * 6                                          # q,k,v,o,w1,w3 per block
    g = m.layers[0].q_proj._input_grid
    lo, hi = surf["act_dict"]["layers.0.q_proj"]["input"]
    s, o, *_ = Q.compute_scale_offset_from_min_max(lo, hi, 8, False)
    assert g.scale.item() == s.item() and g.offset.item() == o.item()
    # the producer (input_layernorm) quantizes its output to exactly this grid
    p = m.layers[0].input_layernorm.output_quantizer
    assert p.scale.item() == g.scale.item() and p.offset.item() == g.offset.item()


def test_reference_import_path_alias():
    """the names the reference's callers import from mobilellm.quantization.qmodule resolve to this package."""
    import sys
    import mobilequant_amd as mq
    saved = {k: v for k, v in sys.modules.items() if k == "mobilellm" or k.startswith("mobilellm.")}
    try:
        mq.install_reference_alias()
        from mobilellm.quantization.qmodule import (QuantConfig, Quantizer, QLinear, QRMSNorm, QLayerNorm, QMatMul, QSiLU,  # noqa: F401
                                                    QGELU, create_fp_model, export_act_range, create_sim_qmodel,
                                                    create_weight_only_qmodel, set_scale_and_offset, update_qcfg, export_qcfg)
        import mobilellm.quantization.qmodule as alias
        assert alias is mq.quantization.qmodule and QLinear is mq.QLinear
        assert QuantConfig(bitwidth=8).to_dict()["bitwidth"] == "8"
    finally:
        for k in [k for k in sys.modules if k == "mobilellm" or k.startswith("mobilellm.")]:
            del sys.modules[k]
        sys.modules.update(saved)


def test_torch_library_registration_and_meta_kernels():
    """SURVEY 8b "native boundary": the core entry points are registered as torch.ops.mobilequant_amd.* with FakeTensor (meta)
    kernels -- shapes / dtypes without a device -- and NO CPU implementation: a CPU tensor finds nothing to dispatch to."""
    import torch
    import mobilequant_amd.torch_ops as T
    ns = torch.ops.mobilequant_amd
    for name in T.OPS:
        assert hasattr(ns, name), name
    x = torch.empty(6, 64, device="meta")
    s = torch.empty(1, device="meta")
    assert ns.fake_quant(x, s, s, 0.0, 255.0).shape == (6, 64)
    q, rs = ns.quantize(x, s, s, 0.0, 255.0, 128)
    assert q.dtype == torch.int8 and q.shape == (6, 64) and rs.dtype == torch.int32 and rs.shape == (6,)
    assert ns.minmax(x, False).shape == (2,) and ns.minmax(x, True).shape == (2, 64)
    xq, wq = torch.empty(6, 64, dtype=torch.int8, device="meta"), torch.empty(32, 64, dtype=torch.int8, device="meta")
    v, vi = torch.empty(32, device="meta"), torch.empty(32, dtype=torch.int32, device="meta")
    assert ns.w8a8_linear(xq, rs, wq, v, vi, vi).shape == (6, 32)
    packed = ns.pack_w4(torch.empty(32, 64, dtype=torch.uint8, device="meta"))
    assert packed.shape == (32, 32) and ns.w4a8_linear(xq, rs, packed, v, vi, vi).dtype == torch.float32
    import pytest
    with pytest.raises(NotImplementedError):
        ns.fake_quant(torch.zeros(4, 8), torch.ones(1), torch.zeros(1), 0.0, 255.0)


def test_committed_fixtures_are_what_the_generator_writes():
    """VERDICT r03 "What's weak" 9: three fixtures had drifted from oracle/gen_golden.py without a test noticing.  When the reference
    checkout is present (the build container), the QUICK list of generators is re-run into a temp dir and every file compared
    with tests/golden/ bit for bit (tools/check_golden.py; `--all` covers the BASELINE-sized ones, minutes).  Skipped where the
    reference does not exist (the GPU box)."""
    import importlib.util
    import os
    import pytest
    ref = os.environ.get("MQ_REFERENCE", "/root/reference")
    if not os.path.isdir(os.path.join(ref, "mobilellm")):
        pytest.skip("no reference checkout")
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    spec = importlib.util.spec_from_file_location("check_golden", os.path.join(root, "tools", "check_golden.py"))
    mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mod)
    # every generator is listed (a new one cannot be forgotten), and the whole quick list runs -- dependent generators behind their
    # producers
    import re
    src = open(os.path.join(root, "oracle", "gen_golden.py")).read()
    gens = set(re.findall(r"^def (gen_\w+)\(", src, re.M))
    assert gens == set(mod.QUICK) | set(mod.SLOW), gens ^ (set(mod.QUICK) | set(mod.SLOW))
    assert all(d in gens for deps in mod.DEPS.values() for d in deps)
    assert mod.run(mod.QUICK, ref) == 0

"""mobilequant_amd/llama.py (the minimal decoder with the reference's leaf-module graph) against the reference's REAL model
classes: fp32 logits of a 2-layer HFForCausalLM frozen by oracle/gen_golden.py (smooth_cases.npz).  CPU only."""
import json

import numpy as np
import pytest
import torch

from conftest import load_npz
from mobilequant_amd.llama import LlamaForCausalLM, LlamaShape


def llama_from_fixture(z, tag, prefix="sd"):
    meta = json.loads(str(z["meta"]))[tag]
    shape = LlamaShape(hidden=64, layers=2, heads=4, kv_heads=meta["kv_heads"], head_dim=16, ffn=96, vocab=50, eps=1e-5, max_pos=64)
    m = LlamaForCausalLM(shape).eval()
    sd = {}
    for k in z.files:
        if k.startswith(f"{tag}|{prefix}|"):
            name = k.split("|", 2)[2]
            name = name[len("model."):] if name.startswith("model.") else name
            if "rotary_emb" in name:
                continue
            sd[name] = torch.from_numpy(z[k])
    missing, unexpected = m.load_state_dict(sd, strict=False)
    assert not unexpected and all("cos" in k or "sin" in k for k in missing), (missing, unexpected)
    return m


@pytest.mark.parametrize("tag", ["gqa", "mha"])
def test_llama_forward_equals_reference_hf_model(tag):
    z = load_npz("smooth_cases.npz")
    m = llama_from_fixture(z, tag)
    ids = torch.from_numpy(z[tag + "_ids"][0:1]).long()
    with torch.no_grad():
        got = m(ids).numpy()
        want = z[tag + "_logits_fp"]
        assert got.shape == want.shape
        assert np.abs(got - want).max() <= 2e-4 * np.abs(want).max(), np.abs(got - want).max()
        # static KV cache: prefill 16 tokens, then 8 single-token steps == the full forward (sim_model.py:160-221 semantics)
        cache = m.new_cache(1, 32)
        outs = [m(ids[:, :16], cache=cache, pos=0)] + [m(ids[:, t:t + 1], cache=cache, pos=t) for t in range(16, 24)]
        inc = torch.cat(outs, dim=1).numpy()
        assert np.abs(inc - want).max() <= 2e-4 * np.abs(want).max()


def test_llama_has_the_reference_leaf_names():
    names = [n for n, _ in LlamaForCausalLM(LlamaShape.toy()).named_modules()]
    for leaf in ("layers.0.self_attn.q_proj", "layers.0.self_attn.k_proj", "layers.0.self_attn.v_proj", "layers.0.self_attn.o_proj",
                 "layers.0.self_attn.qk_bmm", "layers.0.self_attn.pv_bmm", "layers.0.mlp.w1", "layers.0.mlp.w2", "layers.0.mlp.w3",
                 "layers.0.mlp.act_fn", "layers.0.input_layernorm", "layers.0.post_attention_layernorm", "norm", "lm_head"):
        assert leaf in names, leaf


def test_reference_artifact_files_are_reproduced_byte_for_byte(tmp_path):
    """a14: act_dict.json written by save_act_dict == the file the reference's json_save wrote from the same dictionary
    (tests/golden/act_dict_ref.json); act_scales.pth round-trips with the reference's structure (CPU fp32 tensors by
    "<module>_<field>")."""
    import os
    from conftest import GOLDEN, load_json
    from mobilequant_amd import smoothquant as S
    from mobilequant_amd.calibration import save_act_dict
    act = load_json("api_surface.json")["act_dict"]
    out = tmp_path / "act_dict.json"
    save_act_dict(str(out), act)
    assert out.read_bytes() == open(os.path.join(GOLDEN, "act_dict_ref.json"), "rb").read()
    ref = torch.load(os.path.join(GOLDEN, "act_scales_ref.pth"), map_location="cpu")
    p2 = tmp_path / "act_scales.pth"
    S.save_act_scales(str(p2), {k: v.clone() for k, v in ref.items()})
    back = S.load_act_scales(str(p2))
    assert list(back) == list(ref) and all(torch.equal(back[k], ref[k]) and back[k].dtype == torch.float32 and back[k].device.type == "cpu" for k in ref)
    pc = {"m": {"input": torch.arange(8.).reshape(2, 4), "output": torch.ones(2, 3)}}
    p3 = tmp_path / "act_dict_per_channel.pth"
    S.save_act_dict_per_channel(str(p3), pc)
    back = torch.load(str(p3), map_location="cpu")
    assert back.keys() == pc.keys() and torch.equal(back["m"]["input"], pc["m"]["input"]) and back["m"]["output"].shape == (2, 3)


FAMILY_SHAPES = {
    # BASELINE.json configs[2] / [3]: the reference's HFConfig switches for StableLM-2 and Gemma (hf_config.py:101-179) at toy size
    "stablelm": dict(hidden=256, layers=2, heads=4, kv_heads=4, head_dim=64, ffn=512, vocab=96, eps=1e-5, max_pos=64,
                     norm="layernorm", qkv_bias=True, rotary_pct=0.25),
    "gemma": dict(hidden=256, layers=2, heads=2, kv_heads=1, head_dim=256, ffn=512, vocab=96, eps=1e-5, max_pos=64, hidden_act="gelu",
                  embed_scale=True),
}


@pytest.mark.parametrize("tag", ["stablelm", "gemma"])
def test_other_model_families_equal_the_reference_hf_model(tag):
    """LlamaShape's family switches (LayerNorm + q|k|v bias + partial rotary; explicit head_dim + GeGLU + scaled embeddings) against
    the fp32 logits of the reference's HFForCausalLM under the matching HFConfig (tests/golden/decode_case_<tag>.npz; weights from
    tests/seeded.py on both sides), full forward and prefill + single-token steps over the static KV cache."""
    from seeded import seeded_parameters_
    z = load_npz(f"decode_case_{tag}.npz")
    m = LlamaForCausalLM(LlamaShape(**FAMILY_SHAPES[tag])).eval()
    seeded_parameters_(m, std=0.08)
    ids = torch.from_numpy(z["ids"]).long()[None]
    want = z["logits_fp"]
    with torch.no_grad():
        got = m(ids).numpy()
        assert got.shape == want.shape and np.abs(got - want).max() <= 2e-4 * np.abs(want).max(), np.abs(got - want).max()
        cache = m.new_cache(1, 64)
        outs = [m(ids[:, :16], cache=cache, pos=0)] + [m(ids[:, t:t + 1], cache=cache, pos=t) for t in range(16, 40)]
        assert np.abs(torch.cat(outs, dim=1).numpy() - want).max() <= 2e-4 * np.abs(want).max()
    names = dict(m.named_modules())
    assert isinstance(names["layers.0.input_layernorm"], torch.nn.LayerNorm) == (tag == "stablelm")
    assert (names["layers.0.self_attn.q_proj"].bias is not None) == (tag == "stablelm") and names["layers.0.self_attn.o_proj"].bias is None


def test_declared_calibration_alias_groups_name_real_slots():
    """LlamaForCausalLM.calibration_alias_groups() (round 6: tensors hooked under several names are reduced once) must name hooked
    (module, field) pairs of the graph as named_modules() spells them -- a renamed submodule would silently switch the mirroring off --
    and so must the parts the one-pass calibration kernels stand for (_mq_calibration_layer_parts)."""
    import torch
    from mobilequant_amd import llama
    from mobilequant_amd.calibration import ActRangeCollector
    for fam in ("tinyllama", "stablelm_2_1_6b", "gemma_2b"):
        shape = getattr(llama.LlamaShape, fam)(layers=2, max_pos=32, vocab=64)
        model = llama.LlamaForCausalLM(shape)
        col = ActRangeCollector(model, per_channel=False, device="cpu")
        groups = model.calibration_alias_groups()
        assert len(groups) == 4 * shape.layers + 1
        for grp in groups:
            assert len(grp) >= 2 and all(k in col.slots for k in grp), (fam, grp)
        assert len(col._mirror_pending) == len(groups)
        names = dict(model.named_modules())
        for name, m in names.items():
            for part in getattr(m, "_mq_calibration_layer_parts", ()):
                q = m
                for piece in part.split("."):
                    q = getattr(q, piece)
                key = next(n for n, mm in names.items() if mm is q)
                assert (key, "input") in col.slots and (key, "output") in col.slots, (fam, name, part)

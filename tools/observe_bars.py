"""Observed model-level deviations (for setting the test bars at ~2x what is measured).  GPU."""
import json, sys, os
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__)))); sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests"))
import test_gpu_round2 as T2
from conftest import load_npz
from mobilequant_amd.decode import DecodeEngine
from mobilequant_amd import llama
import mobilequant_amd as mq
dev = torch.device("cuda:0")
def stats(name, d, span):
    d = np.abs(d) / span
    print(f"{name:38s} max {d.max():.5f} p99 {np.quantile(d, 0.99):.5f} median {np.median(d):.2e} within1% {(d <= 0.01).mean():.4f}")
m, z = T2._decode_model(dev)
ids = torch.from_numpy(z["ids"]).long(); ref = z["logits_w8a8"][0]; span = float(np.ptp(z["logits_fp"]))
with torch.no_grad(): pre = m(ids[None].to(dev))[0].cpu().numpy()
eng = DecodeEngine(m, cache_len=64)
eager = np.stack([eng.step(int(t)).cpu().numpy().copy() for t in ids])
stats("decode vs reference", eager - ref, span); stats("decode vs own prefill", eager - pre, span); stats("prefill chain vs reference", pre - ref, span)
print("argmax agreement decode/ref", (eager.argmax(-1) == ref.argmax(-1)).mean(), "prefill/ref", (pre.argmax(-1) == ref.argmax(-1)).mean())
with torch.no_grad():
    llama.fuse_decoder_layer(m); fused = m(ids[None].to(dev))[0].cpu().numpy()
stats("fused prefill vs reference", fused - ref, span); print("argmax fused/ref", (fused.argmax(-1) == ref.argmax(-1)).mean())
from seeded import seeded_parameters_
from test_llama_host import FAMILY_SHAPES
for tag, wbits, kv, act in [("w4", 4, 2, "silu"), ("w8pc_mha", 8, 4, "silu"), ("w4_geglu_mqa", 4, 1, "gelu"), ("stablelm", 8, 4, "silu"), ("gemma", 4, 1, "gelu")]:
    z = load_npz(f"decode_case_{tag}.npz")
    kw = FAMILY_SHAPES.get(tag) or dict(hidden=256, layers=2, heads=4, kv_heads=kv, head_dim=64, ffn=512, vocab=96, eps=1e-5, max_pos=64, hidden_act=act)
    m = llama.LlamaForCausalLM(llama.LlamaShape(**kw)).eval(); seeded_parameters_(m, std=0.08); m = m.to(dev)
    strip = lambda d: {(k[len("model."):] if k.startswith("model.") else k): v for k, v in d.items()}
    mq.create_sim_qmodel(m, mq.QuantConfig(bitwidth=8), mq.QuantConfig(bitwidth=8)); mq.update_qcfg(m, strip(json.loads(str(z["qcfg"]))))
    mq.set_scale_and_offset(m, strip(json.loads(str(z["act"]))), "buffer"); mq.wire_integer_inputs(m); m.requires_grad_(False)
    ids = torch.from_numpy(z["ids"]).long(); ref, span = z["logits_w4a8"][0], float(np.ptp(z["logits_fp"]))
    noise = float(np.abs(z["logits_w4a8"] - z["logits_fp"]).max()) / span
    with torch.no_grad(): chain = m(ids[None].to(dev))[0].cpu().numpy()
    eng = DecodeEngine(m, cache_len=64); steps = np.stack([eng.step(int(t)).cpu().numpy().copy() for t in ids])
    with torch.no_grad(): llama.fuse_decoder_layer(m); fused = m(ids[None].to(dev))[0].cpu().numpy()
    print(f"-- {tag}: quantisation noise {noise:.4f} of span")
    for name, got in (("chain", chain), ("decode", steps), ("fused", fused)):
        stats(f"{tag} {name} vs reference", got - ref, span)
    print("   argmax agreement", [(float((g.argmax(-1) == ref.argmax(-1)).mean())) for g in (chain, steps, fused)])

#!/usr/bin/env python3
"""A few eager decode steps of the TinyLlama-shaped DecodeEngine (for rocprofv3 --kernel-trace --stats)."""
import os, sys, time
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench  # noqa: E402

dev = torch.device("cuda:0")
layers = int(os.environ.get("LAYERS", "4"))
import mobilequant_amd as mq
from mobilequant_amd.calibration import get_act_range
from mobilequant_amd.decode import DecodeEngine
from mobilequant_amd.llama import LlamaForCausalLM, LlamaShape
shape = LlamaShape.tinyllama(max_pos=2048, layers=layers)
model = LlamaForCausalLM(shape); model.reset_parameters(seed=1); model = model.to(dev).eval().requires_grad_(False)
g = torch.Generator().manual_seed(1)
act = get_act_range(model, [torch.randint(3, shape.vocab, (1, 256), generator=g)])
a8 = mq.QuantConfig(bitwidth=8)
wbits = int(os.environ.get("WBITS", "8"))                # 4: packed 4-bit per-channel weights (the W4A8 recipe of bench.py)
mq.create_sim_qmodel(model, a8 if wbits == 8 else mq.QuantConfig(bitwidth=wbits, is_per_channel=True), a8)
for name, mod in model.named_modules():
    if isinstance(mod, mq.QLinear):
        if "w2" in name: mod.weight_quantizer.qcfg.is_per_channel = True; mod.output_quantizer.qcfg.bitwidth = 16
        elif "o_proj" in name: mod.output_quantizer.qcfg.bitwidth = 16
    elif isinstance(mod, mq.QRMSNorm): mod.input_quantizer.qcfg.bitwidth = 16; mod.weight_quantizer.qcfg.bitwidth = 16
    elif isinstance(mod, mq.QMatMul):
        if "qk_bmm" in name: mod.output_quantizer.qcfg.bitwidth = 16
        if "pv_bmm" in name: mod.input_quantizer.qcfg.bitwidth = 16
mq.set_scale_and_offset(model, act, "buffer")
eng = DecodeEngine(model, cache_len=1024, attn_splits=int(os.environ.get('SPLITS', '1')), prefetch=float(os.environ.get('PREFETCH', '0.5')), prefetch_delay_us=float(os.environ.get('PFDELAY', '1.5')))
eng.fill_cache_random(256); eng.tok.fill_(17)
for _ in range(3): eng.step()
torch.cuda.synchronize()
t0 = time.perf_counter()
for _ in range(20): eng.step()
torch.cuda.synchronize()
print("eager ms/token", (time.perf_counter() - t0) / 20 * 1e3)
eng.set_position(256)
eng.capture()
for _ in range(3): eng.step()
torch.cuda.synchronize()
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
eng.set_position(256); e0.record()
for _ in range(64): eng.graph.replay()
e1.record(); e1.synchronize()
print("graph ms/token", e0.elapsed_time(e1) / 64, "layers", layers)

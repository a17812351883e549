"""Runs mq_attention_quant alone at TinyLlama's prefill shape (S = 2048, 32 heads / 4 KV heads) for rocprofv3:
    rocprofv3 --kernel-trace --stats -d /tmp/p -o p -- python tools/prof_attention.py
    rocprofv3 --pmc SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_WAIT_INST_ANY SQ_VALU_MFMA_BUSY_CYCLES --kernel-trace ...
"""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from mobilequant_amd import ops  # noqa: E402

S, H, KV, D = int(os.environ.get("MQ_ATT_S", 2048)), 32, 4, 64
dev = torch.device("cuda:0")
torch.manual_seed(0)
q, k, v = torch.randn(S, H * D, device=dev), torch.randn(S, KV * D, device=dev), torch.randn(S, KV * D, device=dev)
inv = 1.0 / (10000.0 ** (torch.arange(0, D, 2, dtype=torch.float32, device=dev) / D))
ang = torch.outer(torch.arange(S, dtype=torch.float32, device=dev), inv)
ang = torch.cat((ang, ang), -1)
cos, sin = ang.cos(), ang.sin()


def grid(lo, hi, bits):
    n = float(2 ** bits - 1)
    sc = (hi - lo) / n
    return (torch.tensor([sc], device=dev), torch.tensor([round(-lo / sc)], device=dev, dtype=torch.float32), 0.0, n)


grids = dict(qk_a=grid(-6.0, 6.0, 8), qk_b=grid(-6.0, 6.0, 8), qk_out=grid(-60.0, 60.0, 16), pv_a=grid(0.0, 1.0, 16), pv_b=grid(-4.5, 4.5, 8),
             pv_out=grid(-2.0, 2.0, 8))
if "MQ_ATT_F16" in os.environ:             # A/B: 0 = int8 score contraction, 1 = fp16 over the centred indices (default)
    import mobilequant_amd._lib as L
    L.load().mq_attention_set_f16(int(os.environ["MQ_ATT_F16"]))
if "MQ_ATT_FUSED_Q" in os.environ:         # A/B: 0 = the prep kernel writes the q image, 1 = the attention workgroups prepare their q rows
    import mobilequant_amd._lib as L
    L.load().mq_attention_set_fused_q(int(os.environ["MQ_ATT_FUSED_Q"]))
kw = {}
if os.environ.get("MQ_ATT_IDX"):           # index input, as in the fused layer (the q|k|v GEMM's uint8 output)
    idx = torch.randint(0, 256, (S, (H + 2 * KV) * D), dtype=torch.uint8, device=dev)
    kw["qkv_idx"] = (idx, tuple((torch.tensor([0.047], device=dev), torch.tensor([128.0], device=dev)) for _ in range(3)))
    q = k = v = None
img = torch.empty(S, H * D, dtype=torch.int8, device=dev)
rs = torch.empty(S, dtype=torch.int32, device=dev)
for _ in range(int(os.environ.get("MQ_ATT_ITERS", 20))):
    ops.attention_quant(q, k, v, cos, sin, H, KV, grids, image=(img, rs, 0, 128, False), want_out=False, **kw)
torch.cuda.synchronize()
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
e0.record()
for _ in range(20):
    ops.attention_quant(q, k, v, cos, sin, H, KV, grids, image=(img, rs, 0, 128, False), want_out=False, **kw)
e1.record()
torch.cuda.synchronize()
print("attention op (prep + core), eager: %.1f us" % (e0.elapsed_time(e1) / 20 * 1e3))

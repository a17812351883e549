"""A/B of the two score contractions of mq_attention_quant at head_dim 64 (int8 MFMA + zero-point terms against fp16 MFMA over the
centred indices, mq_attention_set_f16): identical outputs (fp32 out, int8 image, row sums) and the time of each, prep + core, in one
hipGraph.   python tools/att_f16_ab.py   [MQ_ATT_S=2048  MQ_ATT_KV=4  MQ_ATT_ROT=64]"""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from mobilequant_amd import ops  # noqa: E402
import mobilequant_amd._lib as L  # noqa: E402

S, H, KV, D = int(os.environ.get("MQ_ATT_S", 2048)), 32, int(os.environ.get("MQ_ATT_KV", 4)), 64
ROT = int(os.environ.get("MQ_ATT_ROT", 64))
dev = torch.device("cuda:0")
torch.manual_seed(0)
q, k, v = torch.randn(S, H * D, device=dev), torch.randn(S, KV * D, device=dev), torch.randn(S, KV * D, device=dev)
inv = 1.0 / (10000.0 ** (torch.arange(0, ROT, 2, dtype=torch.float32, device=dev) / ROT))
ang = torch.outer(torch.arange(S, dtype=torch.float32, device=dev), inv)
ang = torch.cat((ang, ang), -1)
cos, sin = ang.cos(), ang.sin()


def grid(lo, hi, bits):
    n = float(2 ** bits - 1)
    sc = (hi - lo) / n
    return (torch.tensor([sc], device=dev), torch.tensor([round(-lo / sc)], device=dev, dtype=torch.float32), 0.0, n)


grids = dict(qk_a=grid(-6.0, 6.0, 8), qk_b=grid(-5.0, 7.0, 8), qk_out=grid(-60.0, 60.0, 16), pv_a=grid(0.0, 1.0, 16), pv_b=grid(-4.5, 4.0, 8),
             pv_out=grid(-2.0, 2.0, 8))
lib = L.load()


def run(f16):
    lib.mq_attention_set_f16(f16)
    img = torch.zeros(S, H * D, dtype=torch.int8, device=dev)
    rs = torch.zeros(S, dtype=torch.int32, device=dev)
    out = ops.attention_quant(q, k, v, cos, sin, H, KV, grids, image=(img, rs, 0, 128, False), want_out=True)
    torch.cuda.synchronize()
    g = torch.cuda.CUDAGraph()
    st = torch.cuda.Stream()
    with torch.cuda.stream(st):
        for _ in range(3):
            ops.attention_quant(q, k, v, cos, sin, H, KV, grids, image=(img, rs, 0, 128, False), want_out=False)
        with torch.cuda.graph(g, stream=st):
            for _ in range(10):
                ops.attention_quant(q, k, v, cos, sin, H, KV, grids, image=(img, rs, 0, 128, False), want_out=False)
    ts = []
    for _ in range(7):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        g.replay()
        e1.record()
        torch.cuda.synchronize()
        ts.append(e0.elapsed_time(e1) / 10 * 1e3)
    return out, img, rs, sorted(ts)[len(ts) // 2]


if "MQ_ATT_PAIR" in os.environ:            # 1 = two heads of a KV group per eight-wave workgroup (the f16 form)
    lib.mq_attention_set_pair(int(os.environ["MQ_ATT_PAIR"]))
o0, i0, r0, t0 = run(0)
o1, i1, r1, t1 = run(1)
o0b, i0b, r0b, t0b = run(0)
o1b, i1b, r1b, t1b = run(1)
print(f"S={S} H={H} KV={KV} rot={ROT}: int8 scores {t0:.1f} / {t0b:.1f} us, f16 scores {t1:.1f} / {t1b:.1f} us (prep + core, hipGraph)")
print("identical: out", bool(torch.equal(o0, o1)), "image", bool(torch.equal(i0, i1)), "row sums", bool(torch.equal(r0, r1)),
      "| max |out diff|", float((o0 - o1).abs().max()))
lib.mq_attention_set_f16(1)

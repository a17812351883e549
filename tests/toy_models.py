"""Small module stacks with the leaf-module names the reference's surgery rules key on.  They mirror
the architectures oracle/gen_golden.py built with the reference's own leaf classes, so the frozen
state dicts in tests/golden load into them."""
import torch
import torch.nn as nn

from mobilequant_amd.quantization.fp_ops import FMatMul, HFRMSNorm


class Block(nn.Module):
    def __init__(self, d=32, f=48):
        super().__init__()
        self.input_layernorm = HFRMSNorm(d)
        self.q_proj, self.k_proj, self.v_proj, self.o_proj = (nn.Linear(d, d, bias=False) for _ in range(4))
        self.qk_bmm, self.pv_bmm = FMatMul(), FMatMul()
        self.post_attention_layernorm = HFRMSNorm(d)
        self.w1, self.w3, self.w2 = nn.Linear(d, f, bias=False), nn.Linear(d, f, bias=False), nn.Linear(f, d, bias=False)
        self.act_fn = nn.SiLU()

    def forward(self, x):
        h = self.input_layernorm(x)
        q, k, v = self.q_proj(h), self.k_proj(h), self.v_proj(h)
        s = self.qk_bmm(q, k.transpose(-1, -2)) / (q.shape[-1] ** 0.5)
        a = self.pv_bmm(torch.softmax(s, dim=-1), v)
        x = x + self.o_proj(a)
        h = self.post_attention_layernorm(x)
        return x + self.w2(self.act_fn(self.w1(h)) * self.w3(h))


class ToyLM(nn.Module):
    def __init__(self):
        super().__init__()
        self.layers = nn.ModuleList([Block(), Block()])
        self.norm = HFRMSNorm(32)
        self.lm_head = nn.Linear(32, 50, bias=False)

    def forward(self, x):
        for l in self.layers:
            x = l(x)
        return self.lm_head(self.norm(x))


class CalibToy(nn.Module):
    def __init__(self):
        super().__init__()
        self.emb = nn.Embedding(50, 32)
        self.fc1 = nn.Linear(32, 48)
        self.act = nn.SiLU()
        self.fc2 = nn.Linear(48, 32)
        self.bmm = FMatMul()
        self.ln = nn.LayerNorm(32)

    def forward(self, ids):
        h = self.emb(ids)
        c = self.fc2(self.act(self.fc1(h)))
        s = self.bmm(c, c.transpose(-1, -2))
        return self.ln(c) + s.mean()


def apply_mixed_precision(model, Q):
    """The rules of ptq/mobilequant.py:175-201 for a W8A8 run."""
    for name, mod in model.named_modules():
        if isinstance(mod, Q.QLinear):
            if "w2" in name:
                mod.weight_quantizer.qcfg.is_per_channel = True
                mod.output_quantizer.qcfg.bitwidth = 16
            elif "o_proj" in name:
                mod.output_quantizer.qcfg.bitwidth = 16
        elif isinstance(mod, Q.QRMSNorm):
            mod.input_quantizer.qcfg.bitwidth = 16
            mod.weight_quantizer.qcfg.bitwidth = 16
        elif isinstance(mod, Q.QMatMul):
            if "qk_bmm" in name:
                mod.output_quantizer.qcfg.bitwidth = 16
            if "pv_bmm" in name:
                mod.input_quantizer.qcfg.bitwidth = 16


from seeded import seeded_parameters_  # noqa: E402,F401  (standalone: oracle/gen_golden.py imports it without this package)

import torch, bench_variants
from mobilequant_amd.quantization import qmodule as Q
dev = torch.device("cuda:0")
with torch.no_grad():
    for fam in ("tinyllama", "gemma_2b"):
        for mode in ("image", "packed", "image", "packed"):
            Q.QLinear.w4_prefill = mode
            try:
                r = bench_variants.bench_layer_full(dev, modes=("fused",), wbits=4, family=fam)
                print(fam, mode, {k: v for k, v in r.items() if "fused_us" in k or "tops" in k}, flush=True)
            except Exception as e:
                print(fam, mode, "ERR", repr(e)[:300], flush=True)

"""A/B of DecodeEngine configurations on ONE box (boxes differ by a few percent): tok/s at context 256 (and others) per configuration.
usage: python tools/r06_decode_ab.py "launches=4,prefetch=0.5" "launches=4,prefetch=0" "launches=5" """
import os, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench  # noqa: E402
import mobilequant_amd.decode as D  # noqa: E402

dev = torch.device("cuda:0")
real = D.DecodeEngine
for spec in sys.argv[1:]:
    kw = {}
    for item in spec.split(","):
        k, v = item.split("=")
        kw[k] = float(v) if "." in v else int(v)
    launches = kw.pop("launches", 4)

    class Eng(real):
        def __init__(self, model, **k):
            k.update(kw)
            super().__init__(model, **k)
    bench_kw = dict(wbits=kw.pop("wbits", 8))
    import mobilequant_amd.decode as DD
    DD.DecodeEngine = Eng
    try:
        with torch.no_grad():
            r = bench.bench_decode_full(dev, cache_len=2176, also_contexts=(256, 512, 1024, 2048), launches=launches, **bench_kw)
    finally:
        DD.DecodeEngine = real
    print(spec, "->", r["decode_tok_s"], "tok/s", r["ms_per_token"], "ms", r["decode_tok_s_by_context"], flush=True)

import torch, sys, os
sys.path.insert(0, os.getcwd())
from mobilequant_amd import llama
from mobilequant_amd.calibration import ActRangeCollector, get_act_range
dev = torch.device("cuda:0")
for fam in ("tinyllama", "gemma_2b", "stablelm_2_1_6b"):
    shape = getattr(llama.LlamaShape, fam)(layers=2, max_pos=2048, vocab=4096)
    model = llama.LlamaForCausalLM(shape)
    model.reset_parameters(seed=1337, std=0.05)
    model = model.to(dev).eval().requires_grad_(False)
    g = torch.Generator().manual_seed(11)
    samples = [torch.randint(0, shape.vocab, (1, 2048), generator=g).to(dev) for _ in range(3)]
    a = get_act_range(model, samples)
    ActRangeCollector.fuse_layer_statistics = False; ActRangeCollector.mirror_declared_aliases = False; ActRangeCollector.keep_causal_zeros = False
    try:
        b = get_act_range(model, samples)
    finally:
        ActRangeCollector.fuse_layer_statistics = True; ActRangeCollector.mirror_declared_aliases = True; ActRangeCollector.keep_causal_zeros = True
    worst = (0.0, None)
    assert a.keys() == b.keys()
    for n in b:
        assert a[n].keys() == b[n].keys(), n
        for f, (lo, hi) in b[n].items():
            for u, v in ((a[n][f][0], lo), (a[n][f][1], hi)):
                r = abs(u - v) / max(abs(v), 1e-3)
                if r > worst[0]: worst = (r, (n, f, u, v))
    print(fam, "slots", sum(len(v) for v in b.values()), "worst relative deviation", worst, flush=True)
    del model
    torch.cuda.empty_cache()

#!/usr/bin/env python3
"""Is the W4A8 perplexity offset of the integer paths against full_depth_stable_case.npz a bias or a sample of summation-order noise?
Runs the 22-layer contractive model on: the simulated path (HIP fake-quant around the fp32 library GEMM / bmm: the reference's op
sequence with rocBLAS summing), the module chain on the integer kernels, the fused prefill; prints per path the mean per-position NLL
difference against the reference, its standard error over the 255 positions, and the perplexity difference."""
import os, sys
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "tests"))
import numpy as np
import torch
import mobilequant_amd as mq
from mobilequant_amd import llama
import test_gpu_round5 as T

dev = torch.device("cuda:0")
for tag in sys.argv[1:] or ["w8a8", "w4a8"]:
    m, z = T._stable_model(dev, tag)
    ids_all, ref, _ = T._stable_reference(z, tag)
    ids_t = torch.from_numpy(ids_all).long().to(dev)

    def report(name, run):
        nll = np.stack([T._nll(run(ids_t[i]), ids_t[i])[0] for i in range(ids_t.shape[0])])
        d = nll - ref
        per_seq = [round(float(np.exp(nll[i].mean()) - np.exp(ref[i].mean())), 4) for i in range(nll.shape[0])]
        print(f"[{tag}] {name:28s} dppl {float(np.exp(nll.mean()) - np.exp(ref.mean())):+.5f}  mean dNLL {d.mean():+.6f}  SE(iid) {d.std() / np.sqrt(d.size):.6f}  "
              f"per sequence {per_seq}", flush=True)
        return nll
    with torch.no_grad():
        for mod in m.modules():
            if isinstance(mod, (mq.QLinear, mq.QMatMul)):
                mod.int8_mode = "off"
        sim = report("simulated (rocBLAS fp32)", lambda t: m(t.view(1, -1))[0])
        for mod in m.modules():
            if isinstance(mod, mq.QLinear):
                mod.int8_mode = "auto"
        lin = report("int8 linears, fp32 bmm", lambda t: m(t.view(1, -1))[0])
        for mod in m.modules():
            if isinstance(mod, mq.QMatMul):
                mod.int8_mode = "auto"
        chain = report("module chain (all integer)", lambda t: m(t.view(1, -1))[0])
        llama.fuse_decoder_layer(m)
        fused = report("fused prefill", lambda t: m(t.view(1, -1))[0])
    d = chain - sim
    print(f"[{tag}] chain vs simulated: mean dNLL {d.mean():+.6f} SE {d.std() / np.sqrt(d.size):.6f}")
    print(f"[{tag}] reference self3: mean dNLL {(T._stable_reference(z, tag + '_self3')[1] - ref).mean():+.6f}")
    del m
    torch.cuda.empty_cache()

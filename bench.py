#!/usr/bin/env python3
"""Hot-path benchmark: the W8A8 real-int8 QLinear step at TinyLlama-1.1B's headline shape.

    python bench.py --gpus N --steps K --warmup W
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 --master-port P \
        bench.py --gpus N --steps K --warmup W

One STEP = one pass of the hot path over one batch of synthetic activations already resident in HBM:
    fp32 x [2048, 2048]  --mq_quantize-->  int8 + row sums  --mq_w8a8_linear (MFMA i8, fused dequant +
    8-bit output quantizer)-->  output indices [2048, 5632]                     (BASELINE.json configs[1])
metric = 2*M*K*N int8 ops per step / step time, summed over ranks (replicas: the quantized forward has no
exchange step -- DESIGN.md "Multi-GPU"; the data-parallel calibration path with its single all-reduce is
exercised by `--workload calibration`).

The JSON line also carries
  roofline     : the dominant kernel (the int8 GEMM) against the dense int8 MFMA peak, from HIP-event timing
                 of that kernel alone on the stream it is launched on;
  cpu_baseline : the numpy oracle of the reference's simulated QLinear (oracle/mq_oracle.py, "port") timed
                 on this box's host cores on the same shape -- a reported baseline, not a target;
  variants     : the same step with fp16 / fp32 outputs, and the drop-in nn.Module forward (fp32 in/out).
"""
from __future__ import annotations

import argparse
import json
import os
import sys
import time

import numpy as np
import torch

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

M, K, N = 2048, 2048, 5632            # bsz*seq, hidden, FFN  (TinyLlama-1.1B w1/w3)
OPS_PER_STEP = 2.0 * M * K * N
INT8_MFMA_PEAK_TOPS = 5000.0          # dense int8 = 2x the 2.5 PF bf16 dense peak (MI355X_MICROARCH.md)
N_BATCHES = 4                         # distinct activation batches rotated through the steps
GRAPH_STEPS = 10                      # steps captured per hipGraph (launch-bound otherwise)


def parse():
    p = argparse.ArgumentParser()
    p.add_argument("--gpus", type=int, default=1)
    p.add_argument("--steps", type=int, default=200)
    p.add_argument("--warmup", type=int, default=20)
    p.add_argument("--workload", default="qlinear", choices=["qlinear", "calibration"])
    p.add_argument("--no-cpu-baseline", action="store_true")
    p.add_argument("--no-graph", action="store_true")
    p.add_argument("--gemm-variant", type=int, default=-1, help="force a GEMM tile variant (experiments)")
    p.add_argument("--row-major-activations", action="store_true",
                   help="A/B: quantise row-major and run the C++ ping-pong GEMM loop instead of the fragment-blocked layout")
    p.add_argument("--overlap", action="store_true",
                   help="experiment: quantize(batch i+1) on a side stream next to GEMM(batch i); measured SLOWER "
                        "(36.3 vs 34.7 us/step): the GEMM's one workgroup per CU leaves no room to co-schedule")
    return p.parse_args()


def dist_setup(args):
    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local = int(os.environ.get("LOCAL_RANK", "0"))
    torch.cuda.set_device(local)
    if world > 1:
        import torch.distributed as dist
        dist.init_process_group("nccl", device_id=torch.device("cuda", local))
    return rank, world, local


def barrier(world):
    if world > 1:
        import torch.distributed as dist
        dist.barrier()
    torch.cuda.synchronize()


def max_over_ranks(v: float, world) -> float:
    if world == 1:
        return v
    import torch.distributed as dist
    t = torch.tensor([v], dtype=torch.float64, device="cuda")
    dist.all_reduce(t, op=dist.ReduceOp.MAX)
    return float(t.item())


class Step:
    """The real-int8 QLinear step on preallocated buffers (no allocation inside the timed region)."""

    tiled_ok = True       # --row-major-activations turns the fragment-blocked layout off (A/B)

    def __init__(self, dev, out_dtype_code, seed):
        from mobilequant_amd import ops
        from mobilequant_amd._lib import MQ_I8
        import mobilequant_amd as mq
        self.ops = ops
        g = torch.Generator(device="cpu").manual_seed(1337 + seed)      # the reference's seed (mobilequant.py:87)
        self.x = [torch.randn(M, K, generator=g).to(dev) for _ in range(N_BATCHES)]
        w = (torch.randn(N, K, generator=g) * 0.02).to(dev)
        lo = min(float(x.min()) for x in self.x)
        hi = max(float(x.max()) for x in self.x)
        self.aq = mq.Quantizer(mq.QuantConfig(bitwidth=8))              # per-tensor asymmetric, static
        self.aq.set_scale_offset_from_minmax(lo, hi, "buffer", dev)
        wq = mq.Quantizer(mq.QuantConfig(bitwidth=8))
        wq(w)                                                           # first forward: range from the weight
        self.w8, colsum, wshift = wq.quantize_to_int(w, MQ_I8, want_row_sum=True, rows=N)
        self.alpha, self.wzp, self.ct = ops.linear_epilogue_prepare(self.aq.scale, self.aq.offset, 128, wq.scale.detach(),
                                                                    wq.offset.detach(), wshift, colsum, K)
        y = torch.nn.functional.linear(self.x[0], w)
        self.oq = mq.Quantizer(mq.QuantConfig(bitwidth=8))
        self.oq.set_scale_offset_from_minmax(float(y.min()), float(y.max()), "buffer", dev)
        del y
        from mobilequant_amd.ops import _OUT_TORCH
        self.code = out_dtype_code
        # two int8 activation buffers: the quantize of batch i+1 may run while the GEMM of batch i reads its own
        self.a8s = [torch.empty(M, K, dtype=torch.int8, device=dev) for _ in range(2)]
        self.rss = [torch.empty(M, dtype=torch.int32, device=dev) for _ in range(2)]
        self.a8, self.rs = self.a8s[0], self.rss[0]
        self.out = torch.empty(M, N, dtype=_OUT_TORCH[out_dtype_code], device=dev)
        self.w_fp = w
        # the layout QLinear._forward_int8 picks for this shape: fragment-blocked activations + generated-ISA GEMM loop
        self.tiled = Step.tiled_ok and ops.gemm_tiled_supported(M, N, K)

    def quantize(self, i, slot=0):
        from mobilequant_amd import _lib
        from mobilequant_amd._lib import MQ_F32, MQ_I8
        x = self.x[i % N_BATCHES]
        if self.tiled:
            _lib.call("mq_quantize_tiled", x.data_ptr(), MQ_F32, M, K, self.aq.scale.data_ptr(), self.aq.offset.data_ptr(),
                      0.0, 255.0, 128, self.a8s[slot].data_ptr(), self.rss[slot].data_ptr(), torch.cuda.current_stream().cuda_stream)
            return
        _lib.call("mq_quantize", x.data_ptr(), MQ_F32, M, K, self.aq.scale.data_ptr(), self.aq.offset.data_ptr(), 1,
                  0.0, 255.0, 128, self.a8s[slot].data_ptr(), MQ_I8, self.rss[slot].data_ptr(),
                  torch.cuda.current_stream().cuda_stream)

    def gemm(self, slot=0):
        self.ops.int8_linear(self.a8s[slot], self.w8, self.rss[slot], self.alpha, self.wzp, self.ct, None,
                             out_scale=self.oq.scale, out_offset=self.oq.offset, out_qmin=0.0, out_qmax=255.0,
                             out_dtype=self.code, out=self.out, a_tiled_rows=M if self.tiled else None)

    def __call__(self, i):
        self.quantize(i)
        self.gemm()

    def pipelined(self, n, side):
        """n steps with the activation quantize of batch i+1 on a side stream next to the GEMM of batch i (the
        GEMM leaves 24 KiB of LDS and ~150 VGPRs per SIMD lane free on every CU; the quantize kernel is HBM-bound).
        Every batch still passes through both kernels; only their order across batches is software-pipelined."""
        main = torch.cuda.current_stream()
        q_done = [torch.cuda.Event() for _ in range(n)]
        g_done = [torch.cuda.Event() for _ in range(n)]
        side.wait_stream(main)
        for i in range(n):
            with torch.cuda.stream(side):
                if i >= 2:
                    side.wait_event(g_done[i - 2])          # slot i%2 was read by GEMM i-2
                self.quantize(i, i % 2)
                q_done[i].record(side)
            main.wait_event(q_done[i])
            self.gemm(i % 2)
            g_done[i].record(main)
        main.wait_stream(side)


def run_steps(fn, steps, warmup, world, use_graph=True, pipelined=False):
    """W untimed warmup steps, then EXACTLY `steps` timed steps bracketed by barrier + synchronize.
    Steps are replayed from hipGraphs of GRAPH_STEPS steps each (the step is ~30 us: launch-bound from
    Python otherwise); a remainder runs eagerly inside the same timed region."""
    for i in range(warmup):
        fn(i)
    torch.cuda.synchronize()
    graph = None
    if use_graph and steps >= GRAPH_STEPS:
        s = torch.cuda.Stream()
        s.wait_stream(torch.cuda.current_stream())
        with torch.cuda.stream(s):
            for i in range(3):
                fn(i)
        torch.cuda.current_stream().wait_stream(s)
        graph = torch.cuda.CUDAGraph()
        side = torch.cuda.Stream()
        with torch.cuda.graph(graph):
            if pipelined:
                fn.pipelined(GRAPH_STEPS, side)
            else:
                for i in range(GRAPH_STEPS):
                    fn(i)
        graph.replay()
        torch.cuda.synchronize()
    barrier(world)
    t0 = time.perf_counter()
    done = 0
    if graph is not None:
        for _ in range(steps // GRAPH_STEPS):
            graph.replay()
        done = (steps // GRAPH_STEPS) * GRAPH_STEPS
    for i in range(done, steps):
        fn(i)
    barrier(world)
    return (time.perf_counter() - t0) / steps


def event_time(fn, iters):
    """Average duration of one `fn` launch: `iters` launches captured in a hipGraph (so the host cannot be
    the bottleneck) and bracketed by HIP events on the stream the kernels run on."""
    fn()
    torch.cuda.synchronize()
    s = torch.cuda.Stream()
    s.wait_stream(torch.cuda.current_stream())
    with torch.cuda.stream(s):
        fn()
    torch.cuda.current_stream().wait_stream(s)
    g = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g):
        for _ in range(iters):
            fn()
    g.replay()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    best = float("inf")
    for _ in range(5):
        e0.record()
        g.replay()
        e1.record()
        e1.synchronize()
        best = min(best, e0.elapsed_time(e1) / iters)
    return best * 1e-3


def pmc_traffic():
    """HBM bytes per GEMM launch from the committed rocprofv3 PMC passes of this same command
    (profiles/r01/pmc_fetch + pmc_write; FETCH_SIZE is doubled as MI355X_MICROARCH.md prescribes for 16 B/lane
    streams on gfx950).  Counters cannot be read from inside the bench, so this is the last profiled value."""
    import re
    d = os.path.join(ROOT, "profiles", "r01")
    vals = {}
    for fn, key in (("pmc_fetch.summary.txt", "FETCH_SIZE"), ("pmc_write.summary.txt", "WRITE_SIZE")):
        try:
            txt = open(os.path.join(d, fn)).read()
        except OSError:
            return None, None
        m = re.search(r"gemm_i8_kernel<256, 176, 8, 1, 3,[^\n]*\n(?:\s+\S+\s+[\d.]+[^\n]*\n)*?\s+" + key + r"\s+([\d.]+)", txt)
        if not m:
            return None, None
        vals[key] = float(m.group(1)) * 1024.0
    return 2.0 * vals["FETCH_SIZE"] + vals["WRITE_SIZE"], "profiles/r01/pmc_fetch.summary.txt + pmc_write.summary.txt"


def cpu_baseline():
    """The oracle's restatement of the reference's simulated QLinear.forward (weight re-quantised on
    every call, as qmodule.py:346-347 does), on this box's host cores, bounded to ~10-30 s."""
    from oracle import mq_oracle as O
    rng = np.random.default_rng(1337)
    x = rng.standard_normal((M, K), dtype=np.float32)
    w = (rng.standard_normal((N, K), dtype=np.float32) * np.float32(0.02)).astype(np.float32)
    wq, iq, oq = O.QuantizerOracle(8), O.QuantizerOracle(8), O.QuantizerOracle(8)
    iq.set_from_minmax(float(x.min()), float(x.max()))
    y = x @ w.T
    oq.set_from_minmax(float(y.min()), float(y.max()))
    try:
        from threadpoolctl import threadpool_info
        blas_threads = max([p.get("num_threads", 1) for p in threadpool_info()] or [1])
    except Exception:
        blas_threads = os.cpu_count() or 1
    O.qlinear_sim(x, w, None, wq, iq, oq)              # warm-up; also caches the weight grid like the reference
    t0 = time.perf_counter()
    reps = 0
    while reps < 3 or (time.perf_counter() - t0 < 10.0 and reps < 40):
        O.qlinear_sim(x, w, None, wq, iq, oq)
        reps += 1
    dt = (time.perf_counter() - t0) / reps
    return {"value": round(OPS_PER_STEP / dt / 1e12, 4), "unit": "TOPS", "cores": int(blas_threads), "kind": "port",
            "seconds_per_step": round(dt, 4),
            "sample": f"{reps} calls of oracle.qlinear_sim (weight fake-quant + input fake-quant + fp32 GEMM + output "
                      f"fake-quant) at M={M},K={K},N={N}; numpy elementwise on 1 thread, OpenBLAS GEMM on {blas_threads}"}


def bench_decode(dev, w4=False):
    """TinyLlama-1.1B decode, linears only: per layer quantize(x) -> GEMV qkv (2048->2560) -> quantize -> GEMV o
    (2048->2048) -> quantize -> GEMV w1|w3 (2048->11264) -> quantize -> GEMV w2 (5632->2048), 22 layers with their own
    int8 weights (0.97 GB streamed per token), one hipGraph per token.  Attention, norms and sampling are outside the
    hot path of this repository and are not included."""
    import mobilequant_amd as mq
    from mobilequant_amd import ops
    from mobilequant_amd._lib import MQ_F32, MQ_I8
    g = torch.Generator(device="cpu").manual_seed(7)
    shapes = [(2048, 2560), (2048, 2048), (2048, 11264), (5632, 2048)]      # (K, N) per layer
    layers = []
    aq = mq.Quantizer(mq.QuantConfig(bitwidth=8)); aq.set_scale_offset_from_minmax(-4.0, 4.0, "buffer", dev)
    oq = mq.Quantizer(mq.QuantConfig(bitwidth=8)); oq.set_scale_offset_from_minmax(-4.0, 4.0, "buffer", dev)
    for _ in range(22):
        lw = []
        for Kk, Nn in shapes:
            if w4:      # packed unsigned nibbles, zero point 8
                nib = torch.randint(0, 16, (Nn, Kk), dtype=torch.uint8, generator=g).to(dev)
                colsum = nib.to(torch.int32).sum(1).to(torch.int32)
                w8 = ops.pack_w4(nib)
                wscale = torch.full((1,), 3e-3, device=dev); woff = torch.full((1,), 8.0, device=dev)
                alpha, wzp, ct = ops.linear_epilogue_prepare(aq.scale, aq.offset, 128, wscale, woff, 0, colsum, Kk)
            else:
                w8 = torch.randint(-128, 128, (Nn, Kk), dtype=torch.int8, generator=g).to(dev)
                colsum = w8.to(torch.int32).sum(1).to(torch.int32)
                wscale = torch.full((1,), 2e-4, device=dev); woff = torch.full((1,), 128.0, device=dev)
                alpha, wzp, ct = ops.linear_epilogue_prepare(aq.scale, aq.offset, 128, wscale, woff, 128, colsum, Kk)
            lw.append((w8, alpha, wzp, ct, torch.empty(1, Nn, device=dev)))
        layers.append(lw)
    xs = {2048: torch.randn(1, 2048, device=dev), 5632: torch.randn(1, 5632, device=dev)}
    a8 = {k: torch.empty(1, k, dtype=torch.int8, device=dev) for k in xs}
    rs = {k: torch.empty(1, dtype=torch.int32, device=dev) for k in xs}
    from mobilequant_amd import _lib
    st = lambda: torch.cuda.current_stream().cuda_stream

    def token():
        for lw in layers:
            for (Kk, Nn), (w8, alpha, wzp, ct, out) in zip(shapes, lw):
                ops.int8_linear_f32in(xs[Kk], aq.scale, aq.offset, 0.0, 255.0, 128, w8, alpha, wzp, ct, None, out_scale=oq.scale,
                                      out_offset=oq.offset, out_qmin=0.0, out_qmax=255.0, out_dtype=MQ_F32, out=out, w4=w4)
    t = event_time(token, 1)      # graph of one token, best of 5 replays
    wbytes = 22 * sum(Kk * Nn for Kk, Nn in shapes) // (2 if w4 else 1)
    return {"decode_tok_s": round(1.0 / t, 1), "ms_per_token": round(t * 1e3, 4), "weight_GB_per_token": round(wbytes / 1e9, 4),
            "achieved_GBps": round(wbytes / t / 1e9, 1), "peak_GBps": 8000.0, "kernels_per_token": 22 * 4,
            "scope": "linears only (22 layers x [qkv, o, w1|w3, w2] %s GEMV with the activation quantize fused in), batch 1, hipGraph" % ("W4A8" if w4 else "W8A8")}


def bench_layer(dev):
    """The quantized-linear path of ONE TinyLlama decoder layer at prefill (S = 2048) through the module API:
    attention_norm -> q/k/v, o_proj, ffn_norm -> w1/w3, w2 (W8A8, 8-bit activations, 16-bit norm inputs; attention,
    RoPE, SiLU*mul and residuals are not part of the hot path and are left out: every linear gets a ready input).
    Reports the hipGraph time of the 2 norms + 7 linears with the fused kernels / integer chaining, and with
    fused_mode = "off" + no chaining (composite norms, every linear quantising its own input)."""
    import mobilequant_amd as mq
    from mobilequant_amd.quantization import qmodule as Q
    from mobilequant_amd.quantization.fp_ops import HFRMSNorm
    a8, a16 = mq.QuantConfig(bitwidth=8), mq.QuantConfig(bitwidth=16)
    S, H, F_, KV = 2048, 2048, 5632, 256
    torch.manual_seed(1337)

    def lin(k, n, own_input_quantizer):
        ql = mq.QLinear.from_float(torch.nn.Linear(k, n, bias=False).to(dev), a8, a8, a8).requires_grad_(False)
        if not own_input_quantizer:
            ql.input_quantizer = None
        ql.set_scale_offset({"input": [-4.0, 4.0], "output": [-3.0, 3.0]}, "buffer")
        return ql

    def norm():
        n = mq.QRMSNorm.from_float(HFRMSNorm(H, eps=1e-5).to(dev), a16, a16, a8).requires_grad_(False)
        n.set_scale_offset({"input": [-5.0, 5.0], "output": [-4.0, 4.0]}, "buffer")
        return n

    class Layer(torch.nn.Module):
        def __init__(self):
            super().__init__()
            self.attention_norm, self.ffn_norm = norm(), norm()
            self.q_proj, self.k_proj, self.v_proj = lin(H, H, False), lin(H, KV, False), lin(H, KV, False)
            self.o_proj, self.w1, self.w3, self.w2 = lin(H, H, True), lin(H, F_, False), lin(H, F_, False), lin(F_, H, True)

        def forward(self, x, attn_out, ffn_mid):
            h = self.attention_norm(x)
            q, k, v = self.q_proj(h), self.k_proj(h), self.v_proj(h)
            o = self.o_proj(attn_out)
            g = self.ffn_norm(x)
            a, b = self.w1(g), self.w3(g)
            d = self.w2(ffn_mid)
            return q, k, v, o, a, b, d
    layer = Layer()
    mq.wire_integer_inputs(layer)
    x, attn_out, ffn_mid = torch.randn(1, S, H, device=dev), torch.randn(1, S, H, device=dev), torch.randn(1, S, F_, device=dev)
    res = {}
    for mode in ("fused", "composite"):
        for m in layer.modules():
            if hasattr(m, "fused_mode"):
                m.fused_mode = "auto" if mode == "fused" else "off"

        def fwd():
            if mode == "composite":
                Q._shared_activation.clear()
            layer(x, attn_out, ffn_mid)
        fwd()
        res[mode + "_us"] = round(event_time(fwd, 5) * 1e6, 1)
    ops_layer = 2.0 * S * (H * H * 2 + H * KV * 2 + H * F_ * 3)
    res["tops_fused"] = round(ops_layer / (res["fused_us"] * 1e-6) / 1e12, 1)
    res["scope"] = "2 QRMSNorm + 7 QLinear of one TinyLlama layer, S = 2048, W8A8, module API, hipGraph"
    return res


def bench_calibration(args, rank, world, dev):
    """Data-parallel activation-range calibration over a TinyLlama-shaped MLP block stack (synthetic)."""
    import torch.nn as nn
    from mobilequant_amd.calibration import ActRangeCollector
    from mobilequant_amd.quantization.fp_ops import HFRMSNorm

    class MLP(nn.Module):
        def __init__(self):
            super().__init__()
            self.norm1 = HFRMSNorm(2048)
            self.w1, self.w3, self.w2 = nn.Linear(2048, 5632, bias=False), nn.Linear(2048, 5632, bias=False), nn.Linear(5632, 2048, bias=False)
            self.act_fn = nn.SiLU()

        def forward(self, x):
            h = self.norm1(x)
            return x + self.w2(self.act_fn(self.w1(h)) * self.w3(h))

    torch.manual_seed(1337)
    model = nn.Sequential(*[MLP() for _ in range(2)]).to(dev).eval()
    n_samples = 64
    xs = [torch.randn(1, 2048, 2048, device=dev) for _ in range(4)]
    col = ActRangeCollector(model).attach()
    with torch.no_grad():
        for i in range(2):
            model(xs[i % 4])
        barrier(world)
        t0 = time.perf_counter()
        for i in range(rank, n_samples, world):
            model(xs[i % 4])
        col.all_reduce()
        barrier(world)
    dt = time.perf_counter() - t0
    col.detach()
    dt = max_over_ranks(dt, world)
    if rank == 0:
        print(json.dumps({"metric": "activation-range calibration samples/s (2 TinyLlama MLP blocks, S=2048, per-tensor)",
                          "value": round(n_samples / dt, 2), "unit": "samples/s", "n_gpus": world, "scaling": "strong",
                          "collectives": 1, "data": "synthetic"}))


def main():
    args = parse()
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs an MI355X (torch.cuda.is_available() is False); there is no CPU path")
    rank, world, local = dist_setup(args)
    dev = torch.device("cuda", local)
    import mobilequant_amd._lib as L
    from mobilequant_amd._lib import MQ_F16, MQ_F32, MQ_U8
    info = L.device_info()
    if args.workload == "calibration":
        return bench_calibration(args, rank, world, dev)

    with torch.no_grad():
        Step.tiled_ok = not args.row_major_activations
        step = Step(dev, MQ_U8, seed=rank)
        if args.gemm_variant >= 0:
            from mobilequant_amd import _lib as _l
            _l.load().mq_gemm_set_variant(args.gemm_variant)
        pipelined = args.overlap and not args.no_graph
        sec = run_steps(step, args.steps, args.warmup, world, use_graph=not args.no_graph, pipelined=pipelined)
        sec = max_over_ranks(sec, world)
        value = world * OPS_PER_STEP / sec / 1e12

        extras = {}
        roof = cpu = decode = None
        if rank == 0:
            # dominant kernel alone, HIP events on its stream
            t_gemm = event_time(step.gemm, 50)
            t_quant = event_time(lambda: step.quantize(0), 50)
            achieved = OPS_PER_STEP / t_gemm / 1e12
            traffic, traffic_src = pmc_traffic()
            roof = {"bound": "mfma", "kernel": "mq::gemm_i8_kernel (%s)" % ("mq_w8a8_linear_tiled: generated-ISA loop, fragment-blocked activations"
                                                                        if step.tiled else "mq_w8a8_linear"), "achieved": round(achieved, 1),
                    "peak": INT8_MFMA_PEAK_TOPS, "unit": "TOPS", "frac": round(achieved / INT8_MFMA_PEAK_TOPS, 4),
                    "avg_launch_us": round(t_gemm * 1e6, 2), "traffic": traffic, "traffic_unit": "bytes/launch",
                    "traffic_source": traffic_src, "algorithmic_bytes_per_launch": M * K + N * K + M * N,
                    "algorithmic_ops_per_launch": OPS_PER_STEP,
                    "quantize_kernel": {"bound": "hbm", "avg_launch_us": round(t_quant * 1e6, 2),
                                        "achieved_GBps": round((M * K * 5 + M * 4) / t_quant / 1e9, 1), "peak_GBps": 8000.0}}
            if pipelined:
                t_ser = run_steps(step, min(args.steps, 100), 5, 1, use_graph=True, pipelined=False)
                extras["serial_u8"] = {"ms_per_step": round(t_ser * 1e3, 5), "value": round(OPS_PER_STEP / t_ser / 1e12, 1),
                                       "note": "quantize and GEMM back to back on one stream"}
            for name, code in (("out_f16", MQ_F16), ("out_f32", MQ_F32)):
                s2 = Step(dev, code, seed=rank)
                t2 = run_steps(s2, min(args.steps, 100), 5, 1, use_graph=not args.no_graph)
                extras[name] = {"ms_per_step": round(t2 * 1e3, 5), "value": round(OPS_PER_STEP / t2 / 1e12, 1)}
                del s2
            # drop-in nn.Module forward: fp32 in -> fp32 out, weight cached as int8 after the first call
            import mobilequant_amd as mq
            lin = torch.nn.Linear(K, N, bias=False, device=dev)
            with torch.no_grad():
                lin.weight.copy_(step.w_fp)
            a8 = mq.QuantConfig(bitwidth=8)
            ql = mq.QLinear.from_float(lin, a8, a8, a8).requires_grad_(False)
            ql.input_quantizer.set_scale_offset_from_minmax(float(step.x[0].min()), float(step.x[0].max()), "buffer", dev)
            ql.output_quantizer.set_scale_offset_from_minmax(-3.0, 3.0, "buffer", dev)
            x3 = step.x[0].view(1, M, K)
            ql(x3)
            tm = event_time(lambda: ql(x3), 30)
            extras["module_forward_f32"] = {"ms_per_step": round(tm * 1e3, 5), "value": round(OPS_PER_STEP / tm / 1e12, 1),
                                            "note": "QLinear.forward from Python, eager (includes host launch overhead)"}
            decode = bench_decode(dev)
            torch.cuda.empty_cache()
            extras["decode_w4a8"] = bench_decode(dev, w4=True)      # the reference's deployment mode: 4-bit weights
            torch.cuda.empty_cache()
            extras["layer_prefill"] = bench_layer(dev)
            if not args.no_cpu_baseline:
                cpu = cpu_baseline()

    if rank == 0:
        line = {
            "metric": "W8A8 QuantLinear GEMM TOPS (% int8 MFMA peak) + TinyLlama-1.1B decode tok/s",
            "value": round(value, 1), "unit": "TOPS", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": round(sec * 1e3, 5), "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
            "dtype": "i8", "data": "synthetic",
            "config": {"workload": "TinyLlama-1.1B W8A8 real-int8 QLinear step (BASELINE.json configs[1]): fp32 x[2048,2048] "
                                   "-> int8 quantize(+row sums) -> MFMA i8 GEMM 2048->5632 with fused dequant + 8-bit output "
                                   "quantizer -> u8 indices; per-tensor asymmetric activation ranges, per-tensor asymmetric weights",
                       "M": M, "K": K, "N": N, "parallelism": f"replicas x{world}", "graph_steps": 0 if args.no_graph else GRAPH_STEPS,
                       "schedule": "quantize(batch i+1) overlapped with GEMM(batch i) on a second stream" if pipelined else "serial",
                       "pct_int8_mfma_peak": round(100 * value / world / INT8_MFMA_PEAK_TOPS, 2),
                       "decode_tok_s": decode["decode_tok_s"] if decode else None, "device": info},
            "roofline": roof, "cpu_baseline": cpu, "decode": decode, "variants": extras,
        }
        print(json.dumps(line))
    if world > 1:
        import torch.distributed as dist
        dist.barrier()            # ranks > 0 wait for rank 0's untimed extras before tearing RCCL down
        dist.destroy_process_group()


if __name__ == "__main__":
    main()

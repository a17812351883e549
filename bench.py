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
  cpu_baseline : the torch-CPU restatement of the reference's simulated QLinear (oracle/mq_oracle_torch.py, "port": the
                 reference's op sequence on torch CPU kernels, bit-checked against the numpy oracle) timed on this box's
                 physical host cores on the same shape -- a reported baseline, not a target;
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
    p.add_argument("--calib-samples", type=int, default=512, help="--workload calibration: samples (generate_act_range.py:30)")
    p.add_argument("--calib-layers", type=int, default=22, help="--workload calibration: decoder layers (TinyLlama-1.1B: 22)")
    p.add_argument("--calib-seq", type=int, default=2048, help="--workload calibration: tokens per sample")
    p.add_argument("--per-channel", action="store_true", help="--workload calibration: per-channel statistics")
    p.add_argument("--calib-stub-gemm", action="store_true",
                   help="--workload calibration: every Linear / FMatMul returns a resident tensor of its output shape instead of running the "
                        "fp32 library GEMM, so the timed region is the hot path (hooked reductions + the collective), not rocBLAS")
    p.add_argument("--no-cpu-baseline", action="store_true")
    p.add_argument("--no-graph", action="store_true")
    p.add_argument("--headline-only", action="store_true", help="only the timed steps and the roofline leg (profiler passes)")
    p.add_argument("--variants", action="store_true", help="also run the non-headline legs (bench_variants.py: layers at prefill, training step, "
                                                           "model prefill, ...): minutes, not seconds")
    p.add_argument("--row-major-activations", action="store_true",
                   help="A/B: quantise row-major and run the C++ ping-pong GEMM loop instead of the fragment-blocked layout")
    p.add_argument("--overlap", action="store_true",
                   help="experiment: quantize(batch i+1) on a side stream next to GEMM(batch i); measured SLOWER "
                        "(36.3 vs 34.7 us/step): the GEMM's one workgroup per CU leaves no room to co-schedule")
    return p.parse_args()


def maybe_spawn(args):
    """`python bench.py --gpus N` without an outer torchrun: re-launch under torch.distributed.run with N ranks (one process
    per GPU, RCCL over xGMI) and exit with its status.  Under an outer torchrun WORLD_SIZE must equal --gpus.  Asking for more
    GPUs than the box has fails loudly -- never a silent 1-rank run."""
    env_world = os.environ.get("WORLD_SIZE")
    if env_world is not None:
        if int(env_world) != args.gpus:
            raise SystemExit(f"bench.py: --gpus {args.gpus} but the launcher set WORLD_SIZE={env_world}")
        return
    if args.gpus == 1:
        return
    have = torch.cuda.device_count()
    if args.gpus > have:
        raise SystemExit(f"bench.py: --gpus {args.gpus} requested but only {have} GPU(s) are visible")
    import socket
    import subprocess
    with socket.socket() as sk:
        sk.bind(("127.0.0.1", 0))
        port = sk.getsockname()[1]
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={args.gpus}", "--master-addr", "127.0.0.1",
           "--master-port", str(port), os.path.abspath(__file__)] + sys.argv[1:]
    raise SystemExit(subprocess.call(cmd))


def dist_setup(args):
    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local = int(os.environ.get("LOCAL_RANK", "0"))
    if local >= torch.cuda.device_count():
        raise SystemExit(f"bench.py: LOCAL_RANK {local} has no GPU ({torch.cuda.device_count()} visible)")
    torch.cuda.set_device(local)
    if world > 1:
        import torch.distributed as dist
        dist.init_process_group("nccl", device_id=torch.device("cuda", local))
    return rank, world, local


def multi_gpu_report(rank, world, local, extras):
    """What each rank saw of the process group: communicator size, backend, device, host -- gathered to rank 0 so the
    parsed line itself shows that N ranks ran under ONE RCCL communicator of size N -- and how many collectives the timed data paths
    issue (GEMM / decode replicas: none; calibration: exactly one all-reduce per run, counted by the collector itself)."""
    import socket
    me = {"rank": rank, "local_rank": local, "device": torch.cuda.get_device_name(local), "host": socket.gethostname(), "communicator_size": 1,
          "backend": None}
    ranks = [me]
    if world > 1:
        import torch.distributed as dist
        me["communicator_size"], me["backend"] = dist.get_world_size(), dist.get_backend()
        ranks = [None] * world
        dist.all_gather_object(ranks, me)
    cal = {k: v.get("collectives") for k, v in extras.items() if k.startswith("calibration")}
    return {"world_size_env": world, "ranks": ranks, "all_ranks_agree": all(r["communicator_size"] == world for r in ranks),
            "gemm_step_collectives": 0, "decode_collectives": 0, "calibration_collectives_per_run": cal,
            "timing_collectives": "barrier before and after every timed repeat + one all_reduce(MAX) of the elapsed time (outside the timed region)"}


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
                      0.0, 255.0, 128, None, self.a8s[slot].data_ptr(), self.rss[slot].data_ptr(), torch.cuda.current_stream().cuda_stream)
            return
        _lib.call("mq_quantize", x.data_ptr(), MQ_F32, M, K, self.aq.scale.data_ptr(), self.aq.offset.data_ptr(), 1,
                  0.0, 255.0, 128, None, self.a8s[slot].data_ptr(), MQ_I8, self.rss[slot].data_ptr(),
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


REPEATS = 7


MAX_GRAPH_STEPS = 400      # steps captured in one hipGraph; longer runs replay several graphs back to back
PREROLL_STEPS = 100        # untimed steps replayed in front of the first event of every repeat (~3 ms: behind a barrier + synchronize the
                           # chip needs about a millisecond to settle at its sustained clock; a 20-step region is 0.6 ms)


def run_steps(fn, steps, warmup, world, use_graph=True, pipelined=False):
    """W untimed warmup steps, then EXACTLY `steps` timed steps.

    Round 4: the timed region is delimited by two HIP events on the stream the steps run on, around hipGraph(s) that
    contain exactly `steps` steps -- no host clock, no Python between the first and the last timed kernel.  An untimed pre-roll graph
    (PREROLL_STEPS steps) runs in front of the first event: the host has then already enqueued the event and the timed graph(s) when the
    GPU reaches them (otherwise one graph-launch latency, 10-16 us, sits inside a 0.6-ms region), and the chip is at its sustained clock
    (measured: without it a 20-step region reads 9 % slower than a 200-step one).  Every repeat is still bracketed by barrier +
    torch.cuda.synchronize() on both sides and the maximum over ranks is taken by the caller; `--steps 20` and `--steps 200` now agree.
    The host-clock figure of the same bracket (pre-roll excluded by subtraction is not possible, so it is the old method on its own
    repeat) is kept in run_steps.host_wall for reference."""
    for i in range(warmup):
        fn(i)
    torch.cuda.synchronize()
    graphs = []
    preroll = None
    if use_graph:
        s = torch.cuda.Stream()
        s.wait_stream(torch.cuda.current_stream())
        with torch.cuda.stream(s):
            for i in range(3):
                fn(i)
        torch.cuda.current_stream().wait_stream(s)
        side = torch.cuda.Stream()

        def capture(n, first):
            g = torch.cuda.CUDAGraph()
            with torch.cuda.graph(g):
                if pipelined:
                    fn.pipelined(n, side)
                else:
                    for i in range(first, first + n):
                        fn(i)
            return g
        done = 0
        while done < steps:
            n = min(MAX_GRAPH_STEPS, steps - done)
            graphs.append(capture(n, done))
            done += n
        preroll = capture(PREROLL_STEPS, 0)
        for g in graphs + [preroll]:
            g.replay()
        torch.cuda.synchronize()
    times, host = [], []
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    # BOTH methods, REPEATS times each, alternating: even repeats = the host-clock bracket of rounds 1-3 around the same K
    # steps (barrier | perf_counter | graphs | barrier | perf_counter, no pre-roll: it carries one graph-launch + synchronize latency);
    # odd repeats = HIP events around the K steps behind the untimed pre-roll (round 4's method, the headline).  Medians of each.
    for rep in range(2 * REPEATS):
        by_events = rep % 2 == 1
        barrier(world)
        t0 = time.perf_counter()
        if preroll is not None and by_events:
            preroll.replay()                     # untimed: the GPU is busy while the host enqueues e0 and the timed graphs
        e0.record()
        if graphs:
            for g in graphs:
                g.replay()
        else:
            for i in range(steps):
                fn(i)
        e1.record()
        barrier(world)
        if by_events:
            times.append(e0.elapsed_time(e1) * 1e-3 / steps)
        else:
            host.append((time.perf_counter() - t0) / steps)
    run_steps.last = sorted(times)
    run_steps.host_last = sorted(host)
    run_steps.host_wall = run_steps.host_last[len(host) // 2]
    return run_steps.last[len(times) // 2]


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
    # pre-roll, as run_steps does: the first replays of a fresh graph read 1-2 us per launch slower than the following ones (the same
    # library measured four times in a row: 21.2, 20.9, 20.1, 20.2 us); the timed replays start on a chip that is already in the loop
    for _ in range(8):
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


def sustained_clock(fn, dev, launches=40):
    """Shader clock sustained inside mq::gemm_i8_fr_kernel during a graph of `launches` back-to-back launches of `fn`: every wave stores
    [s_memtime cycles, s_memrealtime 100-MHz ticks] of its generated program (include/mobilequant_amd_tuning.h: mq_gemm_set_clock_probe);
    clock = 100 MHz x sum(cycles) / sum(ticks) over the 2 048 waves of the last launch; also per XCD (workgroup b runs on XCD b % 8).
    None when the probe is not in the library or the launch is not the generated kernel."""
    import mobilequant_amd._lib as L
    lib = L.load()
    if not hasattr(lib, "mq_gemm_set_clock_probe"):
        return None
    buf = torch.zeros(256 * 8 * 2, dtype=torch.int32, device=dev)
    lib.mq_gemm_set_clock_probe(buf.data_ptr())
    try:
        event_time(fn, launches)
    finally:
        lib.mq_gemm_set_clock_probe(None)
    d = buf.cpu().numpy().astype(np.int64).reshape(256, 8, 2)
    if not d[..., 1].all():
        return None
    mhz = 100.0 * d[..., 0].sum() / d[..., 1].sum()
    xcd = [100.0 * d[x::8, :, 0].sum() / d[x::8, :, 1].sum() for x in range(8)]
    return {"mhz": round(float(mhz), 1), "xcd": [round(float(min(xcd)), 1), round(float(max(xcd)), 1)],
            "cycles_per_wave": round(float(d[..., 0].mean()), 1)}


def cold_time(fn, flush_bytes=768 << 20, reps=10):
    """Average duration of `fn` with the operand caches flushed in front of every launch (the L2-flushing variant): a
    768 MiB fill (3x the 256 MB Infinity Cache, far beyond the 8 x 4 MB L2s) precedes each launch inside one hipGraph; the same graph
    without `fn` is subtracted.  Best of 5 for both."""
    buf = torch.empty(flush_bytes, dtype=torch.uint8, device="cuda")

    def both():
        buf.fill_(1)
        fn()

    def flush_only():
        buf.fill_(1)
    t_both = event_time(both, reps)
    t_flush = event_time(flush_only, reps)
    del buf
    return max(t_both - t_flush, 0.0)


def pmc_traffic():
    """HBM bytes per GEMM launch from the committed rocprofv3 PMC passes of this same command
    (profiles/r03/pmc_fetch + pmc_write, else r02's; FETCH_SIZE is doubled as MI355X_MICROARCH.md prescribes for 16 B/lane
    streams on gfx950).  Counters cannot be read from inside the bench, so this is the last profiled value."""
    import re
    for rnd in ("r06", "r05", "r04", "r03", "r02"):
        d = os.path.join(ROOT, "profiles", rnd)
        vals = {}
        for fn, key in (("pmc_fetch.summary.txt", "FETCH_SIZE"), ("pmc_write.summary.txt", "WRITE_SIZE")):
            try:
                txt = open(os.path.join(d, fn)).read()
            except OSError:
                break
            m = re.search(r"kernel: mq::gemm_i8_fr_kernel\([^\n]*\n(?:\s+\S+\s+[\d.]+[^\n]*\n)*?\s+" + key + r"\s+([\d.]+)", txt)
            if not m:
                break
            vals[key] = float(m.group(1)) * 1024.0
        if len(vals) == 2:
            return 2.0 * vals["FETCH_SIZE"] + vals["WRITE_SIZE"], f"LAST PROFILED, not measured in this run: profiles/{rnd}/pmc_fetch.summary.txt + pmc_write.summary.txt"
    return None, None


def cpu_baseline():
    """The reference's simulated path on this box's host cores: torch CPU kernels with
    set_num_threads(physical cores), op for op what qmodule.py:251-358 executes -- the weight re-quantised on every
    QLinear.forward (qmodule.py:346-347), fp32 F.linear, ~8 elementwise ops per Quantizer.forward
    (oracle/mq_oracle_torch.py, checked bit-exactly against the numpy oracle in tests/).  Two bounded samples: the headline
    QLinear (2048 x 2048 -> 5632), and one whole TinyLlama-shaped decoder layer at S = 2048.  Baseline only, not a target."""
    from oracle import mq_oracle_torch as T
    cores = T.physical_cores()
    prev = torch.get_num_threads()
    # best of >= 5 runs after warm-up, the thread count stated -- and swept (on a 128-core
    # host the all-cores figure of a 47-GOP problem scattered 0.08 ... 0.42 TOPS between rounds: thread wake-up and NUMA placement, not
    # arithmetic).  Every candidate count gets the same bounded budget; `value` is the best single call of the best count, the spread of
    # that count's calls travels with it.
    counts = sorted({c for c in (cores, max(cores // 2, 1), max(cores // 4, 1), 16, 8) if 1 <= c <= cores})
    sweep = {}
    try:
        g = torch.Generator().manual_seed(1337)
        x = torch.randn(M, K, generator=g)
        w = torch.randn(N, K, generator=g) * 0.02
        wq, iq, oq = T.Quantizer(8), T.Quantizer(8).set_range(float(x.min()), float(x.max())), T.Quantizer(8)
        y = x @ w.T
        oq.set_range(float(y.min()), float(y.max()))
        del y
        with torch.no_grad():
            for n in counts:
                torch.set_num_threads(n)
                T.qlinear(x, w, None, wq, iq, oq)          # warm-up; fixes the weight grid like the reference's first forward
                T.qlinear(x, w, None, wq, iq, oq)
                times, t0 = [], time.perf_counter()
                while len(times) < 5 or (time.perf_counter() - t0 < 3.0 and len(times) < 60):
                    t1 = time.perf_counter()
                    T.qlinear(x, w, None, wq, iq, oq)
                    times.append(time.perf_counter() - t1)
                times.sort()
                sweep[n] = {"best_s": round(times[0], 4), "median_s": round(times[len(times) // 2], 4), "worst_s": round(times[-1], 4), "calls": len(times)}
            best_n = min(sweep, key=lambda n: sweep[n]["best_s"])
            dt, reps = sweep[best_n]["best_s"], sweep[best_n]["calls"]
            torch.set_num_threads(best_n)
            layer = T.SimLayer()
            xl = torch.randn(M, K, generator=g)
            layer.forward(xl)
            ltimes, t1 = [], time.perf_counter()
            while len(ltimes) < 5 or (time.perf_counter() - t1 < 8.0 and len(ltimes) < 50):
                t2 = time.perf_counter()
                layer.forward(xl)
                ltimes.append(time.perf_counter() - t2)
            ltimes.sort()
            dtl, lreps = ltimes[0], len(ltimes)
    finally:
        torch.set_num_threads(prev)
    layer_ops = 2.0 * M * (2048 * 2048 * 2 + 2048 * 256 * 2 + 2048 * 5632 * 3) + 2.0 * 2 * 32 * M * M * 64
    # `value` is the MEDIAN call of the best thread count (a trend-worthy figure: the best single call scattered 4 x
    # between boxes); the best and the worst call travel beside it
    med = sweep[best_n]["median_s"]
    return {"value": round(OPS_PER_STEP / med / 1e12, 4), "unit": "TOPS", "cores": int(best_n), "kind": "port",
            "seconds_per_step": round(med, 4), "threads": int(best_n), "physical_cores": int(cores),
            "value_best": round(OPS_PER_STEP / dt / 1e12, 4), "value_worst": round(OPS_PER_STEP / sweep[best_n]["worst_s"] / 1e12, 4),
            "thread_sweep_seconds": {str(n): v for n, v in sweep.items()},
            "sample": f"median of {reps} calls (after 2 warm-up calls) of the torch-CPU restatement of QLinear.forward (weight fake-quant + input "
                      f"fake-quant + fp32 F.linear + output fake-quant) at M={M},K={K},N={N}, torch.set_num_threads({best_n}) -- the fastest of "
                      f"the swept counts {counts} on {cores} physical cores; best / worst of the same calls beside it",
            "layer": {"seconds_per_layer": round(dtl, 4), "value": round(layer_ops / dtl / 1e12, 4), "unit": "TOPS-equivalent", "threads": int(best_n),
                      "sample": f"best of {lreps} forwards of one TinyLlama-shaped W8A8 decoder layer (2 QRMSNorm, 7 QLinear, 2 QMatMul, QSiLU; "
                                "mixed-precision rules of ptq/mobilequant.py:175-201) at S=2048"}}


def bench_decode_full(dev, context=256, steps=64, wbits=8, family="tinyllama", wsym=False, wpc=None, cache_len=1024, attn_splits=None,
                      contexts=None, also_contexts=None, launches=4):
    """TinyLlama-1.1B-shaped W8A8 decode, the WHOLE step (sim_model.py:160-221 on the quantized module graph): random-init fp32
    model -> the reference's surgery (create_sim_qmodel + the mixed-precision rules of ptq/mobilequant.py:175-201) -> ranges from
    one calibration pass of this package -> DecodeEngine: per layer 4 fused launches (launches=4, round 6: norm + q|k|v stream; RoPE / cache /
    qk_bmm / softmax / pv_bmm + o_proj's contraction; o_proj's epilogue + residual + norm + w1|w3 stream + QSiLU * (.) + w2 input quantizer;
    w2 + residual) or the 5 of rounds 2-5 (launches=5: o_proj + residual as a launch of its own), final norm + fp32 lm_head, embedding
    row; one hipGraph per token, position in device memory.  Timed at a context of `context` tokens."""
    import mobilequant_amd as mq
    from mobilequant_amd.calibration import get_act_range
    from mobilequant_amd.decode import DecodeEngine
    from mobilequant_amd.llama import LlamaForCausalLM, LlamaShape
    # family: the leaf graph (BASELINE.json configs[1] / [2] / [3]): tinyllama | stablelm_2_1_6b (LayerNorm, q|k|v bias, 25 % rotary) |
    # gemma_2b (head_dim 256, 8 / 1 heads, GeGLU, FFN 16384, vocab 256000, scaled embeddings)
    shape = getattr(LlamaShape, family)(max_pos=max(2048, cache_len))
    model = LlamaForCausalLM(shape)
    model.reset_parameters(seed=1337)
    model = model.to(dev).eval().requires_grad_(False)
    g = torch.Generator().manual_seed(1337)
    calib = [torch.randint(3, shape.vocab, (1, 256), generator=g) for _ in range(2)]
    act = get_act_range(model, calib)
    a8 = mq.QuantConfig(bitwidth=8)
    per_channel = (wbits != 8) if wpc is None else wpc
    mq.create_sim_qmodel(model, mq.QuantConfig(bitwidth=wbits, is_per_channel=per_channel, is_symmetric=wsym), a8)
    for name, mod in model.named_modules():               # ptq/mobilequant.py:175-201
        if isinstance(mod, mq.QLinear):
            if "w2" in name:
                mod.weight_quantizer.qcfg.is_per_channel = True
                mod.output_quantizer.qcfg.bitwidth = 16
            elif "o_proj" in name:
                mod.output_quantizer.qcfg.bitwidth = 16
        elif isinstance(mod, (mq.QRMSNorm, mq.QLayerNorm)):
            mod.input_quantizer.qcfg.bitwidth = 16
            mod.weight_quantizer.qcfg.bitwidth = 16
            mod.weight_quantizer.qcfg.is_symmetric = False
            mod.weight_quantizer.qcfg.is_per_channel = False
        elif isinstance(mod, mq.QMatMul):
            if "qk_bmm" in name:
                mod.output_quantizer.qcfg.bitwidth = 16
            if "pv_bmm" in name:
                mod.input_quantizer.qcfg.bitwidth = 16
    mq.set_scale_and_offset(model, act, "buffer")
    eng = DecodeEngine(model, cache_len=cache_len, attn_splits=attn_splits, launches=launches)
    for p in model.parameters():                            # the float weights of the decoder layers are no longer needed
        if p.dim() == 2 and p.shape[0] != shape.vocab:
            p.data = torch.empty(0, device=dev)
    torch.cuda.empty_cache()
    eng.fill_cache_random(context)                          # a context's worth of cached keys / values
    eng.tok.fill_(17)
    eng.capture()
    for _ in range(3):
        eng.step()
    torch.cuda.synchronize()
    def timed_at(ctx):
        graph = eng.graph_long if eng.graph_long is not None and ctx >= eng._long_threshold() else eng.graph      # what step() replays there
        best = float("inf")
        for _ in range(3):
            eng.set_position(ctx)
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            for _ in range(steps):
                graph.replay()
            e1.record()
            e1.synchronize()
            best = min(best, e0.elapsed_time(e1) / steps)
        return best
    if contexts:                                            # sweep (tools/decode_context_sweep.py): ms per token at each context
        eng.fill_cache_random(max(contexts))
        return {c: round(timed_at(c), 4) for c in contexts}
    t = timed_at(context) * 1e-3
    by_context = None
    if also_contexts:                                       # the same engine at longer caches: tok/s per context
        eng.fill_cache_random(max(also_contexts))
        by_context = {str(c): round(1e3 / timed_at(c), 1) for c in also_contexts if c + steps < cache_len}
    kv_bytes = shape.layers * 2 * shape.kv_heads * (context + steps // 2) * shape.head_dim      # int8 indices
    total = eng.weight_bytes + eng.head_bytes + kv_bytes
    return {"decode_tok_s": round(1.0 / t, 1), "ms_per_token": round(t * 1e3, 4), "context": context,
            "int8_weight_GB_per_token": round(eng.weight_bytes / 1e9, 4), "lm_head_fp32_GB_per_token": round(eng.head_bytes / 1e9, 4),
            "kv_cache_GB_per_token": round(kv_bytes / 1e9, 4), "achieved_GBps": round(total / t / 1e9, 1),
            "weight_stream_GBps": round(eng.weight_bytes / t / 1e9, 1), "peak_GBps": 8000.0, "frac_of_hbm_peak": round(total / t / 8e12, 4),
            "kernels_per_token": len(eng.phases) + 2, "launches_per_layer": eng.launches, "decode_tok_s_by_context": by_context,
            "scope": f"FULL decode step, {family} shape, W{wbits}A8 recipe (16-bit norm inputs / o_proj / w2 / qk_bmm outputs): embedding, {shape.layers} x "
                     + ("[norm+qkv, RoPE / cache append / attention over the static KV cache + o_proj's contraction (split-K int32 atomics), o_proj's "
                        "epilogue+residual+norm+w1|w3+SiLU*mul+quantize, w2+residual]" if eng.launches == 4 else
                        "[norm+qkv, attention over the static KV cache, o_proj+residual, norm+w1|w3+SiLU*mul+quantize, w2+residual]")
                     + f", final norm + fp32 lm_head; batch 1, one hipGraph per token ({eng.launches} launches per layer; from "
                       f"{eng._long_threshold()} cached positions on a second graph with the long-cache attention launch)"}


def _stub_gemms(model):
    """Replaces the forward of every nn.Linear and FMatMul by 'return a resident tensor of the output's shape' (one buffer per shape,
    N(0, 1) values, allocated at first use).  The hooks still receive -- and fully read -- inputs and outputs of the real shapes; what
    disappears is the fp32 library GEMM time, which is not this package's code.  Statistics are no longer those of the model."""
    import types
    from mobilequant_amd.quantization.fp_ops import FMatMul
    pool = {}

    def buf(shape, dev):
        key = (tuple(shape), str(dev))
        if key not in pool:
            pool[key] = torch.randn(shape, device=dev)
        return pool[key]

    for m in model.modules():
        if isinstance(m, torch.nn.Linear):
            m.forward = types.MethodType(lambda self, x: buf(tuple(x.shape[:-1]) + (self.out_features,), x.device), m)
        elif isinstance(m, FMatMul):
            m.forward = types.MethodType(lambda self, a, b: buf(tuple(a.shape[:-1]) + (b.shape[-1],), a.device), m)
    return pool


def calibration_run(dev, rank, world, layers, n_samples, seq, per_channel, stub_gemm=False):
    """ptq/generate_act_range.py:49-122 data-parallel: every rank holds the same random-init TinyLlama-shaped fp32 model
    (mobilequant_amd/llama.py: the reference's leaf-module graph incl. the two FMatMuls), runs samples rank, rank + world, ...
    through it with the min/max hooks attached (one single-pass HIP reduction per hooked tensor, device-resident running
    statistics, no host sync), then ONE all-reduce(MAX) of the packed [-min, max] buffer (RCCL over xGMI)."""
    from mobilequant_amd.calibration import ActRangeCollector
    from mobilequant_amd.llama import LlamaForCausalLM, LlamaShape
    shape = LlamaShape.tinyllama(layers=layers, max_pos=seq)
    model = LlamaForCausalLM(shape)
    model.reset_parameters(seed=1337)                      # identical weights on every rank
    model = model.to(dev).eval()
    if stub_gemm:
        _stub_gemms(model)
    g = torch.Generator().manual_seed(1337)
    pool = min(n_samples, 16)
    ids = torch.randint(3, shape.vocab, (pool, seq), generator=g).to(dev)      # ids as harness_eval draws them
    col = ActRangeCollector(model, per_channel).attach()
    with torch.no_grad():
        model(ids[0:1])                                    # warm-up (sample 0 is part of the stream anyway: min / max are idempotent)
        barrier(world)
        t0 = time.perf_counter()
        mine = list(range(rank, n_samples, world)) or [rank % n_samples]
        for i in mine:
            model(ids[i % pool][None])
        col.all_reduce()
        barrier(world)
        dt = time.perf_counter() - t0
        hooked_bytes = col.bytes_seen / (len(mine) + 1)    # per sample (the warm-up sample was hooked too)
        fused_bytes = getattr(col, "bytes_fused", 0) / (len(mine) + 1)   # statistics folded into the pass that produces the tensor
        # the same samples in the same calibration-mode graph with the hook reductions switched off: what the model itself costs (fp32
        # library GEMMs, the score chain -- whose two statistics ride in its one pass -- RoPE, SiLU: not the hooked reductions)
        col._update = lambda *a, **kw: None
        k = min(len(mine), 8)
        torch.cuda.synchronize()
        t1 = time.perf_counter()
        for i in mine[:k]:
            model(ids[i % pool][None])
        torch.cuda.synchronize()
        t_model = (time.perf_counter() - t1) / k
        del col._update
        col.detach()
    dt = max_over_ranks(dt, world)
    per_sample = dt / max(len(mine), 1)
    act = col.act_dict()
    return {"seconds": dt, "samples_per_s": n_samples / dt, "collectives": getattr(col, "n_collectives", 0), "tensors": len(col.slots),
            "hooked_bytes_per_sample": hooked_bytes, "fused_bytes_per_sample": fused_bytes, "model_seconds_per_sample": t_model,
            "reduction_seconds_per_sample": max(per_sample - t_model, 0.0), "modules": len(act)}


def decode_block(dev):
    """The decode half of the metric: the FULL step on the TinyLlama-1.1B leaf graph (W8A8; contexts 256 / 1024 / 2048), the reference's
    W4A8 deployment mode, and BASELINE.json configs[2] / [3] on their own leaf graphs (random-init weights of those architectures)."""
    decode = bench_decode_full(dev, cache_len=2176, also_contexts=(256, 1024, 2048))
    torch.cuda.empty_cache()
    w4 = bench_decode_full(dev, wbits=4)                    # the reference's deployment mode: packed 4-bit per-channel weights
    decode["full_step_w4a8"] = {k: w4[k] for k in ("decode_tok_s", "ms_per_token", "int8_weight_GB_per_token", "weight_stream_GBps", "scope")}
    torch.cuda.empty_cache()
    # BASELINE.json configs[2] / [3] on their own leaf graphs (random-init weights of those architectures)
    keys = ("decode_tok_s", "ms_per_token", "int8_weight_GB_per_token", "lm_head_fp32_GB_per_token", "achieved_GBps", "frac_of_hbm_peak",
            "kernels_per_token", "scope")
    sl = bench_decode_full(dev, wbits=8, family="stablelm_2_1_6b", wpc=True)
    decode["stablelm_2_1_6b_w8a8_per_channel"] = {k: sl[k] for k in keys}
    torch.cuda.empty_cache()
    gm = bench_decode_full(dev, wbits=4, family="gemma_2b", wsym=True)
    decode["gemma_2b_w4a8_symmetric"] = {k: gm[k] for k in keys}
    torch.cuda.empty_cache()
    return decode


def bench_minmax(dev, seq):
    """HIP-event time of the calibration reductions alone on the dominant hooked tensors of one TinyLlama sample."""
    from mobilequant_amd import ops
    res = {}
    for name, shape in (("qk_bmm.output [1,32,S,S]", (32 * seq, seq)), ("w1.output [1,S,5632]", (seq, 5632)), ("q_proj.input [1,S,2048]", (seq, 2048))):
        x = torch.randn(shape, device=dev)
        mn, mx = ops.minmax_new(1, dev)
        t = event_time(lambda: ops.minmax_tensor_(x, mn, mx), 20)
        cm, cx = ops.minmax_new(shape[1], dev)
        tc = event_time(lambda: ops.minmax_cols_(x, cm, cx), 20)
        nbytes = x.numel() * 4
        res[name] = {"bytes": nbytes, "minmax_tensor_us": round(t * 1e6, 2), "minmax_tensor_GBps": round(nbytes / t / 1e9, 1),
                     "minmax_cols_us": round(tc * 1e6, 2), "minmax_cols_GBps": round(nbytes / tc / 1e9, 1)}
        del x
    return res


def bench_calibration(args, rank, world, dev):
    """BASELINE.json configs[4]: generate_act_range over 512 calibration samples, data-parallel with one RCCL all-reduce."""
    r = calibration_run(dev, rank, world, args.calib_layers, args.calib_samples, args.calib_seq, args.per_channel, args.calib_stub_gemm)
    if rank != 0:
        return
    mm = bench_minmax(dev, args.calib_seq)
    cpu = None
    if world == 1 and not args.no_cpu_baseline:
        # the oracle's running min / max (oracle/mq_oracle.py: compute_min_max_from_tensor + update_act_range) over a bounded sample of
        # the hooked bytes; samples/s = rate / hooked bytes per sample
        import numpy as np
        from oracle import mq_oracle as O
        x = np.random.default_rng(0).standard_normal((4 * args.calib_seq, args.calib_seq), dtype=np.float32)
        orc = O.ActRangeOracle(args.per_channel)
        t0, n = time.perf_counter(), 0
        while time.perf_counter() - t0 < 10.0:
            orc.update("qk_bmm", "output", x)
            n += 1
        rate = n * x.nbytes / (time.perf_counter() - t0)
        cpu = {"value": round(rate / max(r["hooked_bytes_per_sample"], 1), 5), "unit": "samples/s", "cores": 1, "kind": "port",
               "GBps": round(rate / 1e9, 2),
               "sample": f"{n} ActRangeOracle.update calls (oracle/mq_oracle.py, numpy) over a [{4 * args.calib_seq}, {args.calib_seq}] fp32 tensor (~10 s); "
                         "samples/s = bytes/s / hooked bytes per sample (reductions only, no model forward)"}
    big = mm["qk_bmm.output [1,32,S,S]"]
    key = "minmax_cols" if args.per_channel else "minmax_tensor"
    achieved = big["bytes"] / (big[key + "_us"] * 1e-6) / 1e9
    info = None
    try:
        import mobilequant_amd._lib as L
        info = L.device_info()
    except Exception:
        pass
    print(json.dumps({
        "metric": "activation-range calibration samples/s (ptq/generate_act_range.py, data-parallel + one RCCL all-reduce)",
        "value": round(r["samples_per_s"], 3), "unit": "samples/s", "n_gpus": world, "steps": args.calib_samples, "warmup": 1,
        "ms_per_step": round(1e3 * r["seconds"] / max(args.calib_samples, 1), 4), "higher_is_better": True, "scaling": "strong",
        "vs_baseline": None, "dtype": "f32", "data": "synthetic",
        "config": {"workload": f"BASELINE.json configs[4]: {args.calib_samples} samples x S={args.calib_seq} through a random-init fp32 "
                               f"TinyLlama-1.1B-shaped decoder ({args.calib_layers} layers, hidden 2048, 32/4 heads, FFN 5632; hooks on every "
                               "Linear / HFRMSNorm / SiLU / FMatMul leaf), " + ("per-channel" if args.per_channel else "per-tensor")
                               + " running [min, max], round-robin shards", "parallelism": f"dp{world}", "collectives": r["collectives"],
                   "tensors_tracked": r["tensors"], "device": info,
                   "gemms": "stubbed (resident output tensors; --calib-stub-gemm)" if args.calib_stub_gemm else "fp32 library GEMMs (rocBLAS)"},
        "roofline": {"bound": "hbm", "kernel": f"mq::{key}_kernel on the [32*S, S] attention scores (the largest hooked tensor)",
                     "achieved": round(achieved, 1), "peak": 8000.0, "unit": "GB/s", "frac": round(achieved / 8000.0, 4),
                     "algorithmic_bytes_per_launch": big["bytes"], "traffic": None, "per_tensor_shapes": mm},
        "breakdown": {"hooked_bytes_per_sample": int(r["hooked_bytes_per_sample"]),
                      "model_ms_per_sample (fp32 library GEMMs, softmax: not the hot path)": round(1e3 * r["model_seconds_per_sample"], 3),
                      "reduction_ms_per_sample": round(1e3 * r["reduction_seconds_per_sample"], 3),
                      "reduction_GBps": round(r["hooked_bytes_per_sample"] / max(r["reduction_seconds_per_sample"], 1e-9) / 1e9, 1)},
        "cpu_baseline": cpu}))


def main():
    args = parse()
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs an MI355X (torch.cuda.is_available() is False); there is no CPU path")
    maybe_spawn(args)
    rank, world, local = dist_setup(args)
    dev = torch.device("cuda", local)
    import mobilequant_amd._lib as L
    from mobilequant_amd._lib import MQ_F16, MQ_F32, MQ_U8
    info = L.device_info()
    if args.workload == "calibration":
        bench_calibration(args, rank, world, dev)
        if world > 1:
            import torch.distributed as dist
            dist.barrier()
            dist.destroy_process_group()
        return

    with torch.no_grad():
        Step.tiled_ok = not args.row_major_activations
        step = Step(dev, MQ_U8, seed=rank)
        pipelined = args.overlap and not args.no_graph
        sec = run_steps(step, args.steps, args.warmup, world, use_graph=not args.no_graph, pipelined=pipelined)
        spread = [round(t * 1e3, 5) for t in getattr(run_steps, "last", [sec])]
        run_steps.host_wall_main = max_over_ranks(run_steps.host_wall, world)
        host_spread = [round(t * 1e3, 5) for t in run_steps.host_last]
        sec = max_over_ranks(sec, world)
        value = world * OPS_PER_STEP / sec / 1e12

        extras = {}
        roof = cpu = decode = None
        # the data-parallel half of the hot path, bounded: 8 samples per rank through 2 TinyLlama-shaped layers + the single
        # all-reduce -- so every multi-GPU run of this file also drives the RCCL path (full configs[4]: --workload calibration)
        if not args.headline_only:
            cal = calibration_run(dev, rank, world, layers=2, n_samples=8 * world, seq=2048, per_channel=False)
            extras["calibration_dp"] = {"samples_per_s": round(cal["samples_per_s"], 2), "n_gpus": world, "samples": 8 * world, "layers": 2,
                                        "seq": 2048, "collectives": cal["collectives"], "scaling": "weak (8 samples per rank)"}
            torch.cuda.empty_cache()
            # BASELINE.json configs[4] at its full size in every run of this file: 512 samples x S = 2048 through the 22-layer graph,
            # sharded over the ranks (STRONG scaling: the driver's N = 1, 2, 4, 8 runs time the same 512 samples), GEMMs stubbed so the
            # timed region is the hot path -- hooked reductions + ONE all-reduce (--workload calibration prints the same as the headline)
            cal = calibration_run(dev, rank, world, layers=22, n_samples=512, seq=2048, per_channel=False, stub_gemm=True)
            extras["calibration_512_stub_gemm"] = {
                "samples_per_s": round(cal["samples_per_s"], 2), "seconds": round(cal["seconds"], 3), "n_gpus": world, "samples": 512, "layers": 22,
                "seq": 2048, "collectives": cal["collectives"], "tensors_tracked": cal["tensors"], "scaling": "strong (512 samples over all ranks)",
                "hooked_GB_per_sample": round(cal["hooked_bytes_per_sample"] / 1e9, 2),
                "statistics_taken_in_the_producing_pass_GB_per_sample": round(cal["fused_bytes_per_sample"] / 1e9, 2),
                "reduction_ms_per_sample": round(1e3 * cal["reduction_seconds_per_sample"], 3),
                "reduction_GBps_per_gpu": round(cal["hooked_bytes_per_sample"] / max(cal["reduction_seconds_per_sample"], 1e-9) / 1e9, 1)}
            torch.cuda.empty_cache()
        if rank == 0:
            # dominant kernel alone, HIP events on its stream
            t_gemm = event_time(step.gemm, 50)
            t_quant = event_time(lambda: step.quantize(0), 50)
            achieved = OPS_PER_STEP / t_gemm / 1e12
            t_cold = cold_time(step.gemm)
            clock = sustained_clock(step.gemm, dev)
            # the same launch on zero-filled operands: the chip clocks to the load, so this is the binary's time at
            # the clock an idle datapath sustains -- the gap to avg_launch_us is power / clock, not schedule
            keep_a, keep_w = step.a8s[0].clone(), step.w8.clone()
            step.a8s[0].zero_(); step.w8.zero_()
            t_zero = event_time(step.gemm, 50)
            clock_zero = sustained_clock(step.gemm, dev)
            step.a8s[0].copy_(keep_a); step.w8.copy_(keep_w)
            del keep_a, keep_w
            traffic, traffic_src = pmc_traffic()
            roof = {"bound": "mfma", "kernel": ("mq::gemm_i8_fr_kernel (mq_w8a8_linear_tiled: free-running whole-kernel gfx950 ISA, fragment-blocked "
                                                "activations)" if step.tiled else "mq::gemm_i8_kernel (mq_w8a8_linear)"), "achieved": round(achieved, 1),
                    "peak": INT8_MFMA_PEAK_TOPS, "unit": "TOPS", "frac": round(achieved / INT8_MFMA_PEAK_TOPS, 4),
                    "avg_launch_us": round(t_gemm * 1e6, 2),
                    # the clock the chip SUSTAINED inside this kernel, read by the kernel itself (s_memtime cycles / s_memrealtime ticks of
                    # every wave, mq_gemm_set_clock_probe): the dense peak is quoted at 2.4 GHz, the chip clocks to its power budget
                    "sustained_mhz": clock and clock["mhz"], "sustained_mhz_xcd_min_max": clock and clock["xcd"],
                    "frac_clock_adjusted": clock and round(achieved / (INT8_MFMA_PEAK_TOPS * clock["mhz"] / 2400.0), 4),
                    "ceiling_at_sustained_clock": clock and round(INT8_MFMA_PEAK_TOPS * clock["mhz"] / 2400.0, 1),
                    "clock_note": "ceiling_at_sustained_clock = peak x sustained_mhz / 2400 (TOPS): what the matrix pipe can issue at the clock "
                                  "the silicon holds under THIS kernel and data; frac_clock_adjusted = achieved / that ceiling; frac (against "
                                  "the nominal 2.4 GHz peak) is the headline figure",
                    # tools/mfma_energy_probe.cpp on this chip (profiles/r05/mfma_energy_probe.log): NOTHING but v_mfma_i32_16x16x64_i8 on
                    # register-resident quantised-Gaussian operands, 2 waves per SIMD on all 256 CUs, holds ~2.0-2.1 GHz = 4 160 TOPS in
                    # steady state (zeros: 2.4 GHz, 4 950); the 32x32x32 form holds 1.79 GHz = 3 660 TOPS -- the data-dependent power of
                    # the matrix pipe itself caps a random-data int8 GEMM at ~0.83 of the nominal peak before any operand is moved
                    "mfma_only_ceiling": {"tops_gaussian_16x16x64": 4160, "mhz": 2050, "tops_gaussian_32x32x32": 3660, "tops_zeros": 4950,
                                          "frac_of_it": round(achieved / 4160.0, 4), "source": "profiles/r05/mfma_energy_probe.log (last profiled)"},
                    "cold_caches": {"avg_launch_us": round(t_cold * 1e6, 2), "frac": round(OPS_PER_STEP / t_cold / 1e12 / INT8_MFMA_PEAK_TOPS, 4),
                                    "how": "768 MiB fill in front of every launch in one hipGraph, minus the same graph without the GEMM"},
                    "zero_filled_operands": {"avg_launch_us": round(t_zero * 1e6, 2), "frac": round(OPS_PER_STEP / t_zero / 1e12 / INT8_MFMA_PEAK_TOPS, 4),
                                             "sustained_mhz": clock_zero and clock_zero["mhz"],
                                             "how": "the same launch with all-zero int8 operands (no data-dependent switching power)"},
                    "traffic": traffic, "traffic_unit": "bytes/launch",
                    "traffic_source": traffic_src, "algorithmic_bytes_per_launch": M * K + N * K + M * N,
                    "algorithmic_ops_per_launch": OPS_PER_STEP,
                    "cycles_per_wave": clock and clock["cycles_per_wave"],
                    "notes": "MFMA busy 22 528 cycles per SIMD of cycles_per_wave (the program's own s_memtime span; two waves share a SIMD); "
                             "~2.5 us of each launch period lie outside the waves' lifetime (kernel boundary 1.2 us + write-back of the "
                             "11.5 MB of outputs): DESIGN.md 4.2, profiles/r05",
                    "quantize_kernel": {"bound": "hbm", "avg_launch_us": round(t_quant * 1e6, 2),
                                        "achieved_GBps": round((M * K * 5 + M * 4) / t_quant / 1e9, 1), "peak_GBps": 8000.0}}
            if pipelined:
                t_ser = run_steps(step, min(args.steps, 100), 5, 1, use_graph=True, pipelined=False)
                extras["serial_u8"] = {"ms_per_step": round(t_ser * 1e3, 5), "value": round(OPS_PER_STEP / t_ser / 1e12, 1),
                                       "note": "quantize and GEMM back to back on one stream"}
            for name, code in ((("out_f16", MQ_F16), ("out_f32", MQ_F32)) if args.variants and not args.headline_only else ()):
                s2 = Step(dev, code, seed=rank)
                t2 = run_steps(s2, min(args.steps, 100), 5, 1, use_graph=not args.no_graph)
                extras[name] = {"ms_per_step": round(t2 * 1e3, 5), "value": round(OPS_PER_STEP / t2 / 1e12, 1)}
                del s2
            if not args.headline_only:
                decode = decode_block(dev)
                torch.cuda.empty_cache()
                import bench_variants as BV          # BASELINE.json configs[2] / [3] GEMMs belong to the headline line; the other legs on request
                extras["other_configs"] = BV.bench_other_configs(dev)
                torch.cuda.empty_cache()
                if args.variants:
                    extras.update(BV.bench_variants(dev, step, args))
            if not args.no_cpu_baseline and not args.headline_only:
                cpu = cpu_baseline()

    mgpu = multi_gpu_report(rank, world, local, extras)        # (a collective: every rank calls it)
    if rank == 0:
        line = {
            "metric": "W8A8 QuantLinear GEMM TOPS (% int8 MFMA peak) + TinyLlama-1.1B decode tok/s",
            "value": round(value, 1), "unit": "TOPS", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": round(sec * 1e3, 5), "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
            "timing": {"what": f"exactly {args.steps} steps inside hipGraph(s) between two HIP events on their stream, behind an untimed pre-roll graph of {PREROLL_STEPS} steps, "
                               f"each repeat bracketed by barrier + synchronize; measured {REPEATS} times, value / ms_per_step = the median "
                               "(max over ranks)",
                       "ms_per_step_rank0_sorted": spread,
                       "host_clock_ms_per_step": round(getattr(run_steps, "host_wall_main", 0.0) * 1e3, 5),
                       "host_clock_ms_per_step_rank0_sorted": host_spread,
                       "host_clock_value": round(world * OPS_PER_STEP / max(getattr(run_steps, "host_wall_main", 0.0), 1e-12) / 1e12, 1),
                       "host_clock_note": "the method of rounds 1-3, also the median of the same number of repeats (max over ranks): perf_counter "
                                          "around barrier | the same graphs | barrier, no pre-roll -- it carries the graph-launch and synchronize "
                                          "latency of the host, amortised over only --steps steps.  Bench lines of r01-r03 quote THIS clock, "
                                          "r04 on quote the event clock: compare like with like across rounds"},
            "dtype": "i8", "data": "synthetic",
            "config": {"workload": "TinyLlama-1.1B W8A8 real-int8 QLinear step (BASELINE.json configs[1]): fp32 x[2048,2048] "
                                   "-> int8 quantize(+row sums) -> MFMA i8 GEMM 2048->5632 with fused dequant + 8-bit output "
                                   "quantizer -> u8 indices; per-tensor asymmetric activation ranges, per-tensor asymmetric weights",
                       "M": M, "K": K, "N": N, "parallelism": f"replicas x{world}", "graph_steps": 0 if args.no_graph else min(args.steps, MAX_GRAPH_STEPS),
                       "schedule": "quantize(batch i+1) overlapped with GEMM(batch i) on a second stream" if pipelined else "serial",
                       "pct_int8_mfma_peak": round(100 * value / world / INT8_MFMA_PEAK_TOPS, 2),
                       "decode_tok_s": decode["decode_tok_s"] if decode else None, "device": info},
            "roofline": roof, "cpu_baseline": cpu, "decode": decode, "multi_gpu": mgpu, "variants": extras,
        }
        print(json.dumps(line))
    if world > 1:
        import torch.distributed as dist
        dist.barrier()            # ranks > 0 wait for rank 0's untimed extras before tearing RCCL down
        dist.destroy_process_group()


if __name__ == "__main__":
    main()

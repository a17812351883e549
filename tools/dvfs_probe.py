#!/usr/bin/env python3
"""Steady-state timing of the GEMM variants with the shader clock read from inside the kernel (stamp build of the free-running
kernel: s_memtime vs the 100 MHz s_memrealtime).  Answers: is the headline GEMM schedule-bound or power-bound?
usage (GPU box):  MQ_LIB_PATH=mobilequant_amd/lib/frs/libmobilequant_amd.so python tools/dvfs_probe.py"""
import ctypes
import os
import sys
import time

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from mobilequant_amd import _lib, ops  # noqa: E402
from mobilequant_amd._lib import MQ_U8  # noqa: E402

lib = _lib.load()
dev = torch.device("cuda:0")
M, N, K = 2048, 5632, 2048
stamped = hasattr(lib, "mq_gemm_set_debug_buffer_")
dbg = torch.zeros(256 * 8 * 16, dtype=torch.int64, device=dev)
if stamped:
    lib.mq_gemm_set_debug_buffer_.argtypes = [ctypes.c_void_p]
    lib.mq_gemm_set_debug_buffer_(dbg.data_ptr())
    lib.mq_gemm_set_debug(16)


def problem(fill):
    g = torch.Generator().manual_seed(0)
    if fill == "uniform":
        a = torch.randint(-128, 128, (M, K), generator=g, dtype=torch.int8)
        w = torch.randint(-128, 128, (N, K), generator=g, dtype=torch.int8)
    elif fill == "gauss":      # quantised N(0,1) activations / N(0,0.02) weights on their min/max grids, as bench.py has them
        x = torch.randn(M, K, generator=g); ww = torch.randn(N, K, generator=g)
        a = (torch.round((x - x.min()) / ((x.max() - x.min()) / 255)) - 128).to(torch.int8)
        w = (torch.round((ww - ww.min()) / ((ww.max() - ww.min()) / 255)) - 128).to(torch.int8)
    else:
        a = torch.zeros(M, K, dtype=torch.int8); w = torch.zeros(N, K, dtype=torch.int8)
    a, w = a.to(dev), w.to(dev)
    a_t = a.view(M // 16, 16, K // 64, 4, 16).permute(0, 2, 3, 1, 4).contiguous().view(M, K)
    rs = a.to(torch.int32).sum(1).to(torch.int32)
    colsum = w.to(torch.int32).sum(1).to(torch.int32)
    one = torch.ones(1, device=dev)
    alpha, wzp, ct = ops.linear_epilogue_prepare(one * 0.02, one * 131, 128, one * 7e-4, one * 120, 128, colsum, K)
    out = torch.empty(M, N, dtype=torch.uint8, device=dev)
    return lambda: ops.int8_linear(a_t, w, rs, alpha, wzp, ct, None, out_scale=one * 0.05, out_offset=one * 128, out_qmin=0.,
                                   out_qmax=255., out_dtype=MQ_U8, out=out, a_tiled_rows=M)


def run(fn, ms=400):
    fn(); torch.cuda.synchronize()
    s = torch.cuda.Stream(); s.wait_stream(torch.cuda.current_stream())
    with torch.cuda.stream(s):
        fn()
    torch.cuda.current_stream().wait_stream(s)
    g = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g):
        for _ in range(100):
            fn()
    g.replay(); torch.cuda.synchronize()
    times = []
    t_end = time.perf_counter() + ms * 1e-3
    while time.perf_counter() < t_end:
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record(); g.replay(); e1.record(); e1.synchronize()
        times.append(e0.elapsed_time(e1) * 10.0)       # us per launch
    return times


def sample_power(stop, out):
    import json
    import subprocess
    while not stop.is_set():
        try:
            r = subprocess.run(["rocm-smi", "--showpower", "--showclocks", "--json"], capture_output=True, text=True, timeout=5)
            d = json.loads(r.stdout[r.stdout.index("{"):])["card0"]
            out.append((d.get("Current Socket Graphics Package Power (W)") or d.get("Average Graphics Package Power (W)"),
                        d.get("sclk clock speed:")))
        except Exception as e:      # noqa: BLE001
            out.append(("err", str(e)[:60]))
            return


def eager(fn, ms=400):
    """launch-bound check: the same kernel from a plain host loop (no graph), events around 200 launches"""
    fn(); torch.cuda.synchronize()
    times = []
    t_end = time.perf_counter() + ms * 1e-3
    while time.perf_counter() < t_end:
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(200):
            fn()
        e1.record(); e1.synchronize()
        times.append(e0.elapsed_time(e1) * 5.0)
    return times


def launch_gap():
    """Two stamp buffers, alternated launch by launch in a plain host loop: does launch N+1 start before launch N's last wave
    ends, and how long is the hole between them?  (s_memrealtime is one 100 MHz counter for the chip.)"""
    fn = problem("gauss")
    lib.mq_gemm_set_variant(11)
    bufs = [torch.zeros(256 * 8 * 16, dtype=torch.int64, device=dev) for _ in range(2)]
    for i in range(60):
        lib.mq_gemm_set_debug_buffer_(bufs[i & 1].data_ptr())
        fn()
    torch.cuda.synchronize()
    a, b = (x.cpu().numpy().reshape(-1, 16) for x in bufs)       # a: launch 58, b: launch 59
    print(f"launch N   : first wave start {0:.2f} us, last wave start {(a[:, 4].max() - a[:, 4].min()) / 100:.2f}, first wave end "
          f"{(a[:, 5].min() - a[:, 4].min()) / 100:.2f}, last wave end {(a[:, 5].max() - a[:, 4].min()) / 100:.2f}")
    print(f"launch N+1 : first wave start {(b[:, 4].min() - a[:, 4].min()) / 100:.2f} us after launch N's first, i.e. "
          f"{(b[:, 4].min() - a[:, 5].max()) / 100:.2f} us after launch N's LAST wave ended; its last wave ends at "
          f"{(b[:, 5].max() - a[:, 4].min()) / 100:.2f}", flush=True)
    if a[:, 6].any():          # round-4 stamp builds: s_memrealtime at the kernel's first instruction (before the C++ preamble)
        pre_a, pre_b = (a[:, 4] - a[:, 6]) / 100, (b[:, 4] - b[:, 6]) / 100
        print(f"entry stamps: preamble (kernel entry -> first instruction of the generated program) mean {pre_b.mean():.2f} max {pre_b.max():.2f} us; "
              f"launch N+1's first ENTRY {(b[:, 6].min() - a[:, 5].max()) / 100:.2f} us after launch N's last wave ended "
              f"(last entry {(b[:, 6].max() - a[:, 5].max()) / 100:.2f}); launch N: preamble mean {pre_a.mean():.2f}", flush=True)
    lib.mq_gemm_set_debug_buffer_(dbg.data_ptr())


if os.environ.get("DVFS_GAP") and stamped:
    launch_gap()
fills = os.environ.get("DVFS_FILLS", "uniform,gauss,zero").split(",")
variants = [int(v) for v in os.environ.get("DVFS_VARIANTS", "11,9").split(",")]
if os.environ.get("DVFS_POWER"):
    import threading
    fn = problem("gauss")
    lib.mq_gemm_set_variant(11)
    stop, samples = threading.Event(), []
    th = threading.Thread(target=sample_power, args=(stop, samples)); th.start()
    t = run(fn, ms=4000)
    stop.set(); th.join()
    print(f"power/clock samples during a 4 s steady run of variant 11 (median {np.median(t):.2f} us/launch):", samples, flush=True)
for fill in fills:
    fn = problem(fill)
    for v in variants:
        lib.mq_gemm_set_variant(v)
        t = run(fn)
        line = f"fill={fill:8s} variant {v:2d} {lib.mq_gemm_variant_name(v).decode():24s} us/launch first {t[0]:.2f} median {np.median(t):.2f} last {t[-1]:.2f} min {min(t):.2f} ({len(t)} replays of 100)"
        if stamped and v == 11:
            d = dbg.cpu().numpy().reshape(-1, 16)
            ticks = (d[:, 3] - d[:, 0]).astype(np.float64); rt = (d[:, 5] - d[:, 4]).astype(np.float64)
            alive = rt.mean() / 100
            line += (f" | in-kernel: {ticks.mean():.0f} cycles per wave, shader clock {ticks.sum() / rt.sum() * 100:.0f} MHz, wave alive {alive:.2f} us,"
                     f" outside the waves {np.median(t) - alive:.2f} us per launch")
        if os.environ.get("DVFS_EAGER"):
            te = eager(fn)
            line += f" | eager loop median {np.median(te):.2f} us/launch"
        print(line, flush=True)
lib.mq_gemm_set_variant(-1)

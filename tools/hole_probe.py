#!/usr/bin/env python3
"""What makes the kernel boundary behind the headline GEMM 6 us instead of the 1.3 us of a trivial kernel?  (profiles/r04)
Stamp build only:  MQ_LIB_PATH=mobilequant_amd/lib/frs/libmobilequant_amd.so python tools/hole_probe.py
Every test is a plain host loop of launches on one stream, stamp buffers alternated launch by launch (s_memrealtime is one 100 MHz
counter for the chip); reported: gap = first ENTRY of launch n+1 minus last EXIT of launch n, median over the last launches."""
import ctypes
import os
import sys

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from mobilequant_amd import _lib, ops  # noqa: E402
from mobilequant_amd._lib import MQ_U8  # noqa: E402

lib = _lib.load()
dev = torch.device("cuda:0")
lib.mq_gemm_set_debug_buffer_.argtypes = [ctypes.c_void_p]
if hasattr(lib, "mq_debug_stamp_"):
    lib.mq_debug_stamp_.argtypes = [ctypes.c_void_p, ctypes.c_int, ctypes.c_int, ctypes.c_int, ctypes.c_int, ctypes.c_int, ctypes.c_void_p,
                                    ctypes.c_int, ctypes.c_void_p]
lib.mq_gemm_set_debug(16)
lib.mq_gemm_set_variant(11)
NL = 40
dirty = torch.zeros(64 << 20, dtype=torch.int32, device=dev)


def gemm_problem(fill, K=2048):
    M, N = 2048, 5632
    g = torch.Generator().manual_seed(0)
    if fill == "gauss":
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
    so, oo = one * 0.05, one * 128
    return lambda: ops.int8_linear(a_t, w, rs, alpha, wzp, ct, None, out_scale=so, out_offset=oo, out_qmin=0.,
                                   out_qmax=255., out_dtype=MQ_U8, out=out, a_tiled_rows=M)


class G:
    """the stamped GEMM: entry = column 6 (round-3 builds: program start, column 4), exit = column 5 of the per-wave records"""
    def __init__(self, fill="gauss", K=2048):
        self.fn = gemm_problem(fill, K)
        self.name = f"GEMM({fill}, K={K})"

    def launch(self, buf):
        lib.mq_gemm_set_debug_buffer_(buf.data_ptr())
        self.fn()

    @staticmethod
    def times(buf):
        d = buf.cpu().numpy().reshape(-1, 16)[:2048]
        e = d[:, 6] if d[:, 6].any() else d[:, 4]
        return e.min(), e.max(), d[:, 5].min(), d[:, 5].max()


class S:
    def __init__(self, threads=256, lds=0, fat=0, spin_us=0.0, dirty_mb=0.0, blocks=256):
        self.a = (blocks, threads, lds, fat, int(spin_us * 100))
        self.dw = int(dirty_mb * (1 << 20) / 4 / blocks)
        self.blocks = blocks
        self.name = f"stamp(threads={threads}, lds={lds >> 10}K, regs={'248' if fat else 'few'}, spin={spin_us}us, dirty={dirty_mb}MB)"

    def launch(self, buf):
        b, t, l, f, sp = self.a
        lib.mq_debug_stamp_(buf.data_ptr(), b, t, l, f, sp, dirty.data_ptr(), self.dw, torch.cuda.current_stream().cuda_stream)

    def times(self, buf):
        d = buf.cpu().numpy()[:2 * self.blocks].reshape(-1, 2)
        return d[:, 0].min(), d[:, 0].max(), d[:, 1].min(), d[:, 1].max()


def chain(kernels, label):
    """launch the sequence `kernels` NL times; per position: gap in front of it, entry spread, duration"""
    n = len(kernels)
    bufs = [[torch.zeros(256 * 8 * 16, dtype=torch.int64, device=dev) for _ in range(n)] for _ in range(2)]
    for rep in range(NL):
        for i, k in enumerate(kernels):
            k.launch(bufs[rep & 1][i])
    torch.cuda.synchronize()
    # the last two repetitions: rep NL-2 in bufs[NL & 1], rep NL-1 in bufs[(NL - 1) & 1]
    seq = [(k, k.times(bufs[(NL - 2) & 1][i])) for i, k in enumerate(kernels)] + [(k, k.times(bufs[(NL - 1) & 1][i])) for i, k in enumerate(kernels)]
    print(f"-- {label}")
    for j in range(n, 2 * n):
        k, (e0, e1, x0, x1) = seq[j]
        pe0, pe1, px0, px1 = seq[j - 1][1]
        print(f"   {k.name:78s} gap after the previous kernel's last exit {(e0 - px1) / 100:6.2f} us | entries spread {(e1 - e0) / 100:5.2f} | "
              f"first entry -> last exit {(x1 - e0) / 100:6.2f} | first exit at {(x0 - e0) / 100:6.2f}", flush=True)
    period = (seq[2 * n - 1][1][3] - seq[n - 1][1][3]) / 100
    print(f"   period of the sequence {period:.2f} us", flush=True)


def per_xcd(fill="gauss", K=2048):
    """per-XCD anatomy of one steady-state GEMM launch: do the XCDs run at one clock?  (block b runs on XCD b % 8)"""
    g = G(fill, K)
    bufs = [torch.zeros(256 * 8 * 16, dtype=torch.int64, device=dev) for _ in range(2)]
    for rep in range(NL):
        g.launch(bufs[rep & 1])
    torch.cuda.synchronize()
    d = bufs[(NL - 1) & 1].cpu().numpy().reshape(-1, 16)[:2048].astype(np.float64)
    t0 = d[:, 6].min()
    if not d[:, 6].any():          # round-3 stamp build: no entry stamp
        d[:, 6] = d[:, 4]
        t0 = d[:, 4].min()
    if os.environ.get("HOLE_BRIEF"):
        clk = (d[:, 3] - d[:, 0]).sum() / (d[:, 5] - d[:, 4]).sum() * 100
        per = [((d[[b * 8 + w for b in range(256) if b % 8 == x for w in range(8)]][:, 3] - d[[b * 8 + w for b in range(256) if b % 8 == x for w in range(8)]][:, 0]).sum()
                / (d[[b * 8 + w for b in range(256) if b % 8 == x for w in range(8)]][:, 5] - d[[b * 8 + w for b in range(256) if b % 8 == x for w in range(8)]][:, 4]).sum() * 100) for x in range(8)]
        print(f"-- {g.name}: cycles/wave {(d[:, 3] - d[:, 0]).mean():.0f} (prologue {(d[:, 1] - d[:, 0]).mean():.0f}, to the tail {(d[:, 2] - d[:, 1]).mean():.0f}, "
              f"tail {(d[:, 3] - d[:, 2]).mean():.0f}); clock {clk:.0f} MHz (XCDs {min(per):.0f}..{max(per):.0f}); last exit {(d[:, 5].max() - t0) / 100:.2f} us after the first entry")
        return
    print(f"-- per-XCD anatomy, {g.name}: us after the launch's first entry (mean over the XCD's 256 waves) and shader clock = s_memtime ticks / s_memrealtime")
    for x in range(8):
        rows = np.array([b * 8 + w for b in range(256) if b % 8 == x for w in range(8)])
        e = d[rows]
        clk = (e[:, 3] - e[:, 0]).sum() / (e[:, 5] - e[:, 4]).sum() * 100
        cyc = e[:, 3] - e[:, 0]
        print(f"   XCD {x}: entry {(e[:, 6].mean() - t0) / 100:5.2f} program start {(e[:, 4].mean() - t0) / 100:5.2f} exit mean {(e[:, 5].mean() - t0) / 100:6.2f} "
              f"max {(e[:, 5].max() - t0) / 100:6.2f} | cycles/wave {cyc.mean():7.0f} (prologue {(e[:, 1] - e[:, 0]).mean():5.0f} loop {(e[:, 2] - e[:, 1]).mean():6.0f} "
              f"epilogue {(e[:, 3] - e[:, 2]).mean():5.0f}) | clock {clk:5.0f} MHz", flush=True)


if os.environ.get("HOLE_KSWEEP"):
    for K in (768, 1024, 1280, 1536, 2048, 4096):
        per_xcd("gauss", K)
    lib.mq_gemm_set_variant(-1)
    sys.exit(0)
per_xcd("gauss")
per_xcd("zero")
if os.environ.get("HOLE_ONLY") == "xcd":
    chain([G("gauss")], "GEMM, quantised-Gaussian operands")
    chain([G("zero")], "GEMM, zero-filled operands")
    lib.mq_gemm_set_variant(-1)
    sys.exit(0)
thin = S()
chain([thin], "trivial kernels back to back")
chain([S(spin_us=15.0)], "thin kernel spinning 15 us")
chain([S(threads=512, lds=138 << 10, fat=1, spin_us=15.0)], "the GEMM's footprint (512 threads, 138 KiB LDS, 248 registers), spinning 15 us, nothing written")
chain([S(threads=512, lds=138 << 10, fat=1, spin_us=0.0)], "the GEMM's footprint, no spin")
chain([S(spin_us=15.0, dirty_mb=11.5)], "thin kernel, 15 us, 11.5 MB written")
chain([S(threads=512, lds=138 << 10, fat=1, spin_us=15.0, dirty_mb=11.5)], "GEMM footprint + 15 us + 11.5 MB written")
chain([G("gauss")], "GEMM, quantised-Gaussian operands")
chain([G("zero")], "GEMM, zero-filled operands")
chain([G("gauss", 768)], "GEMM K = 768")
chain([G("gauss", 4096)], "GEMM K = 4096")
chain([G("gauss"), thin], "GEMM, trivial kernel, GEMM, ...")
chain([G("gauss"), S(threads=512, lds=138 << 10, fat=1)], "GEMM, fat trivial kernel, GEMM, ...")
chain([G("gauss"), S(spin_us=6.0)], "GEMM, thin kernel spinning 6 us, ...")
lib.mq_gemm_set_variant(-1)

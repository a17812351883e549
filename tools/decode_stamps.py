#!/usr/bin/env python3
"""Timeline of ONE decode step from in-kernel s_memrealtime stamps (100 MHz, chip-wide clock).

Needs the profiling build of the library:
    python -c "from mobilequant_amd import build; build.build(force=True, tag='stamps', extra_flags=['-DMQ_DECODE_STAMPS'])"
    MQ_LIB_PATH=mobilequant_amd/lib/stamps/libmobilequant_amd.so python tools/decode_stamps.py
Prints, per launch of the captured step graph: gap since the previous launch's last stamp, first-start -> per-phase means, span.
Env: LAYERS (default 4), CONTEXT (256), WBITS (8).
"""
import ctypes, os, sys
import numpy as np
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import mobilequant_amd as mq  # noqa: E402
from mobilequant_amd import _lib  # noqa: E402
from mobilequant_amd.calibration import get_act_range  # noqa: E402
from mobilequant_amd.decode import DecodeEngine  # noqa: E402
from mobilequant_amd.llama import LlamaForCausalLM, LlamaShape  # noqa: E402

KINDS = {0: "gemv norm (qkv)", 1: "gemv norm+gate (w1|w3)", 2: "gemv f32 in (o_proj)", 3: "gemv i8 in (w2)", 4: "attention", 5: "head"}


def build_engine(dev, layers, wbits=8, cache_len=int(os.environ.get("CACHE", "1024"))):
    shape = LlamaShape.tinyllama(max_pos=max(2048, cache_len), layers=layers)
    model = LlamaForCausalLM(shape); model.reset_parameters(seed=1); model = model.to(dev).eval().requires_grad_(False)
    g = torch.Generator().manual_seed(1)
    act = get_act_range(model, [torch.randint(3, shape.vocab, (1, 256), generator=g)])
    a8 = mq.QuantConfig(bitwidth=8)
    mq.create_sim_qmodel(model, a8 if wbits == 8 else mq.QuantConfig(bitwidth=wbits, is_per_channel=True), a8)
    for name, mod in model.named_modules():
        if isinstance(mod, mq.QLinear):
            if "w2" in name: mod.weight_quantizer.qcfg.is_per_channel = True; mod.output_quantizer.qcfg.bitwidth = 16
            elif "o_proj" in name: mod.output_quantizer.qcfg.bitwidth = 16
        elif isinstance(mod, mq.QRMSNorm): mod.input_quantizer.qcfg.bitwidth = 16; mod.weight_quantizer.qcfg.bitwidth = 16
        elif isinstance(mod, mq.QMatMul):
            if "qk_bmm" in name: mod.output_quantizer.qcfg.bitwidth = 16
            if "pv_bmm" in name: mod.input_quantizer.qcfg.bitwidth = 16
    mq.set_scale_and_offset(model, act, "buffer")
    return DecodeEngine(model, cache_len=cache_len, attn_splits=int(os.environ.get('SPLITS', '1')), prefetch=float(os.environ.get('PREFETCH', '0.5')), prefetch_delay_us=float(os.environ.get('PFDELAY', '2.5')), launches=int(os.environ.get('LAUNCHES', '4')))


def main():
    dev = torch.device("cuda:0")
    layers, context, wbits = int(os.environ.get("LAYERS", "4")), int(os.environ.get("CONTEXT", "256")), int(os.environ.get("WBITS", "8"))
    eng = build_engine(dev, layers, wbits)
    if os.environ.get("NOHEAD"):                        # what-if: no 262 MB lm_head stream -> a small model's weights stay in the Infinity Cache
        import types
        real = _lib.call
        def _launch(self):
            _lib.call = lambda name, *a: None if name == "mq_decode_head" else real(name, *a)
            try:
                type(self)._launch(self)
            finally:
                _lib.call = real
        eng._launch = types.MethodType(_launch, eng)
    eng.fill_cache_random(context); eng.tok.fill_(17)
    lib = _lib.load()
    if not hasattr(lib, "mq_decode_set_stamps_"):
        sys.exit("this library was built without -DMQ_DECODE_STAMPS")
    lib.mq_decode_set_stamps_.argtypes = [ctypes.c_void_p, ctypes.c_longlong]
    lib.mq_decode_stamp_log_.argtypes = [ctypes.POINTER(ctypes.c_longlong), ctypes.c_int]
    cap = 1 << 22
    buf = torch.zeros(cap, dtype=torch.int64, device=dev)
    lib.mq_decode_set_stamps_(buf.data_ptr(), cap)
    eng.capture()                                       # warm launch + captured launch: the log holds both, the graph the second
    log = (ctypes.c_longlong * (3 * 4096))()
    n = lib.mq_decode_stamp_log_(log, 4096)
    entries = [(log[3 * i], log[3 * i + 1], log[3 * i + 2]) for i in range(n)][n // 2:]
    for _ in range(5): eng.graph.replay()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(32): eng.graph.replay()
    e1.record(); e1.synchronize()
    print(f"graph ms/token {e0.elapsed_time(e1) / 32:.4f}  layers {layers} context {context} W{wbits} (stamped build: ~+0.1 us per launch)")
    buf.zero_(); eng.set_position(context)
    eng.graph.replay(); torch.cuda.synchronize()
    st = buf.cpu().numpy().astype(np.int64)
    prev_end = None
    print(f"{'launch':28s} {'gap':>6s} {'ramp':>6s} | per-WG means since the launch's first stamp (us): s1 s2 s3 s4 s5 | span")
    agg = {}
    for kind, grid, base in entries:
        s = st[base:base + grid * 16].reshape(grid, 16).astype(np.float64) / 100.0     # us
        t0 = s[:, 0].min()
        gap = t0 - prev_end if prev_end is not None else float("nan")
        ramp = s[:, 0].max() - t0
        cols = []
        for k in (1, 2, 3, 4, 5, 8, 6, 7, 9, 10, 14, 15):
            v = s[:, k][s[:, k] > 0]
            cols.append(v.mean() - t0 if v.size else float("nan"))
        raw = st[base:base + grid * 16].reshape(grid, 16)
        ok = (raw[:, 12] > 0) & (raw[:, 11] > 0) & (raw[:, 5] > raw[:, 13])
        mhz = float(np.mean((raw[ok, 12] - raw[ok, 11]) / ((raw[ok, 5] - raw[ok, 13]) / 100.0))) if ok.any() else float("nan")
        end = s[:, 1:6].max()
        span = end - t0
        prev_end = end
        agg.setdefault(kind, []).append([gap, ramp] + cols + [span, mhz])
    for kind, rows in agg.items():
        r = np.nanmean(np.array(rows[1:] if len(rows) > 1 else rows), axis=0)
        print(f"{KINDS[kind]:28s} {r[0]:6.2f} {r[1]:6.2f} | " + " ".join(f"{x:6.2f}" for x in r[2:7]) + f" | {r[14]:6.2f}   (n={len(rows)})"
              f"  x@w0 {r[8]:.2f} ss-math {r[7]:.2f} wave-sum {r[10]:.2f} [s1] quant-math {r[11]:.2f} [s2]  shader clock {r[15]:.0f} MHz  kernarg@w0 {r[12]:.2f} @w8 {r[13]:.2f}")
    tot = sum(np.nansum(np.array(rows)[:, [0, 14]]) for rows in agg.values())
    print(f"sum of gaps + spans: {tot:.1f} us")


if __name__ == "__main__":
    main()

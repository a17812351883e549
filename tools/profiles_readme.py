#!/usr/bin/env python3
"""Writes profiles/<round>/README.md FROM the files in that directory: every number in the README is parsed out of a committed
summary / log / JSON by this script, so the tables cannot drift from the captures (round-2 review: the hand-written README quoted
numbers its own summaries did not contain).

usage: python tools/profiles_readme.py [profiles/r03]"""
import json
import os
import re
import sys

CAPTIONS = {
    "trace.summary.txt": "`tools/prof_bench.sh <tag> pmc`: `rocprofv3 --kernel-trace --stats -- python bench.py --steps 100 --warmup 10 --no-cpu-baseline` (every leg of the bench)",
    "pmc_sq.summary.txt": "same script, `--pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_VALU_MFMA_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_WAVES SQ_INSTS_VALU_MFMA_MOPS_I8 --kernel-trace -- python bench.py --steps 100 --warmup 10 --headline-only`",
    "pmc_lds.summary.txt": "same, `--pmc SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_ACTIVE_INST_LDS SQ_WAIT_INST_LDS SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INST_CYCLES_VMEM_RD GRBM_GUI_ACTIVE`",
    "pmc_fetch.summary.txt": "same, `--pmc FETCH_SIZE` (KiB; doubled below as MI355X_MICROARCH.md prescribes for gfx950)",
    "pmc_write.summary.txt": "same, `--pmc WRITE_SIZE TCC_HIT_sum TCC_MISS_sum`",
    "layer_trace.summary.txt": "`rocprofv3 --kernel-trace --stats -- python tools/prof_layer.py`: one fused TinyLlama W8A8 decoder layer at S = 2048 (plus its one-off calibration kernels)",
    "layer_trace_w4.summary.txt": "the same with `MQ_LAYER_WBITS=4` (W4A8 recipe), captured BEFORE the 4-bit o_proj took the attention kernel's image (it shows the extra quantize launch)",
    "gemma_layer_trace.summary.txt": "`bench.bench_layer_full(family=\"gemma_2b\", wbits=4)` under `rocprofv3 --kernel-trace --stats`: one Gemma-2B-shaped W4A8 layer at S = 2048",
    "decode_engine_trace.summary.txt": "`LAYERS=22 PREFETCH=0.5 rocprofv3 --kernel-trace --stats -- python tools/prof_decode_engine.py`: eager steps of the TinyLlama-shaped decode engine, context 256",
    "attention_pmc_sq.summary.txt": "`rocprofv3 --pmc SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_WAIT_INST_ANY SQ_VALU_MFMA_BUSY_CYCLES --kernel-trace -- python tools/prof_attention.py` (S = 2048, 32 / 4 heads, head_dim 64)",
}
LOG_CAPTIONS = {
    "decode_stamps_before.log": "`tools/decode_stamps.py` (`-DMQ_DECODE_STAMPS` build): per-launch gap / ramp / in-kernel `s_memrealtime` stamps of the decode step at the START of round 3",
    "decode_stamps_roles_fastdiv.log": "the same after the wave-role split and `div_by_scale`",
    "latency_probe.log": "`tools/latency_probe.cpp`: kernel boundary, kernarg and first-load latencies inside a hipGraph",
    "div_check.log": "`tools/div_check.cpp`: `div_by_scale` against the IEEE divide, every fp32 dividend, for the divisors the log lists (round 3: 16; round 4: 48 in [1e-5, 1e6] + 54 over +-[2^-60, 2^60])",
    "div_check_wide.log": "the same, 32 more divisors (scale clamps, all-ones significands)",
    "fr128_ab.log": "`tools/bench_fr128.py`: the 128-column generated GEMMs against the C++ tile kernels they replace",
    "decode_context_sweep.log": "`tools/decode_context_sweep.py`: ms per token against cached positions for 1 / 2 / 4 / 8 attention workgroups per head and the by-position default",
    # round 6 (tools/evidence_r06.sh, tools/r06_decode_ab.py, tools/r06_whatif.sh)
    "atomic_probe.log": "`tools/atomic_probe.cpp`: price of 65 536 no-return device-scope int32 atomics (32 adders per address) between two launches -- the split-K hand-off of o_proj inside the attention launch",
    "decode_base.log": "the five-launch decode engine at the START of round 6 (tok/s at 256 / 1 024 / 2 048 cached positions)",
    "decode_4launch_vs_5launch.log": "`bench.bench_decode_full(launches=4 | 5)` on one box: the four-launch chain against the five-launch chain",
    "decode_stamps_L4.log": "`LAUNCHES=4 tools/decode_stamps.py` (`-DMQ_DECODE_STAMPS` build, 6 layers, context 256): per-launch gap / ramp / in-kernel stamps of the four-launch decode step",
    "decode_stamps_L5.log": "the same for the five-launch chain (`LAUNCHES=5`) on the same box",
    "decode_prefetch_ab.log": "`tools/r06_decode_ab.py`: share and start time of the attention launch's L2 prefetch rows",
    "decode_whatif_kv_loads.log": "`tools/r06_whatif.sh`: what-if builds of the attention + o_proj launch without its key / value requests",
    "bench_qmatmul.log": "`tools/bench_qmatmul.py`: mq_qmatmul step by step through round 6 (rows per workgroup, what-if builds, the row-panel kernel, branch-free loads, buffer stores, loop orders rotated against HBM channel camping); `tools/hbm_read_probe` ceilings",
    "fuzz_full.log": "the random-shape fuzzers at full length on the final tree (`tests/fuzz_*.py`, `tests/stress_qmatmul_race.py`): every output against the oracles, 0 mismatches",
    "bench_qmatmul_final.log": "`tools/bench_qmatmul.py` and `tools/bench_calib_probs.py` on the final tree",
    "calibration_full_size_check.log": "`tools/r06_calib_full_size_check.py`: the act_dict with every round-6 calibration change on against the plain hooks, full-width 2-layer graphs of three families: worst relative deviation 1.8e-6 ... 2.6e-6",
    "calibration512_stub_gemm.json": "`python bench.py --workload calibration --calib-samples 512 --calib-stub-gemm`: BASELINE.json configs[4] at its full size on one GPU, final tree (180 samples/s; round 5: 81)",
    "calibration64.json": "`python bench.py --workload calibration --calib-samples 64`: the same graph with the fp32 library GEMMs in (22.5 samples/s)",
    "calibration64_per_channel_stub_gemm.json": "`... --calib-samples 64 --calib-stub-gemm --per-channel`: per-channel statistics (38 samples/s: the one-pass layer kernels are per-tensor only)",
    "calibration_trace_after.summary.txt": "the same trace on the final tree (alias groups, `mq_calib_norm` / `mq_calib_gated` / `mq_calib_rope`): 333 -> 195 ms of kernels for the same passes",
    "calibration_trace.summary.txt": "`rocprofv3 --kernel-trace` of `bench.py --workload calibration --calib-samples 16 --calib-stub-gemm`: kernels of the calibration pass by share of GPU time, BEFORE the round-6 calibration changes",
    "fuzz_and_per_sequence_ppl.log": "`pytest tests/test_gpu_fuzz.py tests/test_gpu_round5.py -k 'fuzz or perplexity' -s`: the fuzzer slices inside `-m gpu` and the per-sequence perplexity differences",
    "bench_wall.log": "wall time of `python bench.py --steps 20 --warmup 5` (round 5: 173 s)",
    # round 4 (everything below is rewritten by tools/evidence_r04.sh)
    "boundary_probe.log": "`tools/hole_probe.py` on the stamped build (`tools/build_stamped.sh`): kernel boundary behind the headline GEMM from in-kernel `s_memrealtime` stamps, per-XCD anatomy and clock, K sweep",
    "grid_barrier_probe.log": "`tools/barrier_probe.cpp 2000`: software grid barriers among 256 resident workgroups -- single counter, round 2's hierarchy, and the microarchitecture guide's XCD-hierarchical recipe",
    "decode_stamps_w8.log": "`tools/decode_stamps.py` (`-DMQ_DECODE_STAMPS` build, 6 layers, context 256): per-launch gap / ramp / in-kernel stamps of the decode step, int8 weights",
    "decode_stamps_w4.log": "the same with packed 4-bit weights (`WBITS=4`)",
    "groupm_traffic.log": "`tools/groupm_probe.py` + `rocprofv3 --pmc FETCH_SIZE` per setting: tile order of the N = 2048 residual GEMMs (`mq_gemm_set_group_m`) against time and L2 fetch bytes",
    "train_step_kernels_before.log": "`tools/train_prof.py` (torch.profiler, device time per kernel over 6 e2equant inner steps + set-up) at the START of round 4",
    "train_step_kernels.log": "the same at the end of the round (fused LWC pass, training attention-probabilities pass, vectorised STE backward)",
    "bench_w4.log": "`tools/bench_w4.py`: packed-W4 generated-ISA GEMMs (expanded per workgroup / per-wave unpack) against the int8-image kernel, identical indices",
    "gpu_tests.log": "result line of `python -m pytest tests -m gpu -q` on the box that produced this directory",
    "bench_final.json": "un-profiled `python bench.py --steps 20 --warmup 5` (the driver's settings) at the end of the round",
    "bench_steps200.json": "`python bench.py --steps 200 --warmup 20 --headline-only --no-cpu-baseline` on the same box, minutes later (agreement of the step time with the 20-step run)",
    # round 5 (tools/evidence_r05.sh, tools/r05_probes.sh)
    "mfma_energy_probe.log": "`tools/mfma_energy_probe.cpp 256 200`: NOTHING but int8 MFMAs on all 256 CUs -- 16x16x64 against 32x32x32, operands in registers or W re-read from the LDS every k-step, 8 waves x 32 rows against 4 waves x 64 rows, quantised-Gaussian / uniform-random / zero operands: us per launch, TOPS, the clock each variant holds",
    "mfma_energy_probe_k32.log": "the same at 32 k-steps per wave (= the headline K = 2048): the difference to the 256-k-step run gives the steady-state time per k-step without launch overhead",
    "whatif_operand_paths.log": "`tools/build_stamped.sh <tag> MQ_FR_NO_READ=1 | MQ_FR_NO_W=1 | MQ_FR_NO_A=1` + `tools/ab_stamped.sh frs frs_nord frs_now frs_noa`: the production GEMM program with one operand path removed (wrong results, same schedule), alternated on one box -- cycles per wave, clock, launch period",
    "bench_w4_lds.log": "`tools/bench_w4.py` with the generator's `MQ_FR_W4X_LDS=1` experiment (packed pieces parked in the LDS by LDS-DMA three stages ahead and read back at expansion time; two scheduling variants) against the int8-image kernel: slower than the register path of round 4 -- not built by default",
    "stable_depth_diag.log": "`tools/stable_depth_diag.py`: the contractive 22-layer model of `full_depth_stable_case.npz` over 8 sequences (2 040 predicted tokens) -- perplexity difference to the reference per execution path (the reference's op sequence on rocBLAS, integer linears, all-integer module chain, fused prefill), per sequence",
    "stable_depth_diag_one_sequence.log": "the same diagnostic on the first (one-sequence, 255-position) form of the fixture: every path, incl. the reference's own op sequence on rocBLAS, 0.03-0.075 off for W4A8 -- position-correlated summation-order noise, the reason the fixture grew to eight sequences",
    "stable_ppl.log": "the report line of `test_quantized_perplexity_within_0_05_of_the_reference_at_22_layers` (module chain, fused prefill, decode engine; W8A8 and W4A8)",
    "bench_qmatmul.log": "`tools/bench_qmatmul.py`: `mq_qmatmul` against the simulated QMatMul path (HIP fake-quant kernels around the fp32 library bmm) at the attention block's shapes",
    "calibration64.json": "`python bench.py --workload calibration --calib-samples 64`: configs[4]'s per-sample cost with the fp32 library GEMMs in (the score chain of every attention block takes its two statistics in one pass)",
    "calibration64_stub_gemm.json": "the same with `--calib-stub-gemm` (the timed region is the hot path: hooked reductions + the fused score chain)",
    "decode_stamps_w8.log": "`tools/decode_stamps.py` (`-DMQ_DECODE_STAMPS` build, 6 layers, context 256): per-launch gap / ramp / in-kernel stamps of the decode step, int8 weights",
    "attention_f16_ab.log": "`tools/att_f16_ab.py` (and with `MQ_ATT_ROT=16 MQ_ATT_KV=32`, the StableLM shape): the prefill attention at head_dim 64 with its scores contracted as int8 MFMA + zero-point terms against fp16 MFMA over the centred indices (`mq_attention_set_f16`): identical fp32 output / int8 image / row sums, and the time of each (prep + core, one hipGraph)",
    "attention_stamps.log": "`tools/att_stamps.py` on the `-DMQ_ATT_STAMPS` build: `s_memtime` differences accumulated per wave of the f16 attention kernel -- q preparation, the waits / fragment + DMA issue / chain + MFMAs of sweep 1, waits and the rest of sweep 2, epilogue (the stamps themselves cost ~10 %)",
    "valu_rate_probe.log": "`tools/valu_rate_probe.cpp`: cycles per wave-instruction of the chain's VALU instructions (fma / pk_fma / exp / med3 / add / pk_add and the chain's mix), one and two waves per SIMD",
    "fuzz_round3.log": "`python tests/fuzz_round3.py 120` on the final tree: random shapes / grids -- generated 128-column GEMMs == the C++ tiles, `mq_attention_quant` at head_dim 64 / 128 / 256 against the oracle, chunked == single shot, in-kernel q rows == the prep kernel's image, fp16 score contraction == the int8 one, staged image kernels == the generic ones",
    "attention_whatif.log": "`tools/att_ablate.sh` over `MQ_ATT_ABL` builds (wrong results, same box): the attention op without sweep 2 / sweep 1 / v_exp / barrier / K fragment reads / DMA requests, and with plain loads in place of the DMA pieces",
}
for _v in ("0", "1"):
    for _f, _c in (("sq", "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_VALU_MFMA_BUSY_CYCLES SQ_ACTIVE_INST_ANY"),
                   ("lds", "SQ_INSTS_LDS SQ_ACTIVE_INST_LDS SQ_WAIT_INST_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_VMEM_RD SQ_INST_CYCLES_VMEM_RD SQ_ACTIVE_INST_MISC"),
                   ("misc", "SQ_INSTS_SALU SQ_INSTS_SMEM SQ_ACTIVE_INST_SCA SQ_INST_LEVEL_LDS SQ_INST_LEVEL_VMEM GRBM_GUI_ACTIVE")):
        CAPTIONS[f"attention_pmc_{_f}_f{_v}.summary.txt"] = (f"`tools/prof_attention_pmc.sh`: `rocprofv3 --pmc {_c} --kernel-trace -- python tools/prof_attention.py` with "
                                                            f"`MQ_ATT_F16={_v}` ({'fp16 over the centred indices' if _v == '1' else 'the int8 score contraction'}); the attention core kernel's rows")


def kernel_table(path, only=None, limit=60):
    """lines ' name   n=.. mean X us min Y us' of a --stats summary -> markdown rows"""
    rows = []
    for line in open(path):
        m = re.match(r"\s*(.+?)\s+n=(\d+) mean ([\d.]+) us min ([\d.]+) us", line)
        if not m:
            continue
        name = m.group(1)
        if only and not re.search(only, name):
            continue
        name = re.sub(r"\(.*", "", name).replace("void ", "").strip()
        rows.append((name, int(m.group(2)), float(m.group(3)), float(m.group(4))))
    out = ["| kernel | launches | mean µs | min µs |", "|---|---|---|---|"]
    for name, n, mean, mn in rows[:limit]:
        out.append(f"| `{name}` | {n} | {mean:.2f} | {mn:.2f} |")
    return out if rows else []


def counters(path, kernel_re):
    """{counter: (mean value, n, mean duration us)} of the first kernel block matching kernel_re in a pmc summary"""
    txt = open(path).read()
    res, on = {}, False
    for line in txt.splitlines():
        if line.startswith(" kernel:"):
            if on:
                break
            on = re.search(kernel_re, line) is not None
            continue
        if on:
            m = re.match(r"\s+(\S+)\s+([\d.]+)\s+\(n=(\d+), mean dur ([\d.]+) us\)", line)
            if m:
                res[m.group(1)] = (float(m.group(2)), int(m.group(3)), float(m.group(4)))
    return res


def main(d):
    files = sorted(os.listdir(d))
    L = [f"# {d.rstrip('/')} -- measurements on MI355X (gfx950, ROCm 7.2)", "",
         "GENERATED by `python tools/profiles_readme.py " + d.rstrip("/") + "` from the files in this directory -- do not edit; re-run after adding a capture.",
         "`*.summary.txt` = per-kernel mean duration (and mean counter value) over all dispatches of one rocprofv3 pass (`tools/pmc_summary.py` over the",
         "rocpd database); `*.bench.json` = the bench line printed inside that pass (profiled runs are slower than un-profiled ones).", ""]

    L += ["## Files", "", "| file | what |", "|---|---|"]
    for f in files:
        if f == "README.md":
            continue
        cap = CAPTIONS.get(f) or LOG_CAPTIONS.get(f)
        if cap is None and f.endswith(".bench.json"):
            cap = "bench line printed inside the pass of the same name"
        if cap is None and f.startswith("calibration_512"):
            cap = "`python bench.py --workload calibration` (BASELINE.json configs[4]: 512 samples x S = 2048, 22 layers, one GPU)" + \
                  (" `--calib-stub-gemm`" if "stub" in f else "") + (" `--per-channel`" if "per_channel" in f else "")
        if cap is None and f.startswith("bench_"):
            cap = "un-profiled `python bench.py` line of that point of the round"
        L.append(f"| `{f}` | {cap or ''} |")
    L.append("")

    # ---- dominant kernel ---------------------------------------------------------------------------------------------------------
    sq = os.path.join(d, "pmc_sq.summary.txt")
    if os.path.exists(sq):
        K = r"mq::gemm_i8_fr_kernel"
        c = counters(sq, K)
        lds = counters(os.path.join(d, "pmc_lds.summary.txt"), K) if os.path.exists(os.path.join(d, "pmc_lds.summary.txt")) else {}
        fe = counters(os.path.join(d, "pmc_fetch.summary.txt"), K) if os.path.exists(os.path.join(d, "pmc_fetch.summary.txt")) else {}
        wr = counters(os.path.join(d, "pmc_write.summary.txt"), K) if os.path.exists(os.path.join(d, "pmc_write.summary.txt")) else {}
        L += ["## Dominant kernel `mq::gemm_i8_fr_kernel` (M = 2048, N = 5632, K = 2048; generated gfx950 ISA)", "",
              "| quantity | value | from |", "|---|---|---|"]
        tr = os.path.join(d, "trace.summary.txt")
        if os.path.exists(tr):
            for line in open(tr):
                m = re.match(r"\s*mq::gemm_i8_fr_kernel\(.*n=(\d+) mean ([\d.]+) us min ([\d.]+) us", line)
                if m:
                    L.append(f"| duration in the `--stats` pass | mean {float(m.group(2)):.2f} µs, min {float(m.group(3)):.2f} µs, n = {m.group(1)} | trace.summary.txt |")
        if c:
            dur = c["SQ_WAVES"][2]
            waves = c["SQ_WAVES"][0]
            L.append(f"| duration in the SQ counter pass | {dur:.2f} µs (n = {c['SQ_WAVES'][1]}) | pmc_sq.summary.txt |")
            mfma = c["SQ_VALU_MFMA_BUSY_CYCLES"][0]
            L.append(f"| MFMA busy | `SQ_VALU_MFMA_BUSY_CYCLES` {mfma:,.0f} = {mfma / 1024:,.0f} cycles per SIMD (1 024 SIMDs) | pmc_sq |")
            wc = c["SQ_WAVE_CYCLES"][0] * 4.0 / waves
            L.append(f"| wave lifetime | `SQ_WAVE_CYCLES` {c['SQ_WAVE_CYCLES'][0]:,.0f} quad-cycles / {waves:.0f} waves = {wc:,.0f} cycles per wave -> the SIMD's matrix pipe is busy "
                     f"{100.0 * (mfma / 1024) / wc:.0f} % of its two (concurrent) waves' lifetime | pmc_sq |")
            L.append(f"| waits | `SQ_WAIT_ANY` {100 * c['SQ_WAIT_ANY'][0] / c['SQ_WAVE_CYCLES'][0]:.0f} %, `SQ_WAIT_INST_ANY` "
                     f"{100 * c['SQ_WAIT_INST_ANY'][0] / c['SQ_WAVE_CYCLES'][0]:.0f} % of wave cycles | pmc_sq |")
            L.append(f"| int8 MFMA ops | `SQ_INSTS_VALU_MFMA_MOPS_I8` {c['SQ_INSTS_VALU_MFMA_MOPS_I8'][0]:,.0f} x 512 = "
                     f"{c['SQ_INSTS_VALU_MFMA_MOPS_I8'][0] * 512 / 1e9:.2f} GOP (algorithmic: 47.24) | pmc_sq |")
        if lds:
            L.append(f"| LDS | `SQ_INSTS_LDS` {lds['SQ_INSTS_LDS'][0]:,.0f}, `SQ_LDS_IDX_ACTIVE` {lds['SQ_LDS_IDX_ACTIVE'][0]:,.0f}, bank-conflict cycles "
                     f"{lds['SQ_LDS_BANK_CONFLICT'][0]:,.0f} ({100 * lds['SQ_LDS_BANK_CONFLICT'][0] / max(lds['SQ_LDS_IDX_ACTIVE'][0], 1):.1f} %) | pmc_lds |")
        if fe:
            f_kb = fe["FETCH_SIZE"][0]
            L.append(f"| fabric read | `FETCH_SIZE` {f_kb:,.0f} KiB x 2 (gfx950 correction) = {2 * f_kb * 1024 / 1e6:.1f} MB per launch (algorithmic input 15.7 MB) | pmc_fetch |")
        if wr:
            w_kb = wr["WRITE_SIZE"][0]
            L.append(f"| HBM write | `WRITE_SIZE` {w_kb:,.0f} KiB = {w_kb * 1024 / 1e6:.2f} MB (M x N bytes = 11.53 MB) | pmc_write |")
            if "TCC_HIT_sum" in wr:
                h, m_ = wr["TCC_HIT_sum"][0], wr["TCC_MISS_sum"][0]
                L.append(f"| L2 | hits {h:,.0f}, misses {m_:,.0f}: hit rate {100 * h / (h + m_):.0f} % | pmc_write |")
        if fe and wr:
            L.append(f"| `roofline.traffic` of bench.py | 2 x FETCH + WRITE = {(2 * fe['FETCH_SIZE'][0] + wr['WRITE_SIZE'][0]) * 1024:,.0f} bytes per launch "
                     "(labelled \"last profiled\" in the bench line) | pmc_fetch + pmc_write |")
        L.append("")

    for f, title, only in (("layer_trace.summary.txt", "Fused TinyLlama W8A8 layer, S = 2048 (the nine launches of DESIGN.md 4.6 are the rows with n = 22 / 44 / 54)", r"mq::"),
                           ("gemma_layer_trace.summary.txt", "Gemma-2B-shaped W4A8 layer, S = 2048", r"mq::"),
                           ("decode_engine_trace.summary.txt", "Decode engine, eager steps (TinyLlama shape, 22 layers, context 256)", r"mq::decode|indexSelect"),
                           ("trace.summary.txt", "`mq::` kernels over the whole bench run", r"mq::")):
        p = os.path.join(d, f)
        if os.path.exists(p):
            t = kernel_table(p, only)
            if t:
                L += [f"## {title}", "", f"from `{f}`", ""] + t + [""]

    # ---- per-kernel counters of the fused layer (three separate PMC passes over tools/prof_layer.py) -------------------------------------
    lf, lw, lq = (os.path.join(d, f"layer_pmc_{k}.summary.txt") for k in ("fetch", "write", "sq"))
    if all(os.path.exists(x) for x in (lf, lw, lq)):
        names = []
        for line in open(lf):
            m = re.match(r" kernel: (.*)", line)
            if m and "mq::" in m.group(1):
                names.append(m.group(1).strip())
        L += ["## Fused TinyLlama layer, per-kernel counters", "",
              "from `layer_pmc_fetch / _write / _sq.summary.txt` (`tools/r03_cc.sh`: three rocprofv3 `--pmc` passes over `tools/prof_layer.py`, kernel trace only).",
              "read = 2 x FETCH_SIZE (gfx950 correction), write = WRITE_SIZE; rate = (read + write) / mean duration of the fetch pass;",
              "MFMA busy = SQ_VALU_MFMA_BUSY_CYCLES / 1 024 SIMDs; VALU instructions per wave = SQ_INSTS_VALU / SQ_WAVES.  (fr128<2> averages o_proj and w2.)", "",
              "| kernel | launches | µs | read MB | write MB | TB/s | L2 hit % | MFMA busy cycles / SIMD | VALU instructions / wave |", "|---|---|---|---|---|---|---|---|---|"]
        for k in names:
            kre = re.escape(k[:60])
            f_, w_, q_ = counters(lf, kre), counters(lw, kre), counters(lq, kre)
            if not f_ or not w_:
                continue
            n, dur = f_["FETCH_SIZE"][1], f_["FETCH_SIZE"][2]
            if n < 20 or "minmax" in k:
                continue                      # one-off calibration / set-up kernels of the script
            rd, wr_ = 2 * f_["FETCH_SIZE"][0] * 1024 / 1e6, w_["WRITE_SIZE"][0] * 1024 / 1e6
            hit = 100 * w_["TCC_HIT_sum"][0] / max(w_["TCC_HIT_sum"][0] + w_["TCC_MISS_sum"][0], 1)
            mf = q_.get("SQ_VALU_MFMA_BUSY_CYCLES", (0,))[0] / 1024
            vi = q_.get("SQ_INSTS_VALU", (0,))[0] / max(q_.get("SQ_WAVES", (1,))[0], 1)
            short = re.sub(r"\(.*", "", k).replace("void ", "")
            L.append(f"| `{short}` | {n} | {dur:.2f} | {rd:.1f} | {wr_:.1f} | {(rd + wr_) / dur:.2f} | {hit:.0f} | {mf:,.0f} | {vi:,.0f} |")
        L.append("")

    p = os.path.join(d, "attention_pmc_sq.summary.txt")
    if os.path.exists(p):
        c = counters(p, r"attention_quant_kernel")
        if c:
            L += ["## Prefill attention core (head_dim 64), S = 2048, 32 / 4 heads", "", "from `attention_pmc_sq.summary.txt`", "",
                  "| counter | mean per launch |", "|---|---|"]
            for k in sorted(c):
                L.append(f"| `{k}` | {c[k][0]:,.0f} |")
            dur = next(iter(c.values()))[2]
            L.append(f"| duration in this pass | {dur:.2f} µs |")
            if "SQ_INSTS_VALU" in c and "SQ_ACTIVE_INST_VALU" in c and "SQ_BUSY_CYCLES" in c:
                L.append(f"| derived | {c['SQ_INSTS_VALU'][0] / 67584:,.0f} VALU instructions per 16 x 64 block and wave (67 584 block-waves, both sweeps) |")
            L.append("")

    benches = [f for f in files if f.endswith(".json") and not f.endswith(".bench.json")]
    if benches:
        L += ["## Bench lines kept in this directory", "", "| file | metric | value | ms per step | roofline frac |", "|---|---|---|---|---|"]
        for f in benches:
            try:
                j = json.load(open(os.path.join(d, f)))
            except Exception:
                continue
            L.append(f"| `{f}` | {j.get('metric', '')[:70]} | {j.get('value')} {j.get('unit', '')} | {j.get('ms_per_step')} | {(j.get('roofline') or {}).get('frac')} |")
        L.append("")
    open(os.path.join(d, "README.md"), "w").write("\n".join(L) + "\n")
    print("wrote", os.path.join(d, "README.md"), len(L), "lines")


if __name__ == "__main__":
    main(sys.argv[1] if len(sys.argv) > 1 else os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "profiles", "r03"))

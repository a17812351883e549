#!/bin/bash
# Regenerates the round-5 evidence under gpurun_out/evidence_r05/ on an MI355X (run through gpurun; ~20 min).  The files kept under
# profiles/r05/ are copies of what this and tools/r05_probes.sh write (captions: tools/profiles_readme.py).   usage: evidence_r05.sh [quick]
R=$GRAFT_REPO_ROOT; OUT=$R/gpurun_out/evidence_r05; mkdir -p $OUT; cd $R
# 1. parity: the whole GPU suite, then the smoke entry
timeout 1800 python -m pytest tests -m gpu -q > $OUT/gpu_tests_full.log 2>&1; grep -E "passed|failed" $OUT/gpu_tests_full.log > $OUT/gpu_tests.log; cat $OUT/gpu_tests.log
timeout 300 python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" > $OUT/smoke.log 2>&1; tail -1 $OUT/smoke.log
# 2. the bench line at the driver's settings and at 10 x the steps
timeout 1500 python bench.py --steps 20 --warmup 5 2> $OUT/bench_final.err | grep '^{"metric"' > $OUT/bench_final.json
timeout 600 python bench.py --steps 200 --warmup 20 --headline-only --no-cpu-baseline 2>/dev/null | grep '^{"metric"' > $OUT/bench_steps200.json
python - <<'PY'
import json, os
o = os.path.join(os.environ["GRAFT_REPO_ROOT"], "gpurun_out", "evidence_r05")
for f in ("bench_final.json", "bench_steps200.json"):
    try:
        d = json.load(open(os.path.join(o, f)))
        r = d["roofline"]
        print(f, "value", d["value"], "ms/step", d["ms_per_step"], "host-clock ms/step", d["timing"].get("host_clock_ms_per_step"), "gemm us", r["avg_launch_us"], "frac", r["frac"],
              "MHz", r.get("sustained_mhz"), "ceiling", r.get("ceiling_at_sustained_clock"), "adj", r.get("frac_clock_adjusted"))
        if d.get("decode"):
            print("  decode", d["decode"].get("decode_tok_s"), d["decode"].get("decode_tok_s_by_context"), "cpu", (d.get("cpu_baseline") or {}).get("value"),
                  (d.get("cpu_baseline") or {}).get("thread_sweep_seconds"))
            v = d["variants"]
            print("  calib stub", v.get("calibration_512_stub_gemm", {}).get("samples_per_s"), "pair", v.get("ffn_pair_gemm", {}).get("frac_of_int8_peak"),
                  "layer", v.get("layer_prefill_full", {}).get("fused_us"), "multi_gpu", d.get("multi_gpu", {}).get("all_ranks_agree"))
    except Exception as e:
        print(f, "unreadable:", e)
PY
[ "$1" = quick ] && exit 0
# 3. rocprofv3: kernel trace of every bench leg + the four PMC passes of the headline legs (tools/prof_bench.sh)
bash tools/prof_bench.sh r05 pmc > $OUT/prof_bench.log 2>&1; cp gpurun_out/prof_bench_r05/*.summary.txt gpurun_out/prof_bench_r05/*.bench.json $OUT/ 2>/dev/null
head -30 $OUT/trace.summary.txt
# 4. decode timeline (stamped build)
python -c "from mobilequant_amd import build; build.build(force=True, tag='stamps', extra_flags=['-DMQ_DECODE_STAMPS'])" > /dev/null 2>&1
MQ_LIB_PATH=mobilequant_amd/lib/stamps/libmobilequant_amd.so LAYERS=6 CONTEXT=256 WBITS=8 timeout 300 python tools/decode_stamps.py > $OUT/decode_stamps_w8.log 2>&1
ls -la $OUT
# 5. the prefill attention (round 5: f16 score contraction): A/B against the int8 form, counter passes, in-kernel stamps, VALU issue rates
bash tools/build_r04_attention.sh > /dev/null 2>&1   # (a no-op on the GPU box: build it before gpurun)
for shape in "MQ_ATT_ROT=64 MQ_ATT_KV=4" "MQ_ATT_ROT=16 MQ_ATT_KV=32"; do
  echo "== the round-4 kernel (lib/r04att: both columns time it)" >> $OUT/attention_f16_ab.log
  env $shape MQ_LIB_PATH=mobilequant_amd/lib/r04att/libmobilequant_amd.so python tools/att_f16_ab.py 2>&1 | grep -v amdgpu.ids | head -1 >> $OUT/attention_f16_ab.log
  echo "== this tree" >> $OUT/attention_f16_ab.log
  env $shape python tools/att_f16_ab.py 2>&1 | grep -v amdgpu.ids >> $OUT/attention_f16_ab.log
done; cat $OUT/attention_f16_ab.log
bash tools/prof_attention_pmc.sh att_r05 > /dev/null 2>&1
for f in sq lds misc; do for v in 0 1; do cp gpurun_out/prof_att_r05/${f}_f$v.summary.txt $OUT/attention_pmc_${f}_f$v.summary.txt; done; done
python -c "from mobilequant_amd import build; build.build(force=True, tag='attst', extra_flags=['-DMQ_ATT_STAMPS'], only=['mq_attention.hip'])" > /dev/null 2>&1
MQ_LIB_PATH=mobilequant_amd/lib/attst/libmobilequant_amd.so timeout 300 python tools/att_stamps.py 2>&1 | grep -v amdgpu.ids > $OUT/attention_stamps.log; cat $OUT/attention_stamps.log
hipcc --offload-arch=gfx950 -O3 tools/valu_rate_probe.cpp -o /tmp/valu_rate_probe 2> /dev/null && timeout 120 /tmp/valu_rate_probe > $OUT/valu_rate_probe.log 2>&1; cat $OUT/valu_rate_probe.log
bash tools/prof_cmd.sh layer_r05 -- python $R/tools/prof_layer.py > /dev/null 2>&1; grep "mq::" gpurun_out/prof_layer_r05/trace.summary.txt | cut -c1-170 > $OUT/layer_trace.summary.txt; cat $OUT/layer_trace.summary.txt
ls -la $OUT

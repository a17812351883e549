#!/bin/bash
# Regenerates the round-6 evidence under gpurun_out/evidence_r06/ on an MI355X (run through gpurun; ~12 min).  profiles/r06/ keeps copies.
R=$GRAFT_REPO_ROOT; OUT=$R/gpurun_out/evidence_r06; mkdir -p $OUT; cd $R
# 1. the bench line at the driver's settings (headline + decode + configs[2] / [3] + cpu baseline; the other legs: bench_variants.py)
T0=$(date +%s); timeout 1200 python bench.py --steps 20 --warmup 5 2> $OUT/bench_final.err | grep '^{"metric"' > $OUT/bench_final.json; echo "bench.py wall $(( $(date +%s) - T0 )) s" | tee $OUT/bench_wall.log
python - <<'PY'
import json, os
o = os.path.join(os.environ["GRAFT_REPO_ROOT"], "gpurun_out", "evidence_r06")
d = json.load(open(os.path.join(o, "bench_final.json")))
r = d["roofline"]
print("value", d["value"], "ms/step", d["ms_per_step"], "gemm us", r["avg_launch_us"], "frac", r["frac"], "MHz", r.get("sustained_mhz"))
print("decode", d["decode"]["decode_tok_s"], d["decode"]["decode_tok_s_by_context"], "w4", d["decode"]["full_step_w4a8"]["decode_tok_s"],
      "stablelm", d["decode"]["stablelm_2_1_6b_w8a8_per_channel"]["decode_tok_s"], "gemma", d["decode"]["gemma_2b_w4a8_symmetric"]["decode_tok_s"])
print("cpu", d["cpu_baseline"]["value"], d["cpu_baseline"].get("value_best"), "variants", list(d["variants"]))
PY
[ "$1" = quick ] && exit 0
# 2. rocprofv3: kernel trace of the bench + the four PMC passes of the headline legs
bash tools/prof_bench.sh r06 pmc > $OUT/prof_bench.log 2>&1; cp gpurun_out/prof_bench_r06/*.summary.txt gpurun_out/prof_bench_r06/*.bench.json $OUT/ 2>/dev/null
head -25 $OUT/trace.summary.txt | cut -c1-200
# 3. the decode engine under the tracer (4 launches per layer) and its stamped timeline
LAYERS=22 bash tools/prof_cmd.sh decode_r06 -- python $R/tools/prof_decode_engine.py > /dev/null 2>&1; grep "mq::\|Kernel\|index" gpurun_out/prof_decode_r06/trace.summary.txt | cut -c1-200 > $OUT/decode_engine_trace.summary.txt; cat $OUT/decode_engine_trace.summary.txt
python -c "from mobilequant_amd import build; build.build(force=True, tag='stamps', only=['mq_decode.hip'], extra_flags=['-DMQ_DECODE_STAMPS'])" > /dev/null 2>&1
for L in 4 5; do MQ_LIB_PATH=mobilequant_amd/lib/stamps/libmobilequant_amd.so LAUNCHES=$L LAYERS=6 CONTEXT=256 WBITS=8 timeout 300 python tools/decode_stamps.py 2>&1 | grep -v "amdgpu.ids\|RuntimeWarning\|nanmean" | cut -c1-260 > $OUT/decode_stamps_L$L.log; done
cat $OUT/decode_stamps_L4.log
ls -la $OUT

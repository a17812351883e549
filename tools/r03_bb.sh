#!/bin/bash
O=gpurun_out/r03bb; mkdir -p $O
timeout 900 python bench.py > $O/bench.json 2> $O/bench.err; echo "rc=$?" >> $O/bench.err; tail -3 $O/bench.err
python - <<PY
import json
d=json.load(open("$O/bench.json"))
v=d["variants"]
print(d["value"], d["ms_per_step"], d["roofline"]["avg_launch_us"], d["roofline"]["frac"], d["roofline"]["quantize_kernel"]["avg_launch_us"])
print("decode", d["decode"]["decode_tok_s"], d["decode"]["full_step_w4a8"]["decode_tok_s"])
for k in ("layer_prefill_full","layer_prefill_full_w4a8","layer_prefill_full_stablelm_2_1_6b","layer_prefill_full_gemma_2b_w4a8"):
    print(k, v[k]["fused_us"], v[k].get("attention_op_us"))
print("model_prefill", v["model_prefill"]["fused_ms"], "pair", v["ffn_pair_gemm"]["avg_launch_us"], v["ffn_pair_gemm"]["frac_of_int8_peak"])
print("train", v["train_step_e2equant"]); print("calib512", v["calibration_512_stub_gemm"]["samples_per_s"])
PY

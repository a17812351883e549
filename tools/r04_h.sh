#!/bin/bash
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r04h; mkdir -p $O; cd $R
for t in r3 new; do
  p=$R/mobilequant_amd/lib/$t/libmobilequant_amd.so; [ $t = new ] && p=$R/mobilequant_amd/lib/libmobilequant_amd.so
  MQ_LIB_PATH=$p timeout 900 python bench.py --steps 20 --warmup 5 --no-cpu-baseline 2>/dev/null | grep '^{"metric"' > $O/bench_$t.json
  python - <<PY
import json
d=json.load(open("$O/bench_$t.json")); r=d["roofline"]; v=d.get("variants",{})
print("$t", "value", d["value"], "ms/step", d["ms_per_step"], "gemm_us", r.get("avg_launch_us"), "frac", r["frac"])
for k in ("ffn_pair_gemm","layer_prefill","layer_prefill_full","layer_prefill_full_w4a8","model_prefill"):
    if k in v: print("   ", k, json.dumps(v[k])[:300])
PY
done

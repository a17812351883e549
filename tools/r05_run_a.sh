#!/bin/bash
# round 5, GPU call A: MFMA energy probe, the new round-5 tests, the whole GPU suite, a baseline bench line of this box
R=$GRAFT_REPO_ROOT; OUT=$R/gpurun_out/r05a; mkdir -p $OUT; cd $R
hipcc --offload-arch=gfx950 -O3 tools/mfma_energy_probe.cpp -o /tmp/mfma_energy_probe 2> $OUT/probe_build.log && timeout 300 /tmp/mfma_energy_probe 256 200 > $OUT/mfma_energy_probe.log 2>&1
timeout 120 /tmp/mfma_energy_probe 32 400 > $OUT/mfma_energy_probe_k32.log 2>&1
cat $OUT/mfma_energy_probe.log | tail -12
timeout 1200 python -m pytest tests/test_gpu_round5.py -q -x -s > $OUT/gpu_tests_r5.log 2>&1; tail -25 $OUT/gpu_tests_r5.log
timeout 1500 python -m pytest tests -m gpu -q > $OUT/gpu_tests.log 2>&1; tail -15 $OUT/gpu_tests.log
timeout 900 python bench.py --steps 20 --warmup 5 2> $OUT/bench.err | grep '^{"metric"' > $OUT/bench.json
python - <<'PY'
import json, os
d = json.load(open(os.path.join(os.environ["GRAFT_REPO_ROOT"], "gpurun_out", "r05a", "bench.json")))
r = d["roofline"]
print("value", d["value"], "ms/step", d["ms_per_step"], "gemm us", r.get("avg_launch_us"), "frac", r["frac"], "MHz", r.get("sustained_mhz"), "adj", r.get("frac_clock_adjusted"))
v = d.get("variants", {})
for k in ("decode", "ffn_pair_gemm", "layer_prefill_full"):
    print(k, json.dumps(v.get(k))[:600])
PY

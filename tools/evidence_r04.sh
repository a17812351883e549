#!/bin/bash
# Regenerates the round-4 evidence under gpurun_out/evidence_r04/ on an MI355X (run through gpurun; ~25 min).  The files kept under
# profiles/r04/ are copies of what this writes (captions: tools/profiles_readme.py).    usage: evidence_r04.sh [quick]
R=$GRAFT_REPO_ROOT; OUT=$R/gpurun_out/evidence_r04; mkdir -p $OUT; cd $R
# 1. parity: the whole GPU suite, then the smoke entry
timeout 1800 python -m pytest tests -m gpu -q > $OUT/gpu_tests.log 2>&1; tail -3 $OUT/gpu_tests.log
timeout 300 python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" > $OUT/smoke.log 2>&1; tail -1 $OUT/smoke.log
# 2. the bench line at the driver's settings and at 10 x the steps (they must agree within 2 %)
timeout 1500 python bench.py --steps 20 --warmup 5 2> $OUT/bench_final.err | grep '^{"metric"' > $OUT/bench_final.json
timeout 600 python bench.py --steps 200 --warmup 20 --headline-only --no-cpu-baseline 2>/dev/null | grep '^{"metric"' > $OUT/bench_steps200.json
python - <<'PY'
import json, os
o = os.path.join(os.environ["GRAFT_REPO_ROOT"], "gpurun_out", "evidence_r04")
for f in ("bench_final.json", "bench_steps200.json"):
    try:
        d = json.load(open(os.path.join(o, f)))
        r = d["roofline"]
        print(f, "value", d["value"], "ms/step", d["ms_per_step"], "gemm us", r["avg_launch_us"], "frac", r["frac"], "MHz", r.get("sustained_mhz"))
    except Exception as e:
        print(f, "unreadable:", e)
PY
[ "$1" = quick ] && exit 0
# 3. rocprofv3: kernel trace of every bench leg + the four PMC passes of the headline legs (tools/prof_bench.sh)
bash tools/prof_bench.sh r04 pmc > $OUT/prof_bench.log 2>&1; cp gpurun_out/prof_bench_r04/*.summary.txt gpurun_out/prof_bench_r04/*.bench.json $OUT/ 2>/dev/null
# 4. probes: kernel boundary, grid barrier, decode timeline, tile order vs fetch bytes, division sweep, training step
bash tools/build_stamped.sh frs > /dev/null 2>&1
MQ_LIB_PATH=$R/mobilequant_amd/lib/frs/libmobilequant_amd.so timeout 600 python tools/hole_probe.py 2>&1 | grep -v amdgpu.ids > $OUT/boundary_probe.log
hipcc --offload-arch=gfx950 -O2 tools/barrier_probe.cpp -o /tmp/barrier_probe 2>/dev/null && timeout 120 /tmp/barrier_probe 2000 > $OUT/grid_barrier_probe.log 2>&1
python -c "from mobilequant_amd import build; build.build(force=True, tag='stamps', extra_flags=['-DMQ_DECODE_STAMPS'])" > /dev/null 2>&1
for wb in 8 4; do MQ_LIB_PATH=mobilequant_amd/lib/stamps/libmobilequant_amd.so LAYERS=6 CONTEXT=256 WBITS=$wb timeout 300 python tools/decode_stamps.py > $OUT/decode_stamps_w$wb.log 2>&1; done
python tools/groupm_probe.py 2>&1 | grep group_m > $OUT/groupm_traffic.log
( cd /tmp && export TMPDIR=/tmp; for g in 2 4 8 16; do
    timeout 300 rocprofv3 --pmc FETCH_SIZE --kernel-trace -d /tmp/gm_$g -o p -- python $R/tools/groupm_probe.py $g > /dev/null 2>&1
    echo "== rocprofv3 --pmc FETCH_SIZE, group_m $g" >> $OUT/groupm_traffic.log
    python $R/tools/pmc_summary.py /tmp/gm_$g/p_results.db 2>&1 | grep -A1 "fr128" >> $OUT/groupm_traffic.log; rm -rf /tmp/gm_$g; done )
hipcc --offload-arch=gfx950 -O3 -ffp-contract=off -fhip-fp32-correctly-rounded-divide-sqrt tools/div_check.cpp -o /tmp/div_check 2>/dev/null && timeout 900 /tmp/div_check 1 48 > $OUT/div_check.log 2>&1
timeout 600 python tools/train_prof.py > $OUT/train_step_kernels.log 2>&1
timeout 600 python tools/bench_w4.py > $OUT/bench_w4.log 2>&1
ls -la $OUT

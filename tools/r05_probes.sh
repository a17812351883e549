#!/bin/bash
# round 5 probes (run through gpurun, ~5 min): GPU suite, the stable-depth perplexity report and its bias-or-noise diagnostic, the MFMA
# energy probe, the calibration legs at 64 samples -> gpurun_out/r05_probes/ (copies kept in profiles/r05/)
R=$GRAFT_REPO_ROOT; OUT=$R/gpurun_out/r05_probes; mkdir -p $OUT; cd $R
timeout 1500 python -m pytest tests -m gpu -q > $OUT/gpu_tests.log 2>&1; tail -12 $OUT/gpu_tests.log
timeout 900 python -m pytest tests/test_gpu_round5.py -q -s -k "perplexity" 2>&1 | grep "stable full depth" > $OUT/stable_ppl.log; cat $OUT/stable_ppl.log
timeout 900 python tools/stable_depth_diag.py 2>&1 | grep -v amdgpu.ids > $OUT/stable_depth_diag.log; cat $OUT/stable_depth_diag.log
hipcc --offload-arch=gfx950 -O3 tools/mfma_energy_probe.cpp -o /tmp/mfma_energy_probe 2> $OUT/probe_build.log && timeout 300 /tmp/mfma_energy_probe 256 200 > $OUT/mfma_energy_probe.log 2>&1 && timeout 120 /tmp/mfma_energy_probe 32 400 > $OUT/mfma_energy_probe_k32.log 2>&1
grep "4 waves\|W from LDS gauss" $OUT/mfma_energy_probe.log $OUT/mfma_energy_probe_k32.log
for mode in "" "--calib-stub-gemm"; do
  timeout 600 python bench.py --workload calibration --calib-samples 64 $mode 2>/dev/null | grep '^{"metric"' > $OUT/calibration64$mode.json
  python -c "
import json,sys
d=json.load(open('$OUT/calibration64$mode.json')); print('calibration $mode', d['value'], 'samples/s', d['breakdown'])"
done

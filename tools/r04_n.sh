#!/bin/bash
# lease r04n: full GPU suite after the ADVICE r03 low items + the full 2^32-dividend division sweep over the wide scale range
mkdir -p gpurun_out
timeout 1500 python -m pytest tests -m gpu -x -q > gpurun_out/r04n_tests.log 2>&1; echo "tests rc=$?" >> gpurun_out/r04n_tests.log
tail -5 gpurun_out/r04n_tests.log
hipcc --offload-arch=gfx950 -O3 -ffp-contract=off -fhip-fp32-correctly-rounded-divide-sqrt tools/div_check.cpp -o /tmp/div_check 2>/dev/null
timeout 900 /tmp/div_check 1 48 > gpurun_out/div_check.log 2>&1; echo "rc=$?" >> gpurun_out/div_check.log
tail -4 gpurun_out/div_check.log

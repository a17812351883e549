#!/bin/bash
# round 3, step j: the 128-column generated kernels -- parity tests, then A/B timing
mkdir -p gpurun_out/r03j
timeout 300 python -m pytest tests/test_gpu_round3.py -x -q -k "tiled" > gpurun_out/r03j/tests.log 2>&1; echo "tests rc=$?" >> gpurun_out/r03j/tests.log
tail -15 gpurun_out/r03j/tests.log
timeout 200 python tools/bench_fr128.py > gpurun_out/r03j/bench_fr128.log 2>&1; echo "rc=$?" >> gpurun_out/r03j/bench_fr128.log
cat gpurun_out/r03j/bench_fr128.log

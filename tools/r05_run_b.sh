#!/bin/bash
# round 5, GPU call B: the fixes of call A's failures, the bias-or-noise diagnostic of the stable 22-layer case, QMatMul timing
R=$GRAFT_REPO_ROOT; OUT=$R/gpurun_out/r05b; mkdir -p $OUT; cd $R
timeout 900 python -m pytest tests/test_gpu_round5.py tests/test_gpu_round4.py -q -x -k "qmatmul or per_group or grouped" > $OUT/gpu_tests_fix.log 2>&1; tail -8 $OUT/gpu_tests_fix.log
timeout 900 python tools/stable_depth_diag.py 2>&1 | grep -v amdgpu.ids > $OUT/stable_depth_diag.log; cat $OUT/stable_depth_diag.log
timeout 600 python tools/bench_qmatmul.py 2>&1 | grep -v amdgpu.ids > $OUT/bench_qmatmul.log; cat $OUT/bench_qmatmul.log

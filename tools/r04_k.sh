#!/bin/bash
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r04k; mkdir -p $O; cd $R
timeout 1500 python -m pytest tests/test_gpu_round4.py -m gpu -q -s > $O/t4.log 2>&1; grep -E "passed|failed|^E " $O/t4.log | cut -c1-600

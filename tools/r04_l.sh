#!/bin/bash
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r04l; mkdir -p $O; cd $R
timeout 900 python -m pytest tests/test_gpu_round4.py -m gpu -x -q -k "packed" > $O/w4.log 2>&1; grep -E "passed|failed|^E |Error" $O/w4.log | cut -c1-500 | tail -12

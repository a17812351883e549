#!/bin/bash
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r04j; mkdir -p $O; cd $R
timeout 1500 python -m pytest tests/test_gpu_round4.py -m gpu -q -s > $O/t4.log 2>&1; grep -E "full depth|passed|failed|Error|error|assert|^E " $O/t4.log | cut -c1-1200 | tail -30
timeout 900 python -m pytest tests -m gpu -x -q --deselect tests/test_gpu_round4.py > $O/tests.log 2>&1; grep -E "passed|failed" $O/tests.log | tail -3

#!/bin/bash
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r04e; mkdir -p $O; cd $R
timeout 900 python -m pytest tests -m gpu -x -q > $O/tests.log 2>&1; grep -E "passed|failed|Error|error" $O/tests.log | tail -5
bash tools/r04_d.sh "$@" 2>&1 | grep "^==\|^-- GEMM\|period"

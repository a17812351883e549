#!/bin/bash
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r04m; mkdir -p $O; cd $R
timeout 900 python -m pytest tests -m gpu -x -q -k "gemm or tiled or segmented or gated or int8 or qlinear or layer or packed" > $O/t.log 2>&1; grep -E "passed|failed|^E " $O/t.log | cut -c1-400 | tail -8
bash tools/r04_d.sh "$@" 2>&1 | grep "^==\|^-- GEMM(gauss\|period" | paste - - - - | cut -c1-300

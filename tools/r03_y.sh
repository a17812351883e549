#!/bin/bash
O=gpurun_out/r03y; mkdir -p $O
timeout 900 python -m pytest tests -m gpu -x -q -k "norm or fuse or layer or recipes or prefill or toy_lm" > $O/tests.log 2>&1; tail -6 $O/tests.log
bash tools/r03_x.sh 2>&1 | head -2
bash tools/r03_s.sh 2>&1 | tail -5

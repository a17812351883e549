#!/bin/bash
O=gpurun_out/r03r; mkdir -p $O
timeout 600 python -m pytest tests -m gpu -x -q -k "gated or tiled or fuse or layer or recipes" > $O/tests.log 2>&1; tail -4 $O/tests.log
timeout 200 python tools/bench_fr128.py 2>&1 | grep gated_lookup

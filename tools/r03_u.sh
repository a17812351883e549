#!/bin/bash
O=gpurun_out/r03u2; mkdir -p $O
timeout 900 python -m pytest tests -m gpu -x -q -k "attention or fuse or layer or recipes or prefill or chunk or continuation" > $O/tests.log 2>&1; tail -15 $O/tests.log

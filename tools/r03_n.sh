#!/bin/bash
O=gpurun_out/r03n; mkdir -p $O
timeout 900 python tools/decode_context_sweep.py > $O/decode_context_sweep.log 2>&1; echo "rc=$?" >> $O/decode_context_sweep.log
grep -v amdgpu.ids $O/decode_context_sweep.log
timeout 600 python -m pytest tests -m gpu -x -q -k "decode or generate or fuse or layer" > $O/tests.log 2>&1; tail -3 $O/tests.log

#!/bin/bash
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/${TAG:-r03s}; mkdir -p $O; cd $R
timeout 900 python -m pytest tests -m gpu -x -q -k "decode or producer" > $O/pytest.log 2>&1; grep -E "passed|failed|Error|error|assert" $O/pytest.log | tail -8
LAYERS=22 timeout 600 python tools/prof_decode_engine.py 2>&1 | grep "graph ms/token"
export MQ_LIB_PATH=$R/mobilequant_amd/lib/stamps/libmobilequant_amd.so
LAYERS=6 timeout 600 python tools/decode_stamps.py 2>&1 | grep "gemv\|attention\|graph" | cut -c1-150

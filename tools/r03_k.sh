#!/bin/bash
# round 3, step k: key-split prefill attention -- parity tests, then timing at S = 512 / 2048 / 4096 and a kernel trace
O=gpurun_out/r03k; mkdir -p $O
timeout 600 python -m pytest tests -m gpu -x -q -k "attention or fuse or layer or decode_case or prefill" > $O/tests.log 2>&1; echo "tests rc=$?" >> $O/tests.log
tail -5 $O/tests.log
for S in 512 2048 4096; do MQ_ATT_S=$S timeout 120 python tools/prof_attention.py 2>&1 | tail -1; done
cd /tmp; export TMPDIR=/tmp
timeout 300 rocprofv3 --kernel-trace --stats -d /tmp/pa -o p -- python $GRAFT_REPO_ROOT/tools/prof_attention.py > /dev/null 2>&1
python $GRAFT_REPO_ROOT/tools/pmc_summary.py /tmp/pa/p_results.db 2>/dev/null | grep "mq::"

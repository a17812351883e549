#!/bin/bash
# W4A8 layer kernel trace
O=$GRAFT_REPO_ROOT/gpurun_out/r03m; mkdir -p $O
cd /tmp; export TMPDIR=/tmp
MQ_LAYER_WBITS=4 timeout 600 rocprofv3 --kernel-trace --stats -d /tmp/pl -o p -- python $GRAFT_REPO_ROOT/tools/prof_layer.py > $O/prof_layer_w4.log 2>&1
python $GRAFT_REPO_ROOT/tools/pmc_summary.py /tmp/pl/p_results.db 2>/dev/null | grep "mq::" > $O/layer_trace_w4.summary.txt; cat $O/layer_trace_w4.summary.txt; tail -2 $O/prof_layer_w4.log

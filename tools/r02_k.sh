#!/bin/bash
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r02k; mkdir -p $O; cd /tmp; export TMPDIR=/tmp
cd $R && timeout 600 python -m pytest tests/test_gpu_round2.py -m gpu -x -q -k decode > $O/pytest.log 2>&1; tail -3 $O/pytest.log
cd /tmp
LAYERS=22 timeout 600 python $R/tools/prof_decode_engine.py 2>&1 | grep "ms/token"
timeout 600 rocprofv3 --kernel-trace --stats -d /tmp/prof_dec -o p -- python $R/tools/prof_decode_engine.py > $O/run.log 2>&1
python $R/tools/pmc_summary.py /tmp/prof_dec/p_results.db > $O/decode_trace.summary.txt 2>&1
grep "ms/token" $O/run.log; grep "mq::decode\|decode_gemv" $O/decode_trace.summary.txt

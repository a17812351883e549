#!/bin/bash
# round-3 profile capture: bench.py kernel trace + PMC passes; decode-engine kernel trace (graph replays, 22 layers); attention PMC
R=$GRAFT_REPO_ROOT
bash $R/tools/prof_bench.sh r03 pmc > /dev/null 2>&1
O=$R/gpurun_out/prof_bench_r03
cd /tmp; export TMPDIR=/tmp
LAYERS=22 PREFETCH=0.5 timeout 300 rocprofv3 --kernel-trace --stats -d /tmp/pd -o p -- python $R/tools/prof_decode_engine.py > $O/decode_engine.log 2>&1
python $R/tools/pmc_summary.py /tmp/pd/p_results.db 2>/dev/null | grep "mq::\|index" > $O/decode_engine_trace.summary.txt
timeout 300 rocprofv3 --pmc SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_WAIT_INST_ANY SQ_VALU_MFMA_BUSY_CYCLES --kernel-trace -d /tmp/pa -o p -- python $R/tools/prof_attention.py > /dev/null 2>&1
python $R/tools/pmc_summary.py /tmp/pa/p_results.db > $O/attention_pmc_sq.summary.txt 2>&1
ls -la $O; head -30 $O/trace.summary.txt; cat $O/decode_engine_trace.summary.txt

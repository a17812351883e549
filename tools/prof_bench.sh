#!/bin/bash
# rocprofv3 kernel trace (+ optional PMC passes) of bench.py on the GPU box.  usage: prof_bench.sh <tag> [pmc]
TAG=${1:-r01}; R=$GRAFT_REPO_ROOT; OUT=$R/gpurun_out/prof_bench_$TAG; mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
CMD="python $R/bench.py --steps 100 --warmup 10 --no-cpu-baseline"
timeout 300 rocprofv3 --kernel-trace --stats -d $OUT/trace -o t -- $CMD > $OUT/trace.log 2>&1
if [ "$2" = "pmc" ]; then
  timeout 300 rocprofv3 --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_VALU_MFMA_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_WAVES SQ_INSTS_VALU_MFMA_MOPS_I8 --kernel-trace -d $OUT/pmc1 -o p -- $CMD > $OUT/pmc1.log 2>&1
  timeout 300 rocprofv3 --pmc FETCH_SIZE --kernel-trace -d $OUT/pmc_fetch -o p -- $CMD > $OUT/pmc_fetch.log 2>&1
  timeout 300 rocprofv3 --pmc WRITE_SIZE TCC_HIT_sum TCC_MISS_sum --kernel-trace -d $OUT/pmc_write -o p -- $CMD > $OUT/pmc_write.log 2>&1
fi
tail -n 2 $OUT/trace.log

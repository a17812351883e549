#!/bin/bash
# generic: rocprofv3 --kernel-trace --stats of an arbitrary command, summary only.  usage: prof_cmd.sh <tag> -- <cmd...>
TAG=$1; shift; shift
R=$GRAFT_REPO_ROOT; OUT=$R/gpurun_out/prof_$TAG; mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
timeout 300 rocprofv3 --kernel-trace --stats -d /tmp/prof_$TAG -o p -- "$@" > $OUT/run.log 2>&1
python $R/tools/pmc_summary.py /tmp/prof_$TAG/p_results.db > $OUT/trace.summary.txt 2>&1
rm -rf /tmp/prof_$TAG
grep -E "gemv|quantize_rows|gemm_i8" $OUT/trace.summary.txt

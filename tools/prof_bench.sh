#!/bin/bash
# rocprofv3 kernel trace (+ PMC passes) of bench.py on the GPU box; keeps only text summaries (the rocpd
# databases are tens of MB).  usage: prof_bench.sh <tag> [pmc]     (run through gpurun)
TAG=${1:-r04}; R=$GRAFT_REPO_ROOT; OUT=$R/gpurun_out/prof_bench_$TAG; mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
CMD="python $R/bench.py --steps 100 --warmup 10 --no-cpu-baseline"
# PMC passes profile the headline legs only (timed steps + the GEMM / quantize kernels alone): counter collection over the
# whole variant suite (library GEMMs of the calibration leg, graph replays of the decode engine) crashed rocprofv3 on this pool
PMC_CMD="python $R/bench.py --steps 100 --warmup 10 --headline-only"
pass() {  # name, rocprofv3 args...
  local name=$1; shift
  local cmd=$CMD; case $name in pmc_*) cmd=$PMC_CMD;; esac
  timeout 900 rocprofv3 "$@" -d /tmp/prof_$name -o p -- $cmd > $OUT/$name.log 2>&1
  python $R/tools/pmc_summary.py /tmp/prof_$name/p_results.db > $OUT/$name.summary.txt 2>&1
  grep -E '^\{"metric"' $OUT/$name.log > $OUT/$name.bench.json
  rm -rf /tmp/prof_$name; tail -c 2000 $OUT/$name.log > $OUT/$name.log.tail; rm -f $OUT/$name.log
}
pass trace --kernel-trace --stats
if [ "$2" = "pmc" ]; then
  pass pmc_sq --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_VALU_MFMA_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_WAVES SQ_INSTS_VALU_MFMA_MOPS_I8 --kernel-trace
  pass pmc_lds --pmc SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_ACTIVE_INST_LDS SQ_WAIT_INST_LDS SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INST_CYCLES_VMEM_RD GRBM_GUI_ACTIVE --kernel-trace
  pass pmc_fetch --pmc FETCH_SIZE --kernel-trace
  pass pmc_write --pmc WRITE_SIZE TCC_HIT_sum TCC_MISS_sum --kernel-trace
fi
cat $OUT/trace.summary.txt | head -40

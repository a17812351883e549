#!/bin/bash
# rocprofv3 counter passes of the prefill attention alone (tools/prof_attention.py), int8 against f16 score contraction.
# usage (under gpurun): bash tools/prof_attention_pmc.sh <tag>
TAG=${1:-att}; R=$GRAFT_REPO_ROOT; OUT=$R/gpurun_out/prof_$TAG; mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
run() {  # name f16 counters...
  local name=$1 f16=$2; shift; shift
  MQ_ATT_F16=$f16 MQ_ATT_ITERS=3 timeout 300 rocprofv3 --pmc "$@" --kernel-trace -d /tmp/pa_$name -o p -- python $R/tools/prof_attention.py > $OUT/$name.log 2>&1
  python $R/tools/pmc_summary.py /tmp/pa_$name/p_results.db 2>&1 | grep -A12 "attention_quant_kernel" > $OUT/$name.summary.txt
  rm -rf /tmp/pa_$name
}
for f in 0 1; do
  run sq_f$f $f SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_VALU_MFMA_BUSY_CYCLES SQ_ACTIVE_INST_ANY
  run lds_f$f $f SQ_INSTS_LDS SQ_ACTIVE_INST_LDS SQ_WAIT_INST_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_VMEM_RD SQ_INST_CYCLES_VMEM_RD SQ_ACTIVE_INST_MISC
  run misc_f$f $f SQ_INSTS_SALU SQ_INSTS_SMEM SQ_ACTIVE_INST_SCA SQ_INSTS_VALU_TRANS SQ_ACTIVE_INST_EXP_GDS SQ_INST_LEVEL_LDS SQ_INST_LEVEL_VMEM GRBM_GUI_ACTIVE
done
tail -n +1 $OUT/*.summary.txt

#!/bin/bash
# PMC passes over one GEMM variant (run on the GPU box through gpurun).  usage: prof_gemm.sh <variant> <tag> [probe args...]
V=${1:-1}; TAG=${2:-gemm}; shift 2
R=$GRAFT_REPO_ROOT; OUT=$R/gpurun_out/prof_$TAG; mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
P="$R/tools/mq_probe prof $V 0 10 $@"
rocprofv3 --kernel-trace --stats -d $OUT/trace -o t -- $P > $OUT/trace.log 2>&1
rocprofv3 --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES GRBM_GUI_ACTIVE SQ_VALU_MFMA_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_WAVES --kernel-trace -d $OUT/pmc1 -o p -- $P > $OUT/pmc1.log 2>&1
rocprofv3 --pmc SQ_INSTS_VALU_MFMA_MOPS_I8 SQ_INSTS_MFMA SQ_ACTIVE_INST_LDS SQ_WAIT_INST_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_ACTIVE_INST_VMEM SQ_INST_CYCLES_VMEM_RD --kernel-trace -d $OUT/pmc2 -o p -- $P > $OUT/pmc2.log 2>&1
rocprofv3 --pmc FETCH_SIZE TCC_HIT_sum TCC_MISS_sum --kernel-trace -d $OUT/pmc3 -o p -- $P > $OUT/pmc3.log 2>&1
rocprofv3 --pmc WRITE_SIZE TCC_REQ_sum TA_TA_BUSY_sum --kernel-trace -d $OUT/pmc4 -o p -- $P > $OUT/pmc4.log 2>&1
find $OUT -name "*.csv" | head -30

#!/bin/bash
# PMC passes over the fused TinyLlama layer (tools/prof_layer.py): SQ counters, fabric reads, HBM writes -- separate passes, kernel trace only
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r03cc; mkdir -p $O
cd /tmp; export TMPDIR=/tmp
pass() { name=$1; shift
  timeout 600 rocprofv3 "$@" --kernel-trace -d /tmp/lp_$name -o p -- python $R/tools/prof_layer.py > $O/$name.log 2>&1
  python $R/tools/pmc_summary.py /tmp/lp_$name/p_results.db > $O/layer_$name.summary.txt 2>&1; rm -rf /tmp/lp_$name; tail -2 $O/$name.log; }
pass pmc_sq --pmc SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_WAIT_INST_ANY SQ_VALU_MFMA_BUSY_CYCLES
pass pmc_fetch --pmc FETCH_SIZE
pass pmc_write --pmc WRITE_SIZE TCC_HIT_sum TCC_MISS_sum
grep -A3 "fr128\|norm_tiled8\|gated_lookup\|attention_quant\|fr_pair" $O/layer_pmc_fetch.summary.txt | head -40

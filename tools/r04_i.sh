#!/bin/bash
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r04i; mkdir -p $O; cd $R
for st in "20 5" "200 20" "20 5" "200 20"; do set -- $st
  timeout 600 python bench.py --steps $1 --warmup $2 --headline-only --no-cpu-baseline 2>&1 | grep '^{"metric"' | python -c "
import json,sys
d=json.loads(sys.stdin.read()); r=d['roofline']; t=d['timing']
print('clk', r.get('sustained_mhz'), r.get('sustained_mhz_xcd_min_max'), r.get('frac_clock_adjusted'), r['zero_filled_operands'].get('sustained_mhz'), 'steps', d['steps'], 'value', d['value'], 'ms_per_step', d['ms_per_step'], 'host', t['host_clock_ms_per_step'], 'sorted', t['ms_per_step_rank0_sorted'], 'gemm_us', r.get('avg_launch_us'), 'quant', r['quantize_kernel']['avg_launch_us'])"
done

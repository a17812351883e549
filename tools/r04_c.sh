#!/bin/bash
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r04c; mkdir -p $O; cd $R
timeout 900 python -m pytest tests -m gpu -x -q > $O/tests.log 2>&1; grep -E "passed|failed|Error|error" $O/tests.log | tail -5
for t in r3 new r3 new; do
  p=$R/mobilequant_amd/lib/$t/libmobilequant_amd.so; [ $t = new ] && p=$R/mobilequant_amd/lib/libmobilequant_amd.so
  echo "== $t" >> $O/bench.log
  MQ_LIB_PATH=$p timeout 600 python bench.py --steps 20 --warmup 5 --headline-only --no-cpu-baseline 2>/dev/null | grep '^{"metric"' | python -c "
import json,sys
d=json.loads(sys.stdin.read()); r=d['roofline']
print('value', d['value'], 'ms_per_step', d['ms_per_step'], 'gemm_us', r.get('avg_launch_us'), 'frac', r['frac'], 'zero', r.get('zero_filled_operands',{}).get('avg_launch_us'))" >> $O/bench.log
done
cat $O/bench.log
MQ_LIB_PATH=$R/mobilequant_amd/lib/frs/libmobilequant_amd.so HOLE_ONLY=xcd timeout 600 python tools/hole_probe.py 2>&1 | grep -v amdgpu.ids > $O/hole.log
head -24 $O/hole.log

#!/bin/bash
# round 4, lease a: parity of the new preamble + same-box A/B of the launch hole (entry stamps) and of the headline bench
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r04a; mkdir -p $O; cd $R
timeout 900 python -m pytest tests -m gpu -x -q > $O/tests.log 2>&1; tail -4 $O/tests.log
export DVFS_FILLS=gauss DVFS_VARIANTS=11 DVFS_GAP=1 DVFS_EAGER=1
for t in r3frs frs r3frs frs; do
  echo "== $t" >> $O/gap.log
  MQ_LIB_PATH=$R/mobilequant_amd/lib/$t/libmobilequant_amd.so timeout 300 python tools/dvfs_probe.py 2>&1 | grep -v amdgpu.ids >> $O/gap.log
done
cat $O/gap.log
for t in r3 new r3 new; do
  p=$R/mobilequant_amd/lib/$t/libmobilequant_amd.so; [ $t = new ] && p=$R/mobilequant_amd/lib/libmobilequant_amd.so
  echo "== $t" >> $O/bench.log
  MQ_LIB_PATH=$p timeout 600 python bench.py --steps 20 --warmup 5 --headline-only --no-cpu-baseline 2>/dev/null | grep '^{"metric"' | python -c "
import json,sys
d=json.loads(sys.stdin.read()); r=d['roofline']
print('value', d['value'], 'ms_per_step', d['ms_per_step'], 'gemm_us', r.get('avg_launch_us'), 'frac', r['frac'], 'zero', r.get('zero_filled_operands'))" >> $O/bench.log
done
cat $O/bench.log

#!/bin/bash
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/${TAG:-r03v}; mkdir -p $O; cd $R
timeout 1500 python bench.py > $O/bench.json 2> $O/bench.err; echo "rc=$?" >> $O/bench.err
tail -3 $O/bench.err; python -c "
import json; d=json.load(open('$O/bench.json')); print(d['value'], d['ms_per_step'], d['timing']['ms_per_step_rank0_sorted']); r=d['roofline']; print(r['avg_launch_us'], r['frac'], r['zero_filled_operands'], r['traffic_source']);
print({k:(v if not isinstance(v,dict) else {kk:vv for kk,vv in v.items() if kk!='scope'}) for k,v in d['decode'].items() if 'linears' not in k and k!='scope'})"

#!/bin/bash
O=gpurun_out/r03p; mkdir -p $O
timeout 900 python -m pytest tests -m gpu -x -q -k "attention or fuse or layer or recipes or gemma or prefill" > $O/tests.log 2>&1; tail -5 $O/tests.log
timeout 600 python - > $O/families.log 2>&1 <<'PY'
import torch, bench
dev = torch.device("cuda:0")
for fam, wb in (("gemma_2b", 4), ("tinyllama", 8)):
    print(fam, wb, bench.bench_layer_full(dev, modes=("fused", "composite"), wbits=wb, family=fam), flush=True)
PY
grep -v amdgpu.ids $O/families.log
cd /tmp; export TMPDIR=/tmp
cat > /tmp/pg.py <<'PY'
import sys; sys.path.insert(0, "/root/repo")
import torch, bench
print(bench.bench_layer_full(torch.device("cuda:0"), modes=("fused",), wbits=4, family="gemma_2b"))
PY
timeout 600 rocprofv3 --kernel-trace --stats -d /tmp/pl -o p -- python /tmp/pg.py > /dev/null 2>&1
python $GRAFT_REPO_ROOT/tools/pmc_summary.py /tmp/pl/p_results.db 2>/dev/null | grep "mq::" > $GRAFT_REPO_ROOT/$O/gemma_layer_trace.txt; cat $GRAFT_REPO_ROOT/$O/gemma_layer_trace.txt

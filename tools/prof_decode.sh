#!/bin/bash
# rocprofv3 --kernel-trace of the decode GEMV launches (HBM-resident weight sets), per-kernel durations grouped by grid size.
# usage (on the GPU box): bash tools/prof_decode.sh <tag>
TAG=${1:-decode}
R=$GRAFT_REPO_ROOT; OUT=$R/gpurun_out/prof_$TAG; mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
timeout 300 rocprofv3 --kernel-trace -d /tmp/prof_$TAG -o p -- python $R/tools/bench_gemv_cache.py --hbm-only > $OUT/run.log 2>&1
python - /tmp/prof_$TAG/p_results.db > $OUT/by_grid.txt 2>&1 <<'PY'
import sqlite3, sys
c = sqlite3.connect(sys.argv[1])
cols = [r[1] for r in c.execute("pragma table_info(kernels)")]
print("columns:", cols)
gcol = next((x for x in ("grid_x", "grid_size_x", "grid_size") if x in cols), None)
q = f"select name, {gcol}, count(*), avg(duration), min(duration) from kernels group by name, {gcol}" if gcol else \
    "select name, 0, count(*), avg(duration), min(duration) from kernels group by name"
for k, g, n, d, mn in c.execute(q):
    if "gemv" in k or n > 40:
        print(f"{k[:70]:70s} grid={g} n={n} mean {d/1e3:.2f} us min {mn/1e3:.2f} us")
PY
rm -rf /tmp/prof_$TAG
tail -8 $OUT/run.log; cat $OUT/by_grid.txt

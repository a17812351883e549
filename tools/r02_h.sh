#!/bin/bash
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r02h; mkdir -p $O; cd $R
timeout 1500 python -m pytest tests -m gpu -x -q > $O/pytest.log 2>&1; echo "pytest rc=$?" >> $O/pytest.log
tail -25 $O/pytest.log

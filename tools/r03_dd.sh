#!/bin/bash
for tag in ec0 "" ec3 ec4; do
  if [ -n "$tag" ]; then export MQ_LIB_PATH=$GRAFT_REPO_ROOT/mobilequant_amd/lib/$tag/libmobilequant_amd.so; else unset MQ_LIB_PATH; fi
  echo "== ${tag:-ec2 (default)}: $(for S in 512 2048 4096; do MQ_ATT_S=$S timeout 120 python tools/prof_attention.py 2>&1 | tail -1 | sed 's/.*eager: //'; done | tr '\n' ' ')"
done

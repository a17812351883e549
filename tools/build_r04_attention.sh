#!/bin/bash
# The round-4 attention kernel as a library of its own (mobilequant_amd/lib/r04att/) for same-box A/B timing against the current one:
#   MQ_LIB_PATH=mobilequant_amd/lib/r04att/libmobilequant_amd.so python tools/att_f16_ab.py      (both columns then time the round-4 kernel)
# mq_attention.hip and the headers of commit 779f70f (end of round 4's attention work) + today's other objects; the two tuning
# switches the Python binding looks up are added as no-ops.  Run after `python -m mobilequant_amd.build`.
set -e
R0=$(cd "$(dirname "$0")/.." && pwd)
# needs the git history: run it HERE before gpurun (the built library travels with the snapshot; the GPU box has no .git)
git -C $R0 rev-parse --git-dir > /dev/null 2>&1 || { echo "no git history here: keeping $R0/mobilequant_amd/lib/r04att as it is"; exit 0; }
R=$(cd "$(dirname "$0")/.." && pwd); T=$(mktemp -d); mkdir -p $R/mobilequant_amd/lib/r04att $T/inc
git -C $R show 779f70f:mobilequant_amd/csrc/mq_attention.hip > $T/mq_attention_r04.hip
git -C $R show 779f70f:include/mobilequant_amd.h > $T/inc/mobilequant_amd.h
git -C $R show 779f70f:include/mobilequant_amd_tuning.h > $T/inc/mobilequant_amd_tuning.h
cat >> $T/mq_attention_r04.hip <<'EOC'
extern "C" int mq_attention_set_f16(int) { return 0; }
extern "C" int mq_attention_set_pair(int) { return 1; }
EOC
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -ffp-contract=off -fno-fast-math -fhip-fp32-correctly-rounded-divide-sqrt \
  -I$T/inc -I$R/mobilequant_amd/csrc -mllvm -amdgpu-mfma-vgpr-form=1 -c -x hip $T/mq_attention_r04.hip -o $R/mobilequant_amd/lib/r04att/mq_attention.o
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC $(ls $R/mobilequant_amd/lib/*.o | grep -v mq_attention.o) $R/mobilequant_amd/lib/r04att/mq_attention.o \
  -o $R/mobilequant_amd/lib/r04att/libmobilequant_amd.so
rm -rf $T; ls -la $R/mobilequant_amd/lib/r04att/libmobilequant_amd.so

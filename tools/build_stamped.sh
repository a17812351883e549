#!/bin/bash
# Stamp build of the free-running GEMM (s_memtime / s_memrealtime stamps incl. the kernel-entry time) into mobilequant_amd/lib/frs/.
# usage: tools/build_stamped.sh [tag] [ENV=1 ...]   (generator switches, e.g. MQ_FR_NO_A=1); the production .inc is restored at the end.
cd "$(dirname "$0")/.."
tag=${1:-frs}; shift
env MQ_FR_STAMP=1 "$@" python tools/gen_fr_asm.py fr > /dev/null || exit 1
python - <<PY || exit 1
from mobilequant_amd import build as b
print(b.build(tag="$tag", only=["mq_gemm.hip"], extra_flags=["-DMQ_GEMM_ABLATE"]))
PY
python tools/gen_fr_asm.py fr > /dev/null

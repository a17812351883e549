#!/bin/bash
# what-if builds of the free-running generated GEMM (tools/gen_fr_asm.py switches), each with s_memtime stamps, into
# mobilequant_amd/lib/<tag>/ + tools/mq_probe_<tag>.  Only mq_gemm.hip is recompiled; the .inc in csrc is restored at the end.
cd "$(dirname "$0")/.."
L=mobilequant_amd/lib
FLAGS="--offload-arch=gfx950 -O3 -std=c++17 -fPIC -ffp-contract=off -fno-fast-math -fhip-fp32-correctly-rounded-divide-sqrt -Iinclude -Imobilequant_amd/csrc -DMQ_GEMM_ABLATE"
for cfg in "frs:" "frs_noa:MQ_FR_NO_A=1" "frs_now:MQ_FR_NO_W=1" "frs_nord:MQ_FR_NO_READ=1" "frs_nomfma:MQ_FR_NO_MFMA=1" "frs_mfmaonly:MQ_FR_NO_A=1 MQ_FR_NO_W=1 MQ_FR_NO_READ=1" $EXTRA_CFGS; do
  tag=${cfg%%:*}; envs=${cfg#*:}
  mkdir -p $L/$tag
  env MQ_FR_STAMP=1 $envs python tools/gen_fr_asm.py > /dev/null || exit 1
  /opt/rocm/bin/hipcc $FLAGS -c mobilequant_amd/csrc/mq_gemm.hip -o $L/$tag/mq_gemm.o || exit 1
  /opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC $L/mq_elementwise.o $L/mq_reduce.o $L/$tag/mq_gemm.o $L/mq_gemv.o $L/mq_norm.o -o $L/$tag/libmobilequant_amd.so || exit 1
  python -c "from mobilequant_amd import build as b; b.build_probe('$tag')" || exit 1
  echo "built $tag ($envs)"
done
python tools/gen_fr_asm.py

// Decode-shape (M <= 8) int8 GEMV path behind mq_w8a8_linear; see mq_gemv.hip.
#pragma once
#include "mq_common.h"

namespace mq {

struct GemvArgs {
  const int8_t* a;
  const int8_t* w;
  int M, N, K;
  const int32_t* a_rowsum;
  const float* alpha;
  const int32_t* w_zp;
  const int32_t* col_term;
  const float* bias;
  const float* out_scale;
  const float* out_offset;
  float out_qmin, out_qmax;
  void* out;
  int out_dtype;
  // fused activation quantize (mq_w8a8_linear_f32in): fp32 activations + their grid; a / a_rowsum unused
  const float* x_f32;
  const float* xq_scale;
  const float* xq_offset;
  float xq_qmin, xq_qmax;
  int xq_shift;
  int w4;   // weights are packed unsigned nibbles (mq_pack_w4 layout), row stride K/2 bytes
};

int run_gemv(const GemvArgs& g, hipStream_t st);

}  // namespace mq

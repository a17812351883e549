// Decode-shape QLinear (M <= 16 tokens): int8 GEMV, weight-streaming, HBM-bound.
//
// Same arithmetic and epilogue as the MFMA GEMM (mq_gemm.hip):
//   out[m,n] = alpha[n] * ( sum_k a[m,k]*w[n,k] - w_zp[n]*a_rowsum[m] + col_term[n] ) + bias[n]  (+ output quantizer)
// but at M = 1 the matrix cores would idle on a 16-row tile, and every weight byte is used once per
// token, so the kernel is a stream of 16-byte weight loads straight into registers (no LDS round trip
// for W: it is not shared between waves) feeding v_dot4_i32_i8.  Algorithmic bytes per token: N*K weight
// bytes (+ K activation bytes, L2 resident).  TinyLlama-1.1B W8: 22 x 44.04 MB = 0.969 GB per token.
//
// Mapping: one wave owns ROWS_PER_WAVE consecutive output rows; its 64 lanes split K into 16-byte
// chunks (lane l takes chunks l, l+64, ...), so a wave instruction reads 1 KiB contiguous per row;
// ROWS_PER_WAVE loads are in flight per lane before the first dot product is needed.  Activations
// ([M,K] int8, a few KiB) are staged in LDS once per workgroup and re-read from there.
#include <hip/hip_fp16.h>

#include "mq_gemv.h"

namespace mq {

typedef int v4i __attribute__((ext_vector_type(4)));

__device__ __forceinline__ int dot16(const v4i a, const v4i b, int c) {
#pragma unroll
  for (int e = 0; e < 4; ++e) c = __builtin_amdgcn_sdot4(a[e], b[e], c, false);
  return c;
}

#pragma clang fp contract(off)
// qmodule.py:286-287, the same expression tree as mq_elementwise.hip (bit-exact indices)
__device__ __forceinline__ int q_index_i(float x, float s, float o, float qmin, float qmax) {
  float q = __fadd_rn(rintf(__fdiv_rn(x, s)), o);
  return (int)fminf(fmaxf(q, qmin), qmax);
}

template <int MT, int ROWS, bool FUSEQ>   // MT: tokens per pass; ROWS: output rows per wave; FUSEQ: fp32 activations in
__global__ void __launch_bounds__(256) gemv_i8_kernel(const GemvArgs g) {
  extern __shared__ __attribute__((aligned(16))) char smem[];   // [M][K] int8 activations (+ M row sums when FUSEQ)
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int K = g.K, N = g.N, M = g.M;
  const int kchunks = K >> 4;
  int* s_rs = reinterpret_cast<int*>(smem + (size_t)M * K);
  // First trip of this wave's weight rows goes out BEFORE the activation staging: the kernel is one HBM
  // round trip long for the small decode matrices, so the weight latency must overlap the staging.
  const int n0 = (blockIdx.x * 4 + wave) * ROWS;
  v4i pw[ROWS], pw2[ROWS];
  {
    const int c = lane, c2 = lane + 64;
#pragma unroll
    for (int r = 0; r < ROWS; ++r) {
      const int n = (n0 + r < N) ? n0 + r : N - 1;
      const v4i* wr = reinterpret_cast<const v4i*>(g.w + (size_t)n * K);
      pw[r] = (c < kchunks) ? __builtin_nontemporal_load(wr + c) : v4i{0, 0, 0, 0};
      pw2[r] = (c2 < kchunks) ? __builtin_nontemporal_load(wr + c2) : v4i{0, 0, 0, 0};
    }
  }
  if constexpr (FUSEQ) {
    // quantize the fp32 activations straight into LDS (every workgroup repeats it: M*K*4 bytes from L2)
    // and reduce the row sums of the stored values with LDS atomics
    if (threadIdx.x < M) s_rs[threadIdx.x] = 0;
    __syncthreads();
    const float s = g.xq_scale[0], o = g.xq_offset[0];
    for (int i = threadIdx.x; i < M * (K >> 2); i += 256) {
      const float4 v = reinterpret_cast<const float4*>(g.x_f32)[i];
      const int q0 = q_index_i(v.x, s, o, g.xq_qmin, g.xq_qmax) - g.xq_shift, q1 = q_index_i(v.y, s, o, g.xq_qmin, g.xq_qmax) - g.xq_shift;
      const int q2 = q_index_i(v.z, s, o, g.xq_qmin, g.xq_qmax) - g.xq_shift, q3 = q_index_i(v.w, s, o, g.xq_qmin, g.xq_qmax) - g.xq_shift;
      reinterpret_cast<unsigned*>(smem)[i] = (q0 & 0xff) | ((q1 & 0xff) << 8) | ((q2 & 0xff) << 16) | ((unsigned)(q3 & 0xff) << 24);
      int part = q0 + q1 + q2 + q3;
      // lanes of a wave may straddle rows only if K < 256; K % 128 == 0 and a wave covers 256 elements per trip
      const int row = (i << 2) / K;
      part = wave_sum(part);          // K >= 256 here (checked on the host): the whole wave is in one row
      if (lane == 0) atomicAdd(&s_rs[row], part);
    }
  } else {
    const v4i* src = reinterpret_cast<const v4i*>(g.a);
    v4i* dst = reinterpret_cast<v4i*>(smem);
    for (int i = threadIdx.x; i < M * kchunks; i += 256) dst[i] = src[i];
  }
  __syncthreads();
  if (n0 >= N) return;
  float so = 1.f, oo = 0.f, inv_so = 1.f;
  const bool outq = g.out_scale != nullptr;
  if (outq) {
    so = g.out_scale[0];
    oo = g.out_offset[0];
    inv_so = __fdiv_rn(1.0f, so);
  }
  for (int mb = 0; mb < M; mb += MT) {
    int acc[MT][ROWS];
#pragma unroll
    for (int m = 0; m < MT; ++m)
#pragma unroll
      for (int r = 0; r < ROWS; ++r) acc[m][r] = 0;
    // two K chunks per trip: 2*ROWS independent 16-byte weight loads in flight per lane
    for (int c = lane; c < kchunks; c += 128) {
      const int c2 = c + 64;
      const bool has2 = c2 < kchunks;
      v4i wv[ROWS], wv2[ROWS];
#pragma unroll
      for (int r = 0; r < ROWS; ++r) {
        const int n = (n0 + r < N) ? n0 + r : N - 1;
        const v4i* wr = reinterpret_cast<const v4i*>(g.w + (size_t)n * K);
        if (c == lane) {                                            // first trip: already in registers
          wv[r] = pw[r];
          wv2[r] = pw2[r];
        } else {
          wv[r] = __builtin_nontemporal_load(wr + c);               // streamed once: bypass-friendly
          wv2[r] = has2 ? __builtin_nontemporal_load(wr + c2) : v4i{0, 0, 0, 0};
        }
      }
#pragma unroll
      for (int m = 0; m < MT; ++m) {
        if (mb + m < M) {
          const v4i av = *reinterpret_cast<const v4i*>(smem + (size_t)(mb + m) * K + c * 16);
          const v4i av2 = has2 ? *reinterpret_cast<const v4i*>(smem + (size_t)(mb + m) * K + c2 * 16) : v4i{0, 0, 0, 0};
#pragma unroll
          for (int r = 0; r < ROWS; ++r) acc[m][r] = dot16(wv2[r], av2, dot16(wv[r], av, acc[m][r]));
        }
      }
    }
#pragma unroll
    for (int m = 0; m < MT; ++m)
#pragma unroll
      for (int r = 0; r < ROWS; ++r) acc[m][r] = wave_sum(acc[m][r]);
    if (lane == 0) {
#pragma unroll
      for (int m = 0; m < MT; ++m) {
        if (mb + m >= M) break;
        const int rs = FUSEQ ? s_rs[mb + m] : (g.a_rowsum ? g.a_rowsum[mb + m] : 0);
#pragma unroll
        for (int r = 0; r < ROWS; ++r) {
          const int n = n0 + r;
          if (n >= N) break;
          const int t = (int)((unsigned)acc[m][r] - (unsigned)g.w_zp[n] * (unsigned)rs + (unsigned)g.col_term[n]);
          float f = __fadd_rn(__fmul_rn((float)t, g.alpha[n]), g.bias ? g.bias[n] : 0.f);
          const size_t idx = (size_t)(mb + m) * N + n;
          if (outq) {
            float q = rintf(f * inv_so) + oo;
            q = fminf(fmaxf(q, g.out_qmin), g.out_qmax);
            switch (g.out_dtype) {
              case MQ_F32: reinterpret_cast<float*>(g.out)[idx] = __fmul_rn(__fsub_rn(q, oo), so); break;
              case MQ_F16: reinterpret_cast<__half*>(g.out)[idx] = __float2half_rn(__fmul_rn(__fsub_rn(q, oo), so)); break;
              case MQ_U8: reinterpret_cast<uint8_t*>(g.out)[idx] = (uint8_t)(int)q; break;
              case MQ_I8: reinterpret_cast<int8_t*>(g.out)[idx] = (int8_t)((int)q - (g.out_qmin == 0.f ? 128 : 0)); break;
              case MQ_U16: reinterpret_cast<uint16_t*>(g.out)[idx] = (uint16_t)(int)q; break;
              default: reinterpret_cast<int16_t*>(g.out)[idx] = (int16_t)(int)q; break;
            }
          } else if (g.out_dtype == MQ_F32) {
            reinterpret_cast<float*>(g.out)[idx] = f;
          } else {
            reinterpret_cast<__half*>(g.out)[idx] = __float2half_rn(f);
          }
        }
      }
    }
  }
}

// Called by mq_w8a8_linear / mq_w8a8_linear_f32in for decode shapes (argument checks already done there).
int run_gemv(const GemvArgs& g, hipStream_t st) {
  const bool fuse = g.x_f32 != nullptr;
  const size_t lds = (size_t)g.M * g.K + (fuse ? 64 : 0);
  if (lds > 64 * 1024 || (fuse && g.K < 256)) {
    set_error("mq_w8a8_linear (decode path): M*K = %zu bytes of activations exceed the 64 KiB staging buffer", lds);
    return MQ_EUNSUPPORTED;
  }
  constexpr int ROWS = 2;
  const unsigned grid = (unsigned)((g.N + 4 * ROWS - 1) / (4 * ROWS));
#define MQ_GV(MT)                                                            \
  do {                                                                       \
    if (fuse) gemv_i8_kernel<MT, ROWS, true><<<grid, 256, lds, st>>>(g);    \
    else gemv_i8_kernel<MT, ROWS, false><<<grid, 256, lds, st>>>(g);        \
  } while (0)
  if (g.M == 1) MQ_GV(1);
  else if (g.M == 2) MQ_GV(2);
  else MQ_GV(4);
#undef MQ_GV
  MQ_LAUNCH_CHECK("mq_w8a8_linear(gemv)");
  return MQ_OK;
}

}  // namespace mq

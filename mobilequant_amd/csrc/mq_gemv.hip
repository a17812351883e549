// Decode-shape QLinear (M <= 8 tokens): int8 GEMV, weight-streaming.
//
// Same arithmetic and epilogue as the MFMA GEMM (mq_gemm.hip):
//   out[m,n] = alpha[n] * ( sum_k a[m,k]*w[n,k] - w_zp[n]*a_rowsum[m] + col_term[n] ) + bias[n]  (+ output quantizer)
// but at M = 1 the matrix cores would idle on a 16-row tile, and every weight byte is used once per
// token, so the kernel is a stream of 16-byte weight loads straight into registers (no LDS round trip
// for W: it is not shared between waves) feeding v_dot4_i32_i8.  Algorithmic bytes per token: N*K weight
// bytes (+ K activation bytes, L2 resident).  TinyLlama-1.1B W8: 22 x 44.04 MB = 0.969 GB per token.
//
// Structure: one fat workgroup (16 waves) per CU.  A wave owns whole output rows (row_base + wave + 16 t); its
// 64 lanes split K into 16-byte chunks (lane l takes chunks l, l + 64, ...: a wave instruction reads 1 KiB
// contiguous) and it issues ALL its weight loads (<= 8 per lane per pass; one pass covers every TinyLlama matrix)
// plus the per-row epilogue parameters before anything else.  The activations are quantized ONCE per CU
// (K / 1024 elements per thread) into LDS while those loads fly; reductions are DPP adds; the epilogue runs from
// registers.  Why: the first version (256-1408 small workgroups, 2 rows per wave, several load trips, bpermute
// reductions, parameter loads after the dot products) cost the same whether its weights came from L2, the Infinity
// Cache or HBM -- it was bound by its dependent chain, not by bytes.  Measured (tools/bench_gemv_cache.py, HBM-resident
// 1 GB weight sets, hipGraph): launch time = 3.7 us + bytes / 5.5 TB/s for all four TinyLlama shapes
// (4.2 / 4.6 / 5.8 / 7.9 us for 4.2 / 5.2 / 11.5 / 23 MB; v1: 4.5 / 5.0 / 8.5 / 9.7 us); a trivial kernel in the same
// graph costs 1.76 us per launch.
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
__device__ __forceinline__ int q_index_i(float x, float s, float inv_s, float o, float qmin, float qmax) {
  float q = __fadd_rn(rintf(div_by_scale(x, s, inv_s)), o);
  return (int)fminf(fmaxf(q, qmin), qmax);
}

// 64-lane integer sum with DPP row operations (6 VALU adds instead of 6 LDS-crossbar bpermutes); all lanes active
__device__ __forceinline__ int wave_sum_dpp(int v) {
  v += __builtin_amdgcn_update_dpp(0, v, 0xB1, 0xf, 0xf, true);    // quad_perm [1,0,3,2]
  v += __builtin_amdgcn_update_dpp(0, v, 0x4E, 0xf, 0xf, true);    // quad_perm [2,3,0,1]
  v += __builtin_amdgcn_update_dpp(0, v, 0x141, 0xf, 0xf, true);   // row_half_mirror
  v += __builtin_amdgcn_update_dpp(0, v, 0x140, 0xf, 0xf, true);   // row_mirror: every lane = its row-of-16 total
  v += __builtin_amdgcn_update_dpp(0, v, 0x142, 0xa, 0xf, false);  // row_bcast:15 into rows 1 and 3
  v += __builtin_amdgcn_update_dpp(0, v, 0x143, 0xc, 0xf, false);  // row_bcast:31 into rows 2 and 3
  return __builtin_amdgcn_readlane(v, 63);
}

constexpr int GV2_THREADS = 1024, GV2_WAVES = 16, GV2_INFLIGHT = 8;

template <int MT, bool FUSEQ, bool W4>
__global__ void __launch_bounds__(GV2_THREADS) gemv_i8_fat_kernel(const GemvArgs g, const int rows_per_wg) {
  extern __shared__ __attribute__((aligned(16))) char smem[];   // [M][K] int8 activations, then M row sums
  const int lane = threadIdx.x & 63;
  const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const int K = g.K, N = g.N, M = g.M;
  // a 16-byte weight chunk covers 16 k (int8) or 32 k (packed nibbles: low = k 0..15, high = k 16..31 of the block)
  const int kchunks = W4 ? K >> 5 : K >> 4;
  const int wrow = W4 ? K >> 1 : K;                               // weight row stride in bytes
  const int cpl = (kchunks + 63) >> 6;                            // chunks per lane per row
  int* s_rs = reinterpret_cast<int*>(smem + (size_t)M * K);
  const int row0 = blockIdx.x * rows_per_wg + wave;               // this wave's rows: row0 + 16 t
  const int row_end = (blockIdx.x + 1) * rows_per_wg < N ? (blockIdx.x + 1) * rows_per_wg : N;

  // ---- pass 0 weight loads + per-row parameters go out first ------------------------------------------
  v4i buf[GV2_INFLIGHT];
  auto issue_pass = [&](int t, int j) {                            // (t, j) = first (row slot, chunk slot) of the pass
#pragma unroll
    for (int u = 0; u < GV2_INFLIGHT; ++u) {
      const int row = row0 + GV2_WAVES * t;
      const int c = lane + 64 * j;
      if (row < row_end && c < kchunks)
        buf[u] = __builtin_nontemporal_load(reinterpret_cast<const v4i*>(g.w + (size_t)row * wrow) + c);
      else
        buf[u] = v4i{0, 0, 0, 0};
      if (++j == cpl) { j = 0; ++t; }
    }
  };
  issue_pass(0, 0);
  float p_alpha = 0.f, p_bias = 0.f;
  int p_zp = 0, p_ct = 0;
  {
    const int row = row0 + GV2_WAVES * lane;                       // lane t keeps the parameters of row slot t
    if (row < row_end) {
      p_alpha = g.alpha[row];
      p_zp = g.w_zp[row];
      p_ct = g.col_term[row];
      if (g.bias) p_bias = g.bias[row];
    }
  }
  float so = 1.f, oo = 0.f, inv_so = 1.f;
  const bool outq = g.out_scale != nullptr;
  if (outq) {
    so = g.out_scale[0];
    oo = g.out_offset[0];
    inv_so = __fdiv_rn(1.0f, so);
  }

  // ---- activations -> LDS, once per CU ---------------------------------------------------------------
  if constexpr (FUSEQ) {
    if (threadIdx.x < M) s_rs[threadIdx.x] = 0;
    __syncthreads();
    const float s = g.xq_scale[0], o = g.xq_offset[0];
    const float inv_s = __fdiv_rn(1.0f, s);
    for (int i = threadIdx.x; i < M * (K >> 2); i += GV2_THREADS) {
      const float4 v = reinterpret_cast<const float4*>(g.x_f32)[i];
      const int q0 = q_index_i(v.x, s, inv_s, o, g.xq_qmin, g.xq_qmax) - g.xq_shift, q1 = q_index_i(v.y, s, inv_s, o, g.xq_qmin, g.xq_qmax) - g.xq_shift;
      const int q2 = q_index_i(v.z, s, inv_s, o, g.xq_qmin, g.xq_qmax) - g.xq_shift, q3 = q_index_i(v.w, s, inv_s, o, g.xq_qmin, g.xq_qmax) - g.xq_shift;
      reinterpret_cast<unsigned*>(smem)[i] = (q0 & 0xff) | ((q1 & 0xff) << 8) | ((q2 & 0xff) << 16) | ((unsigned)(q3 & 0xff) << 24);
      // K % 256 == 0 (host check): a wave's 64 float4s never straddle a row and the trip count is a multiple of
      // 64, so every wave is fully active here
      const int part = wave_sum_dpp(q0 + q1 + q2 + q3);
      if (lane == 0) atomicAdd(&s_rs[(i << 2) / K], part);
    }
  } else {
    const v4i* src = reinterpret_cast<const v4i*>(g.a);
    v4i* dst = reinterpret_cast<v4i*>(smem);
    for (int i = threadIdx.x; i < M * (K >> 4); i += GV2_THREADS) dst[i] = src[i];
    if (threadIdx.x < M) s_rs[threadIdx.x] = g.a_rowsum ? g.a_rowsum[threadIdx.x] : 0;
  }
  __syncthreads();
  if (row0 >= row_end) return;

  // ---- dot products, DPP reduction and epilogue per completed row ---------------------------------------
  const int nslots = (row_end - row0 + GV2_WAVES - 1) / GV2_WAVES;   // rows of this wave
  for (int mb = 0; mb < M; mb += MT) {
    int acc[MT];
#pragma unroll
    for (int m = 0; m < MT; ++m) acc[m] = 0;
    int t = 0, j = 0;
    if (mb > 0) issue_pass(0, 0);                                  // M > MT: weights are re-streamed (L2) per token group
    while (t < nslots) {
      int t2 = t, j2 = j;
#pragma unroll
      for (int u = 0; u < GV2_INFLIGHT; ++u) {
        if (t2 < nslots) {                                         // wave-uniform
          int c = lane + 64 * j2;
          c = c < kchunks ? c : kchunks - 1;                       // buf[u] is zero there
#pragma unroll
          for (int m = 0; m < MT; ++m) {
            if (mb + m < M) {
              if constexpr (W4) {
                const v4i* ap = reinterpret_cast<const v4i*>(smem + (size_t)(mb + m) * K + c * 32);
                const v4i lo = buf[u] & 0x0f0f0f0f, hi = (buf[u] >> 4) & 0x0f0f0f0f;
                acc[m] = dot16(hi, ap[1], dot16(lo, ap[0], acc[m]));
              } else {
                const v4i av = *reinterpret_cast<const v4i*>(smem + (size_t)(mb + m) * K + c * 16);
                acc[m] = dot16(buf[u], av, acc[m]);
              }
            }
          }
          if (j2 == cpl - 1) {                                     // row slot t2 complete
            const int row = row0 + GV2_WAVES * t2;
            const float alpha = __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, p_alpha), t2));
            const float bias = __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, p_bias), t2));
            const int zp = __builtin_amdgcn_readlane(p_zp, t2), ct = __builtin_amdgcn_readlane(p_ct, t2);
#pragma unroll
            for (int m = 0; m < MT; ++m) {
              const int sum = wave_sum_dpp(acc[m]);
              acc[m] = 0;
              if (mb + m < M && lane == 0) {
                const int tt = (int)((unsigned)sum - (unsigned)zp * (unsigned)s_rs[mb + m] + (unsigned)ct);
                const float f = __fadd_rn(__fmul_rn((float)tt, alpha), bias);
                const size_t idx = (size_t)(mb + m) * N + row;
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
          if (++j2 == cpl) { j2 = 0; ++t2; }
        }
      }
      t = t2;
      j = j2;
      if (t < nslots) issue_pass(t, j);                            // next pass (matrices beyond 8 loads per lane)
    }
  }
}

// Called by mq_w8a8_linear / mq_w8a8_linear_f32in for decode shapes (argument checks already done there).
int run_gemv(const GemvArgs& g, hipStream_t st) {
  const bool fuse = g.x_f32 != nullptr;
  const size_t lds = (size_t)g.M * g.K + 64;
  if (lds > 64 * 1024 || (fuse && g.K < 256)) {
    set_error("mq_w8a8_linear (decode path): M*K = %zu bytes of activations exceed the 64 KiB staging buffer", lds);
    return MQ_EUNSUPPORTED;
  }
  // one fat workgroup per CU; a wave keeps at most 64 row slots (their parameters live one per lane)
  static std::atomic<int> cus_of[kMaxDevices];      // CU count per device (0 = not read yet)
  const int dev = current_device();
  int cus = cus_of[dev].load(std::memory_order_relaxed);
  if (!cus) {
    hipDeviceProp_t prop;
    cus = (hipGetDeviceProperties(&prop, dev) == hipSuccess && prop.multiProcessorCount > 0) ? prop.multiProcessorCount : 256;
    cus_of[dev].store(cus, std::memory_order_relaxed);
  }
  int rows_per_wg = (g.N + cus - 1) / cus;
  if (rows_per_wg > GV2_WAVES * 64) rows_per_wg = GV2_WAVES * 64;
  const unsigned grid = (unsigned)((g.N + rows_per_wg - 1) / rows_per_wg);
  const size_t lds_bytes = (size_t)g.M * g.K + 64;
#define MQ_GV(MT)                                                                                              \
  do {                                                                                                         \
    if (g.w4) {                                                                                                \
      if (fuse) gemv_i8_fat_kernel<MT, true, true><<<grid, GV2_THREADS, lds_bytes, st>>>(g, rows_per_wg);     \
      else gemv_i8_fat_kernel<MT, false, true><<<grid, GV2_THREADS, lds_bytes, st>>>(g, rows_per_wg);         \
    } else {                                                                                                   \
      if (fuse) gemv_i8_fat_kernel<MT, true, false><<<grid, GV2_THREADS, lds_bytes, st>>>(g, rows_per_wg);    \
      else gemv_i8_fat_kernel<MT, false, false><<<grid, GV2_THREADS, lds_bytes, st>>>(g, rows_per_wg);        \
    }                                                                                                          \
  } while (0)
  if (g.M == 1) MQ_GV(1);
  else if (g.M == 2) MQ_GV(2);
  else MQ_GV(4);
#undef MQ_GV
  MQ_LAUNCH_CHECK("mq_w8a8_linear(gemv)");
  return MQ_OK;
}

}  // namespace mq

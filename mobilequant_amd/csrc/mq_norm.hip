// QRMSNorm.forward (qmodule.py:469-530 around hf_model.py:184-195) in ONE pass over the activations.
//
// Reference op sequence on x [rows, cols] (fp32):
//   xi  = Qin(x)                                 input quantizer (16-bit per-tensor in every recipe), optional
//   r   = rsqrt(mean(xi^2, -1) + eps)            x.pow(2).mean(-1, keepdim=True); torch.rsqrt
//   y   = weight' * (xi * r) (+ bias)            weight' = Qw(weight), fake-quantised once by the caller
//   out = Qout(y)                                8-bit per-tensor activation grid, optional
// = 6 elementwise / reduction launches and ~9 passes over the tensor as composite torch ops.  Here a wave owns a
// row, keeps it in registers (cols <= 4096) or re-reads it (larger), and writes out -- and, when the output grid is
// an 8-bit one, the int8 indices + row sums the consumer linears (q/k/v, w1/w3) feed to the integer GEMM, so their
// activation quantize launches disappear (norm -> int8 -> GEMM chaining).
//
// Numerics: every elementwise op is the reference's (IEEE divide, round-half-even, separate mul/add; 1/sqrt with
// correctly rounded sqrt and divide, which is what the CPU reference computes).  The sum of squares is reduced in
// a different order than torch's (lane-strided partial sums, then a butterfly), so r can differ in the last bit
// and an output that sits within ~1e-7 relative of a rounding boundary can land on the neighbouring grid point:
// tests allow 1 LSB on < 0.1 % of the elements.
#include <hip/hip_fp16.h>

#include "mq_common.h"

namespace mq {

#pragma clang fp contract(off)

__device__ __forceinline__ float nq_clamp_nan(float q, float lo, float hi) {
  const float c = fminf(fmaxf(q, lo), hi);
  return q != q ? q : c;
}
// qmodule.py:286-290 with round_ste = (round(t) - t) + t, as in mq_elementwise.hip
__device__ __forceinline__ float nq_index(float x, float s, float inv_s, float o, float qmin, float qmax) {
  const float t = div_by_scale(x, s, inv_s);          // == x / s on the quantizer's domain (mq_common.h)
  const float r = __fadd_rn(__fsub_rn(rintf(t), t), t);
  return nq_clamp_nan(__fadd_rn(r, o), qmin, qmax);
}
__device__ __forceinline__ float nq_dequant(float q, float s, float o) { return __fmul_rn(__fsub_rn(q, o), s); }

__device__ __forceinline__ float wave_sum_f32(float v) { return wave_sum_f32_dpp(v); }

struct NormArgs {
  const float* x;
  const float* weight;
  const float* bias;
  float eps;
  const float* in_scale;
  const float* in_offset;
  float in_qmin, in_qmax;
  const float* out_scale;
  const float* out_offset;
  float out_qmin, out_qmax;
  float* y;
  int8_t* q_out;
  int8_t* q_tiled;     // same integer image in the fragment-blocked layout of mq_quantize_tiled (nullable)
  int q_shift;
  int32_t* row_sum;
  int64_t rows;
  int cols;
};

// TPR = threads per row: 64 (a wave owns a row, 4 rows per workgroup) for short rows, 256 (a workgroup owns a row,
// reductions through LDS) for cols >= 1024 -- at M = 2048 rows the wave-per-row mapping leaves only 2 waves per SIMD,
// all in the same phase (load, then ~55 VALU ops per element, then store), so nothing overlaps.
// V = float4 vectors held per thread (cols <= 4 * TPR * V); V == 0: the row is re-read instead of kept.
// LN: LayerNorm (QLayerNorm.forward, qmodule.py:624-640 around F.layer_norm) instead of RMSNorm: mean and biased
// variance of the row, y = (xi * rstd + (-rstd * mean)) * gamma + beta -- the expression of torch's CPU kernel.
template <int V, bool LN, int TPR>
__global__ void __launch_bounds__(256) rmsnorm_quant_kernel(const NormArgs a) {
  __shared__ float s_red[3][4];
  __shared__ int s_redi[4];
  const int lane = TPR == 64 ? (threadIdx.x & 63) : threadIdx.x;   // index of this thread inside its row
  const int wv_id = threadIdx.x >> 6;
  const int64_t row = TPR == 64 ? (int64_t)blockIdx.x * 4 + wv_id : (int64_t)blockIdx.x;
  if (TPR == 64 && row >= a.rows) return;      // TPR == 256: grid == rows, and the block-wide reductions need everyone
  auto row_sum_f = [&](float v, int slot) {
    v = wave_sum_f32(v);
    if constexpr (TPR == 64) return v;
    if ((threadIdx.x & 63) == 0) s_red[slot][wv_id] = v;
    __syncthreads();
    return (s_red[slot][0] + s_red[slot][1]) + (s_red[slot][2] + s_red[slot][3]);
  };
  const int cols = a.cols, nvec = cols >> 2;
  const float4* xr = reinterpret_cast<const float4*>(a.x + row * cols);
  const float4* wv = reinterpret_cast<const float4*>(a.weight);
  const float4* bv = reinterpret_cast<const float4*>(a.bias);
  const bool has_in = a.in_scale != nullptr, has_out = a.out_scale != nullptr;
  float si = 1.f, oi = 0.f, so = 1.f, oo = 0.f;
  if (has_in) {
    si = a.in_scale[0];
    oi = a.in_offset[0];
  }
  if (has_out) {
    so = a.out_scale[0];
    oo = a.out_offset[0];
  }
  const float isi = __fdiv_rn(1.0f, si), iso = __fdiv_rn(1.0f, so);
  auto qin = [&](float v) { return has_in ? nq_dequant(nq_index(v, si, isi, oi, a.in_qmin, a.in_qmax), si, oi) : v; };

  constexpr int VV = V > 0 ? V : 1;
  float4 xs[VV];
  float ss = 0.f;
  if constexpr (V > 0) {
#pragma unroll
    for (int k = 0; k < V; ++k) {
      const int i = lane + TPR * k;
      if (i < nvec) {
        float4 v = xr[i];
        v.x = qin(v.x); v.y = qin(v.y); v.z = qin(v.z); v.w = qin(v.w);
        xs[k] = v;
        ss += v.x * v.x;
        ss += v.y * v.y;
        ss += v.z * v.z;
        ss += v.w * v.w;
      }
    }
  } else {
    for (int i = lane; i < nvec; i += TPR) {
      float4 v = xr[i];
      v.x = qin(v.x); v.y = qin(v.y); v.z = qin(v.z); v.w = qin(v.w);
      ss += v.x * v.x;
      ss += v.y * v.y;
      ss += v.z * v.z;
      ss += v.w * v.w;
    }
  }
  float r, shiftv = 0.f;
  if constexpr (LN) {
    // ss holds sum(xi^2) so far; LayerNorm wants sum(xi) and then sum((xi - mean)^2): redo the (register) pass
    float s1 = 0.f;
    if constexpr (V > 0) {
#pragma unroll
      for (int k = 0; k < V; ++k)
        if (lane + TPR * k < nvec) s1 += (xs[k].x + xs[k].y) + (xs[k].z + xs[k].w);
    } else {
      for (int i = lane; i < nvec; i += TPR) {
        float4 v = xr[i];
        s1 += (qin(v.x) + qin(v.y)) + (qin(v.z) + qin(v.w));
      }
    }
    const float mu = __fdiv_rn(row_sum_f(s1, 0), (float)cols);
    float s2 = 0.f;
    auto dev2 = [&](float4 v) {
      const float d0 = v.x - mu, d1 = v.y - mu, d2 = v.z - mu, d3 = v.w - mu;
      s2 += d0 * d0;
      s2 += d1 * d1;
      s2 += d2 * d2;
      s2 += d3 * d3;
    };
    if constexpr (V > 0) {
#pragma unroll
      for (int k = 0; k < V; ++k)
        if (lane + TPR * k < nvec) dev2(xs[k]);
    } else {
      for (int i = lane; i < nvec; i += TPR) {
        float4 v = xr[i];
        v.x = qin(v.x); v.y = qin(v.y); v.z = qin(v.z); v.w = qin(v.w);
        dev2(v);
      }
    }
    const float var = __fdiv_rn(row_sum_f(s2, 1), (float)cols);
    r = __fdiv_rn(1.0f, __fsqrt_rn(__fadd_rn(var, a.eps)));
    shiftv = __fmul_rn(-r, mu);
  } else {
    ss = row_sum_f(ss, 2);
    const float mean = __fdiv_rn(ss, (float)cols);
    r = __fdiv_rn(1.0f, __fsqrt_rn(__fadd_rn(mean, a.eps)));
  }

  int acc = 0;
  auto emit = [&](int i, float4 v) {
    const float4 w = wv[i];
    float y0, y1, y2, y3;
    if constexpr (LN) {
      y0 = __fmul_rn(__fadd_rn(__fmul_rn(v.x, r), shiftv), w.x); y1 = __fmul_rn(__fadd_rn(__fmul_rn(v.y, r), shiftv), w.y);
      y2 = __fmul_rn(__fadd_rn(__fmul_rn(v.z, r), shiftv), w.z); y3 = __fmul_rn(__fadd_rn(__fmul_rn(v.w, r), shiftv), w.w);
    } else {
      y0 = __fmul_rn(w.x, __fmul_rn(v.x, r)); y1 = __fmul_rn(w.y, __fmul_rn(v.y, r));
      y2 = __fmul_rn(w.z, __fmul_rn(v.z, r)); y3 = __fmul_rn(w.w, __fmul_rn(v.w, r));
    }
    if (bv) {
      const float4 b = bv[i];
      y0 = __fadd_rn(y0, b.x); y1 = __fadd_rn(y1, b.y); y2 = __fadd_rn(y2, b.z); y3 = __fadd_rn(y3, b.w);
    }
    if (has_out) {
      const float q0 = nq_index(y0, so, iso, oo, a.out_qmin, a.out_qmax), q1 = nq_index(y1, so, iso, oo, a.out_qmin, a.out_qmax);
      const float q2 = nq_index(y2, so, iso, oo, a.out_qmin, a.out_qmax), q3 = nq_index(y3, so, iso, oo, a.out_qmin, a.out_qmax);
      y0 = nq_dequant(q0, so, oo); y1 = nq_dequant(q1, so, oo); y2 = nq_dequant(q2, so, oo); y3 = nq_dequant(q3, so, oo);
      if (a.q_out || a.q_tiled) {   // NaN has no integer image: saturate to the grid's low end like mq_quantize
        const int s0 = (int)fmaxf(q0, a.out_qmin) - a.q_shift, s1 = (int)fmaxf(q1, a.out_qmin) - a.q_shift;
        const int s2 = (int)fmaxf(q2, a.out_qmin) - a.q_shift, s3 = (int)fmaxf(q3, a.out_qmin) - a.q_shift;
        acc += (s0 + s1) + (s2 + s3);
        const unsigned pk = (unsigned)(s0 & 0xff) | ((unsigned)(s1 & 0xff) << 8) | ((unsigned)(s2 & 0xff) << 16) | ((unsigned)(s3 & 0xff) << 24);
        if (a.q_out) reinterpret_cast<unsigned*>(a.q_out + row * cols)[i] = pk;
        if (a.q_tiled) {            // block (row >> 4, k >> 6); lane (row & 15) + 16 * ((k >> 4) & 3); byte k & 15;  k = 4 i
          const int k = i << 2;
          const int64_t blk = (row >> 4) * (int64_t)(cols >> 6) + (k >> 6);
          *reinterpret_cast<unsigned*>(a.q_tiled + (blk << 10) + ((((int)row & 15) + 16 * ((k >> 4) & 3)) << 4) + (k & 15)) = pk;
        }
      }
    }
    if (a.y) reinterpret_cast<float4*>(a.y + row * cols)[i] = make_float4(y0, y1, y2, y3);
  };
  if constexpr (V > 0) {
#pragma unroll
    for (int k = 0; k < V; ++k) {
      const int i = lane + TPR * k;
      if (i < nvec) emit(i, xs[k]);
    }
  } else {
    for (int i = lane; i < nvec; i += TPR) {
      float4 v = xr[i];
      v.x = qin(v.x); v.y = qin(v.y); v.z = qin(v.z); v.w = qin(v.w);
      emit(i, v);
    }
  }
  if (a.row_sum) {
    acc = wave_sum(acc);
    if constexpr (TPR == 64) {
      if (lane == 0) a.row_sum[row] = acc;
    } else {
      if ((threadIdx.x & 63) == 0) s_redi[wv_id] = acc;
      __syncthreads();
      if (threadIdx.x == 0) a.row_sum[row] = (s_redi[0] + s_redi[1]) + (s_redi[2] + s_redi[3]);
    }
  }
}


// Image-only, fragment-blocked output (the decoder-layer pass: llama.fuse_decoder_layer): the kernel above stores a row's tiled image as
// 16-byte pieces at a 256-byte stride -- 10.5 us at [2048, 2048] against 7.6 us for the row-major image.  Here a workgroup of FOUR
// row groups (256 threads each, the arithmetic of rmsnorm_quant_kernel<V, LN, 256> op for op) owns EIGHT rows, two per group, all
// loads issued up front; the int8 results go to an LDS staging tile in the image's order and leave as 128-byte runs (8 rows x 16 B
// = whole cache lines of a fragment block).
// GRPS: row groups per workgroup (4: eight rows, 1024 threads, one workgroup per CU; 2: four rows, 512 threads, TWO per CU whose load /
// arithmetic / store phases overlap -- mq_norm_tiled_set_rows).
template <int V, bool LN, int GRPS = 4>
__global__ void __launch_bounds__(256 * GRPS) norm_tiled8_kernel(const NormArgs a) {
  constexpr int RW = 2 * GRPS;                                      // rows per workgroup
  extern __shared__ __attribute__((aligned(16))) int8_t stage[];   // [cols / 16 pieces][RW rows][16 B]
  __shared__ float s_red[2][3][GRPS][4];                            // [row of the pair][statistic][group][wave]
  __shared__ int s_redi[2][GRPS][4];
  const int grp = threadIdx.x >> 8, lane = threadIdx.x & 255, wv_id = (threadIdx.x >> 6) & 3;
  const int cols = a.cols, nvec = cols >> 2;
  const int64_t row0 = (int64_t)blockIdx.x * RW;
  const float4* wv = reinterpret_cast<const float4*>(a.weight);
  const float4* bv = reinterpret_cast<const float4*>(a.bias);
  const bool has_in = a.in_scale != nullptr;
  float si = 1.f, oi = 0.f;
  if (has_in) {
    si = a.in_scale[0];
    oi = a.in_offset[0];
  }
  const float so = a.out_scale[0], oo = a.out_offset[0];
  const float isi = __fdiv_rn(1.0f, si), iso = __fdiv_rn(1.0f, so);
  // The arithmetic runs on register PAIRS (v_pk_mul / v_pk_fma / v_pk_add: mq_common.h): the kernel spent ~480 VALU instructions per
  // wave on 16 elements per thread -- as much time as its memory round trip -- and two thirds of them have a packed form with the
  // same bits.  The row statistics keep their element-by-element association (the sums must be those of rmsnorm_quant_kernel).
  // input quantizer, value form.  A NaN / inf element must still poison its row (the reference's clamp propagates NaN): the clamp is a
  // v_med3 (NaN -> qmin) and `probe` = fma(v, 0, probe) turns NaN for such an element; it is added to the row statistic (+ 0.0 otherwise).
  v2f probe2 = {0.f, 0.f};
  auto qin2 = [&](v2f v) -> v2f {
    if (!has_in) return v;
    probe2 = __builtin_elementwise_fma(v, splat2(0.f), probe2);
    const v2f t = div_by_scale2(v, si, isi);
    v2f q = {rintf(t.x), rintf(t.y)};
    q = q + splat2(oi);
    q.x = __builtin_amdgcn_fmed3f(q.x, a.in_qmin, a.in_qmax);
    q.y = __builtin_amdgcn_fmed3f(q.y, a.in_qmin, a.in_qmax);
    return (q - splat2(oi)) * splat2(si);                           // nq_dequant
  };
  const float ubias = (float)(128 - a.q_shift);                    // image_u8f / image_pack4 (mq_common.h)
  float4 xs[2][V];
#pragma unroll
  for (int j = 0; j < 2; ++j) {                                     // every load of both rows goes out before any arithmetic
    const int64_t row = row0 + grp * 2 + j;
    const float4* xr = reinterpret_cast<const float4*>(a.x + (row < a.rows ? row : a.rows - 1) * cols);
#pragma unroll
    for (int k = 0; k < V; ++k) {
      const int i = lane + 256 * k;
      xs[j][k] = xr[i < nvec ? i : nvec - 1];
    }
  }
  float4 wreg[V];
#pragma unroll
  for (int k = 0; k < V; ++k) wreg[k] = wv[lane + 256 * k < nvec ? lane + 256 * k : nvec - 1];
  auto group_sum = [&](float v, int j, int slot) {                  // block-wide barrier: all four groups run the same sequence
    v = wave_sum_f32(v);
    if ((threadIdx.x & 63) == 0) s_red[j][slot][grp][wv_id] = v;
    __syncthreads();
    return (s_red[j][slot][grp][0] + s_red[j][slot][grp][1]) + (s_red[j][slot][grp][2] + s_red[j][slot][grp][3]);
  };
#pragma unroll
  for (int j = 0; j < 2; ++j) {
    const int64_t row = row0 + grp * 2 + j;
    float ss = 0.f;
    probe2 = splat2(0.f);
#pragma unroll
    for (int k = 0; k < V; ++k) {
      if (lane + 256 * k < nvec) {
        const float4 v = xs[j][k];
        const v2f lo = qin2((v2f){v.x, v.y}), hi = qin2((v2f){v.z, v.w});
        xs[j][k] = make_float4(lo.x, lo.y, hi.x, hi.y);
        const v2f sl = lo * lo, sh = hi * hi;
        ss += sl.x;
        ss += sl.y;
        ss += sh.x;
        ss += sh.y;
      }
    }
    const float probe = probe2.x + probe2.y;                        // 0 or NaN
    float r, shiftv = 0.f;
    if constexpr (LN) {
      float s1 = 0.f;
#pragma unroll
      for (int k = 0; k < V; ++k)
        if (lane + 256 * k < nvec) s1 += (xs[j][k].x + xs[j][k].y) + (xs[j][k].z + xs[j][k].w);
      s1 += probe;
      const float mu = __fdiv_rn(group_sum(s1, j, 0), (float)cols);
      float s2 = 0.f;
#pragma unroll
      for (int k = 0; k < V; ++k)
        if (lane + 256 * k < nvec) {
          const float4 v = xs[j][k];
          const v2f d01 = (v2f){v.x, v.y} - splat2(mu), d23 = (v2f){v.z, v.w} - splat2(mu);
          const v2f q01 = d01 * d01, q23 = d23 * d23;
          s2 += q01.x;
          s2 += q01.y;
          s2 += q23.x;
          s2 += q23.y;
        }
      const float var = __fdiv_rn(group_sum(s2, j, 1), (float)cols);
      r = __fdiv_rn(1.0f, __fsqrt_rn(__fadd_rn(var, a.eps)));
      shiftv = __fmul_rn(-r, mu);
    } else {
      ss = group_sum(ss + probe, j, 2);
      const float mean = __fdiv_rn(ss, (float)cols);
      r = __fdiv_rn(1.0f, __fsqrt_rn(__fadd_rn(mean, a.eps)));
    }
    uint32_t usum = 0;
#pragma unroll
    for (int k = 0; k < V; ++k) {
      const int i = lane + 256 * k;
      if (i < nvec) {
        const float4 v = xs[j][k], w = wreg[k];
        v2f y01 = (v2f){v.x, v.y} * splat2(r), y23 = (v2f){v.z, v.w} * splat2(r);
        if constexpr (LN) {
          y01 = (y01 + splat2(shiftv)) * (v2f){w.x, w.y};
          y23 = (y23 + splat2(shiftv)) * (v2f){w.z, w.w};
        } else {
          y01 = (v2f){w.x, w.y} * y01;
          y23 = (v2f){w.z, w.w} * y23;
        }
        if (bv) {
          const float4 b = bv[i];
          y01 = y01 + (v2f){b.x, b.y};
          y23 = y23 + (v2f){b.z, b.w};
        }
        const v2f u01 = image_u8f2(y01, so, iso, oo, a.out_qmin, a.out_qmax, ubias), u23 = image_u8f2(y23, so, iso, oo, a.out_qmin, a.out_qmax, ubias);
        const uint32_t pk = image_pack4(u01.x, u01.y, u23.x, u23.y, usum);
        // staging: piece (k >> 4) = 16-byte chunk column, then the row of the eight, then the byte:  k = 4 i
        *reinterpret_cast<unsigned*>(stage + (i >> 2) * (RW * 16) + ((grp * 2 + j) << 4) + ((i & 3) << 2)) = pk;
      }
    }
    if (a.row_sum) {
      const int acc = wave_sum((int)usum);
      if ((threadIdx.x & 63) == 0) s_redi[j][grp][wv_id] = acc;
    }
  }
  __syncthreads();                                                  // the staging tile and the row-sum partials are complete
  if (a.row_sum && threadIdx.x < RW) {
    const int g = threadIdx.x >> 1, j = threadIdx.x & 1;
    if (row0 + threadIdx.x < a.rows) a.row_sum[row0 + threadIdx.x] = (s_redi[j][g][0] + s_redi[j][g][1]) + (s_redi[j][g][2] + s_redi[j][g][3]) - 128 * cols;
  }
  // copy-out: 16-byte unit p = 8 piece + row;  piece = 4 kb + kq  ->  block (row0 >> 4, kb), byte 256 kq + 16 ((row0 & 15) + row)
  const int units = (cols >> 4) * RW;                               // RW rows x cols / 16
  const int64_t rb = row0 >> 4;
  const int half = (int)(row0 & 15);
  for (int p = threadIdx.x; p < units; p += 256 * GRPS) {
    const int piece = p / RW, r8 = p % RW;
    if (row0 + r8 < a.rows)
      *reinterpret_cast<uint4*>(a.q_tiled + ((rb * (cols >> 6) + (piece >> 2)) << 10) + ((piece & 3) << 8) + ((half + r8) << 4)) =
          *reinterpret_cast<const uint4*>(stage + (p << 4));
  }
}

}  // namespace mq

using namespace mq;

// rows per workgroup of the image-only tiled norm: 0 = by shape (four rows -- two 512-thread workgroups per CU, whose load / arithmetic /
// store phases overlap -- up to 2048 columns: 6.9 -> 6.2 us at [2048, 2048]; eight beyond, where it is a wash), 4 / 8 force one (A/B timing)
static std::atomic<int> g_norm_tiled_rows{0};
extern "C" int mq_norm_tiled_set_rows(int rows) {
  g_norm_tiled_rows = rows == 4 ? 4 : (rows == 8 ? 8 : 0);
  return 0;
}

static int launch_norm(const char* fn, bool ln, const float* x, int64_t rows, int64_t cols, const float* weight, const float* bias,
                       float eps, const float* in_scale, const float* in_offset, float in_qmin, float in_qmax,
                       const float* out_scale, const float* out_offset, float out_qmin, float out_qmax, float* y,
                       int8_t* q_out, int8_t* q_tiled, int q_shift, int32_t* row_sum, mq_stream_t stream) {
  if (rows == 0) return MQ_OK;                   // empty activation (its data pointers may be NULL)
  MQ_REQUIRE(x && weight && (y || q_out || q_tiled), "%s: null pointer", fn);
  MQ_REQUIRE(!q_tiled || (cols % 64 == 0 && aligned(q_tiled, 16)), "%s: the fragment-blocked output needs cols %% 64 == 0 and a 16-byte aligned buffer", fn);
  MQ_REQUIRE(rows >= 0 && cols > 0 && cols % 4 == 0 && cols <= (1 << 20) && rows < (int64_t)0x7fffffff,
             "%s: bad shape %lld x %lld (cols must be a multiple of 4)", fn, (long long)rows, (long long)cols);
  MQ_REQUIRE((in_scale == nullptr) == (in_offset == nullptr) && (out_scale == nullptr) == (out_offset == nullptr),
             "%s: scale/offset must both be set or NULL", fn);
  MQ_REQUIRE(!(q_out || q_tiled) || out_scale, "%s: integer output needs an output quantizer", fn);
  MQ_REQUIRE(!row_sum || q_out || q_tiled, "%s: row sums are those of the integer output", fn);
  MQ_REQUIRE(!(q_out || q_tiled) || (out_qmin - (float)q_shift >= -128.f && out_qmax - (float)q_shift <= 127.f),
             "%s: [%g,%g]-%d does not fit int8", fn, out_qmin, out_qmax, q_shift);
  MQ_REQUIRE(aligned(x, 16) && aligned(weight, 16) && (!bias || aligned(bias, 16)) && (!y || aligned(y, 16)) &&
                 (!q_out || aligned(q_out, 4)),
             "%s: pointers must be 16-byte aligned", fn);
  if (rows == 0) return MQ_OK;
  NormArgs a{x, weight, bias, eps, in_scale, in_offset, in_qmin, in_qmax, out_scale, out_offset, out_qmin, out_qmax,
             y, q_out, q_tiled, q_shift, row_sum, rows, (int)cols};
  hipStream_t st = as_stream(stream);
  // image-only, fragment-blocked: eight rows per workgroup, stores as whole lines of the image (norm_tiled8_kernel)
  if (q_tiled && !y && !q_out && out_scale && cols >= 1024 && cols <= 4096 && cols % 64 == 0 && rows >= 64) {
    const int rows_knob = g_norm_tiled_rows.load();
    const bool four = rows_knob == 4 || (rows_knob == 0 && cols <= 2048);
    const unsigned grid = four ? (unsigned)((rows + 3) / 4) : (unsigned)((rows + 7) / 8);
    const size_t lds = (size_t)cols * (four ? 4 : 8);
#define MQ_NORM8(V)                                                                   \
    do {                                                                              \
      if (four) {                                                                     \
        if (ln) norm_tiled8_kernel<V, true, 2><<<grid, 512, lds, st>>>(a);            \
        else norm_tiled8_kernel<V, false, 2><<<grid, 512, lds, st>>>(a);              \
      } else if (ln) norm_tiled8_kernel<V, true><<<grid, 1024, lds, st>>>(a);         \
      else norm_tiled8_kernel<V, false><<<grid, 1024, lds, st>>>(a);                  \
    } while (0)
    if (cols <= 1024) MQ_NORM8(1);
    else if (cols <= 2048) MQ_NORM8(2);
    else MQ_NORM8(4);
#undef MQ_NORM8
    MQ_LAUNCH_CHECK(fn);
    return MQ_OK;
  }
#define MQ_NORM(V, TPR)                                                                                   \
  do {                                                                                                    \
    const unsigned grid = (TPR) == 64 ? (unsigned)((rows + 3) / 4) : (unsigned)rows;                      \
    if (ln) rmsnorm_quant_kernel<V, true, TPR><<<grid, 256, 0, st>>>(a);                                  \
    else rmsnorm_quant_kernel<V, false, TPR><<<grid, 256, 0, st>>>(a);                                    \
  } while (0)
  if (cols < 1024) MQ_NORM(4, 64);                      // short rows: a wave per row
  else if (cols <= 1024) MQ_NORM(1, 256);               // a workgroup per row from here on
  else if (cols <= 2048) MQ_NORM(2, 256);
  else if (cols <= 4096) MQ_NORM(4, 256);
  else if (cols <= 8192) MQ_NORM(8, 256);
  else MQ_NORM(0, 256);
#undef MQ_NORM
  MQ_LAUNCH_CHECK(fn);
  return MQ_OK;
}

extern "C" int mq_rmsnorm_quant(const float* x, int64_t rows, int64_t cols, const float* weight, const float* bias, float eps,
                                const float* in_scale, const float* in_offset, float in_qmin, float in_qmax,
                                const float* out_scale, const float* out_offset, float out_qmin, float out_qmax, float* y,
                                int8_t* q_out, int8_t* q_tiled, int q_shift, int32_t* row_sum, mq_stream_t stream) {
  return launch_norm("mq_rmsnorm_quant", false, x, rows, cols, weight, bias, eps, in_scale, in_offset, in_qmin, in_qmax, out_scale,
                     out_offset, out_qmin, out_qmax, y, q_out, q_tiled, q_shift, row_sum, stream);
}

extern "C" int mq_layernorm_quant(const float* x, int64_t rows, int64_t cols, const float* weight, const float* bias, float eps,
                                  const float* in_scale, const float* in_offset, float in_qmin, float in_qmax,
                                  const float* out_scale, const float* out_offset, float out_qmin, float out_qmax, float* y,
                                  int8_t* q_out, int8_t* q_tiled, int q_shift, int32_t* row_sum, mq_stream_t stream) {
  return launch_norm("mq_layernorm_quant", true, x, rows, cols, weight, bias, eps, in_scale, in_offset, in_qmin, in_qmax, out_scale,
                     out_offset, out_qmin, out_qmax, y, q_out, q_tiled, q_shift, row_sum, stream);
}

// ---- QSiLU / QGELU.forward in one pass (qmodule.py:739-754, :790-798) -------------------------------------------
// SiLU:  xi = Qin(x);  g = Qmid(sigmoid(xi));  out = Qout(xi * g)       (Qmid: the [0,1] sigmoid grid, qmodule.py:731-734)
// GELU:  xi = Qin(x);  out = Qout(0.5 * xi * (1 + erf(xi / sqrt 2)))
// = 4 (2) launches and 9 (5) passes as composite ops.  exp / erf are the device library's (<= 1-2 ulp), the divide of the
// sigmoid is IEEE: results equal torch's GPU sigmoid / gelu bit for bit and the CPU reference's up to those ulps, i.e.
// after the output quantizer at most one LSB apart on a vanishing fraction of elements.
namespace mq {

struct ActArgs {
  const float* x;
  float* y;
  int64_t numel;
  int act;   // 0 = SiLU, 1 = GELU (erf)
  const float* s[3];   // in / mid / out scale (nullable)
  const float* o[3];
  float qmin[3], qmax[3];
};

__global__ void __launch_bounds__(256) act_quant_kernel(const ActArgs a) {
  float sc[3], of[3], isc[3];
  bool has[3];
#pragma unroll
  for (int k = 0; k < 3; ++k) {
    has[k] = a.s[k] != nullptr;
    sc[k] = has[k] ? a.s[k][0] : 1.f;
    of[k] = has[k] ? a.o[k][0] : 0.f;
    isc[k] = __fdiv_rn(1.0f, sc[k]);
  }
  auto fq = [&](int k, float v) { return has[k] ? nq_dequant(nq_index(v, sc[k], isc[k], of[k], a.qmin[k], a.qmax[k]), sc[k], of[k]) : v; };
  auto f = [&](float v) {
    const float xi = fq(0, v);
    float r;
    if (a.act == 0) {
      const float g = __fdiv_rn(1.0f, __fadd_rn(1.0f, expf(-xi)));
      r = __fmul_rn(xi, fq(1, g));
    } else {
      r = __fmul_rn(__fmul_rn(0.5f, xi), __fadd_rn(1.0f, erff(__fmul_rn(xi, 0.70710678118654752440f))));
    }
    return fq(2, r);
  };
  const int64_t nvec = a.numel >> 2;
  const int64_t stride = (int64_t)gridDim.x * 256;
  for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < nvec; i += stride) {
    float4 v = reinterpret_cast<const float4*>(a.x)[i];
    v.x = f(v.x); v.y = f(v.y); v.z = f(v.z); v.w = f(v.w);
    reinterpret_cast<float4*>(a.y)[i] = v;
  }
  for (int64_t i = (nvec << 2) + (int64_t)blockIdx.x * 256 + threadIdx.x; i < a.numel; i += stride) a.y[i] = f(a.x[i]);
}


// ---- f1: the gated FFN's  act(w1(x)) * w3(x)  -> integer input image of w2, ONE launch ------------------------------------------
// Reference chain (hf_model.py:1057, qmodule.py:739-753): y1 = Qact(va * Qmid(sigmoid(va))) (QSiLU; QGELU: Qact(gelu(va))), the
// plain fp32 product p = y1 * vb (ElementwiseMul is not quantised), then w2's input quantizer.  va / vb arrive either as fp32 values
// or -- the integer chain -- as the 8-bit output INDICES the w1 / w3 GEMMs wrote (va = (qa - oa) * sa: exactly the fp32 value the
// fake-quant path would hold).  Output: int8 storage (index - shift) of p on w2's input grid + row sums (what mq_quantize would
// produce from p), optionally p itself.  3 B per element instead of 17 for the composite chain.  Wave per row, 16 elements per lane.
struct GatedArgs {
  const void* a;
  const void* b;
  int in_index;                // 0: fp32 values, 1: u8 indices
  int64_t rows, cols;
  int act;
  const float* s[5];           // a grid, b grid, mid (sigmoid) grid, activation output grid, w2 input grid
  const float* o[5];
  float qmin[5], qmax[5];
  int shift;
  int8_t* q;
  int32_t* row_sum;
  float* y;
};

template <bool INDEX, bool WRITE_Y>
__global__ void __launch_bounds__(256) gated_act_quant_kernel(const GatedArgs g) {
  float sc[5], of[5], isc[5];
  bool has[5];
#pragma unroll
  for (int k = 0; k < 5; ++k) {
    has[k] = g.s[k] != nullptr;
    sc[k] = has[k] ? g.s[k][0] : 1.f;
    of[k] = has[k] ? g.o[k][0] : 0.f;
    isc[k] = __fdiv_rn(1.0f, sc[k]);
  }
  auto fq = [&](int k, float v) { return has[k] ? nq_dequant(nq_index(v, sc[k], isc[k], of[k], g.qmin[k], g.qmax[k]), sc[k], of[k]) : v; };
  // y1 = Qact(act(x)) of one gate input value
  auto gate_of = [&](float xi) {
    float r;
    if (g.act == 0) {
      const float gate = __fdiv_rn(1.0f, __fadd_rn(1.0f, expf(-xi)));
      r = __fmul_rn(xi, fq(2, gate));
    } else {
      r = __fmul_rn(__fmul_rn(0.5f, xi), __fadd_rn(1.0f, erff(__fmul_rn(xi, 0.70710678118654752440f))));
    }
    return fq(3, r);
  };
  // Index inputs take only 256 values each: the whole activation chain (exp, three exact divides) is evaluated ONCE per index into
  // LDS -- the same arithmetic on the same operands, so the result is bit-identical to evaluating it per element -- and the
  // per-element work is two table reads, the product and w2's input quantizer.  (45 -> ~10 us at [2048, 5632].)
  __shared__ float lut[2][256];
  if constexpr (INDEX) {
    lut[0][threadIdx.x] = gate_of(nq_dequant((float)threadIdx.x, sc[0], of[0]));
    lut[1][threadIdx.x] = nq_dequant((float)threadIdx.x, sc[1], of[1]);
    __syncthreads();
  }
  const int lane = threadIdx.x & 63;
  const int64_t wave0 = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6);
  const int64_t nwaves = (int64_t)gridDim.x * 4;
  for (int64_t row = wave0; row < g.rows; row += nwaves) {
    int acc = 0;
    for (int64_t c = (int64_t)lane * 16; c < g.cols; c += 1024) {
      float y1[16], vb[16];
      const int64_t at = row * g.cols + c;
      if constexpr (INDEX) {
        const uint4 pa = *reinterpret_cast<const uint4*>(reinterpret_cast<const uint8_t*>(g.a) + at);
        const uint4 pb = *reinterpret_cast<const uint4*>(reinterpret_cast<const uint8_t*>(g.b) + at);
        const uint32_t wa[4] = {pa.x, pa.y, pa.z, pa.w}, wb[4] = {pb.x, pb.y, pb.z, pb.w};
#pragma unroll
        for (int e = 0; e < 16; ++e) {
          y1[e] = lut[0][(wa[e >> 2] >> (8 * (e & 3))) & 0xffu];
          vb[e] = lut[1][(wb[e >> 2] >> (8 * (e & 3))) & 0xffu];
        }
      } else {
        const float4* pa = reinterpret_cast<const float4*>(reinterpret_cast<const float*>(g.a) + at);
        const float4* pb = reinterpret_cast<const float4*>(reinterpret_cast<const float*>(g.b) + at);
#pragma unroll
        for (int d = 0; d < 4; ++d) {
          const float4 x = pa[d], z = pb[d];
          y1[4 * d] = gate_of(x.x); y1[4 * d + 1] = gate_of(x.y); y1[4 * d + 2] = gate_of(x.z); y1[4 * d + 3] = gate_of(x.w);
          vb[4 * d] = z.x; vb[4 * d + 1] = z.y; vb[4 * d + 2] = z.z; vb[4 * d + 3] = z.w;
        }
      }
      float p[16];
      uint32_t w[4];
#pragma unroll
      for (int d = 0; d < 4; ++d) {
        uint32_t pk = 0;
#pragma unroll
        for (int e = 0; e < 4; ++e) {
          const float prod = __fmul_rn(y1[4 * d + e], vb[4 * d + e]);
          p[4 * d + e] = prod;
          const float qi = nq_index(prod, sc[4], isc[4], of[4], g.qmin[4], g.qmax[4]);
          // integer storage has no NaN: saturate like mq_quantize does
          const int st_v = (qi != qi ? (int)g.qmin[4] : (int)qi) - g.shift;
          acc += st_v;
          pk |= ((uint32_t)st_v & 0xffu) << (8 * e);
        }
        w[d] = pk;
      }
      *reinterpret_cast<uint4*>(g.q + at) = make_uint4(w[0], w[1], w[2], w[3]);
      if constexpr (WRITE_Y) {
        float4* py = reinterpret_cast<float4*>(g.y + at);
#pragma unroll
        for (int d = 0; d < 4; ++d) py[d] = make_float4(p[4 * d], p[4 * d + 1], p[4 * d + 2], p[4 * d + 3]);
      }
    }
    if (g.row_sum != nullptr) {
      acc = wave_sum(acc);
      if (lane == 0) g.row_sum[row] = acc;
    }
  }
}

// Index inputs without the fp32 side output (the integer chain of fuse_gated_mlp): a WORKGROUP per row and 8 elements per thread
// and trip, so that [2048, 5632] puts 8 waves on every SIMD instead of 2 -- the kernel is a latency-bound stream (two LDS table reads
// and one exact divide per element), occupancy is what it lacks.  Same tables, same per-element arithmetic as the kernel above.
__global__ void __launch_bounds__(256) gated_index_rows_kernel(const GatedArgs g) {
  __shared__ float lut[2][256];
  __shared__ int s_part[4];
  const float so = g.s[4][0], oo = g.o[4][0];
  const float iso = __fdiv_rn(1.0f, so);
  {
    float sc[4], of[4], isc[4];
    bool has[4];
#pragma unroll
    for (int k = 0; k < 4; ++k) {
      has[k] = g.s[k] != nullptr;
      sc[k] = has[k] ? g.s[k][0] : 1.f;
      of[k] = has[k] ? g.o[k][0] : 0.f;
      isc[k] = __fdiv_rn(1.0f, sc[k]);
    }
    auto fq = [&](int k, float v) { return has[k] ? nq_dequant(nq_index(v, sc[k], isc[k], of[k], g.qmin[k], g.qmax[k]), sc[k], of[k]) : v; };
    const float xi = nq_dequant((float)threadIdx.x, sc[0], of[0]);
    float r;
    if (g.act == 0) {
      const float gate = __fdiv_rn(1.0f, __fadd_rn(1.0f, expf(-xi)));
      r = __fmul_rn(xi, fq(2, gate));
    } else {
      r = __fmul_rn(__fmul_rn(0.5f, xi), __fadd_rn(1.0f, erff(__fmul_rn(xi, 0.70710678118654752440f))));
    }
    lut[0][threadIdx.x] = fq(3, r);
    lut[1][threadIdx.x] = nq_dequant((float)threadIdx.x, sc[1], of[1]);
  }
  __syncthreads();
  const int64_t row = blockIdx.x;
  const uint8_t* pa = reinterpret_cast<const uint8_t*>(g.a) + row * g.cols;
  const uint8_t* pb = reinterpret_cast<const uint8_t*>(g.b) + row * g.cols;
  int8_t* pq = g.q + row * g.cols;
  int acc = 0;
  for (int64_t c = (int64_t)threadIdx.x * 8; c < g.cols; c += 2048) {
    const uint2 va = *reinterpret_cast<const uint2*>(pa + c), vb = *reinterpret_cast<const uint2*>(pb + c);
    const uint32_t wa[2] = {va.x, va.y}, wb[2] = {vb.x, vb.y};
    uint32_t w[2];
#pragma unroll
    for (int d = 0; d < 2; ++d) {
      uint32_t pk = 0;
#pragma unroll
      for (int e = 0; e < 4; ++e) {
        const float prod = __fmul_rn(lut[0][(wa[d] >> (8 * e)) & 0xffu], lut[1][(wb[d] >> (8 * e)) & 0xffu]);
        const float qi = nq_index(prod, so, iso, oo, g.qmin[4], g.qmax[4]);
        const int st_v = (qi != qi ? (int)g.qmin[4] : (int)qi) - g.shift;
        acc += st_v;
        pk |= ((uint32_t)st_v & 0xffu) << (8 * e);
      }
      w[d] = pk;
    }
    *reinterpret_cast<uint2*>(pq + c) = make_uint2(w[0], w[1]);
  }
  if (g.row_sum != nullptr) {
    acc = wave_sum(acc);
    if ((threadIdx.x & 63) == 0) s_part[threadIdx.x >> 6] = acc;
    __syncthreads();
    if (threadIdx.x == 0) g.row_sum[row] = (s_part[0] + s_part[1]) + (s_part[2] + s_part[3]);
  }
}

// ---- the same map as a 256 x 256 table --------------------------------------------------------------------------------------------
// With index inputs and static grids, act(a) * b -> w2's input index is a FUNCTION of the two 8-bit indices: 65 536 values, computed
// once per set of grids (mq_gated_table: the per-element arithmetic of the kernels above, evaluated for every (ia, ib) pair -- the
// table IS that arithmetic, so results are bit-identical) and then looked up (mq_gated_lookup): one LDS byte read per element instead
// of two table reads, a multiply and an IEEE divide.  The lookup kernel is a pure stream (2 B in, 1 B out per element).
__global__ void __launch_bounds__(256) gated_table_kernel(const GatedArgs g, int8_t* __restrict__ table) {
  float sc[5], of[5], isc[5];
  bool has[5];
#pragma unroll
  for (int k = 0; k < 5; ++k) {
    has[k] = g.s[k] != nullptr;
    sc[k] = has[k] ? g.s[k][0] : 1.f;
    of[k] = has[k] ? g.o[k][0] : 0.f;
    isc[k] = __fdiv_rn(1.0f, sc[k]);
  }
  auto fq = [&](int k, float v) { return has[k] ? nq_dequant(nq_index(v, sc[k], isc[k], of[k], g.qmin[k], g.qmax[k]), sc[k], of[k]) : v; };
  const int ia = blockIdx.x, ib = threadIdx.x;
  const float xi = nq_dequant((float)ia, sc[0], of[0]);
  float r;
  if (g.act == 0) {
    const float gate = __fdiv_rn(1.0f, __fadd_rn(1.0f, expf(-xi)));
    r = __fmul_rn(xi, fq(2, gate));
  } else {
    r = __fmul_rn(__fmul_rn(0.5f, xi), __fadd_rn(1.0f, erff(__fmul_rn(xi, 0.70710678118654752440f))));
  }
  const float prod = __fmul_rn(fq(3, r), nq_dequant((float)ib, sc[1], of[1]));
  const float qi = nq_index(prod, sc[4], isc[4], of[4], g.qmin[4], g.qmax[4]);
  table[ia * 256 + ib] = (int8_t)((qi != qi ? (int)g.qmin[4] : (int)qi) - g.shift);
}

constexpr int GL_TABLE = 65536;
// 1024 threads = four groups of four waves; a group owns one row at a time (rows strided by 4 * gridDim), requests the whole row up
// front (<= 4 x 8 bytes per lane and operand), then looks up.  Two such workgroups are resident per CU (2 x 64 KiB of LDS): 8 waves
// per SIMD, and at [2048, 5632] every group handles exactly one row -- the kernel is one round of loads, lookups and stores.
// TILED: q is the fragment-blocked image of mq_quantize_tiled (1-KiB blocks of 16 rows x 64 k; a lane's eight bytes stay inside one
// 16-byte fragment chunk, and the four rows of a workgroup's trip -- rows 4 n .. 4 n + 3 -- fill whole 64-byte pieces of a block).
template <bool TILED>
__global__ void __launch_bounds__(1024) gated_lookup_kernel(const uint8_t* __restrict__ a, const uint8_t* __restrict__ b, int64_t rows, int64_t cols,
                                                            const int8_t* __restrict__ table, int8_t* __restrict__ q, int32_t* __restrict__ row_sum) {
  auto dst = [&](int64_t row, int64_t c) -> int8_t* {
    if constexpr (TILED) return q + (((row >> 4) * (cols >> 6) + (c >> 6)) << 10) + ((((int)row & 15) + 16 * (((int)c >> 4) & 3)) << 4) + ((int)c & 15);
    else return q + row * cols + c;
  };
  extern __shared__ __attribute__((aligned(16))) int8_t lut[];      // [256][256]
  __shared__ int s_sum[4];
  const int grp = threadIdx.x >> 8, tid = threadIdx.x & 255;
  constexpr int MAXIT = 4;                                          // cols <= 8192 on the fast path (host-checked); longer rows loop
  const int64_t row0 = (int64_t)blockIdx.x * 4 + grp;
  uint2 va[MAXIT], vb[MAXIT];
  const bool fast = cols <= 2048 * MAXIT;
  if (fast && row0 < rows) {                                        // first row's operands go out before the table copy
#pragma unroll
    for (int it = 0; it < MAXIT; ++it) {
      const int64_t c = (int64_t)tid * 8 + 2048 * it;
      if (c < cols) {
        va[it] = *reinterpret_cast<const uint2*>(a + row0 * cols + c);
        vb[it] = *reinterpret_cast<const uint2*>(b + row0 * cols + c);
      }
    }
  }
  for (int i = threadIdx.x; i < GL_TABLE / 16; i += 1024) reinterpret_cast<uint4*>(lut)[i] = reinterpret_cast<const uint4*>(table)[i];
  __syncthreads();
  auto convert = [&](uint2 x, uint2 y, int& acc) {
    const uint32_t wa[2] = {x.x, x.y}, wb[2] = {y.x, y.y};
    uint32_t w[2];
#pragma unroll
    for (int d = 0; d < 2; ++d) {
      uint32_t pk = 0;
#pragma unroll
      for (int e = 0; e < 4; ++e) {
        const int st_v = lut[(((wa[d] >> (8 * e)) & 0xffu) << 8) | ((wb[d] >> (8 * e)) & 0xffu)];
        acc += st_v;
        pk |= ((uint32_t)st_v & 0xffu) << (8 * e);
      }
      w[d] = pk;
    }
    return make_uint2(w[0], w[1]);
  };
  const int64_t stride = (int64_t)gridDim.x * 4;
  for (int64_t base = (int64_t)blockIdx.x * 4; base < rows; base += stride) {      // uniform trip count for the whole workgroup
    const int64_t row = base + grp;
    int acc = 0;
    if (row < rows) {
      if (fast) {
        if (base != (int64_t)blockIdx.x * 4) {
#pragma unroll
          for (int it = 0; it < MAXIT; ++it) {
            const int64_t c = (int64_t)tid * 8 + 2048 * it;
            if (c < cols) {
              va[it] = *reinterpret_cast<const uint2*>(a + row * cols + c);
              vb[it] = *reinterpret_cast<const uint2*>(b + row * cols + c);
            }
          }
        }
#pragma unroll
        for (int it = 0; it < MAXIT; ++it) {
          const int64_t c = (int64_t)tid * 8 + 2048 * it;
          if (c < cols) *reinterpret_cast<uint2*>(dst(row, c)) = convert(va[it], vb[it], acc);
        }
      } else {
        for (int64_t c = (int64_t)tid * 8; c < cols; c += 2048)
          *reinterpret_cast<uint2*>(dst(row, c)) =
              convert(*reinterpret_cast<const uint2*>(a + row * cols + c), *reinterpret_cast<const uint2*>(b + row * cols + c), acc);
      }
    }
    if (row_sum != nullptr) {
      if (tid < 1) s_sum[grp] = 0;
      __syncthreads();
      acc = wave_sum(acc);
      if ((threadIdx.x & 63) == 0) atomicAdd(&s_sum[grp], acc);
      __syncthreads();
      if (tid == 0 && row < rows) row_sum[row] = s_sum[grp];
    }
  }
}

}  // namespace mq

extern "C" int mq_gated_table(int act, const float* a_scale, const float* a_offset, const float* b_scale, const float* b_offset,
                              const float* mid_scale, const float* mid_offset, float mid_qmin, float mid_qmax, const float* act_scale,
                              const float* act_offset, float act_qmin, float act_qmax, const float* out_scale, const float* out_offset,
                              float out_qmin, float out_qmax, int q_shift, int8_t* table, mq_stream_t stream) {
  using namespace mq;
  MQ_REQUIRE((act == 0 || act == 1) && a_scale && a_offset && b_scale && b_offset && out_scale && out_offset && table,
             "mq_gated_table: null pointer / bad act (0 SiLU, 1 GELU)");
  MQ_REQUIRE((mid_scale == nullptr) == (mid_offset == nullptr) && (act_scale == nullptr) == (act_offset == nullptr),
             "mq_gated_table: scale/offset must both be set or NULL");
  MQ_REQUIRE(out_qmin - (float)q_shift >= -128.f && out_qmax - (float)q_shift <= 127.f, "mq_gated_table: output grid does not fit int8");
  GatedArgs g{nullptr, nullptr, 1, 0, 0, act, {a_scale, b_scale, mid_scale, act_scale, out_scale}, {a_offset, b_offset, mid_offset, act_offset, out_offset},
              {0.f, 0.f, mid_qmin, act_qmin, out_qmin}, {0.f, 0.f, mid_qmax, act_qmax, out_qmax}, q_shift, nullptr, nullptr, nullptr};
  gated_table_kernel<<<256, 256, 0, as_stream(stream)>>>(g, table);
  MQ_LAUNCH_CHECK("mq_gated_table");
  return MQ_OK;
}

static int gated_lookup_launch(const char* fn, bool tiled, const uint8_t* a, const uint8_t* b, int64_t rows, int64_t cols, const int8_t* table,
                               int8_t* q_out, int32_t* row_sum, mq_stream_t stream) {
  using namespace mq;
  MQ_REQUIRE(rows >= 0 && cols >= 0 && cols % (tiled ? 64 : 8) == 0, "%s: cols %% %d == 0", fn, tiled ? 64 : 8);
  if (rows == 0 || cols == 0) return MQ_OK;
  MQ_REQUIRE(a && b && table && q_out && aligned(a, 8) && aligned(b, 8) && aligned(q_out, tiled ? 16 : 8) && aligned(table, 16),
             "%s: null or misaligned pointer", fn);
  static PerDeviceOnce attr_set[2];
  const int dev = current_device();
  if (!attr_set[tiled].done(dev)) {
    hipError_t e = hipFuncSetAttribute(tiled ? (const void*)gated_lookup_kernel<true> : (const void*)gated_lookup_kernel<false>,
                                       hipFuncAttributeMaxDynamicSharedMemorySize, GL_TABLE);
    if (e != hipSuccess) {
      set_error("%s: hipFuncSetAttribute: %s", fn, hipGetErrorString(e));
      return MQ_EHIP;
    }
    attr_set[tiled].mark(dev);
  }
  int64_t blocks = (rows + 3) / 4;                          // four rows per workgroup and trip; two resident workgroups per CU
  if (blocks > 512) blocks = 512;
  if (tiled) gated_lookup_kernel<true><<<(unsigned)blocks, 1024, GL_TABLE, as_stream(stream)>>>(a, b, rows, cols, table, q_out, row_sum);
  else gated_lookup_kernel<false><<<(unsigned)blocks, 1024, GL_TABLE, as_stream(stream)>>>(a, b, rows, cols, table, q_out, row_sum);
  MQ_LAUNCH_CHECK(fn);
  return MQ_OK;
}

extern "C" int mq_gated_lookup(const uint8_t* a, const uint8_t* b, int64_t rows, int64_t cols, const int8_t* table, int8_t* q_out,
                               int32_t* row_sum, mq_stream_t stream) {
  return gated_lookup_launch("mq_gated_lookup", false, a, b, rows, cols, table, q_out, row_sum, stream);
}

extern "C" int mq_gated_lookup_tiled(const uint8_t* a, const uint8_t* b, int64_t rows, int64_t cols, const int8_t* table, int8_t* q_tiled,
                                     int32_t* row_sum, mq_stream_t stream) {
  return gated_lookup_launch("mq_gated_lookup_tiled", true, a, b, rows, cols, table, q_tiled, row_sum, stream);
}

extern "C" int mq_gated_act_quant(const void* a, const void* b, int in_dtype, int64_t rows, int64_t cols, int act,
                                  const float* a_scale, const float* a_offset, const float* b_scale, const float* b_offset,
                                  const float* mid_scale, const float* mid_offset, float mid_qmin, float mid_qmax,
                                  const float* act_scale, const float* act_offset, float act_qmin, float act_qmax,
                                  const float* out_scale, const float* out_offset, float out_qmin, float out_qmax, int q_shift,
                                  int8_t* q_out, int32_t* row_sum, float* y, mq_stream_t stream) {
  using namespace mq;
  MQ_REQUIRE(rows >= 0 && cols >= 0 && cols % 16 == 0 && (act == 0 || act == 1),
             "mq_gated_act_quant: bad arguments (cols %% 16 == 0; act = 0 SiLU, 1 GELU)");
  if (rows == 0 || cols == 0) return MQ_OK;
  MQ_REQUIRE(a && b && q_out && out_scale && out_offset, "mq_gated_act_quant: null pointer (a, b, q_out and the output grid are required)");
  MQ_REQUIRE(in_dtype == MQ_F32 || in_dtype == MQ_U8, "mq_gated_act_quant: inputs are float32 values or uint8 indices");
  MQ_REQUIRE(in_dtype == MQ_F32 || (a_scale && a_offset && b_scale && b_offset), "mq_gated_act_quant: index inputs need their grids");
  MQ_REQUIRE(aligned(a, 16) && aligned(b, 16) && aligned(q_out, 16) && (!y || aligned(y, 16)), "mq_gated_act_quant: pointers must be 16-byte aligned");
  MQ_REQUIRE((mid_scale == nullptr) == (mid_offset == nullptr) && (act_scale == nullptr) == (act_offset == nullptr),
             "mq_gated_act_quant: scale/offset must both be set or NULL");
  MQ_REQUIRE(out_qmin - (float)q_shift >= -128.f && out_qmax - (float)q_shift <= 127.f, "mq_gated_act_quant: output grid does not fit int8");
  const bool idx = in_dtype == MQ_U8;
  GatedArgs g{a, b, idx ? 1 : 0, rows, cols, act,
              {idx ? a_scale : nullptr, idx ? b_scale : nullptr, mid_scale, act_scale, out_scale},
              {idx ? a_offset : nullptr, idx ? b_offset : nullptr, mid_offset, act_offset, out_offset},
              {0.f, 0.f, mid_qmin, act_qmin, out_qmin}, {0.f, 0.f, mid_qmax, act_qmax, out_qmax}, q_shift, q_out, row_sum, y};
  int64_t blocks = (rows + 3) / 4;
  if (blocks > 256 * 16) blocks = 256 * 16;
  hipStream_t st = as_stream(stream);
  if (idx && !y && cols % 8 == 0 && rows < (int64_t)0x7fffffff && aligned(a, 8) && aligned(b, 8)) {
    gated_index_rows_kernel<<<(unsigned)rows, 256, 0, st>>>(g);
  } else if (idx) {
    if (y) gated_act_quant_kernel<true, true><<<(unsigned)blocks, 256, 0, st>>>(g);
    else gated_act_quant_kernel<true, false><<<(unsigned)blocks, 256, 0, st>>>(g);
  } else {
    if (y) gated_act_quant_kernel<false, true><<<(unsigned)blocks, 256, 0, st>>>(g);
    else gated_act_quant_kernel<false, false><<<(unsigned)blocks, 256, 0, st>>>(g);
  }
  MQ_LAUNCH_CHECK("mq_gated_act_quant");
  return MQ_OK;
}

namespace mq {
}  // namespace mq

extern "C" int mq_act_quant(const float* x, int64_t numel, int act, const float* in_scale, const float* in_offset, float in_qmin,
                            float in_qmax, const float* mid_scale, const float* mid_offset, float mid_qmin, float mid_qmax,
                            const float* out_scale, const float* out_offset, float out_qmin, float out_qmax, float* y,
                            mq_stream_t stream) {
  MQ_REQUIRE(numel >= 0 && (act == 0 || act == 1), "mq_act_quant: bad arguments (act = 0 SiLU, 1 GELU)");
  if (numel == 0) return MQ_OK;
  MQ_REQUIRE(x && y && aligned(x, 16) && aligned(y, 16), "mq_act_quant: x / y must be non-null and 16-byte aligned");
  MQ_REQUIRE((in_scale == nullptr) == (in_offset == nullptr) && (mid_scale == nullptr) == (mid_offset == nullptr) &&
                 (out_scale == nullptr) == (out_offset == nullptr),
             "mq_act_quant: scale/offset must both be set or NULL");
  ActArgs a{x, y, numel, act, {in_scale, mid_scale, out_scale}, {in_offset, mid_offset, out_offset},
            {in_qmin, mid_qmin, out_qmin}, {in_qmax, mid_qmax, out_qmax}};
  int64_t g = ((numel >> 2) + 255) / 256;
  if (g < 1) g = 1;
  if (g > 256 * 8) g = 256 * 8;
  act_quant_kernel<<<(unsigned)g, 256, 0, as_stream(stream)>>>(a);
  MQ_LAUNCH_CHECK("mq_act_quant");
  return MQ_OK;
}

// Elementwise kernels of the simulated-quant forward (HBM-bound): scale/offset from min/max,
// fake-quant, quantize-to-integer (+ row sums), GEMM epilogue-vector preparation, W4 packing.
//
// Bit-exactness contract (DESIGN.md "Numerics"): every kernel here evaluates the reference's fp32
// expression tree op for op -- IEEE division (never reciprocal-multiply), round-half-even
// (v_rndne_f32), separate add / clamp / subtract / multiply, no FMA contraction -- so the integer
// indices equal the reference CPU path's bit for bit.  Build flags: -ffp-contract=off, no fast-math,
// -fhip-fp32-correctly-rounded-divide-sqrt (hipcc default, stated explicitly in build.py).
#include <hip/hip_fp16.h>

#include <type_traits>

#include "mq_common.h"

#pragma clang fp contract(off)

namespace mq {

static thread_local char g_err[512] = "";

void set_error(const char* fmt, ...) {
  va_list ap;
  va_start(ap, fmt);
  vsnprintf(g_err, sizeof(g_err), fmt, ap);
  va_end(ap);
}

// ---- the reference's scalar expression tree ---------------------------------------------------
// qmodule.py:286-287
__device__ __forceinline__ float q_index(float x, float s, float inv_s, float o, float qmin, float qmax, bool fast = true) {
  float t = div_by_scale_guarded(x, s, inv_s, fast);
  float r = rintf(t);
  float q = __fadd_rn(r, o);
  return fminf(fmaxf(q, qmin), qmax);          // NaN saturates to qmin: integer storage has no NaN
}
// torch.clamp propagates NaN (v_min/v_max drop it): the float-valued kernels follow the reference there
__device__ __forceinline__ float clamp_nan(float q, float lo, float hi) {
  const float c = fminf(fmaxf(q, lo), hi);
  return q != q ? q : c;
}
// round_ste (qmodule.py:17-21) is (round(t) - t) + t: exact for finite t, NaN for t = +-inf (inf - inf)
__device__ __forceinline__ float round_ste(float t) { return __fadd_rn(__fsub_rn(rintf(t), t), t); }
__device__ __forceinline__ float q_index_fq(float x, float s, float inv_s, float o, float qmin, float qmax, bool fast = true) {
  return clamp_nan(__fadd_rn(round_ste(div_by_scale_guarded(x, s, inv_s, fast)), o), qmin, qmax);
}
// qmodule.py:290
__device__ __forceinline__ float q_dequant(float q, float s, float o) { return __fmul_rn(__fsub_rn(q, o), s); }

// fp16 tensor with 0-dim fp32 scale/offset: result rounded to half after every op (SURVEY 8a' item 4)
__device__ __forceinline__ float h_round(float v) { return __half2float(__float2half_rn(v)); }
__device__ __forceinline__ float q_index_hmath(float x, float s, float inv_s, float o, float qmin, float qmax, bool fast = true) {
  float t = h_round(div_by_scale_guarded(x, s, inv_s, fast));
  float r = h_round(__fadd_rn(h_round(__fsub_rn(h_round(rintf(t)), t)), t));   // round_ste, one half rounding per op
  float q = h_round(__fadd_rn(r, o));
  return clamp_nan(q, qmin, qmax);      // qmin/qmax are exactly representable in half for <= 8 bits
}
__device__ __forceinline__ float q_dequant_hmath(float q, float s, float o) {
  return h_round(__fmul_rn(h_round(__fsub_rn(q, o)), s));
}

// ---- a1 -----------------------------------------------------------------------------------------
__global__ void scale_offset_kernel(const float* __restrict__ mn, const float* __restrict__ mx, int64_t n,
                                    float qmax, int symmetric, float* __restrict__ scale,
                                    float* __restrict__ offset) {
  int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  float lo = mn[i], hi = mx[i];
  float alpha, beta;
  if (symmetric) {
    alpha = (lo != lo || hi != hi) ? __fadd_rn(lo, hi) : fmaxf(fabsf(lo), fabsf(hi));   // torch.max propagates NaN
    beta = 0.0f;
  } else {
    alpha = __fsub_rn(hi, lo);
    beta = lo;
  }
  float s = __fdiv_rn(alpha, qmax);
  s = clamp_nan(s, 1e-5f, 1e6f);               // qmodule.py:58 (CLIPMIN / CLIPMAX)
  scale[i] = s;
  offset[i] = -rintf(__fdiv_rn(beta, s));      // qmodule.py:60 ; symmetric -> -0.0f
}

typedef float vf4 __attribute__((ext_vector_type(4)));

// ---- a5: fake-quant, vectorised 16 B per lane ---------------------------------------------------
template <typename T>
struct Vec16;
template <>
struct Vec16<float> {
  static constexpr int N = 4;
  float v[4];
  __device__ static float get(const Vec16& a, int i) { return a.v[i]; }
  __device__ static void set(Vec16& a, int i, float f) { a.v[i] = f; }
};
template <>
struct Vec16<__half> {
  static constexpr int N = 8;
  __half v[8];
  __device__ static float get(const Vec16& a, int i) { return __half2float(a.v[i]); }
  __device__ static void set(Vec16& a, int i, float f) { a.v[i] = __float2half_rn(f); }
};

template <typename T>
__device__ __forceinline__ float ld(const T* p, int64_t i);
template <>
__device__ __forceinline__ float ld<float>(const float* p, int64_t i) { return p[i]; }
template <>
__device__ __forceinline__ float ld<__half>(const __half* p, int64_t i) { return __half2float(p[i]); }
template <typename T>
__device__ __forceinline__ void st(T* p, int64_t i, float v);
template <>
__device__ __forceinline__ void st<float>(float* p, int64_t i, float v) { p[i] = v; }
template <>
__device__ __forceinline__ void st<__half>(__half* p, int64_t i, float v) { p[i] = __float2half_rn(v); }

template <typename T, bool PER_ROW, bool HMATH>
__global__ void __launch_bounds__(256) fake_quant_vec_kernel(const T* __restrict__ x, T* __restrict__ y,
                                                             int64_t nvec, uint32_t cols,
                                                             const float* __restrict__ scale,
                                                             const float* __restrict__ offset, float qmin,
                                                             float qmax) {
  using V = Vec16<T>;
  const int64_t stride = (int64_t)gridDim.x * blockDim.x;
  float s = 0.f, o = 0.f, inv_s = 0.f;
  bool fast = true;                               // mq_common.h: IEEE divide for a scale outside the fast form's range
  if (!PER_ROW) {
    s = scale[0];
    o = offset[0];
    inv_s = __fdiv_rn(1.0f, s);
    fast = scale_in_fast_range(s);
  }
  const V* xv = reinterpret_cast<const V*>(x);
  V* yv = reinterpret_cast<V*>(y);
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < nvec; i += stride) {
    V a = xv[i];
    if (PER_ROW) {
      int64_t row = (i * V::N) / cols;          // cols % V::N == 0: a vector never straddles rows
      s = scale[row];
      o = offset[row];
      inv_s = __fdiv_rn(1.0f, s);
      fast = scale_in_fast_range(s);
    }
    V r;
#pragma unroll
    for (int j = 0; j < V::N; ++j) {
      float f = V::get(a, j);
      float q = HMATH ? q_index_hmath(f, s, inv_s, o, qmin, qmax, fast) : q_index_fq(f, s, inv_s, o, qmin, qmax, fast);
      V::set(r, j, HMATH ? q_dequant_hmath(q, s, o) : q_dequant(q, s, o));
    }
    yv[i] = r;
  }
}

template <typename T, bool PER_ROW, bool HMATH>
__global__ void __launch_bounds__(256) fake_quant_scalar_kernel(const T* __restrict__ x, T* __restrict__ y,
                                                                int64_t numel, int64_t cols,
                                                                const float* __restrict__ scale,
                                                                const float* __restrict__ offset, float qmin,
                                                                float qmax) {
  const int64_t stride = (int64_t)gridDim.x * blockDim.x;
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < numel; i += stride) {
    int64_t row = PER_ROW ? i / cols : 0;
    float s = scale[row], o = offset[row];
    const float inv_s = __fdiv_rn(1.0f, s);
    const bool fast = scale_in_fast_range(s);
    float f = ld<T>(x, i);
    float q = HMATH ? q_index_hmath(f, s, inv_s, o, qmin, qmax, fast) : q_index_fq(f, s, inv_s, o, qmin, qmax, fast);
    st<T>(y, i, HMATH ? q_dequant_hmath(q, s, o) : q_dequant(q, s, o));
  }
}

// ---- f3: learnable weight clipping + per-row fake-quant of a weight in ONE pass per direction -------------------------------
// The e2equant / omniquant inner step (algorithm.py:187-233, :381, :587) re-derives every weight's grid from the weight itself on
// every forward: amin / amax per output row (qmodule.py:263-268), sigmoid(bound factor) * range (:271-273), scale / offset (:40-61),
// fake-quant (:286-290) -- as modules: a row reduction, ~12 [rows, 1]-sized launches, the fake-quant pass, and in the backward the
// per-row STE pass, ~25 [rows, 1]-sized launches and EIGHT weight-sized passes that scatter the range gradients into the extreme
// elements.  Here a workgroup owns a row, holds it in registers and does all of it: one read + one write of the weight forward, two
// reads + one write backward.  Same fp32 operations in the same order as the module chain (scale_offset_kernel, q_index_fq,
// fake_quant_bwd_kernel above), so forward values are bit-identical; the row sums of the backward associate differently (float4 per
// thread) -- inside every gradient tolerance the goldens use.
template <int VPT>
struct LwcRow {
  vf4 v[VPT];
};

__device__ __forceinline__ float block_reduce4(float v, float* s_part, int slot) {      // 256 threads; every thread gets the result
  const int w = threadIdx.x >> 6;
  if ((threadIdx.x & 63) == 0) s_part[slot * 4 + w] = v;
  __syncthreads();
  return (s_part[slot * 4 + 0] + s_part[slot * 4 + 1]) + (s_part[slot * 4 + 2] + s_part[slot * 4 + 3]);
}

struct LwcGrid {
  float lo, hi, scale, offset, alpha;
};
// sigmoid(bound) * range -> grid: qmodule.py:271-273 then :40-61 (scale_offset_kernel's expression)
__device__ __forceinline__ LwcGrid lwc_grid(float mn, float mx, float sig_lo, float sig_hi, float qmax, int symmetric) {
  LwcGrid g;
  g.lo = __fmul_rn(sig_lo, mn);
  g.hi = __fmul_rn(sig_hi, mx);
  float beta;
  if (symmetric) {
    g.alpha = (g.lo != g.lo || g.hi != g.hi) ? __fadd_rn(g.lo, g.hi) : fmaxf(fabsf(g.lo), fabsf(g.hi));
    beta = 0.0f;
  } else {
    g.alpha = __fsub_rn(g.hi, g.lo);
    beta = g.lo;
  }
  g.scale = clamp_nan(__fdiv_rn(g.alpha, qmax), 1e-5f, 1e6f);
  g.offset = -rintf(__fdiv_rn(beta, g.scale));
  return g;
}

template <int VPT>
__global__ void __launch_bounds__(256) lwc_fake_quant_kernel(const float* __restrict__ w, int cols, const float* __restrict__ sig_lo,
                                                             const float* __restrict__ sig_hi, float qmin, float qmax, int symmetric,
                                                             float* __restrict__ out, float* __restrict__ row_min,
                                                             float* __restrict__ row_max, float* __restrict__ scale,
                                                             float* __restrict__ offset) {
  __shared__ float s_mn[4], s_mx[4];
  __shared__ int s_nan[4];
  const int64_t row = blockIdx.x;
  const int nvec = cols >> 2;
  const vf4* wr = reinterpret_cast<const vf4*>(w + row * cols);
  LwcRow<VPT> r;
#pragma unroll
  for (int k = 0; k < VPT; ++k) {
    const int i = threadIdx.x + 256 * k;
    if (i < nvec) r.v[k] = __builtin_nontemporal_load(wr + i);
  }
  float mn = INFINITY, mx = -INFINITY;
  int nan = 0;
#pragma unroll
  for (int k = 0; k < VPT; ++k) {
    if (threadIdx.x + 256 * k < nvec) {
#pragma unroll
      for (int e = 0; e < 4; ++e) {
        const float x = r.v[k][e];
        mn = fminf(mn, x);
        mx = fmaxf(mx, x);
        nan |= (x != x);
      }
    }
  }
  mn = wave_min(mn);
  mx = wave_max(mx);
  nan = __builtin_amdgcn_ballot_w64(nan != 0) != 0;
  const int wv = threadIdx.x >> 6;
  if ((threadIdx.x & 63) == 0) {
    s_mn[wv] = mn;
    s_mx[wv] = mx;
    s_nan[wv] = nan;
  }
  __syncthreads();
  mn = fminf(fminf(s_mn[0], s_mn[1]), fminf(s_mn[2], s_mn[3]));
  mx = fmaxf(fmaxf(s_mx[0], s_mx[1]), fmaxf(s_mx[2], s_mx[3]));
  if (s_nan[0] | s_nan[1] | s_nan[2] | s_nan[3]) mn = mx = __builtin_nanf("");         // amin / amax propagate NaN
  const LwcGrid g = lwc_grid(mn, mx, sig_lo[row], sig_hi[row], qmax, symmetric);
  if (threadIdx.x == 0) {
    row_min[row] = mn;
    row_max[row] = mx;
    scale[row] = g.scale;
    offset[row] = g.offset;
  }
  const float inv_s = __fdiv_rn(1.0f, g.scale);
  const bool fast = scale_in_fast_range(g.scale);
  vf4* orow = reinterpret_cast<vf4*>(out + row * cols);
#pragma unroll
  for (int k = 0; k < VPT; ++k) {
    const int i = threadIdx.x + 256 * k;
    if (i < nvec) {
      vf4 y;
#pragma unroll
      for (int e = 0; e < 4; ++e) y[e] = q_dequant(q_index_fq(r.v[k][e], g.scale, inv_s, g.offset, qmin, qmax, fast), g.scale, g.offset);
      orow[i] = y;                                     // read back by the GEMM that follows: a normal store
    }
  }
}

// Backward: grad_w = STE-masked grad_out + the range gradients scattered into the row's extreme elements (ties share evenly: what
// torch's amin / amax backward does); grad_sig_lo / grad_sig_hi [rows] for sigmoid(lowbound_factor) / sigmoid(upbound_factor).
// The offset is -round(beta / scale): no gradient reaches the range through it (torch.round has none).
template <int VPT>
__global__ void __launch_bounds__(256) lwc_fake_quant_bwd_kernel(const float* __restrict__ w, const float* __restrict__ gy, int cols,
                                                                 const float* __restrict__ sig_lo, const float* __restrict__ sig_hi,
                                                                 const float* __restrict__ row_min, const float* __restrict__ row_max,
                                                                 float qmin, float qmax, int symmetric, float* __restrict__ gw,
                                                                 float* __restrict__ g_sig_lo, float* __restrict__ g_sig_hi) {
  __shared__ float s_part[12];
  const int64_t row = blockIdx.x;
  const int nvec = cols >> 2;
  const vf4* wr = reinterpret_cast<const vf4*>(w + row * cols);
  const vf4* gr = reinterpret_cast<const vf4*>(gy + row * cols);
  LwcRow<VPT> x, g;
#pragma unroll
  for (int k = 0; k < VPT; ++k) {
    const int i = threadIdx.x + 256 * k;
    if (i < nvec) {
      x.v[k] = __builtin_nontemporal_load(wr + i);
      g.v[k] = __builtin_nontemporal_load(gr + i);
    }
  }
  const float mn = row_min[row], mx = row_max[row], slo = sig_lo[row], shi = sig_hi[row];
  const LwcGrid gd = lwc_grid(mn, mx, slo, shi, qmax, symmetric);
  const float s = gd.scale, o = gd.offset;
  float acc_s = 0.f, n_mn = 0.f, n_mx = 0.f;
#pragma unroll
  for (int k = 0; k < VPT; ++k) {
    if (threadIdx.x + 256 * k < nvec) {
#pragma unroll
      for (int e = 0; e < 4; ++e) {
        const float xv = x.v[k][e], gv = g.v[k][e];
        const float t = __fdiv_rn(xv, s);
        const float r = round_ste(t);
        const float q = __fadd_rn(r, o);
        const bool inside = q >= qmin && q <= qmax;
        const float qc = clamp_nan(q, qmin, qmax);
        acc_s += inside ? gv * (r - t) : gv * (qc - o);
        g.v[k][e] = inside ? __fdiv_rn(__fmul_rn(gv, s), s) : 0.f;
        n_mn += (xv == mn) ? 1.f : 0.f;
        n_mx += (xv == mx) ? 1.f : 0.f;
      }
    }
  }
  const float g_scale = block_reduce4(wave_sum_f32_dpp(acc_s), s_part, 0);
  n_mn = block_reduce4(wave_sum_f32_dpp(n_mn), s_part, 1);
  n_mx = block_reduce4(wave_sum_f32_dpp(n_mx), s_part, 2);
  // scale = clamp(alpha / qmax): the clamp passes the gradient inside [CLIPMIN, CLIPMAX] (bounds included, as torch.clamp does)
  const float raw = __fdiv_rn(gd.alpha, qmax);
  const float g_alpha = (raw >= 1e-5f && raw <= 1e6f) ? __fdiv_rn(g_scale, qmax) : 0.f;
  float g_lo, g_hi;
  if (symmetric) {                                     // alpha = maximum(|lo|, |hi|): the larger takes it, a tie halves it; d|v| = sign(v)
    const float alo = fabsf(gd.lo), ahi = fabsf(gd.hi);
    const float share_lo = alo > ahi ? 1.f : (alo == ahi ? 0.5f : 0.f), share_hi = ahi > alo ? 1.f : (alo == ahi ? 0.5f : 0.f);
    const float sg_lo = gd.lo > 0.f ? 1.f : (gd.lo < 0.f ? -1.f : 0.f), sg_hi = gd.hi > 0.f ? 1.f : (gd.hi < 0.f ? -1.f : 0.f);
    g_lo = __fmul_rn(__fmul_rn(g_alpha, share_lo), sg_lo);
    g_hi = __fmul_rn(__fmul_rn(g_alpha, share_hi), sg_hi);
  } else {                                             // alpha = hi - lo
    g_lo = -g_alpha;
    g_hi = g_alpha;
  }
  if (threadIdx.x == 0) {
    g_sig_lo[row] = __fmul_rn(g_lo, mn);               // lo = sig_lo * mn
    g_sig_hi[row] = __fmul_rn(g_hi, mx);
  }
  const float add_mn = __fdiv_rn(__fmul_rn(g_lo, slo), n_mn), add_mx = __fdiv_rn(__fmul_rn(g_hi, shi), n_mx);
  vf4* orow = reinterpret_cast<vf4*>(gw + row * cols);
#pragma unroll
  for (int k = 0; k < VPT; ++k) {
    const int i = threadIdx.x + 256 * k;
    if (i < nvec) {
      vf4 y;
#pragma unroll
      for (int e = 0; e < 4; ++e) {
        const float xv = x.v[k][e];
        const float scat = __fadd_rn(xv == mn ? add_mn : 0.f, xv == mx ? add_mx : 0.f);   // is_mn * (g_mn / n_mn) + is_mx * (g_mx / n_mx)
        y[e] = __fadd_rn(g.v[k][e], scat);
      }
      orow[i] = y;
    }
  }
}

// ---- f3: the [heads, S, S] chain of a training-mode attention block in one pass per direction --------------------------------
// HFAttention.forward (hf_model.py:511-520) between the two matmuls, as the PTQ training loops run it (algorithm.py:381 / :587):
//   qk_bmm's output quantizer (16-bit, learnable grid) -> / sqrt(d) -> + mask -> softmax (fp32) -> pv_bmm's input quantizer (16-bit)
// is five score-sized launches forward (each reads and writes 0.5 GB at S = 2048, 32 heads) and five backward, with four
// score-sized tensors kept for autograd.  A WAVE owns a row (up to 4096 keys in registers), reads the raw scores once and writes the
// quantised probabilities once; the backward re-derives everything from the raw scores and the incoming gradient: two reads, one
// write, and the four grid gradients as one atomic each per workgroup.  Per-element expressions are those of fake_quant_vec_kernel /
// fake_quant_bwd_kernel and of torch's softmax (max, expf(a - max), sum, e / sum; backward p * (g - sum(g p))); the row sums associate
// as a wave reduction (torch's own order differs between its CPU and GPU kernels as well).
struct AttnProbsArgs {
  const float* raw;        // [rows, cols] q.k^T before the output quantizer
  const float* mask;       // [mask_rows, cols] additive (row r uses mask row r % mask_rows), or NULL
  const float* s1; const float* o1; const float* s2; const float* o2;
  float qmin1, qmax1, qmin2, qmax2, sqrt_d;
  int64_t rows;
  int cols, mask_rows;
};

struct FqPoint {            // the fake-quant expression tree at one element (fake_quant_bwd_kernel's names)
  float t, r, qc, y;
  bool inside;
};

// FAST: both scales inside the reciprocal form's range (mq_common.h scale_in_fast_range) and sqrt(d) a power of two -- decided ONCE per
// kernel (a wave-uniform branch at the top; a guard per element would put a branch between every two instructions): x / s through
// div_by_scale (the IEEE quotient's bits for |t| >= 2^-2, within an ulp below, where r = 0 and t's last bit only enters the scale
// gradient's g * (r - t), a relative 6e-8 of a term summed in unspecified order), x / sqrt(d) as an exact multiplication, and the
// straight-through (g * s) / s within an ulp of the IEEE quotient.  !FAST: IEEE divides throughout.  (The per-tensor / per-row
// backward kernels above keep the IEEE divide: they are the ones quantizer_grads.npz pins bit for bit.)
template <bool FAST>
__device__ __forceinline__ float uquot(float x, float s, float inv_s) { return FAST ? div_by_scale(x, s, inv_s) : __fdiv_rn(x, s); }

template <bool FAST>
__device__ __forceinline__ FqPoint fq_point(float x, float s, float inv_s, float o, float qmin, float qmax) {
  FqPoint f;
  f.t = uquot<FAST>(x, s, inv_s);
  f.r = round_ste(f.t);
  const float q = __fadd_rn(f.r, o);
  f.inside = q >= qmin && q <= qmax;
  f.qc = clamp_nan(q, qmin, qmax);
  f.y = q_dequant(f.qc, s, o);
  return f;
}
__device__ __forceinline__ bool is_pow2f(float v) { return (__float_as_uint(v) & 0x007FFFFFu) == 0u; }

// A lane holds elements 4 * (lane + TPR k) + e of the row, k < VPT (TPR threads per row: a wave, or the whole workgroup); lanes past
// the row's end hold clamped duplicates that are excluded from the statistics by selects and never stored (no divergent control flow
// around the arithmetic).
template <int VPT, int TPR = 64>
__device__ __forceinline__ void load_row(const float* __restrict__ base, int lane, int nvec, float (&v)[VPT * 4]) {
  const vf4* r = reinterpret_cast<const vf4*>(base);
#pragma unroll
  for (int k = 0; k < VPT; ++k) {
    const int i = lane + TPR * k;
    const vf4 t = __builtin_nontemporal_load(r + (i < nvec ? i : nvec - 1));
    v[4 * k] = t.x; v[4 * k + 1] = t.y; v[4 * k + 2] = t.z; v[4 * k + 3] = t.w;
  }
}

struct AttnGrids {
  float s1, o1, s2, o2, is1, is2, isd;
};
__device__ __forceinline__ AttnGrids attn_grids(const AttnProbsArgs& a, bool& fast) {
  AttnGrids g;
  g.s1 = a.s1[0]; g.o1 = a.o1[0]; g.s2 = a.s2[0]; g.o2 = a.o2[0];
  g.is1 = __fdiv_rn(1.0f, g.s1); g.is2 = __fdiv_rn(1.0f, g.s2); g.isd = __fdiv_rn(1.0f, a.sqrt_d);
  fast = scale_in_fast_range(g.s1) && scale_in_fast_range(g.s2) && is_pow2f(a.sqrt_d);
  return g;
}

// scores of one row after Q1, / sqrt(d), + mask -> v[]; returns the lane's running maximum
template <int VPT, bool FAST, int TPR = 64>
__device__ __forceinline__ float attn_logits(const AttnProbsArgs& a, const AttnGrids& g, int64_t row, int lane, int nvec, float (&v)[VPT * 4]) {
  const vf4* mr = a.mask ? reinterpret_cast<const vf4*>(a.mask + (row % a.mask_rows) * a.cols) : nullptr;
  float mx = -INFINITY;
#pragma unroll
  for (int k = 0; k < VPT; ++k) {
    const int i = lane + TPR * k;
    const bool valid = i < nvec;
    vf4 m = {0.f, 0.f, 0.f, 0.f};
    if (mr) m = mr[valid ? i : nvec - 1];
#pragma unroll
    for (int e = 0; e < 4; ++e) {
      const float y = fq_point<FAST>(v[4 * k + e], g.s1, g.is1, g.o1, a.qmin1, a.qmax1).y;
      float x = FAST ? __fmul_rn(y, g.isd) : __fdiv_rn(y, a.sqrt_d);
      if (mr) x = __fadd_rn(x, m[e]);
      v[4 * k + e] = x;
      mx = fmaxf(mx, valid ? x : -INFINITY);
    }
  }
  return mx;
}

template <int VPT, int TPR = 64>
__device__ __forceinline__ float attn_exp(float mx, int lane, int nvec, float (&v)[VPT * 4]) {
  float l = 0.f;
#pragma unroll
  for (int k = 0; k < VPT; ++k) {
    const bool valid = lane + TPR * k < nvec;
#pragma unroll
    for (int e = 0; e < 4; ++e) {
      const float ex = expf(__fsub_rn(v[4 * k + e], mx));
      v[4 * k + e] = ex;
      l += valid ? ex : 0.f;
    }
  }
  return l;
}

template <int VPT, bool FAST>
__device__ __forceinline__ void attn_probs_fwd_body(const AttnProbsArgs& a, const AttnGrids& g, float* __restrict__ out) {
  const int lane = threadIdx.x & 63, nvec = a.cols >> 2;
  const int64_t nw = (int64_t)gridDim.x * 4;
  for (int64_t row = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6); row < a.rows; row += nw) {
    float v[VPT * 4];
    load_row<VPT>(a.raw + row * a.cols, lane, nvec, v);
    const float mx = wave_max(attn_logits<VPT, FAST>(a, g, row, lane, nvec, v));
    const float l = wave_sum_f32_dpp(attn_exp<VPT>(mx, lane, nvec, v));
    vf4* orow = reinterpret_cast<vf4*>(out + row * a.cols);
#pragma unroll
    for (int k = 0; k < VPT; ++k) {
      vf4 y;
#pragma unroll
      for (int e = 0; e < 4; ++e) y[e] = fq_point<FAST>(__fdiv_rn(v[4 * k + e], l), g.s2, g.is2, g.o2, a.qmin2, a.qmax2).y;
      if (lane + 64 * k < nvec) orow[lane + 64 * k] = y;
    }
  }
}

template <int VPT>
__global__ void __launch_bounds__(256) attn_probs_fwd_kernel(const AttnProbsArgs a, float* __restrict__ out) {
  bool fast;
  const AttnGrids g = attn_grids(a, fast);
  if (fast) attn_probs_fwd_body<VPT, true>(a, g, out);
  else attn_probs_fwd_body<VPT, false>(a, g, out);
}

// Row reductions of the backward: TPR = 64 -> the wave's; TPR = 256 -> the four waves of the workgroup meet through the LDS (slot =
// reduction number, double-buffered over the row loop: a slot is rewritten two iterations, i.e. at least three barriers, later).
template <int TPR, class Op>
__device__ __forceinline__ float row_reduce(float v, Op op, float* s_red, int slot) {
  v = wave_reduce_f(v, op);
  if (TPR == 64) return v;
  const int w = threadIdx.x >> 6;
  if ((threadIdx.x & 63) == 0) s_red[slot * 4 + w] = v;
  __syncthreads();
  return op(op(s_red[slot * 4 + 0], s_red[slot * 4 + 1]), op(s_red[slot * 4 + 2], s_red[slot * 4 + 3]));
}

// The backward keeps three values per element (raw score, incoming gradient, probability): with a wave per 2048-key row that is 96
// registers of state plus the division chains of 32 elements in flight -- hipcc allocates 256 VGPRs + 142 AGPRs (one wave per SIMD,
// 0.9 ms).  Rows of more than 512 keys are therefore spread over the WHOLE workgroup (8 elements per thread at 2048 keys).
template <int VPT, bool FAST, int TPR>
__device__ __forceinline__ void attn_probs_bwd_body(const AttnProbsArgs& a, const AttnGrids& gd, const float* __restrict__ gy,
                                                    float* __restrict__ graw, float (&acc)[4], float* s_red) {
  constexpr int RPW = 256 / TPR;                      // rows per workgroup and iteration
  const int lane = threadIdx.x & (TPR - 1), nvec = a.cols >> 2;
  const float s1 = gd.s1, o1 = gd.o1, s2 = gd.s2, o2 = gd.o2;
  const int64_t nr = (int64_t)gridDim.x * RPW;
  const int64_t iters = (a.rows + nr - 1) / nr;       // the same trip count for every thread (barriers inside when TPR = 256)
  auto fmax_op = [](float x, float y) { return fmaxf(x, y); };
  auto add_op = [](float x, float y) { return x + y; };
  for (int64_t it = 0; it < iters; ++it) {
    const int64_t row_raw = it * nr + (int64_t)blockIdx.x * RPW + threadIdx.x / TPR;
    const bool live = row_raw < a.rows;
    const int64_t row = live ? row_raw : a.rows - 1;
    float* red = s_red + (it & 1) * 12;
    float x[VPT * 4], g[VPT * 4], p[VPT * 4];
    load_row<VPT, TPR>(a.raw + row * a.cols, lane, nvec, x);
    load_row<VPT, TPR>(gy + row * a.cols, lane, nvec, g);
#pragma unroll
    for (int i = 0; i < VPT * 4; ++i) p[i] = x[i];
    const float mx = row_reduce<TPR>(attn_logits<VPT, FAST, TPR>(a, gd, row, lane, nvec, p), fmax_op, red, 0);
    const float l = row_reduce<TPR>(attn_exp<VPT, TPR>(mx, lane, nvec, p), add_op, red, 1);
    float dot = 0.f;
#pragma unroll
    for (int k = 0; k < VPT; ++k) {
      const bool valid = live && lane + TPR * k < nvec;
#pragma unroll
      for (int e = 0; e < 4; ++e) {
        const float pv = __fdiv_rn(p[4 * k + e], l), gv = valid ? g[4 * k + e] : 0.f;
        const FqPoint f = fq_point<FAST>(pv, s2, gd.is2, o2, a.qmin2, a.qmax2);
        acc[2] += f.inside ? gv * (f.r - f.t) : gv * (f.qc - o2);
        acc[3] += f.inside ? 0.f : -gv * s2;
        const float gp = f.inside ? uquot<FAST>(__fmul_rn(gv, s2), s2, gd.is2) : 0.f;     // (g * s) / s, as autograd chains it
        p[4 * k + e] = pv;
        g[4 * k + e] = gp;
        dot += gp * pv;
      }
    }
    dot = row_reduce<TPR>(dot, add_op, red, 2);
    vf4* orow = reinterpret_cast<vf4*>(graw + row * a.cols);
#pragma unroll
    for (int k = 0; k < VPT; ++k) {
      const bool valid = live && lane + TPR * k < nvec;
      vf4 y;
#pragma unroll
      for (int e = 0; e < 4; ++e) {
        const float ga = valid ? __fmul_rn(p[4 * k + e], __fsub_rn(g[4 * k + e], dot)) : 0.f;       // softmax backward: p * (g - sum(g p))
        const float gs = FAST ? __fmul_rn(ga, gd.isd) : __fdiv_rn(ga, a.sqrt_d);
        const FqPoint f = fq_point<FAST>(x[4 * k + e], s1, gd.is1, o1, a.qmin1, a.qmax1);
        acc[0] += f.inside ? gs * (f.r - f.t) : gs * (f.qc - o1);
        acc[1] += f.inside ? 0.f : -gs * s1;
        y[e] = f.inside ? uquot<FAST>(__fmul_rn(gs, s1), s1, gd.is1) : 0.f;
      }
      if (valid) __builtin_nontemporal_store(y, orow + lane + TPR * k);
    }
  }
}

template <int VPT, int TPR>
__global__ void __launch_bounds__(256) attn_probs_bwd_kernel(const AttnProbsArgs a, const float* __restrict__ gy, float* __restrict__ graw,
                                                             float* __restrict__ ggrid /* [4]: d s1, d o1, d s2, d o2; zero-initialised */) {
  __shared__ float s_acc[4][4];
  __shared__ float s_red[24];
  bool fast;
  const AttnGrids gd = attn_grids(a, fast);
  float acc[4] = {0.f, 0.f, 0.f, 0.f};
  if (fast) attn_probs_bwd_body<VPT, true, TPR>(a, gd, gy, graw, acc, s_red);
  else attn_probs_bwd_body<VPT, false, TPR>(a, gd, gy, graw, acc, s_red);
  const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    const float v = wave_sum_f32_dpp(acc[i]);
    if (lane == 0) s_acc[i][w] = v;
  }
  __syncthreads();
  if (threadIdx.x < 4) atomicAdd(ggrid + threadIdx.x, (s_acc[threadIdx.x][0] + s_acc[threadIdx.x][1]) + (s_acc[threadIdx.x][2] + s_acc[threadIdx.x][3]));
}

// ---- quantize to integers, one workgroup per row, optional row sum ------------------------------
template <typename QT>
__device__ __forceinline__ QT to_store(float q, int shift) {
  return static_cast<QT>(static_cast<int>(q) - shift);
}

// CS: SmoothQuant per-channel scale fused in front of the quantizer: x[m,k] / chan_scale[k] (IEEE divide), then the index
// arithmetic op for op -- the run-time form of the reference's offline fold `ln.weight /= s; fc.weight *= s`
// (ptq/smoothquant.py:64-69, algorithm.py:47-68) for activations whose producer cannot absorb 1/s.
template <typename T, typename QT, bool PER_ROW, bool CS = false>
__global__ void __launch_bounds__(256) quantize_rows_kernel(const T* __restrict__ x, QT* __restrict__ q,
                                                            int64_t cols, const float* __restrict__ scale,
                                                            const float* __restrict__ offset, float qmin,
                                                            float qmax, int shift, int32_t* __restrict__ row_sum,
                                                            int vec_ok, const float* __restrict__ chan_scale = nullptr) {
  using V = Vec16<T>;
  const int64_t row = blockIdx.x;
  const float s = scale[PER_ROW ? row : 0];
  const float o = offset[PER_ROW ? row : 0];
  const float inv_s = __fdiv_rn(1.0f, s);
  const bool fast = scale_in_fast_range(s);
  const T* xr = x + row * cols;
  QT* qr = q + row * cols;
  int acc = 0;
  if (vec_ok) {
    const int64_t nvec = cols / V::N;
    const V* xv = reinterpret_cast<const V*>(xr);
    for (int64_t i = threadIdx.x; i < nvec; i += 256) {
      V a = xv[i];
      QT out[V::N];
#pragma unroll
      for (int j = 0; j < V::N; ++j) {
        float xv = V::get(a, j);
        if constexpr (CS) xv = __fdiv_rn(xv, chan_scale[i * V::N + j]);
        float qi = q_index(xv, s, inv_s, o, qmin, qmax, fast);
        int st_v = static_cast<int>(qi) - shift;
        acc += st_v;
        out[j] = static_cast<QT>(st_v);
      }
      // V::N elements of QT: 4 B (f32->i8) .. 16 B; a single naturally aligned store
      struct alignas(sizeof(QT) * V::N) Pack { QT e[V::N]; };
      Pack p;
#pragma unroll
      for (int j = 0; j < V::N; ++j) p.e[j] = out[j];
      reinterpret_cast<Pack*>(qr)[i] = p;
    }
  } else {
    for (int64_t i = threadIdx.x; i < cols; i += 256) {
      float xv = ld<T>(xr, i);
      if constexpr (CS) xv = __fdiv_rn(xv, chan_scale[i]);
      float qi = q_index(xv, s, inv_s, o, qmin, qmax, fast);
      int st_v = static_cast<int>(qi) - shift;
      acc += st_v;
      qr[i] = static_cast<QT>(st_v);
    }
  }
  if (row_sum != nullptr) {
    __shared__ int part[4];
    acc = wave_sum(acc);
    if ((threadIdx.x & 63) == 0) part[threadIdx.x >> 6] = acc;
    __syncthreads();
    if (threadIdx.x == 0) row_sum[row] = part[0] + part[1] + part[2] + part[3];
  }
}

// fp32 -> 1-byte indices, wave-per-row: a lane converts 16 consecutive elements (four independent 16-byte
// loads in flight, one 16-byte store), so a wave instruction stores 1 KiB contiguous; the row sum is a
// wave reduction (no LDS, no barrier).  Needs cols % 16 == 0 and 16-byte aligned rows.
template <typename QT, bool PER_ROW, bool CS = false>
__global__ void __launch_bounds__(256) quantize_rows_f32_b16_kernel(const float* __restrict__ x, QT* __restrict__ q,
                                                                    int64_t rows, int64_t cols,
                                                                    const float* __restrict__ scale,
                                                                    const float* __restrict__ offset, float qmin,
                                                                    float qmax, int shift, int32_t* __restrict__ row_sum,
                                                                    const float* __restrict__ chan_scale = nullptr) {
  static_assert(sizeof(QT) == 1, "one byte per index");
  const int lane = threadIdx.x & 63;
  const int64_t wave0 = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6);
  const int64_t nwaves = (int64_t)gridDim.x * 4;
  for (int64_t row = wave0; row < rows; row += nwaves) {
    const float s = scale[PER_ROW ? row : 0];
    const float o = offset[PER_ROW ? row : 0];
    const float inv_s = __fdiv_rn(1.0f, s);
    const bool fast = scale_in_fast_range(s);
    const float* xr = x + row * cols;
    QT* qr = q + row * cols;
    int acc = 0;
    for (int64_t c = (int64_t)lane * 16; c < cols; c += 1024) {
      const float4* p = reinterpret_cast<const float4*>(xr + c);
      const float4 v0 = p[0], v1 = p[1], v2 = p[2], v3 = p[3];
      float f[16] = {v0.x, v0.y, v0.z, v0.w, v1.x, v1.y, v1.z, v1.w, v2.x, v2.y, v2.z, v2.w, v3.x, v3.y, v3.z, v3.w};
      if constexpr (CS) {      // the [cols] vector is shared by every row: L2 / L1 resident after the first rows
        const float4* cp = reinterpret_cast<const float4*>(chan_scale + c);
        const float4 c0 = cp[0], c1 = cp[1], c2 = cp[2], c3 = cp[3];
        const float cs[16] = {c0.x, c0.y, c0.z, c0.w, c1.x, c1.y, c1.z, c1.w, c2.x, c2.y, c2.z, c2.w, c3.x, c3.y, c3.z, c3.w};
#pragma unroll
        for (int e = 0; e < 16; ++e) f[e] = __fdiv_rn(f[e], cs[e]);
      }
      uint32_t w[4];
#pragma unroll
      for (int d = 0; d < 4; ++d) {
        uint32_t pk = 0;
#pragma unroll
        for (int e = 0; e < 4; ++e) {
          const int st_v = static_cast<int>(q_index(f[d * 4 + e], s, inv_s, o, qmin, qmax, fast)) - shift;
          acc += st_v;
          pk |= (static_cast<uint32_t>(st_v) & 0xffu) << (8 * e);
        }
        w[d] = pk;
      }
      *reinterpret_cast<uint4*>(qr + c) = make_uint4(w[0], w[1], w[2], w[3]);
    }
    if (row_sum != nullptr) {
      acc = wave_sum(acc);
      if (lane == 0) row_sum[row] = acc;
    }
  }
}

// ---- a5 backward: straight-through gradients of the fake-quant (training loops, algorithm.py:381/:587) ----
// One workgroup per row (per-row grids) or a grid-stride slab (per-tensor); grad_scale / grad_offset are
// accumulated with float atomics into zero-initialised outputs (one pair of atomics per workgroup).
__device__ __forceinline__ float wave_sum_f(float v) { return wave_sum_f32_dpp(v); }

// Per-tensor grid, 16-byte loads / stores, four vectors per thread in flight: the training step's largest tensors (the [heads, S, S]
// scores and probabilities, 0.5 GB each at S = 2048) go through this pass, which is HBM-bound only if enough loads are outstanding
// (the scalar kernel below, capped at 512 workgroups, reached 2.3 TB/s).  Same per-element expression tree as the scalar kernel.
__global__ void __launch_bounds__(256) fake_quant_bwd_vec_kernel(const vf4* __restrict__ x, const vf4* __restrict__ gy, int64_t nvec,
                                                                 const float* __restrict__ scale, const float* __restrict__ offset,
                                                                 float qmin, float qmax, vf4* __restrict__ gx,
                                                                 float* __restrict__ gscale, float* __restrict__ goffset) {
  __shared__ float s_gs[4], s_go[4];
  const float s = scale[0], o = offset[0];
  float acc_s = 0.f, acc_o = 0.f;
  auto one = [&](float xv, float g) -> float {
    const float t = __fdiv_rn(xv, s);
    const float r = round_ste(t);
    const float q = __fadd_rn(r, o);
    const bool inside = q >= qmin && q <= qmax;
    const float qc = clamp_nan(q, qmin, qmax);
    acc_s += inside ? g * (r - t) : g * (qc - o);
    acc_o += inside ? 0.f : -g * s;
    return inside ? __fdiv_rn(__fmul_rn(g, s), s) : 0.f;            // (g*s) * mask / s, as autograd chains it
  };
  constexpr int U = 4;
  const int64_t stride = (int64_t)gridDim.x * 256;
  for (int64_t i0 = (int64_t)blockIdx.x * 256 + threadIdx.x; i0 < nvec; i0 += stride * U) {
    vf4 xv[U], gv[U];
#pragma unroll
    for (int u = 0; u < U; ++u) {
      const int64_t i = i0 + u * stride;
      if (i < nvec) {
        xv[u] = __builtin_nontemporal_load(x + i);
        gv[u] = __builtin_nontemporal_load(gy + i);
      }
    }
#pragma unroll
    for (int u = 0; u < U; ++u) {
      const int64_t i = i0 + u * stride;
      if (i < nvec) {
        vf4 r;
        r.x = one(xv[u].x, gv[u].x);
        r.y = one(xv[u].y, gv[u].y);
        r.z = one(xv[u].z, gv[u].z);
        r.w = one(xv[u].w, gv[u].w);
        __builtin_nontemporal_store(r, gx + i);
      }
    }
  }
  acc_s = wave_sum_f(acc_s);
  acc_o = wave_sum_f(acc_o);
  const int w = threadIdx.x >> 6;
  if ((threadIdx.x & 63) == 0) {
    s_gs[w] = acc_s;
    s_go[w] = acc_o;
  }
  __syncthreads();
  if (threadIdx.x == 0) {
    atomicAdd(gscale, (s_gs[0] + s_gs[1]) + (s_gs[2] + s_gs[3]));
    atomicAdd(goffset, (s_go[0] + s_go[1]) + (s_go[2] + s_go[3]));
  }
}

template <bool PER_ROW>
__global__ void __launch_bounds__(256) fake_quant_bwd_kernel(const float* __restrict__ x, const float* __restrict__ gy,
                                                             int64_t rows, int64_t cols, const float* __restrict__ scale,
                                                             const float* __restrict__ offset, float qmin, float qmax,
                                                             float* __restrict__ gx, float* __restrict__ gscale,
                                                             float* __restrict__ goffset) {
  __shared__ float s_gs[4], s_go[4];
  float acc_s = 0.f, acc_o = 0.f;
  int64_t begin, end, step;
  float s, o;
  if (PER_ROW) {             // blockIdx.x = row
    begin = (int64_t)blockIdx.x * cols + threadIdx.x;
    end = (int64_t)(blockIdx.x + 1) * cols;
    step = 256;
    s = scale[blockIdx.x];
    o = offset[blockIdx.x];
  } else {
    begin = (int64_t)blockIdx.x * 256 + threadIdx.x;
    end = rows * cols;
    step = (int64_t)gridDim.x * 256;
    s = scale[0];
    o = offset[0];
  }
  for (int64_t i = begin; i < end; i += step) {
    const float xv = x[i], g = gy[i];
    const float t = __fdiv_rn(xv, s);
    const float r = round_ste(t);
    const float q = __fadd_rn(r, o);
    const bool inside = q >= qmin && q <= qmax;
    const float qc = clamp_nan(q, qmin, qmax);
    gx[i] = inside ? __fdiv_rn(__fmul_rn(g, s), s) : 0.f;          // (g*s) * mask / s, as autograd chains it
    acc_s += inside ? g * (r - t) : g * (qc - o);
    acc_o += inside ? 0.f : -g * s;
  }
  acc_s = wave_sum_f(acc_s);
  acc_o = wave_sum_f(acc_o);
  const int w = threadIdx.x >> 6;
  if ((threadIdx.x & 63) == 0) {
    s_gs[w] = acc_s;
    s_go[w] = acc_o;
  }
  __syncthreads();
  if (threadIdx.x == 0) {
    const int slot = PER_ROW ? blockIdx.x : 0;
    atomicAdd(gscale + slot, (s_gs[0] + s_gs[1]) + (s_gs[2] + s_gs[3]));
    atomicAdd(goffset + slot, (s_go[0] + s_go[1]) + (s_go[2] + s_go[3]));
  }
}

// ---- epilogue vectors of one QLinear ------------------------------------------------------------
__global__ void linear_epilogue_prepare_kernel(const float* __restrict__ a_scale, const float* __restrict__ a_offset,
                                               int a_shift, const float* __restrict__ w_scale,
                                               const float* __restrict__ w_offset, int per_row, int w_shift,
                                               const int32_t* __restrict__ w_colsum, int64_t N, int K,
                                               float* __restrict__ alpha, int32_t* __restrict__ w_zp,
                                               int32_t* __restrict__ col_term) {
  int64_t n = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (n >= N) return;
  const float sa = a_scale[0];
  const int za = static_cast<int>(a_offset[0]) - a_shift;
  const float sw = w_scale[per_row ? n : 0];
  const int zw = static_cast<int>(w_offset[per_row ? n : 0]) - w_shift;
  alpha[n] = __fmul_rn(sa, sw);
  w_zp[n] = zw;
  // two's-complement wrap-around is fine: the GEMM's final sum is exact when it fits int32
  col_term[n] = (int32_t)((uint32_t)(-za) * (uint32_t)w_colsum[n] + (uint32_t)K * (uint32_t)za * (uint32_t)zw);
}

// ---- W4 packing ---------------------------------------------------------------------------------
// out byte (n, kb*16 + j) = nib(n, kb*32 + j) | nib(n, kb*32 + 16 + j) << 4
__global__ void pack_w4_kernel(const uint8_t* __restrict__ nib, int64_t total_out, uint8_t* __restrict__ packed) {
  int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= total_out) return;
  int64_t grp = i >> 4;      // 16-byte output group == 32-element input block
  int j = (int)(i & 15);
  const uint8_t* src = nib + grp * 32;
  packed[i] = (uint8_t)((src[j] & 15) | ((src[j + 16] & 15) << 4));
}

static int grid_for(int64_t work_items, int block) {
  int64_t g = (work_items + block - 1) / block;
  const int64_t cap = 256 * 8;   // 256 CUs x 8 blocks, grid-stride beyond that
  return (int)(g < 1 ? 1 : (g > cap ? cap : g));
}

template <typename T>
static int launch_fake_quant(const T* x, T* y, int64_t rows, int64_t cols, const float* scale, const float* offset,
                             bool per_row, bool hmath, float qmin, float qmax, hipStream_t st) {
  const int64_t numel = rows * cols;
  constexpr int VN = Vec16<T>::N;
  const bool vec = aligned(x, 16) && aligned(y, 16) && (per_row ? (cols % VN == 0) : (numel % VN == 0)) &&
                   cols < (int64_t)0xffffffffu;
#define MQ_FQ(PR, HM)                                                                                            \
  if (vec)                                                                                                       \
    fake_quant_vec_kernel<T, PR, HM><<<grid_for(numel / VN, 256), 256, 0, st>>>(x, y, numel / VN, (uint32_t)cols, \
                                                                               scale, offset, qmin, qmax);       \
  else                                                                                                           \
    fake_quant_scalar_kernel<T, PR, HM><<<grid_for(numel, 256), 256, 0, st>>>(x, y, numel, cols, scale, offset,  \
                                                                             qmin, qmax);
  if (per_row) {
    if (hmath) { MQ_FQ(true, true) } else { MQ_FQ(true, false) }
  } else {
    if (hmath) { MQ_FQ(false, true) } else { MQ_FQ(false, false) }
  }
#undef MQ_FQ
  MQ_LAUNCH_CHECK("mq_fake_quant");
  return MQ_OK;
}

// ---- a5 -> int8, fragment-blocked ("tiled") output for the generated-ISA GEMM loop ------------------------------
// Layout: 1-KiB blocks of 16 rows x 64 k, ordered [row block][k block]; inside a block lane l = (row & 15) + 16 * ((k & 63) >> 4)
// owns the 16 bytes k & 15 -- the register image of a v_mfma_i32_16x16x64_i8 operand, so the GEMM loads one fragment
// with ONE fully coalesced global_load_dwordx4 (a row-major fragment is 16 rows x 64 B = 16 half cache lines: measured
// 4 us slower per launch).  A workgroup owns a row block: its 8 waves split the k blocks, every lane reads 64 B of
// fp32 per block (four float4; the four lanes of a row cover 256 contiguous bytes) and writes its 16 bytes; row sums
// of the stored values go through LDS atomics.  Same index arithmetic as quantize_rows_* (bit-exact indices).
// A workgroup owns 8 rows (half a row block: at M = 2048 that is 256 workgroups, one per CU; the conversion is
// ~20 VALU ops per element, so leaving half the CUs idle doubles the kernel).  A wave converts 8 rows x 2 k blocks per
// step: lane = r + 8 * kq + 32 * ksel reads the 64 bytes (16 fp32) of row r, k block kb0 + ksel, quarter kq and stores
// its 16 bytes at the fragment position; the 8 lanes (kq, ksel) of a row reduce the row sum, LDS atomics across waves.
template <typename T, bool HAS_SUM, int STEPS, bool CS = false>   // STEPS: (k block pairs per wave) held in flight at once (0: generic loop)
__global__ void __launch_bounds__(512) quantize_tiled_kernel(const T* __restrict__ x, int8_t* __restrict__ q, int64_t rows,
                                                             int64_t cols, const float* __restrict__ scale,
                                                             const float* __restrict__ offset, float qmin, float qmax,
                                                             int shift, int32_t* __restrict__ row_sum,
                                                             const float* __restrict__ chan_scale = nullptr) {
  __shared__ int s_sum[8];
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int r = lane & 7, kq = (lane >> 3) & 3, ksel = lane >> 5;
  const int64_t row_raw = (int64_t)blockIdx.x * 8 + r;
  const int64_t row = row_raw < rows ? row_raw : rows - 1;      // rows past the end are padding (written, never used)
  const int64_t rb = row_raw >> 4;
  const int r16 = (int)(row_raw & 15);
  const float s = scale[0], o = offset[0];
  const float inv_s = __fdiv_rn(1.0f, s);
  const int kblocks = (int)(cols >> 6), kpairs = kblocks >> 1;
  if (HAS_SUM && threadIdx.x < 8) s_sum[threadIdx.x] = 0;
  if (HAS_SUM) __syncthreads();
  int acc = 0;
  const T* xrow = x + row * cols + kq * 16;
  int8_t* qdst = q + ((rb * kblocks) << 10) + ((r16 + 16 * kq) << 4);
  auto emit = [&](int kb, const float (&fin)[16]) {
    float f[16];
#pragma unroll
    for (int e = 0; e < 16; ++e) f[e] = fin[e];
    if constexpr (CS) {        // this lane's 16 channels: k = kb * 64 + kq * 16 + e
      const float4* cp = reinterpret_cast<const float4*>(chan_scale + (int64_t)kb * 64 + kq * 16);
      const float4 c0 = cp[0], c1 = cp[1], c2 = cp[2], c3 = cp[3];
      const float cs[16] = {c0.x, c0.y, c0.z, c0.w, c1.x, c1.y, c1.z, c1.w, c2.x, c2.y, c2.z, c2.w, c3.x, c3.y, c3.z, c3.w};
#pragma unroll
      for (int e = 0; e < 16; ++e) f[e] = __fdiv_rn(f[e], cs[e]);
    }
    uint32_t w[4];
#pragma unroll
    for (int d = 0; d < 4; ++d) {
      uint32_t pk = 0;
#pragma unroll
      for (int e = 0; e < 4; ++e) {
        const int st_v = static_cast<int>(q_index(f[d * 4 + e], s, inv_s, o, qmin, qmax)) - shift;
        acc += st_v;
        pk |= (static_cast<uint32_t>(st_v) & 0xffu) << (8 * e);
      }
      w[d] = pk;
    }
    *reinterpret_cast<uint4*>(qdst + ((int64_t)kb << 10)) = make_uint4(w[0], w[1], w[2], w[3]);
  };
  if constexpr (STEPS > 0 && std::is_same<T, float>::value) {
    float4 v[STEPS][4];       // everything this wave converts is requested before the first conversion
#pragma unroll
    for (int i = 0; i < STEPS; ++i) {
      const float4* p = reinterpret_cast<const float4*>(xrow + (int64_t)(2 * (wave + 8 * i) + ksel) * 64);
#pragma unroll
      for (int d = 0; d < 4; ++d) v[i][d] = p[d];
    }
#pragma unroll
    for (int i = 0; i < STEPS; ++i) {
      const float f[16] = {v[i][0].x, v[i][0].y, v[i][0].z, v[i][0].w, v[i][1].x, v[i][1].y, v[i][1].z, v[i][1].w,
                           v[i][2].x, v[i][2].y, v[i][2].z, v[i][2].w, v[i][3].x, v[i][3].y, v[i][3].z, v[i][3].w};
      emit(2 * (wave + 8 * i) + ksel, f);
    }
  } else {
    for (int kp = wave; kp < kpairs; kp += 8) {
      const int kb = 2 * kp + ksel;
      float f[16];
#pragma unroll
      for (int e = 0; e < 16; ++e) f[e] = ld<T>(xrow + (int64_t)kb * 64, e);
      emit(kb, f);
    }
  }
  if (HAS_SUM) {
    acc += __shfl_xor(acc, 8, 64);
    acc += __shfl_xor(acc, 16, 64);
    acc += __shfl_xor(acc, 32, 64);
    if (lane < 8) atomicAdd(&s_sum[r], acc);
    __syncthreads();
    if (threadIdx.x < 8 && (int64_t)blockIdx.x * 8 + threadIdx.x < rows) row_sum[(int64_t)blockIdx.x * 8 + threadIdx.x] = s_sum[threadIdx.x];
  }
}

template <typename T, typename QT>
static int launch_quantize(const T* x, QT* q, int64_t rows, int64_t cols, const float* scale, const float* offset,
                           bool per_row, float qmin, float qmax, int shift, int32_t* row_sum, const float* chan_scale,
                           hipStream_t st) {
  constexpr int VN = Vec16<T>::N;
  const int vec_ok = aligned(x, 16) && aligned(q, sizeof(QT) * VN) && (cols % VN == 0);
  if (chan_scale != nullptr) {       // SmoothQuant channel scale: per-tensor grids of fp32 activations (checked by the caller)
    if constexpr (std::is_same<T, float>::value) {
      if constexpr (sizeof(QT) == 1) {
        if (aligned(x, 16) && aligned(q, 16) && aligned(chan_scale, 16) && cols % 16 == 0 && cols >= 256) {
          int64_t blocks = (rows + 3) / 4;
          if (blocks > 256 * 16) blocks = 256 * 16;
          quantize_rows_f32_b16_kernel<QT, false, true><<<(unsigned)blocks, 256, 0, st>>>(x, q, rows, cols, scale, offset, qmin,
                                                                                          qmax, shift, row_sum, chan_scale);
          MQ_LAUNCH_CHECK("mq_quantize");
          return MQ_OK;
        }
      }
      quantize_rows_kernel<T, QT, false, true><<<(unsigned)rows, 256, 0, st>>>(x, q, cols, scale, offset, qmin, qmax, shift,
                                                                             row_sum, vec_ok, chan_scale);
      MQ_LAUNCH_CHECK("mq_quantize");
      return MQ_OK;
    }
  }
  if constexpr (std::is_same<T, float>::value && sizeof(QT) == 1) {
    if (aligned(x, 16) && aligned(q, 16) && cols % 16 == 0 && cols >= 256) {
      int64_t blocks = (rows + 3) / 4;
      if (blocks > 256 * 16) blocks = 256 * 16;           // 16 workgroups per CU, wave-stride over the rest
      if (per_row)
        quantize_rows_f32_b16_kernel<QT, true><<<(unsigned)blocks, 256, 0, st>>>(x, q, rows, cols, scale, offset, qmin,
                                                                                 qmax, shift, row_sum);
      else
        quantize_rows_f32_b16_kernel<QT, false><<<(unsigned)blocks, 256, 0, st>>>(x, q, rows, cols, scale, offset, qmin,
                                                                                  qmax, shift, row_sum);
      MQ_LAUNCH_CHECK("mq_quantize");
      return MQ_OK;
    }
  }
  if (per_row)
    quantize_rows_kernel<T, QT, true><<<(unsigned)rows, 256, 0, st>>>(x, q, cols, scale, offset, qmin, qmax, shift,
                                                                    row_sum, vec_ok);
  else
    quantize_rows_kernel<T, QT, false><<<(unsigned)rows, 256, 0, st>>>(x, q, cols, scale, offset, qmin, qmax, shift,
                                                                     row_sum, vec_ok);
  MQ_LAUNCH_CHECK("mq_quantize");
  return MQ_OK;
}


// fp32 rows of 1024 .. 4096 columns: loads along the rows (a wave reads 1 KiB runs), the int8 results staged in an LDS tile in the
// image's order, stores as 128-byte runs (8 rows x 16 B: whole lines of a fragment block).  1024 threads = four groups of 256, two
// rows each, every load in flight before the first conversion.  Same index arithmetic as quantize_tiled_kernel: identical images.
// GRPS = 2: four rows per 512-thread workgroup, TWO workgroups per CU -- one's loads fly while the other converts and stores
// (mq_quantize_tiled_set_rows; the same change as in mq_norm.hip's norm_tiled8_kernel).
template <int V, bool HAS_SUM, int GRPS = 4>
__global__ void __launch_bounds__(256 * GRPS) quantize_tiled8_kernel(const float* __restrict__ x, int8_t* __restrict__ q, int64_t rows, int64_t cols,
                                                               const float* __restrict__ scale, const float* __restrict__ offset, float qmin,
                                                               float qmax, int shift, int32_t* __restrict__ row_sum) {
  constexpr int RW = 2 * GRPS;                                      // rows per workgroup
  extern __shared__ __attribute__((aligned(16))) int8_t stage8[];   // [cols / 16 pieces][RW rows][16 B]
  __shared__ int s_part[RW][4];
  const int grp = threadIdx.x >> 8, lane = threadIdx.x & 255, wv_id = (threadIdx.x >> 6) & 3;
  const int nvec = (int)(cols >> 2);
  const int64_t row0 = (int64_t)blockIdx.x * RW;
  const float s = scale[0], o = offset[0];
  const float inv_s = __fdiv_rn(1.0f, s);
  const float ubias = (float)(128 - shift);               // image_u8f / image_pack4 (mq_common.h)
  float4 xs[2][V];
#pragma unroll
  for (int j = 0; j < 2; ++j) {
    const int64_t row = row0 + grp * 2 + j;
    const float4* xr = reinterpret_cast<const float4*>(x + (row < rows ? row : rows - 1) * cols);
#pragma unroll
    for (int k = 0; k < V; ++k) xs[j][k] = xr[lane + 256 * k < nvec ? lane + 256 * k : nvec - 1];
  }
#pragma unroll
  for (int j = 0; j < 2; ++j) {
    uint32_t usum = 0;
#pragma unroll
    for (int k = 0; k < V; ++k) {
      const int i = lane + 256 * k;
      if (i < nvec) {
        const float4 f = xs[j][k];
        // two elements per VALU instruction where a packed form exists (mq_common.h image_u8f2: the same bits)
        const v2f u01 = image_u8f2((v2f){f.x, f.y}, s, inv_s, o, qmin, qmax, ubias), u23 = image_u8f2((v2f){f.z, f.w}, s, inv_s, o, qmin, qmax, ubias);
        const uint32_t pk = image_pack4(u01.x, u01.y, u23.x, u23.y, usum);
        *reinterpret_cast<uint32_t*>(stage8 + (i >> 2) * (RW * 16) + ((grp * 2 + j) << 4) + ((i & 3) << 2)) = pk;
      }
    }
    if (HAS_SUM) {
      const int acc = mq::wave_sum((int)usum);
      if ((threadIdx.x & 63) == 0) s_part[grp * 2 + j][wv_id] = acc;
    }
  }
  __syncthreads();
  if (HAS_SUM && threadIdx.x < RW && row0 + threadIdx.x < rows)
    row_sum[row0 + threadIdx.x] = (s_part[threadIdx.x][0] + s_part[threadIdx.x][1]) + (s_part[threadIdx.x][2] + s_part[threadIdx.x][3]) - 128 * (int)cols;
  const int units = (int)(cols >> 4) * RW;                          // RW rows x cols / 16 sixteen-byte units
  const int64_t rb = row0 >> 4;
  const int half = (int)(row0 & 15);
  for (int p = threadIdx.x; p < units; p += 256 * GRPS) {           // rows past `rows` are padding of the image: written like the others
    const int piece = p / RW, r8 = p % RW;
    *reinterpret_cast<uint4*>(q + ((rb * (cols >> 6) + (piece >> 2)) << 10) + ((piece & 3) << 8) + ((half + r8) << 4)) =
        *reinterpret_cast<const uint4*>(stage8 + (p << 4));
  }
}

}  // namespace mq

using namespace mq;

extern "C" {

int mq_version(void) { return MQ_VERSION; }

const char* mq_last_error(void) { return g_err; }

int mq_device_info(int* cu_count, int* max_clock_khz, char* arch_name, size_t arch_name_len) {
  int dev = 0;
  hipDeviceProp_t p;
  if (hipGetDevice(&dev) != hipSuccess || hipGetDeviceProperties(&p, dev) != hipSuccess) {
    set_error("mq_device_info: no HIP device");
    return MQ_EHIP;
  }
  if (cu_count) *cu_count = p.multiProcessorCount;
  if (max_clock_khz) *max_clock_khz = p.clockRate;
  if (arch_name && arch_name_len) snprintf(arch_name, arch_name_len, "%s", p.gcnArchName);
  return MQ_OK;
}

int mq_scale_offset_from_minmax(const float* min_val, const float* max_val, int64_t n, int bitwidth,
                                int is_symmetric, float* scale, float* offset, mq_stream_t stream) {
  MQ_REQUIRE(min_val && max_val && scale && offset, "mq_scale_offset_from_minmax: null pointer");
  MQ_REQUIRE(n >= 0 && bitwidth >= 2 && bitwidth <= 16, "mq_scale_offset_from_minmax: n=%lld bitwidth=%d",
             (long long)n, bitwidth);
  if (n == 0) return MQ_OK;
  const float qmax = is_symmetric ? (float)((1 << (bitwidth - 1)) - 1) : (float)((1 << bitwidth) - 1);
  scale_offset_kernel<<<(unsigned)((n + 255) / 256), 256, 0, as_stream(stream)>>>(min_val, max_val, n, qmax,
                                                                                 is_symmetric, scale, offset);
  MQ_LAUNCH_CHECK("mq_scale_offset_from_minmax");
  return MQ_OK;
}

int mq_fake_quant(const void* x, void* y, int dtype, int64_t rows, int64_t cols, const float* scale,
                  const float* offset, int64_t n_scale, float qmin, float qmax, mq_stream_t stream) {
  MQ_REQUIRE(rows >= 0 && cols >= 0, "mq_fake_quant: negative shape");
  MQ_REQUIRE(n_scale == 1 || n_scale == rows, "mq_fake_quant: n_scale=%lld must be 1 or rows=%lld",
             (long long)n_scale, (long long)rows);
  MQ_REQUIRE(qmin <= qmax, "mq_fake_quant: qmin > qmax");
  if (rows == 0 || cols == 0) return MQ_OK;      // empty tensor (its data pointer may be NULL): nothing to do
  MQ_REQUIRE(x && y && scale && offset, "mq_fake_quant: null pointer");
  const bool per_row = (n_scale == rows) && rows > 1;
  if (dtype == MQ_F32)
    return launch_fake_quant<float>((const float*)x, (float*)y, rows, cols, scale, offset, per_row, false, qmin, qmax,
                                    as_stream(stream));
  if (dtype == MQ_F16)  // per-tensor: half math per op; per-row: fp32 math, one final rounding
    return launch_fake_quant<__half>((const __half*)x, (__half*)y, rows, cols, scale, offset, per_row, !per_row,
                                     qmin, qmax, as_stream(stream));
  set_error("mq_fake_quant: dtype %d not supported (MQ_F32, MQ_F16)", dtype);
  return MQ_EUNSUPPORTED;
}

int mq_fake_quant_backward(const float* x, const float* grad_y, int64_t rows, int64_t cols, const float* scale,
                           const float* offset, int64_t n_scale, float qmin, float qmax, float* grad_x,
                           float* grad_scale, float* grad_offset, mq_stream_t stream) {
  MQ_REQUIRE(rows >= 0 && cols >= 0 && (n_scale == 1 || n_scale == rows), "mq_fake_quant_backward: bad shape");
  if (rows == 0 || cols == 0) return MQ_OK;      // empty tensor: the (zero-initialised) scale / offset gradients stay zero
  MQ_REQUIRE(x && grad_y && scale && offset && grad_x && grad_scale && grad_offset, "mq_fake_quant_backward: null pointer");
  MQ_REQUIRE(rows < (int64_t)0x7fffffff, "mq_fake_quant_backward: too many rows");
  if (n_scale == rows && rows > 1) {
    fake_quant_bwd_kernel<true><<<(unsigned)rows, 256, 0, as_stream(stream)>>>(x, grad_y, rows, cols, scale, offset, qmin,
                                                                             qmax, grad_x, grad_scale, grad_offset);
  } else {
    const int64_t numel = rows * cols;
    if (numel >= (1 << 16) && numel % 4 == 0 && aligned(x, 16) && aligned(grad_y, 16) && aligned(grad_x, 16)) {
      int64_t gv = (numel / 4 + 1023) / 1024;      // four 16-byte vectors per thread and sweep
      if (gv > 2048) gv = 2048;                    // 8 workgroups per CU; 2 048 pairs of same-address atomics at most
      fake_quant_bwd_vec_kernel<<<(unsigned)gv, 256, 0, as_stream(stream)>>>((const vf4*)x, (const vf4*)grad_y, numel / 4, scale, offset,
                                                                            qmin, qmax, (vf4*)grad_x, grad_scale, grad_offset);
      MQ_LAUNCH_CHECK("mq_fake_quant_backward");
      return MQ_OK;
    }
    int64_t g = (numel + 1023) / 1024;
    if (g < 1) g = 1;
    if (g > 512) g = 512;     // 512 pairs of same-address atomics at most
    fake_quant_bwd_kernel<false><<<(unsigned)g, 256, 0, as_stream(stream)>>>(x, grad_y, rows, cols, scale, offset, qmin,
                                                                           qmax, grad_x, grad_scale, grad_offset);
  }
  MQ_LAUNCH_CHECK("mq_fake_quant_backward");
  return MQ_OK;
}

int mq_lwc_fake_quant(const float* w, int64_t rows, int64_t cols, const float* sig_lo, const float* sig_hi, int bitwidth,
                      int is_symmetric, float* out, float* row_min, float* row_max, float* scale, float* offset, mq_stream_t stream) {
  MQ_REQUIRE(rows >= 0 && cols >= 0 && rows < (int64_t)0x7fffffff, "mq_lwc_fake_quant: bad shape %lld x %lld", (long long)rows, (long long)cols);
  MQ_REQUIRE(bitwidth >= 2 && bitwidth <= 16, "mq_lwc_fake_quant: bitwidth=%d", bitwidth);
  if (rows == 0 || cols == 0) return MQ_OK;
  MQ_REQUIRE(w && sig_lo && sig_hi && out && row_min && row_max && scale && offset, "mq_lwc_fake_quant: null pointer");
  MQ_REQUIRE(cols % 4 == 0 && cols <= 16384 && aligned(w, 16) && aligned(out, 16),
             "mq_lwc_fake_quant: rows of up to 16384 columns, a multiple of 4, 16-byte aligned (cols=%lld)", (long long)cols);
  const float qmin = is_symmetric ? -(float)(1 << (bitwidth - 1)) : 0.0f;
  const float qmax = is_symmetric ? (float)((1 << (bitwidth - 1)) - 1) : (float)((1 << bitwidth) - 1);
  hipStream_t st = as_stream(stream);
#define MQ_LWC(V)                                                                                                              \
  lwc_fake_quant_kernel<V><<<(unsigned)rows, 256, 0, st>>>(w, (int)cols, sig_lo, sig_hi, qmin, qmax, is_symmetric, out, row_min, \
                                                         row_max, scale, offset)
  if (cols <= 2048) MQ_LWC(2);
  else if (cols <= 4096) MQ_LWC(4);
  else if (cols <= 6144) MQ_LWC(6);
  else if (cols <= 8192) MQ_LWC(8);
  else MQ_LWC(16);
#undef MQ_LWC
  MQ_LAUNCH_CHECK("mq_lwc_fake_quant");
  return MQ_OK;
}

int mq_lwc_fake_quant_backward(const float* w, const float* grad_out, int64_t rows, int64_t cols, const float* sig_lo,
                               const float* sig_hi, const float* row_min, const float* row_max, int bitwidth, int is_symmetric,
                               float* grad_w, float* grad_sig_lo, float* grad_sig_hi, mq_stream_t stream) {
  MQ_REQUIRE(rows >= 0 && cols >= 0 && rows < (int64_t)0x7fffffff, "mq_lwc_fake_quant_backward: bad shape");
  MQ_REQUIRE(bitwidth >= 2 && bitwidth <= 16, "mq_lwc_fake_quant_backward: bitwidth=%d", bitwidth);
  if (rows == 0 || cols == 0) return MQ_OK;
  MQ_REQUIRE(w && grad_out && sig_lo && sig_hi && row_min && row_max && grad_w && grad_sig_lo && grad_sig_hi,
             "mq_lwc_fake_quant_backward: null pointer");
  MQ_REQUIRE(cols % 4 == 0 && cols <= 16384 && aligned(w, 16) && aligned(grad_out, 16) && aligned(grad_w, 16),
             "mq_lwc_fake_quant_backward: rows of up to 16384 columns, a multiple of 4, 16-byte aligned (cols=%lld)", (long long)cols);
  const float qmin = is_symmetric ? -(float)(1 << (bitwidth - 1)) : 0.0f;
  const float qmax = is_symmetric ? (float)((1 << (bitwidth - 1)) - 1) : (float)((1 << bitwidth) - 1);
  hipStream_t st = as_stream(stream);
#define MQ_LWC(V)                                                                                                               \
  lwc_fake_quant_bwd_kernel<V><<<(unsigned)rows, 256, 0, st>>>(w, grad_out, (int)cols, sig_lo, sig_hi, row_min, row_max, qmin, qmax, \
                                                             is_symmetric, grad_w, grad_sig_lo, grad_sig_hi)
  if (cols <= 2048) MQ_LWC(2);
  else if (cols <= 4096) MQ_LWC(4);
  else if (cols <= 6144) MQ_LWC(6);
  else if (cols <= 8192) MQ_LWC(8);
  else MQ_LWC(16);
#undef MQ_LWC
  MQ_LAUNCH_CHECK("mq_lwc_fake_quant_backward");
  return MQ_OK;
}

static int attn_probs_check(const char* fn, const float* raw, int64_t rows, int64_t cols, const float* mask, int64_t mask_rows,
                            const float* s1, const float* o1, const float* s2, const float* o2, float sqrt_d) {
  MQ_REQUIRE(rows >= 0 && cols >= 0, "%s: negative shape", fn);
  if (rows == 0 || cols == 0) return MQ_OK;
  MQ_REQUIRE(raw && s1 && o1 && s2 && o2, "%s: null pointer", fn);
  MQ_REQUIRE(cols % 4 == 0 && cols <= 4096 && aligned(raw, 16) && (mask == nullptr || aligned(mask, 16)),
             "%s: rows of up to 4096 columns, a multiple of 4, 16-byte aligned (cols=%lld)", fn, (long long)cols);
  MQ_REQUIRE(mask == nullptr || (mask_rows >= 1 && rows % mask_rows == 0), "%s: rows=%lld is not a multiple of mask_rows=%lld", fn,
             (long long)rows, (long long)mask_rows);
  MQ_REQUIRE(sqrt_d > 0.f, "%s: sqrt_d=%g", fn, sqrt_d);
  return MQ_OK;
}

#define MQ_ATTN_PROBS_DISPATCH(KERNEL, GRID, ...)                         \
  do {                                                                    \
    if (cols <= 256) KERNEL<1><<<GRID, 256, 0, st>>>(__VA_ARGS__);        \
    else if (cols <= 512) KERNEL<2><<<GRID, 256, 0, st>>>(__VA_ARGS__);   \
    else if (cols <= 1024) KERNEL<4><<<GRID, 256, 0, st>>>(__VA_ARGS__);  \
    else if (cols <= 2048) KERNEL<8><<<GRID, 256, 0, st>>>(__VA_ARGS__);  \
    else KERNEL<16><<<GRID, 256, 0, st>>>(__VA_ARGS__);                   \
  } while (0)

int mq_attention_probs_train(const float* raw, int64_t rows, int64_t cols, const float* mask, int64_t mask_rows, const float* s1,
                             const float* o1, float qmin1, float qmax1, const float* s2, const float* o2, float qmin2, float qmax2,
                             float sqrt_d, float* out, mq_stream_t stream) {
  const int rc = attn_probs_check("mq_attention_probs_train", raw, rows, cols, mask, mask_rows, s1, o1, s2, o2, sqrt_d);
  if (rc != MQ_OK || rows == 0 || cols == 0) return rc;
  MQ_REQUIRE(out && aligned(out, 16), "mq_attention_probs_train: out must be 16-byte aligned");
  AttnProbsArgs a{raw, mask, s1, o1, s2, o2, qmin1, qmax1, qmin2, qmax2, sqrt_d, rows, (int)cols, (int)(mask ? mask_rows : 1)};
  hipStream_t st = as_stream(stream);
  int64_t grid = (rows + 3) / 4;
  if (grid > 256 * 32) grid = 256 * 32;
  MQ_ATTN_PROBS_DISPATCH(attn_probs_fwd_kernel, (unsigned)grid, a, out);
  MQ_LAUNCH_CHECK("mq_attention_probs_train");
  return MQ_OK;
}

int mq_attention_probs_train_backward(const float* raw, const float* grad_out, int64_t rows, int64_t cols, const float* mask,
                                      int64_t mask_rows, const float* s1, const float* o1, float qmin1, float qmax1, const float* s2,
                                      const float* o2, float qmin2, float qmax2, float sqrt_d, float* grad_raw, float* grad_grids,
                                      mq_stream_t stream) {
  const int rc = attn_probs_check("mq_attention_probs_train_backward", raw, rows, cols, mask, mask_rows, s1, o1, s2, o2, sqrt_d);
  if (rc != MQ_OK || rows == 0 || cols == 0) return rc;
  MQ_REQUIRE(grad_out && grad_raw && grad_grids && aligned(grad_out, 16) && aligned(grad_raw, 16),
             "mq_attention_probs_train_backward: null or unaligned pointer");
  AttnProbsArgs a{raw, mask, s1, o1, s2, o2, qmin1, qmax1, qmin2, qmax2, sqrt_d, rows, (int)cols, (int)(mask ? mask_rows : 1)};
  hipStream_t st = as_stream(stream);
  // rows of up to 512 keys: a wave per row; longer rows: the workgroup per row.  At most 2 048 workgroups (4 same-address atomics each).
  if (cols <= 512) {
    int64_t grid = (rows + 3) / 4;
    if (grid > 2048) grid = 2048;
    if (cols <= 256) attn_probs_bwd_kernel<1, 64><<<(unsigned)grid, 256, 0, st>>>(a, grad_out, grad_raw, grad_grids);
    else attn_probs_bwd_kernel<2, 64><<<(unsigned)grid, 256, 0, st>>>(a, grad_out, grad_raw, grad_grids);
  } else {
    const int64_t grid = rows > 2048 ? 2048 : rows;
    if (cols <= 1024) attn_probs_bwd_kernel<1, 256><<<(unsigned)grid, 256, 0, st>>>(a, grad_out, grad_raw, grad_grids);
    else if (cols <= 2048) attn_probs_bwd_kernel<2, 256><<<(unsigned)grid, 256, 0, st>>>(a, grad_out, grad_raw, grad_grids);
    else attn_probs_bwd_kernel<4, 256><<<(unsigned)grid, 256, 0, st>>>(a, grad_out, grad_raw, grad_grids);
  }
  MQ_LAUNCH_CHECK("mq_attention_probs_train_backward");
  return MQ_OK;
}
#undef MQ_ATTN_PROBS_DISPATCH

int mq_quantize(const void* x, int dtype, int64_t rows, int64_t cols, const float* scale, const float* offset,
                int64_t n_scale, float qmin, float qmax, int shift, const float* chan_scale, void* q, int q_dtype,
                int32_t* row_sum, mq_stream_t stream) {
  MQ_REQUIRE(rows >= 0 && cols >= 0 && rows < (int64_t)0x7fffffff, "mq_quantize: bad shape %lld x %lld",
             (long long)rows, (long long)cols);
  MQ_REQUIRE(n_scale == 1 || n_scale == rows, "mq_quantize: n_scale=%lld must be 1 or rows=%lld", (long long)n_scale,
             (long long)rows);
  if (rows == 0 || cols == 0) return MQ_OK;      // empty tensor (its data pointer may be NULL)
  MQ_REQUIRE(x && q && scale && offset, "mq_quantize: null pointer");
  MQ_REQUIRE(chan_scale == nullptr || (dtype == MQ_F32 && n_scale == 1),
             "mq_quantize: chan_scale needs float32 activations and a per-tensor grid");
  const bool per_row = (n_scale == rows) && rows > 1 && chan_scale == nullptr;
  const float lo = qmin - (float)shift, hi = qmax - (float)shift;
  hipStream_t st = as_stream(stream);
#define MQ_Q(T, QT, LO, HI)                                                                                          \
  do {                                                                                                               \
    MQ_REQUIRE(lo >= (float)(LO) && hi <= (float)(HI), "mq_quantize: [%g,%g]-%d does not fit the storage type", qmin, \
               qmax, shift);                                                                                         \
    return launch_quantize<T, QT>((const T*)x, (QT*)q, rows, cols, scale, offset, per_row, qmin, qmax, shift,        \
                                  row_sum, chan_scale, st);                                                          \
  } while (0)
#define MQ_QD(T)                                          \
  switch (q_dtype) {                                      \
    case MQ_I8: MQ_Q(T, int8_t, -128, 127);               \
    case MQ_U8: MQ_Q(T, uint8_t, 0, 255);                 \
    case MQ_I16: MQ_Q(T, int16_t, -32768, 32767);         \
    case MQ_U16: MQ_Q(T, uint16_t, 0, 65535);             \
    case MQ_I32: MQ_Q(T, int32_t, -2147483648.0, 2147483520.0); \
    default: break;                                       \
  }
  if (dtype == MQ_F32) { MQ_QD(float) }
  else if (dtype == MQ_F16) { MQ_QD(__half) }
#undef MQ_QD
#undef MQ_Q
  set_error("mq_quantize: dtype %d -> q_dtype %d not supported", dtype, q_dtype);
  return MQ_EUNSUPPORTED;
}

static std::atomic<int> g_tiled8_rows{0};       // tuning hook: rows per workgroup of the staged kernel: 0 = by shape (4 up to 2048 columns), 4 / 8 forced
extern "C" int mq_quantize_tiled_set_rows(int rows) {
  g_tiled8_rows = rows == 4 ? 4 : (rows == 8 ? 8 : 0);
  return 0;
}
static std::atomic<int> g_tiled8{1};            // tuning hook: 0 = the lane-per-fragment kernel for every shape
int mq_quantize_tiled_set_staged(int on) {
  g_tiled8 = on ? 1 : 0;
  return 0;
}

int mq_quantize_tiled(const void* x, int dtype, int64_t rows, int64_t cols, const float* scale, const float* offset,
                      float qmin, float qmax, int shift, const float* chan_scale, int8_t* q_tiled, int32_t* row_sum,
                      mq_stream_t stream) {
  MQ_REQUIRE(rows != 0 ? (x && q_tiled && scale && offset) : true, "mq_quantize_tiled: null pointer");
  MQ_REQUIRE(rows >= 0 && cols > 0 && cols % 128 == 0 && (rows + 15) / 8 < (int64_t)0x7fffffff,
             "mq_quantize_tiled: bad shape %lld x %lld (cols must be a multiple of 128)", (long long)rows, (long long)cols);
  MQ_REQUIRE(qmin - (float)shift >= -128.f && qmax - (float)shift <= 127.f, "mq_quantize_tiled: [%g,%g]-%d does not fit int8",
             qmin, qmax, shift);
  if (rows == 0) return MQ_OK;
  MQ_REQUIRE(aligned(x, 16) && aligned(q_tiled, 16), "mq_quantize_tiled: pointers must be 16-byte aligned");
  MQ_REQUIRE(chan_scale == nullptr || (dtype == MQ_F32 && aligned(chan_scale, 16)),
             "mq_quantize_tiled: chan_scale needs float32 activations and a 16-byte aligned vector");
  const unsigned grid = (unsigned)(((rows + 15) / 16) * 2);      // 8 rows per workgroup, padding rows included
  hipStream_t st = as_stream(stream);
#define MQ_QT(T, KBW)                                                                                                 \
  do {                                                                                                                \
    if (row_sum) quantize_tiled_kernel<T, true, KBW><<<grid, 512, 0, st>>>((const T*)x, q_tiled, rows, cols, scale, offset, qmin, qmax, shift, row_sum); \
    else quantize_tiled_kernel<T, false, KBW><<<grid, 512, 0, st>>>((const T*)x, q_tiled, rows, cols, scale, offset, qmin, qmax, shift, row_sum);       \
  } while (0)
  const int64_t kblocks = cols >> 6;
  if (chan_scale != nullptr) {       // the SmoothQuant form: x / chan_scale[k] in front of the same index arithmetic
    if (row_sum) quantize_tiled_kernel<float, true, 0, true><<<grid, 512, 0, st>>>((const float*)x, q_tiled, rows, cols, scale, offset, qmin, qmax, shift, row_sum, chan_scale);
    else quantize_tiled_kernel<float, false, 0, true><<<grid, 512, 0, st>>>((const float*)x, q_tiled, rows, cols, scale, offset, qmin, qmax, shift, row_sum, chan_scale);
  } else if (dtype == MQ_F32 && cols >= 1024 && cols <= 4096 && cols % 1024 == 0 && rows >= 64 && g_tiled8.load()) {
    const int rows_knob = g_tiled8_rows.load();
    const bool four = rows_knob == 4 || (rows_knob == 0 && cols <= 2048);
    const unsigned grid8 = (unsigned)(((rows + 15) / 16) * (four ? 4 : 2));
    const size_t lds = (size_t)cols * (four ? 4 : 8);
#define MQ_QT8(V)                                                                                                                              \
  do {                                                                                                                                         \
    if (four) {                                                                                                                                \
      if (row_sum) quantize_tiled8_kernel<V, true, 2><<<grid8, 512, lds, st>>>((const float*)x, q_tiled, rows, cols, scale, offset, qmin, qmax, shift, row_sum);  \
      else quantize_tiled8_kernel<V, false, 2><<<grid8, 512, lds, st>>>((const float*)x, q_tiled, rows, cols, scale, offset, qmin, qmax, shift, row_sum);        \
    } else if (row_sum) quantize_tiled8_kernel<V, true><<<grid8, 1024, lds, st>>>((const float*)x, q_tiled, rows, cols, scale, offset, qmin, qmax, shift, row_sum);  \
    else quantize_tiled8_kernel<V, false><<<grid8, 1024, lds, st>>>((const float*)x, q_tiled, rows, cols, scale, offset, qmin, qmax, shift, row_sum);        \
  } while (0)
    if (cols == 1024) MQ_QT8(1);
    else if (cols == 2048) MQ_QT8(2);
    else if (cols == 3072) MQ_QT8(3);
    else MQ_QT8(4);
#undef MQ_QT8
  } else if (dtype == MQ_F32) {
    if (kblocks == 32) MQ_QT(float, 2);            // K = 2048: 2 steps of 2 k blocks per wave, all in flight
    else if (kblocks == 16) MQ_QT(float, 1);
    else MQ_QT(float, 0);
  } else if (dtype == MQ_F16) {
    MQ_QT(__half, 0);
  } else {
    set_error("mq_quantize_tiled: dtype %d not supported", dtype);
    return MQ_EUNSUPPORTED;
  }
#undef MQ_QT
  MQ_LAUNCH_CHECK("mq_quantize_tiled");
  return MQ_OK;
}

int mq_linear_epilogue_prepare(const float* a_scale, const float* a_offset, int a_shift, const float* w_scale,
                               const float* w_offset, int64_t n_wscale, int w_shift, const int32_t* w_colsum,
                               int64_t N, int64_t K, float* alpha, int32_t* w_zp, int32_t* col_term,
                               mq_stream_t stream) {
  MQ_REQUIRE(a_scale && a_offset && w_scale && w_offset && w_colsum && alpha && w_zp && col_term,
             "mq_linear_epilogue_prepare: null pointer");
  MQ_REQUIRE(N > 0 && K > 0 && K < (1 << 24), "mq_linear_epilogue_prepare: N=%lld K=%lld", (long long)N, (long long)K);
  MQ_REQUIRE(n_wscale == 1 || n_wscale == N, "mq_linear_epilogue_prepare: n_wscale=%lld must be 1 or N",
             (long long)n_wscale);
  linear_epilogue_prepare_kernel<<<(unsigned)((N + 255) / 256), 256, 0, as_stream(stream)>>>(
      a_scale, a_offset, a_shift, w_scale, w_offset, n_wscale == N && N > 1, w_shift, w_colsum, N, (int)K, alpha, w_zp,
      col_term);
  MQ_LAUNCH_CHECK("mq_linear_epilogue_prepare");
  return MQ_OK;
}

int mq_pack_w4(const uint8_t* nibbles, int64_t N, int64_t K, uint8_t* packed, mq_stream_t stream) {
  MQ_REQUIRE(nibbles && packed, "mq_pack_w4: null pointer");
  MQ_REQUIRE(N > 0 && K > 0 && K % 64 == 0, "mq_pack_w4: K=%lld must be a positive multiple of 64", (long long)K);
  const int64_t total = N * K / 2;
  pack_w4_kernel<<<(unsigned)((total + 255) / 256), 256, 0, as_stream(stream)>>>(nibbles, total, packed);
  MQ_LAUNCH_CHECK("mq_pack_w4");
  return MQ_OK;
}

}  // extern "C"

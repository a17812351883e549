// Min/max reductions (HBM-bound, one pass yields both): per-tensor, per-row, per-column.
//
// Replaces compute_min_max_from_tensor (qmodule.py:26-34) and the calibration statistics of
// update_act_range (ptq/generate_act_range.py:55-69).  The reference does two reduction passes plus
// two host syncs per tensor; here one pass keeps a device-resident running [min, max] with no host
// readback: lane-private min/max over 16-byte loads -> 64-lane wave reduce -> LDS across the
// workgroup's waves -> one exact float atomic per workgroup (bit-pattern ordered, mq_common.h).
// min/max are exact and order independent, so any sharding of the data gives identical results.
#include <hip/hip_fp16.h>

#include "mq_common.h"

namespace mq {

// torch.min / torch.max / aminmax return NaN when the tensor holds one (v_min_f32 / v_max_f32 would drop it):
// v_minimum3_f32 / v_maximum3_f32 (new on gfx950) propagate NaN at no cost.
__device__ __forceinline__ float min_p(float a, float b) { return __builtin_elementwise_minimum(a, b); }
__device__ __forceinline__ float max_p(float a, float b) { return __builtin_elementwise_maximum(a, b); }
__device__ __forceinline__ float wave_min_p(float v) {
  return wave_reduce_f(v, [](float a, float b) { return min_p(a, b); });
}
__device__ __forceinline__ float wave_max_p(float v) {
  return wave_reduce_f(v, [](float a, float b) { return max_p(a, b); });
}

template <typename T>
struct Ld16;
template <>
struct Ld16<float> {
  static constexpr int N = 4;
  __device__ static void minmax(const float* p, float& lo, float& hi) {
    float4 v = *reinterpret_cast<const float4*>(p);
    lo = min_p(min_p(lo, v.x), min_p(v.y, min_p(v.z, v.w)));
    hi = max_p(max_p(hi, v.x), max_p(v.y, max_p(v.z, v.w)));
  }
  __device__ static float one(const float* p) { return *p; }
};
template <>
struct Ld16<__half> {
  static constexpr int N = 8;
  __device__ static void minmax(const __half* p, float& lo, float& hi) {
    struct alignas(16) H8 { __half h[8]; };
    H8 v = *reinterpret_cast<const H8*>(p);
#pragma unroll
    for (int j = 0; j < 8; ++j) {
      float f = __half2float(v.h[j]);
      lo = min_p(lo, f);
      hi = max_p(hi, f);
    }
  }
  __device__ static float one(const __half* p) { return __half2float(*p); }
};

__global__ void minmax_init_kernel(float* mn, float* mx, int64_t n) {
  int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i < n) {
    mn[i] = __int_as_float(0x7f800000);   // +inf
    mx[i] = __int_as_float(0xff800000);   // -inf
  }
}

// canonicalise -0.0 -> +0.0 so the bit-pattern atomics treat the two zeros alike
__device__ __forceinline__ float canon(float v) { return v + 0.0f; }

// Same-address atomics serialise at the L2 (~12 ns each): with 1-2 k workgroups committing to ONE running
// [min, max] the atomics alone took longer than the streaming pass.  A relaxed agent-scope read filters them:
// only a workgroup that would actually improve the statistic issues the atomic (expected O(log #workgroups)
// per launch, and none at all once a calibration statistic has settled).  A stale read can only cause a
// redundant atomic, never a wrong result.
__device__ __forceinline__ void commit_min(float* addr, float v) {
  if (v < __hip_atomic_load(addr, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT)) atomic_min_f32(addr, v);
}
__device__ __forceinline__ void commit_max(float* addr, float v) {
  if (v > __hip_atomic_load(addr, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT)) atomic_max_f32(addr, v);
}

// NaN is sticky under the bit-pattern atomics: 0xffffffff is the largest unsigned and the smallest-but-one
// signed pattern, so neither branch of atomic_min_f32 can replace it; 0x7fffffff likewise for atomic_max_f32.
__device__ __forceinline__ void commit_nan(float* mn, float* mx) {
  atomicMax(reinterpret_cast<unsigned int*>(mn), 0xffffffffu);
  atomicMax(reinterpret_cast<int*>(mx), 0x7fffffff);
}
__device__ __forceinline__ void commit_pair(float lo, float hi, float* mn, float* mx) {
  if (lo != lo || hi != hi) {
    commit_nan(mn, mx);
  } else if (lo <= hi) {   // false only when nothing was seen
    commit_min(mn, canon(lo));
    commit_max(mx, canon(hi));
  }
}

__device__ __forceinline__ void block_commit(float lo, float hi, float* mn, float* mx) {
  __shared__ float s_lo[4], s_hi[4];
  lo = wave_min_p(lo);
  hi = wave_max_p(hi);
  const int w = threadIdx.x >> 6;
  if ((threadIdx.x & 63) == 0) {
    s_lo[w] = lo;
    s_hi[w] = hi;
  }
  __syncthreads();
  if (threadIdx.x == 0) {
    lo = min_p(min_p(s_lo[0], s_lo[1]), min_p(s_lo[2], s_lo[3]));
    hi = max_p(max_p(s_hi[0], s_hi[1]), max_p(s_hi[2], s_hi[3]));
    commit_pair(lo, hi, mn, mx);
  }
}

// per-tensor: grid-stride over 16-byte vectors; `head`/`tail` scalars cover unaligned ends
template <typename T>
__device__ __forceinline__ void stream_minmax(const T* __restrict__ x, int64_t numel, int64_t head, int64_t nvec,
                                              float& lo, float& hi) {
  using L = Ld16<T>;
  lo = __int_as_float(0x7f800000);
  hi = __int_as_float(0xff800000);
  const int64_t tid = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  const int64_t stride = (int64_t)gridDim.x * blockDim.x;
  const T* xv = x + head;
  // four independent 16-byte loads in flight per lane (one load per trip leaves HBM latency exposed:
  // measured 1.5 TB/s), each with its own running min/max so the loads do not serialise on the compare chain
  float lo1 = lo, hi1 = hi, lo2 = lo, hi2 = hi, lo3 = lo, hi3 = hi;
  int64_t i = tid;
  for (; i + 3 * stride < nvec; i += 4 * stride) {
    L::minmax(xv + i * L::N, lo, hi);
    L::minmax(xv + (i + stride) * L::N, lo1, hi1);
    L::minmax(xv + (i + 2 * stride) * L::N, lo2, hi2);
    L::minmax(xv + (i + 3 * stride) * L::N, lo3, hi3);
  }
  for (; i < nvec; i += stride) L::minmax(xv + i * L::N, lo, hi);
  lo = min_p(min_p(lo, lo1), min_p(lo2, lo3));
  hi = max_p(max_p(hi, hi1), max_p(hi2, hi3));
  // scalar ends: [0, head) and [head + nvec*N, numel)
  const int64_t tail0 = head + nvec * L::N;
  const int64_t nscalar = head + (numel - tail0);
  for (int64_t i2 = tid; i2 < nscalar; i2 += stride) {
    float f = L::one(x + (i2 < head ? i2 : tail0 + (i2 - head)));
    lo = min_p(lo, f);
    hi = max_p(hi, f);
  }
}

template <typename T>
__global__ void __launch_bounds__(256) minmax_tensor_kernel(const T* __restrict__ x, int64_t numel, int64_t head,
                                                            int64_t nvec, float* mn, float* mx) {
  float lo, hi;
  stream_minmax<T>(x, numel, head, nvec, lo, hi);
  block_commit(lo, hi, mn, mx);
}

// Fresh statistic (dynamic quantizers, first-forward weight ranges): every workgroup would improve a statistic that
// starts at +-inf, i.e. up to 1024 same-address atomics (~12 ns each) behind a 3 us streaming pass.  Two launches
// without atomics instead: per-workgroup partials into caller scratch, then one workgroup folds them and WRITES the
// result (no init kernel either).
template <typename T>
__global__ void __launch_bounds__(256) minmax_partials_kernel(const T* __restrict__ x, int64_t numel, int64_t head,
                                                              int64_t nvec, float* __restrict__ partials) {
  __shared__ float s_lo[4], s_hi[4];
  float lo, hi;
  stream_minmax<T>(x, numel, head, nvec, lo, hi);
  lo = wave_min_p(lo);
  hi = wave_max_p(hi);
  const int w = threadIdx.x >> 6;
  if ((threadIdx.x & 63) == 0) {
    s_lo[w] = lo;
    s_hi[w] = hi;
  }
  __syncthreads();
  if (threadIdx.x == 0) {
    partials[2 * blockIdx.x] = min_p(min_p(s_lo[0], s_lo[1]), min_p(s_lo[2], s_lo[3]));
    partials[2 * blockIdx.x + 1] = max_p(max_p(s_hi[0], s_hi[1]), max_p(s_hi[2], s_hi[3]));
  }
}

__global__ void __launch_bounds__(256) minmax_fold_kernel(const float* __restrict__ partials, int n, float* mn, float* mx) {
  __shared__ float s_lo[4], s_hi[4];
  float lo = __int_as_float(0x7f800000), hi = __int_as_float(0xff800000);
  for (int i = threadIdx.x; i < n; i += 256) {
    lo = min_p(lo, partials[2 * i]);
    hi = max_p(hi, partials[2 * i + 1]);
  }
  lo = wave_min_p(lo);
  hi = wave_max_p(hi);
  const int w = threadIdx.x >> 6;
  if ((threadIdx.x & 63) == 0) {
    s_lo[w] = lo;
    s_hi[w] = hi;
  }
  __syncthreads();
  if (threadIdx.x == 0) {
    lo = min_p(min_p(s_lo[0], s_lo[1]), min_p(s_lo[2], s_lo[3]));
    hi = max_p(max_p(s_hi[0], s_hi[1]), max_p(s_hi[2], s_hi[3]));
    mn[0] = (lo != lo) ? __int_as_float(0xffffffff) : (lo <= hi ? canon(lo) : lo);   // same NaN patterns as the atomics
    mx[0] = (hi != hi) ? __int_as_float(0x7fffffff) : (lo <= hi ? canon(hi) : hi);
  }
}

// per-row: one wave per row, 4 rows per workgroup; accumulates into mn[row], mx[row] (plain RMW:
// each row is owned by exactly one wave of one launch)
template <typename T>
__global__ void __launch_bounds__(256) minmax_rows_kernel(const T* __restrict__ x, int64_t rows, int64_t cols,
                                                          int vec_ok, float* __restrict__ mn,
                                                          float* __restrict__ mx) {
  using L = Ld16<T>;
  const int lane = threadIdx.x & 63;
  const int64_t row = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6);
  if (row >= rows) return;
  const T* xr = x + row * cols;
  float lo = __int_as_float(0x7f800000), hi = __int_as_float(0xff800000);
  if (vec_ok) {
    const int64_t nvec = cols / L::N;
    for (int64_t i = lane; i < nvec; i += 64) L::minmax(xr + i * L::N, lo, hi);
  } else {
    for (int64_t i = lane; i < cols; i += 64) {
      float f = L::one(xr + i);
      lo = min_p(lo, f);
      hi = max_p(hi, f);
    }
  }
  lo = wave_min_p(lo);
  hi = wave_max_p(hi);
  if (lane == 0) {
    mn[row] = min_p(mn[row], lo);
    mx[row] = max_p(mx[row], hi);
  }
}

// per-column: workgroup = 64 lanes x 4 row-groups; a lane owns Ld16::N adjacent columns, the four
// waves interleave rows; grid = (column tiles, row chunks); one atomic per column per workgroup.
template <typename T>
__global__ void __launch_bounds__(256) minmax_cols_kernel(const T* __restrict__ x, int64_t rows, int64_t cols,
                                                          int64_t rows_per_block, float* mn, float* mx) {
  using L = Ld16<T>;
  constexpr int N = L::N;
  __shared__ float s_lo[4][64 * N], s_hi[4][64 * N];
  const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
  const int64_t c0 = ((int64_t)blockIdx.x * 64 + lane) * N;
  const int64_t r0 = (int64_t)blockIdx.y * rows_per_block;
  const int64_t r1 = (r0 + rows_per_block < rows) ? r0 + rows_per_block : rows;
  float lo[N], hi[N];
#pragma unroll
  for (int j = 0; j < N; ++j) {
    lo[j] = __int_as_float(0x7f800000);
    hi[j] = __int_as_float(0xff800000);
  }
  if (c0 < cols) {
    for (int64_t r = r0 + w; r < r1; r += 4) {
      const T* p = x + r * cols + c0;
      struct alignas(16) VV { T e[N]; };
      VV v = *reinterpret_cast<const VV*>(p);
#pragma unroll
      for (int j = 0; j < N; ++j) {
        float f = (float)v.e[j];
        lo[j] = min_p(lo[j], f);
        hi[j] = max_p(hi[j], f);
      }
    }
  }
#pragma unroll
  for (int j = 0; j < N; ++j) {
    s_lo[w][lane * N + j] = lo[j];
    s_hi[w][lane * N + j] = hi[j];
  }
  __syncthreads();
  for (int i = threadIdx.x; i < 64 * N; i += 256) {
    const int64_t c = (int64_t)blockIdx.x * 64 * N + i;
    if (c < cols) {
      float l = min_p(min_p(s_lo[0][i], s_lo[1][i]), min_p(s_lo[2][i], s_lo[3][i]));
      float h = max_p(max_p(s_hi[0][i], s_hi[1][i]), max_p(s_hi[2][i], s_hi[3][i]));
      commit_pair(l, h, mn + c, mx + c);
    }
  }
}

// generic (unaligned / cols % N != 0) per-column fallback: a thread owns one column
template <typename T>
__global__ void __launch_bounds__(256) minmax_cols_scalar_kernel(const T* __restrict__ x, int64_t rows,
                                                                 int64_t cols, int64_t rows_per_block, float* mn,
                                                                 float* mx) {
  const int64_t c = (int64_t)blockIdx.x * 256 + threadIdx.x;
  if (c >= cols) return;
  const int64_t r0 = (int64_t)blockIdx.y * rows_per_block;
  const int64_t r1 = (r0 + rows_per_block < rows) ? r0 + rows_per_block : rows;
  float lo = __int_as_float(0x7f800000), hi = __int_as_float(0xff800000);
  for (int64_t r = r0; r < r1; ++r) {
    float f = Ld16<T>::one(x + r * cols + c);
    lo = min_p(lo, f);
    hi = max_p(hi, f);
  }
  commit_pair(lo, hi, mn + c, mx + c);
}

template <typename T>
static int launch_tensor(const T* x, int64_t numel, float* mn, float* mx, hipStream_t st) {
  constexpr int N = Ld16<T>::N;
  // leading scalars up to the first 16-byte boundary (tensor views may start anywhere)
  int64_t head = 0;
  const uintptr_t a = reinterpret_cast<uintptr_t>(x);
  if (a % 16) head = (int64_t)((16 - a % 16) / sizeof(T));
  if (a % sizeof(T)) head = numel;  // cannot happen for real tensors
  if (head > numel) head = numel;
  const int64_t nvec = (numel - head) / N;
  int64_t g = (nvec + 1023) / 1024;   // >= 4 vectors per lane before the grid-stride loop wraps
  if (g < 1) g = 1;
  if (g > 512) g = 512;               // 2 workgroups per CU keep 8 MB in flight; a fresh statistic sees <= 1024 atomics
  minmax_tensor_kernel<T><<<(unsigned)g, 256, 0, st>>>(x, numel, head, nvec, mn, mx);
  MQ_LAUNCH_CHECK("mq_minmax_tensor");
  return MQ_OK;
}

constexpr int kFreshMaxBlocks = 512;

template <typename T>
static int launch_tensor_fresh(const T* x, int64_t numel, float* mn, float* mx, float* scratch, hipStream_t st) {
  constexpr int N = Ld16<T>::N;
  int64_t head = 0;
  const uintptr_t a = reinterpret_cast<uintptr_t>(x);
  if (a % 16) head = (int64_t)((16 - a % 16) / sizeof(T));
  if (head > numel) head = numel;
  const int64_t nvec = (numel - head) / N;
  int64_t g = (nvec + 1023) / 1024;
  if (g < 1) g = 1;
  if (g > kFreshMaxBlocks) g = kFreshMaxBlocks;
  minmax_partials_kernel<T><<<(unsigned)g, 256, 0, st>>>(x, numel, head, nvec, scratch);
  MQ_LAUNCH_CHECK("mq_minmax_tensor_fresh");
  minmax_fold_kernel<<<1, 256, 0, st>>>(scratch, (int)g, mn, mx);
  MQ_LAUNCH_CHECK("mq_minmax_tensor_fresh(fold)");
  return MQ_OK;
}

template <typename T>
static int launch_cols(const T* x, int64_t rows, int64_t cols, float* mn, float* mx, hipStream_t st) {
  constexpr int N = Ld16<T>::N;
  const bool vec = aligned(x, 16) && cols % N == 0;
  const int64_t ctiles = vec ? (cols + 64 * N - 1) / (64 * N) : (cols + 255) / 256;
  // enough row chunks to fill ~8 workgroups per CU, at least 64 rows each
  int64_t chunks = (2048 + ctiles - 1) / ctiles;
  int64_t rpb = (rows + chunks - 1) / chunks;
  if (rpb < 64) rpb = 64;
  chunks = (rows + rpb - 1) / rpb;
  MQ_REQUIRE(chunks <= 65535 && ctiles < (int64_t)0x7fffffff, "mq_minmax_cols: shape too large");
  dim3 grid((unsigned)ctiles, (unsigned)chunks);
  if (vec)
    minmax_cols_kernel<T><<<grid, 256, 0, st>>>(x, rows, cols, rpb, mn, mx);
  else
    minmax_cols_scalar_kernel<T><<<grid, 256, 0, st>>>(x, rows, cols, rpb, mn, mx);
  MQ_LAUNCH_CHECK("mq_minmax_cols");
  return MQ_OK;
}


// ---- calibration-mode score chain of an attention block (round 5) -------------------------------------------------------------------
// generate_act_range.py:55-69 hooks qk_bmm's OUTPUT (the raw scores) and pv_bmm's INPUT (the probabilities): at S = 2048 two 537-MB
// tensors per layer, which the reference's graph (hf_model.py:513-530) also walks for `/ sqrt(d)`, `+ mask` and the softmax -- about nine
// passes over [heads, S, S] per layer, three quarters of a calibration sample's time.  Both statistics can be taken where the values
// exist: ONE pass reads a row of raw scores, folds its min / max into qk_bmm.output's running statistic, forms the reference's
// probabilities (x = raw * (1 / sqrt_d) [+ mask]; torch.softmax(dim = -1, fp32): max, exp(x - max), sum, quotient), folds THEIR min / max
// into pv_bmm.input's statistic and writes them over the scores.  A wave owns a row (up to 4096 keys in registers); the four
// running statistics are committed once per workgroup through the filtered bit-pattern atomics above (NaN sticky, as torch's amin / amax).
// CAUSAL (round 6): the mask is the causal one of a square block (row r of every [mask_rows, cols = mask_rows] matrix masks the columns
// > r): -inf is added without reading a mask, and -- store_masked == 0 -- quads that lie wholly above the diagonal are NOT stored: the
// caller hands a `out` buffer whose upper triangles already hold the zeros a previous call left there (calibration.ActRangeCollector keeps
// one per shape), which takes a quarter of the pass's bytes away.  Same arithmetic on every element as with the explicit mask.
template <int VPT, bool CAUSAL = false>
__global__ void __launch_bounds__(256) calib_probs_kernel(const float* raw, float* out /* may be raw: no __restrict__ */, const int64_t rows, const int cols,
                                                          const float* __restrict__ mask, const int mask_rows, const float inv_sqrt_d,
                                                          float* mn_raw, float* mx_raw, float* mn_p, float* mx_p, const int store_masked = 1) {
  typedef float v4f_ __attribute__((ext_vector_type(4)));
  const int lane = threadIdx.x & 63, nvec = cols >> 2;
  const int64_t nw = (int64_t)gridDim.x * 4;
  const float pinf = __int_as_float(0x7f800000), ninf = __int_as_float(0xff800000);
  float rlo = pinf, rhi = ninf, plo = rlo, phi = rhi;
  int trip = 0;
  for (int64_t lrow = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6); lrow < rows; lrow += nw, ++trip) {
    // CAUSAL: a row's work grows with its position in the block (see klive below) and a wave would meet the same position on every
    // trip (the launch covers whole blocks per trip: nw % mask_rows == 0): odd trips walk their blocks bottom-up, so every wave sees
    // light and heavy rows in turn
    int64_t row = lrow;
    if (CAUSAL && (trip & 1)) row = lrow - lrow % mask_rows + (mask_rows - 1 - lrow % mask_rows);
    const float4* rr = reinterpret_cast<const float4*>(raw + row * cols);
    const float4* mr = (!CAUSAL && mask) ? reinterpret_cast<const float4*>(mask + (row % mask_rows) * (int64_t)cols) : nullptr;
    const int diag = CAUSAL ? (int)(row % mask_rows) : cols;      // CAUSAL: columns > diag are masked
    float v[VPT * 4];
    float qlo = pinf, qhi = ninf;                                   // this row's raw statistic
#pragma unroll
    for (int k = 0; k < VPT; ++k) {
      const int i = lane + 64 * k;
      const bool valid = i < nvec;
      // read once, overwritten (in place or into the kept buffer): streaming hints on both sides (S = 2048, 32 heads: 220 -> 201 us)
      const v4f_ xv = __builtin_nontemporal_load(reinterpret_cast<const v4f_*>(rr) + (valid ? i : nvec - 1));
#pragma unroll
      for (int e = 0; e < 4; ++e) {
        v[4 * k + e] = xv[e];
        if (valid) {
          qlo = min_p(qlo, xv[e]);
          qhi = max_p(qhi, xv[e]);
        }
      }
    }
    rlo = min_p(rlo, qlo);
    rhi = max_p(rhi, qhi);
    // CAUSAL: the 256-column blocks that lie wholly above the diagonal hold -inf after the mask: no scaling, no exponential, no divide
    // for them -- unless the row holds a NaN or an infinity somewhere (inf - inf poisons the row in the masked chain: the full pass then)
    int klive = VPT;
    if (CAUSAL) {
      const bool odd = qlo != qlo || qhi != qhi || qlo == ninf || qhi == pinf;
      if (__builtin_amdgcn_ballot_w64(odd) == 0) klive = min(VPT, (diag >> 8) + 1);
    }
    float mx = ninf;
#pragma unroll
    for (int k = 0; k < VPT; ++k) {
      if (k < klive) {
        const int i = lane + 64 * k;
        const bool valid = i < nvec;
        float4 m4 = make_float4(0.f, 0.f, 0.f, 0.f);
        if (mr) m4 = mr[valid ? i : nvec - 1];
        const float ms[4] = {m4.x, m4.y, m4.z, m4.w};
#pragma unroll
        for (int e = 0; e < 4; ++e) {
          float x = __fmul_rn(v[4 * k + e], inv_sqrt_d);           // torch: tensor / python scalar == tensor * (1 / scalar)
          if (mr) x = __fadd_rn(x, ms[e]);
          if (CAUSAL) x = __fadd_rn(x, 4 * i + e > diag ? ninf : 0.f);
          v[4 * k + e] = x;
          mx = valid ? max_p(mx, x) : mx;
        }
      }
    }
    mx = wave_max_p(mx);
    float l = 0.f;
#pragma unroll
    for (int k = 0; k < VPT; ++k) {
      if (k < klive) {
        const bool valid = lane + 64 * k < nvec;
#pragma unroll
        for (int e = 0; e < 4; ++e) {
          const float ex = expf(__fsub_rn(v[4 * k + e], mx));
          v[4 * k + e] = ex;
          l += valid ? ex : 0.f;
        }
      }
    }
    l = wave_sum_f32_dpp(l);
    float4* orow = reinterpret_cast<float4*>(out + row * cols);
    // ex / l through RN(1 / l) and one fma correction (mq_common.h: div_by_scale, 3 instructions for the ~10 of the IEEE sequence -- this
    // pass is VALU-bound once the masked stores are gone): l in [1, cols] is normal and its reciprocal correctly rounded, so the
    // quotient is the IEEE one except for a divisor with an all-ones significand (wave-uniform: the true divide then) and faithfully
    // rounded where it underflows into the denormals (probabilities below 1e-38)
    const float inv_l = __fdiv_rn(1.0f, l);
    const bool quick = __builtin_amdgcn_readfirstlane((int)((__float_as_uint(l) & 0x7fffffu) != 0x7fffffu && l >= 1.0f && l <= 8192.0f)) != 0;
#pragma unroll
    for (int k = 0; k < VPT; ++k) {
      if (k < klive) {
        if (quick) {                                                // (a scalar branch: the select form computes both quotients)
#pragma unroll
          for (int e = 0; e < 4; ++e) v[4 * k + e] = div_by_scale(v[4 * k + e], l, inv_l);
        } else {
#pragma unroll
          for (int e = 0; e < 4; ++e) v[4 * k + e] = __fdiv_rn(v[4 * k + e], l);
        }
      } else {
#pragma unroll
        for (int e = 0; e < 4; ++e) v[4 * k + e] = 0.f;             // exp(-inf) / l
      }
    }
#pragma unroll
    for (int k = 0; k < VPT; ++k) {
      if (lane + 64 * k < nvec) {
#pragma unroll
        for (int e = 0; e < 4; ++e) {
          plo = min_p(plo, v[4 * k + e]);
          phi = max_p(phi, v[4 * k + e]);
        }
        if (!CAUSAL || store_masked || 4 * (lane + 64 * k) <= diag)
          __builtin_nontemporal_store((v4f_){v[4 * k], v[4 * k + 1], v[4 * k + 2], v[4 * k + 3]}, reinterpret_cast<v4f_*>(orow) + lane + 64 * k);
      }
    }
  }
  block_commit(rlo, rhi, mn_raw, mx_raw);
  __syncthreads();                                                // (block_commit's LDS slots are reused)
  block_commit(plo, phi, mn_p, mx_p);
}

}  // namespace mq
namespace mq {

// ---- calibration of a decoder layer's glue (round 6): statistics where the tensors are produced -------------------------------------
// The fp32 calibration forward (generate_act_range.py:49-122 around hf_model.py) spends a third of its GPU time in torch's elementwise
// launches between the linears -- an RMSNorm is six of them (pow, mean, + eps, rsqrt, *, * weight), each a pass over [S, hidden] -- and
// every hooked tensor is then read again for its statistic.  Two one-pass kernels take both jobs for the leaf graph of this package
// (llama.DecoderLayer / MLP, calibration.ActRangeCollector.norm_pass / gated_pass): the values are those of the module chain up to the
// order of the row sums (a few ulp, like the fused score chain above; tests bound the act_dict at 1e-5 relative).

// h = x (+ delta), running [min, max] of h (the norm's INPUT slot), y = norm(h) * weight (+ bias), running [min, max] of y (its OUTPUT
// slot).  A workgroup per row, VPT float4 per thread.  RMSNorm as hf_model.py:183-186 (x * rsqrt(mean(x^2) + eps), then weight * x);
// LayerNorm as torch.nn.functional.layer_norm (biased variance).
template <int VPT, bool LN>
__global__ void __launch_bounds__(256) calib_norm_kernel(const float* __restrict__ x, const float* __restrict__ delta, float* __restrict__ h_out,
                                                         float* __restrict__ y_out, const int64_t rows, const int cols,
                                                         const float* __restrict__ weight, const float* __restrict__ bias, const float eps,
                                                         float* mn_h, float* mx_h, float* mn_y, float* mx_y, float* mn_d, float* mx_d) {
  __shared__ float s_red[2][4];
  const int tid = threadIdx.x, lane = tid & 63, wv = tid >> 6, nvec = cols >> 2;
  typedef float v4f __attribute__((ext_vector_type(4)));
  float hlo = __int_as_float(0x7f800000), hhi = __int_as_float(0xff800000), ylo = hlo, yhi = hhi, dlo = hlo, dhi = hhi;
  for (int64_t row = blockIdx.x; row < rows; row += gridDim.x) {
    const v4f* xr = reinterpret_cast<const v4f*>(x + row * cols);
    const v4f* dr = delta ? reinterpret_cast<const v4f*>(delta + row * cols) : nullptr;
    v4f h[VPT];
    float s1 = 0.f;
#pragma unroll
    for (int k = 0; k < VPT; ++k) {
      const int i = tid + 256 * k;
      const bool valid = i < nvec;
      v4f v = xr[valid ? i : nvec - 1];
      if (dr) {
        const v4f d = dr[valid ? i : nvec - 1];
        if (valid && mn_d) {                                        // the branch's own statistic (the producing linear's output hook)
          dlo = min_p(min_p(dlo, d[0]), min_p(d[1], min_p(d[2], d[3])));
          dhi = max_p(max_p(dhi, d[0]), max_p(d[1], max_p(d[2], d[3])));
        }
        v += d;
      }
      h[k] = v;
      if (valid) {
        hlo = min_p(min_p(hlo, v[0]), min_p(v[1], min_p(v[2], v[3])));
        hhi = max_p(max_p(hhi, v[0]), max_p(v[1], max_p(v[2], v[3])));
        s1 += LN ? (v[0] + v[1]) + (v[2] + v[3]) : (v[0] * v[0] + v[1] * v[1]) + (v[2] * v[2] + v[3] * v[3]);
        if (h_out) reinterpret_cast<v4f*>(h_out + row * cols)[i] = v;
      }
    }
    auto block_sum = [&](float v, int slot) {
      v = wave_sum_f32_dpp(v);
      if (lane == 0) s_red[slot][wv] = v;
      __syncthreads();
      return (s_red[slot][0] + s_red[slot][1]) + (s_red[slot][2] + s_red[slot][3]);
    };
    float mu = 0.f, r;
    if constexpr (LN) {
      mu = __fdiv_rn(block_sum(s1, 0), (float)cols);
      float s2 = 0.f;
#pragma unroll
      for (int k = 0; k < VPT; ++k)
        if (tid + 256 * k < nvec) {
          const v4f d = h[k] - mu;
          s2 += (d[0] * d[0] + d[1] * d[1]) + (d[2] * d[2] + d[3] * d[3]);
        }
      r = __fdiv_rn(1.0f, __fsqrt_rn(__fadd_rn(__fdiv_rn(block_sum(s2, 1), (float)cols), eps)));
    } else {
      r = __fdiv_rn(1.0f, __fsqrt_rn(__fadd_rn(__fdiv_rn(block_sum(s1, 0), (float)cols), eps)));
    }
#pragma unroll
    for (int k = 0; k < VPT; ++k) {
      const int i = tid + 256 * k;
      if (i < nvec) {
        const v4f w = reinterpret_cast<const v4f*>(weight)[i];
        v4f y = LN ? (h[k] - mu) * r * w : w * (h[k] * r);
        if (LN && bias) y += reinterpret_cast<const v4f*>(bias)[i];
        ylo = min_p(min_p(ylo, y[0]), min_p(y[1], min_p(y[2], y[3])));
        yhi = max_p(max_p(yhi, y[0]), max_p(y[1], max_p(y[2], y[3])));
        reinterpret_cast<v4f*>(y_out + row * cols)[i] = y;
      }
    }
    __syncthreads();                                                // (s_red is reused by the next row)
  }
  block_commit(hlo, hhi, mn_h, mx_h);
  __syncthreads();
  block_commit(ylo, yhi, mn_y, mx_y);
  if (mn_d) {
    __syncthreads();
    block_commit(dlo, dhi, mn_d, mx_d);
  }
}

// p = act(a) * b with the running [min, max] of a (w1's output = the activation's input), act(a) (the activation's output), b (w3's
// output) and p (w2's input): hf_model.py:1057 between four hooks.  act: 0 = SiLU (x * 1 / (1 + exp(-x)): mq_act_quant's expression),
// 1 = GELU (erf).  stats: [mn_a, mx_a, mn_s, mx_s, mn_b, mx_b, mn_p, mx_p].
struct CalibGatedStats {
  float* p[8];
};
__global__ void __launch_bounds__(256) calib_gated_kernel(const float* __restrict__ a, const float* __restrict__ b, float* __restrict__ out, const int64_t numel,
                                                          const int act, const CalibGatedStats st) {
  typedef float v4f __attribute__((ext_vector_type(4)));
  const float pinf = __int_as_float(0x7f800000), ninf = __int_as_float(0xff800000);
  float lo[4] = {pinf, pinf, pinf, pinf}, hi[4] = {ninf, ninf, ninf, ninf};
  auto upd = [&](int k, const v4f v) {
    lo[k] = min_p(min_p(lo[k], v[0]), min_p(v[1], min_p(v[2], v[3])));
    hi[k] = max_p(max_p(hi[k], v[0]), max_p(v[1], max_p(v[2], v[3])));
  };
  auto f = [&](float x) {
    if (act == 0) return __fmul_rn(x, __fdiv_rn(1.0f, __fadd_rn(1.0f, expf(-x))));
    return __fmul_rn(__fmul_rn(0.5f, x), __fadd_rn(1.0f, erff(__fmul_rn(x, 0.70710678118654752440f))));
  };
  const int64_t nvec = numel >> 2, stride = (int64_t)gridDim.x * 256;
  for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < nvec; i += stride) {
    const v4f va = __builtin_nontemporal_load(reinterpret_cast<const v4f*>(a) + i), vb = __builtin_nontemporal_load(reinterpret_cast<const v4f*>(b) + i);
    const v4f vs = {f(va[0]), f(va[1]), f(va[2]), f(va[3])};
    const v4f vp = vs * vb;
    upd(0, va);
    upd(1, vs);
    upd(2, vb);
    upd(3, vp);
    reinterpret_cast<v4f*>(out)[i] = vp;
  }
#pragma unroll
  for (int k = 0; k < 4; ++k) {
    block_commit(lo[k], hi[k], st.p[2 * k], st.p[2 * k + 1]);
    __syncthreads();
  }
}


// RoPE on the q and k projections of one attention block with the four statistics around it (hf_model.py:486-501 between the hooks of
// q_proj / k_proj (outputs) and qk_bmm (input, input2)): x [B, S, heads * D] as the linear wrote it -> [B, heads, S, D] contiguous,
// out[d] = x[d] * cos[s][d] + rot[d] * sin[s][d] for d < rot (rot[d] = -x[d + rot / 2] | x[d - rot / 2]), x[d] beyond -- two rounded
// products and a rounded sum, as torch's three launches: the same bits.  One thread per 4 consecutive d; the partner quad is a second
// (cached) load.  Segment 0 = q (heads[0] heads), segment 1 = k (heads[1]).
struct CalibRopeArgs {
  const float* x[3];
  float* out[3];
  int heads[3];                  // heads of the INPUT of a segment (q: heads, k / v: kv_heads)
  int rep[3];                    // every input head is written to `rep` consecutive output heads (repeat_kv, hf_model.py:509-510)
  int rotate[3];                 // 0: the segment passes through (v)
  int nseg;
  long long quads[3];            // B * S * heads * D / 4
  int S, D, rot, dq_shift;       // dq_shift: log2(D / 4) or -1
  const float* cos;              // [S, rot]
  const float* sin;
  float* st[12];                 // {min, max} of x[s], out[s] per segment (an out pair may be NULL)
};
__global__ void __launch_bounds__(256) calib_rope_kernel(const CalibRopeArgs a) {
  typedef float v4f __attribute__((ext_vector_type(4)));
  const float pinf = __int_as_float(0x7f800000), ninf = __int_as_float(0xff800000);
  float lo[6] = {pinf, pinf, pinf, pinf, pinf, pinf}, hi[6] = {ninf, ninf, ninf, ninf, ninf, ninf};
  auto upd = [&](int k, const v4f v) {
    lo[k] = min_p(min_p(lo[k], v[0]), min_p(v[1], min_p(v[2], v[3])));
    hi[k] = max_p(max_p(hi[k], v[0]), max_p(v[1], max_p(v[2], v[3])));
  };
  // a workgroup per (batch, position) row of the projections; thread -> (head slot threadIdx.x / dq, quad threadIdx.x % dq) with dq = D / 4
  // a launch constant the host passes as a shift when it is a power of two (no integer division per quad: 14 -> 9 us)
  const int dq = a.D >> 2, half = a.rot >> 1;
  const int tq = a.dq_shift >= 0 ? (int)(threadIdx.x & (dq - 1)) : (int)(threadIdx.x % dq);
  const int th = a.dq_shift >= 0 ? (int)(threadIdx.x >> a.dq_shift) : (int)(threadIdx.x / dq);
  const int hstep = 256 / dq;                                       // heads covered per sweep of the workgroup (host: dq <= 256)
  const int d = tq * 4;
  const long long nrows = a.quads[0] / ((long long)a.heads[0] * dq);          // batch * S
  for (long long bs = blockIdx.x; bs < nrows; bs += gridDim.x) {
    const int s_ = (int)(bs % a.S);
    const long long b = bs / a.S;
    v4f c = {0, 0, 0, 0}, sn = {0, 0, 0, 0};
    if (d < a.rot && th < hstep) {
      c = *reinterpret_cast<const v4f*>(a.cos + (long long)s_ * a.rot + d);
      sn = *reinterpret_cast<const v4f*>(a.sin + (long long)s_ * a.rot + d);
    }
#pragma unroll
    for (int seg = 0; seg < 3; ++seg) {
      if (seg >= a.nseg) break;
      const int H = a.heads[seg], rep = a.rep[seg], HO = H * rep;
      const v4f* x = reinterpret_cast<const v4f*>(a.x[seg]) + bs * H * dq;
      for (int h = th; h < H; h += hstep) {
        if (th >= hstep) break;                                     // (256 % dq != 0: the last partial head slot idles)
        const int i = h * dq + tq;
        const v4f v = x[i];
        upd(2 * seg, v);
        v4f o = v;
        if (a.rotate[seg] && d < a.rot) {
          v4f p = x[d < half ? i + (half >> 2) : i - (half >> 2)];
          if (d < half) p = -p;
#pragma unroll
          for (int e = 0; e < 4; ++e) o[e] = __fadd_rn(__fmul_rn(v[e], c[e]), __fmul_rn(p[e], sn[e]));
        }
        upd(2 * seg + 1, o);
        for (int j = 0; j < rep; ++j)
          reinterpret_cast<v4f*>(a.out[seg])[(((b * HO + h * rep + j) * a.S + s_) * (long long)a.D + d) >> 2] = o;
      }
    }
  }
#pragma unroll
  for (int k = 0; k < 6; ++k) {
    if (k < 2 * a.nseg && a.st[2 * k]) block_commit(lo[k], hi[k], a.st[2 * k], a.st[2 * k + 1]);
    __syncthreads();
  }
}

}  // namespace mq

using namespace mq;

extern "C" {

int mq_minmax_init(float* min_out, float* max_out, int64_t n, mq_stream_t stream) {
  MQ_REQUIRE(min_out && max_out && n >= 0, "mq_minmax_init: bad arguments");
  if (n == 0) return MQ_OK;
  minmax_init_kernel<<<(unsigned)((n + 255) / 256), 256, 0, as_stream(stream)>>>(min_out, max_out, n);
  MQ_LAUNCH_CHECK("mq_minmax_init");
  return MQ_OK;
}

int mq_minmax_tensor(const void* x, int dtype, int64_t numel, float* min_out, float* max_out, mq_stream_t stream) {
  MQ_REQUIRE(min_out && max_out && numel >= 0, "mq_minmax_tensor: bad arguments");
  if (numel == 0) return MQ_OK;   // empty input leaves the running statistic untouched
  MQ_REQUIRE(x, "mq_minmax_tensor: null input");
  if (dtype == MQ_F32) return launch_tensor<float>((const float*)x, numel, min_out, max_out, as_stream(stream));
  if (dtype == MQ_F16) return launch_tensor<__half>((const __half*)x, numel, min_out, max_out, as_stream(stream));
  set_error("mq_minmax_tensor: dtype %d not supported", dtype);
  return MQ_EUNSUPPORTED;
}

int mq_minmax_tensor_fresh(const void* x, int dtype, int64_t numel, float* min_out, float* max_out, float* scratch,
                           int64_t scratch_floats, mq_stream_t stream) {
  MQ_REQUIRE(min_out && max_out && numel >= 0, "mq_minmax_tensor_fresh: bad arguments");
  if (numel == 0) return mq_minmax_init(min_out, max_out, 1, stream);   // empty tensor: the identity statistic
  MQ_REQUIRE(x && scratch && scratch_floats >= 2 * kFreshMaxBlocks, "mq_minmax_tensor_fresh: needs %d floats of scratch",
             2 * kFreshMaxBlocks);
  if (dtype == MQ_F32) return launch_tensor_fresh<float>((const float*)x, numel, min_out, max_out, scratch, as_stream(stream));
  if (dtype == MQ_F16) return launch_tensor_fresh<__half>((const __half*)x, numel, min_out, max_out, scratch, as_stream(stream));
  set_error("mq_minmax_tensor_fresh: dtype %d not supported", dtype);
  return MQ_EUNSUPPORTED;
}

int mq_minmax_rows(const void* x, int dtype, int64_t rows, int64_t cols, float* min_out, float* max_out,
                   mq_stream_t stream) {
  MQ_REQUIRE(min_out && max_out && rows >= 0 && cols >= 0, "mq_minmax_rows: bad arguments");
  if (rows == 0 || cols == 0) return MQ_OK;
  MQ_REQUIRE(x, "mq_minmax_rows: null input");
  MQ_REQUIRE((rows + 3) / 4 < (int64_t)0x7fffffff, "mq_minmax_rows: too many rows");
  const unsigned grid = (unsigned)((rows + 3) / 4);
  if (dtype == MQ_F32) {
    const int v = aligned(x, 16) && cols % 4 == 0;
    minmax_rows_kernel<float><<<grid, 256, 0, as_stream(stream)>>>((const float*)x, rows, cols, v, min_out, max_out);
  } else if (dtype == MQ_F16) {
    const int v = aligned(x, 16) && cols % 8 == 0;
    minmax_rows_kernel<__half><<<grid, 256, 0, as_stream(stream)>>>((const __half*)x, rows, cols, v, min_out, max_out);
  } else {
    set_error("mq_minmax_rows: dtype %d not supported", dtype);
    return MQ_EUNSUPPORTED;
  }
  MQ_LAUNCH_CHECK("mq_minmax_rows");
  return MQ_OK;
}

int mq_minmax_cols(const void* x, int dtype, int64_t rows, int64_t cols, float* min_out, float* max_out,
                   mq_stream_t stream) {
  MQ_REQUIRE(min_out && max_out && rows >= 0 && cols >= 0, "mq_minmax_cols: bad arguments");
  if (rows == 0 || cols == 0) return MQ_OK;
  MQ_REQUIRE(x, "mq_minmax_cols: null input");
  if (dtype == MQ_F32) return launch_cols<float>((const float*)x, rows, cols, min_out, max_out, as_stream(stream));
  if (dtype == MQ_F16) return launch_cols<__half>((const __half*)x, rows, cols, min_out, max_out, as_stream(stream));
  set_error("mq_minmax_cols: dtype %d not supported", dtype);
  return MQ_EUNSUPPORTED;
}

int mq_calib_attention_probs(const float* raw, float* probs, int64_t rows, int64_t cols, const float* mask, int64_t mask_rows, double sqrt_d,
                             float* raw_min, float* raw_max, float* probs_min, float* probs_max, mq_stream_t stream) {
  const char* fn = "mq_calib_attention_probs";
  MQ_REQUIRE(rows >= 0 && cols >= 0, "%s: negative shape", fn);
  if (rows == 0 || cols == 0) return MQ_OK;
  MQ_REQUIRE(raw && probs && raw_min && raw_max && probs_min && probs_max, "%s: null pointer", fn);
  MQ_REQUIRE(sqrt_d > 0.0, "%s: sqrt_d=%g", fn, sqrt_d);
  MQ_REQUIRE(mask == nullptr || (mask_rows >= 1 && rows % mask_rows == 0), "%s: rows=%lld is not a multiple of mask_rows=%lld", fn,
             (long long)rows, (long long)mask_rows);
  if (cols % 4 != 0 || cols > 4096 || !aligned(raw, 16) || !aligned(probs, 16) || (mask && !aligned(mask, 16))) {
    set_error("%s: rows of up to 4096 keys, a multiple of 4, 16-byte aligned (cols=%lld)", fn, (long long)cols);
    return MQ_EUNSUPPORTED;
  }
  hipStream_t st = as_stream(stream);
  int64_t grid = (rows + 3) / 4;
  if (grid > 2048) grid = 2048;
  const float inv = 1.0f / (float)sqrt_d;              // torch's CUDA div-by-scalar: tensor * (accscalar_t(1) / scalar), accscalar_t = float
#define MQ_CALIB(V) calib_probs_kernel<V><<<(unsigned)grid, 256, 0, st>>>(raw, probs, rows, (int)cols, mask, (int)(mask ? mask_rows : 1), inv, \
                                                                         raw_min, raw_max, probs_min, probs_max)
  if (cols <= 256) MQ_CALIB(1);
  else if (cols <= 512) MQ_CALIB(2);
  else if (cols <= 1024) MQ_CALIB(4);
  else if (cols <= 2048) MQ_CALIB(8);
  else MQ_CALIB(16);
#undef MQ_CALIB
  MQ_LAUNCH_CHECK(fn);
  return MQ_OK;
}

int mq_calib_attention_probs_causal(const float* raw, float* probs, int64_t rows, int64_t seq, double sqrt_d, int store_masked, float* raw_min,
                                    float* raw_max, float* probs_min, float* probs_max, mq_stream_t stream) {
  const char* fn = "mq_calib_attention_probs_causal";
  MQ_REQUIRE(rows >= 0 && seq >= 0, "%s: negative shape", fn);
  if (rows == 0 || seq == 0) return MQ_OK;
  MQ_REQUIRE(raw && probs && raw_min && raw_max && probs_min && probs_max, "%s: null pointer", fn);
  MQ_REQUIRE(sqrt_d > 0.0 && rows % seq == 0, "%s: sqrt_d=%g, rows=%lld of square blocks of %lld", fn, sqrt_d, (long long)rows, (long long)seq);
  MQ_REQUIRE(store_masked || raw != probs, "%s: skipping the masked stores needs an output buffer of its own", fn);
  if (seq % 4 != 0 || seq > 4096 || !aligned(raw, 16) || !aligned(probs, 16)) {
    set_error("%s: square blocks of up to 4096 keys, a multiple of 4, 16-byte aligned (seq=%lld)", fn, (long long)seq);
    return MQ_EUNSUPPORTED;
  }
  hipStream_t st = as_stream(stream);
  // whole blocks per trip of the grid (4 rows per workgroup, seq % 4 == 0): the kernel reverses the row order of odd trips
  const int64_t blocks = rows / seq;
  int64_t per_trip = 8192 / seq < 1 ? 1 : 8192 / seq;
  if (per_trip > blocks) per_trip = blocks;
  const int64_t grid = per_trip * seq / 4;
  const float inv = 1.0f / (float)sqrt_d;
#define MQ_CALIBC(V) calib_probs_kernel<V, true><<<(unsigned)grid, 256, 0, st>>>(raw, probs, rows, (int)seq, nullptr, (int)seq, inv, raw_min, raw_max, \
                                                                                 probs_min, probs_max, store_masked)
  if (seq <= 256) MQ_CALIBC(1);
  else if (seq <= 512) MQ_CALIBC(2);
  else if (seq <= 1024) MQ_CALIBC(4);
  else if (seq <= 2048) MQ_CALIBC(8);
  else MQ_CALIBC(16);
#undef MQ_CALIBC
  MQ_LAUNCH_CHECK(fn);
  return MQ_OK;
}

int mq_calib_norm(const float* x, const float* delta, float* h_out, float* y_out, int64_t rows, int64_t cols, const float* weight, const float* bias,
                  float eps, int layernorm, float* in_min, float* in_max, float* out_min, float* out_max, float* delta_min, float* delta_max,
                  mq_stream_t stream) {
  const char* fn = "mq_calib_norm";
  MQ_REQUIRE(rows >= 0 && cols > 0, "%s: bad shape", fn);
  if (rows == 0) return MQ_OK;
  MQ_REQUIRE(x && y_out && weight && in_min && in_max && out_min && out_max, "%s: null pointer", fn);
  MQ_REQUIRE(!delta || h_out, "%s: a residual needs h_out", fn);
  MQ_REQUIRE((delta_min == nullptr) == (delta_max == nullptr) && (!delta_min || delta), "%s: the residual's statistic needs both slots and a residual", fn);
  if (cols % 4 != 0 || cols > 8192 || !aligned(x, 16) || !aligned(y_out, 16) || !aligned(weight, 16) || (delta && !aligned(delta, 16)) ||
      (h_out && !aligned(h_out, 16)) || (bias && !aligned(bias, 16))) {
    set_error("%s: not served: cols %% 4 == 0, cols <= 8192, 16-byte aligned pointers", fn);
    return MQ_EUNSUPPORTED;
  }
  hipStream_t st = as_stream(stream);
  const unsigned grid = (unsigned)(rows < 8192 ? rows : 8192);
#define MQ_CN(V)                                                                                                                     \
  do {                                                                                                                               \
    if (layernorm) calib_norm_kernel<V, true><<<grid, 256, 0, st>>>(x, delta, h_out, y_out, rows, (int)cols, weight, bias, eps, in_min, in_max, out_min, out_max, delta_min, delta_max); \
    else calib_norm_kernel<V, false><<<grid, 256, 0, st>>>(x, delta, h_out, y_out, rows, (int)cols, weight, bias, eps, in_min, in_max, out_min, out_max, delta_min, delta_max);      \
  } while (0)
  if (cols <= 1024) MQ_CN(1);
  else if (cols <= 2048) MQ_CN(2);
  else if (cols <= 4096) MQ_CN(4);
  else MQ_CN(8);
#undef MQ_CN
  MQ_LAUNCH_CHECK(fn);
  return MQ_OK;
}

int mq_calib_gated(const float* a, const float* b, float* out, int64_t numel, int act, float* const* stats, mq_stream_t stream) {
  const char* fn = "mq_calib_gated";
  MQ_REQUIRE(numel >= 0 && (act == 0 || act == 1), "%s: bad arguments", fn);
  if (numel == 0) return MQ_OK;
  MQ_REQUIRE(a && b && out && stats, "%s: null pointer", fn);
  if (numel % 4 != 0 || !aligned(a, 16) || !aligned(b, 16) || !aligned(out, 16)) {
    set_error("%s: not served: numel %% 4 == 0, 16-byte aligned pointers", fn);
    return MQ_EUNSUPPORTED;
  }
  CalibGatedStats cs;
  for (int k = 0; k < 8; ++k) {
    MQ_REQUIRE(stats[k], "%s: null statistic %d", fn, k);
    cs.p[k] = stats[k];
  }
  int64_t grid = (numel / 4 + 255) / 256;
  if (grid > 4096) grid = 4096;
  calib_gated_kernel<<<(unsigned)grid, 256, 0, as_stream(stream)>>>(a, b, out, numel, act, cs);
  MQ_LAUNCH_CHECK(fn);
  return MQ_OK;
}


static int launch_calib_rope(const char* fn, const float* const* in, float* const* out, int nseg, int64_t batch, int64_t seq, int heads, int kv_heads,
                             int head_dim, int rot_dim, int expand, const float* cos, const float* sin, float* const* stats, mq_stream_t stream) {
  MQ_REQUIRE(batch >= 0 && seq >= 0 && heads > 0 && kv_heads > 0 && head_dim > 0 && rot_dim > 0 && heads % kv_heads == 0, "%s: bad shape", fn);
  if (batch == 0 || seq == 0) return MQ_OK;
  MQ_REQUIRE(cos && sin && stats, "%s: null pointer", fn);
  bool ok = head_dim % 4 == 0 && head_dim <= 1024 && rot_dim <= head_dim && rot_dim % 8 == 0 && seq < (1ll << 31) && aligned(cos, 16) && aligned(sin, 16);
  for (int k = 0; k < nseg; ++k) {
    MQ_REQUIRE(in[k] && out[k], "%s: null pointer", fn);
    ok = ok && aligned(in[k], 16) && aligned(out[k], 16);
  }
  if (!ok) {
    set_error("%s: not served: head_dim %% 4 == 0, head_dim <= 1024, rot_dim %% 8 == 0, rot_dim <= head_dim, 16-byte aligned pointers", fn);
    return MQ_EUNSUPPORTED;
  }
  CalibRopeArgs a;
  a.nseg = nseg;
  for (int k = 0; k < 3; ++k) {
    a.x[k] = k < nseg ? in[k] : nullptr;
    a.out[k] = k < nseg ? out[k] : nullptr;
    a.heads[k] = k == 0 ? heads : kv_heads;
    a.rep[k] = (k == 0 || !expand) ? 1 : heads / kv_heads;
    a.rotate[k] = k < 2;
    a.quads[k] = batch * seq * a.heads[k] * head_dim / 4;
  }
  a.S = (int)seq; a.D = head_dim; a.rot = rot_dim; a.cos = cos; a.sin = sin;
  for (int k = 0; k < 12; ++k) a.st[k] = nullptr;
  for (int k = 0; k < 4 * nseg; ++k) a.st[k] = stats[k];
  for (int k = 0; k < nseg; ++k) MQ_REQUIRE(a.st[4 * k] && a.st[4 * k + 1] && (a.st[4 * k + 2] == nullptr) == (a.st[4 * k + 3] == nullptr), "%s: statistics of segment %d", fn, k);
  const int dq = head_dim / 4;
  a.dq_shift = (dq & (dq - 1)) == 0 ? __builtin_ctz((unsigned)dq) : -1;
  long long grid = batch * seq;
  if (grid > 8192) grid = 8192;
  calib_rope_kernel<<<(unsigned)grid, 256, 0, as_stream(stream)>>>(a);
  MQ_LAUNCH_CHECK(fn);
  return MQ_OK;
}

int mq_calib_rope(const float* q_in, const float* k_in, float* q_out, float* k_out, int64_t batch, int64_t seq, int heads, int kv_heads, int head_dim,
                  int rot_dim, const float* cos, const float* sin, float* const* stats, mq_stream_t stream) {
  const float* in[2] = {q_in, k_in};
  float* out[2] = {q_out, k_out};
  if (stats) for (int k = 0; k < 8; ++k) MQ_REQUIRE(stats[k], "mq_calib_rope: null statistic %d", k);
  return launch_calib_rope("mq_calib_rope", in, out, 2, batch, seq, heads, kv_heads, head_dim, rot_dim, 0, cos, sin, stats, stream);
}

int mq_calib_rope_qkv(const float* q_in, const float* k_in, const float* v_in, float* q_out, float* k_out, float* v_out, int64_t batch, int64_t seq, int heads,
                      int kv_heads, int head_dim, int rot_dim, const float* cos, const float* sin, float* const* stats, mq_stream_t stream) {
  const float* in[3] = {q_in, k_in, v_in};
  float* out[3] = {q_out, k_out, v_out};
  return launch_calib_rope("mq_calib_rope_qkv", in, out, 3, batch, seq, heads, kv_heads, head_dim, rot_dim, 1, cos, sin, stats, stream);
}

}  // extern "C"

// Shared helpers for the libmobilequant_amd translation units (gfx950 only).
#pragma once
#include <hip/hip_runtime.h>
#include <atomic>
#include <stdarg.h>
#include <stdint.h>
#include <stdio.h>

#include "mobilequant_amd.h"

namespace mq {

// Thread-local error text behind mq_last_error(); defined in mq_elementwise.hip.
void set_error(const char* fmt, ...);

inline hipStream_t as_stream(mq_stream_t s) { return reinterpret_cast<hipStream_t>(s); }

inline bool aligned(const void* p, size_t a) { return (reinterpret_cast<uintptr_t>(p) & (a - 1)) == 0; }

// The library may be driven on several devices of one process (a caller that switches devices, one model per GPU in one
// process): everything cached per kernel or per launch configuration is cached PER DEVICE.
constexpr int kMaxDevices = 64;
inline int current_device() {
  int d = 0;
  return (hipGetDevice(&d) == hipSuccess && d >= 0 && d < kMaxDevices) ? d : 0;
}
// once-per-device flag set (one bit per device; setting an attribute twice in a race is harmless)
struct PerDeviceOnce {
  std::atomic<unsigned long long> mask{0};
  bool done(int dev) const { return (mask.load(std::memory_order_acquire) >> dev) & 1ull; }
  void mark(int dev) { mask.fetch_or(1ull << dev, std::memory_order_release); }
};

#define MQ_REQUIRE(cond, ...)      \
  do {                             \
    if (!(cond)) {                 \
      mq::set_error(__VA_ARGS__);  \
      return MQ_EINVAL;            \
    }                              \
  } while (0)

#define MQ_LAUNCH_CHECK(name)                                                        \
  do {                                                                               \
    hipError_t e__ = hipGetLastError();                                              \
    if (e__ != hipSuccess) {                                                         \
      mq::set_error("%s: kernel launch failed: %s", name, hipGetErrorString(e__));   \
      return MQ_EHIP;                                                                \
    }                                                                                \
  } while (0)

// x / s through the correctly rounded reciprocal inv_s = RN(1 / s) and one fma correction (Markstein): bit-identical to the IEEE
// divide for |x / s| in [2^-2, 1e30] -- tools/div_check.cpp compares every fp32 dividend for 48 divisors on the GPU, incl. all-ones
// significands and the scale clamps 1e-5 / 1e6 (profiles/r03/div_check.log) -- and faithful below 2^-2, where round(x / s) = 0 and
// round_ste's (round(t) - t) + t = 0 whatever the last bit of t is: a quantizer INDEX (qmodule.py:286-287) can not tell the two
// apart.  +-inf and NaN dividends give NaN, which is what the reference's round_ste makes of them.  3 VALU instructions instead of
// the ~10 + two mode switches of v_div_scale / v_rcp / v_fma x4 / v_div_fmas / v_div_fixup: the quantize / norm kernels spend
// most of their issue slots on this division.  NOT for quotients that are themselves results (x / chan_scale, gradients).
__device__ __forceinline__ float div_by_scale(float x, float s, float inv_s) {
  const float q0 = __fmul_rn(x, inv_s);
  return __builtin_fmaf(__builtin_fmaf(-q0, s, x), inv_s, q0);
}
// The fast form needs a NORMAL reciprocal and no over / underflow in its intermediates.  tools/div_check.cpp sweeps every dividend for
// divisors drawn over +-[2^-60, 2^60] (profiles/r04/div_check.log): inside that range it is the IEEE quotient on the quantizer's
// domain; outside (a scale of 0, a denormal, inf, NaN, |s| beyond 2^+-60 -- nothing set_scale_offset_from_minmax's [1e-5, 1e6] clamp
// (qmodule.py:58) produces, but a trained scale parameter or a C-ABI caller is not bound by it) the reciprocal is inf / denormal and
// the fast form returns NaN where x / s is +-inf or finite.  The PUBLIC element-wise entry points (mq_fake_quant / mq_quantize and the
// per-row weight grids of training: HBM-bound kernels, the select is free) therefore take the IEEE divide for such a scale
// (div_by_scale_guarded, a per-tensor / per-row uniform choice).  The fused image kernels (norm, GEMV, decode, attention, tiled
// quantize) are reached through static calibrated grids only and keep the unguarded form: with a scale outside the range their
// indices saturate to qmin (NaN -> qmin) where the reference saturates to qmin or qmax -- a degenerate grid either way
// (dequantised values are (q - o) * s with s = 0 / inf / NaN).
__device__ __forceinline__ bool scale_in_fast_range(float s) {
  const float a = __builtin_fabsf(s);
  return a >= 0x1p-60f && a <= 0x1p60f;          // false for NaN
}
__device__ __forceinline__ float div_by_scale_guarded(float x, float s, float inv_s, bool fast) {
  return fast ? div_by_scale(x, s, inv_s) : __fdiv_rn(x, s);
}

// Four activations -> the dword of their int8 image bytes (index - shift), for the image-only kernels.  index = clamp(rint(x / s) + o):
// (rint(t) - t) + t, the reference's round_ste, IS rint(t) in fp32 for every t (|t| >= 0.5: rint(t) and t are within a factor of two,
// the difference is exact and adding t back lands on the representable rint(t); |t| < 0.5: (0 - t) + t = 0; x = +-inf / NaN: t is
// already NaN, div_by_scale).  The clamp is one v_med3_f32, which returns min3 when an operand is a (quiet) NaN: NaN -> qmin, the
// integer image's convention.  u = index + (128 - shift) lies in [0, 255] (the host checks that index - shift fits int8), so
// v_cvt_pk_u8_f32 converts AND packs in one instruction; the int8 bytes are u ^ 0x80 and sum(index - shift) = sum(u) - 128 n with
// sum(u) from one v_sad_u8 per dword (`usum` accumulates it).  ~8.5 VALU instructions per element instead of ~16.
__device__ __forceinline__ float image_idxf(float x, float s, float inv_s, float o, float qmin, float qmax) {
  const float t = div_by_scale(x, s, inv_s);
  return __builtin_amdgcn_fmed3f(__fadd_rn(rintf(t), o), qmin, qmax);
}
__device__ __forceinline__ float image_u8f(float x, float s, float inv_s, float o, float qmin, float qmax, float bias) {
  return __fadd_rn(image_idxf(x, s, inv_s, o, qmin, qmax), bias);
}
// Two elements per instruction: v_pk_mul_f32 / v_pk_fma_f32 / v_pk_add_f32 are IEEE fp32 operations on register pairs (full rate on
// CDNA3 / 4), so the packed forms below return the bits of the scalar ones; rint, med3 and the u8 conversion have no packed form.
typedef float v2f __attribute__((ext_vector_type(2)));
__device__ __forceinline__ v2f splat2(float v) { return (v2f)(v); }
__device__ __forceinline__ v2f div_by_scale2(v2f x, float s, float inv_s) {
  const v2f q0 = x * splat2(inv_s);
  return __builtin_elementwise_fma(__builtin_elementwise_fma(-q0, splat2(s), x), splat2(inv_s), q0);
}
__device__ __forceinline__ v2f image_u8f2(v2f x, float s, float inv_s, float o, float qmin, float qmax, float bias) {
  const v2f t = div_by_scale2(x, s, inv_s);
  v2f r = {rintf(t.x), rintf(t.y)};
  r = r + splat2(o);
  r.x = __builtin_amdgcn_fmed3f(r.x, qmin, qmax);
  r.y = __builtin_amdgcn_fmed3f(r.y, qmin, qmax);
  return r + splat2(bias);
}
__device__ __forceinline__ uint32_t image_pack4(float u0, float u1, float u2, float u3, uint32_t& usum) {
  uint32_t pk = __builtin_amdgcn_cvt_pk_u8_f32(u0, 0u, 0u);
  pk = __builtin_amdgcn_cvt_pk_u8_f32(u1, 1u, pk);
  pk = __builtin_amdgcn_cvt_pk_u8_f32(u2, 2u, pk);
  pk = __builtin_amdgcn_cvt_pk_u8_f32(u3, 3u, pk);
  usum = __builtin_amdgcn_sad_u8(pk, 0u, usum);
  return pk ^ 0x80808080u;
}

// 64-lane wave reductions on DPP moves (wave = 64 on gfx950): quad permutes, row_half_mirror, row_mirror leave every lane of a
// 16-lane row with the row's result; the four rows meet through v_readlane.  A __shfl_xor is a ds_bpermute -- an LDS round trip
// of ~100 cycles -- and six dependent ones cost a short kernel more than its arithmetic (decode: 0.25 us of a 3 us launch).  The
// result is wave-uniform.  (Float sums: another association than the xor butterfly -- every caller's tolerance covers the order
// of a row reduction; min / max are exact either way.)
template <int CTRL>
__device__ __forceinline__ int dpp_mov_i(int v) { return __builtin_amdgcn_update_dpp(0, v, CTRL, 0xf, 0xf, true); }
template <int CTRL>
__device__ __forceinline__ float dpp_mov_f(float v) { return __builtin_bit_cast(float, dpp_mov_i<CTRL>(__builtin_bit_cast(int, v))); }
__device__ __forceinline__ float readlane_f(float v, int l) { return __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, v), l)); }
template <class Op>
__device__ __forceinline__ float wave_reduce_f(float v, Op op) {
  v = op(v, dpp_mov_f<0xB1>(v));                                    // quad_perm [1,0,3,2]
  v = op(v, dpp_mov_f<0x4E>(v));                                    // quad_perm [2,3,0,1]
  v = op(v, dpp_mov_f<0x141>(v));                                   // row_half_mirror
  v = op(v, dpp_mov_f<0x140>(v));                                   // row_mirror
  return op(op(readlane_f(v, 15), readlane_f(v, 31)), op(readlane_f(v, 47), readlane_f(v, 63)));
}
__device__ __forceinline__ float wave_min(float v) {
  return wave_reduce_f(v, [](float a, float b) { return fminf(a, b); });
}
__device__ __forceinline__ float wave_max(float v) {
  return wave_reduce_f(v, [](float a, float b) { return fmaxf(a, b); });
}
__device__ __forceinline__ float wave_sum_f32_dpp(float v) {
  return wave_reduce_f(v, [](float a, float b) { return a + b; });
}
__device__ __forceinline__ int wave_sum(int v) {
  v += dpp_mov_i<0xB1>(v);
  v += dpp_mov_i<0x4E>(v);
  v += dpp_mov_i<0x141>(v);
  v += dpp_mov_i<0x140>(v);
  return (__builtin_amdgcn_readlane(v, 15) + __builtin_amdgcn_readlane(v, 31)) + (__builtin_amdgcn_readlane(v, 47) + __builtin_amdgcn_readlane(v, 63));
}

// Exact float atomic min/max on the IEEE bit pattern (no CAS loop): non-negative floats order as
// signed ints, negative floats order reversed as unsigned ints.
__device__ __forceinline__ void atomic_max_f32(float* addr, float v) {
  if (v >= 0.0f) atomicMax(reinterpret_cast<int*>(addr), __float_as_int(v));
  else atomicMin(reinterpret_cast<unsigned int*>(addr), __float_as_uint(v));
}
__device__ __forceinline__ void atomic_min_f32(float* addr, float v) {
  if (v >= 0.0f) atomicMin(reinterpret_cast<int*>(addr), __float_as_int(v));
  else atomicMax(reinterpret_cast<unsigned int*>(addr), __float_as_uint(v));
}

}  // namespace mq

// Exhaustive check of the two-correction reciprocal division against the IEEE divide on gfx950:
//   q0 = x * y;  r0 = fma(-q0, s, x);  q1 = fma(r0, y, q0);  r1 = fma(-q1, s, x);  q2 = fma(r1, y, q1),   y = RN(1 / s)
// for every finite fp32 x (2^32 bit patterns) and a list of divisors s: 48 in the quantizer's clamp range [1e-5, 1e6]
// (qmodule.py:58) and -- round 4 -- `wide` more drawn log-uniformly over +-[2^-60, 2^60], the range mq_common.h's
// scale_in_fast_range() admits to the one-correction form (div_by_scale).
//   usage: div_check [dividend_stride = 1] [wide = 48]      (stride 251: every 251st bit pattern, ~2 s, what the GPU test runs)
//   hipcc --offload-arch=gfx950 -O3 -ffp-contract=off -fhip-fp32-correctly-rounded-divide-sqrt tools/div_check.cpp -o tools/div_check
#include <hip/hip_runtime.h>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <random>
#include <vector>
#pragma clang fp contract(off)
__device__ __forceinline__ float div2(float x, float s, float y) {
  const float q0 = x * y;
  const float r0 = __builtin_fmaf(-q0, s, x);
  const float q1 = __builtin_fmaf(r0, y, q0);
  const float r1 = __builtin_fmaf(-q1, s, x);
  return __builtin_fmaf(r1, y, q1);
}
__device__ __forceinline__ float div1(float x, float s, float y) {
  const float q0 = x * y;
  const float r0 = __builtin_fmaf(-q0, s, x);
  return __builtin_fmaf(r0, y, q0);
}
__global__ void check(const float* scales, int n, unsigned long long* bad2, unsigned long long* bad1, unsigned* example, unsigned stride) {
  const int si = blockIdx.y;
  const float s = scales[si], y = __fdiv_rn(1.0f, s);
  unsigned long long b2 = 0, b1 = 0;
  for (unsigned long long i = (unsigned long long)blockIdx.x * blockDim.x + threadIdx.x; i * stride < (1ull << 32); i += (unsigned long long)gridDim.x * blockDim.x) {
    const float x = __uint_as_float((unsigned)(i * stride));
    if (!(fabsf(x) <= 3.0e38f)) continue;                          // finite only
    const float want = __fdiv_rn(x, s);
    // the quantizer's domain: |x / s| in [2^-2, 1e30].  Below, round(x / s) = 0 whatever the last bits of the quotient are (and
    // (round(t) - t) + t = 0 exactly); above, the index is clamped (qmax <= 65535) and every float is an integer anyway.
    if (!(fabsf(want) <= 1.0e30f) || fabsf(want) < 0.25f) continue;
    const float g2 = div2(x, s, y), g1 = div1(x, s, y);
    if (__float_as_uint(g2) != __float_as_uint(want)) {
      if (!b2) { example[2 * si] = (unsigned)(i * stride); example[2 * si + 1] = __float_as_uint(g2); }
      ++b2;
    }
    if (__float_as_uint(g1) != __float_as_uint(want)) ++b1;
  }
  if (b2) atomicAdd(&bad2[si], b2);
  if (b1) atomicAdd(&bad1[si], b1);
}
int main(int argc, char** argv) {
  const unsigned stride = argc > 1 ? (unsigned)atoi(argv[1]) : 1u;
  const int wide = argc > 2 ? atoi(argv[2]) : 48;
  std::vector<float> sc = {1e-5f, 1e6f, 1.0f, 0.0117647f, 0.015748f, 0.285714f, 3.0f, 7.0f, 0.1f, 1.9999999f, 1.0000001f, 1.5f, 65535.0f, 1.0f / 255.0f, 1.0f / 65535.0f};
  unsigned allones = 0x3F7FFFFFu;                                  // 0.99999994: significand all ones
  float f; memcpy(&f, &allones, 4); sc.push_back(f);
  allones = 0x3AFFFFFFu; memcpy(&f, &allones, 4); sc.push_back(f);
  std::mt19937 rng(1337);
  std::uniform_real_distribution<float> lg(-11.5f, 13.8f);         // ln(1e-5) .. ln(1e6)
  while (sc.size() < 48) sc.push_back(expf(lg(rng)));
  std::uniform_real_distribution<float> lg2(-60.0f, 60.0f);        // exponents of the wide range, both signs, plus its ends
  for (float e : {0x1p-60f, 0x1p60f, -0x1p-60f, -0x1p60f, 0x1.fffffep59f, 0x1.000002p-60f}) sc.push_back(e);
  for (int i = 0; i < wide; ++i) sc.push_back((i & 1 ? -1.0f : 1.0f) * exp2f(lg2(rng)) * (1.0f + (rng() & 0x7FFFFF) * 0x1p-23f) * 0.5f);
  for (float& v : sc) { if (fabsf(v) < 0x1p-60f) v = copysignf(0x1p-60f, v); if (fabsf(v) > 0x1p60f) v = copysignf(0x1p60f, v); }
  const int n = (int)sc.size();
  float* d_s; unsigned long long *d_b2, *d_b1; unsigned* d_ex;
  hipMalloc(&d_s, n * 4); hipMalloc(&d_b2, n * 8); hipMalloc(&d_b1, n * 8); hipMalloc(&d_ex, n * 8);
  hipMemcpy(d_s, sc.data(), n * 4, hipMemcpyHostToDevice); hipMemset(d_b2, 0, n * 8); hipMemset(d_b1, 0, n * 8); hipMemset(d_ex, 0, n * 8);
  check<<<dim3(2048, n), 256>>>(d_s, n, d_b2, d_b1, d_ex, stride ? stride : 1u);
  if (hipDeviceSynchronize() != hipSuccess) { printf("kernel failed\n"); return 1; }
  std::vector<unsigned long long> b2(n), b1(n); std::vector<unsigned> ex(2 * n);
  hipMemcpy(b2.data(), d_b2, n * 8, hipMemcpyDeviceToHost); hipMemcpy(b1.data(), d_b1, n * 8, hipMemcpyDeviceToHost); hipMemcpy(ex.data(), d_ex, n * 8, hipMemcpyDeviceToHost);
  unsigned long long t2 = 0, t1 = 0;
  for (int i = 0; i < n; ++i) {
    t2 += b2[i]; t1 += b1[i];
    if (b2[i] || b1[i] || i < 17 || (i >= 48 && i < 56)) printf("s = %-14.9g two corrections: %llu mismatches   one correction: %llu   (first x bits %08x)\n", sc[i], b2[i], b1[i], ex[2 * i]);
  }
  printf("TOTAL over %d divisors (48 in [1e-5, 1e6], %d over +-[2^-60, 2^60]) x every %u-th of 2^32 dividends: two corrections %llu mismatches, one correction %llu\n",
         n, n - 48, stride, t2, t1);
  return (t2 || t1) ? 2 : 0;
}

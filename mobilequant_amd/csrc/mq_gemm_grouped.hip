// QLinear with PER-GROUP weight grids on the integer path (round 4) -- the last configuration of mobilellm/quantization/qmodule.py's
// Quantizer (group_size != -1: the weight is viewed as [-1, group_size] and every group of `group_size` consecutive input channels of
// an output row has its own scale / offset, qmodule.py:259-260, :292-293; CLI --group_size, ptq/mobilequant.py:41, :157) that QLinear
// could only run simulated (fake-quant + fp32 library GEMM).
//
//   y[m, n] = sum_g  s_a s_w[n, g] * sum_{k in g} (ia[m, k] - z_a) (iw[n, k] - o_w[n, g])  + bias[n]
//
// With the stored bytes a' = ia - sh_a, w' = iw - sh_w the inner sum is an exact integer
//   P_g[m, n] + c_w[n, g] A_g[m] + T[n, g],     P_g = sum_{k in g} a' w'  (int8 MFMA),   A_g[m] = sum_{k in g} a'[m, k],
//   c_w = sh_w - o_w,   T = c_a W_g[n] + group_size c_a c_w   (c_a = sh_a - z_a, W_g = sum_{k in g} w'),
// folded into an fp32 accumulator once per group: acc_f += alpha[n, g] * float(...), alpha = s_a s_w[n, g].  The fold is the price
// of the recipe (one convert + one fma per output and GROUP, where the per-channel kernels pay them once per output): at
// group_size 128 a wave spends 64 x 4 VALU instructions per 32 MFMAs, so this kernel is VALU-bound by construction; it is written for
// exactness and a ~10 x margin over the simulated path, not for the MFMA roofline.
//
// Tile 128 x 128, eight waves (4 x 2, each 32 x 64 = 2 x 4 MFMA tiles of v_mfma_i32_16x16x64_i8), k-steps of 64 through a double-
// buffered LDS tile (rows padded to 80 bytes: the 16-byte fragment reads of 16 consecutive rows fall into different banks).
// Row-major int8 operands; group vectors are [G, N] / [G, M] (group-major), staged per group in the LDS.
#include <hip/hip_runtime.h>

#include "mobilequant_amd.h"
#include "mq_common.h"

namespace mq {

typedef int v4i __attribute__((ext_vector_type(4)));
typedef float v4f __attribute__((ext_vector_type(4)));

struct GroupedArgs {
  const int8_t* a;        // [M, K] stored activations (index - a_shift)
  const int8_t* w;        // [N, K] stored weights (index - w_shift)
  int M, N, K, gs;
  const int32_t* a_gsum;  // [G, M]
  const float* alpha;     // [G, N]
  const int32_t* cw;      // [G, N]
  const int32_t* t;       // [G, N]
  const float* bias;      // [N] or NULL
  float* out;             // [M, N]
};

constexpr int GT = 128;         // tile edge

// Eight waves, 4 (M) x 2 (N), each 32 x 64 = 2 x 4 MFMA tiles: 32 integer + 32 fp32 accumulator registers per lane; the group's
// vectors go through the LDS (read 16 bytes at a time at the fold) instead of living in 52 registers -- ~126 VGPRs, four waves per
// SIMD, so another workgroup's MFMAs and folds cover this one's global-load latency (the first version: 64 x 64 per wave, 254 VGPRs +
// 64 AGPRs, ONE wave per SIMD, 141 us at the headline shape; this layout with 64-byte steps 76-92 us: one global round trip per 8
// MFMAs).  BKS = bytes of K per LDS buffer and barrier: 128 when group_size % 128 == 0 (two MFMA k-steps per round trip), else 64.
template <int BKS>
__global__ void __launch_bounds__(512, 2) gemm_i8_grouped_kernel(const GroupedArgs g) {
  constexpr int GP = BKS + 16;                                            // LDS row pitch: the 16-byte reads of 16 consecutive rows hit different banks
  constexpr int SUB = BKS / 64;                                           // MFMA k-steps per buffer
  constexpr int PCS = BKS / 64;                                           // 16-byte pieces per thread and operand (128 rows x BKS bytes / 512 threads)
  __shared__ __attribute__((aligned(16))) int8_t lds[2][2][GT * GP];      // [buffer][A | W][row][pitch]
  __shared__ __attribute__((aligned(16))) float s_al[2][GT];              // [group parity][column of the tile]
  __shared__ __attribute__((aligned(16))) int s_cw[2][GT], s_tt[2][GT], s_ag[2][GT];
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int wm = wave >> 1, wn = wave & 1;
  const int tiles_n = g.N / GT;
  const int m0 = (blockIdx.x / tiles_n) * GT, n0 = (blockIdx.x % tiles_n) * GT;
  const int frow = lane & 15, fq = lane >> 4;
  const int KS = g.K / BKS, per_group = g.gs / 64;                        // buffers in K; MFMA k-steps per group
  // global -> LDS: piece p of a thread = 16 bytes at chunk (tid + 512 p) % (BKS / 16) of row (tid + 512 p) / (BKS / 16)
  constexpr int CPR = BKS / 16;
  int prow[PCS], pch[PCS];
  const int8_t* ap[PCS];
  const int8_t* wp[PCS];
#pragma unroll
  for (int p = 0; p < PCS; ++p) {
    const int piece = threadIdx.x + 512 * p;
    prow[p] = piece / CPR;
    pch[p] = piece % CPR;
    ap[p] = g.a + (size_t)(m0 + prow[p] < g.M ? m0 + prow[p] : g.M - 1) * g.K + pch[p] * 16;
    wp[p] = g.w + (size_t)(n0 + prow[p]) * g.K + pch[p] * 16;
  }
  const int vcol = threadIdx.x & 127, vsel = threadIdx.x >> 7;           // group vectors: thread -> (which vector, tile column / row)

  v4i acc[2][4];
  v4f accf[2][4];
#pragma unroll
  for (int i = 0; i < 2; ++i)
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      acc[i][j] = (v4i){0, 0, 0, 0};
      accf[i][j] = (v4f){0.f, 0.f, 0.f, 0.f};
    }
  auto group_vec = [&](int grp) -> int {                                  // this thread's element of group grp's vectors (as bits)
    if (vsel == 0) return __float_as_int(g.alpha[(size_t)grp * g.N + n0 + vcol]);
    if (vsel == 1) return g.cw[(size_t)grp * g.N + n0 + vcol];
    if (vsel == 2) return g.t[(size_t)grp * g.N + n0 + vcol];
    return g.a_gsum[(size_t)grp * g.M + (m0 + vcol < g.M ? m0 + vcol : g.M - 1)];
  };
  auto park_vec = [&](int par, int bits) {
    if (vsel == 0) s_al[par][vcol] = __int_as_float(bits);
    else if (vsel == 1) s_cw[par][vcol] = bits;
    else if (vsel == 2) s_tt[par][vcol] = bits;
    else s_ag[par][vcol] = bits;
  };
  // BKS = 64: a buffer is one k-step, a group >= 1 buffers.  BKS = 128: a buffer is two k-steps of ONE group (group_size % 128 == 0).
  // Either way at most one group ends per buffer, and the vectors of the group that opens in the next buffer are fetched a buffer ahead.
  const int bufs_per_group = g.gs / BKS;
  v4i ra[PCS], rw[PCS];
#pragma unroll
  for (int p = 0; p < PCS; ++p) {
    ra[p] = *reinterpret_cast<const v4i*>(ap[p]);
    rw[p] = *reinterpret_cast<const v4i*>(wp[p]);
  }
  int rv = group_vec(0);
#pragma unroll
  for (int p = 0; p < PCS; ++p) {
    *reinterpret_cast<v4i*>(&lds[0][0][prow[p] * GP + pch[p] * 16]) = ra[p];
    *reinterpret_cast<v4i*>(&lds[0][1][prow[p] * GP + pch[p] * 16]) = rw[p];
  }
  park_vec(0, rv);
  __syncthreads();
  for (int kb = 0; kb < KS; ++kb) {
    const int buf = kb & 1;
    const bool more = kb + 1 < KS;
    const int grp = kb / bufs_per_group, gpar = grp & 1;
    const bool fold = (kb + 1) % bufs_per_group == 0;
    const bool next_group = more && fold;                                 // the next buffer opens group grp + 1: fetch its vectors now
    if (more) {
#pragma unroll
      for (int p = 0; p < PCS; ++p) {
        ra[p] = *reinterpret_cast<const v4i*>(ap[p] + (size_t)(kb + 1) * BKS);
        rw[p] = *reinterpret_cast<const v4i*>(wp[p] + (size_t)(kb + 1) * BKS);
      }
      if (next_group) rv = group_vec(grp + 1);
    }
#pragma unroll
    for (int sub = 0; sub < SUB; ++sub) {
      v4i fa[2], fw[4];
#pragma unroll
      for (int i = 0; i < 2; ++i) fa[i] = *reinterpret_cast<const v4i*>(&lds[buf][0][(32 * wm + 16 * i + frow) * GP + sub * 64 + fq * 16]);
#pragma unroll
      for (int j = 0; j < 4; ++j) fw[j] = *reinterpret_cast<const v4i*>(&lds[buf][1][(64 * wn + 16 * j + frow) * GP + sub * 64 + fq * 16]);
#pragma unroll
      for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int j = 0; j < 4; ++j)   // D[n][m]: lane owns n = 4 fq + e (e = 0..3) of W fragment j for row m = frow of X fragment i
          acc[i][j] = __builtin_amdgcn_mfma_i32_16x16x64_i8(fw[j], fa[i], acc[i][j], 0, 0, 0);
    }
    if (fold) {
      const int ag0 = s_ag[gpar][32 * wm + frow], ag1 = s_ag[gpar][32 * wm + 16 + frow];
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        const int c = 64 * wn + 16 * j + 4 * fq;
        const v4f al = *reinterpret_cast<const v4f*>(&s_al[gpar][c]);
        const v4i cw = *reinterpret_cast<const v4i*>(&s_cw[gpar][c]);
        const v4i tt = *reinterpret_cast<const v4i*>(&s_tt[gpar][c]);
#pragma unroll
        for (int e = 0; e < 4; ++e) {
          // |cw| < 2^9, |a_gsum| <= 128 group_size: a 24-bit multiply (full rate; v_mul_lo_u32 issues at a quarter of it)
          accf[0][j][e] = __builtin_fmaf((float)(acc[0][j][e] + __mul24(cw[e], ag0) + tt[e]), al[e], accf[0][j][e]);
          accf[1][j][e] = __builtin_fmaf((float)(acc[1][j][e] + __mul24(cw[e], ag1) + tt[e]), al[e], accf[1][j][e]);
        }
        acc[0][j] = (v4i){0, 0, 0, 0};
        acc[1][j] = (v4i){0, 0, 0, 0};
      }
    }
    if (more) {
#pragma unroll
      for (int p = 0; p < PCS; ++p) {
        *reinterpret_cast<v4i*>(&lds[buf ^ 1][0][prow[p] * GP + pch[p] * 16]) = ra[p];
        *reinterpret_cast<v4i*>(&lds[buf ^ 1][1][prow[p] * GP + pch[p] * 16]) = rw[p];
      }
      if (next_group) park_vec(gpar ^ 1, rv);
      __syncthreads();
    }
  }
  // store: a lane's four consecutive n of one row as 16 bytes
#pragma unroll
  for (int i = 0; i < 2; ++i) {
    const int m = m0 + 32 * wm + 16 * i + frow;
    if (m >= g.M) continue;
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      const int n = n0 + 64 * wn + 16 * j + 4 * fq;
      v4f y = accf[i][j];
      if (g.bias) y = y + *reinterpret_cast<const v4f*>(g.bias + n);
      *reinterpret_cast<v4f*>(g.out + (size_t)m * g.N + n) = y;
    }
  }
}

}  // namespace mq

using namespace mq;

extern "C" int mq_w8a8_linear_grouped(const int8_t* a_q, const int8_t* w_q, int64_t M, int64_t N, int64_t K, int64_t group_size,
                                      const int32_t* a_gsum, const float* alpha, const int32_t* cw, const int32_t* t, const float* bias,
                                      float* out, mq_stream_t stream) {
  const char* fn = "mq_w8a8_linear_grouped";
  MQ_REQUIRE(M >= 0 && N > 0 && K > 0, "%s: bad shape M=%lld N=%lld K=%lld", fn, (long long)M, (long long)N, (long long)K);
  if (M == 0) return MQ_OK;
  MQ_REQUIRE(a_q && w_q && a_gsum && alpha && cw && t && out, "%s: null pointer", fn);
  MQ_REQUIRE(group_size <= 32768, "%s: group_size=%lld: the per-group activation sums must fit 24 bits", fn, (long long)group_size);
  MQ_REQUIRE(group_size > 0 && group_size % 64 == 0 && K % group_size == 0, "%s: group_size=%lld must be a multiple of 64 that divides K=%lld",
             fn, (long long)group_size, (long long)K);
  MQ_REQUIRE(N % 128 == 0, "%s: N=%lld must be a multiple of 128", fn, (long long)N);
  MQ_REQUIRE(M * K < (1ll << 31) && N * K < (1ll << 31) && M * N < (1ll << 31) && (K / group_size) * (M > N ? M : N) < (1ll << 31),
             "%s: operand too large", fn);
  MQ_REQUIRE(aligned(a_q, 16) && aligned(w_q, 16) && aligned(alpha, 16) && aligned(cw, 16) && aligned(t, 16) && aligned(out, 16) &&
                 (!bias || aligned(bias, 16)) && K % 16 == 0,
             "%s: pointers must be 16-byte aligned", fn);
  GroupedArgs g{a_q, w_q, (int)M, (int)N, (int)K, (int)group_size, a_gsum, alpha, cw, t, bias, out};
  const int64_t tiles = ((M + GT - 1) / GT) * (N / GT);
  if (group_size % 128 == 0) gemm_i8_grouped_kernel<128><<<(unsigned)tiles, 512, 0, as_stream(stream)>>>(g);
  else gemm_i8_grouped_kernel<64><<<(unsigned)tiles, 512, 0, as_stream(stream)>>>(g);
  MQ_LAUNCH_CHECK(fn);
  return MQ_OK;
}

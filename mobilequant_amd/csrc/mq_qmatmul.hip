// mq_qmatmul -- QMatMul.forward (reference: mobilellm/quantization/qmodule.py:453-466) as ONE launch on the int8 matrix pipe.
//
//   out = Qout( matmul( Qa(x1), Qb(x2) ) )          x1 [batch, M, K], x2 [batch, K, N] (either memory order), fp32
//
// The reference fake-quantises both operands (two elementwise passes each), multiplies them in fp32 (rocBLAS here) and fake-quantises
// the product (another two passes): for the attention block's qk_bmm / pv_bmm that is six passes over the [heads, S, T] tensor.  Here a
// workgroup owns a 64 x 64 output tile and walks K in 64-wide chunks: both operand chunks are loaded as fp32, quantised in registers
// with the reference's exact index arithmetic (mq_common.h: image_idxf = clamp(rint(x / s) + o)), stored as int8 (index - shift) in
// the LDS -- a 9 ... 16-bit first operand (pv_bmm's probabilities: ptq/mobilequant.py:198) as a high and a low byte plane -- and
// contracted with v_mfma_i32_16x16x64_i8.  Row sums of the stored A bytes and column sums of the stored B bytes come from two more
// MFMAs against an all-ones fragment, so the zero-point correction
//     sum_k (ia - za)(ib - zb) = sum a'b' + cb sum a' + ca sum b' + K ca cb        (a' = ia - sha, ca = sha - za, likewise b)
// is exact integer arithmetic (64 bit) and the product is rounded ONCE: float(t) * fl(sa sb) for <= 8-bit operands, one rounding of
// the double product for a 16-bit operand (the conventions of mq_attention.hip / mq_decode.hip and oracle.mq_oracle._qmatmul_exact).
// The output quantizer is the reference's expression op for op (IEEE quotient); the result is the fp32 tensor the module returns.
//
// HBM-bound by the [M, N] or [M, K] fp32 tensor (roofline: sizeof(float) (M K + K N + M N) per batch element); the integer
// contraction is there for exactness (every output is the exact quantised sum, where the fp32 library GEMM rounds per product) and to
// remove five of the six passes.  No mask or causality assumption, arbitrary M / N / K (an N-contiguous x2 needs N % 4 == 0: the host
// wrapper hands other widths over K-contiguous).
#include "mq_common.h"

namespace mq {

typedef int v4i __attribute__((ext_vector_type(4)));
typedef float v4f __attribute__((ext_vector_type(4)));

struct QmmArgs {
  const float* a;
  const float* b;
  float* out;
  int batch, M, N, K;
  long long a_bs, b_bs, o_bs;          // batch strides (elements)
  mq_grid ga, gb, go;                  // go.scale == NULL: no output quantizer
  int a_shift, b_shift;                // stored byte(s) = index - shift
  int tiles_m, tiles_n;
  long long nblk;
  int a_vec, b_vec, o_vec;             // 16-byte accesses allowed: K (N for the output) % 4 == 0 and a 16-byte aligned base
};

constexpr int QP = 80;                 // LDS row pitch in bytes: 64 k + 16 (ds_read_b128 of 16 rows lands on distinct bank groups)

// 4 fp32 -> the dword(s) of their stored bytes (index - shift); a 16-bit grid gives a low and a high byte plane.  Only the first
// `nlive` elements exist (k < K): the others become zero bytes, which contribute nothing to the products nor to the row / column sums.
template <bool W16>
__device__ __forceinline__ void quant4(const v4f x, int nlive, float s, float inv_s, float o, float qmin, float qmax, float bias, int& lo, int& hi) {
  const uint32_t keep = nlive >= 4 ? 0xffffffffu : (nlive <= 0 ? 0u : (1u << (8 * nlive)) - 1u);
  if constexpr (!W16) {
    uint32_t usum = 0;
    const uint32_t pk = image_pack4(image_u8f(x[0], s, inv_s, o, qmin, qmax, bias), image_u8f(x[1], s, inv_s, o, qmin, qmax, bias),
                                    image_u8f(x[2], s, inv_s, o, qmin, qmax, bias), image_u8f(x[3], s, inv_s, o, qmin, qmax, bias), usum);
    lo = (int)(pk & keep);
  } else {
    uint32_t l = 0, h = 0;
#pragma unroll
    for (int e = 0; e < 4; ++e) {
      // u = index - qmin in [0, 65535] (bias = -qmin); NaN -> 0 through the med3 clamp, like the 8-bit image
      const uint32_t u = (uint32_t)image_u8f(x[e], s, inv_s, o, qmin, qmax, bias);
      l |= (u & 255u) << (8 * e);
      h |= (u >> 8) << (8 * e);
    }
    lo = (int)((l ^ 0x80808080u) & keep);
    hi = (int)((h ^ 0x80808080u) & keep);
  }
}

// RF: 16-row fragments per wave = 64 RF rows per workgroup (round 6).  At RF = 1 (rounds 5) the 64 x 64 tile re-read 2 B of fp32 operand
// through the L2s per output byte: every 64-row block of pv_bmm quantised the WHOLE of x2 again (as many bytes as its own rows of
// x1), every tile of qk_bmm read two 16-KB operand tiles for one 16-KB output tile.  RF = 2 (128 rows): half of the x2 traffic and
// of its quantiser arithmetic per output row (RF = 4 needs 256 VGPRs: one workgroup per CU, slower again).  Wave w owns rows 64 r + 16 w + (0 .. 15), r < RF: every round r of the epilogue stages and
// stores 64 consecutive rows.
template <bool A16, bool BKM, int RF>
__global__ void __launch_bounds__(256) qmatmul_kernel(const QmmArgs g) {
  // one LDS block: the int8 operand tiles of the K loop, then (behind the loop's last barrier) the fp32 output tile, staged so that
  // the stores are whole 256-byte rows instead of the MFMA layout's 64-byte pieces of 16 different rows
  constexpr int OP = 68;                                         // output staging pitch in floats (272 B: float4 writes of 16 rows spread over the banks)
  constexpr int MT = 64 * RF;
  constexpr int A_BYTES = (A16 ? 2 : 1) * MT * QP, SMEM = 64 * OP * 4 > A_BYTES + 64 * QP ? 64 * OP * 4 : A_BYTES + 64 * QP;
  __shared__ __attribute__((aligned(16))) char smem[SMEM];
  char (*sA)[MT * QP] = reinterpret_cast<char (*)[MT * QP]>(smem);
  char* sB = smem + A_BYTES;
  float* sO = reinterpret_cast<float*>(smem);
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int frow = lane & 15, fq = lane >> 4;
  // XCD-aware order: consecutive logical tiles (n fastest: they share the A rows) go to ONE XCD's L2 (block b runs on XCD b % 8)
  long long lb = blockIdx.x;
  if (g.nblk % 8 == 0) lb = (lb % 8) * (g.nblk / 8) + lb / 8;
  const int tn = (int)(lb % g.tiles_n);
  const int tm = (int)((lb / g.tiles_n) % g.tiles_m);
  const int bi = (int)(lb / ((long long)g.tiles_n * g.tiles_m));
  const int m0 = tm * MT, n0 = tn * 64;
  const int M = g.M, N = g.N, K = g.K;
  const float* A = g.a + (long long)bi * g.a_bs;
  const float* B = g.b + (long long)bi * g.b_bs;

  auto uniform = [](float v) { return __builtin_bit_cast(float, __builtin_amdgcn_readfirstlane(__builtin_bit_cast(int, v))); };   // -> SGPR
  const float sa = uniform(g.ga.scale[0]), oa = uniform(g.ga.offset[0]), sb = uniform(g.gb.scale[0]), ob = uniform(g.gb.offset[0]);
  const float inv_sa = __fdiv_rn(1.0f, sa), inv_sb = __fdiv_rn(1.0f, sb);
  // 8-bit: stored u8 = index + (128 - shift), byte ^ 0x80 = index - shift.  16-bit: u = index - qmin, planes hi / lo ^ 0x80
  const float a_bias = A16 ? -g.ga.qmin : (float)(128 - g.a_shift);
  const float b_bias = (float)(128 - g.b_shift);

  // loader roles.  K-contiguous operand: thread -> row tid / 4, 16 consecutive k from (tid % 4) * 16.
  // N-contiguous B (x2 as [K, N]): thread -> 4 k (tid / 16 * 4 ..) x 4 n ((tid % 16) * 4 ..): four 16-byte loads along n, transposed in
  // registers into four dwords of 4 consecutive k each.
  const int lr = tid >> 2, lk = (tid & 3) * 16;
  const float* arow[RF];
#pragma unroll
  for (int rf = 0; rf < RF; ++rf) arow[rf] = A + (long long)min(m0 + 64 * rf + lr, M - 1) * K;
  const int bn = min(n0 + lr, N - 1);
  const int tk4 = (tid >> 4) * 4, tn4 = (tid & 15) * 4;
  const float* brow = BKM ? B + (long long)bn * K : B;

  v4f xa[RF][4], xb[4];
  // rows of K floats on a 16-byte aligned base: one dwordx4 per 4 k; else element loads
  auto load_k4 = [&](const float* row, int k, bool vec) -> v4f {
    v4f r = {0.f, 0.f, 0.f, 0.f};
    if (vec) {
      if (k < K) r = *reinterpret_cast<const v4f*>(row + k);
    } else {
#pragma unroll
      for (int e = 0; e < 4; ++e)
        if (k + e < K) r[e] = row[k + e];
    }
    return r;
  };
  auto load_chunk = [&](int k0) {
#pragma unroll
    for (int rf = 0; rf < RF; ++rf)
#pragma unroll
      for (int c = 0; c < 4; ++c) xa[rf][c] = load_k4(arow[rf], k0 + lk + 4 * c, g.a_vec != 0);
    if constexpr (BKM) {
#pragma unroll
      for (int c = 0; c < 4; ++c) xb[c] = load_k4(brow, k0 + lk + 4 * c, g.b_vec != 0);
    } else {
      const int n = min(n0 + tn4, N - 4);        // (N % 4 == 0 in this memory order: the host sends other widths K-contiguous)
#pragma unroll
      for (int c = 0; c < 4; ++c) {
        const int k = k0 + tk4 + c;
        xb[c] = k < K ? *reinterpret_cast<const v4f*>(brow + (long long)k * N + n) : (v4f){0.f, 0.f, 0.f, 0.f};
      }
    }
  };

  v4i acc_lo[RF][4], acc_hi[A16 ? RF : 1][4], cs[4], rs_lo[RF], rs_hi[A16 ? RF : 1];
#pragma unroll
  for (int j = 0; j < 4; ++j) cs[j] = (v4i){0, 0, 0, 0};
#pragma unroll
  for (int rf = 0; rf < RF; ++rf) {
    rs_lo[rf] = (v4i){0, 0, 0, 0};
    if (A16) rs_hi[rf] = (v4i){0, 0, 0, 0};
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      acc_lo[rf][j] = (v4i){0, 0, 0, 0};
      if (A16) acc_hi[rf][j] = (v4i){0, 0, 0, 0};
    }
  }
  const v4i ones = {0x01010101, 0x01010101, 0x01010101, 0x01010101};

  load_chunk(0);
  for (int k0 = 0; k0 < K; k0 += 64) {
    // quantise the chunk held in registers and park it in the LDS; k >= K contributes zero bytes (to the products AND to both sums)
#pragma unroll
    for (int rf = 0; rf < RF; ++rf) {
      v4i lo, hi;
#pragma unroll
      for (int c = 0; c < 4; ++c) {
        int l1, h1 = 0;
        quant4<A16>(xa[rf][c], K - (k0 + lk + 4 * c), sa, inv_sa, oa, g.ga.qmin, g.ga.qmax, a_bias, l1, h1);
        lo[c] = l1;
        hi[c] = h1;
      }
      *reinterpret_cast<v4i*>(&sA[0][(64 * rf + lr) * QP + lk]) = lo;
      if constexpr (A16) *reinterpret_cast<v4i*>(&sA[1][(64 * rf + lr) * QP + lk]) = hi;
    }
    if constexpr (BKM) {
      v4i lo;
#pragma unroll
      for (int c = 0; c < 4; ++c) {
        int l1, h1 = 0;
        quant4<false>(xb[c], K - (k0 + lk + 4 * c), sb, inv_sb, ob, g.gb.qmin, g.gb.qmax, b_bias, l1, h1);
        lo[c] = l1;
      }
      *reinterpret_cast<v4i*>(&sB[lr * QP + lk]) = lo;
    } else {
      // xb[c][e] = b[k0 + tk4 + c][n + e]: the dword of column e holds k = tk4 .. tk4 + 3
      uint32_t u[4][4];
#pragma unroll
      for (int c = 0; c < 4; ++c)
#pragma unroll
        for (int e = 0; e < 4; ++e)
          u[c][e] = (k0 + tk4 + c < K) ? ((uint32_t)image_u8f(xb[c][e], sb, inv_sb, ob, g.gb.qmin, g.gb.qmax, b_bias) ^ 0x80u) & 255u : 0u;
      const int nn = min(n0 + tn4, N - 4) - n0;              // the tile's last columns may be re-covered by a clamped thread: same bytes
#pragma unroll
      for (int e = 0; e < 4; ++e)
        *reinterpret_cast<uint32_t*>(&sB[(nn + e) * QP + tk4]) = u[0][e] | (u[1][e] << 8) | (u[2][e] << 16) | (u[3][e] << 24);
    }
    __syncthreads();
    if (k0 + 64 < K) load_chunk(k0 + 64);                    // next chunk's loads fly under this chunk's MFMAs
    v4i fa[RF], fah[A16 ? RF : 1];
#pragma unroll
    for (int rf = 0; rf < RF; ++rf) {
      fa[rf] = *reinterpret_cast<const v4i*>(&sA[0][(64 * rf + 16 * wave + frow) * QP + fq * 16]);
      if constexpr (A16) fah[rf] = *reinterpret_cast<const v4i*>(&sA[1][(64 * rf + 16 * wave + frow) * QP + fq * 16]);
      rs_lo[rf] = __builtin_amdgcn_mfma_i32_16x16x64_i8(ones, fa[rf], rs_lo[rf], 0, 0, 0);
      if constexpr (A16) rs_hi[rf] = __builtin_amdgcn_mfma_i32_16x16x64_i8(ones, fah[rf], rs_hi[rf], 0, 0, 0);
    }
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      const v4i fb = *reinterpret_cast<const v4i*>(&sB[(16 * j + frow) * QP + fq * 16]);
#pragma unroll
      for (int rf = 0; rf < RF; ++rf) {
        acc_lo[rf][j] = __builtin_amdgcn_mfma_i32_16x16x64_i8(fb, fa[rf], acc_lo[rf][j], 0, 0, 0);      // D[n][m]: lane holds n = 4 fq + e of row m = frow
        if constexpr (A16) acc_hi[rf][j] = __builtin_amdgcn_mfma_i32_16x16x64_i8(fb, fah[rf], acc_hi[rf][j], 0, 0, 0);
      }
      cs[j] = __builtin_amdgcn_mfma_i32_16x16x64_i8(fb, ones, cs[j], 0, 0, 0);
    }
    __syncthreads();
  }

  // epilogue: exact integer bracket -> one rounding -> the output quantizer (qmodule.py:286-290 op for op) -> fp32
  const float zaf = rintf(oa), zbf = rintf(ob);
  const bool sane = __builtin_fabsf(zaf) < 1048576.f && __builtin_fabsf(zbf) < 1048576.f;     // else: NaN out (a grid far from zero)
  const long long ca = (long long)g.a_shift - (long long)zaf, cb = (long long)g.b_shift - (long long)zbf;
  const float alpha = __fmul_rn(sa, sb);
  const bool has_q = g.go.scale != nullptr;
  const float so = has_q ? uniform(g.go.scale[0]) : 1.f, oo = has_q ? uniform(g.go.offset[0]) : 0.f;
  const float inv_so = __fdiv_rn(1.0f, so);
  const bool fast = scale_in_fast_range(so);
  // <= 8-bit operands whose whole bracket provably fits 32 bits (K (255 + |ca|)(255 + |cb|) < 2^31: every real grid) take 32-bit
  // integer arithmetic and ONE v_cvt_f32_i32 -- the same rounding of the same integer as the 64-bit / double route, which costs ~2 x the
  // instructions of this VALU-bound epilogue (wave-uniform choice)
  const bool fits32 = __builtin_amdgcn_readfirstlane(
      (int)(!A16 && (double)K * (255.0 + (double)(ca < 0 ? -ca : ca)) * (255.0 + (double)(cb < 0 ? -cb : cb)) < 2147483648.0)) != 0;
  float* obase = g.out + (long long)bi * g.o_bs;
#pragma unroll
  for (int rf = 0; rf < RF; ++rf) {
  if (m0 + 64 * rf >= M) break;                                  // (uniform) row blocks past the matrix
  const long long rs = A16 ? 256ll * rs_hi[A16 ? rf : 0][0] + rs_lo[rf][0] : (long long)rs_lo[rf][0];
  const int ca32 = (int)ca, row32 = (int)(cb * rs + (long long)K * ca * cb);
#pragma unroll
  for (int j = 0; j < 4; ++j) {
    v4f y;
#pragma unroll
    for (int e = 0; e < 4; ++e) {
      float v;
      if (fits32) {
        v = __fmul_rn((float)(acc_lo[rf][j][e] + ca32 * cs[j][e] + row32), alpha);
      } else {
        const long long p = A16 ? 256ll * acc_hi[A16 ? rf : 0][j][e] + acc_lo[rf][j][e] : (long long)acc_lo[rf][j][e];
        const long long t = p + cb * rs + ca * (long long)cs[j][e] + (long long)K * ca * cb;
        v = A16 ? (float)((double)t * (double)alpha) : __fmul_rn((float)(double)t, alpha);
      }
      if (!sane) v = __builtin_nanf("");
      if (has_q) {
        const float q = div_by_scale_guarded(v, so, inv_so, fast);
        const float r = __fadd_rn(__fsub_rn(rintf(q), q), q);              // round_ste (qmodule.py:17-19): +-inf / NaN -> NaN
        const float idx = fminf(fmaxf(__fadd_rn(r, oo), g.go.qmin), g.go.qmax);
        v = r != r ? r : __fmul_rn(__fsub_rn(idx, oo), so);                // torch.clamp propagates NaN, fminf / fmaxf drop it
      }
      y[e] = v;
    }
    *reinterpret_cast<v4f*>(&sO[(16 * wave + frow) * OP + 16 * j + 4 * fq]) = y;        // (rows of this wave only: no workgroup barrier)
  }
  // copy-out: the wave's 16 rows x 64 columns of this round, 16 lanes per row
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    const int idx = lane + 64 * i, r = idx >> 4, c4 = (idx & 15) * 4;
    const int mm = m0 + 64 * rf + 16 * wave + r, n = n0 + c4;
    const v4f y = *reinterpret_cast<const v4f*>(&sO[(16 * wave + r) * OP + c4]);
    if (mm < M) {
      float* orow = obase + (long long)mm * N;
      if (n + 4 <= N && g.o_vec) {
        *reinterpret_cast<v4f*>(orow + n) = y;
      } else {
#pragma unroll
        for (int e = 0; e < 4; ++e)
          if (n + e < N) orow[n + e] = y[e];
      }
    }
  }
  }
}

}  // namespace mq

using namespace mq;

extern "C" int mq_qmatmul(const float* x1, const float* x2, float* out, int64_t batch, int64_t M, int64_t N, int64_t K, int x2_k_contiguous,
                          const mq_grid* grid1, const mq_grid* grid2, const mq_grid* grid_out, mq_stream_t stream) {
  const char* fn = "mq_qmatmul";
  MQ_REQUIRE(batch >= 0 && M >= 0 && N >= 0 && K > 0, "%s: bad shape batch=%lld M=%lld N=%lld K=%lld", fn, (long long)batch, (long long)M,
             (long long)N, (long long)K);
  if (batch == 0 || M == 0 || N == 0) return MQ_OK;
  MQ_REQUIRE(x1 && x2 && out && grid1 && grid2, "%s: null pointer", fn);
  MQ_REQUIRE(aligned(x1, 4) && aligned(x2, 4) && aligned(out, 4), "%s: operands must be 4-byte aligned", fn);
  MQ_REQUIRE(x2_k_contiguous || aligned(x2, 16), "%s: an N-contiguous x2 must be 16-byte aligned", fn);
  MQ_REQUIRE(grid1->scale && grid1->offset && grid2->scale && grid2->offset, "%s: both operands need a (static per-tensor) grid", fn);
  MQ_REQUIRE(!grid_out || !grid_out->scale || grid_out->offset, "%s: output grid without an offset", fn);
  const double span1 = (double)grid1->qmax - (double)grid1->qmin, span2 = (double)grid2->qmax - (double)grid2->qmin;
  if (!(span1 >= 1 && span1 <= 65535 && span2 >= 1 && span2 <= 255) || (!x2_k_contiguous && N % 4 != 0) || K > 131071 ||
      M >= (1ll << 31) - 64 || N >= (1ll << 31) - 64 || batch >= (1ll << 31) || M * K >= (1ll << 40) || N * K >= (1ll << 40)) {
    set_error("%s: not served: grids of at most 16 (x1) / 8 (x2) bits, N %% 4 == 0 for an N-contiguous x2 (pass other widths K-contiguous), K <= 131071 (the int32 MFMA accumulators hold K x 128 x 128)", fn);
    return MQ_EUNSUPPORTED;
  }
  QmmArgs g;
  g.a = x1; g.b = x2; g.out = out;
  g.batch = (int)batch; g.M = (int)M; g.N = (int)N; g.K = (int)K;
  g.a_bs = M * K; g.b_bs = N * K; g.o_bs = M * N;
  g.ga = *grid1; g.gb = *grid2;
  g.go = (grid_out && grid_out->scale) ? *grid_out : mq_grid{nullptr, nullptr, 0.f, 0.f};
  const bool a16 = span1 > 255;
  // stored bytes: <= 8-bit grids as index - (qmin + 128); a wider x1 as the two byte planes of index - qmin, each minus 128
  g.a_shift = a16 ? (int)grid1->qmin + 32896 : (int)grid1->qmin + 128;
  g.b_shift = (int)grid2->qmin + 128;
  // 256-row tiles where that still gives every CU several workgroups (the x2 tile is quantised once per 256 rows instead of per 64)
  // 128-row tiles where that still gives every CU two workgroups: the x2 tile is fetched and quantised once per 128 rows instead of per 64
  // (S = 2048: qk_bmm 229 -> 202 us, pv_bmm 184 -> 152 us; 256-row tiles fall back to one workgroup per CU -- 256 VGPRs -- and lose it
  // again: 236 / 187 us, profiles/r06/bench_qmatmul.log)
  const int rf = (M >= 256 && batch * ((M + 127) / 128) * ((N + 63) / 64) >= 512) ? 2 : 1;
  g.tiles_m = (int)((M + 64 * rf - 1) / (64 * rf));
  g.tiles_n = (int)((N + 63) / 64);
  g.nblk = (long long)batch * g.tiles_m * g.tiles_n;
  g.a_vec = K % 4 == 0 && aligned(x1, 16);
  g.b_vec = x2_k_contiguous ? (K % 4 == 0 && aligned(x2, 16)) : 1;
  g.o_vec = N % 4 == 0 && aligned(out, 16);
  MQ_REQUIRE(g.nblk < (1ll << 31), "%s: too many tiles", fn);
  hipStream_t st = as_stream(stream);
  const dim3 grid((unsigned)g.nblk), block(256);
#define MQ_QMM_LAUNCH(A16, BKM)                                              \
  do {                                                                       \
    if (rf == 2) qmatmul_kernel<A16, BKM, 2><<<grid, block, 0, st>>>(g);     \
    else qmatmul_kernel<A16, BKM, 1><<<grid, block, 0, st>>>(g);             \
  } while (0)
  if (a16) {
    if (x2_k_contiguous) MQ_QMM_LAUNCH(true, true);
    else MQ_QMM_LAUNCH(true, false);
  } else {
    if (x2_k_contiguous) MQ_QMM_LAUNCH(false, true);
    else MQ_QMM_LAUNCH(false, false);
  }
#undef MQ_QMM_LAUNCH
  MQ_LAUNCH_CHECK(fn);
  return MQ_OK;
}

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

#include <algorithm>
#include <type_traits>

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
  int nsplit, tiles_per_split;          // row-panel kernel: column tiles per workgroup
  long long nblk;
  int a_vec, b_vec, o_vec;             // 16-byte accesses allowed: K (N for the output) % 4 == 0 and a 16-byte aligned base
};

constexpr int QP = 80;                 // LDS row pitch in bytes: 64 k + 16 (ds_read_b128 of 16 rows lands on distinct bank groups)

// 4 fp32 -> the dword(s) of their stored bytes (index - shift); a 16-bit grid gives a low and a high byte plane.  Only the first
// `nlive` elements exist (k < K): the others become zero bytes, which contribute nothing to the products nor to the row / column sums.
template <bool W16>
__device__ __forceinline__ void quant4(const v4f x, int nlive, float s, float inv_s, float o, float qmin, float qmax, float bias, int& lo, int& hi) {
  const uint32_t keep = nlive >= 4 ? 0xffffffffu : (nlive <= 0 ? 0u : (1u << (8 * nlive)) - 1u);
  // two elements per instruction where a packed fp32 form exists (mq_common.h: the bits of the scalar expressions)
  const v2f u01 = image_u8f2((v2f){x[0], x[1]}, s, inv_s, o, qmin, qmax, bias), u23 = image_u8f2((v2f){x[2], x[3]}, s, inv_s, o, qmin, qmax, bias);
  if constexpr (!W16) {
    uint32_t pk = __builtin_amdgcn_cvt_pk_u8_f32(u01.x, 0u, 0u);
    pk = __builtin_amdgcn_cvt_pk_u8_f32(u01.y, 1u, pk);
    pk = __builtin_amdgcn_cvt_pk_u8_f32(u23.x, 2u, pk);
    pk = __builtin_amdgcn_cvt_pk_u8_f32(u23.y, 3u, pk);
    lo = (int)((pk ^ 0x80808080u) & keep);
  } else {
    // u = index - qmin in [0, 65535] (bias = -qmin); NaN -> 0 through the med3 clamp, like the 8-bit image.  Two 16-bit halves per
    // dword, then one byte permute per plane
    const uint32_t p01 = (uint32_t)u01.x | ((uint32_t)u01.y << 16), p23 = (uint32_t)u23.x | ((uint32_t)u23.y << 16);
    const uint32_t l = __builtin_amdgcn_perm(p23, p01, 0x06040200u), h = __builtin_amdgcn_perm(p23, p01, 0x07050301u);
    lo = (int)((l ^ 0x80808080u) & keep);
    hi = (int)((h ^ 0x80808080u) & keep);
  }
}

// One 64-k chunk of the x2 tile (64 columns) from registers into the LDS as stored bytes, `klive` = K - k0 of them real.
// K-contiguous x2: thread -> row lr, 16 consecutive k from lk.  N-contiguous: thread holds 4 k (tk4 ..) x 4 n (tn4 ..), xb[c][e] =
// b[k0 + tk4 + c][n + e]: the dword of column e holds k = tk4 .. tk4 + 3.
template <bool BKM>
__device__ __forceinline__ void quantise_b(const v4f (&xb)[4], char* sB, int pitch, int klive, int lr, int lk, int tk4, int tn4, int n0, int N,
                                           float sb, float inv_sb, float ob, float qmin, float qmax, float b_bias) {
  if constexpr (BKM) {
    v4i lo;
#pragma unroll
    for (int c = 0; c < 4; ++c) {
      int l1, h1 = 0;
      quant4<false>(xb[c], klive - (lk + 4 * c), sb, inv_sb, ob, qmin, qmax, b_bias, l1, h1);
      lo[c] = l1;
    }
    *reinterpret_cast<v4i*>(&sB[lr * pitch + lk]) = lo;
  } else {
    const int nlive = klive - tk4;
    const uint32_t keep = nlive >= 4 ? 0xffffffffu : (nlive <= 0 ? 0u : (1u << (8 * nlive)) - 1u);
    uint32_t pk[4] = {0u, 0u, 0u, 0u};
#pragma unroll
    for (int c = 0; c < 4; ++c) {
      const v2f u01 = image_u8f2((v2f){xb[c][0], xb[c][1]}, sb, inv_sb, ob, qmin, qmax, b_bias), u23 = image_u8f2((v2f){xb[c][2], xb[c][3]}, sb, inv_sb, ob, qmin, qmax, b_bias);
      pk[0] = __builtin_amdgcn_cvt_pk_u8_f32(u01.x, (uint32_t)c, pk[0]);
      pk[1] = __builtin_amdgcn_cvt_pk_u8_f32(u01.y, (uint32_t)c, pk[1]);
      pk[2] = __builtin_amdgcn_cvt_pk_u8_f32(u23.x, (uint32_t)c, pk[2]);
      pk[3] = __builtin_amdgcn_cvt_pk_u8_f32(u23.y, (uint32_t)c, pk[3]);
    }
    const int nn = min(n0 + tn4, N - 4) - n0;              // the tile's last columns may be re-covered by a clamped thread: same bytes
#pragma unroll
    for (int e = 0; e < 4; ++e) *reinterpret_cast<uint32_t*>(&sB[(nn + e) * pitch + tk4]) = (pk[e] ^ 0x80808080u) & keep;
  }
}

// ---- the epilogue both kernels share: exact integer bracket -> one rounding -> the output quantizer (qmodule.py:286-290) -> fp32 ------
struct Epi {                            // wave-uniform
  float alpha, so, oo, inv_so, qmin, qmax;
  long long ca, cb, kcc;                // kcc = K ca cb
  bool has_q, fast, sane, fits32, plain;
};

__device__ __forceinline__ float uniformf(float v) { return __builtin_bit_cast(float, __builtin_amdgcn_readfirstlane(__builtin_bit_cast(int, v))); }   // -> SGPR

template <bool A16>
__device__ __forceinline__ Epi make_epi(const QmmArgs& g, float sa, float oa, float sb, float ob) {
  Epi c;
  const float zaf = rintf(oa), zbf = rintf(ob);
  c.sane = __builtin_fabsf(zaf) < 1048576.f && __builtin_fabsf(zbf) < 1048576.f;     // else: NaN out (a grid far from zero)
  c.ca = (long long)g.a_shift - (long long)zaf;
  c.cb = (long long)g.b_shift - (long long)zbf;
  c.kcc = (long long)g.K * c.ca * c.cb;
  c.alpha = __fmul_rn(sa, sb);
  c.has_q = g.go.scale != nullptr;
  c.so = c.has_q ? uniformf(g.go.scale[0]) : 1.f;
  c.oo = c.has_q ? uniformf(g.go.offset[0]) : 0.f;
  c.inv_so = __fdiv_rn(1.0f, c.so);
  c.fast = scale_in_fast_range(c.so);
  c.qmin = g.go.qmin;
  c.qmax = g.go.qmax;
  // <= 8-bit operands whose whole bracket provably fits 32 bits (K (255 + |ca|)(255 + |cb|) < 2^31: every real grid) take 32-bit
  // integer arithmetic and ONE v_cvt_f32_i32 -- the same rounding of the same integer as the 64-bit / double route, which costs ~2 x the
  // instructions of this VALU-bound epilogue (wave-uniform choice)
  c.fits32 = __builtin_amdgcn_readfirstlane(
      (int)(!A16 && (double)g.K * (255.0 + (double)(c.ca < 0 ? -c.ca : c.ca)) * (255.0 + (double)(c.cb < 0 ? -c.cb : c.cb)) < 2147483648.0)) != 0;
  // Every value of the chain is finite and every grid ordinary (all real grids): the quantizer's NaN / inf handling drops out --
  // (rint(q) - q) + q IS rint(q), fminf(fmaxf(.)) IS v_med3_f32 -- and its multiplies / adds go two outputs per instruction
  // (v_pk_*_f32: the bits of the scalar operations); the zero-point term of a column is multiplied once per tile, not per row block
  // (v_mul_lo_u32 is quarter rate).  ~8 VALU per output instead of ~20.
  c.plain = __builtin_amdgcn_readfirstlane((int)(c.fits32 && c.sane && __builtin_fabsf(c.alpha) <= 0x1p30f &&
                (!c.has_q || (c.fast && __builtin_fabsf(c.oo) <= 0x1p60f && c.qmin <= c.qmax && __builtin_fabsf(c.qmin) <= 0x1p60f &&
                              __builtin_fabsf(c.qmax) <= 0x1p60f)))) != 0;
  return c;
}

constexpr int OP = 68;                  // output staging pitch in floats (272 B: float4 writes of 16 rows spread over the banks)

// One round of 64 rows (16 per wave) x 64 columns: lo / hi = the wave's accumulators (D[n][m]: lane holds n = 4 fq + e of row m = frow),
// cs = column sums of the stored B bytes, colc = ca cs (plain route), rs = row sum of the stored A bytes.  Staged through the wave's own
// 16 rows of sO (no workgroup barrier), stored as whole 256-byte rows.
template <bool A16, int PLAIN, bool BUF = false>
__device__ __forceinline__ void store_round(const QmmArgs& g, const Epi& c, const v4i (&lo)[4], const v4i (&hi)[4], const v4i (&cs)[4], const int (&colc)[4][4],
                                            long long rs, float* sO, float* obase, int mrow0, int n0, int lane, int wave) {
  const int frow = lane & 15, fq = lane >> 4;
  const int ca32 = (int)c.ca, row32 = (int)(c.cb * rs + c.kcc);
#pragma unroll
  for (int j = 0; j < 4; ++j) {
    v4f y;
    if constexpr (PLAIN != 0) {
#pragma unroll
      for (int e = 0; e < 4; e += 2) {
        v2f v = {(float)(lo[j][e] + colc[j][e] + row32), (float)(lo[j][e + 1] + colc[j][e + 1] + row32)};
        v = v * splat2(c.alpha);
        if constexpr (PLAIN == 2) {
          const v2f q = div_by_scale2(v, c.so, c.inv_so);
          v2f idx = (v2f){rintf(q.x), rintf(q.y)} + splat2(c.oo);
          idx.x = __builtin_amdgcn_fmed3f(idx.x, c.qmin, c.qmax);
          idx.y = __builtin_amdgcn_fmed3f(idx.y, c.qmin, c.qmax);
          v = (idx - splat2(c.oo)) * splat2(c.so);
        }
        y[e] = v.x;
        y[e + 1] = v.y;
      }
    } else {
#pragma unroll
      for (int e = 0; e < 4; ++e) {
        float v;
        if (c.fits32) {
          v = __fmul_rn((float)(lo[j][e] + ca32 * cs[j][e] + row32), c.alpha);
        } else {
          const long long p = A16 ? 256ll * hi[j][e] + lo[j][e] : (long long)lo[j][e];
          const long long t = p + c.cb * rs + c.ca * (long long)cs[j][e] + c.kcc;
          v = A16 ? (float)((double)t * (double)c.alpha) : __fmul_rn((float)(double)t, c.alpha);
        }
        if (!c.sane) v = __builtin_nanf("");
        if (c.has_q) {
          const float q = div_by_scale_guarded(v, c.so, c.inv_so, c.fast);
          const float r = __fadd_rn(__fsub_rn(rintf(q), q), q);              // round_ste (qmodule.py:17-19): +-inf / NaN -> NaN
          const float idx = fminf(fmaxf(__fadd_rn(r, c.oo), c.qmin), c.qmax);
          v = r != r ? r : __fmul_rn(__fsub_rn(idx, c.oo), c.so);            // torch.clamp propagates NaN, fminf / fmaxf drop it
        }
        y[e] = v;
      }
    }
    *reinterpret_cast<v4f*>(&sO[(16 * wave + frow) * OP + 16 * j + 4 * fq]) = y;
  }
  // copy-out: the wave's 16 rows x 64 columns of this round, 16 lanes per row
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    const int idx = lane + 64 * i, r = idx >> 4, c4 = (idx & 15) * 4;
    const int mm = mrow0 + 16 * wave + r, n = n0 + c4;
    const v4f y = *reinterpret_cast<const v4f*>(&sO[(16 * wave + r) * OP + c4]);
    if constexpr (BUF) {
      // buffer store: a lane outside the matrix gets an offset past num_records and the hardware drops it -- no branch around the
      // store, so hipcc can count the stores behind a pending load exactly (s_waitcnt vmcnt(N) instead of draining them)
      const unsigned off = (mm < g.M && n < g.N) ? (unsigned)(mm * g.N + n) * 4u : 0x80000000u;
      __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(v4i, y), __builtin_amdgcn_make_buffer_rsrc(obase, 0, g.M * g.N * 4, 0x00020000), (int)off, 0, 0);
    } else if (mm < g.M) {
      float* orow = obase + (long long)mm * g.N;
      if (n + 4 <= g.N && g.o_vec) {
        *reinterpret_cast<v4f*>(orow + n) = y;
      } else {
#pragma unroll
        for (int e = 0; e < 4; ++e)
          if (n + e < g.N) orow[n + e] = y[e];
      }
    }
  }
}

template <int PLAIN>
__device__ __forceinline__ void column_terms(const Epi& c, const v4i (&cs)[4], int (&colc)[4][4]) {
  if constexpr (PLAIN != 0) {
#pragma unroll
    for (int j = 0; j < 4; ++j)
#pragma unroll
      for (int e = 0; e < 4; ++e) colc[j][e] = (int)c.ca * cs[j][e];
  }
}

// rows of K floats.  VEC (K % 4 == 0 on a 16-byte aligned base): ONE unconditional dwordx4 per 4 k -- past the row's end the address
// is clamped to its last quad and the caller's `nlive` mask turns those bytes into zeros, so the hot loops carry no branch around a load
// (a conditional load makes hipcc drain the whole vector-memory queue, stores included, at the merge).  Else element loads.
template <bool VEC>
__device__ __forceinline__ v4f load_k4(const float* row, int k, int K) {
  if constexpr (VEC) {
    return *reinterpret_cast<const v4f*>(row + min(k, K - 4));
  } else {
    v4f r = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int e = 0; e < 4; ++e)
      if (k + e < K) r[e] = row[k + e];
    return r;
  }
}

// RF: 16-row fragments per wave = 64 RF rows per workgroup (round 6).  At RF = 1 (rounds 5) the 64 x 64 tile re-read 2 B of fp32 operand
// through the L2s per output byte: every 64-row block of pv_bmm quantised the WHOLE of x2 again (as many bytes as its own rows of
// x1), every tile of qk_bmm read two 16-KB operand tiles for one 16-KB output tile.  RF = 2 (128 rows): half of the x2 traffic and
// of its quantiser arithmetic per output row (RF = 4 needs 256 VGPRs: one workgroup per CU, slower again).  Wave w owns rows 64 r + 16 w + (0 .. 15), r < RF: every round r of the epilogue stages and
// stores 64 consecutive rows.
template <bool A16, bool BKM, int RF, bool VEC>
__global__ void __launch_bounds__(256) qmatmul_kernel(const QmmArgs g) {
  // one LDS block: the int8 operand tiles of the K loop, then (behind the loop's last barrier) the fp32 output tile, staged so that
  // the stores are whole 256-byte rows instead of the MFMA layout's 64-byte pieces of 16 different rows
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

  const float sa = uniformf(g.ga.scale[0]), oa = uniformf(g.ga.offset[0]), sb = uniformf(g.gb.scale[0]), ob = uniformf(g.gb.offset[0]);
  const float inv_sa = __fdiv_rn(1.0f, sa), inv_sb = __fdiv_rn(1.0f, sb);
  // 8-bit: stored u8 = index + (128 - shift), byte ^ 0x80 = index - shift.  16-bit: u = index - qmin, planes hi / lo ^ 0x80
  const float a_bias = A16 ? -g.ga.qmin : (float)(128 - g.a_shift);
  const float b_bias = (float)(128 - g.b_shift);

  // loader roles.  K-contiguous operand: thread -> row tid / 4, 16 consecutive k from (tid % 4) * 16.
  // N-contiguous B (x2 as [K, N]): thread -> 4 k (tid / 16 * 4 ..) x 4 n ((tid % 16) * 4 ..): four 16-byte loads along n, transposed in
  // registers into four dwords of 4 consecutive k each.
  const int lr = tid >> 2, lk = (tid & 3) * 16;
  // x1: 16 lanes read 256 contiguous bytes of one row (one request of a quarter wave = two cache lines; the row-per-four-lanes order of
  // x2 below makes such a request touch eight), so a thread holds ONE quad of k (ak ..) of the rows ar + 16 i
  const int ar = tid >> 4, ak = (tid & 15) * 4;
  int arow[4 * RF];
#pragma unroll
  for (int i = 0; i < 4 * RF; ++i) arow[i] = min(m0 + ar + 16 * i, M - 1);
  const int bn = min(n0 + lr, N - 1);
  const int tk4 = (tid >> 4) * 4, tn4 = (tid & 15) * 4;
  const float* brow = BKM ? B + (long long)bn * K : B;

  v4f xa[RF][4], xb[4];
  auto load_a = [&](int rf, int k0) {
#pragma unroll
    for (int c = 0; c < 4; ++c) xa[rf][c] = load_k4<VEC>(A + (long long)arow[4 * rf + c] * K, k0 + ak, K);
  };
  auto load_b = [&](int k0) {
    if constexpr (BKM) {
#pragma unroll
      for (int c = 0; c < 4; ++c) xb[c] = load_k4<VEC>(brow, k0 + lk + 4 * c, K);
    } else {
      const int n = min(n0 + tn4, N - 4);        // (N % 4 == 0 in this memory order: the host sends other widths K-contiguous)
#pragma unroll
      for (int c = 0; c < 4; ++c) {
        const int k = min(k0 + tk4 + c, K - 1);                 // (rows past K: masked to zero bytes by quantise_b)
        xb[c] = *reinterpret_cast<const v4f*>(brow + (long long)k * N + n);
      }
    }
  };
  auto load_chunk = [&](int k0) {
#pragma unroll
    for (int rf = 0; rf < RF; ++rf) load_a(rf, k0);
    load_b(k0);
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

  // The K loop starts at a different chunk in every workgroup and wraps around (exact integer sums: any order gives the same bits).
  // Rows are K floats apart, so with every workgroup at the same k0 the whole chip asks for addresses that agree in their low bits --
  // a few HBM channels at a time.
  const int nch = (K + 63) >> 6, rot = (int)(lb % nch);
  auto chunk_k0 = [&](int i) { const int c = i + rot; return (c >= nch ? c - nch : c) << 6; };
  load_chunk(chunk_k0(0));
  for (int ci = 0; ci < nch; ++ci) {
    const int k0 = chunk_k0(ci), k0n = chunk_k0(ci + 1 < nch ? ci + 1 : ci);
    // quantise the chunk held in registers and park it in the LDS; k >= K contributes zero bytes (to the products AND to both sums)
#pragma unroll
    for (int rf = 0; rf < RF; ++rf) {
#pragma unroll
      for (int c = 0; c < 4; ++c) {
        int l1, h1 = 0;
        quant4<A16>(xa[rf][c], K - (k0 + ak), sa, inv_sa, oa, g.ga.qmin, g.ga.qmax, a_bias, l1, h1);
        *reinterpret_cast<int*>(&sA[0][(ar + 16 * (4 * rf + c)) * QP + ak]) = l1;
        if constexpr (A16) *reinterpret_cast<int*>(&sA[1][(ar + 16 * (4 * rf + c)) * QP + ak]) = h1;
      }
    }
    quantise_b<BKM>(xb, sB, QP, K - k0, lr, lk, tk4, tn4, n0, N, sb, inv_sb, ob, g.gb.qmin, g.gb.qmax, b_bias);
    __syncthreads();
    // next chunk's loads fly under this chunk's MFMAs (VEC: unconditional -- past the last chunk it is read again -- so that no
    // branch sits between the requests and their wait; requesting them earlier, as soon as a register quad is packed, measured 7 % slower)
    if (VEC || ci + 1 < nch) load_chunk(k0n);                  // next chunk's loads fly under this chunk's MFMAs
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

  const Epi c = make_epi<A16>(g, sa, oa, sb, ob);
  float* obase = g.out + (long long)bi * g.o_bs;
  auto tail = [&](auto plain) {
    constexpr int PLAIN = decltype(plain)::value;          // 0: every special case; 1 / 2: ordinary grids without / with an output quantizer
    int colc[4][4];
    column_terms<PLAIN>(c, cs, colc);
#pragma unroll
    for (int rf = 0; rf < RF; ++rf) {
      if (m0 + 64 * rf >= M) break;                                // (uniform) row blocks past the matrix
      const long long rs = A16 ? 256ll * rs_hi[A16 ? rf : 0][0] + rs_lo[rf][0] : (long long)rs_lo[rf][0];
      store_round<A16, PLAIN>(g, c, acc_lo[rf], acc_hi[A16 ? rf : 0], cs, colc, rs, sO, obase, m0 + 64 * rf, n0, lane, wave);
    }
  };
  if (!c.plain) tail(std::integral_constant<int, 0>{});
  else if (!c.has_q) tail(std::integral_constant<int, 1>{});
  else tail(std::integral_constant<int, 2>{});
}

// ---- row-panel kernel (round 6): x1 of <= 8 bits, K <= 64 KC, several 64-column tiles -- qk_bmm ---------------------------------------
// The tile kernel above quantises BOTH operand tiles again for every 128 x 64 output tile: at qk_bmm's shape (K = 64, S = T = 2048)
// that is 1.5 operand elements quantised per output element, more arithmetic than the epilogue, and a workgroup lives through
// load -> quantise -> MFMA -> epilogue -> store once, its stores overlapping nothing (measured: 137 us with the stores removed, 190 with
// them).  Here a workgroup owns 128 rows of x1 for a RANGE of column tiles: the row panel is quantised once (its MFMA fragments stay in
// registers, its row sums too), and the loop over column tiles quantises only the 64 x K tile of x2 (0.5 elements per output), with
// tile t + 1's loads requested before tile t's MFMAs and tile t's stores in flight under tile t + 1's arithmetic.  The x2 tile is
// double-buffered in the LDS: one barrier per tile.
// (K <= 64: three workgroups' worth of registers per SIMD -- the ordinary-grid loops fit, the special-case loop spills a little)
template <bool BKM, int KC, bool VEC>
__global__ void __launch_bounds__(256, KC == 1 ? 3 : 1) qmatmul_panel_kernel(const QmmArgs g) {
  constexpr int RF = 2, MT = 128, KP = 64 * KC + 16;             // LDS row pitch in bytes (ds_read_b128 of 16 rows: distinct bank groups)
  constexpr int PANEL = MT * KP, STAGE = 64 * OP * 4;
  // the x1 panel is only a transposing buffer (its fragments live in registers afterwards): the output staging tile takes its place
  __shared__ __attribute__((aligned(16))) char smem[(PANEL > STAGE ? PANEL : STAGE) + 2 * 64 * KP];
  char* sA = smem;
  float* sO = reinterpret_cast<float*>(smem);
  char* sB = smem + (PANEL > STAGE ? PANEL : STAGE);
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int frow = lane & 15, fq = lane >> 4;
  long long lb = blockIdx.x;
  if (g.nblk % 8 == 0) lb = (lb % 8) * (g.nblk / 8) + lb / 8;    // the workgroups of one batch element (one x2) share an XCD's L2
  const int sp = (int)(lb % g.nsplit);
  const int tm = (int)((lb / g.nsplit) % g.tiles_m);
  const int bi = (int)(lb / ((long long)g.nsplit * g.tiles_m));
  const int t0 = sp * g.tiles_per_split, t1 = min(t0 + g.tiles_per_split, g.tiles_n);
  const int m0 = tm * MT;
  const int M = g.M, N = g.N, K = g.K;
  const float* A = g.a + (long long)bi * g.a_bs;
  const float* B = g.b + (long long)bi * g.b_bs;
  const float sa = uniformf(g.ga.scale[0]), oa = uniformf(g.ga.offset[0]), sb = uniformf(g.gb.scale[0]), ob = uniformf(g.gb.offset[0]);
  const float inv_sa = __fdiv_rn(1.0f, sa), inv_sb = __fdiv_rn(1.0f, sb);
  const float a_bias = (float)(128 - g.a_shift), b_bias = (float)(128 - g.b_shift);
  const int lr = tid >> 2, lk = (tid & 3) * 16, tk4 = (tid >> 4) * 4, tn4 = (tid & 15) * 4;

  v4f xb[KC][4];
  auto load_b = [&](int t) {
    const int n0 = t * 64;
    if constexpr (BKM) {
      const float* brow = B + (long long)min(n0 + lr, N - 1) * K;
#pragma unroll
      for (int kc = 0; kc < KC; ++kc)
#pragma unroll
        for (int c = 0; c < 4; ++c) xb[kc][c] = load_k4<VEC>(brow, 64 * kc + lk + 4 * c, K);
    } else {
      const int n = min(n0 + tn4, N - 4);
#pragma unroll
      for (int kc = 0; kc < KC; ++kc)
#pragma unroll
        for (int c = 0; c < 4; ++c) {
          const int k = min(64 * kc + tk4 + c, K - 1);
          xb[kc][c] = *reinterpret_cast<const v4f*>(B + (long long)k * N + n);
        }
    }
  };
  // column tiles in a rotated order per workgroup (output rows are N floats apart: with every workgroup on the same tile column the
  // chip would write addresses that agree in their low bits -- a few HBM channels at a time); any order gives the same outputs
  const int nt = t1 - t0, trot = (int)((lb / g.nsplit) % nt);
  auto tile_of = [&](int i) { const int c = i + trot; return t0 + (c >= nt ? c - nt : c); };
  load_b(tile_of(0));                                            // the first x2 tile flies under the panel's quantisation

  // the panel: 128 rows x K, quantised once
#pragma unroll
  for (int rf = 0; rf < RF; ++rf) {
    const float* arow = A + (long long)min(m0 + 64 * rf + lr, M - 1) * K;
    v4f xa[KC][4];
#pragma unroll
    for (int kc = 0; kc < KC; ++kc)
#pragma unroll
      for (int c = 0; c < 4; ++c) xa[kc][c] = load_k4<VEC>(arow, 64 * kc + lk + 4 * c, K);
#pragma unroll
    for (int kc = 0; kc < KC; ++kc) {
      v4i lo;
#pragma unroll
      for (int c = 0; c < 4; ++c) {
        int l1, h1 = 0;
        quant4<false>(xa[kc][c], K - (64 * kc + lk + 4 * c), sa, inv_sa, oa, g.ga.qmin, g.ga.qmax, a_bias, l1, h1);
        lo[c] = l1;
      }
      *reinterpret_cast<v4i*>(&sA[(64 * rf + lr) * KP + 64 * kc + lk]) = lo;
    }
  }
  __syncthreads();
  const v4i ones = {0x01010101, 0x01010101, 0x01010101, 0x01010101};
  v4i fa[RF][KC], rsv[RF];
#pragma unroll
  for (int rf = 0; rf < RF; ++rf) {
    rsv[rf] = (v4i){0, 0, 0, 0};
#pragma unroll
    for (int kc = 0; kc < KC; ++kc) {
      fa[rf][kc] = *reinterpret_cast<const v4i*>(&sA[(64 * rf + 16 * wave + frow) * KP + 64 * kc + fq * 16]);
      rsv[rf] = __builtin_amdgcn_mfma_i32_16x16x64_i8(ones, fa[rf][kc], rsv[rf], 0, 0, 0);
    }
  }
  const long long rs[RF] = {(long long)rsv[0][0], (long long)rsv[1][0]};
  // (the first tile's barrier below separates these fragment reads from the first staging writes into the same LDS bytes)

  const Epi c = make_epi<false>(g, sa, oa, sb, ob);
  float* obase = g.out + (long long)bi * g.o_bs;
  // Tile t: barrier (x2 tile t is in the LDS) -> MFMAs -> epilogue + stores of tile t -> tile t + 1's registers (requested one tile
  // ago) quantised into the other LDS buffer -> request tile t + 2.  VEC: straight-line code -- buffer stores (no branch), the request
  // past the last tile re-reads it -- so the wait for tile t + 1's loads is a counted one that leaves tile t's stores in flight.
  auto tiles = [&](auto plain) {
    constexpr int PLAIN = decltype(plain)::value;          // 0: every special case; 1 / 2: ordinary grids without / with an output quantizer
    auto park = [&](int i) {                                       // tile number i of this workgroup's order -> LDS buffer i & 1
      char* sBt = sB + (i & 1) * (64 * KP);
#pragma unroll
      for (int kc = 0; kc < KC; ++kc)
        quantise_b<BKM>(xb[kc], sBt + 64 * kc, KP, K - 64 * kc, lr, lk, tk4, tn4, tile_of(i) * 64, N, sb, inv_sb, ob, g.gb.qmin, g.gb.qmax, b_bias);
    };
    park(0);
    if (VEC || 1 < nt) load_b(tile_of(min(1, nt - 1)));
    for (int i = 0; i < nt; ++i) {
      const int t = tile_of(i);
      const char* sBt = sB + (i & 1) * (64 * KP);
      // This barrier opens the loop's header block and the latch ends in park()'s LDS writes: hipcc (ROCm 7.2) placed the release
      // fence's wait BEHIND the s_barrier in the variants whose latch is conditional, and another wave read the tile before the last
      // dword had landed (tests/fuzz_qmatmul.py case 85; tools/barrier_audit.py finds the pattern).  An explicit lgkmcnt(0) in front.
      __builtin_amdgcn_s_waitcnt(0xc07f);
      __syncthreads();
      v4i acc[RF][4], cs[4];
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        cs[j] = (v4i){0, 0, 0, 0};
#pragma unroll
        for (int rf = 0; rf < RF; ++rf) acc[rf][j] = (v4i){0, 0, 0, 0};
#pragma unroll
        for (int kc = 0; kc < KC; ++kc) {
          const v4i fb = *reinterpret_cast<const v4i*>(&sBt[(16 * j + frow) * KP + 64 * kc + fq * 16]);
#pragma unroll
          for (int rf = 0; rf < RF; ++rf) acc[rf][j] = __builtin_amdgcn_mfma_i32_16x16x64_i8(fb, fa[rf][kc], acc[rf][j], 0, 0, 0);
          cs[j] = __builtin_amdgcn_mfma_i32_16x16x64_i8(fb, ones, cs[j], 0, 0, 0);
        }
      }
      int colc[4][4];
      column_terms<PLAIN>(c, cs, colc);
#pragma unroll
      for (int rf = 0; rf < RF; ++rf) {
        if (!VEC && m0 + 64 * rf >= M) break;
        store_round<false, PLAIN, VEC>(g, c, acc[rf], acc[rf], cs, colc, rs[rf], sO, obase, m0 + 64 * rf, t * 64, lane, wave);
      }
      if (VEC || i + 1 < nt) park(i + 1);                        // (past the last tile: its successor's bytes into the idle buffer)
      if (VEC || i + 2 < nt) load_b(tile_of(min(i + 2, nt - 1)));
    }
  };
  if (!c.plain) tiles(std::integral_constant<int, 0>{});
  else if (!c.has_q) tiles(std::integral_constant<int, 1>{});
  else tiles(std::integral_constant<int, 2>{});
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
  const bool vec = g.a_vec && g.b_vec;          // every K-contiguous operand takes unconditional 16-byte loads
  g.nsplit = 1;
  g.tiles_per_split = g.tiles_n;
  // row-panel kernel: an x1 of at most 8 bits with K <= 256 against several column tiles (qk_bmm).  Column ranges of at least four
  // tiles (the panel's quantisation amortised), split until the launch has ~8 workgroups per CU (S = T = 2048, 32 heads: 1 / 2 / 3 / 4 /
  // 8 ranges per row panel = 145 / 141 / 125 / 120 / 137 us)
  if (!a16 && K <= 256 && M > 64 && N > 64) {
    const long long tiles_m = (M + 127) / 128, tiles_n = (N + 63) / 64, base = batch * tiles_m;
    long long nsplit = std::min<long long>(tiles_n, std::max<long long>(1, (2048 + base - 1) / base));
    long long per = std::max<long long>((tiles_n + nsplit - 1) / nsplit, std::min<long long>(4, tiles_n));
    nsplit = (tiles_n + per - 1) / per;
    g.tiles_m = (int)tiles_m;
    g.tiles_n = (int)tiles_n;
    g.nsplit = (int)nsplit;
    g.tiles_per_split = (int)per;
    g.nblk = base * nsplit;
    MQ_REQUIRE(g.nblk < (1ll << 31), "%s: too many tiles", fn);
    const dim3 pgrid((unsigned)g.nblk), pblock(256);
#define MQ_QMM_PANEL(BKM, VEC)                                                            \
  do {                                                                                    \
    if (K <= 64) qmatmul_panel_kernel<BKM, 1, VEC><<<pgrid, pblock, 0, st>>>(g);          \
    else if (K <= 128) qmatmul_panel_kernel<BKM, 2, VEC><<<pgrid, pblock, 0, st>>>(g);    \
    else qmatmul_panel_kernel<BKM, 4, VEC><<<pgrid, pblock, 0, st>>>(g);                  \
  } while (0)
    const bool vecp = vec && g.o_vec && M * N < (1ll << 29);       // + buffer stores of whole quads with 32-bit byte offsets
    if (x2_k_contiguous) {
      if (vecp) MQ_QMM_PANEL(true, true);
      else MQ_QMM_PANEL(true, false);
    } else {
      if (vecp) MQ_QMM_PANEL(false, true);
      else MQ_QMM_PANEL(false, false);
    }
#undef MQ_QMM_PANEL
    MQ_LAUNCH_CHECK(fn);
    return MQ_OK;
  }
  const dim3 grid((unsigned)g.nblk), block(256);
#define MQ_QMM_LAUNCH(A16, BKM)                                                           \
  do {                                                                                    \
    if (rf == 2 && vec) qmatmul_kernel<A16, BKM, 2, true><<<grid, block, 0, st>>>(g);     \
    else if (rf == 2) qmatmul_kernel<A16, BKM, 2, false><<<grid, block, 0, st>>>(g);      \
    else if (vec) qmatmul_kernel<A16, BKM, 1, true><<<grid, block, 0, st>>>(g);           \
    else qmatmul_kernel<A16, BKM, 1, false><<<grid, block, 0, st>>>(g);                   \
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

// W8A8 / W4A8 QLinear GEMM for gfx950: int8 x int8 -> int32 on v_mfma_i32_16x16x64_i8, with the
// zero-point-corrected dequant (+ optional output quantizer) fused into the epilogue.
//
//   out[m,n] = alpha[n] * ( sum_k a[m,k]*w[n,k] - w_zp[n]*a_rowsum[m] + col_term[n] ) + bias[n]
//
// replaces QLinear.forward's fp32 simulation (qmodule.py:341-358; integer equivalence: SURVEY 8a' item 9).
//
// Structure (DESIGN.md "GEMM"):
//   * Both operands are K-contiguous ([M,K] activations, [N,K] weights), so both MFMA operands are
//     16 rows x 64 bytes and a lane's fragment is one 16-byte ds_read_b128.
//   * HBM/L2 -> LDS by LDS-DMA (global_load_lds_dwordx4, 16 B/lane, no VGPR round trip).  A wave
//     instruction moves 8 rows x 128 B, full 128-byte lines.  The DMA destination is lane-linear, so
//     the bank-conflict swizzle (16-byte chunk c of row r lives at chunk c ^ (r & 7)) is applied to
//     the per-lane SOURCE address and undone on the ds_read address (same involution both sides).
//   * Two LDS stages of BK = 128 bytes of K; the DMA of stage t+1 is issued before the MFMAs of stage
//     t and retired by one counted s_waitcnt + one s_barrier per stage.
//   * The MFMA is issued as mfma(W fragment, X fragment): D[n][m], so a lane ends up holding FOUR
//     CONSECUTIVE n for one m -- the epilogue loads its per-n vectors as float4 and stores 16 bytes
//     (fp32) per lane without any cross-lane shuffle.
//   * blockIdx -> tile mapping is XCD aware: the eight XCDs (block b runs on XCD b % 8) each get a
//     contiguous run of the grouped tile order, so the tiles sharing an A panel / W panel hit the
//     same private L2.
#include <hip/hip_fp16.h>

#include <type_traits>

#include "mq_common.h"

namespace mq {

typedef int v4i __attribute__((ext_vector_type(4)));
typedef float v4f __attribute__((ext_vector_type(4)));

#define MQ_GLOBAL_PTR(p) ((const __attribute__((address_space(1))) void*)(p))
#define MQ_LDS_PTR(p) ((__attribute__((address_space(3))) void*)(p))

struct GemmArgs {
  const int8_t* a;
  const void* w;      // int8 [N,K] or packed nibbles [N,K/2]
  int M, N, K;
  const int32_t* a_rowsum;
  const float* alpha;
  const int32_t* w_zp;
  const int32_t* col_term;
  const float* bias;
  const float* out_scale;
  const float* out_offset;
  float out_qmin, out_qmax;
  void* out;
  int out_dtype;
  int grid_m, grid_n;
};

constexpr int BK = 128;   // bytes of K per LDS stage (two MFMA k-steps of 64)

// Bijective XCD-aware remap of the linear block id, then grouped (GROUP_M tall) tile order.
__device__ __forceinline__ void tile_of_block(int bid, int nblk, int grid_m, int grid_n, int& tm, int& tn) {
  const int q = nblk >> 3, r = nblk & 7;
  const int xcd = bid & 7, idx = bid >> 3;
  const int L = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + idx;
  constexpr int GROUP_M = 4;
  const int per_group = GROUP_M * grid_n;
  const int g = L / per_group;
  const int first_m = g * GROUP_M;
  const int gm = (grid_m - first_m) < GROUP_M ? (grid_m - first_m) : GROUP_M;
  const int in_g = L - g * per_group;
  tm = first_m + in_g % gm;
  tn = in_g / gm;
}

template <int OUT>
struct OutT;
template <> struct OutT<MQ_F32> { using type = float; };
template <> struct OutT<MQ_F16> { using type = __half; };
template <> struct OutT<MQ_U8> { using type = uint8_t; };
template <> struct OutT<MQ_I8> { using type = int8_t; };
template <> struct OutT<MQ_U16> { using type = uint16_t; };
template <> struct OutT<MQ_I16> { using type = int16_t; };

// Store 4 consecutive-n results of one lane.
template <int OUT>
__device__ __forceinline__ void store4(void* out, size_t idx, const float (&v)[4], int shift) {
  if constexpr (OUT == MQ_F32) {
    *reinterpret_cast<float4*>(reinterpret_cast<float*>(out) + idx) = make_float4(v[0], v[1], v[2], v[3]);
  } else if constexpr (OUT == MQ_F16) {
    struct alignas(8) H4 { __half h[4]; };
    H4 p;
#pragma unroll
    for (int j = 0; j < 4; ++j) p.h[j] = __float2half_rn(v[j]);
    *reinterpret_cast<H4*>(reinterpret_cast<__half*>(out) + idx) = p;
  } else if constexpr (OUT == MQ_U8 || OUT == MQ_I8) {
    uint32_t p = 0;
#pragma unroll
    for (int j = 0; j < 4; ++j) p |= (uint32_t)((int)v[j] - shift & 0xff) << (8 * j);
    *reinterpret_cast<uint32_t*>(reinterpret_cast<uint8_t*>(out) + idx) = p;
  } else {
    uint2 p;
    p.x = (uint32_t)((int)v[0] & 0xffff) | ((uint32_t)((int)v[1] & 0xffff) << 16);
    p.y = (uint32_t)((int)v[2] & 0xffff) | ((uint32_t)((int)v[3] & 0xffff) << 16);
    *reinterpret_cast<uint2*>(reinterpret_cast<uint16_t*>(out) + idx) = p;
  }
}

// ABL: compile-time ablation for profiling builds (-DMQ_GEMM_ABLATE): bit0 = no LDS-DMA after the
// first stage, bit1 = no MFMA loop body, bit2 = no epilogue.  Production instantiates ABL = 0 only.
template <int BM, int BN, int WM, int WN, int OUT, bool OUTQ, bool W4, int ABL = 0>
__global__ void __launch_bounds__(64 * WM * WN)
    gemm_i8_kernel(const GemmArgs args) {
  constexpr int NW = WM * WN;
  constexpr int TM = BM / WM, TN = BN / WN;
  constexpr int FM = TM / 16, FN = TN / 16;
  static_assert(TM % 16 == 0 && TN % 16 == 0 && BM % 8 == 0 && BN % 8 == 0, "tile shape");
  // K step in ELEMENTS is 128 for both operands: A stage = BM x 128 B, W8 stage = BN x 128 B,
  // W4 stage = BN x 64 B (128 packed nibbles).
  constexpr int WROW = W4 ? BK / 2 : BK;            // bytes per weight row per stage
  constexpr int A_BYTES = BM * BK;
  constexpr int W_BYTES = BN * WROW;
  constexpr int STAGE = A_BYTES + W_BYTES;
  constexpr int A_INSTR = BM / 8;                   // LDS-DMA wave instructions per stage (8 rows x 128 B)
  constexpr int W_INSTR = W4 ? BN / 16 : BN / 8;    // W4: 16 rows x 64 B
  static_assert(A_INSTR % NW == 0, "A tile DMA instructions must split evenly over the waves");
  constexpr int A_ROUNDS = A_INSTR / NW;
  constexpr int W_ROUNDS = (W_INSTR + NW - 1) / NW;
  constexpr bool W_TAIL = (W_INSTR % NW) != 0;      // last round: only some waves have an instruction
  constexpr int N_DMA = A_ROUNDS + W_ROUNDS;        // DMA instructions per wave per stage (max)
  constexpr int DMA_PER_GROUP = (N_DMA + FN - 1) / FN;
  constexpr int PAR = 2 * STAGE;                    // LDS offset of the per-n epilogue vectors

  extern __shared__ __attribute__((aligned(16))) char smem[];

  const int lane = threadIdx.x & 63;
  const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const int wave_m = wave / WN, wave_n = wave % WN;

  int tm, tn;
  tile_of_block(blockIdx.x, gridDim.x, args.grid_m, args.grid_n, tm, tn);
  const int m0 = tm * BM, n0 = tn * BN;
  const int M = args.M, N = args.N, K = args.K;
  const int KT = K / BK;

  // ---- LDS-DMA source offsets (per lane, per instruction this wave owns) ------------------------
  // Round i, wave w owns instruction j = w + i*NW (rows 8j..8j+7 of the tile, 1 KiB of LDS).
  // A / W8 instruction: lane -> row (lane>>3), stored chunk (lane&7) holds logical chunk (lane&7)^(row&7).
  // W4 instruction (64-byte rows): lane -> row (lane>>2), stored chunk (lane&3) holds logical chunk
  // (lane&3) ^ g(row), g = {0,3,2,1}[(row>>2)&3]  (conflict-free for ds_read_b128 at a 64-byte pitch).
  int src_a[A_ROUNDS], src_w[W_ROUNDS];
#pragma unroll
  for (int i = 0; i < A_ROUNDS; ++i) {
    int row = m0 + (wave + i * NW) * 8 + (lane >> 3);
    row = row < M ? row : M - 1;
    src_a[i] = row * K + (((lane & 7) ^ (lane >> 3)) << 4);
  }
#pragma unroll
  for (int i = 0; i < W_ROUNDS; ++i) {
    const int jw = wave + i * NW;
    if constexpr (W4) {
      const int r = lane >> 2;
      int row = n0 + jw * 16 + r;
      row = row < N ? row : N - 1;
      const int g = (4 - ((r >> 2) & 3)) & 3;
      src_w[i] = row * (K >> 1) + ((((lane & 3) ^ g)) << 4);
    } else {
      int row = n0 + jw * 8 + (lane >> 3);
      row = row < N ? row : N - 1;
      src_w[i] = row * K + (((lane & 7) ^ (lane >> 3)) << 4);
    }
  }
  const int8_t* a_ptr = args.a;
  const int8_t* w_ptr = reinterpret_cast<const int8_t*>(args.w);

  // one DMA instruction: index d in [0, N_DMA) = A rounds first, then W rounds
  auto issue_one = [&](int d, int buf, int kt) {
    char* base = smem + buf * STAGE + wave * 1024;
    if (d < A_ROUNDS) {
      __builtin_amdgcn_global_load_lds(MQ_GLOBAL_PTR(a_ptr + src_a[d] + kt * BK), MQ_LDS_PTR(base + d * NW * 1024), 16, 0, 0);
    } else {
      const int i = d - A_ROUNDS;
      if (!W_TAIL || i < W_ROUNDS - 1 || wave + i * NW < W_INSTR)
        __builtin_amdgcn_global_load_lds(MQ_GLOBAL_PTR(w_ptr + src_w[i] + kt * WROW),
                                         MQ_LDS_PTR(base + A_BYTES + i * NW * 1024), 16, 0, 0);
    }
  };

  // stage 0 first: its latency overlaps the parameter staging below
#pragma unroll
  for (int d = 0; d < N_DMA; ++d) issue_one(d, 0, 0);

  // ---- per-n epilogue vectors -> LDS (read back as 16-byte vectors in the epilogue) --------------
  {
    float* p_alpha = reinterpret_cast<float*>(smem + PAR);
    float* p_bias = p_alpha + BN;
    int* p_zw = reinterpret_cast<int*>(p_bias + BN);
    int* p_ct = p_zw + BN;
    for (int t = threadIdx.x; t < BN; t += 64 * NW) {
      const int n = n0 + t;
      const bool ok = n < N;
      p_alpha[t] = ok ? args.alpha[n] : 0.f;
      p_bias[t] = (ok && args.bias != nullptr) ? args.bias[n] : 0.f;
      p_zw[t] = ok ? args.w_zp[n] : 0;
      p_ct[t] = ok ? args.col_term[n] : 0;
    }
  }
  const int frow = lane & 15, kq = lane >> 4;
  int rs[FM];
#pragma unroll
  for (int i = 0; i < FM; ++i) {
    const int m = m0 + wave_m * TM + i * 16 + frow;
    rs[i] = (args.a_rowsum != nullptr && m < M) ? args.a_rowsum[m] : 0;
  }

  // ---- ds_read offsets (per lane) -----------------------------------------------------------------
  // 128-byte rows: chunk (kq + 4*ks) ^ (row & 7); ks toggles bit 2 -> XOR 64 on the byte address
  const int x_off = (wave_m * TM + frow) * BK + ((kq ^ (lane & 7)) << 4);
  int w_off;
  if constexpr (W4) {
    w_off = A_BYTES + (wave_n * TN + frow) * WROW;   // group index added per use (depends on kq/ks)
  } else {
    w_off = A_BYTES + (wave_n * TN + frow) * BK + ((kq ^ (lane & 7)) << 4);
  }

  v4i acc[FM][FN];
#pragma unroll
  for (int i = 0; i < FM; ++i)
#pragma unroll
    for (int j = 0; j < FN; ++j) acc[i][j] = v4i{0, 0, 0, 0};

  auto k_step = [&](int kt, auto more_tag) {
    constexpr bool more = decltype(more_tag)::value && !(ABL & 1);
    const int cur = kt & 1;
    // own DMA of stage kt retired, then everyone's; also: every wave has finished reading buf cur^1
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();
    const char* sb = smem + cur * STAGE;
    if constexpr (ABL & 2) {
      if constexpr (more) {
#pragma unroll
        for (int d = 0; d < N_DMA; ++d) issue_one(d, cur ^ 1, kt + 1);
      }
      return;
    }
#pragma unroll
    for (int ks = 0; ks < 2; ++ks) {
      v4i xf[FM];
#pragma unroll
      for (int i = 0; i < FM; ++i)
        xf[i] = *reinterpret_cast<const v4i*>(sb + ((x_off + i * 16 * BK) ^ (ks << 6)));
      if constexpr (!W4) {
        v4i wf[FN];
#pragma unroll
        for (int j = 0; j < FN; ++j)
          wf[j] = *reinterpret_cast<const v4i*>(sb + ((w_off + j * 16 * BK) ^ (ks << 6)));
#pragma unroll
        for (int j = 0; j < FN; ++j) {
#pragma unroll
          for (int i = 0; i < FM; ++i)
            acc[i][j] = __builtin_amdgcn_mfma_i32_16x16x64_i8(wf[j], xf[i], acc[i][j], 0, 0, 0);
          // next stage's DMA, interleaved with the first k-step's MFMAs (wave-uniform branch)
          if constexpr (more) if (ks == 0) {
#pragma unroll
            for (int d = j * DMA_PER_GROUP; d < (j + 1) * DMA_PER_GROUP && d < N_DMA; ++d) issue_one(d, cur ^ 1, kt + 1);
          }
        }
      } else {
        // Packed group G (16 B) = K-elements [32G, 32G+32): low nibbles = first 16, high = next 16.
        // MFMA k-step ks covers elements [64ks, 64ks+64): lane quarter kq needs [64ks+16kq, +16)
        //  -> group 2ks + (kq>>1), half (kq&1): every lane reads one group and shifts its half down.
        const int grp = 2 * ks + (kq >> 1);
        const int g = (4 - ((frow >> 2) & 3)) & 3;
#pragma unroll
        for (int j = 0; j < FN; ++j) {
          const v4i p = *reinterpret_cast<const v4i*>(sb + w_off + j * 16 * WROW + ((grp ^ g) << 4));
          v4i wf;
#pragma unroll
          for (int e = 0; e < 4; ++e) wf[e] = (int)(((unsigned)p[e] >> ((kq & 1) * 4)) & 0x0f0f0f0fu);
#pragma unroll
          for (int i = 0; i < FM; ++i)
            acc[i][j] = __builtin_amdgcn_mfma_i32_16x16x64_i8(wf, xf[i], acc[i][j], 0, 0, 0);
          if constexpr (more) if (ks == 0) {
#pragma unroll
            for (int d = j * DMA_PER_GROUP; d < (j + 1) * DMA_PER_GROUP && d < N_DMA; ++d) issue_one(d, cur ^ 1, kt + 1);
          }
        }
      }
    }
  };
  for (int kt = 0; kt < KT - 1; ++kt) k_step(kt, std::true_type{});
  k_step(KT - 1, std::false_type{});

  // ---- epilogue -----------------------------------------------------------------------------------
  // lane holds, for fragment (i, j): m = frow, n = 4*kq + {0..3}
  if constexpr (ABL & 4) {
    if (acc[0][0][0] == 0x7fffffff) reinterpret_cast<int*>(args.out)[0] = 1;
    return;
  }
  float inv_so = 0.f, so = 0.f, oo = 0.f;
  int oshift = 0;
  if constexpr (OUTQ) {
    so = args.out_scale[0];
    oo = args.out_offset[0];
    inv_so = __fdiv_rn(1.0f, so);
    if constexpr (OUT == MQ_I8) oshift = (args.out_qmin == 0.0f) ? 128 : 0;
  }
  const float qmin = args.out_qmin, qmax = args.out_qmax;
  const v4f* p_alpha = reinterpret_cast<const v4f*>(smem + PAR);
  const v4f* p_bias = reinterpret_cast<const v4f*>(smem + PAR + BN * 4);
  const v4i* p_zw = reinterpret_cast<const v4i*>(smem + PAR + BN * 8);
  const v4i* p_ct = reinterpret_cast<const v4i*>(smem + PAR + BN * 12);
#pragma unroll
  for (int j = 0; j < FN; ++j) {
    const int nl = wave_n * TN + j * 16 + kq * 4;   // column within the tile
    const int n = n0 + nl;
    if (n + 3 < N) {
      const v4f al = p_alpha[nl >> 2];
      const v4f bs = p_bias[nl >> 2];
      const v4i zw = p_zw[nl >> 2];
      const v4i ct = p_ct[nl >> 2];
#pragma unroll
      for (int i = 0; i < FM; ++i) {
        const int m = m0 + wave_m * TM + i * 16 + frow;
        if (m < M) {
          float v[4];
#pragma unroll
          for (int e = 0; e < 4; ++e) {
            const int t = (int)((unsigned)acc[i][j][e] - (unsigned)zw[e] * (unsigned)rs[i] + (unsigned)ct[e]);
            float f = __fmul_rn((float)t, al[e]);
            f = __fadd_rn(f, bs[e]);
            if constexpr (OUTQ) {
              float q = rintf(f * inv_so) + oo;
              q = fminf(fmaxf(q, qmin), qmax);
              if constexpr (OUT == MQ_F32 || OUT == MQ_F16) f = __fmul_rn(__fsub_rn(q, oo), so);
              else f = q;
            }
            v[e] = f;
          }
          store4<OUT>(args.out, (size_t)m * N + n, v, oshift);
        }
      }
    }
  }
}

// ---- variants & dispatch --------------------------------------------------------------------------
struct Variant {
  const char* name;
  int bm, bn, threads;
};

static const Variant kVariants[] = {
    {"t256x176_w4x1", 256, 176, 256},
    {"t256x176_w8x1", 256, 176, 512},
    {"t256x256_w2x4", 256, 256, 512},
    {"t128x128_w2x2", 128, 128, 256},
    {"t128x256_w2x2", 128, 256, 256},
    {"t256x128_w4x2", 256, 128, 512},
    {"t64x64_w2x2", 64, 64, 256},
};
constexpr int kNumVariants = sizeof(kVariants) / sizeof(kVariants[0]);
static int g_forced_variant = -1;
static int g_debug = 0;

template <int BM, int BN, int WM, int WN, int OUT, bool OQ, bool W4, int ABL>
static int launch_one(const GemmArgs& a, int lds, hipStream_t st) {
  auto kfn = gemm_i8_kernel<BM, BN, WM, WN, OUT, OQ, W4, ABL>;
  static bool attr_set = false;   // per instantiation; one device per process
  if (!attr_set) {
    hipError_t e = hipFuncSetAttribute((const void*)kfn, hipFuncAttributeMaxDynamicSharedMemorySize, lds);
    if (e != hipSuccess) {
      set_error("mq_gemm: hipFuncSetAttribute(%d B LDS): %s", lds, hipGetErrorString(e));
      return MQ_EHIP;
    }
    attr_set = true;
  }
  kfn<<<a.grid_m * a.grid_n, 64 * WM * WN, lds, st>>>(a);
  MQ_LAUNCH_CHECK("mq_gemm");
  return MQ_OK;
}

template <int BM, int BN, int WM, int WN, int OUT, bool OQ, bool W4>
static int launch_typed(const GemmArgs& a, hipStream_t st) {
  constexpr int WROW = W4 ? BK / 2 : BK;
  constexpr int LDS = 2 * (BM * BK + BN * WROW) + 16 * BN;
#ifdef MQ_GEMM_ABLATE
  if constexpr (OUT == MQ_U8 && OQ && !W4 && BM == 256) {
    switch (g_debug) {
      case 1: return launch_one<BM, BN, WM, WN, OUT, OQ, W4, 1>(a, LDS, st);
      case 2: return launch_one<BM, BN, WM, WN, OUT, OQ, W4, 2>(a, LDS, st);
      case 3: return launch_one<BM, BN, WM, WN, OUT, OQ, W4, 3>(a, LDS, st);
      case 4: return launch_one<BM, BN, WM, WN, OUT, OQ, W4, 4>(a, LDS, st);
      case 5: return launch_one<BM, BN, WM, WN, OUT, OQ, W4, 5>(a, LDS, st);
      case 6: return launch_one<BM, BN, WM, WN, OUT, OQ, W4, 6>(a, LDS, st);
      case 7: return launch_one<BM, BN, WM, WN, OUT, OQ, W4, 7>(a, LDS, st);
      default: break;
    }
  }
#endif
  return launch_one<BM, BN, WM, WN, OUT, OQ, W4, 0>(a, LDS, st);
}

template <int BM, int BN, int WM, int WN, bool W4>
static int launch_cfg(const GemmArgs& a, bool outq, hipStream_t st) {
  if (outq) {
    switch (a.out_dtype) {
      case MQ_F32: return launch_typed<BM, BN, WM, WN, MQ_F32, true, W4>(a, st);
      case MQ_F16: return launch_typed<BM, BN, WM, WN, MQ_F16, true, W4>(a, st);
      case MQ_U8: return launch_typed<BM, BN, WM, WN, MQ_U8, true, W4>(a, st);
      case MQ_I8: return launch_typed<BM, BN, WM, WN, MQ_I8, true, W4>(a, st);
      case MQ_U16: return launch_typed<BM, BN, WM, WN, MQ_U16, true, W4>(a, st);
      case MQ_I16: return launch_typed<BM, BN, WM, WN, MQ_I16, true, W4>(a, st);
      default: set_error("mq_gemm: out_dtype %d not supported", a.out_dtype); return MQ_EUNSUPPORTED;
    }
  }
  switch (a.out_dtype) {
    case MQ_F32: return launch_typed<BM, BN, WM, WN, MQ_F32, false, W4>(a, st);
    case MQ_F16: return launch_typed<BM, BN, WM, WN, MQ_F16, false, W4>(a, st);
    default:
      set_error("mq_gemm: integer out_dtype %d needs an output quantizer", a.out_dtype);
      return MQ_EINVAL;
  }
}

static int pick_variant(int M, int N) {
  if (g_forced_variant >= 0) return g_forced_variant;
  auto blocks = [&](int v) {
    return (long)((M + kVariants[v].bm - 1) / kVariants[v].bm) * ((N + kVariants[v].bn - 1) / kVariants[v].bn);
  };
  // Measured on MI355X (tools/mq_probe, profiles/): the 8-wave 256x176 tile is the fastest whenever it
  // tiles N exactly and fills the chip in one round (TinyLlama / StableLM FFN: N = 5632 = 32 x 176);
  // otherwise pick the largest tile that still gives every CU a workgroup, else the small tiles.
  if (N % 176 == 0 && blocks(1) >= 192) return 1;
  const int order[] = {2, 5, 3, 6};
  for (int v : order)
    if (blocks(v) >= 224) return v;
  return blocks(3) >= 96 ? 3 : 6;
}

template <bool W4>
static int run_gemm(GemmArgs a, hipStream_t st) {
  const bool outq = a.out_scale != nullptr;
  const int v = pick_variant(a.M, a.N);
  a.grid_m = (a.M + kVariants[v].bm - 1) / kVariants[v].bm;
  a.grid_n = (a.N + kVariants[v].bn - 1) / kVariants[v].bn;
  switch (v) {
    case 0: return launch_cfg<256, 176, 4, 1, W4>(a, outq, st);
    case 1: return launch_cfg<256, 176, 8, 1, W4>(a, outq, st);
    case 2: return launch_cfg<256, 256, 2, 4, W4>(a, outq, st);
    case 3: return launch_cfg<128, 128, 2, 2, W4>(a, outq, st);
    case 4: return launch_cfg<128, 256, 2, 2, W4>(a, outq, st);
    case 5: return launch_cfg<256, 128, 4, 2, W4>(a, outq, st);
    case 6: return launch_cfg<64, 64, 2, 2, W4>(a, outq, st);
    default: set_error("mq_gemm: bad variant %d", v); return MQ_EINVAL;
  }
}

static int check_common(const char* fn, const void* a, const void* w, int64_t M, int64_t N, int64_t K,
                        const int32_t* a_rowsum, const float* alpha, const int32_t* w_zp, const int32_t* col_term,
                        const float* bias, const float* out_scale, const float* out_offset, void* out, int kdiv) {
  MQ_REQUIRE(a && w && alpha && w_zp && col_term && out, "%s: null pointer", fn);
  MQ_REQUIRE(M > 0 && N > 0 && K > 0, "%s: bad shape M=%lld N=%lld K=%lld", fn, (long long)M, (long long)N, (long long)K);
  MQ_REQUIRE(K % 128 == 0, "%s: K=%lld must be a multiple of 128", fn, (long long)K);
  MQ_REQUIRE(N % 4 == 0, "%s: N=%lld must be a multiple of 4", fn, (long long)N);
  MQ_REQUIRE(M * K < (1ll << 31) && N * K / kdiv < (1ll << 31) && M * N < (1ll << 40), "%s: operand too large", fn);
  MQ_REQUIRE(aligned(a, 16) && aligned(w, 16) && aligned(out, 16) && aligned(alpha, 16) && aligned(w_zp, 16) &&
                 aligned(col_term, 16) && (!bias || aligned(bias, 16)),
             "%s: pointers must be 16-byte aligned", fn);
  MQ_REQUIRE((out_scale == nullptr) == (out_offset == nullptr), "%s: out_scale/out_offset must both be set or NULL", fn);
  (void)a_rowsum;
  return MQ_OK;
}

}  // namespace mq

using namespace mq;

extern "C" {

int mq_gemm_set_variant(int variant) {
  g_forced_variant = (variant >= 0 && variant < kNumVariants) ? variant : -1;
  return kNumVariants;
}

int mq_gemm_set_debug(int flags) {
  g_debug = flags;
  return 0;
}

const char* mq_gemm_variant_name(int variant) {
  return (variant >= 0 && variant < kNumVariants) ? kVariants[variant].name : "";
}

int mq_w8a8_linear(const int8_t* a, const int8_t* w, int64_t M, int64_t N, int64_t K, const int32_t* a_rowsum,
                   const float* alpha, const int32_t* w_zp, const int32_t* col_term, const float* bias,
                   const float* out_scale, const float* out_offset, float out_qmin, float out_qmax, void* out,
                   int out_dtype, mq_stream_t stream) {
  int rc = check_common("mq_w8a8_linear", a, w, M, N, K, a_rowsum, alpha, w_zp, col_term, bias, out_scale, out_offset,
                        out, 1);
  if (rc != MQ_OK) return rc;
  GemmArgs g{a, w, (int)M, (int)N, (int)K, a_rowsum, alpha, w_zp, col_term, bias, out_scale, out_offset,
             out_qmin, out_qmax, out, out_dtype, 0, 0};
  return run_gemm<false>(g, as_stream(stream));
}

int mq_w4a8_linear(const int8_t* a, const uint8_t* w_packed, int64_t M, int64_t N, int64_t K, const int32_t* a_rowsum,
                   const float* alpha, const int32_t* w_zp, const int32_t* col_term, const float* bias,
                   const float* out_scale, const float* out_offset, float out_qmin, float out_qmax, void* out,
                   int out_dtype, mq_stream_t stream) {
  int rc = check_common("mq_w4a8_linear", a, w_packed, M, N, K, a_rowsum, alpha, w_zp, col_term, bias, out_scale,
                        out_offset, out, 2);
  if (rc != MQ_OK) return rc;
  GemmArgs g{a, w_packed, (int)M, (int)N, (int)K, a_rowsum, alpha, w_zp, col_term, bias, out_scale, out_offset,
             out_qmin, out_qmax, out, out_dtype, 0, 0};
  return run_gemm<true>(g, as_stream(stream));
}

}  // extern "C"

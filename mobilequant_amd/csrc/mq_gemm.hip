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
#include <mutex>
#include <hip/hip_fp16.h>

#include <type_traits>

#include "mq_common.h"
#include "mobilequant_amd_tuning.h"
#include "mq_gemv.h"
#include "mq_gemm_pp_asm.inc"
#include "mq_gemm_fr_asm.inc"
#include "mq_gemm_fr128_asm.inc"
#include "mq_gemm_frg_asm.inc"
#include "mq_gemm_frg128_asm.inc"
#include "mq_gemm_fr160_asm.inc"
#include "mq_gemm_fr128r_asm.inc"
#include "mq_gemm_fr128r8_asm.inc"
// measured negatives (DESIGN.md 7 / NOTES.md): compiled only into experiment builds (`python -m mobilequant_amd.build --experiments`,
// -DMQ_BUILD_EXPERIMENTS); the production library answers their knobs with 1 (= not built) and keeps the production kernel
#ifdef MQ_BUILD_EXPERIMENTS
#include "mq_gemm_fr128rs_asm.inc"
#include "mq_gemm_frw4_asm.inc"
#include "mq_gemm_frw4_128_asm.inc"
#else
#define MQ_FR128RS_LDS_BYTES 0
#define MQ_FRW4_LDS_BYTES 0
#define MQ_FRW4_128_LDS_BYTES 0
#endif
#include "mq_gemm_frw4x_128r_asm.inc"
#include "mq_gemm_frgw4x_asm.inc"
#include "mq_gemm_frgw4x_128_asm.inc"
#include "mq_gemm_frw4x_asm.inc"
#include "mq_gemm_frw4x_128_asm.inc"

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
  int has_bias;                 // bias == NULL is replaced by a valid dummy pointer on the host
  int a_tiled;                  // A is in the fragment-blocked layout of mq_quantize_tiled (generated-ISA variant only)
  int has_rowsum;               // a_rowsum == NULL (caller guarantees w_zp == 0): dummy pointer, element 0 only
  unsigned long long* dbg_ts;   // ablation builds only: per-wave s_memtime stamps [block][wave][4]
  const float* resid;           // fp32 [M, N] added to the (quantised) output at the store (o_proj / w2 + the residual stream); fp32 out only
  // column segments with their own 8-bit output grids (q | k | v in one launch; integer index outputs only): columns
  // [seg_end[i-1], seg_end[i]) use seg_scale[i-1] / seg_offset[i-1]; columns below seg_end[0] use out_scale / out_offset
  int seg_end[2];
  const float* seg_scale[2];
  const float* seg_offset[2];
  // gated pair (mq_w8a8_linear_tiled_gated): the first launch (w1) also zeroes the row sums the second one accumulates into; the
  // second launch (w3, gemm_i8_frg_kernel) reads w1's indices back and writes w2's input image through the gated table
  int32_t* zero_buf;
  int zero_count;
  const uint8_t* gate_aidx;     // w1's u8 output indices [M, N] row-major
  const int8_t* gate_table;     // [256][256] (mq_gated_table)
  int8_t* gate_q;               // w2's input image, fragment-blocked [ceil16(M), N]
  int32_t* gate_rowsum;         // [M], zeroed by the first launch
  int group_m;                  // tile order: M-tiles per group (0 = 4); mq_gemm_set_group_m, fr128 family only (traffic experiments)
};

#ifndef MQ_PP_PRIO
#define MQ_PP_PRIO 1   // 1: raise the MFMA-phase wave's priority; 0: none; 2: raise the READ/DMA-phase wave instead
#endif

constexpr int BK = 128;   // bytes of K per LDS stage (two MFMA k-steps of 64)
static std::atomic<int> g_group_m{0};   // mobilequant_amd_tuning.h: M-tiles per group of the fr128 family's tile order (0 = 4)

// Bijective XCD-aware remap of the linear block id, then grouped (GROUP_M tall) tile order.
__device__ __forceinline__ void tile_of_block(int bid, int nblk, int grid_m, int grid_n, int& tm, int& tn, int group_m = 0) {
  const int q = nblk >> 3, r = nblk & 7;
  const int xcd = bid & 7, idx = bid >> 3;
  const int L = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + idx;
  const int GROUP_M = group_m > 0 ? group_m : 4;
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

// Bytes of LDS in front of the per-n epilogue vectors: the stage buffers, or the epilogue's staging tiles
// (NW waves x 16 rows x padded fp32 row) if those are larger.
constexpr int lds_main_bytes(int BM, int BN, int WM, int WN, bool W4, bool PP) {
  const int wrow = W4 ? BK / 2 : BK;
  const int stages = PP ? 2 * BM * BK + 3 * BN * wrow : 2 * (BM * BK + BN * wrow);
  const int staging = WM * WN * 16 * ((BN / WN) * 4 + 16);
  return stages > staging ? stages : staging;
}

// ABL: compile-time ablation for profiling builds (-DMQ_GEMM_ABLATE): bit0 = no LDS-DMA after the
// first stage, bit1 = no MFMA loop body, bit2 = no epilogue.  Production instantiates ABL = 0 only.
template <int BM, int BN, int WM, int WN, int OUT, bool OUTQ, bool W4, int ABL = 0, int PP = 0>
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
  // LDS map.  classic: [stage 0: A|W][stage 1: A|W][params].  ping-pong: [A0][A1][W0][W1][W2][params] --
  // three W buffers give the shared weight rows five phases of flight time (A rows are private per wave).
  constexpr int W_BASE = PP ? 2 * A_BYTES : A_BYTES;
  constexpr int PAR = lds_main_bytes(BM, BN, WM, WN, W4, PP != 0);  // LDS offset of the per-n epilogue vectors
  // PP == 3: the ping-pong main loop is the generated ISA of mq_gemm_pp_asm.inc (tools/gen_pp_asm.py): A fragments go
  // HBM/L2 -> AGPRs directly, accumulators live in AGPRs; K % 256 == 0.  Prologue and epilogue are the C++ below.
  constexpr bool ASMK = PP == 3;

  extern __shared__ __attribute__((aligned(16))) char smem[];

  const int lane = threadIdx.x & 63;
  const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const int wave_m = wave / WN, wave_n = wave % WN;

  unsigned long long ts0 = 0, ts1 = 0, ts2 = 0, tsp[5] = {0, 0, 0, 0, 0};
  (void)tsp;
  if constexpr (ABL & 16) ts0 = __builtin_readcyclecounter();
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
  unsigned src_a[A_ROUNDS], src_w[W_ROUNDS];   // byte offsets < 2^31 (checked by the host wrapper)
#pragma unroll
  for (int i = 0; i < A_ROUNDS; ++i) {
    // classic loop: instruction j = wave + i*NW (strided);  ping-pong: j = wave*A_ROUNDS + i, i.e. exactly the
    // rows this wave's own MFMAs consume (WN == 1), so its A slice of a stage buffer is private to it
    int row = m0 + (PP ? wave * A_ROUNDS + i : wave + i * NW) * 8 + (lane >> 3);
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

  // one DMA instruction: index d in [0, N_DMA) = A rounds first, then W rounds.  `buf` selects the stage
  // buffer (classic) or the A buffer (ping-pong); `wbuf` the W buffer of the ping-pong layout.
  auto issue_one = [&](int d, int buf, int kt, int wbuf = 0) {
    if (d < A_ROUNDS) {
      char* dst = PP ? smem + buf * A_BYTES + (wave * A_ROUNDS + d) * 1024
                     : smem + buf * STAGE + (wave + d * NW) * 1024;
      __builtin_amdgcn_global_load_lds(MQ_GLOBAL_PTR(a_ptr + (size_t)(src_a[d] + (unsigned)(kt * BK))), MQ_LDS_PTR(dst), 16, 0, 0);
    } else {
      const int i = d - A_ROUNDS;
      char* dst = PP ? smem + W_BASE + wbuf * W_BYTES + (wave + i * NW) * 1024
                     : smem + buf * STAGE + A_BYTES + (wave + i * NW) * 1024;
      if (!W_TAIL || i < W_ROUNDS - 1 || wave + i * NW < W_INSTR)
        __builtin_amdgcn_global_load_lds(MQ_GLOBAL_PTR(w_ptr + (size_t)(src_w[i] + (unsigned)(kt * WROW))), MQ_LDS_PTR(dst), 16, 0, 0);
    }
  };

  // ---- prologue ----------------------------------------------------------------------------------
  // Ordinary global loads first (per-n epilogue vectors, row sums), THEN the LDS-DMA of the first
  // stage(s): vmcnt retires in issue order, so waiting for the parameter loads does not drain the DMA.
  const int frow = lane & 15, kq = lane >> 4;
  // Straight-line, unpredicated loads (indices clamped; the host passes valid dummy pointers for absent
  // row sums / bias): with no control flow around them hipcc counts vmcnt exactly and the LDS-DMA issued
  // right after is not drained when their values are first used.
  int rs[FM];
#pragma unroll
  for (int i = 0; i < FM; ++i) {
    int m = m0 + wave_m * TM + i * 16 + frow;
    m = args.has_rowsum ? (m < M ? m : M - 1) : 0;
    rs[i] = args.a_rowsum[m];
  }
  // per-n epilogue vectors: global loads first, then the LDS-DMA of the first stage(s); the vectors are parked
  // in LDS with inline-asm ds_writes (a compiler-visible ds_write behind an in-flight LDS-DMA makes hipcc drain
  // every outstanding DMA first) so the accumulators can be INITIALISED with the integer zero-point
  // correction while the first stage is still in flight -- the epilogue then needs no integer arithmetic.
  constexpr int PR = (BN + 64 * NW - 1) / (64 * NW);
  float pa[PR], pb[PR];
  int pz[PR], pc[PR];
#pragma unroll
  for (int r = 0; r < PR; ++r) {
    int n = n0 + (int)threadIdx.x + r * 64 * NW;
    n = n < N ? n : N - 1;
    pa[r] = args.alpha[n];
    pb[r] = args.bias[n];
    pz[r] = args.w_zp[n];
    pc[r] = args.col_term[n];
  }
  // 8-bit output grids carry the (integer-valued, <= 255) offset inside the fma's addend -- for EVERY storage type, so the index
  // a linear produces does not depend on whether its consumer asked for indices or for the dequantised value; 16-bit grids keep
  // it out (an addend of up to 65535 would cost the fma 8 more fraction bits)
  const bool FOLD_OO = OUTQ && (args.out_qmax - args.out_qmin <= 255.0f);
  float so = 1.f, oo = 0.f, inv_so = 1.f;
  if constexpr (OUTQ) {
    so = args.out_scale[0];
    oo = args.out_offset[0];
    inv_so = __fdiv_rn(1.0f, so);
  }
  {
    // layout at PAR: alpha'[BN] | bias'[BN] | w_zp[BN] | col_term[BN]; with an output quantizer alpha' = alpha/so,
    // bias' = bias/so (+ oo for 8-bit storage: the offset <= 255 rides in the bias; 16-bit grids keep it out)
#pragma unroll
    for (int r = 0; r < PR; ++r) {
      const int tt = (int)threadIdx.x + r * 64 * NW;
      if (tt < BN) {
        float inv_c = inv_so, oo_c = oo;
        if constexpr (OUTQ && (OUT == MQ_U8 || OUT == MQ_I8)) {
          if (args.seg_scale[0] != nullptr) {          // this column's own output grid (block-uniform branch)
            const int n = n0 + tt;
            const int sg = (n >= args.seg_end[0]) + (args.seg_scale[1] != nullptr && n >= args.seg_end[1]);
            if (sg > 0) {
              inv_c = __fdiv_rn(1.0f, args.seg_scale[sg - 1][0]);
              oo_c = args.seg_offset[sg - 1][0];
            }
          }
        }
        const float a_v = OUTQ ? pa[r] * inv_c : pa[r];
        const float bias_v = args.has_bias ? pb[r] : 0.f;
        const float b_v = OUTQ ? (FOLD_OO ? bias_v * inv_c + oo_c : bias_v * inv_c) : bias_v;
        const unsigned base = (unsigned)(size_t)MQ_LDS_PTR(smem + PAR) + tt * 4;     // LDS byte address
        asm volatile("ds_write_b32 %0, %1\n\tds_write_b32 %0, %2 offset:%5\n\tds_write_b32 %0, %3 offset:%6\n\t"
                     "ds_write_b32 %0, %4 offset:%7"
                     :: "v"(base), "v"(a_v), "v"(b_v), "v"(pz[r]), "v"(pc[r]), "n"(BN * 4), "n"(BN * 8), "n"(BN * 12)
                     : "memory");
      }
    }
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
  }
  // first stage(s) go out now; the barrier that publishes the parked vectors follows the issue
#pragma unroll
  for (int d = ASMK ? A_ROUNDS : 0; d < N_DMA; ++d) issue_one(d, 0, 0, 0);     // ASMK: A never passes through the LDS
  if constexpr (PP) {
    if (KT > 1) {
#pragma unroll
      for (int d = ASMK ? A_ROUNDS : 0; d < N_DMA; ++d) issue_one(d, 1, 1, 1);
    }
#if MQ_PP_ASM_STAGE_MODE
    if constexpr (ASMK) {       // stage-granular loop: the waves of group 0 prefetch W three stages ahead
      if (KT > 2 && wave < 4) {
#pragma unroll
        for (int d = A_ROUNDS; d < N_DMA; ++d) issue_one(d, 0, 2, 2);
      }
    }
#endif
  }
  asm volatile("s_barrier" ::: "memory");

  // ---- ds_read offsets (per lane) -----------------------------------------------------------------
  // 128-byte rows: chunk (kq + 4*ks) ^ (row & 7); ks toggles bit 2 -> XOR 64 on the byte address
  const int x_off = (wave_m * TM + frow) * BK + ((kq ^ (lane & 7)) << 4);
  int w_off;
  if constexpr (W4) {
    w_off = A_BYTES + (wave_n * TN + frow) * WROW;   // group index added per use (depends on kq/ks)
  } else {
    w_off = W_BASE + (wave_n * TN + frow) * BK + ((kq ^ (lane & 7)) << 4);
  }

  const int x_off1 = x_off ^ 64, w_off1 = w_off ^ 64;
  (void)x_off1; (void)w_off1;

  // accumulators start at the zero-point correction  col_term[n] - w_zp[n] * a_rowsum[m]  (exact in int32
  // wrap-around arithmetic; |w_zp|, |row sum| < 2^23 so the 24-bit multiply is exact); the MFMAs add the
  // contraction on top.  Runs while the first LDS-DMA stage is in flight.
  v4i acc[FM][FN];
  {
    const v4i* p_zw0 = reinterpret_cast<const v4i*>(smem + PAR + BN * 8);
    const v4i* p_ct0 = reinterpret_cast<const v4i*>(smem + PAR + BN * 12);
#pragma unroll
    for (int j = 0; j < FN; ++j) {
      const int nl = wave_n * TN + j * 16 + kq * 4;
      const v4i zw = p_zw0[nl >> 2], ct = p_ct0[nl >> 2];
#pragma unroll
      for (int i = 0; i < FM; ++i)
#pragma unroll
        for (int e = 0; e < 4; ++e) acc[i][j][e] = (int)((unsigned)__mul24(-zw[e], rs[i]) + (unsigned)ct[e]);
    }
  }

  if constexpr (ASMK) {
    static_assert(!ASMK || (BM == 256 && BN == 176 && NW == 8 && WN == 1 && !W4 && FM == 2 && FN == 11), "generated loop: 256x176, 8x1");
    // accumulators -> AGPRs (22 per asm statement: operand limit), the generated loop, AGPRs -> accumulators
    int flat[88];
#pragma unroll
    for (int j = 0; j < FN; ++j)
#pragma unroll
      for (int i = 0; i < FM; ++i)
#pragma unroll
        for (int e = 0; e < 4; ++e) flat[(2 * j + i) * 4 + e] = acc[i][j][e];
#define MQ_OPS22(C, Q) Q(flat[C*22+0]), Q(flat[C*22+1]), Q(flat[C*22+2]), Q(flat[C*22+3]), Q(flat[C*22+4]), Q(flat[C*22+5]), Q(flat[C*22+6]), \
    Q(flat[C*22+7]), Q(flat[C*22+8]), Q(flat[C*22+9]), Q(flat[C*22+10]), Q(flat[C*22+11]), Q(flat[C*22+12]), Q(flat[C*22+13]),                  \
    Q(flat[C*22+14]), Q(flat[C*22+15]), Q(flat[C*22+16]), Q(flat[C*22+17]), Q(flat[C*22+18]), Q(flat[C*22+19]), Q(flat[C*22+20]), Q(flat[C*22+21])
#define MQ_IN(x) "v"(x)
#define MQ_OUT(x) "=v"(x)
    if constexpr (ABL & 16) ts1 = __builtin_readcyclecounter();
    asm volatile(MQ_PP_ASM_COPYIN0 ::MQ_OPS22(0, MQ_IN) : MQ_PP_ASM_ACLOBBERS);
    asm volatile(MQ_PP_ASM_COPYIN1 ::MQ_OPS22(1, MQ_IN) : MQ_PP_ASM_ACLOBBERS);
    asm volatile(MQ_PP_ASM_COPYIN2 ::MQ_OPS22(2, MQ_IN) : MQ_PP_ASM_ACLOBBERS);
    asm volatile(MQ_PP_ASM_COPYIN3 ::MQ_OPS22(3, MQ_IN) : MQ_PP_ASM_ACLOBBERS);
    {
// fragment-blocked A (mq_quantize_tiled): row block rb, k block kb at ((rb * K/64) + kb) KiB, lane-linear inside
      const unsigned rb_max = (unsigned)((M + 15) >> 4) - 1;
      unsigned rb0 = (unsigned)((m0 + wave_m * TM) >> 4), rb1 = rb0 + 1;
      rb0 = rb0 < rb_max ? rb0 : rb_max;
      rb1 = rb1 < rb_max ? rb1 : rb_max;
      const unsigned av0 = (rb0 * (unsigned)(K >> 6)) * 1024u + ((unsigned)lane << 4);
      const unsigned av1 = (rb1 * (unsigned)(K >> 6)) * 1024u + ((unsigned)lane << 4);
      const unsigned woff0 = (unsigned)w_off, woff1 = (unsigned)w_off1;
      asm volatile(MQ_PP_ASM_BODY
                   :
                   : [kt] "s"(KT), [wave] "s"(wave), [aptr] "s"(a_ptr), [wptr] "s"(w_ptr), [woff0] "v"(woff0), [woff1] "v"(woff1),
                     [av0] "v"(av0), [av1] "v"(av1), [sw0] "v"(src_w[0]), [sw1] "v"(src_w[1]), [sw2] "v"(src_w[2])
                   : MQ_PP_ASM_CLOBBERS);
    }
    asm volatile(MQ_PP_ASM_COPYOUT0 : MQ_OPS22(0, MQ_OUT) : : MQ_PP_ASM_ACLOBBERS);
    asm volatile(MQ_PP_ASM_COPYOUT1 : MQ_OPS22(1, MQ_OUT) : : MQ_PP_ASM_ACLOBBERS);
    asm volatile(MQ_PP_ASM_COPYOUT2 : MQ_OPS22(2, MQ_OUT) : : MQ_PP_ASM_ACLOBBERS);
    asm volatile(MQ_PP_ASM_COPYOUT3 : MQ_OPS22(3, MQ_OUT) : : MQ_PP_ASM_ACLOBBERS);
#undef MQ_OPS22
#undef MQ_IN
#undef MQ_OUT
#pragma unroll
    for (int j = 0; j < FN; ++j)
#pragma unroll
      for (int i = 0; i < FM; ++i)
#pragma unroll
        for (int e = 0; e < 4; ++e) acc[i][j][e] = flat[(2 * j + i) * 4 + e];
  } else if constexpr (PP) {
    // ---- ping-pong main loop (8 waves = two groups of four, one wave of each group per SIMD) --------
    // Unit of work u = (stage t, k-step ks) = one MFMA k-step of 64 over the wave's whole tile.
    //   phase E_u : group 0 issues the MFMAs of unit u from registers | group 1 ds_reads unit u
    //   phase O_u : group 0 ds_reads unit u+1                         | group 1 issues the MFMAs of unit u
    // The matrix pipe of every SIMD alternates between its group-0 and its group-1 wave and never waits for
    // LDS latency; MFMA phases contain nothing but MFMAs; one s_barrier ends every phase.
    // LDS-DMA rides in the READ phases only, A_ROUNDS or W_ROUNDS pieces per phase and wave:
    //   A(t+2): a wave's A rows are private (WN == 1); it refills them right after its own read of (t,1)
    //           -- group 0 in O_(t,0), group 1 in E_(t,1) -- no barrier involved, two A buffers.
    //   W(t+2): shared rows in a ring of THREE buffers; buffer (t+2)%3 held stage t-1 and is free after the
    //           barrier ending E_(t-1,1): group 0 issues its share in O_(t-1,1), group 1 in E_(t,0).
    // Retirement: before the barrier ending E_(t,1) every wave waits until only its W(t+2) and A(t+2) pieces
    // (the youngest W share + A_ROUNDS) are outstanding, i.e. its pieces of stage t+1 have landed; stage t+1
    // is first read in O_(t,1).  Every piece gets >= 5 phases of flight time.
    static_assert(NW == 8 && WN == 1 && !W4, "ping-pong schedule: 8 waves stacked along M, int8 weights");
    const int grp = wave >> 2;
    v4i xf[FM], wf[FN];
    bool skip_reads = false, skip_a = false;
    (void)skip_reads; (void)skip_a;
    auto read_unit = [&](int abuf, int wbuf, auto ks_tag) {
      constexpr int ks = decltype(ks_tag)::value;
#if MQ_PP_PRIO == 2
      __builtin_amdgcn_s_setprio(1);
#endif
      if constexpr (ABL & 8) {   // ablation: no fragment reads (only the first unit is read so registers are defined)
        if (skip_reads) return;
        skip_reads = true;
      }
      // (off + i*2048) ^ 64 == (off ^ 64) + i*2048: one base VGPR per operand and k-step, the fragment
      // index goes into the ds_read immediate offset
      const char* xb = smem + abuf * A_BYTES + (ks ? x_off1 : x_off);
      const char* wb = smem + wbuf * W_BYTES + (ks ? w_off1 : w_off);
      if (!((ABL & 32) && skip_a)) {
#pragma unroll
        for (int i = 0; i < FM; ++i) xf[i] = *reinterpret_cast<const v4i*>(xb + i * 16 * BK);
      }
      if constexpr (ABL & 32) skip_a = true;
#pragma unroll
      for (int j = 0; j < FN; ++j) wf[j] = *reinterpret_cast<const v4i*>(wb + j * 16 * BK);
    };
    auto all_frags_read = []() { asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory"); };
    auto mfma_unit = [&]() {
      if constexpr (ABL & 2) return;
#if MQ_PP_PRIO == 1
      __builtin_amdgcn_s_setprio(1);
#endif
#pragma unroll
      for (int j = 0; j < FN; ++j)
#pragma unroll
        for (int i = 0; i < FM; ++i) acc[i][j] = __builtin_amdgcn_mfma_i32_16x16x64_i8(wf[j], xf[i], acc[i][j], 0, 0, 0);
#if MQ_PP_PRIO == 1
      __builtin_amdgcn_s_setprio(0);
#endif
    };
    auto issue_a = [&](int abuf, int kt) {
      if constexpr (ABL & (1 | 32)) return;        // 32: "A operand leaves the LDS" what-if (no A DMA, no A reads)
#pragma unroll
      for (int d = 0; d < A_ROUNDS; ++d) issue_one(d, abuf, kt, 0);
    };
    auto issue_w = [&](int wbuf, int kt) {
      if constexpr (ABL & 1) return;
#pragma unroll
      for (int d = A_ROUNDS; d < N_DMA; ++d) issue_one(d, 0, kt, wbuf);
    };
    bool tail_owner = true;                                     // does this wave own a piece in the last W round?
    if constexpr (W_TAIL) tail_owner = wave + (W_ROUNDS - 1) * NW < W_INSTR;
    // wait until at most (a ? A_ROUNDS : 0) + (w ? this wave's W share : 0) DMA pieces are outstanding
    auto leave_in_flight = [&](bool a, bool w) {
      if constexpr (ABL & 1) return;
      if constexpr (ABL & 32) a = false;           // no A pieces are in flight in that what-if
      if (a && w) {
        if (tail_owner) asm volatile("s_waitcnt vmcnt(%0)" ::"n"(A_ROUNDS + W_ROUNDS) : "memory");
        else asm volatile("s_waitcnt vmcnt(%0)" ::"n"(A_ROUNDS + W_ROUNDS - 1) : "memory");
      } else if (a) {
        asm volatile("s_waitcnt vmcnt(%0)" ::"n"(A_ROUNDS) : "memory");
      } else if (w) {
        if (tail_owner) asm volatile("s_waitcnt vmcnt(%0)" ::"n"(W_ROUNDS) : "memory");
        else asm volatile("s_waitcnt vmcnt(%0)" ::"n"(W_ROUNDS - 1) : "memory");
      } else {
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
      }
    };
    auto phase_end = []() {
#if MQ_PP_PRIO == 2
      __builtin_amdgcn_s_setprio(0);
#endif
      asm volatile("s_barrier" ::: "memory");
    };
    using K0 = std::integral_constant<int, 0>;
    using K1 = std::integral_constant<int, 1>;

    // LDS-DMA schedule (pieces ride at the END of READ phases, after the phase's ds_reads are issued, with
    // no dependency wait in front of them; MFMA phases hold nothing but MFMAs):
    //   group 0:  O_(t,0): reads (t,1)   then W(t+2) -> W buffer (t+2)%3  [held stage t-1, consumed at E_(t-1,1)]
    //             O_(t,1): reads (t+1,0) then A(t+2) -> A buffer t&1      [own rows, read back in O_(t,0)]
    //   group 1:  E_(t,0): reads (t,0)   then A(t+1) -> A buffer (t+1)&1  [own rows, read back in E_(t-1,1)]
    //             E_(t,1): reads (t,1)   then W(t+2) -> W buffer (t+2)%3
    // Shared rows (W) must be retired before the barrier ending E_(t,1) (stage t+1 is first read in O_(t,1));
    // private rows (A) only before the owner's own read.  Every piece has >= 4 phases of flight time.
    leave_in_flight(KT > 1, KT > 1);                           // stage 0 landed; stage 1 may be in flight
    phase_end();
    if constexpr (ABL & 16) ts1 = __builtin_readcyclecounter();
    if (grp == 0) {
      read_unit(0, 0, K0{});
      all_frags_read();
      phase_end();                                             // O_(-1)
      int w3 = 0;                                              // t % 3
      for (int t = 0; t < KT; ++t) {
        const int w3n = w3 == 2 ? 0 : w3 + 1;                  // (t+1) % 3
        const int w3p = w3 == 0 ? 2 : w3 - 1;                  // (t+2) % 3
        const bool stamp = (ABL & 16) && t == 8;
        if (stamp) tsp[0] = __builtin_readcyclecounter();
        mfma_unit();                                           // E_(t,0)
        phase_end();
        if (stamp) tsp[1] = __builtin_readcyclecounter();
        read_unit(t & 1, w3, K1{});                            // O_(t,0)
        if (t + 2 < KT) issue_w(w3p, t + 2);
        all_frags_read();
        phase_end();
        if (stamp) tsp[2] = __builtin_readcyclecounter();
        mfma_unit();                                           // E_(t,1)
        leave_in_flight(false, t + 2 < KT);                    //   W(t+1) and A(t+1) of this wave have landed
        phase_end();
        if (stamp) tsp[3] = __builtin_readcyclecounter();
        if (t + 1 < KT) read_unit((t + 1) & 1, w3n, K0{});     // O_(t,1)
        if (t + 2 < KT) issue_a(t & 1, t + 2);
        all_frags_read();
        phase_end();
        if (stamp) tsp[4] = __builtin_readcyclecounter();
        w3 = w3n;
      }
    } else {
      phase_end();                                             // O_(-1): nothing to do yet
      int w3 = 0;
      for (int t = 0; t < KT; ++t) {
        const int w3n = w3 == 2 ? 0 : w3 + 1;
        const int w3p = w3 == 0 ? 2 : w3 - 1;
        const bool stamp = (ABL & 16) && t == 8;
        if (stamp) tsp[0] = __builtin_readcyclecounter();
        if (t > 0) leave_in_flight(false, t + 1 < KT);         // E_(t,0): own A(t) has landed (W(t+1) may fly)
        read_unit(t & 1, w3, K0{});
        if (t > 0 && t + 1 < KT) issue_a((t + 1) & 1, t + 1);
        all_frags_read();
        if (stamp) tsp[1] = __builtin_readcyclecounter();      //   (before the barrier: own work of the phase)
        phase_end();
        mfma_unit();                                           // O_(t,0)
        phase_end();
        if (stamp) tsp[2] = __builtin_readcyclecounter();
        read_unit(t & 1, w3, K1{});                            // E_(t,1)
        if (t + 2 < KT) issue_w(w3p, t + 2);
        all_frags_read();
        leave_in_flight(t > 0 && t + 1 < KT, t + 2 < KT);      //   this wave's W(t+1) pieces have landed
        if (stamp) tsp[3] = __builtin_readcyclecounter();
        phase_end();
        mfma_unit();                                           // O_(t,1)
        phase_end();
        if (stamp) tsp[4] = __builtin_readcyclecounter();
        w3 = w3n;
      }
    }
  } else {
  auto k_step = [&](int kt, auto more_tag) {
    constexpr bool more = decltype(more_tag)::value && !(ABL & 1);
    const int cur = kt & 1;
    // own DMA of stage kt retired, then everyone's; also: every wave has finished reading buf cur^1
    asm volatile("s_waitcnt vmcnt(0)\n\ts_barrier" ::: "memory");
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
  }

  if constexpr (ABL & 16) ts2 = __builtin_readcyclecounter();
  // ---- epilogue -----------------------------------------------------------------------------------
  // (1) park the per-n vectors in LDS (with the output quantizer folded in: q = rint(t*A' + B'),
  //     A' = alpha/so, B' = bias/so + oo);  (2) per 16-row block: dequant (+quantize) in registers,
  //     transpose through a wave-private LDS tile so that (3) every lane stores 16 contiguous bytes and
  //     a wave instruction writes whole rows -- instead of 16-byte fragments of 16 different rows.
  // every wave must be done with the stage buffers before they become staging tiles -- except for the generated-ISA
  // loop, whose A operand never passes through the LDS: the staging tiles (NW x 16 x ROWP bytes from offset 0) lie
  // inside the unused A region in front of the W ring, so a wave can start its epilogue while others still read W
  if constexpr (!ASMK) __syncthreads();
  if constexpr (ABL & 4) {
    if (acc[0][0][0] == 0x7fffffff) reinterpret_cast<int*>(args.out)[0] = 1;
    return;
  }
  using OT = typename OutT<OUT>::type;
  constexpr int ESZ = sizeof(OT);
  constexpr int ROWB = TN * ESZ;                                   // bytes of one output row of the wave tile
  constexpr int ROWP = ROWB + (((ROWB / 4) % 8 == 0) ? 16 : 0);    // padded pitch: keeps the LDS writes <= 2-way
  constexpr int CH = ROWB / 16;                                    // 16-byte chunks per row
  constexpr int EPC = 16 / ESZ;                                    // elements per chunk
  static_assert(NW * 16 * ROWP <= PAR, "staging tiles must fit in front of the epilogue vectors");
  char* stg = smem + wave * (16 * ROWP);
  const float qmin = args.out_qmin, qmax = args.out_qmax;
  const bool i8_unsigned_grid = (OUT == MQ_I8) && (args.out_qmin == 0.0f);
  const v4f* p_alpha = reinterpret_cast<const v4f*>(smem + PAR);
  const v4f* p_bias = reinterpret_cast<const v4f*>(smem + PAR + BN * 4);
  // unsigned 8-bit grid: v_cvt_pk_u8_f32 rounds to nearest even and saturates to [0,255] (measured:
  // tools/cvt_probe.cpp), i.e. it IS clamp(rint(v), qmin, qmax) -- no separate rint / med3
  const bool u8_grid = (OUT == MQ_U8 || OUT == MQ_I8) && args.out_qmin == 0.0f && args.out_qmax == 255.0f;
  const bool rows_vec = (N % EPC) == 0;        // 16-byte row stores need N*ESZ % 16 == 0 (always true for LLM shapes)
  OT* outp = reinterpret_cast<OT*>(args.out);
  // the unsigned 8-bit fast path (FOLD_OO && u8_grid, wave-uniform) is a separate instantiation of the block loop so
  // that it carries no rint / med3 / select per value (88 values per lane)
  auto store_tile = [&](auto fast_tag) {
    constexpr bool FAST = decltype(fast_tag)::value;
#pragma unroll
    for (int i = 0; i < FM; ++i) {
      const int mrow0 = m0 + wave_m * TM + i * 16;
#pragma unroll
      for (int j = 0; j < FN; ++j) {
        const int nl = wave_n * TN + j * 16 + kq * 4;   // column within the block tile
        const v4f al = p_alpha[nl >> 2];
        const v4f bs = p_bias[nl >> 2];
        float v[4];
#pragma unroll
        for (int e = 0; e < 4; ++e) {
          const int t = acc[i][j][e];              // contraction + zero-point correction (accumulator init)
          if constexpr (OUTQ) {
            float q = __builtin_fmaf((float)t, al[e], bs[e]);
            if constexpr (!FAST) {                 // the u8 fast path leaves rint and clamp to cvt_pk_u8
              q = rintf(q);
              if (!FOLD_OO) q += oo;
              q = __builtin_amdgcn_fmed3f(q, qmin, qmax);
            }
            if constexpr (OUT == MQ_F32 || OUT == MQ_F16) v[e] = __fmul_rn(__fsub_rn(q, oo), so);
            else v[e] = q;
          } else {
            v[e] = __fadd_rn(__fmul_rn((float)t, al[e]), bs[e]);
          }
        }
        char* dst = stg + frow * ROWP + (j * 16 + kq * 4) * ESZ;
        if constexpr (OUT == MQ_F32) {
          *reinterpret_cast<v4f*>(dst) = v4f{v[0], v[1], v[2], v[3]};
        } else if constexpr (OUT == MQ_F16) {
          struct alignas(8) H4 { __half h[4]; };
          H4 h;
#pragma unroll
          for (int e = 0; e < 4; ++e) h.h[e] = __float2half_rn(v[e]);
          *reinterpret_cast<H4*>(dst) = h;
        } else if constexpr (OUT == MQ_U8 || OUT == MQ_I8) {
          unsigned pk = 0;
          if (OUT == MQ_U8 || i8_unsigned_grid) {      // values in [0,255]; i8 storage = value - 128 = byte ^ 0x80
#pragma unroll
            for (int e = 0; e < 4; ++e) pk = __builtin_amdgcn_cvt_pk_u8_f32(v[e], e, pk);
            if (OUT == MQ_I8) pk ^= 0x80808080u;
          } else {
#pragma unroll
            for (int e = 0; e < 4; ++e) pk |= ((unsigned)(int)v[e] & 0xffu) << (8 * e);
          }
          *reinterpret_cast<unsigned*>(dst) = pk;
        } else {
          uint2 pk;
          pk.x = ((unsigned)(int)v[0] & 0xffffu) | (((unsigned)(int)v[1] & 0xffffu) << 16);
          pk.y = ((unsigned)(int)v[2] & 0xffffu) | (((unsigned)(int)v[3] & 0xffffu) << 16);
          *reinterpret_cast<uint2*>(dst) = pk;
        }
      }
      // the wave's own LDS accesses execute in order: its writes above are visible to its reads below
      if (rows_vec) {
#pragma unroll
        for (int c0 = 0; c0 < 16 * CH; c0 += 64) {
          const int c = c0 + lane;
          if (16 * CH % 64 == 0 || c < 16 * CH) {
            const int row = c / CH, ch = c - row * CH;
            const int m = mrow0 + row, n = n0 + wave_n * TN + ch * EPC;
            v4i val = *reinterpret_cast<const v4i*>(stg + row * ROWP + ch * 16);
            if (m < M && n < N) {
              if constexpr (OUT == MQ_F32) {
                if (args.resid != nullptr) {           // x + Qout(linear): the residual add of the decoder layer, plain fp32 add
                  const v4f r = *reinterpret_cast<const v4f*>(args.resid + (size_t)m * N + n);
                  v4f f = __builtin_bit_cast(v4f, val);
                  f[0] = __fadd_rn(r[0], f[0]); f[1] = __fadd_rn(r[1], f[1]); f[2] = __fadd_rn(r[2], f[2]); f[3] = __fadd_rn(r[3], f[3]);
                  val = __builtin_bit_cast(v4i, f);
                }
              }
              __builtin_nontemporal_store(val, reinterpret_cast<v4i*>(outp + (size_t)m * N + n));
            }
          }
        }
      } else {
        for (int c = lane; c < 16 * TN; c += 64) {        // ragged N: element-wise
          const int row = c / TN, col = c - row * TN;
          const int m = mrow0 + row, n = n0 + wave_n * TN + col;
          if (m < M && n < N) {
            OT o = *reinterpret_cast<const OT*>(stg + row * ROWP + col * ESZ);
            if constexpr (OUT == MQ_F32) {
              if (args.resid != nullptr) o = __fadd_rn(args.resid[(size_t)m * N + n], o);
            }
            outp[(size_t)m * N + n] = o;
          }
        }
      }
    }
  };
  if constexpr (OUTQ && (OUT == MQ_U8 || OUT == MQ_I8)) {
    if (u8_grid) store_tile(std::true_type{});
    else store_tile(std::false_type{});
  } else {
    store_tile(std::false_type{});
  }
  if constexpr (ABL & 16) {
    if (args.dbg_ts != nullptr && lane == 0) {
      asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
      unsigned long long* d = args.dbg_ts + ((size_t)blockIdx.x * NW + wave) * 16;
      d[0] = ts0; d[1] = ts1; d[2] = ts2; d[3] = __builtin_readcyclecounter();
      for (int q = 0; q < 5; ++q) d[4 + q] = tsp[q];
    }
  }
}


// ---- free-running generated kernel (tools/gen_fr_asm.py -> mq_gemm_fr_asm.inc) ---------------------------------------
// The whole workgroup program (prologue, software-pipelined main loop, epilogue) is generated gfx950 ISA; C++ only forms
// the per-lane addresses and the scalar arguments.  256 x 176 tile, fragment-blocked activations, int8 weights, 8-bit
// UNSIGNED output grid (u8 storage, or i8 storage = index - 128), K % 256 == 0, K >= 768.
__device__ __forceinline__ void gemm_i8_fr_body(const GemmArgs& args, int bid, int nblk) {
  // Nothing in front of the generated program may wait for memory beyond the kernel arguments: the first LDS-DMA requests leave as soon
  // as the tile's addresses are formed (the output grid is loaded and inverted INSIDE the program, behind them; the row-sum zeroing of
  // the gated pair's first launch follows the program).
#if MQ_FR_ASM_STAMP
  const unsigned long long t_entry = __builtin_amdgcn_s_memrealtime();
#endif
  const int lane = threadIdx.x & 63;
  const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  int tm, tn;
  tile_of_block(bid, nblk, args.grid_m, args.grid_n, tm, tn);
  const int m0 = tm * 256, n0 = tn * 176;
  const int M = args.M, N = args.N, K = args.K;
  const int KT = K / BK;
  // LDS-DMA source offsets of this wave's W pieces (8 rows x 128 B each; piece wave + 8 i), XOR-swizzled like the other variants
  unsigned sw[3];
#pragma unroll
  for (int i = 0; i < 3; ++i) {
    int row = n0 + (wave + i * 8) * 8 + (lane >> 3);
    row = row < N ? row : N - 1;
    sw[i] = (unsigned)row * (unsigned)K + (unsigned)(((lane & 7) ^ (lane >> 3)) << 4);
  }
  // fragment-blocked A (mq_quantize_tiled): row block rb, k block kb at ((rb * K/64) + kb) KiB, lane-linear inside
  const int m0w = m0 + wave * 32;
  const unsigned rb_max = (unsigned)((M + 15) >> 4) - 1;
  unsigned rb0 = (unsigned)(m0w >> 4), rb1 = rb0 + 1;
  rb0 = rb0 < rb_max ? rb0 : rb_max;
  rb1 = rb1 < rb_max ? rb1 : rb_max;
  const unsigned av0 = (rb0 * (unsigned)(K >> 6)) * 1024u + ((unsigned)lane << 4);
  const unsigned av1 = (rb1 * (unsigned)(K >> 6)) * 1024u + ((unsigned)lane << 4);
  unsigned rsofs[2];
#pragma unroll
  for (int i = 0; i < 2; ++i) {
    int m = m0w + i * 16 + (lane & 15);
    m = args.has_rowsum ? (m < M ? m : M - 1) : 0;
    rsofs[i] = (unsigned)m * 4u;
  }
  const float* so_ptr = args.out_scale;
  const float* oo_ptr = args.out_offset;
  const int8_t* a_ptr = args.a;
  const int8_t* w_ptr = reinterpret_cast<const int8_t*>(args.w);
  const float* alpha_p = args.alpha + n0;
  const float* bias_p = args.bias + n0;
  const int32_t* wzp_p = args.w_zp + n0;
  const int32_t* ct_p = args.col_term + n0;
  const int32_t* rs_p = args.a_rowsum;
  uint8_t* outw = reinterpret_cast<uint8_t*>(args.out) + (size_t)m0w * N + n0;
  // (explicit readfirstlane: an "s" constraint alone does not stop hipcc from handing over a VGPR)
  const int mrem = __builtin_amdgcn_readfirstlane(M - m0w);
  const int flags = __builtin_amdgcn_readfirstlane((args.has_bias ? 1 : 0) | (args.has_rowsum ? 2 : 0));
  const int xorv = __builtin_amdgcn_readfirstlane(args.out_dtype == MQ_I8 ? (int)0x80808080u : 0);
  const int ldn = __builtin_amdgcn_readfirstlane(N), kt = __builtin_amdgcn_readfirstlane(KT);
  const unsigned tid = threadIdx.x;
#if MQ_FR_ASM_PROBE
  // clock probe (mq_gemm_set_clock_probe): 8 bytes per wave = [shader cycles, 100-MHz real-time ticks] of the generated program
  unsigned long long* dbg = args.dbg_ts ? args.dbg_ts + ((size_t)bid * 8 + wave) : nullptr;
#endif
#if MQ_FR_ASM_STAMP
  unsigned long long* dbg = args.dbg_ts + ((size_t)bid * 8 + wave) * 16;
  const int te_lo = __builtin_amdgcn_readfirstlane((int)(unsigned)t_entry), te_hi = __builtin_amdgcn_readfirstlane((int)(unsigned)(t_entry >> 32));
#endif
  asm volatile(MQ_FR_ASM_BODY
               :
               : [kt] "s"(kt), [wave] "s"(wave), [aptr] "s"(a_ptr), [wptr] "s"(w_ptr), [outw] "s"(outw), [alpha] "s"(alpha_p),
                 [bias] "s"(bias_p), [wzp] "s"(wzp_p), [ct] "s"(ct_p), [rsptr] "s"(rs_p), [soptr] "s"(so_ptr), [ooptr] "s"(oo_ptr),
                 [ldn] "s"(ldn), [mrem] "s"(mrem), [flags] "s"(flags), [xorv] "s"(xorv),
#if MQ_FR_ASM_STAMP
                 [dbg] "s"(dbg), [tentry_lo] "s"(te_lo), [tentry_hi] "s"(te_hi),
#endif
#if MQ_FR_ASM_PROBE
                 [dbg] "s"(dbg),
#endif
                 [sw0] "v"(sw[0]), [sw1] "v"(sw[1]), [sw2] "v"(sw[2]), [av0] "v"(av0), [av1] "v"(av1), [tid] "v"(tid),
                 [rsofs0] "v"(rsofs[0]), [rsofs1] "v"(rsofs[1])
               : MQ_FR_ASM_CLOBBERS);
  if (args.zero_buf != nullptr) {                                   // (gated pair, first launch: 512 ints per workgroup)
    const int zi = bid * 512 + (int)threadIdx.x;
    if (zi < args.zero_count) args.zero_buf[zi] = 0;
  }
}

__global__ void __launch_bounds__(512) gemm_i8_fr_kernel(const GemmArgs args) { gemm_i8_fr_body(args, blockIdx.x, args.grid_m * args.grid_n); }

// ---- the free-running program on 128-column tiles (tools/gen_fr_asm.py variants fr128 / fr128r / fr128r8) ------------------------------
// N = 2048 / 2560 outputs do not tile by 176.  FR128: 256 x 128 tiles, eight waves, 8-bit unsigned output grid PER COLUMN (the q | k | v
// segments of mq_w8a8_linear_tiled_segmented).  FR128R: 128 x 128 tiles, FOUR waves (one per SIMD; 2048 x 2048 outputs = 256 tiles = one
// per CU), fp32 output x + Q16(linear) with the residual add in the store (o_proj / w2).  FR128R8: that epilogue on 256 x 128 tiles.
// FR160: 128 x 160 tiles, four waves, per-column 8-bit grids: q | k | v at M = 2048 is 16 x 16 = 256 tiles, one per CU (256 x 128 tiles: 160).
enum { FR128 = 1, FR128R = 2, FR128R8 = 3, FR160 = 4, FR128RS = 5 };   // FR128RS: FR128R8's tile with the K loop split over two workgroups
template <int VAR>
__device__ __forceinline__ void gemm_i8_fr128_body(const GemmArgs& args) {
  constexpr int NWV = (VAR == FR128R || VAR == FR160) ? 4 : 8;
  constexpr int BMT = 32 * NWV;
  constexpr int BNT = VAR == FR160 ? 160 : 128;
  constexpr int PCS = BNT / 8 / NWV;            // W LDS-DMA pieces (8 rows x 128 B) per wave and stage
  const int lane = threadIdx.x & 63;
  const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  int tm, tn;
  const int ntiles = args.grid_m * args.grid_n;
  // split-K (FR128RS): blocks [0, tiles) take the first half of K, blocks [tiles, 2 tiles) the second (dispatched after every first-half
  // workgroup: the partner a wave waits for is always resident or finished)
  const int role = VAR == FR128RS ? __builtin_amdgcn_readfirstlane((int)blockIdx.x >= ntiles ? 1 : 0) : 0;
  const int tile = (int)blockIdx.x - role * ntiles;
  tile_of_block(tile, ntiles, args.grid_m, args.grid_n, tm, tn, args.group_m);
  const int m0 = tm * BMT, n0 = tn * BNT;
  const int M = args.M, N = args.N, K = args.K;
  const int KT = (VAR == FR128RS ? K / 2 : K) / BK;
  unsigned sw[5] = {0, 0, 0, 0, 0};
#pragma unroll
  for (int i = 0; i < PCS; ++i) {
    int row = n0 + (wave + i * NWV) * 8 + (lane >> 3);
    row = row < N ? row : N - 1;
    sw[i] = (unsigned)row * (unsigned)K + (unsigned)(((lane & 7) ^ (lane >> 3)) << 4);
  }
  const int m0w = m0 + wave * 32;
  const unsigned rb_max = (unsigned)((M + 15) >> 4) - 1;
  unsigned rb0 = (unsigned)(m0w >> 4), rb1 = rb0 + 1;
  rb0 = rb0 < rb_max ? rb0 : rb_max;
  rb1 = rb1 < rb_max ? rb1 : rb_max;
  const unsigned av0 = (rb0 * (unsigned)(K >> 6)) * 1024u + ((unsigned)lane << 4);
  const unsigned av1 = (rb1 * (unsigned)(K >> 6)) * 1024u + ((unsigned)lane << 4);
  unsigned rsofs[2];
#pragma unroll
  for (int i = 0; i < 2; ++i) {
    int m = m0w + i * 16 + (lane & 15);
    m = args.has_rowsum ? (m < M ? m : M - 1) : 0;
    rsofs[i] = (unsigned)m * 4u;
  }
  // second K half: k blocks K / 128 .. of the fragment-blocked image (1 KiB each), byte K / 2 of every weight row
  const int8_t* a_ptr = args.a + (size_t)role * (size_t)(K >> 7) * 1024u;
  const int8_t* w_ptr = reinterpret_cast<const int8_t*>(args.w) + (size_t)role * (size_t)(K >> 1);
  const float* alpha_p = args.alpha + n0;
  const float* bias_p = args.bias + n0;
  const int32_t* wzp_p = args.w_zp + n0;
  const int32_t* ct_p = args.col_term + n0;
  const int32_t* rs_p = args.a_rowsum;
  const int mrem = __builtin_amdgcn_readfirstlane(M - m0w);
  const int flags = __builtin_amdgcn_readfirstlane((args.has_bias ? 1 : 0) | (args.has_rowsum ? 2 : 0) | (role ? 4 : 0));
  const int ldn = __builtin_amdgcn_readfirstlane(N), kt = __builtin_amdgcn_readfirstlane(KT);
  const unsigned tid = threadIdx.x;
  if constexpr (VAR == FR128 || VAR == FR160) {
    // this thread's column (tid < BNT) and its output grid: segment 0 = out_scale / out_offset, later segments their own
    const int n = n0 + (int)(tid < (unsigned)BNT ? tid : (unsigned)BNT - 1u);
    float sc = args.out_scale[0], ooc = args.out_offset[0];
    if (args.seg_scale[0] != nullptr) {
      const int sg = (n >= args.seg_end[0]) + (args.seg_scale[1] != nullptr && n >= args.seg_end[1]);
      if (sg > 0) {
        sc = args.seg_scale[sg - 1][0];
        ooc = args.seg_offset[sg - 1][0];
      }
    }
    const float invc = __fdiv_rn(1.0f, sc);
    uint8_t* outw = reinterpret_cast<uint8_t*>(args.out) + (size_t)m0w * N + n0;
    const int xorv = __builtin_amdgcn_readfirstlane(args.out_dtype == MQ_I8 ? (int)0x80808080u : 0);
#define MQ_FR128U_OPERANDS                                                                                                        \
    [kt] "s"(kt), [wave] "s"(wave), [aptr] "s"(a_ptr), [wptr] "s"(w_ptr), [outw] "s"(outw), [alpha] "s"(alpha_p),                  \
        [bias] "s"(bias_p), [wzp] "s"(wzp_p), [ct] "s"(ct_p), [rsptr] "s"(rs_p), [ldn] "s"(ldn), [mrem] "s"(mrem),                 \
        [flags] "s"(flags), [xorv] "s"(xorv), [sw0] "v"(sw[0]), [sw1] "v"(sw[1]), [av0] "v"(av0), [av1] "v"(av1),                  \
        [tid] "v"(tid), [rsofs0] "v"(rsofs[0]), [rsofs1] "v"(rsofs[1]), [invc] "v"(invc), [ooc] "v"(ooc)
    if constexpr (VAR == FR128) {
      asm volatile(MQ_FR128_ASM_BODY : : MQ_FR128U_OPERANDS : MQ_FR128_ASM_CLOBBERS);
    } else {
      asm volatile(MQ_FR160_ASM_BODY : : MQ_FR128U_OPERANDS, [sw2] "v"(sw[2]), [sw3] "v"(sw[3]), [sw4] "v"(sw[4]) : MQ_FR160_ASM_CLOBBERS);
    }
#undef MQ_FR128U_OPERANDS
  } else {
    const float* so_ptr = args.out_scale;
    const float* oo_ptr = args.out_offset;
    const int qmin_bits = __builtin_amdgcn_readfirstlane(__float_as_int(args.out_qmin));
    const int qmax_bits = __builtin_amdgcn_readfirstlane(__float_as_int(args.out_qmax));
    float* outw = reinterpret_cast<float*>(args.out) + (size_t)m0w * N + n0;
    const float* resid = args.resid + (size_t)m0w * N + n0;
#define MQ_FR128R_OPERANDS                                                                                                          \
    [kt] "s"(kt), [wave] "s"(wave), [aptr] "s"(a_ptr), [wptr] "s"(w_ptr), [outw] "s"(outw), [resid] "s"(resid), [alpha] "s"(alpha_p), \
        [bias] "s"(bias_p), [wzp] "s"(wzp_p), [ct] "s"(ct_p), [rsptr] "s"(rs_p), [soptr] "s"(so_ptr), [ooptr] "s"(oo_ptr),       \
        [qmin] "s"(qmin_bits), [qmax] "s"(qmax_bits), [ldn] "s"(ldn), [mrem] "s"(mrem), [flags] "s"(flags),      \
        [av0] "v"(av0), [av1] "v"(av1), [tid] "v"(tid), [rsofs0] "v"(rsofs[0]), [rsofs1] "v"(rsofs[1]), [sw0] "v"(sw[0]), [sw1] "v"(sw[1])
    if constexpr (VAR == FR128R) {
      asm volatile(MQ_FR128R_ASM_BODY : : MQ_FR128R_OPERANDS, [sw2] "v"(sw[2]), [sw3] "v"(sw[3]) : MQ_FR128R_ASM_CLOBBERS);
#ifdef MQ_BUILD_EXPERIMENTS
    } else if constexpr (VAR == FR128RS) {
      // the tile's exchange area: 2 x 8 x 8 KiB of partial sums, 2 x 8 flags (gemm_splitk_scratch)
      const char* xch = reinterpret_cast<const char*>(args.gate_q) + (size_t)tile * 131072u;
      const int* xfl = reinterpret_cast<const int*>(args.gate_rowsum) + (size_t)tile * 16u;
      asm volatile(MQ_FR128RS_ASM_BODY : : MQ_FR128R_OPERANDS, [xch] "s"(xch), [xfl] "s"(xfl) : MQ_FR128RS_ASM_CLOBBERS);
#endif
    } else {
      asm volatile(MQ_FR128R8_ASM_BODY : : MQ_FR128R_OPERANDS : MQ_FR128R8_ASM_CLOBBERS);
    }
#undef MQ_FR128R_OPERANDS
  }
  if (args.zero_buf != nullptr) {               // (gated pair, first launch)
    const int zi = (int)blockIdx.x * (64 * NWV) + (int)threadIdx.x;
    if (zi < args.zero_count) args.zero_buf[zi] = 0;
  }
}

template <int VAR>
__global__ void __launch_bounds__((VAR == FR128R || VAR == FR160) ? 256 : 512) gemm_i8_fr128_kernel(const GemmArgs args) { gemm_i8_fr128_body<VAR>(args); }

// shapes the 128-column generated kernels serve (fragment-blocked activations, int8 weights)
static bool gemm_fr128_shape(int64_t M, int64_t N, int64_t K) { return M > 0 && N % 128 == 0 && K % 256 == 0 && K >= 768; }

static int device_cu_count() {
  static std::atomic<int> cus[64];
  const int dev = current_device();
  if (dev < 0 || dev >= 64) return 0;
  int c = cus[dev].load();
  if (c == 0) {
    hipDeviceProp_t prop;
    c = (hipGetDeviceProperties(&prop, dev) == hipSuccess && prop.multiProcessorCount > 0) ? prop.multiProcessorCount : 1;
    cus[dev] = c;
  }
  return c;
}

// Exchange area of the split-K residual GEMM: per device, grown on demand, never freed (128 tiles = 16.8 MB at M = 2048).  The flags
// are zero between launches (every wave clears the flag it consumed).
static int gemm_splitk_scratch(int tiles, char** xch, int** xfl) {
  static std::mutex mu;
  static char* buf[64] = {};
  static int cap[64] = {};
  const int dev = current_device();
  std::lock_guard<std::mutex> lock(mu);
  if (dev < 0 || dev >= 64) return MQ_EHIP;
  if (cap[dev] < tiles) {
    const size_t bytes = (size_t)tiles * (131072u + 64u);
    char* p = nullptr;
    if (hipMalloc(&p, bytes) != hipSuccess || hipMemset(p, 0, bytes) != hipSuccess || hipDeviceSynchronize() != hipSuccess) {
      set_error("mq_gemm: split-K exchange buffer (%zu bytes): allocation failed", bytes);
      return MQ_EHIP;
    }
    buf[dev] = p;                       // (an older, smaller buffer may still be in use by a launch in flight: it is leaked, once per growth)
    cap[dev] = tiles;
  }
  *xch = buf[dev];
  *xfl = reinterpret_cast<int*>(buf[dev] + (size_t)cap[dev] * 131072u);
  return MQ_OK;
}

template <int VAR>
static int launch_fr128(GemmArgs a, hipStream_t st) {
  constexpr int LDS = VAR == FR128 ? MQ_FR128_LDS_BYTES
                                   : (VAR == FR128R ? MQ_FR128R_LDS_BYTES
                                                    : (VAR == FR160 ? MQ_FR160_LDS_BYTES : (VAR == FR128RS ? MQ_FR128RS_LDS_BYTES : MQ_FR128R8_LDS_BYTES)));
  constexpr int BMT = (VAR == FR128R || VAR == FR160) ? 128 : 256;
  constexpr int BNT = VAR == FR160 ? 160 : 128;
  static PerDeviceOnce attr_set;
  const int dev = current_device();
  if (!attr_set.done(dev)) {
    hipError_t e = hipFuncSetAttribute((const void*)gemm_i8_fr128_kernel<VAR>, hipFuncAttributeMaxDynamicSharedMemorySize, LDS);
    if (e != hipSuccess) {
      set_error("mq_gemm: hipFuncSetAttribute(%d B LDS): %s", LDS, hipGetErrorString(e));
      return MQ_EHIP;
    }
    attr_set.mark(dev);
  }
  a.has_rowsum = a.a_rowsum != nullptr;
  if (a.a_rowsum == nullptr) a.a_rowsum = a.col_term;
  if (a.bias == nullptr) a.bias = a.alpha;
  a.grid_m = (a.M + BMT - 1) / BMT;
  a.grid_n = a.N / BNT;
  a.group_m = g_group_m.load();
  if constexpr (VAR == FR128RS) {
    char* xch = nullptr;
    int* xfl = nullptr;
    const int rc = gemm_splitk_scratch(a.grid_m * a.grid_n, &xch, &xfl);
    if (rc != MQ_OK) return rc;
    a.gate_q = reinterpret_cast<int8_t*>(xch);            // (fields of the gated pair, unused by the residual GEMMs)
    a.gate_rowsum = reinterpret_cast<int32_t*>(xfl);
    gemm_i8_fr128_kernel<VAR><<<2 * a.grid_m * a.grid_n, 512, LDS, st>>>(a);
    MQ_LAUNCH_CHECK("mq_gemm");
    return MQ_OK;
  }
  gemm_i8_fr128_kernel<VAR><<<a.grid_m * a.grid_n, (VAR == FR128R || VAR == FR160) ? 256 : 512, LDS, st>>>(a);
  MQ_LAUNCH_CHECK("mq_gemm");
  return MQ_OK;
}

// ---- packed 4-bit weights on the free-running program (tools/gen_fr_asm.py variants frw4 / frw4_128, round 4) --------------------------
// mq_pack_w4's image (two unsigned nibbles per byte; 16 bytes = 32 consecutive k, element p low / p + 16 high in byte p) goes through the
// LDS ring as it is (LDS-DMA pieces of 16 rows x 64 B); a lane's single ds_read_b128 per 16 columns and stage holds both of its MFMA
// operands of that stage, split in registers.  The activation fragments are gathered to match (type A = k 32 q + 0..15, type B = + 16..31
// of the stage: per-lane offsets below).  256 x BNT tiles, eight waves, 8-bit unsigned output grid (BNT = 176: one grid; 128: per column).
// X (frw4x / frw4x_128): the packed pieces go wave-privately through registers, are expanded ONCE per workgroup into the int8 W ring, and
// the loop is the int8 kernel's (standard activation fragments); !X (frw4 / frw4_128): every wave unpacks its own fragments.
template <int BNT, bool X>
__global__ void __launch_bounds__(512) gemm_i8_frw4_kernel(const GemmArgs args) {
  constexpr int FNT = BNT / 16;
  constexpr int PCS = (FNT + 7) / 8;
  const int lane = threadIdx.x & 63;
  const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  int tm, tn;
  tile_of_block(blockIdx.x, args.grid_m * args.grid_n, args.grid_m, args.grid_n, tm, tn);
  const int m0 = tm * 256, n0 = tn * BNT;
  const int M = args.M, N = args.N, K = args.K;
  const int KT = K / BK;
  unsigned sw[3] = {0, 0, 0};
#pragma unroll
  for (int i = 0; i < PCS; ++i) {
    const int r = lane >> 2;                                   // row of the piece (16 rows x 64 B)
    int piece = wave + i * 8;
    piece = piece < FNT ? piece : FNT - 1;
    int row = n0 + piece * 16 + r;
    row = row < N ? row : N - 1;
    const int g = X ? 0 : (4 - ((r >> 2) & 3)) & 3;            // LDS image of the packed rows: {0, 3, 2, 1}[(r >> 2) & 3]; X: no LDS image
    sw[i] = (unsigned)row * (unsigned)(K >> 1) + (unsigned)((((lane & 3) ^ g)) << 4);
  }
  const int m0w = m0 + wave * 32;
  const unsigned rb_max = (unsigned)((M + 15) >> 4) - 1;
  unsigned rb0 = (unsigned)(m0w >> 4), rb1 = rb0 + 1;
  rb0 = rb0 < rb_max ? rb0 : rb_max;
  rb1 = rb1 < rb_max ? rb1 : rb_max;
  // type-A fragment of lane (frow, kq): x[row, 128 s + 32 kq + 0..15] = k block 2 s + (kq >> 1), quarter 2 (kq & 1); type B: + 256 bytes
  const unsigned frow = (unsigned)lane & 15u, kq = (unsigned)lane >> 4;
  const unsigned lofs = X ? ((unsigned)lane << 4) : (kq >> 1) * 1024u + ((2u * (kq & 1u)) * 16u + frow) * 16u;
  const unsigned av0 = (rb0 * (unsigned)(K >> 6)) * 1024u + lofs;
  const unsigned av1 = (rb1 * (unsigned)(K >> 6)) * 1024u + lofs;
  unsigned rsofs[2];
#pragma unroll
  for (int i = 0; i < 2; ++i) {
    int m = m0w + i * 16 + (lane & 15);
    m = args.has_rowsum ? (m < M ? m : M - 1) : 0;
    rsofs[i] = (unsigned)m * 4u;
  }
  const int8_t* a_ptr = args.a;
  const int8_t* w_ptr = reinterpret_cast<const int8_t*>(args.w);
  const float* alpha_p = args.alpha + n0;
  const float* bias_p = args.bias + n0;
  const int32_t* wzp_p = args.w_zp + n0;
  const int32_t* ct_p = args.col_term + n0;
  const int32_t* rs_p = args.a_rowsum;
  uint8_t* outw = reinterpret_cast<uint8_t*>(args.out) + (size_t)m0w * N + n0;
  const int mrem = __builtin_amdgcn_readfirstlane(M - m0w);
  const int flags = __builtin_amdgcn_readfirstlane((args.has_bias ? 1 : 0) | (args.has_rowsum ? 2 : 0));
  const int xorv = __builtin_amdgcn_readfirstlane(args.out_dtype == MQ_I8 ? (int)0x80808080u : 0);
  const int ldn = __builtin_amdgcn_readfirstlane(N), kt = __builtin_amdgcn_readfirstlane(KT);
  const unsigned tid = threadIdx.x;
#define MQ_FRW4_OPERANDS                                                                                                          \
  [kt] "s"(kt), [wave] "s"(wave), [aptr] "s"(a_ptr), [wptr] "s"(w_ptr), [outw] "s"(outw), [alpha] "s"(alpha_p), [bias] "s"(bias_p),  \
      [wzp] "s"(wzp_p), [ct] "s"(ct_p), [rsptr] "s"(rs_p), [ldn] "s"(ldn), [mrem] "s"(mrem), [flags] "s"(flags), [xorv] "s"(xorv),   \
      [av0] "v"(av0), [av1] "v"(av1), [tid] "v"(tid), [rsofs0] "v"(rsofs[0]), [rsofs1] "v"(rsofs[1])
  if constexpr (BNT == 176) {
    const float* so_ptr = args.out_scale;
    const float* oo_ptr = args.out_offset;
    if constexpr (X) {
      asm volatile(MQ_FRW4X_ASM_BODY
                   : [sw0] "+v"(sw[0]), [sw1] "+v"(sw[1]), [sw2] "+v"(sw[2])
                   : MQ_FRW4_OPERANDS, [soptr] "s"(so_ptr), [ooptr] "s"(oo_ptr)
                   : MQ_FRW4X_ASM_CLOBBERS);
    } else {
#ifdef MQ_BUILD_EXPERIMENTS
      asm volatile(MQ_FRW4_ASM_BODY
                   : [sw0] "+v"(sw[0]), [sw1] "+v"(sw[1]), [sw2] "+v"(sw[2])
                   : MQ_FRW4_OPERANDS, [soptr] "s"(so_ptr), [ooptr] "s"(oo_ptr)
                   : MQ_FRW4_ASM_CLOBBERS);
#endif
    }
  } else {
    const int n = n0 + (int)(tid < (unsigned)BNT ? tid : (unsigned)BNT - 1u);
    float sc = args.out_scale[0], ooc = args.out_offset[0];
    if (args.seg_scale[0] != nullptr) {
      const int sg = (n >= args.seg_end[0]) + (args.seg_scale[1] != nullptr && n >= args.seg_end[1]);
      if (sg > 0) {
        sc = args.seg_scale[sg - 1][0];
        ooc = args.seg_offset[sg - 1][0];
      }
    }
    const float invc = __fdiv_rn(1.0f, sc);
    if constexpr (X) {
      asm volatile(MQ_FRW4X_128_ASM_BODY
                   : [sw0] "+v"(sw[0]), [sw1] "+v"(sw[1]), [sw2] "+v"(sw[2])
                   : MQ_FRW4_OPERANDS, [invc] "v"(invc), [ooc] "v"(ooc)
                   : MQ_FRW4X_128_ASM_CLOBBERS);
    } else {
#ifdef MQ_BUILD_EXPERIMENTS
      asm volatile(MQ_FRW4_128_ASM_BODY
                   : [sw0] "+v"(sw[0]), [sw1] "+v"(sw[1]), [sw2] "+v"(sw[2])
                   : MQ_FRW4_OPERANDS, [invc] "v"(invc), [ooc] "v"(ooc)
                   : MQ_FRW4_128_ASM_CLOBBERS);
#endif
    }
  }
#undef MQ_FRW4_OPERANDS
  if (args.zero_buf != nullptr) {               // (packed gated pair, first launch: 512 ints per workgroup)
    const int zi = (int)blockIdx.x * 512 + (int)threadIdx.x;
    if (zi < args.zero_count) args.zero_buf[zi] = 0;
  }
}

// Packed 4-bit weights in front of the RESIDUAL epilogue (tools/gen_fr_asm.py variant frw4x_128r): o_proj / w2 from the mq_pack_w4 image
// on 128 x 128 tiles, four waves -- fr128r's program with the W pieces (16 rows x 64 B, two per wave and stage) loaded into registers,
// split and written into the int8 ring once per workgroup.  fp32 out = resid + Q16(linear).
__global__ void __launch_bounds__(256) gemm_i8_frw4r_kernel(const GemmArgs args) {
  const int lane = threadIdx.x & 63;
  const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  int tm, tn;
  tile_of_block(blockIdx.x, args.grid_m * args.grid_n, args.grid_m, args.grid_n, tm, tn, args.group_m);
  const int m0 = tm * 128, n0 = tn * 128;
  const int M = args.M, N = args.N, K = args.K;
  const int KT = K / BK;
  unsigned sw[2];
#pragma unroll
  for (int i = 0; i < 2; ++i) {
    int row = n0 + (wave + i * 4) * 16 + (lane >> 2);          // piece wave + 4 i: 16 rows x 64 packed bytes per stage
    row = row < N ? row : N - 1;
    sw[i] = (unsigned)row * (unsigned)(K >> 1) + (unsigned)((lane & 3) << 4);
  }
  const int m0w = m0 + wave * 32;
  const unsigned rb_max = (unsigned)((M + 15) >> 4) - 1;
  unsigned rb0 = (unsigned)(m0w >> 4), rb1 = rb0 + 1;
  rb0 = rb0 < rb_max ? rb0 : rb_max;
  rb1 = rb1 < rb_max ? rb1 : rb_max;
  const unsigned av0 = (rb0 * (unsigned)(K >> 6)) * 1024u + ((unsigned)lane << 4);
  const unsigned av1 = (rb1 * (unsigned)(K >> 6)) * 1024u + ((unsigned)lane << 4);
  unsigned rsofs[2];
#pragma unroll
  for (int i = 0; i < 2; ++i) {
    int m = m0w + i * 16 + (lane & 15);
    m = args.has_rowsum ? (m < M ? m : M - 1) : 0;
    rsofs[i] = (unsigned)m * 4u;
  }
  const int8_t* a_ptr = args.a;
  const int8_t* w_ptr = reinterpret_cast<const int8_t*>(args.w);
  const float* alpha_p = args.alpha + n0;
  const float* bias_p = args.bias + n0;
  const int32_t* wzp_p = args.w_zp + n0;
  const int32_t* ct_p = args.col_term + n0;
  const int32_t* rs_p = args.a_rowsum;
  const float* so_ptr = args.out_scale;
  const float* oo_ptr = args.out_offset;
  const int qmin_bits = __builtin_amdgcn_readfirstlane(__float_as_int(args.out_qmin));
  const int qmax_bits = __builtin_amdgcn_readfirstlane(__float_as_int(args.out_qmax));
  float* outw = reinterpret_cast<float*>(args.out) + (size_t)m0w * N + n0;
  const float* resid = args.resid + (size_t)m0w * N + n0;
  const int mrem = __builtin_amdgcn_readfirstlane(M - m0w);
  const int flags = __builtin_amdgcn_readfirstlane((args.has_bias ? 1 : 0) | (args.has_rowsum ? 2 : 0));
  const int ldn = __builtin_amdgcn_readfirstlane(N), kt = __builtin_amdgcn_readfirstlane(KT);
  const unsigned tid = threadIdx.x;
  asm volatile(MQ_FRW4X_128R_ASM_BODY
               : [sw0] "+v"(sw[0]), [sw1] "+v"(sw[1])
               : [kt] "s"(kt), [wave] "s"(wave), [aptr] "s"(a_ptr), [wptr] "s"(w_ptr), [outw] "s"(outw), [resid] "s"(resid), [alpha] "s"(alpha_p),
                 [bias] "s"(bias_p), [wzp] "s"(wzp_p), [ct] "s"(ct_p), [rsptr] "s"(rs_p), [soptr] "s"(so_ptr), [ooptr] "s"(oo_ptr),
                 [qmin] "s"(qmin_bits), [qmax] "s"(qmax_bits), [ldn] "s"(ldn), [mrem] "s"(mrem), [flags] "s"(flags), [av0] "v"(av0), [av1] "v"(av1),
                 [tid] "v"(tid), [rsofs0] "v"(rsofs[0]), [rsofs1] "v"(rsofs[1])
               : MQ_FRW4X_128R_ASM_CLOBBERS);
}

static int launch_frw4r(GemmArgs a, hipStream_t st) {
  static PerDeviceOnce attr_set;
  const int dev = current_device();
  if (!attr_set.done(dev)) {
    hipError_t e = hipFuncSetAttribute((const void*)gemm_i8_frw4r_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, MQ_FRW4X_128R_LDS_BYTES);
    if (e != hipSuccess) {
      set_error("mq_gemm: hipFuncSetAttribute(%d B LDS): %s", MQ_FRW4X_128R_LDS_BYTES, hipGetErrorString(e));
      return MQ_EHIP;
    }
    attr_set.mark(dev);
  }
  a.has_rowsum = a.a_rowsum != nullptr;
  if (a.a_rowsum == nullptr) a.a_rowsum = a.col_term;
  if (a.bias == nullptr) a.bias = a.alpha;
  a.grid_m = (a.M + 127) / 128;
  a.grid_n = a.N / 128;
  a.group_m = g_group_m.load();
  gemm_i8_frw4r_kernel<<<a.grid_m * a.grid_n, 256, MQ_FRW4X_128R_LDS_BYTES, st>>>(a);
  MQ_LAUNCH_CHECK("mq_gemm");
  return MQ_OK;
}

static bool gemm_frw4_shape(int64_t M, int64_t N, int64_t K) { return M > 0 && (N % 176 == 0 || N % 128 == 0) && K % 256 == 0 && K >= 768; }

static std::atomic<int> g_w4_mode{1};           // mobilequant_amd_tuning.h: 1 = expand once per workgroup (frw4x), 0 = per-wave unpack (frw4)

template <int BNT, bool X>
static int launch_frw4(GemmArgs a, hipStream_t st) {
  constexpr int LDS = X ? (BNT == 176 ? MQ_FRW4X_LDS_BYTES : MQ_FRW4X_128_LDS_BYTES) : (BNT == 176 ? MQ_FRW4_LDS_BYTES : MQ_FRW4_128_LDS_BYTES);
  static PerDeviceOnce attr_set;
  const int dev = current_device();
  if (!attr_set.done(dev)) {
    hipError_t e = hipFuncSetAttribute((const void*)gemm_i8_frw4_kernel<BNT, X>, hipFuncAttributeMaxDynamicSharedMemorySize, LDS);
    if (e != hipSuccess) {
      set_error("mq_gemm: hipFuncSetAttribute(%d B LDS): %s", LDS, hipGetErrorString(e));
      return MQ_EHIP;
    }
    attr_set.mark(dev);
  }
  a.has_rowsum = a.a_rowsum != nullptr;
  if (a.a_rowsum == nullptr) a.a_rowsum = a.col_term;
  if (a.bias == nullptr) a.bias = a.alpha;
  a.grid_m = (a.M + 255) / 256;
  a.grid_n = a.N / BNT;
  gemm_i8_frw4_kernel<BNT, X><<<a.grid_m * a.grid_n, 512, LDS, st>>>(a);
  MQ_LAUNCH_CHECK("mq_gemm");
  return MQ_OK;
}

// ---- w3 of a gated FFN with the gate in its epilogue (tools/gen_fr_asm.py variant frg) ------------------------------------------------
// The free-running 256 x 176 program; its epilogue turns the tile's 8-bit output indices and w1's (gate_aidx, written by the launch
// before) into w2's int8 input image through the LDS-resident 64-KiB gated table: no index tensor of w3, no lookup launch.
template <int BNT>
__global__ void __launch_bounds__(512) gemm_i8_frg_kernel(const GemmArgs args) {
  const int lane = threadIdx.x & 63;
  const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  int tm, tn;
  tile_of_block(blockIdx.x, args.grid_m * args.grid_n, args.grid_m, args.grid_n, tm, tn);
  const int m0 = tm * 256, n0 = tn * BNT;
  const int M = args.M, N = args.N, K = args.K;
  const int KT = K / BK;
  unsigned sw[3] = {0, 0, 0};
#pragma unroll
  for (int i = 0; i < (BNT == 176 ? 3 : 2); ++i) {
    int row = n0 + (wave + i * 8) * 8 + (lane >> 3);
    row = row < N ? row : N - 1;
    sw[i] = (unsigned)row * (unsigned)K + (unsigned)(((lane & 7) ^ (lane >> 3)) << 4);
  }
  const int m0w = m0 + wave * 32;
  const unsigned rb_max = (unsigned)((M + 15) >> 4) - 1;
  unsigned rb0 = (unsigned)(m0w >> 4), rb1 = rb0 + 1;
  rb0 = rb0 < rb_max ? rb0 : rb_max;
  rb1 = rb1 < rb_max ? rb1 : rb_max;
  const unsigned av0 = (rb0 * (unsigned)(K >> 6)) * 1024u + ((unsigned)lane << 4);
  const unsigned av1 = (rb1 * (unsigned)(K >> 6)) * 1024u + ((unsigned)lane << 4);
  unsigned rsofs[2];
#pragma unroll
  for (int i = 0; i < 2; ++i) {
    int m = m0w + i * 16 + (lane & 15);
    m = args.has_rowsum ? (m < M ? m : M - 1) : 0;
    rsofs[i] = (unsigned)m * 4u;
  }
  const float* so_ptr = args.out_scale;
  const float* oo_ptr = args.out_offset;
  const int8_t* a_ptr = args.a;
  const int8_t* w_ptr = reinterpret_cast<const int8_t*>(args.w);
  const float* alpha_p = args.alpha + n0;
  const float* bias_p = args.bias + n0;
  const int32_t* wzp_p = args.w_zp + n0;
  const int32_t* ct_p = args.col_term + n0;
  const int32_t* rs_p = args.a_rowsum;
  const uint8_t* aidx = args.gate_aidx + (size_t)m0w * N + n0;
  const int8_t* table = args.gate_table;
  int8_t* qout = args.gate_q;
  int32_t* rsout = args.gate_rowsum + m0w;
  const int mrem = __builtin_amdgcn_readfirstlane(M - m0w);
  const int flags = __builtin_amdgcn_readfirstlane((args.has_bias ? 1 : 0) | (args.has_rowsum ? 2 : 0));
  const int ldn = __builtin_amdgcn_readfirstlane(N), kt = __builtin_amdgcn_readfirstlane(KT);
  const int cg0 = __builtin_amdgcn_readfirstlane(tn * (BNT / 16)), mb0 = __builtin_amdgcn_readfirstlane(m0w >> 4);
  const unsigned tid = threadIdx.x;
#define MQ_FRG_OPERANDS                                                                                                            \
  [kt] "s"(kt), [wave] "s"(wave), [aptr] "s"(a_ptr), [wptr] "s"(w_ptr), [aidx] "s"(aidx), [alpha] "s"(alpha_p),                    \
      [bias] "s"(bias_p), [wzp] "s"(wzp_p), [ct] "s"(ct_p), [rsptr] "s"(rs_p), [soptr] "s"(so_ptr), [ooptr] "s"(oo_ptr),       \
      [ldn] "s"(ldn), [mrem] "s"(mrem), [flags] "s"(flags), [table] "s"(table), [qout] "s"(qout), [rsout] "s"(rsout),              \
      [cg0] "s"(cg0), [mb0] "s"(mb0), [sw0] "v"(sw[0]), [sw1] "v"(sw[1]), [av0] "v"(av0), [av1] "v"(av1),                          \
      [tid] "v"(tid), [rsofs0] "v"(rsofs[0]), [rsofs1] "v"(rsofs[1])
  if constexpr (BNT == 176) {
    asm volatile(MQ_FRG_ASM_BODY : : MQ_FRG_OPERANDS, [sw2] "v"(sw[2]) : MQ_FRG_ASM_CLOBBERS);
  } else {
    asm volatile(MQ_FRG128_ASM_BODY : : MQ_FRG_OPERANDS : MQ_FRG128_ASM_CLOBBERS);
  }
#undef MQ_FRG_OPERANDS
}

template <int BNT>
static int launch_frg(const GemmArgs& a, hipStream_t st) {
  constexpr int LDS = BNT == 176 ? MQ_FRG_LDS_BYTES : MQ_FRG128_LDS_BYTES;
  static PerDeviceOnce attr_set;
  const int dev = current_device();
  if (!attr_set.done(dev)) {
    hipError_t e = hipFuncSetAttribute((const void*)gemm_i8_frg_kernel<BNT>, hipFuncAttributeMaxDynamicSharedMemorySize, LDS);
    if (e != hipSuccess) {
      set_error("mq_gemm: hipFuncSetAttribute(%d B LDS): %s", LDS, hipGetErrorString(e));
      return MQ_EHIP;
    }
    attr_set.mark(dev);
  }
  gemm_i8_frg_kernel<BNT><<<a.grid_m * a.grid_n, 512, LDS, st>>>(a);
  MQ_LAUNCH_CHECK("mq_gemm");
  return MQ_OK;
}

// w3 of a gated FFN from PACKED 4-bit weights with the gate in its epilogue (tools/gen_fr_asm.py variants frgw4x / frgw4x_128): frg's
// program with the W pieces (16 rows x 64 packed bytes) loaded into registers and expanded once per workgroup into the int8 ring.
template <int BNT>
__global__ void __launch_bounds__(512) gemm_i8_frgw4_kernel(const GemmArgs args) {
  constexpr int FNT = BNT / 16;
  const int lane = threadIdx.x & 63;
  const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  int tm, tn;
  tile_of_block(blockIdx.x, args.grid_m * args.grid_n, args.grid_m, args.grid_n, tm, tn);
  const int m0 = tm * 256, n0 = tn * BNT;
  const int M = args.M, N = args.N, K = args.K;
  const int KT = K / BK;
  unsigned sw[2] = {0, 0};
#pragma unroll
  for (int i = 0; i < (BNT == 176 ? 2 : 1); ++i) {
    int piece = wave + i * 8;
    piece = piece < FNT ? piece : FNT - 1;
    int row = n0 + piece * 16 + (lane >> 2);
    row = row < N ? row : N - 1;
    sw[i] = (unsigned)row * (unsigned)(K >> 1) + (unsigned)((lane & 3) << 4);
  }
  const int m0w = m0 + wave * 32;
  const unsigned rb_max = (unsigned)((M + 15) >> 4) - 1;
  unsigned rb0 = (unsigned)(m0w >> 4), rb1 = rb0 + 1;
  rb0 = rb0 < rb_max ? rb0 : rb_max;
  rb1 = rb1 < rb_max ? rb1 : rb_max;
  const unsigned av0 = (rb0 * (unsigned)(K >> 6)) * 1024u + ((unsigned)lane << 4);
  const unsigned av1 = (rb1 * (unsigned)(K >> 6)) * 1024u + ((unsigned)lane << 4);
  unsigned rsofs[2];
#pragma unroll
  for (int i = 0; i < 2; ++i) {
    int m = m0w + i * 16 + (lane & 15);
    m = args.has_rowsum ? (m < M ? m : M - 1) : 0;
    rsofs[i] = (unsigned)m * 4u;
  }
  const float* so_ptr = args.out_scale;
  const float* oo_ptr = args.out_offset;
  const int8_t* a_ptr = args.a;
  const int8_t* w_ptr = reinterpret_cast<const int8_t*>(args.w);
  const float* alpha_p = args.alpha + n0;
  const float* bias_p = args.bias + n0;
  const int32_t* wzp_p = args.w_zp + n0;
  const int32_t* ct_p = args.col_term + n0;
  const int32_t* rs_p = args.a_rowsum;
  const uint8_t* aidx = args.gate_aidx + (size_t)m0w * N + n0;
  const int8_t* table = args.gate_table;
  int8_t* qout = args.gate_q;
  int32_t* rsout = args.gate_rowsum + m0w;
  const int mrem = __builtin_amdgcn_readfirstlane(M - m0w);
  const int flags = __builtin_amdgcn_readfirstlane((args.has_bias ? 1 : 0) | (args.has_rowsum ? 2 : 0));
  const int ldn = __builtin_amdgcn_readfirstlane(N), kt = __builtin_amdgcn_readfirstlane(KT);
  const int cg0 = __builtin_amdgcn_readfirstlane(tn * (BNT / 16)), mb0 = __builtin_amdgcn_readfirstlane(m0w >> 4);
  const unsigned tid = threadIdx.x;
#define MQ_FRGW4_OPERANDS                                                                                                          \
  [kt] "s"(kt), [wave] "s"(wave), [aptr] "s"(a_ptr), [wptr] "s"(w_ptr), [aidx] "s"(aidx), [alpha] "s"(alpha_p),                    \
      [bias] "s"(bias_p), [wzp] "s"(wzp_p), [ct] "s"(ct_p), [rsptr] "s"(rs_p), [soptr] "s"(so_ptr), [ooptr] "s"(oo_ptr),       \
      [ldn] "s"(ldn), [mrem] "s"(mrem), [flags] "s"(flags), [table] "s"(table), [qout] "s"(qout), [rsout] "s"(rsout),              \
      [cg0] "s"(cg0), [mb0] "s"(mb0), [av0] "v"(av0), [av1] "v"(av1), [tid] "v"(tid), [rsofs0] "v"(rsofs[0]), [rsofs1] "v"(rsofs[1])
  if constexpr (BNT == 176) {
    asm volatile(MQ_FRGW4X_ASM_BODY : [sw0] "+v"(sw[0]), [sw1] "+v"(sw[1]) : MQ_FRGW4_OPERANDS : MQ_FRGW4X_ASM_CLOBBERS);
  } else {
    asm volatile(MQ_FRGW4X_128_ASM_BODY : [sw0] "+v"(sw[0]) : MQ_FRGW4_OPERANDS : MQ_FRGW4X_128_ASM_CLOBBERS);
  }
#undef MQ_FRGW4_OPERANDS
}

template <int BNT>
static int launch_frgw4(const GemmArgs& a, hipStream_t st) {
  constexpr int LDS = BNT == 176 ? MQ_FRGW4X_LDS_BYTES : MQ_FRGW4X_128_LDS_BYTES;
  static PerDeviceOnce attr_set;
  const int dev = current_device();
  if (!attr_set.done(dev)) {
    hipError_t e = hipFuncSetAttribute((const void*)gemm_i8_frgw4_kernel<BNT>, hipFuncAttributeMaxDynamicSharedMemorySize, LDS);
    if (e != hipSuccess) {
      set_error("mq_gemm: hipFuncSetAttribute(%d B LDS): %s", LDS, hipGetErrorString(e));
      return MQ_EHIP;
    }
    attr_set.mark(dev);
  }
  gemm_i8_frgw4_kernel<BNT><<<a.grid_m * a.grid_n, 512, LDS, st>>>(a);
  MQ_LAUNCH_CHECK("mq_gemm");
  return MQ_OK;
}

// Two QLinears that consume the SAME activation (w1 / w3 of an FFN: hf_model.py:1057) in ONE launch: workgroups
// [0, nblk) compute problem 0, [nblk, 2 nblk) problem 1 with the same block -> tile map, so the second problem's tiles land on
// the CUs / XCD that just streamed the same activation panel.  Twice the tiles per launch: the kernel-boundary cost and the
// prologue burst are paid once, and a CU's second workgroup starts while other CUs are still in their first epilogue.
struct GemmPairArgs {
  GemmArgs p[2];
};
__global__ void __launch_bounds__(512) gemm_i8_fr_pair_kernel(const GemmPairArgs args) {
  const int nblk = args.p[0].grid_m * args.p[0].grid_n;      // (= gridDim.x / 2, without the hidden-argument load)
  const int which = __builtin_amdgcn_readfirstlane(blockIdx.x >= (unsigned)nblk ? 1 : 0);
  if (which) gemm_i8_fr_body(args.p[1], (int)blockIdx.x - nblk, nblk);
  else gemm_i8_fr_body(args.p[0], (int)blockIdx.x, nblk);
}

// The same two problems on HALF the workgroups: a workgroup runs its tile of problem 0, then the same tile of problem 1 (same rows
// of the activation image, L2-warm).  What it saves is the dispatcher's hand-over of the CU from one 512-thread / 138-KiB workgroup to
// the next; the two programs still run back to back (mq_gemm_set_pair_mode, A/B in bench.py's ffn_pair_gemm).
__global__ void __launch_bounds__(512) gemm_i8_fr_pair_persistent_kernel(const GemmPairArgs args) {
  const int nblk = args.p[0].grid_m * args.p[0].grid_n;
#pragma unroll 1
  for (int t = 0; t < 2; ++t) {
    gemm_i8_fr_body(args.p[t], (int)blockIdx.x, nblk);
    __syncthreads();                                     // the next program's first LDS-DMA pieces land in the ring the slow waves still read
  }
}
static std::atomic<int> g_pair_mode{0};

static bool gemm_fr_supported(const GemmArgs& a) {
  return a.a_tiled && (a.out_dtype == MQ_U8 || a.out_dtype == MQ_I8) && a.out_scale != nullptr && a.out_qmin == 0.0f &&
         a.out_qmax == 255.0f && a.K % 256 == 0 && a.K >= 768 && a.N % 176 == 0;
}

static int launch_fr_pair(const GemmArgs& a0, const GemmArgs& a1, hipStream_t st) {
  static PerDeviceOnce attr_set;
  const int dev = current_device();
  if (!attr_set.done(dev)) {
    hipError_t e = hipFuncSetAttribute((const void*)gemm_i8_fr_pair_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, MQ_FR_LDS_BYTES);
    if (e != hipSuccess) {
      set_error("mq_gemm: hipFuncSetAttribute(%d B LDS): %s", MQ_FR_LDS_BYTES, hipGetErrorString(e));
      return MQ_EHIP;
    }
    attr_set.mark(dev);
  }
  GemmPairArgs pa;
  pa.p[0] = a0;
  pa.p[1] = a1;
  if (g_pair_mode.load() == 1) {
    static PerDeviceOnce attr_set_p;
    if (!attr_set_p.done(dev)) {
      hipError_t e = hipFuncSetAttribute((const void*)gemm_i8_fr_pair_persistent_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, MQ_FR_LDS_BYTES);
      if (e != hipSuccess) {
        set_error("mq_gemm: hipFuncSetAttribute(%d B LDS): %s", MQ_FR_LDS_BYTES, hipGetErrorString(e));
        return MQ_EHIP;
      }
      attr_set_p.mark(dev);
    }
    gemm_i8_fr_pair_persistent_kernel<<<a0.grid_m * a0.grid_n, 512, MQ_FR_LDS_BYTES, st>>>(pa);
  } else {
    gemm_i8_fr_pair_kernel<<<2 * a0.grid_m * a0.grid_n, 512, MQ_FR_LDS_BYTES, st>>>(pa);
  }
  MQ_LAUNCH_CHECK("mq_gemm");
  return MQ_OK;
}

static int launch_fr(const GemmArgs& a, hipStream_t st) {
  static PerDeviceOnce attr_set;
  const int dev = current_device();
  if (!attr_set.done(dev)) {
    hipError_t e = hipFuncSetAttribute((const void*)gemm_i8_fr_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, MQ_FR_LDS_BYTES);
    if (e != hipSuccess) {
      set_error("mq_gemm: hipFuncSetAttribute(%d B LDS): %s", MQ_FR_LDS_BYTES, hipGetErrorString(e));
      return MQ_EHIP;
    }
    attr_set.mark(dev);
  }
  gemm_i8_fr_kernel<<<a.grid_m * a.grid_n, 512, MQ_FR_LDS_BYTES, st>>>(a);
  MQ_LAUNCH_CHECK("mq_gemm");
  return MQ_OK;
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
    {"t256x176_w8x1_pp", 256, 176, 512},
    {"t64x32_w2x2", 64, 32, 256},
    {"t256x176_w8x1_pp_asm", 256, 176, 512},
    {"t128x128_w2x4", 128, 128, 512},
    {"t256x176_w8x1_fr_asm", 256, 176, 512},
};
constexpr int kNumVariants = sizeof(kVariants) / sizeof(kVariants[0]);
static std::atomic<int> g_forced_variant{-1};   // mobilequant_amd_tuning.h: process-wide, atomic
static std::atomic<int> g_debug{0};
static unsigned long long* g_dbg_ts = nullptr;

template <int BM, int BN, int WM, int WN, int OUT, bool OQ, bool W4, int ABL, int PP>
static int launch_one(const GemmArgs& a, int lds, hipStream_t st) {
  auto kfn = gemm_i8_kernel<BM, BN, WM, WN, OUT, OQ, W4, ABL, PP>;
  static PerDeviceOnce attr_set;   // per instantiation and per device
  const int dev = current_device();
  if (!attr_set.done(dev)) {
    hipError_t e = hipFuncSetAttribute((const void*)kfn, hipFuncAttributeMaxDynamicSharedMemorySize, lds);
    if (e != hipSuccess) {
      set_error("mq_gemm: hipFuncSetAttribute(%d B LDS): %s", lds, hipGetErrorString(e));
      return MQ_EHIP;
    }
    attr_set.mark(dev);
  }
  kfn<<<a.grid_m * a.grid_n, 64 * WM * WN, lds, st>>>(a);
  MQ_LAUNCH_CHECK("mq_gemm");
  return MQ_OK;
}

template <int BM, int BN, int WM, int WN, int OUT, bool OQ, bool W4, int PP>
static int launch_typed(const GemmArgs& a, hipStream_t st) {
  constexpr int LDS = lds_main_bytes(BM, BN, WM, WN, W4, PP != 0) + 16 * BN;
#ifdef MQ_GEMM_ABLATE
  if constexpr (OUT == MQ_U8 && OQ && !W4 && BM == 256) {
    switch (g_debug.load()) {
      case 1: return launch_one<BM, BN, WM, WN, OUT, OQ, W4, 1, PP>(a, LDS, st);
      case 2: return launch_one<BM, BN, WM, WN, OUT, OQ, W4, 2, PP>(a, LDS, st);
      case 3: return launch_one<BM, BN, WM, WN, OUT, OQ, W4, 3, PP>(a, LDS, st);
      case 4: return launch_one<BM, BN, WM, WN, OUT, OQ, W4, 4, PP>(a, LDS, st);
      case 5: return launch_one<BM, BN, WM, WN, OUT, OQ, W4, 5, PP>(a, LDS, st);
      case 6: return launch_one<BM, BN, WM, WN, OUT, OQ, W4, 6, PP>(a, LDS, st);
      case 7: return launch_one<BM, BN, WM, WN, OUT, OQ, W4, 7, PP>(a, LDS, st);
      case 8: if constexpr (PP) return launch_one<BM, BN, WM, WN, OUT, OQ, W4, 8, PP>(a, LDS, st); else break;
      case 9: if constexpr (PP) return launch_one<BM, BN, WM, WN, OUT, OQ, W4, 9, PP>(a, LDS, st); else break;
      case 16: if constexpr (PP) return launch_one<BM, BN, WM, WN, OUT, OQ, W4, 16, PP>(a, LDS, st); else break;
      case 17: if constexpr (PP) return launch_one<BM, BN, WM, WN, OUT, OQ, W4, 17, PP>(a, LDS, st); else break;
      case 18: if constexpr (PP) return launch_one<BM, BN, WM, WN, OUT, OQ, W4, 18, PP>(a, LDS, st); else break;
      case 24: if constexpr (PP) return launch_one<BM, BN, WM, WN, OUT, OQ, W4, 24, PP>(a, LDS, st); else break;
      case 25: if constexpr (PP) return launch_one<BM, BN, WM, WN, OUT, OQ, W4, 25, PP>(a, LDS, st); else break;
      case 48: if constexpr (PP) return launch_one<BM, BN, WM, WN, OUT, OQ, W4, 48, PP>(a, LDS, st); else break;
      default: break;
    }
  }
#endif
  return launch_one<BM, BN, WM, WN, OUT, OQ, W4, 0, PP>(a, LDS, st);
}

template <int BM, int BN, int WM, int WN, bool W4, int PP = 0>
static int launch_cfg(const GemmArgs& a, bool outq, hipStream_t st) {
  if (outq) {
    switch (a.out_dtype) {
      case MQ_F32: return launch_typed<BM, BN, WM, WN, MQ_F32, true, W4, PP>(a, st);
      case MQ_F16: return launch_typed<BM, BN, WM, WN, MQ_F16, true, W4, PP>(a, st);
      case MQ_U8: return launch_typed<BM, BN, WM, WN, MQ_U8, true, W4, PP>(a, st);
      case MQ_I8: return launch_typed<BM, BN, WM, WN, MQ_I8, true, W4, PP>(a, st);
      case MQ_U16: return launch_typed<BM, BN, WM, WN, MQ_U16, true, W4, PP>(a, st);
      case MQ_I16: return launch_typed<BM, BN, WM, WN, MQ_I16, true, W4, PP>(a, st);
      default: set_error("mq_gemm: out_dtype %d not supported", a.out_dtype); return MQ_EUNSUPPORTED;
    }
  }
  switch (a.out_dtype) {
    case MQ_F32: return launch_typed<BM, BN, WM, WN, MQ_F32, false, W4, PP>(a, st);
    case MQ_F16: return launch_typed<BM, BN, WM, WN, MQ_F16, false, W4, PP>(a, st);
    default:
      set_error("mq_gemm: integer out_dtype %d needs an output quantizer", a.out_dtype);
      return MQ_EINVAL;
  }
}

// The generated-ISA loop serves the shapes the 256x176 ping-pong tile serves (N a multiple of 176, at least 192 tiles)
// with whole pairs of K = 128 stages.
static bool gemm_tiled_supported(int64_t M, int64_t N, int64_t K) {
  return M > 0 && N % 176 == 0 && K % 256 == 0 && ((M + 255) / 256) * (N / 176) >= 192;
}

static int pick_variant(int M, int N, bool w4) {
  if (g_forced_variant.load() >= 0) return g_forced_variant.load();
  auto blocks = [&](int v) {
    return (long)((M + kVariants[v].bm - 1) / kVariants[v].bm) * ((N + kVariants[v].bn - 1) / kVariants[v].bn);
  };
  // Measured on MI355X (tools/bench_shapes.py --sweep, profiles/r01/linear_shapes_all_variants.txt):
  //  * the 8-wave 256x176 ping-pong tile is the fastest whenever it tiles N exactly and fills the chip in one
  //    round (TinyLlama / StableLM FFN: N = 5632 = 32 x 176); int8 weights only;
  //  * otherwise the largest tile that still gives (nearly) every CU a workgroup: 256x256 (Gemma FFN), 256x128
  //    (fused q|k|v, N = 2560: wins from 160 workgroups);
  //  * N = 2048 outputs (q/o, w2) and everything mid-sized: 128x128 with EIGHT waves (2 x 4, wave tile 64 x 32): twice
  //    the waves of the 2 x 2 layout hide the LDS / DMA latency of the simple two-stage loop (4 x 2, 4 x 4 and 16-wave 256x256 / 64x64 / 256x128 layouts measured no better) -- 11.0 vs 15.2 us
  //    (q/o), 30.7 vs 39.1 us (w2), 75 vs 95 us (Gemma w2);
  //  * packed 4-bit weights: 256x256 from 224 workgroups (Gemma FFN), else the same 8-wave 128x128 tile (also for
  //    N = 5632: 26.3 vs 28.1 us), small problems 64x64 / 64x32.
  if (w4) {
    if (blocks(2) >= 224) return 2;
    if (blocks(10) >= 96) return 10;
    return blocks(6) >= 256 ? 6 : 8;
  }
  if (N % 176 == 0 && blocks(1) >= 192) return 7;
  if (blocks(2) >= 224) return 2;
  if (blocks(5) >= 160) return 5;
  if (blocks(10) >= 96) return 10;
  return blocks(6) >= 256 ? 6 : 8;
}

template <bool W4>
static int run_gemm(GemmArgs a, hipStream_t st) {
  if (a.a_tiled && !W4 && !gemm_tiled_supported(a.M, a.N, a.K) && gemm_fr128_shape(a.M, a.N, a.K) && a.resid == nullptr &&
      (a.out_dtype == MQ_U8 || a.out_dtype == MQ_I8) && a.out_scale != nullptr && a.out_qmin == 0.0f && a.out_qmax == 255.0f) {
    return launch_fr128<FR128>(a, st);      // 8-bit unsigned index outputs on 256 x 128 tiles (Gemma's w1 / w3: N = 16384)
  }
  const bool outq = a.out_scale != nullptr;
  a.has_rowsum = a.a_rowsum != nullptr;
  if (a.a_rowsum == nullptr) a.a_rowsum = a.col_term;                 // element 0 only (M may exceed N); multiplied by w_zp == 0
  if (a.bias == nullptr) a.bias = a.alpha;                            // masked by has_bias
  int v = pick_variant(a.M, a.N, W4);
  if (a.a_tiled) {
    if (W4 || !gemm_tiled_supported(a.M, a.N, a.K)) {
      set_error("mq_w8a8_linear_tiled: shape %dx%dx%d is not served by the fragment-blocked path (see mq_gemm_tiled_supported)", a.M,
                a.N, a.K);
      return MQ_EUNSUPPORTED;
    }
    // the free-running generated kernel serves the 8-bit unsigned output grid; every other output type keeps the
    // ping-pong generated loop with the C++ epilogue (mq_gemm_set_variant(9) forces that one for A/B timing)
    v = (gemm_fr_supported(a) && g_forced_variant != 9) ? 11 : 9;
  } else if (v == 9 || v == 11) {
    v = 7;                                  // variants 9 and 11 read fragment-blocked activations only
  }
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
    case 7:
      if constexpr (!W4) return launch_cfg<256, 176, 8, 1, false, 1>(a, outq, st);
      else return launch_cfg<256, 176, 8, 1, W4>(a, outq, st);
    case 9:     // generated-ISA main loop on fragment-blocked activations (run_gemm checked the shape)
      if constexpr (!W4) return launch_cfg<256, 176, 8, 1, false, 3>(a, outq, st);
      else return launch_cfg<256, 176, 8, 1, W4>(a, outq, st);
    case 11: return launch_fr(a, st);
    case 8: return launch_cfg<64, 32, 2, 2, W4>(a, outq, st);
    case 10: return launch_cfg<128, 128, 2, 4, W4>(a, outq, st);
    default: set_error("mq_gemm: bad variant %d", v); return MQ_EINVAL;
  }
}

static int check_common(const char* fn, const void* a, const void* w, int64_t M, int64_t N, int64_t K,
                        const int32_t* a_rowsum, const float* alpha, const int32_t* w_zp, const int32_t* col_term,
                        const float* bias, const float* out_scale, const float* out_offset, void* out, int kdiv) {
  MQ_REQUIRE(a && w && alpha && w_zp && col_term && out, "%s: null pointer", fn);
  MQ_REQUIRE(M > 0 && N > 0 && K > 0, "%s: bad shape M=%lld N=%lld K=%lld", fn, (long long)M, (long long)N, (long long)K);
  MQ_REQUIRE(K % 128 == 0 && K <= 65536, "%s: K=%lld must be a multiple of 128, at most 65536", fn, (long long)K);
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

#ifdef MQ_GEMM_ABLATE
// tools/hole_probe.py: a do-nothing kernel with a chosen footprint (threads, LDS bytes, register count), duration (busy-wait on the
// 100 MHz s_memrealtime) and amount of dirty L2 data; block b stamps [entry, exit] realtime into out[2 b .. 2 b + 1].  What does a kernel
// boundary cost behind a kernel of the GEMM's footprint?
namespace mq {
template <bool FAT>
__global__ void __launch_bounds__(FAT ? 512 : 1024) debug_stamp_kernel(unsigned long long* out, int spin_ticks, int* dirty, int dirty_words) {
  const unsigned long long t0 = __builtin_amdgcn_s_memrealtime();
  if constexpr (FAT) asm volatile("v_mov_b32 v127, 0\n\tv_accvgpr_write_b32 a119, 0" ::: "v127", "a119");
  extern __shared__ int lds_probe[];
  if (spin_ticks < 0) lds_probe[threadIdx.x] = (int)t0;      // (never: keeps the dynamic LDS referenced)
  for (int i = (int)threadIdx.x; i < dirty_words; i += (int)blockDim.x) dirty[(size_t)blockIdx.x * dirty_words + i] = i;
  while ((long long)(__builtin_amdgcn_s_memrealtime() - t0) < (long long)spin_ticks) __builtin_amdgcn_s_sleep(4);
  if (threadIdx.x == 0) {
    out[2 * blockIdx.x] = t0;
    out[2 * blockIdx.x + 1] = __builtin_amdgcn_s_memrealtime();
  }
}
}  // namespace mq
#endif

extern "C" {

int mq_gemm_set_variant(int variant) {
  g_forced_variant = (variant >= 0 && variant < kNumVariants) ? variant : -1;
  return kNumVariants;
}

int mq_gemm_set_pair_mode(int mode) {
  g_pair_mode = mode;
  return 0;
}

int mq_gemm_set_group_m(int group_m) {
  g_group_m = group_m;
  return 0;
}

int mq_gemm_set_w4_mode(int mode) {
#ifdef MQ_BUILD_EXPERIMENTS
  g_w4_mode = mode;
  return 0;
#else
  g_w4_mode = 1;                    // the per-wave unpack kernels (frw4 / frw4_128) exist in experiment builds only
  return mode != 0 ? 0 : 1;         // 1 = asked for a variant this library was not built with
#endif
}

int mq_gemm_set_clock_probe(void* buf) {
  g_dbg_ts = reinterpret_cast<unsigned long long*>(buf);
  return 0;
}

int mq_gemm_set_debug(int flags) {
  g_debug = flags;
  return 0;
}

#ifdef MQ_GEMM_ABLATE
// not part of the public header: probe-only hook for the s_memtime stamps of ablation builds
extern "C" int mq_gemm_set_debug_buffer_(void* p) {
  g_dbg_ts = reinterpret_cast<unsigned long long*>(p);
  return 0;
}

extern "C" int mq_debug_stamp_(void* out, int blocks, int threads, int lds_bytes, int fat, int spin_ticks, void* dirty, int dirty_words,
                               void* stream) {
  hipStream_t st = reinterpret_cast<hipStream_t>(stream);
  auto o = reinterpret_cast<unsigned long long*>(out);
  auto d = reinterpret_cast<int*>(dirty);
  static bool attr = false;
  if (!attr) {
    (void)hipFuncSetAttribute((const void*)mq::debug_stamp_kernel<false>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
    (void)hipFuncSetAttribute((const void*)mq::debug_stamp_kernel<true>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
    attr = true;
  }
  if (fat) mq::debug_stamp_kernel<true><<<blocks, threads, lds_bytes, st>>>(o, spin_ticks, d, dirty_words);
  else mq::debug_stamp_kernel<false><<<blocks, threads, lds_bytes, st>>>(o, spin_ticks, d, dirty_words);
  return hipGetLastError() == hipSuccess ? 0 : 1;
}
#endif

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
  if (M <= 8 && M * K <= 64 * 1024 - 64 && g_forced_variant < 0) {   // decode shapes: weight-streaming GEMV (mq_gemv.hip)
    if (out_scale == nullptr && out_dtype != MQ_F32 && out_dtype != MQ_F16) {
      set_error("mq_w8a8_linear: integer out_dtype %d needs an output quantizer", out_dtype);
      return MQ_EINVAL;
    }
    GemvArgs v{a, w, (int)M, (int)N, (int)K, a_rowsum, alpha, w_zp, col_term, bias, out_scale, out_offset,
               out_qmin, out_qmax, out, out_dtype, nullptr, nullptr, nullptr, 0.f, 0.f, 0, 0};
    return run_gemv(v, as_stream(stream));
  }
  GemmArgs g{a, w, (int)M, (int)N, (int)K, a_rowsum, alpha, w_zp, col_term, bias, out_scale, out_offset,
             out_qmin, out_qmax, out, out_dtype, 0, 0, bias != nullptr, 0, 0, g_dbg_ts};
  return run_gemm<false>(g, as_stream(stream));
}

int mq_w8a8_linear_residual(const int8_t* a, const int8_t* w, int64_t M, int64_t N, int64_t K, const int32_t* a_rowsum,
                            const float* alpha, const int32_t* w_zp, const int32_t* col_term, const float* bias,
                            const float* out_scale, const float* out_offset, float out_qmin, float out_qmax,
                            const float* resid, float* out, mq_stream_t stream) {
  int rc = check_common("mq_w8a8_linear_residual", a, w, M, N, K, a_rowsum, alpha, w_zp, col_term, bias, out_scale, out_offset, out, 1);
  if (rc != MQ_OK) return rc;
  MQ_REQUIRE(resid != nullptr && aligned(resid, 16) && M > 8, "mq_w8a8_linear_residual: resid must be non-null and 16-byte aligned; M > 8");
  GemmArgs g{a, w, (int)M, (int)N, (int)K, a_rowsum, alpha, w_zp, col_term, bias, out_scale, out_offset,
             out_qmin, out_qmax, out, MQ_F32, 0, 0, bias != nullptr, 0, 0, g_dbg_ts, resid};
  return run_gemm<false>(g, as_stream(stream));
}

static int linear_segmented(const char* fn, bool w4, const int8_t* a, const void* w, int64_t M, int64_t N, int64_t K, const int32_t* a_rowsum,
                            const float* alpha, const int32_t* w_zp, const int32_t* col_term, const float* bias, int n_segments,
                            const int64_t* seg_end, const mq_grid* grids, uint8_t* out, mq_stream_t stream) {
  MQ_REQUIRE(n_segments >= 1 && n_segments <= 3 && seg_end != nullptr && grids != nullptr, "%s: 1..3 segments", fn);
  int rc = check_common(fn, a, w, M, N, K, a_rowsum, alpha, w_zp, col_term, bias, grids[0].scale, grids[0].offset, out, w4 ? 2 : 1);
  if (rc != MQ_OK) return rc;
  MQ_REQUIRE(M > 8, "%s: M > 8 (decode shapes: mq_decode_gemv)", fn);
  MQ_REQUIRE(!w4 || K % 64 == 0, "%s: packed 4-bit weights need K %% 64 == 0", fn);
  int64_t prev = 0;
  for (int i = 0; i < n_segments; ++i) {
    MQ_REQUIRE(grids[i].scale && grids[i].offset && grids[i].qmin == 0.f && grids[i].qmax == 255.f,
               "%s: segment %d needs an 8-bit unsigned output grid", fn, i);
    MQ_REQUIRE(seg_end[i] > prev && seg_end[i] <= N && seg_end[i] % 4 == 0, "%s: segment ends must increase, be multiples of 4, <= N", fn);
    prev = seg_end[i];
  }
  MQ_REQUIRE(prev == N, "%s: the last segment must end at N", fn);
  GemmArgs g{a, w, (int)M, (int)N, (int)K, a_rowsum, alpha, w_zp, col_term, bias, grids[0].scale, grids[0].offset,
             0.f, 255.f, out, MQ_U8, 0, 0, bias != nullptr, 0, 0, g_dbg_ts, nullptr, {0, 0}, {nullptr, nullptr}, {nullptr, nullptr}};
  for (int i = 1; i < n_segments; ++i) {
    g.seg_end[i - 1] = (int)seg_end[i - 1];
    g.seg_scale[i - 1] = grids[i].scale;
    g.seg_offset[i - 1] = grids[i].offset;
  }
  return w4 ? run_gemm<true>(g, as_stream(stream)) : run_gemm<false>(g, as_stream(stream));
}

int mq_w8a8_linear_segmented(const int8_t* a, const int8_t* w, int64_t M, int64_t N, int64_t K, const int32_t* a_rowsum,
                             const float* alpha, const int32_t* w_zp, const int32_t* col_term, const float* bias, int n_segments,
                             const int64_t* seg_end, const mq_grid* grids, uint8_t* out, mq_stream_t stream) {
  return linear_segmented("mq_w8a8_linear_segmented", false, a, w, M, N, K, a_rowsum, alpha, w_zp, col_term, bias, n_segments, seg_end, grids,
                          out, stream);
}

int mq_w4a8_linear_segmented(const int8_t* a, const uint8_t* w_packed, int64_t M, int64_t N, int64_t K, const int32_t* a_rowsum,
                             const float* alpha, const int32_t* w_zp, const int32_t* col_term, const float* bias, int n_segments,
                             const int64_t* seg_end, const mq_grid* grids, uint8_t* out, mq_stream_t stream) {
  return linear_segmented("mq_w4a8_linear_segmented", true, a, w_packed, M, N, K, a_rowsum, alpha, w_zp, col_term, bias, n_segments, seg_end,
                          grids, out, stream);
}

int mq_gemm_tiled_supported(int64_t M, int64_t N, int64_t K) { return gemm_tiled_supported(M, N, K) ? 1 : 0; }

int mq_w8a8_linear_tiled(const int8_t* a_tiled, const int8_t* w, int64_t M, int64_t N, int64_t K, const int32_t* a_rowsum,
                         const float* alpha, const int32_t* w_zp, const int32_t* col_term, const float* bias,
                         const float* out_scale, const float* out_offset, float out_qmin, float out_qmax, void* out,
                         int out_dtype, mq_stream_t stream) {
  int rc = check_common("mq_w8a8_linear_tiled", a_tiled, w, M, N, K, a_rowsum, alpha, w_zp, col_term, bias, out_scale,
                        out_offset, out, 1);
  if (rc != MQ_OK) return rc;
  MQ_REQUIRE(((M + 15) / 16) * 16 * K < (1ll << 32), "mq_w8a8_linear_tiled: activation too large");
  GemmArgs g{a_tiled, w, (int)M, (int)N, (int)K, a_rowsum, alpha, w_zp, col_term, bias, out_scale, out_offset,
             out_qmin, out_qmax, out, out_dtype, 0, 0, bias != nullptr, 1, 0, g_dbg_ts};
  return run_gemm<false>(g, as_stream(stream));
}

int mq_gemm_tiled_w4_supported(int64_t M, int64_t N, int64_t K) { return gemm_frw4_shape(M, N, K) ? 1 : 0; }

int mq_w4a8_linear_tiled(const int8_t* a_tiled, const uint8_t* w_packed, int64_t M, int64_t N, int64_t K, const int32_t* a_rowsum,
                         const float* alpha, const int32_t* w_zp, const int32_t* col_term, const float* bias, int n_segments,
                         const int64_t* seg_end, const mq_grid* grids, void* out, int out_dtype, mq_stream_t stream) {
  const char* fn = "mq_w4a8_linear_tiled";
  MQ_REQUIRE(n_segments >= 1 && n_segments <= 3 && grids != nullptr && (n_segments == 1 || seg_end != nullptr), "%s: 1..3 segments", fn);
  int rc = check_common(fn, a_tiled, w_packed, M, N, K, a_rowsum, alpha, w_zp, col_term, bias, grids[0].scale, grids[0].offset, out, 2);
  if (rc != MQ_OK) return rc;
  MQ_REQUIRE(((M + 15) / 16) * 16 * K < (1ll << 32), "%s: activation too large", fn);
  MQ_REQUIRE(out_dtype == MQ_U8 || out_dtype == MQ_I8, "%s: u8 (indices) or i8 (index - 128) output", fn);
  int64_t prev = 0;
  for (int i = 0; i < n_segments; ++i) {
    MQ_REQUIRE(grids[i].scale && grids[i].offset && grids[i].qmin == 0.f && grids[i].qmax == 255.f,
               "%s: segment %d needs an 8-bit unsigned output grid", fn, i);
    if (n_segments > 1) {
      MQ_REQUIRE(seg_end[i] > prev && seg_end[i] <= N && seg_end[i] % 4 == 0, "%s: segment ends must increase, be multiples of 4, <= N", fn);
      prev = seg_end[i];
    }
  }
  MQ_REQUIRE(n_segments == 1 || prev == N, "%s: the last segment must end at N", fn);
  if (!gemm_frw4_shape(M, N, K) || (n_segments > 1 && N % 128 != 0)) {
    set_error("%s: shape %lldx%lldx%lld is not served (mq_gemm_tiled_w4_supported: N %% 176 == 0 or N %% 128 == 0 (segments: 128), K %% 256 == 0, "
              "K >= 768)", fn, (long long)M, (long long)N, (long long)K);
    return MQ_EUNSUPPORTED;
  }
  GemmArgs g{a_tiled, w_packed, (int)M, (int)N, (int)K, a_rowsum, alpha, w_zp, col_term, bias, grids[0].scale, grids[0].offset,
             0.f, 255.f, out, out_dtype, 0, 0, bias != nullptr, 1, 0, g_dbg_ts, nullptr, {0, 0}, {nullptr, nullptr}, {nullptr, nullptr}};
  for (int i = 1; i < n_segments; ++i) {
    g.seg_end[i - 1] = (int)seg_end[i - 1];
    g.seg_scale[i - 1] = grids[i].scale;
    g.seg_offset[i - 1] = grids[i].offset;
  }
#ifdef MQ_BUILD_EXPERIMENTS
  const bool x = g_w4_mode.load() != 0;
  if (n_segments == 1 && N % 176 == 0) return x ? launch_frw4<176, true>(g, as_stream(stream)) : launch_frw4<176, false>(g, as_stream(stream));
  return x ? launch_frw4<128, true>(g, as_stream(stream)) : launch_frw4<128, false>(g, as_stream(stream));
#else
  if (n_segments == 1 && N % 176 == 0) return launch_frw4<176, true>(g, as_stream(stream));
  return launch_frw4<128, true>(g, as_stream(stream));
#endif
}

int mq_w8a8_linear_tiled_pair(const int8_t* a_tiled, int64_t M, int64_t N, int64_t K, const int32_t* a_rowsum,
                              const int8_t* w0, const float* alpha0, const int32_t* w_zp0, const int32_t* col_term0,
                              const float* bias0, const float* out_scale0, const float* out_offset0, void* out0,
                              const int8_t* w1, const float* alpha1, const int32_t* w_zp1, const int32_t* col_term1,
                              const float* bias1, const float* out_scale1, const float* out_offset1, void* out1,
                              int out_dtype, mq_stream_t stream) {
  const char* fn = "mq_w8a8_linear_tiled_pair";
  int rc = check_common(fn, a_tiled, w0, M, N, K, a_rowsum, alpha0, w_zp0, col_term0, bias0, out_scale0, out_offset0, out0, 1);
  if (rc != MQ_OK) return rc;
  rc = check_common(fn, a_tiled, w1, M, N, K, a_rowsum, alpha1, w_zp1, col_term1, bias1, out_scale1, out_offset1, out1, 1);
  if (rc != MQ_OK) return rc;
  MQ_REQUIRE(((M + 15) / 16) * 16 * K < (1ll << 32), "%s: activation too large", fn);
  MQ_REQUIRE(out_scale0 && out_scale1 && (out_dtype == MQ_U8 || out_dtype == MQ_I8),
             "%s: both outputs carry an 8-bit unsigned output grid (u8 / i8 storage)", fn);
  GemmArgs g0{a_tiled, w0, (int)M, (int)N, (int)K, a_rowsum, alpha0, w_zp0, col_term0, bias0, out_scale0, out_offset0,
              0.f, 255.f, out0, out_dtype, 0, 0, bias0 != nullptr, 1, 0, g_dbg_ts};
  GemmArgs g1{a_tiled, w1, (int)M, (int)N, (int)K, a_rowsum, alpha1, w_zp1, col_term1, bias1, out_scale1, out_offset1,
              0.f, 255.f, out1, out_dtype, 0, 0, bias1 != nullptr, 1, 0, g_dbg_ts};
  if (!gemm_tiled_supported(M, N, K) || !gemm_fr_supported(g0)) {
    set_error("%s: shape %lldx%lldx%lld is not served (mq_gemm_tiled_supported, K %% 256 == 0, K >= 768)", fn, (long long)M, (long long)N,
              (long long)K);
    return MQ_EUNSUPPORTED;
  }
  for (GemmArgs* g : {&g0, &g1}) {
    g->has_rowsum = g->a_rowsum != nullptr;
    if (g->a_rowsum == nullptr) g->a_rowsum = g->col_term;
    if (g->bias == nullptr) g->bias = g->alpha;
    g->grid_m = (g->M + 255) / 256;
    g->grid_n = (g->N + 175) / 176;
  }
  return launch_fr_pair(g0, g1, as_stream(stream));
}

int mq_gemm_tiled128_supported(int64_t M, int64_t N, int64_t K) { return gemm_fr128_shape(M, N, K) ? 1 : 0; }

static std::atomic<int> g_seg_tile{0};       // tuning hook: 128 = keep the 256 x 128 tile for the segmented GEMM
int mq_gemm_set_segmented_tile(int cols) {
  g_seg_tile = cols == 128 ? 128 : 0;
  return 0;
}

static std::atomic<int> g_fr128r_tile{0};     // tuning hook (mobilequant_amd_tuning.h): 0 = by shape, 128 / 256 = force the tile height
int mq_gemm_set_residual_tile(int rows) {
#ifdef MQ_BUILD_EXPERIMENTS
  g_fr128r_tile = (rows == 128 || rows == 256 || rows == 512) ? rows : 0;     // 512: 256-row tiles, K split over two workgroups
  return 0;
#else
  g_fr128r_tile = (rows == 128 || rows == 256) ? rows : 0;                    // (the split-K variant exists in experiment builds only)
  return rows == 512 ? 1 : 0;
#endif
}

int mq_w8a8_linear_tiled_residual(const int8_t* a_tiled, const int8_t* w, int64_t M, int64_t N, int64_t K, const int32_t* a_rowsum,
                                  const float* alpha, const int32_t* w_zp, const int32_t* col_term, const float* bias,
                                  const float* out_scale, const float* out_offset, float out_qmin, float out_qmax,
                                  const float* resid, float* out, mq_stream_t stream) {
  const char* fn = "mq_w8a8_linear_tiled_residual";
  int rc = check_common(fn, a_tiled, w, M, N, K, a_rowsum, alpha, w_zp, col_term, bias, out_scale, out_offset, out, 1);
  if (rc != MQ_OK) return rc;
  MQ_REQUIRE(((M + 15) / 16) * 16 * K < (1ll << 32) && M * N * 4 < (1ll << 32), "%s: operand too large", fn);
  MQ_REQUIRE(resid != nullptr && aligned(resid, 16), "%s: resid must be non-null and 16-byte aligned", fn);
  MQ_REQUIRE(out_scale != nullptr && out_qmax - out_qmin > 255.0f, "%s: a 16-bit output grid is required (8-bit grids: mq_w8a8_linear_residual)", fn);
  if (!gemm_fr128_shape(M, N, K)) {
    set_error("%s: shape %lldx%lldx%lld is not served (mq_gemm_tiled128_supported: N %% 128 == 0, K %% 256 == 0, K >= 768)", fn, (long long)M,
              (long long)N, (long long)K);
    return MQ_EUNSUPPORTED;
  }
  GemmArgs g{a_tiled, w, (int)M, (int)N, (int)K, a_rowsum, alpha, w_zp, col_term, bias, out_scale, out_offset,
             out_qmin, out_qmax, out, MQ_F32, 0, 0, bias != nullptr, 1, 0, g_dbg_ts, resid};
  // 128-row tiles while they give every CU at most ~two rounds of work; taller tiles (twice the MFMAs per W fragment) beyond
  const int forced = g_fr128r_tile.load();
  const int64_t tiles128 = ((M + 127) / 128) * (N / 128);
  // split-K on 256-row tiles (mq_gemm_set_residual_tile(512)): both halves of every tile resident at once (2 x tiles <= CUs), the two
  // workgroups of a tile on one XCD (tiles % 8 == 0), each half a valid K for the program (multiple of 256, >= 768)
  const int64_t tiles256 = ((M + 255) / 256) * (N / 128);
  const bool can_split = K % 512 == 0 && K / 2 >= 768 && tiles256 % 8 == 0 && 2 * tiles256 <= (int64_t)device_cu_count();
#ifdef MQ_BUILD_EXPERIMENTS
  if (forced == 512 && can_split) return launch_fr128<FR128RS>(g, as_stream(stream));
#else
  (void)can_split;
#endif
  const bool tall = forced ? forced == 256 : tiles128 > 512;
  return tall ? launch_fr128<FR128R8>(g, as_stream(stream)) : launch_fr128<FR128R>(g, as_stream(stream));
}

int mq_w4a8_linear_tiled_residual(const int8_t* a_tiled, const uint8_t* w_packed, int64_t M, int64_t N, int64_t K, const int32_t* a_rowsum,
                                  const float* alpha, const int32_t* w_zp, const int32_t* col_term, const float* bias,
                                  const float* out_scale, const float* out_offset, float out_qmin, float out_qmax,
                                  const float* resid, float* out, mq_stream_t stream) {
  const char* fn = "mq_w4a8_linear_tiled_residual";
  int rc = check_common(fn, a_tiled, w_packed, M, N, K, a_rowsum, alpha, w_zp, col_term, bias, out_scale, out_offset, out, 2);
  if (rc != MQ_OK) return rc;
  MQ_REQUIRE(((M + 15) / 16) * 16 * K < (1ll << 32) && M * N * 4 < (1ll << 32), "%s: operand too large", fn);
  MQ_REQUIRE(resid != nullptr && aligned(resid, 16), "%s: resid must be non-null and 16-byte aligned", fn);
  MQ_REQUIRE(out_scale != nullptr && out_qmax - out_qmin > 255.0f, "%s: a 16-bit output grid is required", fn);
  if (!gemm_fr128_shape(M, N, K)) {
    set_error("%s: shape %lldx%lldx%lld is not served (mq_gemm_tiled128_supported: N %% 128 == 0, K %% 256 == 0, K >= 768)", fn, (long long)M,
              (long long)N, (long long)K);
    return MQ_EUNSUPPORTED;
  }
  GemmArgs g{a_tiled, w_packed, (int)M, (int)N, (int)K, a_rowsum, alpha, w_zp, col_term, bias, out_scale, out_offset,
             out_qmin, out_qmax, out, MQ_F32, 0, 0, bias != nullptr, 1, 0, g_dbg_ts, resid};
  return launch_frw4r(g, as_stream(stream));
}

int mq_w8a8_linear_tiled_segmented(const int8_t* a_tiled, const int8_t* w, int64_t M, int64_t N, int64_t K, const int32_t* a_rowsum,
                                   const float* alpha, const int32_t* w_zp, const int32_t* col_term, const float* bias, int n_segments,
                                   const int64_t* seg_end, const mq_grid* grids, uint8_t* out, mq_stream_t stream) {
  const char* fn = "mq_w8a8_linear_tiled_segmented";
  MQ_REQUIRE(n_segments >= 1 && n_segments <= 3 && seg_end != nullptr && grids != nullptr, "%s: 1..3 segments", fn);
  int rc = check_common(fn, a_tiled, w, M, N, K, a_rowsum, alpha, w_zp, col_term, bias, grids[0].scale, grids[0].offset, out, 1);
  if (rc != MQ_OK) return rc;
  MQ_REQUIRE(((M + 15) / 16) * 16 * K < (1ll << 32), "%s: activation too large", fn);
  int64_t prev = 0;
  for (int i = 0; i < n_segments; ++i) {
    MQ_REQUIRE(grids[i].scale && grids[i].offset && grids[i].qmin == 0.f && grids[i].qmax == 255.f,
               "%s: segment %d needs an 8-bit unsigned output grid", fn, i);
    MQ_REQUIRE(seg_end[i] > prev && seg_end[i] <= N && seg_end[i] % 4 == 0, "%s: segment ends must increase, be multiples of 4, <= N", fn);
    prev = seg_end[i];
  }
  MQ_REQUIRE(prev == N, "%s: the last segment must end at N", fn);
  if (!gemm_fr128_shape(M, N, K)) {
    set_error("%s: shape %lldx%lldx%lld is not served (mq_gemm_tiled128_supported: N %% 128 == 0, K %% 256 == 0, K >= 768)", fn, (long long)M,
              (long long)N, (long long)K);
    return MQ_EUNSUPPORTED;
  }
  GemmArgs g{a_tiled, w, (int)M, (int)N, (int)K, a_rowsum, alpha, w_zp, col_term, bias, grids[0].scale, grids[0].offset,
             0.f, 255.f, out, MQ_U8, 0, 0, bias != nullptr, 1, 0, g_dbg_ts, nullptr, {0, 0}, {nullptr, nullptr}, {nullptr, nullptr}};
  for (int i = 1; i < n_segments; ++i) {
    g.seg_end[i - 1] = (int)seg_end[i - 1];
    g.seg_scale[i - 1] = grids[i].scale;
    g.seg_offset[i - 1] = grids[i].offset;
  }
  // 128 x 160 tiles when they give every CU exactly (at most) one tile -- q | k | v at M = 2048: 256 tiles against 160 of 256 x 128
  if (N % 160 == 0 && ((M + 127) / 128) * (N / 160) <= 256 && g_seg_tile.load() != 128) return launch_fr128<FR160>(g, as_stream(stream));
  return launch_fr128<FR128>(g, as_stream(stream));
}

int mq_w8a8_linear_tiled_gated(const int8_t* a_tiled, int64_t M, int64_t N, int64_t K, const int32_t* a_rowsum,
                               const int8_t* w0, const float* alpha0, const int32_t* w_zp0, const int32_t* col_term0,
                               const float* bias0, const float* out_scale0, const float* out_offset0,
                               const int8_t* w1, const float* alpha1, const int32_t* w_zp1, const int32_t* col_term1,
                               const float* bias1, const float* out_scale1, const float* out_offset1,
                               const int8_t* table, uint8_t* idx_scratch, int8_t* q_tiled, int32_t* row_sum, mq_stream_t stream) {
  const char* fn = "mq_w8a8_linear_tiled_gated";
  int rc = check_common(fn, a_tiled, w0, M, N, K, a_rowsum, alpha0, w_zp0, col_term0, bias0, out_scale0, out_offset0, idx_scratch, 1);
  if (rc != MQ_OK) return rc;
  rc = check_common(fn, a_tiled, w1, M, N, K, a_rowsum, alpha1, w_zp1, col_term1, bias1, out_scale1, out_offset1, q_tiled, 1);
  if (rc != MQ_OK) return rc;
  MQ_REQUIRE(((M + 15) / 16) * 16 * K < (1ll << 32) && ((M + 15) / 16) * 16 * N < (1ll << 32), "%s: operand too large", fn);
  MQ_REQUIRE(out_scale0 && out_scale1 && table && row_sum && aligned(table, 16) && aligned(idx_scratch, 16) && aligned(q_tiled, 16),
             "%s: both linears carry an 8-bit unsigned output grid; table / scratch / image must be non-null and 16-byte aligned", fn);
  GemmArgs g0{a_tiled, w0, (int)M, (int)N, (int)K, a_rowsum, alpha0, w_zp0, col_term0, bias0, out_scale0, out_offset0,
              0.f, 255.f, idx_scratch, MQ_U8, 0, 0, bias0 != nullptr, 1, 0, g_dbg_ts};
  GemmArgs g1{a_tiled, w1, (int)M, (int)N, (int)K, a_rowsum, alpha1, w_zp1, col_term1, bias1, out_scale1, out_offset1,
              0.f, 255.f, q_tiled, MQ_U8, 0, 0, bias1 != nullptr, 1, 0, g_dbg_ts};
  const bool wide = gemm_tiled_supported(M, N, K) && gemm_fr_supported(g0) && N % 64 == 0;         // 256 x 176 tiles
  const bool narrow = !wide && gemm_fr128_shape(M, N, K) && ((M + 255) / 256) * (N / 128) >= 192;  // 256 x 128 tiles (Gemma: N = 16384)
  if (!(wide || narrow) || M > 256 * 512) {
    set_error("%s: shape %lldx%lldx%lld is not served (mq_gemm_tiled_supported or N %% 128 == 0 with >= 192 tiles; K %% 256 == 0, K >= 768)",
              fn, (long long)M, (long long)N, (long long)K);
    return MQ_EUNSUPPORTED;
  }
  g0.zero_buf = row_sum;
  g0.zero_count = (int)M;
  g1.gate_aidx = idx_scratch;
  g1.gate_table = table;
  g1.gate_q = q_tiled;
  g1.gate_rowsum = row_sum;
  if (narrow) {
    rc = launch_fr128<FR128>(g0, as_stream(stream));
    if (rc != MQ_OK) return rc;
    g1.has_rowsum = g1.a_rowsum != nullptr;
    if (g1.a_rowsum == nullptr) g1.a_rowsum = g1.col_term;
    if (g1.bias == nullptr) g1.bias = g1.alpha;
    g1.grid_m = (g1.M + 255) / 256;
    g1.grid_n = g1.N / 128;
    return launch_frg<128>(g1, as_stream(stream));
  }
  for (GemmArgs* g : {&g0, &g1}) {
    g->has_rowsum = g->a_rowsum != nullptr;
    if (g->a_rowsum == nullptr) g->a_rowsum = g->col_term;
    if (g->bias == nullptr) g->bias = g->alpha;
    g->grid_m = (g->M + 255) / 256;
    g->grid_n = (g->N + 175) / 176;
  }
  rc = launch_fr(g0, as_stream(stream));
  if (rc != MQ_OK) return rc;
  return launch_frg<176>(g1, as_stream(stream));
}

int mq_w4a8_linear_tiled_gated(const int8_t* a_tiled, int64_t M, int64_t N, int64_t K, const int32_t* a_rowsum,
                               const uint8_t* w0, const float* alpha0, const int32_t* w_zp0, const int32_t* col_term0,
                               const float* bias0, const float* out_scale0, const float* out_offset0,
                               const uint8_t* w1, const float* alpha1, const int32_t* w_zp1, const int32_t* col_term1,
                               const float* bias1, const float* out_scale1, const float* out_offset1,
                               const int8_t* table, uint8_t* idx_scratch, int8_t* q_tiled, int32_t* row_sum, mq_stream_t stream) {
  const char* fn = "mq_w4a8_linear_tiled_gated";
  int rc = check_common(fn, a_tiled, w0, M, N, K, a_rowsum, alpha0, w_zp0, col_term0, bias0, out_scale0, out_offset0, idx_scratch, 2);
  if (rc != MQ_OK) return rc;
  rc = check_common(fn, a_tiled, w1, M, N, K, a_rowsum, alpha1, w_zp1, col_term1, bias1, out_scale1, out_offset1, q_tiled, 2);
  if (rc != MQ_OK) return rc;
  MQ_REQUIRE(((M + 15) / 16) * 16 * K < (1ll << 32) && ((M + 15) / 16) * 16 * N < (1ll << 32), "%s: operand too large", fn);
  MQ_REQUIRE(out_scale0 && out_scale1 && table && row_sum && aligned(table, 16) && aligned(idx_scratch, 16) && aligned(q_tiled, 16),
             "%s: both linears carry an 8-bit unsigned output grid; table / scratch / image must be non-null and 16-byte aligned", fn);
  GemmArgs g0{a_tiled, w0, (int)M, (int)N, (int)K, a_rowsum, alpha0, w_zp0, col_term0, bias0, out_scale0, out_offset0,
              0.f, 255.f, idx_scratch, MQ_U8, 0, 0, bias0 != nullptr, 1, 0, g_dbg_ts};
  GemmArgs g1{a_tiled, w1, (int)M, (int)N, (int)K, a_rowsum, alpha1, w_zp1, col_term1, bias1, out_scale1, out_offset1,
              0.f, 255.f, q_tiled, MQ_U8, 0, 0, bias1 != nullptr, 1, 0, g_dbg_ts};
  const bool wide = gemm_tiled_supported(M, N, K) && N % 176 == 0 && N % 64 == 0 && K % 256 == 0 && K >= 768;   // 256 x 176 tiles
  const bool narrow = !wide && gemm_fr128_shape(M, N, K) && ((M + 255) / 256) * (N / 128) >= 192;                 // 256 x 128 tiles
  if (!(wide || narrow) || M > 256 * 512) {
    set_error("%s: shape %lldx%lldx%lld is not served (as mq_w8a8_linear_tiled_gated)", fn, (long long)M, (long long)N, (long long)K);
    return MQ_EUNSUPPORTED;
  }
  g0.zero_buf = row_sum;
  g0.zero_count = (int)M;
  g1.gate_aidx = idx_scratch;
  g1.gate_table = table;
  g1.gate_q = q_tiled;
  g1.gate_rowsum = row_sum;
  rc = wide ? launch_frw4<176, true>(g0, as_stream(stream)) : launch_frw4<128, true>(g0, as_stream(stream));
  if (rc != MQ_OK) return rc;
  g1.has_rowsum = g1.a_rowsum != nullptr;
  if (g1.a_rowsum == nullptr) g1.a_rowsum = g1.col_term;
  if (g1.bias == nullptr) g1.bias = g1.alpha;
  g1.grid_m = (g1.M + 255) / 256;
  g1.grid_n = wide ? g1.N / 176 : g1.N / 128;
  return wide ? launch_frgw4<176>(g1, as_stream(stream)) : launch_frgw4<128>(g1, as_stream(stream));
}

static int linear_f32in(const char* fn, int w4, const float* x, const float* a_scale, const float* a_offset, float a_qmin,
                        float a_qmax, int a_shift, const void* w, int64_t M, int64_t N, int64_t K, const float* alpha,
                        const int32_t* w_zp, const int32_t* col_term, const float* bias, const float* out_scale,
                        const float* out_offset, float out_qmin, float out_qmax, void* out, int out_dtype,
                        mq_stream_t stream) {
  int rc = check_common(fn, x, w, M, N, K, nullptr, alpha, w_zp, col_term, bias, out_scale, out_offset, out, w4 ? 2 : 1);
  if (rc != MQ_OK) return rc;
  MQ_REQUIRE(a_scale && a_offset, "%s: null activation grid", fn);
  if (!(M <= 8 && M * K <= 64 * 1024 - 64 && K % 256 == 0)) {
    set_error("%s: decode shapes only (M <= 8, M*K < 64 KiB, K %% 256 == 0); use mq_quantize + the int8-input entry point", fn);
    return MQ_EUNSUPPORTED;
  }
  if (out_scale == nullptr && out_dtype != MQ_F32 && out_dtype != MQ_F16) {
    set_error("%s: integer out_dtype %d needs an output quantizer", fn, out_dtype);
    return MQ_EINVAL;
  }
  GemvArgs v{nullptr, (const int8_t*)w, (int)M, (int)N, (int)K, nullptr, alpha, w_zp, col_term, bias, out_scale, out_offset,
             out_qmin, out_qmax, out, out_dtype, x, a_scale, a_offset, a_qmin, a_qmax, a_shift, w4};
  return run_gemv(v, as_stream(stream));
}

int mq_w8a8_linear_f32in(const float* x, const float* a_scale, const float* a_offset, float a_qmin, float a_qmax,
                         int a_shift, const int8_t* w, int64_t M, int64_t N, int64_t K, const float* alpha,
                         const int32_t* w_zp, const int32_t* col_term, const float* bias, const float* out_scale,
                         const float* out_offset, float out_qmin, float out_qmax, void* out, int out_dtype,
                         mq_stream_t stream) {
  return linear_f32in("mq_w8a8_linear_f32in", 0, x, a_scale, a_offset, a_qmin, a_qmax, a_shift, w, M, N, K, alpha, w_zp,
                      col_term, bias, out_scale, out_offset, out_qmin, out_qmax, out, out_dtype, stream);
}

int mq_w4a8_linear_f32in(const float* x, const float* a_scale, const float* a_offset, float a_qmin, float a_qmax,
                         int a_shift, const uint8_t* w_packed, int64_t M, int64_t N, int64_t K, const float* alpha,
                         const int32_t* w_zp, const int32_t* col_term, const float* bias, const float* out_scale,
                         const float* out_offset, float out_qmin, float out_qmax, void* out, int out_dtype,
                         mq_stream_t stream) {
  return linear_f32in("mq_w4a8_linear_f32in", 1, x, a_scale, a_offset, a_qmin, a_qmax, a_shift, w_packed, M, N, K, alpha,
                      w_zp, col_term, bias, out_scale, out_offset, out_qmin, out_qmax, out, out_dtype, stream);
}

int mq_w4a8_linear(const int8_t* a, const uint8_t* w_packed, int64_t M, int64_t N, int64_t K, const int32_t* a_rowsum,
                   const float* alpha, const int32_t* w_zp, const int32_t* col_term, const float* bias,
                   const float* out_scale, const float* out_offset, float out_qmin, float out_qmax, void* out,
                   int out_dtype, mq_stream_t stream) {
  int rc = check_common("mq_w4a8_linear", a, w_packed, M, N, K, a_rowsum, alpha, w_zp, col_term, bias, out_scale,
                        out_offset, out, 2);
  if (rc != MQ_OK) return rc;
  if (M <= 8 && M * K <= 64 * 1024 - 64 && g_forced_variant < 0) {   // decode shapes: nibble-streaming GEMV (mq_gemv.hip)
    if (out_scale == nullptr && out_dtype != MQ_F32 && out_dtype != MQ_F16) {
      set_error("mq_w4a8_linear: integer out_dtype %d needs an output quantizer", out_dtype);
      return MQ_EINVAL;
    }
    GemvArgs v{a, (const int8_t*)w_packed, (int)M, (int)N, (int)K, a_rowsum, alpha, w_zp, col_term, bias, out_scale,
               out_offset, out_qmin, out_qmax, out, out_dtype, nullptr, nullptr, nullptr, 0.f, 0.f, 0, 1};
    return run_gemv(v, as_stream(stream));
  }
  GemmArgs g{a, w_packed, (int)M, (int)N, (int)K, a_rowsum, alpha, w_zp, col_term, bias, out_scale, out_offset,
             out_qmin, out_qmax, out, out_dtype, 0, 0, bias != nullptr, 0, 0, g_dbg_ts};
  return run_gemm<true>(g, as_stream(stream));
}

}  // extern "C"

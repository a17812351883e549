// Quantized causal attention for prefill: the core of HFAttention.forward (mobilellm/model/hf_model.py:486-534) with its two QMatMuls
// (mobilellm/quantization/qmodule.py:453-466) as REAL integer matrix products, the S x S score tensor never written to memory.
//
//   reference (per head):  q, k <- RoPE (fp32)
//     s  = Qqk_out( matmul(Qqk_a(q), Qqk_b(k^T)) ) / sqrt(D)  (+ causal mask)          8-bit x 8-bit -> 16-bit grid,  [S, S] fp32
//     p  = softmax(s)                                                                   fp32
//     o  = Qpv_out( matmul(Qpv_a(p), Qpv_b(v)) )                                        16-bit x 8-bit -> 8-bit grid
//   The simulated form makes ~6 passes over 32 x S x S floats per layer (537 MB each at S = 2048): that, not the linears, is where a
//   prefill layer's time goes.
//
// Here:  mq_attention_prep   RoPE + the three input quantizers -> int8 images: q [H][S][D], k [KV][S][D] (+ row sums of the stored
//                            values) and v TRANSPOSED and key-permuted per 64-key block: vT [KV][S/64][D][64]
//        mq_attention_quant  one workgroup per (64-query block, head), 4 waves x 16 queries.  Two sweeps over the key blocks up to the
//                            diagonal, both on v_mfma_i32_16x16x64_i8:
//                              sweep 1: integer q.k^T -> zero-point correction -> 16-bit output grid -> /sqrt(D) -> mask -> online row
//                                       max / sum of exp
//                              sweep 2: the same scores again (bit-identical), p = exp(s - m) / l -> 16-bit grid index -> split into
//                                       two unsigned bytes -> integer p.v as TWO int8 products (high and low byte) -> exact integer
//                                       combination in double -> 8-bit output grid
//                            The MFMA is issued as mfma(K tile, Q tile): D[t][s], a lane owns ONE query row and 4 keys per tile, so the
//                            softmax statistics are lane-local plus two cross-lane steps.  The D layout of the scores is used directly
//                            as the A-operand layout of the p.v product by permuting the keys inside a 64-block (kappa = 16 tq + 4 j + e
//                            <-> t = 16 j + 4 tq + e); vT is stored in that order, so no data moves between the two products.
// Arithmetic: integer contractions are exact; quantizers use the reciprocal-multiply form of the GEMM epilogues (index within one grid
// step of the divide form on a vanishing fraction of elements); softmax in fp32.
#include "mq_common.h"
#include <type_traits>

namespace mq {

typedef int v4i __attribute__((ext_vector_type(4)));
typedef float v4f __attribute__((ext_vector_type(4)));
typedef _Float16 v8h __attribute__((ext_vector_type(8)));
// two centred indices (integers of magnitude <= 255: exact in fp16) -> one dword of two halves
__device__ __forceinline__ unsigned pack_h2(float a, float b) { return __builtin_bit_cast(unsigned, __builtin_amdgcn_cvt_pkrtz(a, b)); }

#pragma clang fp contract(off)

#ifndef MQ_ATT_EREGS
#define MQ_ATT_EREGS 5    // ... and of the blocks in front of those, kept in registers (16 VGPRs each; two waves per SIMD leave room)
#endif
#ifndef MQ_ATT_ECACHE
#define MQ_ATT_ECACHE 4   // key blocks per row block whose sweep-1 exponentials are kept for sweep 2 (16 KiB of LDS each: 64 KiB = two workgroups per CU)
#endif
// the f16 form stages K / vT tiles in an LDS ring of MQ_ATT_F16_STAGES x 12 KiB, requested STAGES - 1 blocks ahead: with three stages,
// two parked blocks in the LDS (32 + 36 = 68 KiB: two workgroups per CU) and, with no K tile living in registers across the quantizer
// chain, seven in registers
#ifndef MQ_ATT_PK
#define MQ_ATT_PK 0       // 1 = v_pk_fma_f32 in the f16 form's quantizer chain (identical bits): measured 2-3 % SLOWER -- a packed fma costs the
#endif                    // VALU what two scalar ones do (tools/valu_rate_probe.cpp: 4.6 against 2 x 2.5 cycles) and pairs constrain the schedule
#ifndef MQ_ATT_F16_STAGES
#define MQ_ATT_F16_STAGES 3
#endif
#ifndef MQ_ATT_EREGS_F16
#define MQ_ATT_EREGS_F16 7
#endif
#ifndef MQ_ATT_ECACHE_F16
#define MQ_ATT_ECACHE_F16 (MQ_ATT_F16_STAGES == 2 ? 3 : 2)
#endif
// One LDS-DMA piece (1 KiB per wave: lane l's 16 bytes land at lds_base + 16 l) as inline assembly: hipcc's waitcnt pass cannot tell
// which LDS bytes a DMA it knows about will write, and puts s_waitcnt vmcnt(0) in front of EVERY later ds_read -- the block being read
// then waits for the block just requested (measured: 40 % of the wave cycles in s_waitcnt).  The asm form is invisible to that pass;
// the waits are counted by hand (wait_dma below).
__device__ __forceinline__ void lds_dma16(const void* sbase, unsigned voff, unsigned lds_base) {
#if defined(MQ_ATT_ABL) && MQ_ATT_ABL == 8     // what-if: the same bytes as a plain load into registers (nothing reaches the LDS)
  v4i sink;
  asm volatile("global_load_dwordx4 %0, %1, %2" : "=v"(sink) : "v"(voff), "s"(sbase) : "memory");
  return;
#endif
  unsigned keep;                                   // m0 is the compiler's to use: hand it back as found
  asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %1\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %2, %3\n\ts_mov_b32 m0, %0"
               : "=&s"(keep)
               : "s"(lds_base), "v"(voff), "s"(sbase)
               : "memory");
}
// a 256-byte piece: lane l's dword lands at lds_base + 4 l
__device__ __forceinline__ void lds_dma4(const void* sbase, unsigned voff, unsigned lds_base) {
  unsigned keep;
  asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %1\n\ts_nop 0\n\tglobal_load_lds_dword %2, %3\n\ts_mov_b32 m0, %0"
               : "=&s"(keep)
               : "s"(lds_base), "v"(voff), "s"(sbase)
               : "memory");
}
// two consecutive pieces (the instruction offset moves the memory AND the LDS address)
__device__ __forceinline__ void lds_dma16x2(const void* sbase, unsigned voff, unsigned lds_base) {
#if defined(MQ_ATT_ABL) && MQ_ATT_ABL == 8
  v4i sink0, sink1;
  asm volatile("global_load_dwordx4 %0, %2, %3\n\tglobal_load_dwordx4 %1, %2, %3 offset:1024" : "=v"(sink0), "=v"(sink1) : "v"(voff), "s"(sbase) : "memory");
  return;
#endif
  unsigned keep;
  asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %1\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %2, %3\n\tglobal_load_lds_dwordx4 %2, %3 offset:1024\n\ts_mov_b32 m0, %0"
               : "=&s"(keep)
               : "s"(lds_base), "v"(voff), "s"(sbase)
               : "memory");
}
// A fragment read the compiler neither waits for nor moves: issued here, complete after lds_fragments_wait (which names every
// destination, so no consumer can be scheduled above it)
template <int OFF>
__device__ __forceinline__ void lds_read_frag(v4i& dst, unsigned addr) {
  asm volatile("ds_read_b128 %0, %1 offset:%2" : "=v"(dst) : "v"(addr), "n"(OFF));
}
__device__ __forceinline__ void lds_fragments_wait(v4i (&f)[8]) {
  asm volatile("s_waitcnt lgkmcnt(0)" : "+v"(f[0]), "+v"(f[1]), "+v"(f[2]), "+v"(f[3]), "+v"(f[4]), "+v"(f[5]), "+v"(f[6]), "+v"(f[7]));
}
template <int BASE, int... I>
__device__ __forceinline__ void lds_read_frags(v4i* dst, unsigned addr, std::integer_sequence<int, I...>) {
  (lds_read_frag<BASE + I * 1024>(dst[I], addr), ...);
}
__device__ __forceinline__ void lds_fragments_wait(v4i (&f)[4]) {
  asm volatile("s_waitcnt lgkmcnt(0)" : "+v"(f[0]), "+v"(f[1]), "+v"(f[2]), "+v"(f[3]));
}
__device__ __forceinline__ void lds_fragments_wait(v4i (&f)[16]) {
  asm volatile("s_waitcnt lgkmcnt(0)" : "+v"(f[0]), "+v"(f[1]), "+v"(f[2]), "+v"(f[3]), "+v"(f[4]), "+v"(f[5]), "+v"(f[6]), "+v"(f[7]), "+v"(f[8]), "+v"(f[9]),
               "+v"(f[10]), "+v"(f[11]), "+v"(f[12]), "+v"(f[13]), "+v"(f[14]), "+v"(f[15]));
}
__device__ __forceinline__ void tie16(float (&x)[16]) {   // an ordering point for the sixteen values (no instruction)
  asm volatile("" : "+v"(x[0]), "+v"(x[1]), "+v"(x[2]), "+v"(x[3]), "+v"(x[4]), "+v"(x[5]), "+v"(x[6]), "+v"(x[7]), "+v"(x[8]), "+v"(x[9]), "+v"(x[10]),
               "+v"(x[11]), "+v"(x[12]), "+v"(x[13]), "+v"(x[14]), "+v"(x[15]));
}

struct AGrid {
  float s, o, qmin, qmax, inv_s;
  bool on;
};
__device__ __forceinline__ AGrid a_load_grid(const mq_grid& g) {
  AGrid r;
  r.on = g.scale != nullptr;
  r.s = r.on ? g.scale[0] : 1.f;
  r.o = r.on ? g.offset[0] : 0.f;
  r.qmin = g.qmin;
  r.qmax = g.qmax;
  r.inv_s = __fdiv_rn(1.0f, r.s);
  return r;
}
// reciprocal-multiply form, used inside the attention kernel
__device__ __forceinline__ float a_index_fast(float x, const AGrid& g) {
  return fminf(fmaxf(rintf(x * g.inv_s) + g.o, g.qmin), g.qmax);
}

// batch > 1 (no cache continuation): sequence b = blockIdx.z of one launch.  Inputs, outputs and scratch are laid out [batch][...] with
// the per-sequence shapes of the single-sequence call (include/mobilequant_amd.h), cos / sin are shared; the view below is that call's
// argument block for sequence b.  (By value: a reference would pin the kernel's argument block in scratch.)
__device__ __forceinline__ mq_attention_args batch_view(mq_attention_args a, const int b) {
  if (a.batch > 1) {
    const size_t S = (size_t)a.seq, H = (size_t)a.heads, KV = (size_t)a.kv_heads, D = (size_t)a.head_dim;
    if (a.qkv_idx) a.qkv_idx += b * S * (H + 2 * KV) * D;
    if (a.q) { a.q += b * S * H * D; a.k += b * S * KV * D; a.v += b * S * KV * D; }
    if (a.out) a.out += b * S * H * D;
    a.out_row0 += (int64_t)b * a.seq_real;
    a.q_i8 += b * H * S * D; a.k_i8 += b * KV * S * D; a.vt_i8 += b * KV * S * D;
    a.q_rowsum += b * H * S; a.k_rowsum += b * KV * S;
    if (a.q_f16) { a.q_f16 += b * H * S * D; a.k_f16 += b * KV * S * D; }
    if (a.v_prefix) a.v_prefix += b * KV * (S / 64) * D;
  }
  return a;
}

// ---- prep: RoPE + input quantizers -> integer images ----------------------------------------------------------------------------
// grid: (S / 64, H + 2 KV).  Block b of part p: rows s = 64 b .. 64 b + 63.  256 threads: thread (r = tid >> 2, c = tid & 3) handles
// row r, 16 columns 64 dc + 16 c .. + 15 of every 64-column slab dc (D = 64: one slab; D = 256: four).
template <int D>
__global__ void __launch_bounds__(256) attention_prep_kernel(const mq_attention_args a, const int part0) {
  const int H = a.heads, KV = a.kv_heads, S = a.seq;
  // batch > 1: sequence blockIdx.z (batch_view's offsets, as plain locals: a modified copy of the argument block would live in scratch here)
  const size_t bz = a.batch > 1 ? blockIdx.z : 0, bS = bz * (size_t)S;
  const float* const q_in = a.q ? a.q + bS * H * D : nullptr;
  const float* const k_in = a.k ? a.k + bS * KV * D : nullptr;
  const float* const v_in = a.v ? a.v + bS * KV * D : nullptr;
  const uint8_t* const idx_in = a.qkv_idx ? a.qkv_idx + bS * (H + 2 * KV) * D : nullptr;
  int8_t* const q_img = a.q_i8 + bz * H * S * D;
  int8_t* const k_img = a.k_i8 + bz * KV * S * D;
  int8_t* const vt_img = a.vt_i8 + bz * KV * S * D;
  int32_t* const q_rs = a.q_rowsum + bz * H * S;
  int32_t* const k_rs = a.k_rowsum + bz * KV * S;
  uint16_t* const q_h = a.q_f16 ? a.q_f16 + bz * H * S * D : nullptr;
  uint16_t* const k_h = a.k_f16 ? a.k_f16 + bz * KV * S * D : nullptr;
  int32_t* const v_pre = a.v_prefix ? a.v_prefix + bz * KV * (S / 64) * D : nullptr;
  const int64_t row0 = a.out_row0 + (int64_t)bz * a.seq_real;
  // cache continuation (chunked prefill): the K / vT images, their row sums and the v prefix sums are caller-owned caches of cache_seq
  // rows; this chunk's rows go to positions pos0 .. pos0 + seq - 1 (pos0 % 64 == 0).  cache_seq = 0: scratch of seq rows, pos0 = 0.
  const int CS = a.cache_seq > 0 ? a.cache_seq : S, P0 = a.cache_seq > 0 ? a.pos0 : 0;
  const int part = blockIdx.y + part0;               // [0, H): q head; [H, H+KV): k head; [H+KV, H+2KV): v head  (part0 = H: the core kernel prepares its own q rows)
  const int r = threadIdx.x >> 2, c = threadIdx.x & 3;
  const int s = blockIdx.x * 64 + r;
  __shared__ int8_t s_v[64][64 + 4];
  if (a.out_i8 != nullptr && blockIdx.y == 0 && threadIdx.x < 64 && (int)(blockIdx.x * 64 + threadIdx.x) < a.seq_real)
    a.out_rowsum[row0 + blockIdx.x * 64 + threadIdx.x] = 0;          // the core kernel accumulates one share per head
  __syncthreads();
  const bool is_q = part < H, is_k = !is_q && part < H + KV;
  const int head = is_q ? part : (is_k ? part - H : part - H - KV);
  const float* src = (is_q ? q_in : (is_k ? k_in : v_in)) + (size_t)s * (is_q ? H : KV) * D + (size_t)head * D;
  auto load16 = [](const float* p, float (&d)[16]) {
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      const float4 t = reinterpret_cast<const float4*>(p)[i];
      d[4 * i] = t.x; d[4 * i + 1] = t.y; d[4 * i + 2] = t.z; d[4 * i + 3] = t.w;
    }
  };
  // index input: the uint8 output indices of the fused q|k|v GEMM, dequantised as that linear's fp32 output would read
  const uint8_t* isrc = idx_in ? idx_in + ((size_t)s * (H + 2 * KV) + part) * D : nullptr;
  const AGrid gin = a_load_grid(is_q ? a.q_in : (is_k ? a.k_in : a.v_in));
  auto load16_idx = [&](const uint8_t* p, float (&d)[16]) {
    const uint4 t = *reinterpret_cast<const uint4*>(p);
    const unsigned w4[4] = {t.x, t.y, t.z, t.w};
#pragma unroll
    for (int i = 0; i < 16; ++i) d[i] = __fmul_rn(__fsub_rn((float)((w4[i >> 2] >> (8 * (i & 3))) & 0xffu), gin.o), gin.s);
  };
  const AGrid g = a_load_grid(is_q ? a.qk_a : (is_k ? a.qk_b : a.pv_b));
  const int rot = a.rot_dim > 0 ? a.rot_dim : D;
  uint32_t usum = 0;                                 // sum of the stored bytes + 128 each (image_pack4, mq_common.h)
#pragma unroll 1
  for (int dc = 0; dc < D / 64; ++dc) {
  const int col0 = 64 * dc + 16 * c;
  float x[16];
  if (isrc) load16_idx(isrc + col0, x);
  else load16(src + col0, x);
  float y16[16];                                     // the values the input quantizer sees
  if ((is_q || is_k) && rot != D) {
    // partial rotary (hf_model.py:489-500; StableLM-2: 16 of 64 dims): dims d < rot rotate with partner d +- rot/2 and cos / sin
    // [S, rot]; the rest passes through.  Element-wise form (the full-rotary path below keeps its vector loads).
    const int half = rot >> 1;
    auto one = [&](int d) {
      if (isrc) return __fmul_rn(__fsub_rn((float)isrc[d], gin.o), gin.s);
      return src[d];
    };
#pragma unroll
    for (int i = 0; i < 16; ++i) {
      const int d = col0 + i;
      float y = x[i];
      if (d < rot) {
        const float p = one(d < half ? d + half : d - half);
        const float sg = d < half ? -1.f : 1.f;
        y = __fadd_rn(__fmul_rn(x[i], a.cos[(size_t)s * rot + d]), __fmul_rn(sg * p, a.sin[(size_t)s * rot + d]));
      }
      y16[i] = y;
    }
  } else if (is_q || is_k) {                          // RoPE (rotate-half): x * cos + rot(x) * sin, rot(x)[d] = d < D/2 ? -x[d + D/2] : x[d - D/2]
    float pr[16], cs[16], sn[16];
    if (isrc) load16_idx(isrc + ((col0 + D / 2) & (D - 1)), pr);
    else load16(src + ((col0 + D / 2) & (D - 1)), pr);
    load16(a.cos + (size_t)s * D + col0, cs);
    load16(a.sin + (size_t)s * D + col0, sn);
    const float sign = col0 < D / 2 ? -1.f : 1.f;     // (-x) * sin == -(x * sin) exactly
#pragma unroll
    for (int i = 0; i < 16; ++i) y16[i] = __fadd_rn(__fmul_rn(x[i], cs[i]), __fmul_rn(sign * pr[i], sn[i]));
  } else {
#pragma unroll
    for (int i = 0; i < 16; ++i) y16[i] = x[i];
  }
  // exact divide form of the quantizer (qmodule.py:286-287), bytes = index - 128 (8-bit unsigned grids: the host checks), NaN -> qmin
  unsigned w[4];
  float qi[16];
#pragma unroll
  for (int i = 0; i < 16; ++i) qi[i] = image_idxf(y16[i], g.s, g.inv_s, g.o, g.qmin, g.qmax);
#pragma unroll
  for (int d4 = 0; d4 < 4; ++d4) w[d4] = image_pack4(qi[4 * d4], qi[4 * d4 + 1], qi[4 * d4 + 2], qi[4 * d4 + 3], usum);
  if constexpr (D == 64) {
    // fp16 images of the centred indices for the f16 score contraction (attention_quant_kernel<.., F16>); index - offset is exact
    // q: row-major [H][S][64].  k: FRAGMENT-BLOCKED per 64-key block (8 KiB = 8 fragments of 1 KiB): fragment 2 j + hf holds keys
    // 16 j .. + 15, lane l = (key & 15) + 16 tq at byte 16 l: the eight halves d = 16 tq + 8 hf .. + 7 -- one LDS-DMA instruction of
    // the attention kernel moves a fragment as 1 KiB of consecutive bytes into the layout its ds_read_b128 feeds the MFMA from.
    uint16_t* hdst = is_q ? q_h : (is_k ? k_h : nullptr);
    if (hdst != nullptr && q_h != nullptr && k_h != nullptr) {
      unsigned hw[8];
#pragma unroll
      for (int i = 0; i < 8; ++i) hw[i] = pack_h2(__fsub_rn(qi[2 * i], g.o), __fsub_rn(qi[2 * i + 1], g.o));
      // index - offset is exact in fp16 and sum_d (iq - zq)(ik - zk) < 2^24 only while |index - offset| <= 511, i.e. for an
      // offset in [-256, 511].  compute_scale_offset_from_min_max (qmodule.py:55-62) does not force zero into the range, so a narrow
      // range far from zero (or a loaded / trained offset) lies outside: poison the image (NaN scores -> NaN outputs, as mq_qmatmul's
      // `sane`) instead of contracting inexactly in silence; ops.attention_quant(f16=False) serves such a grid on the int8 contraction
      if (!(g.o >= -256.f && g.o <= 511.f)) {
#pragma unroll
        for (int i = 0; i < 8; ++i) hw[i] = 0x7e007e00u;
      }
      if (is_q) {
        hdst += ((size_t)head * S + s) * D + col0;
        reinterpret_cast<uint4*>(hdst)[0] = make_uint4(hw[0], hw[1], hw[2], hw[3]);
        reinterpret_cast<uint4*>(hdst)[1] = make_uint4(hw[4], hw[5], hw[6], hw[7]);
      } else {
        hdst += (((size_t)head * (CS >> 6) + (P0 >> 6) + blockIdx.x) * 8 + 2 * (r >> 4)) * 512 + ((r & 15) + 16 * c) * 8;
        reinterpret_cast<uint4*>(hdst)[0] = make_uint4(hw[0], hw[1], hw[2], hw[3]);
        reinterpret_cast<uint4*>(hdst + 512)[0] = make_uint4(hw[4], hw[5], hw[6], hw[7]);
      }
    }
  }
  if (is_q || is_k) {
    int8_t* dst = (is_q ? q_img + ((size_t)head * S + s) * D : k_img + ((size_t)head * CS + P0 + s) * D) + col0;
    *reinterpret_cast<uint4*>(dst) = make_uint4(w[0], w[1], w[2], w[3]);
    if (dc == D / 64 - 1) {
      int sum = (int)usum - 32 * D;                    // D / 4 bytes per thread, each stored + 128
      sum += __shfl_xor(sum, 1, 64);
      sum += __shfl_xor(sum, 2, 64);
      // the zero-point terms of sum_d (qi - zq)(ki - zk) = sum qs ks - zq' rowsum(ks) - zk' rowsum(qs) + D zq' zk'  (primes: - 128)
      const int zq = (int)a_load_grid(a.qk_a).o - 128, zk = (int)a_load_grid(a.qk_b).o - 128;
      if (c == 0) {
        if (is_q) q_rs[(size_t)head * S + s] = D * zq * zk - zk * sum;
        else k_rs[(size_t)head * CS + P0 + s] = -zq * sum;
      }
    }
  } else {
    // vT [KV][S/64][D][64]: position kappa of key t (inside its 64-block): t = 16 j + 4 tq + e  <->  kappa = 16 tq + 4 j + e
#pragma unroll
    for (int i = 0; i < 16; ++i) s_v[16 * c + i][r] = (int8_t)(w[i >> 2] >> (8 * (i & 3)));
    __syncthreads();
    // thread (d = tid >> 2, quarter c): 16 kappa = 16 c .. 16 c + 15  -> tq = c, (j, e) = (i >> 2, i & 3) -> t = 16 j + 4 c + e
    const int d = threadIdx.x >> 2;
    unsigned o4[4];
    int csum = 0;
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      unsigned pk = 0;
#pragma unroll
      for (int e = 0; e < 4; ++e) {
        const int vv = s_v[d][16 * j + 4 * c + e];
        csum += vv;
        pk |= ((unsigned)vv & 0xffu) << (8 * e);
      }
      o4[j] = pk;
    }
    int8_t* dst = vt_img + (((size_t)head * (CS >> 6) + (P0 >> 6) + blockIdx.x) * D + 64 * dc + d) * 64 + 16 * c;
    *reinterpret_cast<uint4*>(dst) = make_uint4(o4[0], o4[1], o4[2], o4[3]);
    if constexpr (D > 64) {                            // column sums of the stored values over this block's 64 keys (prefix-summed below)
      csum += __shfl_xor(csum, 1, 64);
      csum += __shfl_xor(csum, 2, 64);
      if (c == 0) v_pre[((size_t)head * (CS >> 6) + (P0 >> 6) + blockIdx.x) * D + 64 * dc + d] = csum;
      __syncthreads();                                 // s_v is rewritten by the next slab
    }
  }
  }   // slabs
}

// head_dim > 64: v_prefix[kv][kb][d] <- sum over blocks 0 .. kb of the per-block column sums (the core kernel needs sum_t v[t][d] over
// the keys it processed; with one extra all-ones MFMA per block it would also need D / 4 more accumulator registers per lane)
__global__ void __launch_bounds__(256) attention_vprefix_kernel(int32_t* __restrict__ v_prefix, int first, int nblk, int head_blocks, int D) {
  const int d = blockIdx.y * 256 + threadIdx.x;
  if (d >= D) return;
  int32_t* p = v_prefix + ((size_t)blockIdx.z * gridDim.x + blockIdx.x) * head_blocks * D + d;   // [batch][kv_heads][blocks][D]
  int run = first > 0 ? p[(size_t)(first - 1) * D] : 0;   // blocks before `first` already hold their prefix sums (cache continuation)
  for (int kb0 = first; kb0 < first + nblk; kb0 += 32) {  // 32 independent loads in flight, then the scan in registers
    int v[32];
#pragma unroll
    for (int i = 0; i < 32; ++i) v[i] = kb0 + i < first + nblk ? p[(size_t)(kb0 + i) * D] : 0;
#pragma unroll
    for (int i = 0; i < 32; ++i) {
      run += v[i];
      if (kb0 + i < first + nblk) p[(size_t)(kb0 + i) * D] = run;
    }
  }
}

// ---- the attention kernel ---------------------------------------------------------------------------------------------------------
// Per score element the VALU chain is what bounds this kernel (2 sweeps x ~10 instructions; the MFMAs run beside it), so the two
// quantizers are evaluated in the "magic number" form: f = fma(float(ti), beta, o + 1.5 * 2^23) IS round-to-nearest-even of
// ti * beta + o in the low mantissa bits; the clamp is one v_med3_f32 against [magic + qmin, magic + qmax]; differences of two such
// values are exact integers; the 16-bit probability index is read from the mantissa bytes with v_perm_b32.
constexpr float kMagic = 12582912.0f;             // 1.5 * 2^23
constexpr float kLog2e = 1.4426950408889634f;

// MQ_ATT_ABL (what-if builds through tools/build.py tags: WRONG results, timing only): 1 = no sweep 2, 2 = no sweep 1, 3 = no v_exp,
// 4 = no workgroup barrier in the f16 ring, 5 = no clamp (v_med3) in the score chain, 6 = no K fragment reads in sweep 1, 7 = no DMA requests
#ifndef MQ_ATT_ABL
#define MQ_ATT_ABL 0
#endif
__device__ __forceinline__ float fast_exp2(float x) {
#if MQ_ATT_ABL == 3
  return x;
#else
  return __builtin_amdgcn_exp2f(x);
#endif
}

// D = 64: the tuned kernel (three waves per SIMD).  D = 256 (Gemma: 8 heads / 1 KV head): four MFMA k-steps per score tile, 16 output
// d-tiles (128 accumulator registers: one wave per SIMD), v tiles loaded per d-tile, and sum_t v[t][d] from the prep kernel's prefix
// sums instead of an all-ones MFMA.
// BIG (head_dim 64): the deep exponential cache (four blocks in the LDS, five in registers: two waves per SIMD), the production
// configuration; !BIG: two blocks in the LDS, three waves per SIMD (mq_attention_set_cache(1), A/B timing).  Prep + core at S = 2048,
// deep vs small: 77.3 vs 85.6 us (32 / 4 heads), 73.5 vs 82.8 (32 / 8), 80.6 vs 93.5 (32 / 32); identical images.
// QPREP (head_dim 64, full rotary): the workgroup prepares its own 64 query rows -- the prep kernel's arithmetic for a q part, op for
// op, with thread (wave, lane) on row 16 wave + (lane & 15), columns 16 (lane >> 4) .. + 15: exactly the 16 bytes this lane feeds the
// score MFMAs, so the q image never exists in memory and the prep launch shrinks to the k / v parts (H + 2 KV -> 2 KV workgroups per
// row block; 80 % of its work at 32 / 4 heads).
// F16 (head_dim 64, 16-bit score grid, deep cache): the score contraction on v_mfma_f32_16x16x32_f16 over fp16 images of the CENTRED
// indices (prep kernel / QPREP).  sum_d (qi - zq)(ki - zk) < 2^24 is exact in the fp32 accumulator and arrives as a float: the
// zero-point terms (an add per score), the int -> float conversion (one per score) and the accumulator initialisation (C = 0 is an
// inline constant) leave the VALU stream, which is what bounds this kernel; the matrix pipe has the room for twice the MFMAs.
// PAIR (with F16): a workgroup of EIGHT waves serves two heads of one KV group (waves 0-3 the even head's 64 query rows, waves 4-7 the odd
// head's) over ONE copy of every K / vT tile: half the DMA requests per wave (they cost a wave ~130 cycles each), a third parked block
// (one workgroup of 132 KiB per CU instead of two of 68).  mq_attention_set_pair.
template <int D, bool QK_OUT, bool BIG = false, bool QPREP = false, bool F16 = false, bool PAIR = false>
#ifndef MQ_ATT_F16_WAVES
#define MQ_ATT_F16_WAVES 2   // waves per SIMD of the f16 form (the cache depths above must fit: 512 / WAVES registers, 160 KiB / WAVES of LDS per two... workgroups)
#endif
__global__ void __launch_bounds__(PAIR ? 512 : 256) __attribute__((amdgpu_waves_per_eu(F16 ? MQ_ATT_F16_WAVES : (D == 256 ? 1 : (D == 64 && !BIG ? 3 : 2)), F16 ? MQ_ATT_F16_WAVES : (D == 256 ? 1 : (D == 64 && !BIG ? 3 : 2)))))
    attention_quant_kernel(const mq_attention_args a_in) {
  static_assert(D == 64 || D == 128 || D == 256, "head_dim 64, 128 or 256");
  const mq_attention_args a = batch_view(a_in, (int)blockIdx.z);
  static_assert(!F16 || (D == 64 && QK_OUT && BIG), "f16 score contraction: the production configuration only");
  static_assert(!PAIR || F16, "two heads per workgroup: the f16 form");
  using T_ = std::true_type;
  using F_ = std::false_type;
#ifdef MQ_ATT_STAMPS
  // timing build (tools/att_stamps.py; results are still right, `out` receives the stamps): cycles per wave spent in
  // [0] q preparation, [1] sweep-1 waits (DMA + barrier), [2] sweep-1 LDS reads + MFMA issue, [3] sweep-1 quantizer chain,
  // [4] sweep-2 waits, [5] sweep-2 rest, [6] epilogue, [7] whole wave, [8] key blocks
  unsigned long long st_[11] = {0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0};   // [9] sweep-1 DMA issue, [10] sweep-1 ds_read issue .. data back
  auto now_ = []() { return (unsigned long long)__builtin_readcyclecounter(); };
  const unsigned long long st_begin = now_();
  unsigned long long st_t = st_begin;
#define MQ_ST(i) do { const unsigned long long n_ = now_(); st_[i] += n_ - st_t; st_t = n_; } while (0)
#else
#define MQ_ST(i) do { } while (0)
#endif
  constexpr int NKS = D / 64, NDT = D / 16;
  constexpr float kInvSqrtD = D == 64 ? 0.125f : (D == 256 ? 0.0625f : 0.08838834764831845f);   // 1 / sqrt(D); D = 128: RN(1 / sqrt(128))
  const int S = a.seq, H = a.heads, KV = a.kv_heads;
  const int CS = a.cache_seq > 0 ? a.cache_seq : S, PB = a.cache_seq > 0 ? a.pos0 >> 6 : 0;     // cached key blocks in front of this chunk
  // Work per workgroup is proportional to qb + 1 (causal).  The hardware hands out workgroups in id order to whichever slot frees
  // up, so the ids run over ALL heads of the longest query block first, then the next block, ...: a longest-first list schedule.
  // With 2 resident workgroups per CU and H * S/64 = 2 * (2 * 256) of them, slots pair up (S/64 - i) with (i + 1): even finish.
  // (Per-head ordering instead measured 138 us vs the 74 us of perfectly packed wave cycles: the last heads' long blocks started late.)
  const int lane = threadIdx.x & 63;
  const int wave_s = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));   // 0 .. 3 (PAIR: 0 .. 7)
  const int wave = PAIR ? wave_s & 3 : wave_s;                      // the 16-row group of the row block this wave serves
  const int h = PAIR ? 2 * ((int)blockIdx.x % (H / 2)) + (wave_s >> 2) : (int)blockIdx.x % H, kvh = h / (H / KV);
  const int qb = S / 64 - 1 - (int)blockIdx.x / (PAIR ? H / 2 : H);
  const int srow = lane & 15, tq = lane >> 4;
  const int s_abs = qb * 64 + wave * 16 + srow;                     // this lane's query row
  const AGrid gqa = a_load_grid(a.qk_a), gqb = a_load_grid(a.qk_b), gqo = a_load_grid(a.qk_out);
  const AGrid gpa = a_load_grid(a.pv_a), gpb = a_load_grid(a.pv_b), gpo = a_load_grid(a.pv_out);
  const int zv = (int)gpb.o - 128, zp = (int)gpa.o;
  const float alpha_qk = __fmul_rn(gqa.s, gqb.s);
  // scores: QK_OUT: f = magic + index on the 16-bit grid; value = (index - o) * s / 8.   else: f = ti * alpha / 8 (the value itself)
  const float beta = QK_OUT ? alpha_qk * gqo.inv_s : alpha_qk * kInvSqrtD;
  const float fbias = QK_OUT ? gqo.o + kMagic : 0.f;
  const float flo = kMagic + gqo.qmin, fhi = kMagic + gqo.qmax;
  const float cexp = QK_OUT ? gqo.s * kInvSqrtD * kLog2e : kLog2e;  // exp(value - max) = exp2((f - fmax) * cexp)

  constexpr int kKBytes = F16 ? 8192 : 4 * NKS * 1024;               // K fragments [0, kKBytes) | vT fragments [kKBytes, ...)  (F16: 8 KiB of halves + 4 KiB)
  constexpr int kTileBytes = F16 ? 12288 : (D == 64 ? 16 : 2 * kKBytes + 256);   // D != 64: + the block's 64 key terms
  constexpr int NST = F16 ? MQ_ATT_F16_STAGES : 2;                  // ring stages of the F16 form (the D != 64 form: two buffers)
  __shared__ __attribute__((aligned(16))) char s_tile[NST][kTileBytes];
  const unsigned tile_lds = (unsigned)(uintptr_t)(__attribute__((address_space(3))) char*)&s_tile[0][0];
  const char* khbase = F16 ? reinterpret_cast<const char*>(a.k_f16) + (size_t)kvh * CS * D * 2 : nullptr;   // fragment-blocked halves, 8 KiB per key block
  // this wave's pieces of tile t into the stage at byte offset `off`: K = 8 pieces (two per wave; PAIR: one), vT = 4 (one per wave; PAIR:
  // waves 0-3)
  constexpr int kKPieces = PAIR ? 1 : 2;
  const int v_mine = PAIR ? (wave_s < 4 ? 1 : 0) : 1;
  auto req_k = [&](int t, unsigned off) {
    if constexpr (F16 && MQ_ATT_ABL != 7) {
      if constexpr (PAIR) lds_dma16(khbase + (size_t)t * 8192 + wave_s * 1024, (unsigned)(lane * 16), tile_lds + off + wave_s * 1024);
      else lds_dma16x2(khbase + (size_t)t * 8192 + wave_s * 2048, (unsigned)(lane * 16), tile_lds + off + wave_s * 2048);
    }
  };
  auto req_v = [&](int t, unsigned off) {
    if constexpr (F16 && MQ_ATT_ABL != 7) {
      if (!PAIR || wave_s < 4)
        lds_dma16(a.vt_i8 + ((size_t)kvh * (CS >> 6) + t) * D * 64 + wave_s * 1024, (unsigned)(srow * 64 + tq * 16), tile_lds + off + kKBytes + wave_s * 1024);
    }
  };
  // The ring is driven without conditionals: request j carries tile min(j, last) into stage j mod NST (past the end the last tile is
  // requested again into a stage nobody reads any more), so every wait is the same counted constant and no branch guards a request;
  // stage offsets are running scalars.  Per block that removes ~20 scalar / branch instructions of ~40 beside ~90 VALU.
  v4i qf[NKS];
  v8h qh[2];                                                        // F16: this lane's 16 centred q indices as halves (d = 16 tq .. + 15)
  int qconst = 0;                                                   // D zq zk - zk * rowsum(q)
  if constexpr (QPREP) {
    static_assert(D == 64 || !QPREP, "in-kernel q preparation: head_dim 64");
    const int col0 = 16 * tq, colp = (col0 + 32) & 63;
    float x[16], pr[16], cs[16], sn[16];
    auto load16 = [](const float* p, float (&d)[16]) {
#pragma unroll
      for (int i = 0; i < 4; ++i) {
        const float4 t = reinterpret_cast<const float4*>(p)[i];
        d[4 * i] = t.x; d[4 * i + 1] = t.y; d[4 * i + 2] = t.z; d[4 * i + 3] = t.w;
      }
    };
    // rot_dim 16 (StableLM-2: 16 of 64 dims rotate, hf_model.py:489-500): only the lanes of the first 16 dims rotate, and a dim's
    // partner (d +- 8) lies among the lane's own 16 values -- no partner load, cos / sin [S, 16] read by those lanes only
    const bool part16 = a.rot_dim == 16;
    uint4 t0 = {0, 0, 0, 0}, t1 = {0, 0, 0, 0};
    if (a.qkv_idx) {                                                // the fused q|k|v GEMM's uint8 indices
      const uint8_t* ip = a.qkv_idx + ((size_t)s_abs * (H + 2 * KV) + h) * D;
      t0 = *reinterpret_cast<const uint4*>(ip + col0);
      if (!part16) t1 = *reinterpret_cast<const uint4*>(ip + colp);
    } else {
      const float* src = a.q + (size_t)s_abs * H * D + (size_t)h * D;
      load16(src + col0, x);
      if (!part16) load16(src + colp, pr);
    }
    if (!part16) {
      load16(a.cos + (size_t)s_abs * D + col0, cs);
      load16(a.sin + (size_t)s_abs * D + col0, sn);
    } else if (tq == 0) {
      load16(a.cos + (size_t)s_abs * 16, cs);
      load16(a.sin + (size_t)s_abs * 16, sn);
    }
    // the ring's first K requests: behind the q LOADS (an asm request fences the compiler's loads), in front of the q arithmetic
    if constexpr (F16 && MQ_ATT_ABL != 7 && MQ_ATT_ABL != 2) {
#pragma unroll
      for (int i = 0; i < NST - 1; ++i) {
        const int t = i < PB + qb ? i : PB + qb;
        req_k(t, i * kTileBytes);
      }
    }
    if (a.qkv_idx) {                                                // ... dequantised as that linear's fp32 output would read
      const AGrid gin = a_load_grid(a.q_in);
      const unsigned w0[4] = {t0.x, t0.y, t0.z, t0.w}, w1[4] = {t1.x, t1.y, t1.z, t1.w};
#pragma unroll
      for (int i = 0; i < 16; ++i) {
        x[i] = __fmul_rn(__fsub_rn((float)((w0[i >> 2] >> (8 * (i & 3))) & 0xffu), gin.o), gin.s);
        pr[i] = __fmul_rn(__fsub_rn((float)((w1[i >> 2] >> (8 * (i & 3))) & 0xffu), gin.o), gin.s);
      }
    }
    const float sign = col0 < D / 2 ? -1.f : 1.f;
    float ya[16];                                                   // the values the input quantizer sees
    if (!part16) {
#pragma unroll
      for (int i = 0; i < 16; ++i) ya[i] = __fadd_rn(__fmul_rn(x[i], cs[i]), __fmul_rn(sign * pr[i], sn[i]));
    } else {
#pragma unroll
      for (int i = 0; i < 16; ++i) ya[i] = x[i];
      if (tq == 0) {
#pragma unroll
        for (int i = 0; i < 16; ++i)                                // x * cos + rot(x) * sin, rot(x)[d] = d < 8 ? -x[d + 8] : x[d - 8]
          ya[i] = __fadd_rn(__fmul_rn(x[i], cs[i]), __fmul_rn((i < 8 ? -1.f : 1.f) * x[i < 8 ? i + 8 : i - 8], sn[i]));
      }
    }
    uint32_t usum = 0;
    unsigned hw[8];
#pragma unroll
    for (int d4 = 0; d4 < 4; ++d4) {
      float qi[4];
#pragma unroll
      for (int e = 0; e < 4; ++e) qi[e] = image_idxf(ya[4 * d4 + e], gqa.s, gqa.inv_s, gqa.o, gqa.qmin, gqa.qmax);
      if constexpr (F16) {
        hw[2 * d4] = pack_h2(__fsub_rn(qi[0], gqa.o), __fsub_rn(qi[1], gqa.o));
        hw[2 * d4 + 1] = pack_h2(__fsub_rn(qi[2], gqa.o), __fsub_rn(qi[3], gqa.o));
      } else {
        qf[0][d4] = (int)image_pack4(qi[0], qi[1], qi[2], qi[3], usum);
      }
    }
    if constexpr (F16) {
      if (!(gqa.o >= -256.f && gqa.o <= 511.f)) {                  // (see the prep kernel: a q grid the f16 contraction cannot hold exactly)
#pragma unroll
        for (int i = 0; i < 8; ++i) hw[i] = 0x7e007e00u;
      }
      qh[0] = __builtin_bit_cast(v8h, v4i{(int)hw[0], (int)hw[1], (int)hw[2], (int)hw[3]});
      qh[1] = __builtin_bit_cast(v8h, v4i{(int)hw[4], (int)hw[5], (int)hw[6], (int)hw[7]});
    } else {
      int sum = (int)usum - 128 * 16;
      sum += __shfl_xor(sum, 16, 64);
      sum += __shfl_xor(sum, 32, 64);
      const int zq = (int)gqa.o - 128, zk = (int)gqb.o - 128;
      qconst = D * zq * zk - zk * sum;
    }
  } else if constexpr (F16) {
    const v4i* qp = reinterpret_cast<const v4i*>(reinterpret_cast<const _Float16*>(a.q_f16) + ((size_t)h * S + (size_t)qb * 64 + wave * 16 + srow) * D + tq * 16);
    qh[0] = __builtin_bit_cast(v8h, qp[0]);
    qh[1] = __builtin_bit_cast(v8h, qp[1]);
  if constexpr (F16 && MQ_ATT_ABL != 7 && MQ_ATT_ABL != 2) {         // the ring's first K requests: behind the q LOADS (an asm request is a fence for the compiler's loads), in front of the q arithmetic
#pragma unroll
    for (int i = 0; i < NST - 1; ++i) {
      const int t = i < PB + qb ? i : PB + qb;
      req_k(t, i * kTileBytes);
    }
  }
  } else {
    const int8_t* qbase = a.q_i8 + ((size_t)h * S + (size_t)qb * 64 + wave * 16) * D;
#pragma unroll
    for (int ks = 0; ks < NKS; ++ks) qf[ks] = *reinterpret_cast<const v4i*>(qbase + srow * D + ks * 64 + tq * 16);
    qconst = a.q_rowsum[(size_t)h * S + s_abs];                     // from the prep kernel
  }
  const int8_t* kbase = a.k_i8 + (size_t)kvh * CS * D;
  const int* kterm = a.k_rowsum + (size_t)kvh * CS;                 // -zq * rowsum(k)
  const int nkb = PB + qb + 1;                                      // key blocks 0 .. PB + qb (causal; PB cached blocks in front)
  const int kdiag = PB + qb;                                        // the block that holds the diagonal
  const int s_key = PB * 64 + s_abs;                                // this row's absolute position (the mask compares keys against it)
  const v4i cinit = {qconst, qconst, qconst, qconst};

  struct KTile {
    v4i kf[4][NKS];
    int4 kt[4];
  };
  auto load_k = [&](int kb, KTile& t) {
    const int8_t* kp = kbase + (size_t)kb * 64 * D;
#pragma unroll
    for (int j = 0; j < 4; ++j) {
#pragma unroll
      for (int ks = 0; ks < NKS; ++ks) t.kf[j][ks] = *reinterpret_cast<const v4i*>(kp + (16 * j + srow) * D + ks * 64 + tq * 16);
      t.kt[j] = *reinterpret_cast<const int4*>(kterm + kb * 64 + 16 * j + 4 * tq);
    }
  };
  // D = 256: a key block's K tile (16 KiB) and vT tile (16 KiB) do not fit the 32-KiB L1 next to each other, so four waves loading
  // them privately miss four times (measured: 183 us at 8 heads x S = 2048, L2-bound).  Instead the workgroup stages ONE copy per
  // block in the LDS by LDS-DMA, fragment-blocked: fragment f (K: f = 4 j + ks, 16 keys x 64 d;  vT: f = dt, 16 d x 64 keys) is the
  // 1-KiB block whose lane l holds exactly the 16 bytes lane l feeds the MFMA -- conflict-free ds_read_b128, and the DMA's lane-linear
  // destination is that layout when every lane sources its own fragment bytes.  Two buffers, one barrier per block.
  const int8_t* vbase = a.vt_i8 + (size_t)kvh * (CS >> 6) * D * 64;
  // (asm LDS-DMA, lds_dma16 above: with the builtin, hipcc puts s_waitcnt vmcnt(0) in front of every ds_read that follows a request,
  // i.e. block kb's reads waited for block kb + 1's transfer -- no overlap at all.  The key terms ride along as a 256-byte piece,
  // so no compiler-counted vector load sits between the requests and the reads either.)
  auto dma_block = [&](int kb, int buf, bool with_v) {
    if constexpr (D != 64) {
      const int8_t* kp = kbase + (size_t)kb * 64 * D + (size_t)(16 * wave_s) * D;
      const int8_t* vp = vbase + (size_t)kb * D * 64;
      const unsigned base = tile_lds + buf * kTileBytes;
#pragma unroll
      for (int u = 0; u < NKS; ++u)                                  // this wave's K fragments (j = wave, ks = u) ...
        lds_dma16(kp + u * 64, (unsigned)(srow * D + tq * 16), base + (NKS * wave_s + u) * 1024);
      if (with_v) {
#pragma unroll
        for (int u = 0; u < NDT / 4; ++u) {                          // ... and its vT fragments (dt = f)
          const int f = (NDT / 4) * wave_s + u;
          lds_dma16(vp + (size_t)(16 * f) * 64, (unsigned)(srow * 64 + tq * 16), base + kKBytes + f * 1024);
        }
      }
      if (wave_s == 0) lds_dma4(kterm + kb * 64, (unsigned)(lane * 4), base + 2 * kKBytes);
    }
  };
  auto block_ready = []() { asm volatile("s_waitcnt vmcnt(0)\n\ts_barrier" ::: "memory"); };
  // (one wave per SIMD at head_dim 256: a compiler-placed ds_read in front of every MFMA cost an LDS round trip each -- 19 / 27
  // s_waitcnt per block.  All 4 NKS fragments of the block are requested at once by asm reads the compiler does not wait for, ONE
  // counted wait names them.)
  auto int_scores_lds = [&](int kb, int buf, int (&ti)[16]) {
    constexpr int NF = 4 * NKS;
    v4i kf[NF];
    if constexpr (D != 64) lds_read_frags<0>(kf, tile_lds + buf * kTileBytes + lane * 16, std::make_integer_sequence<int, NF>{});
    int4 kt[4];
#pragma unroll
    for (int j = 0; j < 4; ++j) kt[j] = *reinterpret_cast<const int4*>(s_tile[buf] + 2 * kKBytes + (16 * j + 4 * tq) * 4);
    if constexpr (NF == 16 || NF == 8) {
      if constexpr (NF == 16) lds_fragments_wait(kf);
      else lds_fragments_wait(reinterpret_cast<v4i(&)[8]>(kf));
    }
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      v4i acc = cinit;
#pragma unroll
      for (int ks = 0; ks < NKS; ++ks) acc = __builtin_amdgcn_mfma_i32_16x16x64_i8(kf[NKS * j + ks], qf[ks], acc, 0, 0, 0);
      ti[4 * j] = acc[0] + kt[j].x; ti[4 * j + 1] = acc[1] + kt[j].y; ti[4 * j + 2] = acc[2] + kt[j].z; ti[4 * j + 3] = acc[3] + kt[j].w;
    }
  };
  // integer scores of this lane's row against keys t = 64 kb + 16 j + 4 tq + e: sum_d (qi - zq)(ki - zk), exact (< 2^24).  After this
  // the tile's registers are dead: the caller requests the NEXT block into the same registers, and the loads land under the ~1 000
  // cycles of VALU work that follow (single-buffered tiles: 3 waves per SIMD fit, and they hide what is left).
  auto int_scores = [&](const KTile& t, int (&ti)[16]) {
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      v4i acc = cinit;
#pragma unroll
      for (int ks = 0; ks < NKS; ++ks) acc = __builtin_amdgcn_mfma_i32_16x16x64_i8(t.kf[j][ks], qf[ks], acc, 0, 0, 0);
      ti[4 * j] = acc[0] + t.kt[j].x; ti[4 * j + 1] = acc[1] + t.kt[j].y; ti[4 * j + 2] = acc[2] + t.kt[j].z; ti[4 * j + 3] = acc[3] + t.kt[j].w;
    }
  };
  // F16: the same sums as floats, from two k-halves of fp16 products (C = 0: an inline constant, nothing to initialise); the K
  // fragments come from the workgroup's LDS copy of the block (one fetch per workgroup instead of one per wave: with the VALU
  // chain trimmed, four private copies per block through the L1 were what bounded the loop)
  auto f_scores_lds = [&](int buf, float (&tf)[16]) {
    const char* tb = s_tile[buf] + lane * 16;
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      v4f acc = __builtin_amdgcn_mfma_f32_16x16x32_f16(__builtin_bit_cast(v8h, *reinterpret_cast<const v4i*>(tb + (2 * j) * 1024)), qh[0],
                                                       v4f{0.f, 0.f, 0.f, 0.f}, 0, 0, 0);
      acc = __builtin_amdgcn_mfma_f32_16x16x32_f16(__builtin_bit_cast(v8h, *reinterpret_cast<const v4i*>(tb + (2 * j + 1) * 1024)), qh[1], acc, 0, 0, 0);
      tf[4 * j] = acc[0]; tf[4 * j + 1] = acc[1]; tf[4 * j + 2] = acc[2]; tf[4 * j + 3] = acc[3];
    }
  };
  // this wave's share of block kb into buffer buf: K fragments 2 wave, 2 wave + 1 (consecutive KiB of the fragment-blocked image) and
  // vT fragment dt = wave (rows d = 16 wave .. + 15 of the [d][64] tile, lane (srow, tq) sourcing its own 16 bytes)
  // (the vT piece first: wait_dma counts on that order)
  // (scalar bases, one 32-bit lane offset each: no vector address arithmetic per request)
  const unsigned voff_k = lane * 16, voff_v = srow * 64 + tq * 16;
  auto dma_f16 = [&](int kb, int buf, bool with_k, bool with_v) {
    if constexpr (F16 && MQ_ATT_ABL != 7) {
      if (with_v) req_v(kb, buf * kTileBytes);
      if (with_k) req_k(kb, buf * kTileBytes);
    }
  };
  // block kb's pieces have landed for every wave once each wave has at most `pending` later DMA instructions in flight (they return
  // in order) and the workgroup has met; behind the barrier the stage read one block ago is free
  auto wait_dma = [](int pending) {
    if (pending >= 3) asm volatile("s_waitcnt vmcnt(3)" ::: "memory");
    else if (pending == 2) asm volatile("s_waitcnt vmcnt(2)" ::: "memory");
    else if (pending == 1) asm volatile("s_waitcnt vmcnt(1)" ::: "memory");
    else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
#if MQ_ATT_ABL != 4
    asm volatile("s_barrier" ::: "memory");
#endif
  };
  auto ring_step = [&](unsigned& off) {
    off += kTileBytes;
    if (off == NST * kTileBytes) off = 0;
  };
  auto ring_wait = [&](auto pieces) {                              // the tile about to be read has landed once at most (NST - 2) later tiles are in flight
    constexpr int n = (NST - 2) * decltype(pieces)::value;
    if constexpr (n == 0) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    else if constexpr (n == 1) asm volatile("s_waitcnt vmcnt(1)" ::: "memory");
    else if constexpr (n == 2) asm volatile("s_waitcnt vmcnt(2)" ::: "memory");
    else asm volatile("s_waitcnt vmcnt(3)" ::: "memory");
#if MQ_ATT_ABL != 4
    asm volatile("s_barrier" ::: "memory");
#endif
  };
  using Tile = KTile;
  using SC = int;
  auto load_t = [&](int kb, Tile& t) { load_k(kb, t); };
  auto scores_t = [&](const Tile& t, SC (&ti)[16]) { int_scores(t, ti); };
  // -> f: the score on its 16-bit grid in magic-number form (QK_OUT), or the score value itself.  The causal mask only ever cuts the
  // LAST key block of a row block (kb == kdiag == nkb - 1): `diag` is a compile-time tag, so the other blocks carry no select.
  auto grid_scores = [&](const auto (&ti)[16], auto diag, int kb, float (&f)[16]) {
    if constexpr (F16 && MQ_ATT_PK) {                                // two scores per v_pk_fma_f32 (the same IEEE operation per half)
#pragma unroll
      for (int i = 0; i < 16; i += 2) {
        const v2f v = __builtin_elementwise_fma(v2f{(float)ti[i], (float)ti[i + 1]}, splat2(beta), splat2(fbias));
        f[i] = __builtin_amdgcn_fmed3f(v.x, flo, fhi);
        f[i + 1] = __builtin_amdgcn_fmed3f(v.y, flo, fhi);
      }
    } else {
#pragma unroll
    for (int i = 0; i < 16; ++i) {
      float v = __builtin_fmaf((float)ti[i], beta, fbias);
#if MQ_ATT_ABL != 5
      if (QK_OUT) v = __builtin_amdgcn_fmed3f(v, flo, fhi);
#endif
      f[i] = v;
    }
    }
    if constexpr (decltype(diag)::value) {
#pragma unroll
      for (int i = 0; i < 16; ++i) {
        const int t_abs = kb * 64 + 16 * (i >> 2) + 4 * tq + (i & 3);
        f[i] = t_abs <= s_key ? f[i] : -INFINITY;
      }
    }
  };

  // ---- sweep 1: row max and sum of exp ----------------------------------------------------------------------------------------
  // exp(value - max) = exp2(f * cexp - R) with ONE fma per element: R = fl(fmax * cexp) is a per-row constant, so its rounding error
  // shifts every exponent of the row alike and cancels in e / l (softmax is shift invariant); R - R' below is exact (Sterbenz).
  float m = -INFINITY, l = 0.f, R = -INFINITY;
  auto sweep1 = [&](const auto (&ti)[16], int kb, auto diag) {
    float f[16];
    grid_scores(ti, diag, kb, f);
    float bm = fmaxf(fmaxf(fmaxf(f[0], f[1]), fmaxf(f[2], f[3])), fmaxf(fmaxf(f[4], f[5]), fmaxf(f[6], f[7])));
    bm = fmaxf(bm, fmaxf(fmaxf(fmaxf(f[8], f[9]), fmaxf(f[10], f[11])), fmaxf(fmaxf(f[12], f[13]), fmaxf(f[14], f[15]))));
    bm = fmaxf(bm, __shfl_xor(bm, 16, 64));
    bm = fmaxf(bm, __shfl_xor(bm, 32, 64));
    const float mn = fmaxf(m, bm);                                   // finite: key 0 is never masked
    const float Rn = mn * cexp;
    float bs = 0.f;
#pragma unroll
    for (int i = 0; i < 16; ++i) bs += fast_exp2(__builtin_fmaf(f[i], cexp, -Rn));
    bs += __shfl_xor(bs, 16, 64);
    bs += __shfl_xor(bs, 32, 64);
    l = l * fast_exp2(R - Rn) + bs;                                  // first block: exp2(-inf) = 0
    m = mn;
    R = Rn;
  };
  // With a score grid whose whole span is < 2^96 in exp2 units (any sane 16-bit range: 65 535 steps x cexp ~ 20), the grid's TOP can
  // stand in for the row maximum: no running max, no rescale, no dependency between blocks -- exp2((f - top) c) can neither overflow
  // nor flush the row's largest term, and the common factor cancels in e / l like the rounding of R does.
  const bool fixed_ref = QK_OUT && (fhi - flo) * cexp < 96.f;
  // With the grid top as reference exponent the sweep-1 exponentials ARE the sweep-2 ones (no rescale in between): those of a row
  // block's LAST kEC key blocks are parked in the LDS (a thread reads back only what it wrote: no barrier), and sweep 2 takes them from
  // there instead of recomputing scores, grid and exp2 -- ~70 % of a block's sweep-2 instructions for min(kEC, nkb) / nkb of the blocks.
  constexpr int kEC = D == 64 ? (F16 ? (PAIR ? 3 : MQ_ATT_ECACHE_F16) : (BIG ? MQ_ATT_ECACHE : 2)) : 0;
  static_assert(!PAIR || NST == 3, "two heads per workgroup: a three-stage ring");
  __shared__ float4 s_e[kEC > 0 ? kEC : 1][4][kEC > 0 ? (PAIR ? 512 : 256) : 1];     // [block][quad of keys][thread]: whole-dword-quad rows, ds_*_b128 without bank conflicts
  constexpr int kER = D == 64 && BIG ? (F16 ? MQ_ATT_EREGS_F16 : MQ_ATT_EREGS) : 0;   // ... and those of the kER blocks in front of them in registers
  const int n_lds0 = nkb - kEC > 0 ? nkb - kEC : 0;                 // first block parked in the LDS
  const int n_reg0 = n_lds0 - kER > 0 ? n_lds0 - kER : 0;           // first block parked in registers (blocks before it are recomputed)
  float ereg[kER > 0 ? kER : 1][16];
  auto exps = [&](const auto (&ti)[16], int kb, float (&ex)[16], auto diag) {
    float f[16];
    grid_scores(ti, diag, kb, f);
    float bs = 0.f;
    if constexpr (F16 && MQ_ATT_PK) {
#pragma unroll
      for (int i = 0; i < 16; i += 2) {
        const v2f t = __builtin_elementwise_fma(v2f{f[i], f[i + 1]}, splat2(cexp), splat2(-R));
        ex[i] = fast_exp2(t.x);
        ex[i + 1] = fast_exp2(t.y);
        bs += ex[i];
        bs += ex[i + 1];
      }
    } else {
#pragma unroll
    for (int i = 0; i < 16; ++i) {
      ex[i] = fast_exp2(__builtin_fmaf(f[i], cexp, -R));
      bs += ex[i];
    }
    }
    l += bs;
  };
  auto sweep1_fixed = [&](const auto (&ti)[16], int kb, auto diag, auto park) {
    float ex[16];
    exps(ti, kb, ex, diag);
    if constexpr (kEC > 0 && decltype(park)::value) {
#pragma unroll
      for (int i = 0; i < 4; ++i) s_e[kb - n_lds0][i][threadIdx.x] = make_float4(ex[4 * i], ex[4 * i + 1], ex[4 * i + 2], ex[4 * i + 3]);
    }
  };
  MQ_ST(0);
  if constexpr (F16 && MQ_ATT_ABL == 2) {
    R = fhi * cexp;
    l = 1.f;
  } else if constexpr (F16) {
    auto next_scores = [&](int kb, float (&ti)[16]) {              // (the form with a running row maximum: guarded requests, counted per block)
      const int ahead = nkb - 1 - kb < NST - 2 ? nkb - 1 - kb : NST - 2;
      wait_dma(kKPieces * ahead);
      if (kb + NST - 1 < nkb) dma_f16(kb + NST - 1, (kb + NST - 1) % NST, true, false);
      f_scores_lds(kb % NST, ti);
    };
    if (fixed_ref) {
      R = fhi * cexp;
      // Software pipeline: a wave issues in order, and per block the wait at the barrier, the DMA request, the LDS round trip of the
      // fragments and the MFMAs cost it twice the cycles of the quantizer chain (tools/att_stamps.py).  So block kb + 1's fragment
      // reads and the ring's next request are issued IN FRONT of block kb's chain, and its MFMAs behind it: the LDS latency lies under
      // the chain, the MFMA latency under the next block's barrier / request.
      float sc[16];
      v4i fr[8];
      unsigned rd_off = 0, wr_off = (NST - 1) * kTileBytes;          // the stage the next fetch reads / the next request fills
      int req = NST - 1 < nkb - 1 ? NST - 1 : nkb - 1;               // the tile the next request carries
      auto fetch = [&]() {                                           // the next block has landed -> its fragments requested, the next DMA request issued
        MQ_ST(2);
        ring_wait(std::integral_constant<int, kKPieces>{});
        MQ_ST(1);
        const unsigned tb = tile_lds + rd_off + lane * 16;
#if MQ_ATT_ABL == 6
#pragma unroll
        for (int f = 0; f < 8; ++f) fr[f] = v4i{(int)tb, kb, f, 1};
#else
        lds_read_frag<0>(fr[0], tb); lds_read_frag<1024>(fr[1], tb); lds_read_frag<2048>(fr[2], tb); lds_read_frag<3072>(fr[3], tb);
        lds_read_frag<4096>(fr[4], tb); lds_read_frag<5120>(fr[5], tb); lds_read_frag<6144>(fr[6], tb); lds_read_frag<7168>(fr[7], tb);
#endif
        req_k(req, wr_off);
        wr_off = rd_off;                                             // (the stage just read is the one after next to fill: wr = rd + (NST - 1) stages)
        ring_step(rd_off);
        req = req + 1 < nkb - 1 ? req + 1 : nkb - 1;
        MQ_ST(9);
      };
      auto contract = [&](float (&o)[16]) {
#pragma unroll
        for (int j = 0; j < 4; ++j) {
          v4f acc = __builtin_amdgcn_mfma_f32_16x16x32_f16(__builtin_bit_cast(v8h, fr[2 * j]), qh[0], v4f{0.f, 0.f, 0.f, 0.f}, 0, 0, 0);
          acc = __builtin_amdgcn_mfma_f32_16x16x32_f16(__builtin_bit_cast(v8h, fr[2 * j + 1]), qh[1], acc, 0, 0, 0);
          o[4 * j] = acc[0]; o[4 * j + 1] = acc[1]; o[4 * j + 2] = acc[2]; o[4 * j + 3] = acc[3];
        }
      };
      // one block: the next block's fragments are requested, the ring's next DMA request goes out (its issue time covers the LDS
      // round trip), then the next block's eight MFMAs are spread through this block's chain (one per ~10 VALU instructions: their
      // issue and dependency latency, ~500 cycles when issued back to back beside the other wave's, disappears under the chain)
      auto step = [&](int kb, auto last, auto&& chain) {
        if constexpr (!decltype(last)::value) {
          float nx[16];
          fetch();
          lds_fragments_wait(fr);
          tie16(sc);
          contract(nx);
          chain();
#pragma unroll
          for (int g = 0; g < 8; ++g) {
            __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
            __builtin_amdgcn_sched_group_barrier(0x002, 10, 0);
          }
          MQ_ST(3);
#pragma unroll
          for (int i = 0; i < 16; ++i) sc[i] = nx[i];
        } else {
          tie16(sc);
          chain();
          MQ_ST(3);
        }
      };
      fetch();
      lds_fragments_wait(fr);
      contract(sc);
      for (int kb = 0; kb < n_reg0; ++kb) step(kb, F_{}, [&]() { sweep1_fixed(sc, kb, F_{}, F_{}); });
#pragma unroll
      for (int u = 0; u < kER; ++u) {                                // static register indices: unrolled; the guard is workgroup-uniform
        const int kb = n_reg0 + u;
        if (kb < n_lds0) step(kb, F_{}, [&]() { exps(sc, kb, ereg[u], F_{}); });
      }
      for (int kb = n_lds0; kb < nkb - 1; ++kb) step(kb, F_{}, [&]() { sweep1_fixed(sc, kb, F_{}, T_{}); });
      step(nkb - 1, T_{}, [&]() { sweep1_fixed(sc, nkb - 1, T_{}, T_{}); });   // the diagonal block: always the last, always parked (kEC >= 1)
      l += __shfl_xor(l, 16, 64);
      l += __shfl_xor(l, 32, 64);
    } else {
      asm volatile("s_waitcnt vmcnt(0)" ::: "memory");               // (restart the ring with guarded requests)
      __syncthreads();
#pragma unroll
      for (int i = 0; i < NST - 1; ++i)
        if (i < nkb) dma_f16(i, i, true, false);
      for (int kb = 0; kb < nkb; ++kb) {
        float ti[16];
        next_scores(kb, ti);
        if (kb == kdiag) sweep1(ti, kb, T_{});
        else sweep1(ti, kb, F_{});
      }
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");                 // the ring's trailing requests
    __syncthreads();                                               // sweep 2 starts over in buffer 0
  } else if constexpr (D != 64) {
    if (fixed_ref) R = fhi * cexp;
    dma_block(0, 0, false);
    for (int kb = 0; kb < nkb; ++kb) {
      block_ready();                                               // block kb landed (everyone's pieces); buffer (kb + 1) & 1 is free
      if (kb + 1 < nkb) dma_block(kb + 1, (kb + 1) & 1, false);
      int ti[16];
      int_scores_lds(kb, kb & 1, ti);
      if (fixed_ref) {
        if (kb == kdiag) sweep1_fixed(ti, kb, T_{}, F_{});
        else sweep1_fixed(ti, kb, F_{}, F_{});
      } else {
        if (kb == kdiag) sweep1(ti, kb, T_{});
        else sweep1(ti, kb, F_{});
      }
    }
    if (fixed_ref) {
      l += __shfl_xor(l, 16, 64);
      l += __shfl_xor(l, 32, 64);
    }
    __syncthreads();                                               // sweep 2 starts over in buffer 0
  } else {
    Tile t;
    load_t(0, t);
    if (fixed_ref) {
      R = fhi * cexp;
      for (int kb = 0; kb < n_reg0; ++kb) {                          // n_reg0 <= nkb - kEC: a block follows
        SC ti[16];
        scores_t(t, ti);
        load_t(kb + 1, t);
        sweep1_fixed(ti, kb, F_{}, F_{});
      }
      if constexpr (kER > 0) {
#pragma unroll
        for (int u = 0; u < kER; ++u) {                              // static register indices: the loop is unrolled, the guard wave-uniform
          const int kb = n_reg0 + u;
          if (kb < n_lds0) {
            SC ti[16];
            scores_t(t, ti);
            load_t(kb + 1, t);
            exps(ti, kb, ereg[u], F_{});
          }
        }
      }
      for (int kb = n_lds0; kb < nkb - 1; ++kb) {
        SC ti[16];
        scores_t(t, ti);
        load_t(kb + 1, t);
        sweep1_fixed(ti, kb, F_{}, T_{});
      }
      {                                                              // the diagonal block: always the last, always parked (kEC >= 1)
        SC ti[16];
        scores_t(t, ti);
        sweep1_fixed(ti, nkb - 1, T_{}, T_{});
      }
      l += __shfl_xor(l, 16, 64);
      l += __shfl_xor(l, 32, 64);
    } else {
      for (int kb = 0; kb < nkb; ++kb) {
        SC ti[16];
        scores_t(t, ti);
        if (kb + 1 < nkb) load_t(kb + 1, t);
        if (kb == kdiag) sweep1(ti, kb, T_{});
        else sweep1(ti, kb, F_{});
      }
    }
  }
  // p index = clamp(rint((e / l) / s_p) + z_p): g = fma(e, 1 / (l s_p), z_p + magic), index = low mantissa bits of med3(g, ...)
  MQ_ST(3);
  const float rp = __fdiv_rn(gpa.inv_s, l);
  const float pbias = gpa.o + kMagic, plo = kMagic + gpa.qmin, phi = kMagic + gpa.qmax;

  // ---- sweep 2: probabilities on their 16-bit grid, integer p.v ------------------------------------------------------------------
  v4i acc_hi[NDT], acc_lo[NDT], acc_v[D == 64 ? NDT : 1];          // acc_v (D = 64): column sums of the stored v over the processed keys -- one
#pragma unroll                                                      // more MFMA against an all-ones tile (the MFMA pipe idles, the VALU does not)
  for (int dt = 0; dt < NDT; ++dt) acc_hi[dt] = acc_lo[dt] = v4i{0, 0, 0, 0};
#pragma unroll
  for (int dt = 0; dt < (D == 64 ? NDT : 1); ++dt) acc_v[dt] = v4i{0, 0, 0, 0};
  const v4i ones = {0x01010101, 0x01010101, 0x01010101, 0x01010101};
  unsigned psum_hi = 0, psum_lo = 0;                                // sums of the unsigned high / low bytes (this lane's share)
  v4i acc_ph = {0, 0, 0, 0}, acc_pl = {0, 0, 0, 0};                 // F16: the row's sums of the SIGNED high / low bytes, every row of D alike
  struct VTile {
    v4i vf[4];
  };
  auto load_v = [&](int kb, VTile& t) {
    const int8_t* vt = vbase + (size_t)kb * D * 64;
#pragma unroll
    for (int dt = 0; dt < 4; ++dt) t.vf[dt] = *reinterpret_cast<const v4i*>(vt + (16 * dt + srow) * 64 + tq * 16);   // rows d, key-permuted
  };
  // The clamp of the probability index is dead when the grid holds [0, 1]: e / l <= 1, so the index stays within
  // [z_p, rint(1 / s_p) + z_p]; `pcl` (a compile-time tag, chosen per launch from the grid: p_clamp below) keeps or drops the v_med3.
  auto probs_from = [&](const float (&exv)[16], v4i& pf_hi, v4i& pf_lo, auto pcl) {
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      unsigned b[4];
#pragma unroll
      for (int e = 0; e < 4; ++e) {
        float g = __builtin_fmaf(exv[4 * j + e], rp, pbias);
        if constexpr (decltype(pcl)::value) g = __builtin_amdgcn_fmed3f(g, plo, phi);
        b[e] = __float_as_uint(g);
      }
      const unsigned p01 = __builtin_amdgcn_perm(b[1], b[0], 0x05040100u), p23 = __builtin_amdgcn_perm(b[3], b[2], 0x05040100u);
      const unsigned lo = __builtin_amdgcn_perm(p23, p01, 0x06040200u), hi = __builtin_amdgcn_perm(p23, p01, 0x07050301u);
      if constexpr (!F16) {                                          // (F16: the byte sums come from two more MFMAs against all-ones, pv_lds)
        psum_lo = __builtin_amdgcn_sad_u8(lo, 0u, psum_lo);
        psum_hi = __builtin_amdgcn_sad_u8(hi, 0u, psum_hi);
      }
      pf_lo[j] = (int)(lo ^ 0x80808080u);
      pf_hi[j] = (int)(hi ^ 0x80808080u);
    }
  };
  auto probs = [&](const auto (&ti)[16], int kb, v4i& pf_hi, v4i& pf_lo, auto diag, auto pcl) {
    float f[16];
    grid_scores(ti, diag, kb, f);
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      unsigned b[4];
#pragma unroll
      for (int e = 0; e < 4; ++e) {
        const float ex = fast_exp2(__builtin_fmaf(f[4 * j + e], cexp, -R));     // masked keys: exp2(-inf) = 0 -> index z_p -> (z_p - z_p) = 0
        float g = __builtin_fmaf(ex, rp, pbias);
        if constexpr (decltype(pcl)::value) g = __builtin_amdgcn_fmed3f(g, plo, phi);
        b[e] = __float_as_uint(g);
      }
      const unsigned p01 = __builtin_amdgcn_perm(b[1], b[0], 0x05040100u), p23 = __builtin_amdgcn_perm(b[3], b[2], 0x05040100u);
      const unsigned lo = __builtin_amdgcn_perm(p23, p01, 0x06040200u), hi = __builtin_amdgcn_perm(p23, p01, 0x07050301u);
      if constexpr (!F16) {                                          // (F16: the byte sums come from two more MFMAs against all-ones, pv_lds)
        psum_lo = __builtin_amdgcn_sad_u8(lo, 0u, psum_lo);
        psum_hi = __builtin_amdgcn_sad_u8(hi, 0u, psum_hi);
      }
      pf_lo[j] = (int)(lo ^ 0x80808080u);
      pf_hi[j] = (int)(hi ^ 0x80808080u);
    }
  };
  if constexpr (F16 && MQ_ATT_ABL == 1) {
    acc_hi[0][0] = (int)l;
  } else if constexpr (F16) {
    const int nrec = fixed_ref ? n_reg0 : nkb;                       // blocks whose scores are recomputed
    // the ring as in sweep 1: request j = tile min(j, last), its vT piece and -- for a recomputed tile -- its two K pieces
    auto request = [&](int t, unsigned off) {
      if constexpr (MQ_ATT_ABL != 7) {
        req_v(t, off);
        if (t < nrec) req_k(t, off);
      }
    };
#pragma unroll
    for (int i = 0; i < NST - 1; ++i) request(i < nkb - 1 ? i : nkb - 1, i * kTileBytes);
    unsigned rd_off = 0, wr_off = (NST - 1) * kTileBytes;
    int req = NST - 1 < nkb - 1 ? NST - 1 : nkb - 1;
    // A block of sweep 2: behind the barrier the block's vT fragments (and, for a recomputed block, its K fragments) are requested from
    // the LDS, THEN the ring's next DMA request goes out -- its issue time (100-400 cycles for one to three pieces) covers the LDS round
    // trip -- and one counted wait names every fragment register.  `pieces`: DMA instructions of the tile requested after this one.
    v4i vf[4], kf[8];
    auto next_block = [&](auto rec, auto pieces) {                   // -> vf (and kf) hold the next block's fragments
      MQ_ST(5);
      if constexpr (PAIR) wait_dma(v_mine + (decltype(pieces)::value == 3 ? 1 : 0));   // the next tile's pieces of THIS wave (NST == 3)
      else ring_wait(pieces);
      MQ_ST(4);
      const unsigned tb = tile_lds + rd_off + lane * 16;
      lds_read_frag<kKBytes>(vf[0], tb); lds_read_frag<kKBytes + 1024>(vf[1], tb); lds_read_frag<kKBytes + 2048>(vf[2], tb); lds_read_frag<kKBytes + 3072>(vf[3], tb);
      if constexpr (decltype(rec)::value) {
        lds_read_frag<0>(kf[0], tb); lds_read_frag<1024>(kf[1], tb); lds_read_frag<2048>(kf[2], tb); lds_read_frag<3072>(kf[3], tb);
        lds_read_frag<4096>(kf[4], tb); lds_read_frag<5120>(kf[5], tb); lds_read_frag<6144>(kf[6], tb); lds_read_frag<7168>(kf[7], tb);
      }
      if constexpr (decltype(rec)::value) request(req, wr_off);
      else if constexpr (MQ_ATT_ABL != 7) req_v(req, wr_off);        // behind the recomputed tiles every request is a vT piece only
      wr_off = rd_off;
      ring_step(rd_off);
      req = req + 1 < nkb - 1 ? req + 1 : nkb - 1;
      if constexpr (decltype(rec)::value) {
        asm volatile("s_waitcnt lgkmcnt(0)" : "+v"(vf[0]), "+v"(vf[1]), "+v"(vf[2]), "+v"(vf[3]), "+v"(kf[0]), "+v"(kf[1]), "+v"(kf[2]), "+v"(kf[3]), "+v"(kf[4]),
                     "+v"(kf[5]), "+v"(kf[6]), "+v"(kf[7]));
      } else {
        asm volatile("s_waitcnt lgkmcnt(0)" : "+v"(vf[0]), "+v"(vf[1]), "+v"(vf[2]), "+v"(vf[3]));
      }
    };
    auto pv_lds = [&](const v4i& pf_hi, const v4i& pf_lo) {
#pragma unroll
      for (int dt = 0; dt < 4; ++dt) {
        acc_hi[dt] = __builtin_amdgcn_mfma_i32_16x16x64_i8(vf[dt], pf_hi, acc_hi[dt], 0, 0, 0);
        acc_lo[dt] = __builtin_amdgcn_mfma_i32_16x16x64_i8(vf[dt], pf_lo, acc_lo[dt], 0, 0, 0);
        acc_v[dt] = __builtin_amdgcn_mfma_i32_16x16x64_i8(vf[dt], ones, acc_v[dt], 0, 0, 0);
      }
      // sum over the block's 64 keys of the stored probability bytes, per query: ones (as the 16 x 64 A tile) x p -- the matrix pipe
      // has the room, the VALU (8 v_sad_u8 per block) does not
      acc_ph = __builtin_amdgcn_mfma_i32_16x16x64_i8(ones, pf_hi, acc_ph, 0, 0, 0);
      acc_pl = __builtin_amdgcn_mfma_i32_16x16x64_i8(ones, pf_lo, acc_pl, 0, 0, 0);
    };
    using P1 = std::integral_constant<int, 1>;
    using P3 = std::integral_constant<int, 3>;
    auto sweep2 = [&](auto pcl) {
    auto recomputed_block = [&](int kb, auto with_diag, auto pieces) {
      v4i pf_hi, pf_lo;
      float ti[16];
      next_block(T_{}, pieces);
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        v4f acc = __builtin_amdgcn_mfma_f32_16x16x32_f16(__builtin_bit_cast(v8h, kf[2 * j]), qh[0], v4f{0.f, 0.f, 0.f, 0.f}, 0, 0, 0);
        acc = __builtin_amdgcn_mfma_f32_16x16x32_f16(__builtin_bit_cast(v8h, kf[2 * j + 1]), qh[1], acc, 0, 0, 0);
        ti[4 * j] = acc[0]; ti[4 * j + 1] = acc[1]; ti[4 * j + 2] = acc[2]; ti[4 * j + 3] = acc[3];
      }
      if (decltype(with_diag)::value && kb == kdiag) probs(ti, kb, pf_hi, pf_lo, T_{}, pcl);
      else probs(ti, kb, pf_hi, pf_lo, F_{}, pcl);
      pv_lds(pf_hi, pf_lo);
    };
    // the tile requested after a recomputed one is recomputed too (three pieces) -- except after the last of them (one piece; with a
    // running maximum every tile is recomputed and the last is followed by its own repetition)
    auto recompute = [&](auto with_diag) {
      for (int kb = 0; kb + 1 < nrec; ++kb) recomputed_block(kb, with_diag, P3{});
      if (nrec > 0) {
        if (nrec < nkb) recomputed_block(nrec - 1, with_diag, P1{});
        else recomputed_block(nrec - 1, with_diag, P3{});
      }
    };
    if (fixed_ref) {
      recompute(F_{});
#pragma unroll
      for (int u = 0; u < kER; ++u) {
        const int kb = n_reg0 + u;
        if (kb < n_lds0) {
          v4i pf_hi, pf_lo;
          next_block(F_{}, P1{});
          probs_from(ereg[u], pf_hi, pf_lo, pcl);
          pv_lds(pf_hi, pf_lo);
        }
      }
      for (int kb = n_lds0; kb < nkb; ++kb) {
        v4i pf_hi, pf_lo;
        float exv[16];
        next_block(F_{}, P1{});
#pragma unroll
        for (int i = 0; i < 4; ++i) {
          const float4 t4 = s_e[kEC > 0 ? kb - n_lds0 : 0][i][kEC > 0 ? threadIdx.x : 0];
          exv[4 * i] = t4.x; exv[4 * i + 1] = t4.y; exv[4 * i + 2] = t4.z; exv[4 * i + 3] = t4.w;
        }
        probs_from(exv, pf_hi, pf_lo, pcl);
        pv_lds(pf_hi, pf_lo);
      }
    } else {
      recompute(T_{});
    }
    };
    // e / l <= 1: the index stays within [z_p, rint(1 / s_p) + z_p] -- inside the grid whenever that holds [0, 1] (the reference's
    // calibrated softmax range does: its maximum is the first position's p = 1)
    const bool p_clamp = !(gpa.o >= gpa.qmin && gpa.inv_s * (1.0f + 0x1p-20f) + gpa.o < gpa.qmax + 0.5f);
    if (p_clamp) sweep2(T_{});
    else sweep2(F_{});
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");                 // the ring's trailing requests must not outlive the workgroup's LDS
  } else if constexpr (D == 64) {
    Tile t;
    VTile vt;
    const int nrec = fixed_ref ? n_reg0 : nkb;                       // blocks whose scores are recomputed
    if (nrec > 0) load_t(0, t);
    load_v(0, vt);
    auto pv_block = [&](int kb, const v4i& pf_hi, const v4i& pf_lo) {
#pragma unroll
      for (int dt = 0; dt < 4; ++dt) {
        acc_hi[dt] = __builtin_amdgcn_mfma_i32_16x16x64_i8(vt.vf[dt], pf_hi, acc_hi[dt], 0, 0, 0);
        acc_lo[dt] = __builtin_amdgcn_mfma_i32_16x16x64_i8(vt.vf[dt], pf_lo, acc_lo[dt], 0, 0, 0);
        acc_v[dt] = __builtin_amdgcn_mfma_i32_16x16x64_i8(vt.vf[dt], ones, acc_v[dt], 0, 0, 0);
      }
      if (kb + 1 < nkb) load_v(kb + 1, vt);
    };
    auto recompute = [&](auto with_diag) {                            // (with the grid top as reference the diagonal block is a parked one)
      for (int kb = 0; kb < nrec; ++kb) {
        v4i pf_hi, pf_lo;
        SC ti[16];
        scores_t(t, ti);
        if (kb + 1 < nrec) load_t(kb + 1, t);
        if (decltype(with_diag)::value && kb == kdiag) probs(ti, kb, pf_hi, pf_lo, T_{}, T_{});
        else probs(ti, kb, pf_hi, pf_lo, F_{}, T_{});
        pv_block(kb, pf_hi, pf_lo);
      }
    };
    recompute(T_{});
    if (fixed_ref) {
      if constexpr (kER > 0) {
#pragma unroll
        for (int u = 0; u < kER; ++u) {
          const int kb = n_reg0 + u;
          if (kb < n_lds0) {
            v4i pf_hi, pf_lo;
            probs_from(ereg[u], pf_hi, pf_lo, T_{});
            pv_block(kb, pf_hi, pf_lo);
          }
        }
      }
      for (int kb = n_lds0; kb < nkb; ++kb) {
        v4i pf_hi, pf_lo;
        float exv[16];
#pragma unroll
        for (int i = 0; i < 4; ++i) {
          const float4 t4 = s_e[kEC > 0 ? kb - n_lds0 : 0][i][kEC > 0 ? threadIdx.x : 0];
          exv[4 * i] = t4.x; exv[4 * i + 1] = t4.y; exv[4 * i + 2] = t4.z; exv[4 * i + 3] = t4.w;
        }
        probs_from(exv, pf_hi, pf_lo, T_{});
        pv_block(kb, pf_hi, pf_lo);
      }
    }
  } else {
    dma_block(0, 0, true);
    for (int kb = 0; kb < nkb; ++kb) {
      block_ready();
      if (kb + 1 < nkb) dma_block(kb + 1, (kb + 1) & 1, true);
      int ti[16];
      int_scores_lds(kb, kb & 1, ti);
      v4i pf_hi, pf_lo;
      if (kb == kdiag) probs(ti, kb, pf_hi, pf_lo, T_{}, T_{});
      else probs(ti, kb, pf_hi, pf_lo, F_{}, T_{});
      // p.v: the vT fragments four at a time by asm reads, the next four requested in front of this batch's eight MFMAs
      const unsigned vb = tile_lds + (kb & 1) * kTileBytes + kKBytes + lane * 16;
      v4i vfa[4], vfb[4];
      lds_read_frags<0>(vfa, vb, std::make_integer_sequence<int, 4>{});
      auto pv_batch = [&](auto G, v4i (&cur)[4], v4i (&nxt)[4]) {
        constexpr int g = decltype(G)::value;
        lds_fragments_wait(cur);
        if constexpr (4 * (g + 1) < NDT) lds_read_frags<4096 * (g + 1)>(nxt, vb, std::make_integer_sequence<int, 4>{});
#pragma unroll
        for (int i = 0; i < 4; ++i) {
          acc_hi[4 * g + i] = __builtin_amdgcn_mfma_i32_16x16x64_i8(cur[i], pf_hi, acc_hi[4 * g + i], 0, 0, 0);
          acc_lo[4 * g + i] = __builtin_amdgcn_mfma_i32_16x16x64_i8(cur[i], pf_lo, acc_lo[4 * g + i], 0, 0, 0);
        }
      };
      pv_batch(std::integral_constant<int, 0>{}, vfa, vfb);
      pv_batch(std::integral_constant<int, 1>{}, vfb, vfa);
      if constexpr (NDT > 8) {
        pv_batch(std::integral_constant<int, 2>{}, vfa, vfb);
        pv_batch(std::integral_constant<int, 3>{}, vfb, vfa);
      }
    }
  }
  MQ_ST(5);
  long long psum = 256ll * psum_hi + psum_lo;
  if constexpr (F16) {                                              // stored byte = unsigned byte - 128, 64 keys per block
    psum = 256ll * ((long long)acc_ph[0] + 8192ll * nkb) + ((long long)acc_pl[0] + 8192ll * nkb);
  } else {
    psum += __shfl_xor(psum, 16, 64);
    psum += __shfl_xor(psum, 32, 64);
  }
  const long long nproc = (long long)nkb * 64;
  // out[s][d] = sp * sv * sum_t (p_idx - zp)(v_st - zv),  p_idx = 256 (hi_s + 128) + (lo_s + 128)
  //           = sp * sv * [ 256 A_hi + A_lo + 32896 V - zv P - zp V + zp zv T' ],  A_* = sum byte * v_st, V = sum v_st, P = sum p_idx
  const float alpha_pv = __fmul_rn(gpa.s, gpb.s);
  float* orow = a.out + (size_t)s_abs * H * D + (size_t)h * D;
  int rsum = 0;
  unsigned opk = 0;
  const int32_t* vpre = D == 64 ? nullptr : a.v_prefix + ((size_t)kvh * (CS >> 6) + (nkb - 1)) * D;
#pragma unroll
  for (int dt = 0; dt < NDT; ++dt) {
    float o4[4];
    int4 vsum = {0, 0, 0, 0};
    if constexpr (D != 64) vsum = *reinterpret_cast<const int4*>(vpre + 16 * dt + 4 * tq);
    const int vs4[4] = {vsum.x, vsum.y, vsum.z, vsum.w};
#pragma unroll
    for (int e = 0; e < 4; ++e) {
      const long long V = D == 64 ? acc_v[D == 64 ? dt : 0][e] : vs4[e];
      const long long tot = 256ll * acc_hi[dt][e] + (long long)acc_lo[dt][e] + 32896ll * V - (long long)zv * psum - (long long)zp * V +
                            (long long)zp * zv * nproc;
      const float pre = (float)((double)tot * (double)alpha_pv);
      const float qi = gpo.on ? a_index_fast(pre, gpo) : 0.f;
      o4[e] = gpo.on ? __fmul_rn(__fsub_rn(qi, gpo.o), gpo.s) : pre;
      const int st = (int)qi - a.out_shift;
      rsum += st;
      opk |= ((unsigned)st & 0xffu) << (8 * e);
    }
#ifndef MQ_ATT_STAMPS
    if (a.out != nullptr) *reinterpret_cast<float4*>(orow + 16 * dt + 4 * tq) = make_float4(o4[0], o4[1], o4[2], o4[3]);
#endif
    // o_proj's int8 input image, row-major or fragment-blocked (layout: include/mobilequant_amd.h): 1-KiB block (row >> 4, k >> 6 = h), byte
    // 16 * ((row & 15) + 16 * ((k & 63) >> 4)) + (k & 15), k & 63 = 16 dt + 4 tq + e
    if (a.out_i8 != nullptr && s_abs < a.seq_real) {
      const int64_t row = a.out_row0 + s_abs;
      int8_t* dst = a.out_i8_tiled ? a.out_i8 + ((row >> 4) * (H * NKS) + h * NKS + (dt >> 2)) * 1024 + 16 * ((row & 15) + 16 * (dt & 3)) + 4 * tq
                                   : a.out_i8 + row * H * D + h * D + 16 * dt + 4 * tq;
      *reinterpret_cast<unsigned*>(dst) = opk;
    }
    opk = 0;
  }
  if (a.out_i8 != nullptr) {
    rsum += __shfl_xor(rsum, 16, 64);
    rsum += __shfl_xor(rsum, 32, 64);
    if (tq == 0 && s_abs < a.seq_real) atomicAdd(a.out_rowsum + a.out_row0 + s_abs, rsum);
  }
#ifdef MQ_ATT_STAMPS
  MQ_ST(6);
  st_[7] = now_() - st_begin;
  st_[8] = (unsigned long long)nkb;
  if (a.out != nullptr && lane == 0) {
    unsigned long long* dst = reinterpret_cast<unsigned long long*>(a.out) + ((size_t)blockIdx.x * 4 + wave) * 11;
#pragma unroll
    for (int i = 0; i < 11; ++i) dst[i] = st_[i];
  }
#endif
}

}  // namespace mq

using namespace mq;

static std::atomic<int> g_att_qprep{1};       // tuning hook: 0 = the prep kernel writes the q image too (A/B timing)
extern "C" int mq_attention_set_fused_q(int on) {
  g_att_qprep = on ? 1 : 0;
  return 0;
}
static std::atomic<int> g_att_cache{0};       // tuning hook: 1 = small exponential cache, anything else = the deep cache
extern "C" int mq_attention_set_cache(int mode) {
  g_att_cache = mode == 1 ? 1 : 0;
  return 0;
}

// Two heads of a KV group per eight-wave workgroup (PAIR): identical results, measured SLOWER (70-73 against 66-69 us: one workgroup per
// CU in lock step at its barriers, 512 instead of 1 024 workgroups to balance) -- built only with -DMQ_BUILD_EXPERIMENTS
// (python -m mobilequant_amd.build --experiments); the production library answers the knob with 1 (= not built).
static std::atomic<int> g_att_pair{0};
extern "C" int mq_attention_set_pair(int on) {
#ifdef MQ_BUILD_EXPERIMENTS
  g_att_pair = on ? 1 : 0;
  return 0;
#else
  (void)on;
  return 1;
#endif
}
static std::atomic<int> g_att_f16{1};         // tuning hook: 0 = int8 score contraction even when the fp16 images are supplied (A/B timing)
extern "C" int mq_attention_set_f16(int on) {
  g_att_f16 = on ? 1 : 0;
  return 0;
}

extern "C" int mq_attention_quant(const mq_attention_args* args, mq_stream_t stream) {
  MQ_REQUIRE(args != nullptr, "mq_attention_quant: null argument block");
  mq_attention_args a = *args;
  // the f16 score contraction serves head_dim 64 with a 16-bit score grid and the deep cache; anything else runs the int8 form
  if (!(a.head_dim == 64 && a.qk_out.scale != nullptr && a.q_f16 != nullptr && a.k_f16 != nullptr && g_att_cache.load() != 1 && g_att_f16.load() != 0))
    a.q_f16 = a.k_f16 = nullptr;
  MQ_REQUIRE(a.q_f16 == nullptr || (aligned(a.q_f16, 16) && aligned(a.k_f16, 16)), "mq_attention_quant: q_f16 / k_f16 must be 16-byte aligned");
  MQ_REQUIRE(((a.q && a.k && a.v) || a.qkv_idx) && a.cos && a.sin && (a.out || a.out_i8) && a.q_i8 && a.k_i8 && a.vt_i8 && a.q_rowsum && a.k_rowsum,
             "mq_attention_quant: null pointer");
  MQ_REQUIRE(a.seq <= 65536, "mq_attention_quant: seq = %d exceeds 65536 (int32 accumulators of the p.v products)", a.seq);
  MQ_REQUIRE(a.cache_seq == 0 ? a.pos0 == 0
                              : (a.cache_seq % 64 == 0 && a.pos0 >= 0 && a.pos0 % 64 == 0 && a.pos0 + a.seq <= a.cache_seq && a.cache_seq <= 65536),
             "mq_attention_quant: cache continuation needs pos0 %% 64 == 0, cache_seq %% 64 == 0, pos0 + seq <= cache_seq <= 65536 (pos0=%d cache_seq=%d); "
             "without a cache (cache_seq = 0) pos0 must be 0", a.pos0, a.cache_seq);
  MQ_REQUIRE((a.head_dim == 64 || a.head_dim == 128 || a.head_dim == 256) && a.seq > 0 && a.seq % 64 == 0 && a.heads > 0 && a.kv_heads > 0 && a.heads % a.kv_heads == 0,
             "mq_attention_quant: head_dim 64, 128 or 256, seq %% 64 == 0 (got head_dim=%d seq=%d heads=%d kv_heads=%d)", a.head_dim, a.seq, a.heads, a.kv_heads);
  MQ_REQUIRE(a.rot_dim >= 0 && a.rot_dim <= a.head_dim && a.rot_dim % 2 == 0, "mq_attention_quant: rot_dim = %d (0 = head_dim; even, <= head_dim)", a.rot_dim);
  MQ_REQUIRE(a.head_dim == 64 || (a.v_prefix != nullptr && aligned(a.v_prefix, 16)),
             "mq_attention_quant: head_dim 128 / 256 need the v_prefix scratch ([kv_heads][seq/64][head_dim] int32, 16-byte aligned)");
  MQ_REQUIRE(a.qk_a.scale && a.qk_b.scale && a.pv_a.scale && a.pv_b.scale && a.qk_a.qmax == 255.f && a.qk_b.qmax == 255.f && a.pv_b.qmax == 255.f &&
                 a.qk_a.qmin == 0.f && a.qk_b.qmin == 0.f && a.pv_b.qmin == 0.f && a.pv_a.qmin == 0.f && a.pv_a.qmax <= 65535.f,
             "mq_attention_quant: q / k / v need 8-bit unsigned grids, the probabilities an unsigned grid of at most 16 bits");
  MQ_REQUIRE((a.qkv_idx ? aligned(a.qkv_idx, 16) && a.q_in.scale && a.k_in.scale && a.v_in.scale : aligned(a.q, 16) && aligned(a.k, 16) && aligned(a.v, 16)) &&
                 (!a.out || aligned(a.out, 16)) && aligned(a.q_i8, 16) && aligned(a.k_i8, 16) &&
                 aligned(a.vt_i8, 16) && aligned(a.k_rowsum, 16),
             "mq_attention_quant: pointers must be 16-byte aligned");
  if (a.out_i8 != nullptr) {
    MQ_REQUIRE(a.out_rowsum != nullptr && a.pv_out.scale != nullptr && a.pv_out.qmin - (float)a.out_shift >= -128.f &&
                   a.pv_out.qmax - (float)a.out_shift <= 127.f && a.out_row0 >= 0 && a.seq_real > 0 && a.seq_real <= a.seq &&
                   aligned(a.out_i8, 16),
               "mq_attention_quant: the int8 output image needs an 8-bit pv_out grid that fits int8 after out_shift, out_rowsum, "
               "0 < seq_real <= seq");
  }
  MQ_REQUIRE(a.batch >= 0 && a.batch <= 65535 && (a.batch <= 1 || a.cache_seq == 0),
             "mq_attention_quant: batch = %d (0 / 1: one sequence; > 1 only without cache continuation)", a.batch);
  hipStream_t st = as_stream(stream);
  const unsigned nb = a.batch > 1 ? (unsigned)a.batch : 1u;
  const dim3 pgrid((unsigned)(a.seq / 64), (unsigned)(a.heads + 2 * a.kv_heads), nb), cgrid((unsigned)(a.seq / 64 * a.heads), 1, nb);
  if (a.head_dim == 64) {
    const bool big = g_att_cache.load() != 1;
    // production configuration (16-bit score grid, deep cache, full rotary or StableLM-2's 16 rotating dims): the core kernel prepares its own q rows
    const bool qprep = big && a.qk_out.scale != nullptr && (a.rot_dim == 0 || a.rot_dim == 64 || a.rot_dim == 16) && g_att_qprep.load() != 0;
    if (qprep) attention_prep_kernel<64><<<dim3(pgrid.x, (unsigned)(2 * a.kv_heads), nb), 256, 0, st>>>(a, a.heads);
    else attention_prep_kernel<64><<<pgrid, 256, 0, st>>>(a, 0);
    MQ_LAUNCH_CHECK("mq_attention_quant(prep)");
    const bool f16 = a.q_f16 != nullptr;
    if (a.qk_out.scale == nullptr) attention_quant_kernel<64, false><<<cgrid, 256, 0, st>>>(a);
#ifdef MQ_BUILD_EXPERIMENTS
    else if (qprep && f16 && g_att_pair.load() != 0 && a.heads % 2 == 0 && (a.heads / a.kv_heads) % 2 == 0)
      attention_quant_kernel<64, true, true, true, true, true><<<dim3(cgrid.x / 2, 1, nb), 512, 0, st>>>(a);
#endif
    else if (qprep && f16) attention_quant_kernel<64, true, true, true, true><<<cgrid, 256, 0, st>>>(a);
    else if (f16) attention_quant_kernel<64, true, true, false, true><<<cgrid, 256, 0, st>>>(a);
    else if (qprep) attention_quant_kernel<64, true, true, true><<<cgrid, 256, 0, st>>>(a);
    else if (big) attention_quant_kernel<64, true, true><<<cgrid, 256, 0, st>>>(a);
    else attention_quant_kernel<64, true><<<cgrid, 256, 0, st>>>(a);
  } else {
    if (a.head_dim == 128) attention_prep_kernel<128><<<pgrid, 256, 0, st>>>(a, 0);
    else attention_prep_kernel<256><<<pgrid, 256, 0, st>>>(a, 0);
    MQ_LAUNCH_CHECK("mq_attention_quant(prep)");
    attention_vprefix_kernel<<<dim3((unsigned)a.kv_heads, 1, nb), 256, 0, st>>>(a.v_prefix, a.cache_seq > 0 ? a.pos0 / 64 : 0, a.seq / 64,
                                                                               (a.cache_seq > 0 ? a.cache_seq : a.seq) / 64, a.head_dim);
    MQ_LAUNCH_CHECK("mq_attention_quant(prefix)");
    if (a.head_dim == 128) {
      if (a.qk_out.scale != nullptr) attention_quant_kernel<128, true><<<cgrid, 256, 0, st>>>(a);
      else attention_quant_kernel<128, false><<<cgrid, 256, 0, st>>>(a);
    } else {
      if (a.qk_out.scale != nullptr) attention_quant_kernel<256, true><<<cgrid, 256, 0, st>>>(a);
      else attention_quant_kernel<256, false><<<cgrid, 256, 0, st>>>(a);
    }
  }
  MQ_LAUNCH_CHECK("mq_attention_quant");
  return MQ_OK;
}

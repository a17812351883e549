// Single-token decode step of a llama-style W8A8 / W4A8-free simulated-quant model: the MI355X-native counterpart of the per-token
// body of SimModel.generate (mobilellm/model/sim_model.py:160-221) with the quantized module graph of qmodule.py / hf_model.py.
//
// A decoder layer at M = 1 is a chain of five dependent, HBM-latency-bound phases; each one is ONE launch here:
//   (1) mq_decode_gemv, NORM mode   : input_layernorm (QRMSNorm: 16-bit input grid, fake-quantised weight vector, 8-bit output grid)
//                                     fused into the prologue of the q|k|v weight stream; three row segments with their own output grids
//   (2) mq_decode_attention         : RoPE, static KV cache append, QMatMul qk_bmm (8/8 -> 16-bit), /sqrt(d), softmax, QMatMul pv_bmm
//                                     (16/8 -> 8-bit) over the cached positions
//   (3) mq_decode_gemv, RESID       : o_proj (input on pv_bmm's output grid, 16-bit output grid) + residual add
//   (4) mq_decode_gemv, NORM + GATE : post_attention_layernorm fused in front of the row-interleaved w1|w3 stream; the epilogue holds
//                                     both halves of a pair and applies QSiLU / QGELU, the product and w2's input quantizer: int8 out
//   (5) mq_decode_gemv, INT8 + RESID: w2 from the int8 image (16-bit output grid) + residual add
// and mq_decode_head: final HFRMSNorm (floating point, not quantised: qmodule.py:843) fused in front of the fp32 lm_head stream.
// Every weight byte is read once per token; the activations (<= 22 KiB) live in LDS per CU.  The GEMV body is the fat-workgroup
// kernel of mq_gemv.hip (one 1024-thread workgroup per CU, all weight loads of a wave issued before anything else, DPP reductions);
// the arithmetic of every quantizer is op for op that of the prefill kernels (mq_norm.hip, mq_elementwise.hip, mq_gemm.hip).
// The token position is read from device memory, so one captured hipGraph serves every step of a generation.
#include "mq_common.h"

namespace mq {

typedef int v4i __attribute__((ext_vector_type(4)));

#pragma clang fp contract(off)

__device__ __forceinline__ float dq_clamp_nan(float q, float lo, float hi) {
  const float c = fminf(fmaxf(q, lo), hi);
  return q != q ? q : c;
}
// x / s through the correctly rounded reciprocal y = RN(1 / s) and one fma correction (Markstein): bit-identical to the IEEE divide
// for |x / s| in [2^-2, 1e30] -- tools/div_check.cpp compares every fp32 dividend for 48 divisors on the GPU, incl. all-ones
// significands and the scale clamps 1e-5 / 1e6 (profiles/r03/div_check.log) -- and faithful below 2^-2, where round(x / s) = 0 and
// (round(t) - t) + t = 0 whatever the last bit of t is.  3 VALU instructions instead of the 10 + two mode switches of v_div_*: at
// M = 1 every CU quantises the whole activation row, and that arithmetic is the launch's critical path.
__device__ __forceinline__ float div_by_scale(float x, float s, float inv_s) {
  const float q0 = __fmul_rn(x, inv_s);
  return __builtin_fmaf(__builtin_fmaf(-q0, s, x), inv_s, q0);
}
// qmodule.py:286-290 with round_ste = (round(t) - t) + t (as mq_norm.hip / mq_elementwise.hip)
__device__ __forceinline__ float dq_index(float x, float s, float inv_s, float o, float qmin, float qmax) {
  const float t = div_by_scale(x, s, inv_s);
  const float r = __fadd_rn(__fsub_rn(rintf(t), t), t);
  return dq_clamp_nan(__fadd_rn(r, o), qmin, qmax);
}
__device__ __forceinline__ float dq_dequant(float q, float s, float o) { return __fmul_rn(__fsub_rn(q, o), s); }

struct Grid {          // device view of mq_grid
  float s, o, qmin, qmax, inv_s;
  bool on;
  __device__ __forceinline__ float fq(float v) const { return on ? dq_dequant(dq_index(v, s, inv_s, o, qmin, qmax), s, o) : v; }
};
__device__ __forceinline__ Grid load_grid(const mq_grid& g) {
  Grid r;
  r.on = g.scale != nullptr;
  r.s = r.on ? g.scale[0] : 1.f;
  r.o = r.on ? g.offset[0] : 0.f;
  r.qmin = g.qmin;
  r.qmax = g.qmax;
  r.inv_s = __fdiv_rn(1.0f, r.s);
  return r;
}

__device__ __forceinline__ int dot16(const v4i a, const v4i b, int c) {
#pragma unroll
  for (int e = 0; e < 4; ++e) c = __builtin_amdgcn_sdot4(a[e], b[e], c, false);
  return c;
}
__device__ __forceinline__ int wave_sum_dpp(int v) {
  v += __builtin_amdgcn_update_dpp(0, v, 0xB1, 0xf, 0xf, true);
  v += __builtin_amdgcn_update_dpp(0, v, 0x4E, 0xf, 0xf, true);
  v += __builtin_amdgcn_update_dpp(0, v, 0x141, 0xf, 0xf, true);
  v += __builtin_amdgcn_update_dpp(0, v, 0x140, 0xf, 0xf, true);
  v += __builtin_amdgcn_update_dpp(0, v, 0x142, 0xa, 0xf, false);
  v += __builtin_amdgcn_update_dpp(0, v, 0x143, 0xc, 0xf, false);
  return __builtin_amdgcn_readlane(v, 63);
}
__device__ __forceinline__ float wave_sum_f(float v) {
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
  return v;
}
__device__ __forceinline__ float wave_max_f(float v) {
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) v = fmaxf(v, __shfl_xor(v, o, 64));
  return v;
}

constexpr int DG_THREADS = 1024, DG_WAVES = 16, DG_INFLIGHT = 8;

// Profiling builds (-DMQ_DECODE_STAMPS, tools/decode_stamps.py): every workgroup leaves s_memrealtime stamps (100 MHz, one clock for
// the whole chip) at its phase boundaries.  Production builds compile the stamps out; the pointer argument stays null.
#ifdef MQ_DECODE_STAMPS
#define DG_STAMP(k)                                                                                   \
  do {                                                                                                \
    if (stamps && threadIdx.x == 0) stamps[(size_t)blockIdx.x * 16 + (k)] = __builtin_amdgcn_s_memrealtime(); \
  } while (0)
// stamp of wave `w` once the register `reg` has arrived
#define DG_STAMP_ARRIVED(k, w, reg)                                                                   \
  do {                                                                                                \
    if (stamps && wave == (w)) {                                                                      \
      asm volatile("" ::"v"(reg));                                                                    \
      if (lane == 0) stamps[(size_t)blockIdx.x * 16 + (k)] = __builtin_amdgcn_s_memrealtime();       \
    }                                                                                                 \
  } while (0)
#else
#define DG_STAMP(k) do { } while (0)
#define DG_STAMP_ARRIVED(k, w, reg) do { } while (0)
#endif

// Constants block of a launch (mq_decode_pack_grids): grid k at floats [4k .. 4k+2] = scale, offset, 1 / scale.  ONE 128-byte load
// per wave at the very top of the kernel replaces up to 16 dependent pointer chases behind the barriers.
__device__ __forceinline__ Grid const_grid(const float cv, const int k, const mq_grid& g) {
  Grid r;
  r.on = g.scale != nullptr;
  r.s = __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, cv), 4 * k));
  r.o = __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, cv), 4 * k + 1));
  r.inv_s = __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, cv), 4 * k + 2));
  r.qmin = g.qmin;
  r.qmax = g.qmax;
  return r;
}
enum { CG_NORM_IN = 0, CG_A = 1, CG_OUT0 = 2, CG_OUT1 = 3, CG_OUT2 = 4, CG_GATE_MID = 5, CG_GATE_ACTOUT = 6, CG_GATE_OUT = 7, CG_COUNT = 8 };

__global__ void decode_pack_grids_kernel(const mq_decode_grid_pack p, float* __restrict__ out) {
  const int k = threadIdx.x;
  if (k >= p.n) return;
  const float s = p.grids[k].scale ? p.grids[k].scale[0] : 1.f, o = p.grids[k].offset ? p.grids[k].offset[0] : 0.f;
  out[4 * k] = s;
  out[4 * k + 1] = o;
  out[4 * k + 2] = __fdiv_rn(1.0f, s);
  out[4 * k + 3] = 0.f;
}

// GATE: a logical row r is the weight-row pair (2r, 2r+1) = (w1 row r, w3 row r), 2K contiguous bytes.
// W4: weight rows hold packed unsigned nibbles (mq_pack_w4: K/2 bytes; a 16-byte group = 32 consecutive k, element j in the low and
//     j + 16 in the high nibble of byte j): unpacked in registers, two dot products per loaded chunk; w_zp / col_term are in the
//     unsigned-nibble domain as for mq_w4a8_linear.
//
// Wave roles.  The kernel is a chain of dependent round trips, and two machine facts decide its shape (profiles/r03/decode_stamps_*.log,
// tools/latency_probe.cpp): (1) a CU's vector-memory pipe takes ~16 cycles per 1-KiB wave load and serves the waves' requests in
// arrival order, so an activation load issued behind other waves' weight loads returns behind them -- microseconds, not the 0.1 us
// of a lone load; (2) hipcc counts outstanding loads statically: one conditional load between the activation loads and their first
// use and it waits with vmcnt(0), i.e. for the whole weight stream.  So the waves split the work:
//   * PROLOGUE waves 0 .. DG_PRO-1 (the first to start) request ONLY the launch constants, the activation row and the norm weights,
//     build the int8 image + row sum in LDS (norm -> quantize, arithmetic of mq_rmsnorm_quant / mq_quantize) and exit;
//   * STREAM waves DG_PRO .. 15 request ONLY weights and epilogue parameters, meet the image at the barrier, then run the dot
//     products as their loads return, and the epilogue (one row per lane).
// Both roles execute the same number of s_barrier (2 with a fused norm, 1 without).
enum { XM_NORM = 0, XM_F32 = 1, XM_I8 = 2 };
constexpr int DG_PRO = 8, DG_STR = DG_WAVES - DG_PRO, DG_XPRE = 4;      // 512 prologue threads x 4 float4 -> K <= 8192 (fp32)

template <int XMODE, bool GATE, bool W4>
__global__ void __launch_bounds__(DG_THREADS) decode_gemv_kernel(const mq_decode_gemv_args g, const int rows_per_wg, unsigned long long* stamps) {
  DG_STAMP(0);
  extern __shared__ __attribute__((aligned(16))) char smem[];   // [K] int8 activation image
  __shared__ float s_red[DG_PRO];
  __shared__ int s_redi[DG_PRO];
  const int lane = threadIdx.x & 63;
  const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const int K = g.K;
  const float cv = g.consts[lane & 63];                            // one 256-byte line per wave, both roles

  if (wave < DG_PRO) {
    // ================================================ PROLOGUE role ====================================================================
    const int p = threadIdx.x;                                     // 0 .. 511
    int my_sum = 0;
    if constexpr (XMODE == XM_I8) {                                // ready int8 image (w2 after the gated epilogue): copy + row sum
      const int nq = K >> 4;
      for (int i = p; i < nq; i += DG_PRO * 64) {
        const v4i v = reinterpret_cast<const v4i*>(g.xq)[i];
        reinterpret_cast<v4i*>(smem)[i] = v;
#pragma unroll
        for (int e = 0; e < 4; ++e) my_sum = __builtin_amdgcn_sdot4(v[e], 0x01010101, my_sum, false);
      }
    } else {
      const int nvec = K >> 2;
      float4 xv[DG_XPRE], nw[DG_XPRE];
#pragma unroll
      for (int u = 0; u < DG_XPRE; ++u) {
        if (u * DG_PRO * 64 < nvec) {                              // wave-uniform
          const int i = p + u * DG_PRO * 64;
          xv[u] = reinterpret_cast<const float4*>(g.x)[i < nvec ? i : nvec - 1];
          if constexpr (XMODE == XM_NORM) nw[u] = reinterpret_cast<const float4*>(g.norm_w)[i < nvec ? i : nvec - 1];
        }
      }
      DG_STAMP_ARRIVED(6, 0, xv[0].x);
      const Grid ag = const_grid(cv, CG_A, g.a_grid);            // (the constants were requested before the row: they are here)
      float r = 1.f;
      if constexpr (XMODE == XM_NORM) {                            // QRMSNorm.forward (qmodule.py:515-531), arithmetic of mq_rmsnorm_quant
        const Grid ng = const_grid(cv, CG_NORM_IN, g.norm_in);
        float ss = 0.f;
#pragma unroll
        for (int u = 0; u < DG_XPRE; ++u) {
          if (u * DG_PRO * 64 < nvec) {
            float4& v = xv[u];
            v.x = ng.fq(v.x); v.y = ng.fq(v.y); v.z = ng.fq(v.z); v.w = ng.fq(v.w);
            if (p + u * DG_PRO * 64 < nvec) {
              ss += v.x * v.x;
              ss += v.y * v.y;
              ss += v.z * v.z;
              ss += v.w * v.w;
            }
          }
        }
        DG_STAMP_ARRIVED(8, 0, ss);
        ss = wave_sum_f(ss);
        if (lane == 0) s_red[wave] = ss;
        DG_STAMP_ARRIVED(9, 0, ss);
        __syncthreads();                                           // barrier 1 of 2 (the stream waves pass it right after their requests)
        float tot = 0.f;
#pragma unroll
        for (int w = 0; w < DG_PRO; ++w) tot += s_red[w];
        const float mean = __fdiv_rn(tot, (float)K);
        r = __fdiv_rn(1.0f, __fsqrt_rn(__fadd_rn(mean, g.eps)));
        DG_STAMP(1);
      }
#pragma unroll
      for (int u = 0; u < DG_XPRE; ++u) {
        if (u * DG_PRO * 64 < nvec) {
          const int i = p + u * DG_PRO * 64;
          float4 v = xv[u];
          if constexpr (XMODE == XM_NORM) {
            const float4 w = nw[u];
            v.x = __fmul_rn(w.x, __fmul_rn(v.x, r)); v.y = __fmul_rn(w.y, __fmul_rn(v.y, r));
            v.z = __fmul_rn(w.z, __fmul_rn(v.z, r)); v.w = __fmul_rn(w.w, __fmul_rn(v.w, r));
          }
          const float f[4] = {v.x, v.y, v.z, v.w};
          unsigned pk = 0;
          int sum4 = 0;
#pragma unroll
          for (int e = 0; e < 4; ++e) {
            const float qi = dq_index(f[e], ag.s, ag.inv_s, ag.o, ag.qmin, ag.qmax);
            const int st = (qi != qi ? (int)ag.qmin : (int)qi) - 128;
            sum4 += st;
            pk |= ((unsigned)st & 0xffu) << (8 * e);
          }
          if (i < nvec) {
            my_sum += sum4;
            reinterpret_cast<unsigned*>(smem)[i] = pk;
          }
        }
      }
    }
    DG_STAMP_ARRIVED(10, 0, my_sum);
    const int part = wave_sum_dpp(my_sum);
    if (lane == 0) s_redi[wave] = part;
    __syncthreads();                                               // the image and the row sum are complete
    DG_STAMP(2);
    return;
  }

  // ================================================== STREAM role ======================================================================
#ifdef MQ_DECODE_STAMPS
  if (stamps && wave == DG_PRO && lane == 0) {
    stamps[(size_t)blockIdx.x * 16 + 11] = __builtin_amdgcn_s_memtime();
    stamps[(size_t)blockIdx.x * 16 + 13] = __builtin_amdgcn_s_memrealtime();
  }
#endif
  const int sw = wave - DG_PRO;                                    // 0 .. DG_STR-1
  const int NL = GATE ? g.N >> 1 : g.N;                            // logical rows
  const int kchunks = W4 ? K >> 5 : K >> 4;                        // 16-byte chunks per weight row
  const int wrow = W4 ? K >> 1 : K;                                // bytes per weight row
  const int lchunks = GATE ? 2 * kchunks : kchunks;                // chunks per logical row
  const int cpl = (lchunks + 63) >> 6;
  const int row0 = blockIdx.x * rows_per_wg + sw;
  const int row_end = (blockIdx.x + 1) * rows_per_wg < NL ? (blockIdx.x + 1) * rows_per_wg : NL;
  const int prow = row0 + DG_STR * lane;                           // lane t keeps the parameters / result of row slot t
  const bool prow_ok = prow < row_end;
  v4i buf[DG_INFLIGHT];
  auto issue_pass = [&](int t, int j) {
#pragma unroll
    for (int u = 0; u < DG_INFLIGHT; ++u) {
      const int row = row0 + DG_STR * t;
      const int c = lane + 64 * j;
      if (row < row_end && c < lchunks)
        buf[u] = __builtin_nontemporal_load(reinterpret_cast<const v4i*>(g.w + (size_t)row * (GATE ? 2 : 1) * wrow) + c);
      else
        buf[u] = v4i{0, 0, 0, 0};
      if (++j == cpl) { j = 0; ++t; }
    }
  };
  issue_pass(0, 0);
  float p_alpha[GATE ? 2 : 1], p_bias[GATE ? 2 : 1];
  int p_zp[GATE ? 2 : 1], p_ct[GATE ? 2 : 1];
#pragma unroll
  for (int h = 0; h < (GATE ? 2 : 1); ++h) {
    const int wr = GATE ? 2 * prow + h : prow;
    p_alpha[h] = prow_ok ? g.alpha[wr] : 0.f;
    p_zp[h] = prow_ok ? g.w_zp[wr] : 0;
    p_ct[h] = prow_ok ? g.col_term[wr] : 0;
    p_bias[h] = (prow_ok && g.bias) ? g.bias[wr] : 0.f;
  }
  float p_res = 0.f;
  if (!GATE && g.resid && prow_ok) p_res = g.resid[prow];
  if constexpr (XMODE == XM_NORM) __syncthreads();                 // barrier 1 of 2: the prologue waves' sum of squares
  __syncthreads();                                                 // the image and the row sum are complete
  if (row0 >= row_end) return;
  int rs = 0;
#pragma unroll
  for (int w = 0; w < DG_PRO; ++w) rs += s_redi[w];

  // ---- output grids -----------------------------------------------------------------------------------------------------------------
  const Grid og0 = const_grid(cv, CG_OUT0, g.out_grid[0]), og1 = const_grid(cv, CG_OUT1, g.out_grid[1]), og2 = const_grid(cv, CG_OUT2, g.out_grid[2]);
  const Grid gmid = const_grid(cv, CG_GATE_MID, g.gate_mid), gact = const_grid(cv, CG_GATE_ACTOUT, g.gate_actout),
             gout = const_grid(cv, CG_GATE_OUT, g.gate_out);
  auto out_q = [&](const Grid& q, float f) {        // output quantizer as the GEMM / GEMV epilogues evaluate it (reciprocal multiply)
    if (!q.on) return f;
    float v = rintf(f * q.inv_s) + q.o;
    v = fminf(fmaxf(v, q.qmin), q.qmax);
    return __fmul_rn(__fsub_rn(v, q.o), q.s);
  };

  // ---- dot products, DPP reduction and epilogue per completed logical row ----------------------------------------------------
  const int nslots = (row_end - row0 + DG_STR - 1) / DG_STR;
  int acc0 = 0, acc1 = 0;
  int sum0 = 0, sum1 = 0;
  int t = 0, j = 0;
  while (t < nslots) {
    int t2 = t, j2 = j;
#pragma unroll
    for (int u = 0; u < DG_INFLIGHT; ++u) {
      if (t2 < nslots) {
        int c = lane + 64 * j2;
        c = c < lchunks ? c : lchunks - 1;                       // buf[u] is zero there
        const int ck = GATE ? (c >= kchunks ? c - kchunks : c) : c;
        int part;
        if constexpr (W4) {
          const v4i a_lo = *reinterpret_cast<const v4i*>(smem + (size_t)ck * 32), a_hi = *reinterpret_cast<const v4i*>(smem + (size_t)ck * 32 + 16);
          v4i w_lo, w_hi;
#pragma unroll
          for (int e = 0; e < 4; ++e) {
            w_lo[e] = buf[u][e] & 0x0f0f0f0f;
            w_hi[e] = (int)(((unsigned)buf[u][e] >> 4) & 0x0f0f0f0fu);
          }
          part = dot16(w_hi, a_hi, dot16(w_lo, a_lo, 0));
        } else {
          part = dot16(buf[u], *reinterpret_cast<const v4i*>(smem + (size_t)ck * 16), 0);
        }
        if (GATE && c >= kchunks) acc1 += part;
        else acc0 += part;
        if (j2 == cpl - 1) {                                     // logical row slot t2 complete
          const int s0 = wave_sum_dpp(acc0), s1 = GATE ? wave_sum_dpp(acc1) : 0;
          acc0 = acc1 = 0;
          if (lane == t2) {                                      // lane t keeps the contraction(s) of row slot t
            sum0 = s0;
            sum1 = s1;
          }
        }
        if (++j2 == cpl) { j2 = 0; ++t2; }
      }
    }
    t = t2;
    j = j2;
    if (t < nslots) issue_pass(t, j);
  }
  DG_STAMP_ARRIVED(3, DG_PRO, sum0);
  // ---- epilogue, one row per LANE: every row of the wave goes through its quantizers at the same time -----------------------
  if (lane < nslots) {
    const int row = prow;
    float e0, e1 = 0.f;
    {
      const int tt = (int)((unsigned)sum0 - (unsigned)p_zp[0] * (unsigned)rs + (unsigned)p_ct[0]);
      e0 = __fadd_rn(__fmul_rn((float)tt, p_alpha[0]), p_bias[0]);
    }
    if constexpr (GATE) {
      const int tt = (int)((unsigned)sum1 - (unsigned)p_zp[GATE ? 1 : 0] * (unsigned)rs + (unsigned)p_ct[GATE ? 1 : 0]);
      e1 = __fadd_rn(__fmul_rn((float)tt, p_alpha[GATE ? 1 : 0]), p_bias[GATE ? 1 : 0]);
    }
    if constexpr (GATE) {
      const float fa = out_q(og0, e0), fb = out_q(og1, e1);
      float rr;
      if (g.gate_act == 0) {                                     // QSiLU (qmodule.py:739-753)
        const float gate = __fdiv_rn(1.0f, __fadd_rn(1.0f, expf(-fa)));
        rr = __fmul_rn(fa, gmid.fq(gate));
      } else {                                                   // QGELU (qmodule.py:790-798), erf form
        rr = __fmul_rn(__fmul_rn(0.5f, fa), __fadd_rn(1.0f, erff(__fmul_rn(fa, 0.70710678118654752440f))));
      }
      const float prod = __fmul_rn(gact.fq(rr), fb);
      const float qi = dq_index(prod, gout.s, gout.inv_s, gout.o, gout.qmin, gout.qmax);
      g.gate_q[row] = (int8_t)((qi != qi ? (int)gout.qmin : (int)qi) - 128);
      if (g.y) g.y[row] = prod;
    } else {
      float v = row < g.seg_end[0] ? out_q(og0, e0) : (row < g.seg_end[1] ? out_q(og1, e0) : out_q(og2, e0));
      if (g.resid) v = __fadd_rn(p_res, v);
      g.y[row] = v;
    }
  }
#ifdef MQ_DECODE_STAMPS
  if (stamps && wave == DG_PRO) {
    if (lane == 0) stamps[(size_t)blockIdx.x * 16 + 4] = __builtin_amdgcn_s_memrealtime();
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    if (lane == 0) stamps[(size_t)blockIdx.x * 16 + 5] = __builtin_amdgcn_s_memrealtime();
    if (lane == 0) stamps[(size_t)blockIdx.x * 16 + 12] = __builtin_amdgcn_s_memtime();
  }
#endif
}

// ---- attention for ONE query token over a static KV cache -------------------------------------------------------------------------
// One workgroup (256 threads) per query head.  hf_model.py:486-534 with the QMatMul pair of qmodule.py:453-466:
//   q, k_new <- RoPE;  cache[pos] <- (k_new, v_new)           (written by the first head of each KV group)
//   s[t] = Qqk_out( sum_d Qqk_a(q[d]) * Qqk_b(K[t][d]) ) / sqrt(D),   t = 0 .. pos        (the mask admits exactly these)
//   p = softmax(s)  (fp32)
//   o[d] = Qpv_out( sum_t Qpv_a(p[t]) * Qpv_b(V[t][d]) )
__global__ void __launch_bounds__(256) decode_attention_kernel(const mq_decode_attention_args a, unsigned long long* stamps) {
  DG_STAMP(0);
  extern __shared__ __attribute__((aligned(16))) char smem_raw[];
  float* s_p = reinterpret_cast<float*>(smem_raw);               // [T] scores / probabilities
  __shared__ float s_q[256], s_k[256], s_v[256], s_red[4], s_o[4][256];
  const int D = a.head_dim, h = blockIdx.x, kvh = h / (a.heads / a.kv_heads);
  const int pos = a.pos[0];
  const int T = pos + 1;
  const int tid = threadIdx.x, lane = tid & 63, wv = tid >> 6;
  const Grid qa = load_grid(a.qk_a), qb = load_grid(a.qk_b), qo = load_grid(a.qk_out);
  const Grid pa = load_grid(a.pv_a), pb = load_grid(a.pv_b), po = load_grid(a.pv_out);
  const float* qp = a.qkv + (size_t)h * D;
  const float* kp = a.qkv + (size_t)a.heads * D + (size_t)kvh * D;
  const float* vp = a.qkv + (size_t)(a.heads + a.kv_heads) * D + (size_t)kvh * D;
  float* kc = a.k_cache + (size_t)kvh * a.cache_len * D;
  float* vc = a.v_cache + (size_t)kvh * a.cache_len * D;
  if (tid < D) {                                                   // RoPE (rotate-half): x * cos + rot(x) * sin
    const int half = D >> 1;
    const float c = a.cos[(size_t)pos * D + tid], s = a.sin[(size_t)pos * D + tid];
    const float qr = tid < half ? -qp[tid + half] : qp[tid - half];
    const float kr = tid < half ? -kp[tid + half] : kp[tid - half];
    const float qv = __fadd_rn(__fmul_rn(qp[tid], c), __fmul_rn(qr, s));
    const float kv = __fadd_rn(__fmul_rn(kp[tid], c), __fmul_rn(kr, s));
    // The cache holds the keys / values ON THEIR QMatMul input grids (qk_bmm.input2, pv_bmm.input2): the reference re-quantises
    // the whole cached tensor at every step (qmodule.py:453-466) with static grids, which is idempotent -- quantising once at
    // append time gives the same numbers and takes ~2 IEEE divisions per cached element per step out of this kernel.
    s_q[tid] = qa.fq(qv);
    s_k[tid] = qb.fq(kv);
    s_v[tid] = pb.fq(vp[tid]);
    if (h % (a.heads / a.kv_heads) == 0) {                         // the group's first head appends to the cache
      kc[(size_t)pos * D + tid] = s_k[tid];
      vc[(size_t)pos * D + tid] = s_v[tid];
    }
  }
  __syncthreads();
  DG_STAMP(1);
  // At M = 1 this kernel is a chain of dependent memory round trips, so every phase issues its loads in bulk: positions are
  // processed in chunks of 256; a thread holds 16 float4 of keys (4 lanes per position, 64 positions per unit, 4 units) and 16
  // float4 of values (16 lanes per position row, 16 rows per unit, 16 units) -- and the FIRST chunk's values are requested
  // before the scores are even started (they do not depend on them).
  const float inv_sqrt_d = a.inv_sqrt_d;
  const bool fast = D == 64;                                        // the vectorised mapping (TinyLlama / StableLM heads)
  const int sub = tid & 3, grp = tid >> 2;                          // keys : lane sub of position grp (+ 64 u)
  const int vq = tid & 15, vr = tid >> 4;                           // values: float4 column vq of position row vr (+ 16 u)
  float4 vbuf[16];
  auto load_values = [&](int c0) {
#pragma unroll
    for (int u = 0; u < 16; ++u) {
      const int t = c0 + 16 * u + vr;
      vbuf[u] = (t < T && t != pos) ? reinterpret_cast<const float4*>(vc + (size_t)t * D)[vq] : reinterpret_cast<const float4*>(s_v)[vq];
    }
  };
  if (fast) load_values(0);
  float lmax = -INFINITY;
  if (fast) {
    for (int c0 = 0; c0 < T; c0 += 256) {
      float4 kbuf[4][4];
#pragma unroll
      for (int u = 0; u < 4; ++u) {
        const int t = c0 + 64 * u + grp;
        const float* kr = (t < T && t != pos) ? kc + (size_t)t * D : s_k;     // the new key never makes the round trip through memory
#pragma unroll
        for (int c = 0; c < 4; ++c) kbuf[u][c] = reinterpret_cast<const float4*>(kr + sub * 16)[c];
      }
      const float* q4 = s_q + sub * 16;
#pragma unroll
      for (int u = 0; u < 4; ++u) {
        const int t = c0 + 64 * u + grp;
        float acc = 0.f;
#pragma unroll
        for (int c = 0; c < 4; ++c) {
          acc += q4[4 * c] * kbuf[u][c].x;
          acc += q4[4 * c + 1] * kbuf[u][c].y;
          acc += q4[4 * c + 2] * kbuf[u][c].z;
          acc += q4[4 * c + 3] * kbuf[u][c].w;
        }
        acc += __shfl_xor(acc, 2, 64);
        acc += __shfl_xor(acc, 1, 64);
        if (t < T && sub == 0) {
          const float sc = __fmul_rn(qo.fq(acc), inv_sqrt_d);        // qk_bmm(...) / sqrt(head_dim)  (hf_model.py:513)
          s_p[t] = sc;
          lmax = fmaxf(lmax, sc);
        }
      }
    }
  } else {                                                          // generic head_dim: one thread per position
    for (int t = tid; t < T; t += 256) {
      const float* kr = (t == pos) ? s_k : kc + (size_t)t * D;
      float acc = 0.f;
      for (int d = 0; d < D; ++d) acc += s_q[d] * kr[d];
      const float sc = __fmul_rn(qo.fq(acc), inv_sqrt_d);
      s_p[t] = sc;
      lmax = fmaxf(lmax, sc);
    }
  }
  lmax = wave_max_f(lmax);
  if (lane == 0) s_red[wv] = lmax;
  __syncthreads();
  DG_STAMP(2);
  const float mx = fmaxf(fmaxf(s_red[0], s_red[1]), fmaxf(s_red[2], s_red[3]));
  __syncthreads();
  float lsum = 0.f;
  for (int t = tid; t < T; t += 256) {
    const float e = expf(s_p[t] - mx);
    s_p[t] = e;
    lsum += e;
  }
  lsum = wave_sum_f(lsum);
  if (lane == 0) s_red[wv] = lsum;
  __syncthreads();
  const float inv_sum = __fdiv_rn(1.0f, (s_red[0] + s_red[1]) + (s_red[2] + s_red[3]));
  for (int t = tid; t < T; t += 256) s_p[t] = pa.fq(__fmul_rn(s_p[t], inv_sum));      // pv_bmm's input quantizer, once per position
  __syncthreads();
  DG_STAMP(3);
  float* s_acc = &s_o[0][0];                                        // [16][64] partial outputs (fast path) / [4][256] (generic)
  if (fast) {
    float4 acc = make_float4(0.f, 0.f, 0.f, 0.f);
    for (int c0 = 0; c0 < T; c0 += 256) {
      if (c0 > 0) load_values(c0);
#pragma unroll
      for (int u = 0; u < 16; ++u) {
        const int t = c0 + 16 * u + vr;
        if (t < T) {
          const float p = s_p[t];
          acc.x += p * vbuf[u].x;
          acc.y += p * vbuf[u].y;
          acc.z += p * vbuf[u].z;
          acc.w += p * vbuf[u].w;
        }
      }
    }
    reinterpret_cast<float4*>(s_acc + vr * 64)[vq] = acc;
    __syncthreads();
    DG_STAMP(4);
    if (tid < 64) {
      float tot = 0.f;
#pragma unroll
      for (int r = 0; r < 16; ++r) tot += s_acc[r * 64 + tid];
      a.out[(size_t)h * D + tid] = po.fq(tot);
    }
  } else {
    for (int d = lane; d < D; d += 64) {
      float acc = 0.f;
      for (int t = wv; t < T; t += 4) acc += s_p[t] * ((t == pos) ? s_v[d] : vc[(size_t)t * D + d]);
      s_o[wv][d] = acc;
    }
    __syncthreads();
    if (tid < D) a.out[(size_t)h * D + tid] = po.fq((s_o[0][tid] + s_o[1][tid]) + (s_o[2][tid] + s_o[3][tid]));
  }
#ifdef MQ_DECODE_STAMPS
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  DG_STAMP(5);
#endif
}

// ---- final norm (floating point HFRMSNorm) + lm_head (fp32 weights) ----------------------------------------------------------------
__global__ void __launch_bounds__(256) decode_head_kernel(const float* __restrict__ x, const float* __restrict__ norm_w, float eps,
                                                          const float* __restrict__ w, const float* __restrict__ bias, int K, int V,
                                                          float* __restrict__ logits, unsigned long long* stamps) {
  DG_STAMP(0);
  extern __shared__ __attribute__((aligned(16))) char smem_raw[];
  float* s_x = reinterpret_cast<float*>(smem_raw);               // [K] normalised activation
  __shared__ float s_red[4];
  const int tid = threadIdx.x, lane = tid & 63, wv = tid >> 6;
  float ss = 0.f;
  for (int i = tid; i < K; i += 256) ss += x[i] * x[i];
  ss = wave_sum_f(ss);
  if (lane == 0) s_red[wv] = ss;
  __syncthreads();
  const float mean = __fdiv_rn((s_red[0] + s_red[1]) + (s_red[2] + s_red[3]), (float)K);
  const float r = __fdiv_rn(1.0f, __fsqrt_rn(__fadd_rn(mean, eps)));      // hf_model.py:183 (x * rsqrt(mean + eps)), then weight *
  for (int i = tid; i < K; i += 256) s_x[i] = norm_w ? __fmul_rn(norm_w[i], __fmul_rn(x[i], r)) : x[i];
  __syncthreads();
  // a wave per vocabulary row, float4 loads (16 B per lane)
  const int nvec = K >> 2;
  for (int row = blockIdx.x * 4 + wv; row < V; row += gridDim.x * 4) {
    typedef float v4f __attribute__((ext_vector_type(4)));
    const v4f* wr = reinterpret_cast<const v4f*>(w + (size_t)row * K);
    float acc = 0.f;
    for (int i = lane; i < nvec; i += 64) {
      const v4f a = __builtin_nontemporal_load(wr + i);
      const float4 b = reinterpret_cast<const float4*>(s_x)[i];
      acc += a[0] * b.x;
      acc += a[1] * b.y;
      acc += a[2] * b.z;
      acc += a[3] * b.w;
    }
    acc = wave_sum_f(acc);
    if (lane == 0) logits[row] = bias ? acc + bias[row] : acc;
  }
  DG_STAMP(4);
}

}  // namespace mq

using namespace mq;

#ifdef MQ_DECODE_STAMPS
// profiling builds only (not in any header): stamp buffer + a host log of (kind, grid, base slot) per launch
static unsigned long long* g_stamp_buf = nullptr;
static long long g_stamp_cap = 0, g_stamp_next = 0;
static long long g_stamp_log[3 * 4096];
static int g_stamp_nlog = 0;
static unsigned long long* stamp_slot(int kind, unsigned grid) {
  if (!g_stamp_buf || g_stamp_next + (long long)grid * 16 > g_stamp_cap || g_stamp_nlog >= 4096) return nullptr;
  unsigned long long* p = g_stamp_buf + g_stamp_next;
  g_stamp_log[3 * g_stamp_nlog] = kind, g_stamp_log[3 * g_stamp_nlog + 1] = grid, g_stamp_log[3 * g_stamp_nlog + 2] = g_stamp_next;
  ++g_stamp_nlog;
  g_stamp_next += (long long)grid * 16;
  return p;
}
extern "C" void mq_decode_set_stamps_(void* buf, long long cap_words) {
  g_stamp_buf = static_cast<unsigned long long*>(buf), g_stamp_cap = cap_words, g_stamp_next = 0, g_stamp_nlog = 0;
}
extern "C" int mq_decode_stamp_log_(long long* out, int cap) {
  const int n = g_stamp_nlog < cap ? g_stamp_nlog : cap;
  for (int i = 0; i < 3 * n; ++i) out[i] = g_stamp_log[i];
  return n;
}
#define STAMP_SLOT(kind, grid) stamp_slot(kind, grid)
#else
#define STAMP_SLOT(kind, grid) nullptr
#endif

extern "C" {

int mq_decode_pack_grids(const mq_grid* grids, int n, float* consts, mq_stream_t stream) {
  MQ_REQUIRE(grids && consts && n > 0 && n <= MQ_DECODE_MAX_GRIDS, "mq_decode_pack_grids: 1..%d grids", MQ_DECODE_MAX_GRIDS);
  mq_decode_grid_pack p;
  p.n = n;
  for (int k = 0; k < n; ++k) p.grids[k] = grids[k];
  decode_pack_grids_kernel<<<1, 64, 0, as_stream(stream)>>>(p, consts);
  MQ_LAUNCH_CHECK("mq_decode_pack_grids");
  return MQ_OK;
}

int mq_decode_gemv(const mq_decode_gemv_args* args, mq_stream_t stream) {
  MQ_REQUIRE(args != nullptr, "mq_decode_gemv: null argument block");
  const mq_decode_gemv_args& g = *args;
  MQ_REQUIRE(g.w && g.alpha && g.w_zp && g.col_term && (g.x || g.xq), "mq_decode_gemv: null pointer");
  MQ_REQUIRE(g.consts != nullptr && aligned(g.consts, 16), "mq_decode_gemv: consts (64 floats: mq_decode_pack_grids of this launch's 8 grids, zero padded) is required");
  MQ_REQUIRE(g.xq || g.K <= DG_XPRE * 4 * DG_PRO * 64, "mq_decode_gemv: K=%d exceeds the fp32 activation row the prologue waves hold in registers (8192)", g.K);
  MQ_REQUIRE(g.K > 0 && g.K % 256 == 0 && g.K <= 32768 && g.N > 0, "mq_decode_gemv: K=%d must be a positive multiple of 256 (<= 32768), N=%d", g.K, g.N);
  MQ_REQUIRE(g.xq || (g.a_grid.scale && g.a_grid.offset && g.a_grid.qmin == 0.f && g.a_grid.qmax == 255.f),
             "mq_decode_gemv: fp32 activations need an 8-bit unsigned activation grid");
  MQ_REQUIRE(aligned(g.w, 16) && (!g.x || aligned(g.x, 16)) && (!g.xq || aligned(g.xq, 4)) && (!g.norm_w || aligned(g.norm_w, 16)),
             "mq_decode_gemv: pointers must be 16-byte aligned");
  const bool gate = g.gate_q != nullptr;
  MQ_REQUIRE(gate || g.y, "mq_decode_gemv: no output");
  MQ_REQUIRE(!gate || (g.norm_w && !g.xq), "mq_decode_gemv: gate mode is served for the norm-fused prologue (fp32 x + norm_w)");
  MQ_REQUIRE(!gate || (g.N % 2 == 0 && g.gate_out.scale && g.out_grid[0].scale && g.out_grid[1].scale && (g.gate_act == 0 || g.gate_act == 1)),
             "mq_decode_gemv: gate mode needs an even N (interleaved w1 / w3 rows), both output grids and the w2 input grid");
  static std::atomic<int> cus_of[kMaxDevices];
  const int dev = current_device();
  int cus = cus_of[dev].load(std::memory_order_relaxed);
  if (!cus) {
    hipDeviceProp_t prop;
    cus = (hipGetDeviceProperties(&prop, dev) == hipSuccess && prop.multiProcessorCount > 0) ? prop.multiProcessorCount : 256;
    cus_of[dev].store(cus, std::memory_order_relaxed);
  }
  const int NL = gate ? g.N / 2 : g.N;
  int rows_per_wg = (NL + cus - 1) / cus;
  if (rows_per_wg > DG_STR * 64) rows_per_wg = DG_STR * 64;
  const unsigned grid = (unsigned)((NL + rows_per_wg - 1) / rows_per_wg);
  const size_t lds = (size_t)g.K + 64;
  hipStream_t st = as_stream(stream);
  unsigned long long* stamps = STAMP_SLOT(gate ? 1 : (g.norm_w ? 0 : (g.xq ? 3 : 2)), grid);
  const int xmode = g.xq ? XM_I8 : (g.norm_w ? XM_NORM : XM_F32);
#define MQ_DG_LAUNCH(XM, GT, W4)                                                                                   \
  decode_gemv_kernel<XM, GT, W4><<<grid, DG_THREADS, lds, st>>>(g, rows_per_wg, stamps)
  if (gate) {
    if (g.w4) MQ_DG_LAUNCH(XM_NORM, true, true);
    else MQ_DG_LAUNCH(XM_NORM, true, false);
  } else if (xmode == XM_NORM) {
    if (g.w4) MQ_DG_LAUNCH(XM_NORM, false, true);
    else MQ_DG_LAUNCH(XM_NORM, false, false);
  } else if (xmode == XM_F32) {
    if (g.w4) MQ_DG_LAUNCH(XM_F32, false, true);
    else MQ_DG_LAUNCH(XM_F32, false, false);
  } else {
    if (g.w4) MQ_DG_LAUNCH(XM_I8, false, true);
    else MQ_DG_LAUNCH(XM_I8, false, false);
  }
#undef MQ_DG_LAUNCH
  MQ_LAUNCH_CHECK("mq_decode_gemv");
  return MQ_OK;
}

int mq_decode_attention(const mq_decode_attention_args* args, mq_stream_t stream) {
  MQ_REQUIRE(args != nullptr, "mq_decode_attention: null argument block");
  const mq_decode_attention_args& a = *args;
  MQ_REQUIRE(a.qkv && a.k_cache && a.v_cache && a.cos && a.sin && a.pos && a.out, "mq_decode_attention: null pointer");
  MQ_REQUIRE(a.heads > 0 && a.kv_heads > 0 && a.heads % a.kv_heads == 0 && a.head_dim >= 16 && a.head_dim <= 256 && a.head_dim % 16 == 0 &&
                 a.cache_len > 0 && a.cache_len <= 16384,
             "mq_decode_attention: heads=%d kv_heads=%d head_dim=%d cache_len=%d", a.heads, a.kv_heads, a.head_dim, a.cache_len);
  decode_attention_kernel<<<(unsigned)a.heads, 256, (size_t)a.cache_len * sizeof(float), as_stream(stream)>>>(a, STAMP_SLOT(4, (unsigned)a.heads));
  MQ_LAUNCH_CHECK("mq_decode_attention");
  return MQ_OK;
}

int mq_decode_head(const float* x, const float* norm_weight, float eps, const float* w, const float* bias, int64_t K, int64_t V,
                   float* logits, mq_stream_t stream) {
  MQ_REQUIRE(x && w && logits && K > 0 && K % 4 == 0 && K <= 12288 && V > 0, "mq_decode_head: bad arguments (K %% 4 == 0, K <= 12288)");
  MQ_REQUIRE(aligned(w, 16), "mq_decode_head: the weight must be 16-byte aligned");
  int64_t blocks = (V + 3) / 4;
  if (blocks > 256 * 8) blocks = 256 * 8;
  decode_head_kernel<<<(unsigned)blocks, 256, (size_t)K * sizeof(float), as_stream(stream)>>>(x, norm_weight, eps, w, bias, (int)K, (int)V, logits,
                                                                                    STAMP_SLOT(5, (unsigned)blocks));
  MQ_LAUNCH_CHECK("mq_decode_head");
  return MQ_OK;
}

}  // extern "C"

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

// torch.clamp propagates NaN; so do v_maximum3_f32 / v_minimum3_f32 (gfx950) -- two instructions where fmaxf / fminf + a NaN select
// cost five (round 5: every quantizer of the decode step sits on a launch's critical path)
__device__ __forceinline__ float dq_clamp_nan(float q, float lo, float hi) {
  return __builtin_elementwise_minimum(__builtin_elementwise_maximum(q, lo), hi);
}
// (x / s: div_by_scale of mq_common.h -- at M = 1 every CU quantises the whole activation row, and that arithmetic is on the
// launch's critical path)
// qmodule.py:286-290.  round_ste = (round(t) - t) + t IS rint(t) in fp32 for every t div_by_scale returns (finite or NaN, never inf:
// mq_common.h image_idxf has the argument) -- two instructions fewer on the M = 1 critical path
__device__ __forceinline__ float dq_index(float x, float s, float inv_s, float o, float qmin, float qmax) {
  const float t = div_by_scale(x, s, inv_s);
  return dq_clamp_nan(__fadd_rn(rintf(t), o), qmin, qmax);
}
__device__ __forceinline__ float dq_dequant(float q, float s, float o) { return __fmul_rn(__fsub_rn(q, o), s); }

struct Grid {          // device view of mq_grid
  float s, o, qmin, qmax, inv_s;
  bool on;
  __device__ __forceinline__ float fq(float v) const { return on ? dq_dequant(dq_index(v, s, inv_s, o, qmin, qmax), s, o) : v; }
  // two elements per instruction where a packed form exists (v_pk_mul / v_pk_fma / v_pk_add are IEEE fp32 on register pairs: the same
  // bits as fq on each half; rint and the clamp stay scalar).  q - o == q + (-o) exactly.
  __device__ __forceinline__ v2f fq2(v2f v) const {
    if (!on) return v;
    const v2f t = div_by_scale2(v, s, inv_s);
    v2f r = {rintf(t.x), rintf(t.y)};
    r = r + splat2(o);
    r.x = dq_clamp_nan(r.x, qmin, qmax);
    r.y = dq_clamp_nan(r.y, qmin, qmax);
    return (r + splat2(-o)) * splat2(s);
  }
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
// float wave reductions on DPP moves (a __shfl_xor is an LDS round trip of ~100 cycles; six of them in a row cost ~0.25 us of
// a kernel that lasts 3): same lane pattern as wave_sum_dpp; the result is the value of lane 63, broadcast
template <int CTRL, int ROW_MASK, bool BOUND>
__device__ __forceinline__ float dpp_f(float v) {
  return __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), CTRL, ROW_MASK, 0xf, BOUND));
}
__device__ __forceinline__ float wave_sum_f(float v) {
  v += dpp_f<0xB1, 0xf, true>(v);
  v += dpp_f<0x4E, 0xf, true>(v);
  v += dpp_f<0x141, 0xf, true>(v);
  v += dpp_f<0x140, 0xf, true>(v);
  v += dpp_f<0x142, 0xa, false>(v);                                 // row_bcast15 into rows 1 and 3 (0 elsewhere: x + 0 = x)
  v += dpp_f<0x143, 0xc, false>(v);                                 // row_bcast31 into rows 2 and 3
  return __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, v), 63));
}
__device__ __forceinline__ float wave_max_f(float v) {              // inputs are finite or -inf, never NaN
  v = fmaxf(v, dpp_f<0xB1, 0xf, true>(v));
  v = fmaxf(v, dpp_f<0x4E, 0xf, true>(v));
  v = fmaxf(v, dpp_f<0x141, 0xf, true>(v));
  v = fmaxf(v, dpp_f<0x140, 0xf, true>(v));
  const float r1 = __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, v), 15));
  const float r2 = __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, v), 31));
  const float r3 = __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, v), 47));
  const float r4 = __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, v), 63));
  return fmaxf(fmaxf(r1, r2), fmaxf(r3, r4));
}

constexpr int DG_THREADS = 1024, DG_WAVES = 16, DG_INFLIGHT = 12;   // 12: the w1|w3 launch (22 row pairs per CU over 8 stream waves: 3 x 4 chunks per lane) has every load in flight at once

// Profiling builds (-DMQ_DECODE_STAMPS, tools/decode_stamps.py): every workgroup leaves s_memrealtime stamps (100 MHz, one clock for
// the whole chip) at its phase boundaries.  Production builds compile the stamps out; the pointer argument stays null.
#ifdef MQ_DECODE_STAMPS
#define DG_STAMP(k)                                                                                   \
  do {                                                                                                \
    if (stamps && threadIdx.x == 0) stamps[((size_t)blockIdx.y * gridDim.x + blockIdx.x) * 16 + (k)] = __builtin_amdgcn_s_memrealtime(); \
  } while (0)
// stamp of wave `w` once the register `reg` has arrived
#define DG_STAMP_ARRIVED(k, w, reg)                                                                   \
  do {                                                                                                \
    if (stamps && wave == (w)) {                                                                      \
      asm volatile("" ::"v"(reg));                                                                    \
      if (lane == 0) stamps[((size_t)blockIdx.y * gridDim.x + blockIdx.x) * 16 + (k)] = __builtin_amdgcn_s_memrealtime();       \
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
enum { CG_NORM_IN = 0, CG_A = 1, CG_OUT0 = 2, CG_OUT1 = 3, CG_OUT2 = 4, CG_GATE_MID = 5, CG_GATE_ACTOUT = 6, CG_GATE_OUT = 7, CG_COUNT = 8,
       CG_O_OUT = 8, CG_COUNT_R6 = 9 };   // round 6: slot 8 = o_proj's output grid (OPRE prologue; zero when unused)

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
//   XM_LNORM: QLayerNorm.forward (qmodule.py:624-640 around F.layer_norm; StableLM-2's norm) instead of QRMSNorm: mean, biased variance,
//     y = (xi * rstd + (-rstd * mean)) * gamma + beta -- the arithmetic of mq_layernorm_quant; three barriers instead of two.
enum { XM_NORM = 0, XM_F32 = 1, XM_I8 = 2, XM_LNORM = 3 };
constexpr int DG_PRO = 8, DG_STR = DG_WAVES - DG_PRO, DG_XPRE = 4;      // 512 prologue threads x 4 float4 -> K <= 8192 (fp32)

// PAIR_GATE: two consecutive weight rows (w1 row i, w3 row i) -> the gated activation (above).
// OPRE (round 6; the w1 | w3 stream): the activation row is not x but x + Qo(alpha (acc + ct) + bias): o_proj's epilogue and the
//   residual add on the integer sums mq_decode_attention_oproj left in o_acc -- the o_proj launch is gone.  Each workgroup also stores
//   a share of that row to x_mid (the residual input of the w2 launch).
// zero_acc (round 6; the q | k | v stream): the launch clears the o_proj accumulators the NEXT launch adds into.
// (Tried and dropped in round 6: RoPE + the QMatMul input quantizers + the cache append as this launch's epilogue on rotation-partner
// row pairs -- bit-identical, but one row pair per LANE is a 250-instruction serial chain at the launch's tail: +0.9 us on the q | k | v
// launch against 0.15 us of 64-wide arithmetic in the attention launch; profiles/r06/decode_stamps_L4_rope_epilogue.log.)
enum { PAIR_NONE = 0, PAIR_GATE = 1 };

template <int XMODE, int PAIR, bool W4, bool OPRE>
__global__ void __launch_bounds__(DG_THREADS) decode_gemv_kernel(const mq_decode_gemv_args g, const int rows_per_wg, unsigned long long* stamps) {
  DG_STAMP(0);
  constexpr bool GATE = PAIR == PAIR_GATE, PAIRED = PAIR != PAIR_NONE;
  constexpr int XPRE = OPRE ? 2 : DG_XPRE;                         // OPRE holds four more vectors per element: K <= 4096 there
  extern __shared__ __attribute__((aligned(16))) char smem[];   // [K] int8 activation image
  __shared__ float s_red[DG_PRO], s_red2[DG_PRO];
  __shared__ int s_redi[DG_PRO];
  const int lane = threadIdx.x & 63;
  const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const int K = g.K;
  const float cv = g.consts[lane & 63];                            // one 256-byte line per wave, both roles

  if (wave < DG_PRO) {
    // ================================================ PROLOGUE role ====================================================================
    // (s_setprio 3 for these waves -- the launch's critical chain, sharing SIMDs with the stream waves -- measured neutral: round 6)
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
      constexpr bool ANYNORM = XMODE == XM_NORM || XMODE == XM_LNORM;
      float4 xv[XPRE], nw[XPRE], nb[XPRE];
      typedef int v4i_t __attribute__((ext_vector_type(4)));
      v4i_t oa[OPRE ? XPRE : 1], oc[OPRE ? XPRE : 1];
      float4 oal[OPRE ? XPRE : 1], ob[OPRE ? XPRE : 1], hmid[OPRE ? XPRE : 1];
#pragma unroll
      for (int u = 0; u < XPRE; ++u) {
        if (u * DG_PRO * 64 < nvec) {                              // wave-uniform
          const int i = p + u * DG_PRO * 64;
          const int ic = i < nvec ? i : nvec - 1;
          xv[u] = reinterpret_cast<const float4*>(g.x)[ic];
          if constexpr (OPRE) {
            oa[u] = reinterpret_cast<const v4i_t*>(g.o_acc)[ic];
            oal[u] = reinterpret_cast<const float4*>(g.o_alpha)[ic];
            oc[u] = reinterpret_cast<const v4i_t*>(g.o_ct)[ic];
            ob[u] = g.o_bias ? reinterpret_cast<const float4*>(g.o_bias)[ic] : make_float4(0.f, 0.f, 0.f, 0.f);
          }
          if constexpr (ANYNORM) nw[u] = reinterpret_cast<const float4*>(g.norm_w)[ic];
          if constexpr (XMODE == XM_LNORM)
            nb[u] = g.norm_bias ? reinterpret_cast<const float4*>(g.norm_bias)[ic] : make_float4(0.f, 0.f, 0.f, 0.f);
        }
      }
      DG_STAMP_ARRIVED(6, 0, xv[0].x);
      const Grid ag = const_grid(cv, CG_A, g.a_grid);            // (the constants were requested before the row: they are here)
      if constexpr (OPRE) {
        // o_proj's epilogue, op for op that of the RESID launch it replaces (one row per lane there, four elements per thread here):
        // tt = sum - zp rs + ct (o_acc already holds sum - zp rs), e = float(tt) alpha + bias, Qo in the GEMV epilogues' reciprocal form,
        // resid + (.)
        const Grid oo = const_grid(cv, CG_O_OUT, g.o_out);
        auto oq1 = [&](float x0, int acc, int ct, float al, float bi) {
          const int tt = (int)((unsigned)acc + (unsigned)ct);
          float f = __fadd_rn(__fmul_rn((float)tt, al), bi);
          if (oo.on) {
            float v = rintf(f * oo.inv_s) + oo.o;
            v = fminf(fmaxf(v, oo.qmin), oo.qmax);
            f = __fmul_rn(__fsub_rn(v, oo.o), oo.s);
          }
          return __fadd_rn(x0, f);
        };
#pragma unroll
        for (int u = 0; u < XPRE; ++u) {
          if (u * DG_PRO * 64 < nvec) {
            float4& v = xv[u];
            v = make_float4(oq1(v.x, oa[u][0], oc[u][0], oal[u].x, ob[u].x), oq1(v.y, oa[u][1], oc[u][1], oal[u].y, ob[u].y),
                            oq1(v.z, oa[u][2], oc[u][2], oal[u].z, ob[u].z), oq1(v.w, oa[u][3], oc[u][3], oal[u].w, ob[u].w));
            hmid[u] = v;                                           // stored to x_mid at the END of the prologue: a store in front of the norm
          }                                                        // weights' first use makes hipcc wait vmcnt(0) there, i.e. for the store (+0.8 us)
        }
      }
      float r = 1.f, shiftv = 0.f;
      if constexpr (XMODE == XM_LNORM) {                           // QLayerNorm.forward, arithmetic of mq_layernorm_quant (mq_norm.hip)
        const Grid ng = const_grid(cv, CG_NORM_IN, g.norm_in);
        float s1 = 0.f;
#pragma unroll
        for (int u = 0; u < XPRE; ++u) {
          if (u * DG_PRO * 64 < nvec) {
            float4& v = xv[u];
            const v2f lo2 = ng.fq2((v2f){v.x, v.y}), hi2 = ng.fq2((v2f){v.z, v.w});
            v = make_float4(lo2.x, lo2.y, hi2.x, hi2.y);
            if (p + u * DG_PRO * 64 < nvec) s1 += (v.x + v.y) + (v.z + v.w);
          }
        }
        s1 = wave_sum_f(s1);
        if (lane == 0) s_red[wave] = s1;
        __syncthreads();                                           // barrier 1 of 3
        float tot = 0.f;
#pragma unroll
        for (int w = 0; w < DG_PRO; ++w) tot += s_red[w];
        const float mu = __fdiv_rn(tot, (float)K);
        float s2 = 0.f;
#pragma unroll
        for (int u = 0; u < XPRE; ++u) {
          if (u * DG_PRO * 64 < nvec && p + u * DG_PRO * 64 < nvec) {
            const float4 v = xv[u];
            const float d0 = v.x - mu, d1 = v.y - mu, d2 = v.z - mu, d3 = v.w - mu;
            s2 += d0 * d0;
            s2 += d1 * d1;
            s2 += d2 * d2;
            s2 += d3 * d3;
          }
        }
        s2 = wave_sum_f(s2);
        if (lane == 0) s_red2[wave] = s2;
        __syncthreads();                                           // barrier 2 of 3
        float tot2 = 0.f;
#pragma unroll
        for (int w = 0; w < DG_PRO; ++w) tot2 += s_red2[w];
        const float var = __fdiv_rn(tot2, (float)K);
        r = __fdiv_rn(1.0f, __fsqrt_rn(__fadd_rn(var, g.eps)));
        shiftv = __fmul_rn(-r, mu);
      }
      if constexpr (XMODE == XM_NORM) {                            // QRMSNorm.forward (qmodule.py:515-531), arithmetic of mq_rmsnorm_quant
        const Grid ng = const_grid(cv, CG_NORM_IN, g.norm_in);
        float ss = 0.f;
#pragma unroll
        for (int u = 0; u < XPRE; ++u) {
          if (u * DG_PRO * 64 < nvec) {
            float4& v = xv[u];
            const v2f lo2 = ng.fq2((v2f){v.x, v.y}), hi2 = ng.fq2((v2f){v.z, v.w});
            v = make_float4(lo2.x, lo2.y, hi2.x, hi2.y);
            if (p + u * DG_PRO * 64 < nvec) {
              const v2f sl = lo2 * lo2, sh = hi2 * hi2;             // the squares in pairs, the sum in the reference kernel's order
              ss += sl.x;
              ss += sl.y;
              ss += sh.x;
              ss += sh.y;
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
      for (int u = 0; u < XPRE; ++u) {
        if (u * DG_PRO * 64 < nvec) {
          const int i = p + u * DG_PRO * 64;
          float4 v = xv[u];
          if constexpr (XMODE == XM_NORM) {
            const float4 w = nw[u];
            const v2f a2 = (v2f){w.x, w.y} * ((v2f){v.x, v.y} * splat2(r)), b2 = (v2f){w.z, w.w} * ((v2f){v.z, v.w} * splat2(r));
            v = make_float4(a2.x, a2.y, b2.x, b2.y);
          }
          if constexpr (XMODE == XM_LNORM) {
            const float4 w = nw[u], b = nb[u];
            v.x = __fadd_rn(__fmul_rn(__fadd_rn(__fmul_rn(v.x, r), shiftv), w.x), b.x);
            v.y = __fadd_rn(__fmul_rn(__fadd_rn(__fmul_rn(v.y, r), shiftv), w.y), b.y);
            v.z = __fadd_rn(__fmul_rn(__fadd_rn(__fmul_rn(v.z, r), shiftv), w.z), b.z);
            v.w = __fadd_rn(__fmul_rn(__fadd_rn(__fmul_rn(v.w, r), shiftv), w.w), b.w);
          }
          // the 8-bit unsigned activation grid's image bytes (index - 128): mq_common.h's packed form -- NaN -> qmin through v_med3,
          // v_cvt_pk_u8_f32 converts and packs, v_sad_u8 sums the four indices (as the image-only prefill kernels)
          const v2f u01 = image_u8f2((v2f){v.x, v.y}, ag.s, ag.inv_s, ag.o, ag.qmin, ag.qmax, 0.f);
          const v2f u23 = image_u8f2((v2f){v.z, v.w}, ag.s, ag.inv_s, ag.o, ag.qmin, ag.qmax, 0.f);
          uint32_t usum = 0;
          const uint32_t pk = image_pack4(u01.x, u01.y, u23.x, u23.y, usum);
          if (i < nvec) {
            my_sum += (int)usum - 512;
            reinterpret_cast<unsigned*>(smem)[i] = pk;
          }
        }
      }
      if constexpr (OPRE) {
#pragma unroll
        for (int u = 0; u < XPRE; ++u) {
          const int i = p + u * DG_PRO * 64;
          if (u * DG_PRO * 64 < nvec && i < nvec && (unsigned)i % gridDim.x == blockIdx.x) reinterpret_cast<float4*>(g.x_mid)[i] = hmid[u];   // this workgroup's share
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
  const int NL = PAIRED ? g.N >> 1 : g.N;                          // logical rows
  const int kchunks = W4 ? K >> 5 : K >> 4;                        // 16-byte chunks per weight row
  const int wrow = W4 ? K >> 1 : K;                                // bytes per weight row
  const int lchunks = PAIRED ? 2 * kchunks : kchunks;              // chunks per logical row
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
        buf[u] = __builtin_nontemporal_load(reinterpret_cast<const v4i*>(g.w + (size_t)row * (PAIRED ? 2 : 1) * wrow) + c);
      else
        buf[u] = v4i{0, 0, 0, 0};
      if (++j == cpl) { j = 0; ++t; }
    }
  };
  issue_pass(0, 0);
  float p_alpha[PAIRED ? 2 : 1], p_bias[PAIRED ? 2 : 1];
  int p_zp[PAIRED ? 2 : 1], p_ct[PAIRED ? 2 : 1];
#pragma unroll
  for (int h = 0; h < (PAIRED ? 2 : 1); ++h) {
    const int wr = PAIRED ? 2 * prow + h : prow;
    p_alpha[h] = prow_ok ? g.alpha[wr] : 0.f;
    p_zp[h] = prow_ok ? g.w_zp[wr] : 0;
    p_ct[h] = prow_ok ? g.col_term[wr] : 0;
    p_bias[h] = (prow_ok && g.bias) ? g.bias[wr] : 0.f;
  }
  float p_res = 0.f;
  if (!PAIRED && g.resid && prow_ok) p_res = g.resid[prow];
  if (g.zero_acc && sw == 0) {                                     // clear this workgroup's share of the o_proj accumulators (the next launch adds)
    const int per = (g.zero_n + (int)gridDim.x - 1) / (int)gridDim.x, lo = (int)blockIdx.x * per;
    const int hi = lo + per < g.zero_n ? lo + per : g.zero_n;
    for (int i = lo + lane; i < hi; i += 64) g.zero_acc[i] = 0;
  }
  if constexpr (XMODE == XM_LNORM) {                               // the prologue waves' mean and variance reductions
    __syncthreads();
    __syncthreads();
  }
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
        const int ck = PAIRED ? (c >= kchunks ? c - kchunks : c) : c;
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
        if (PAIRED && c >= kchunks) acc1 += part;
        else acc0 += part;
        if (j2 == cpl - 1) {                                     // logical row slot t2 complete
          const int s0 = wave_sum_dpp(acc0), s1 = PAIRED ? wave_sum_dpp(acc1) : 0;
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
    if constexpr (PAIRED) {
      const int tt = (int)((unsigned)sum1 - (unsigned)p_zp[PAIRED ? 1 : 0] * (unsigned)rs + (unsigned)p_ct[PAIRED ? 1 : 0]);
      e1 = __fadd_rn(__fmul_rn((float)tt, p_alpha[PAIRED ? 1 : 0]), p_bias[PAIRED ? 1 : 0]);
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

// ---- attention for ONE query token over a static INTEGER KV cache ------------------------------------------------------------------
// hf_model.py:486-534 with the QMatMul pair of qmodule.py:453-466, as the prefill kernel computes it (mq_attention.hip): the two
// contractions are exact integer sums over the quantizer indices, everything else is the reference's fp32 arithmetic in its exact
// (divide) form:
//   q, k_new <- RoPE (rotate-half over the first rot_dim dims);  cache[pos] <- (index of k_new on qk_b, index of v_new on pv_b)
//   s[t] = Qqk_out( s_q s_k sum_d (iq[d] - zq)(ik[t][d] - zk) ) / sqrt(D),   t = 0 .. pos        (the mask admits exactly these)
//   p = softmax(s)  (fp32: exp(s - max) / sum)
//   o[d] = Qpv_out( s_p s_v sum_t (ip[t] - zp)(iv[t][d] - zv) )  ->  index on the consumer linear's input grid (o_proj's int8 image)
// The cache holds the keys / values as int8 indices (index - 128) ON THEIR QMatMul input grids (qk_bmm.input2, pv_bmm.input2): the
// reference re-quantises the whole cached tensor at every step with static grids, which is idempotent, so quantising once at append
// time gives the same numbers -- and a quarter of the bytes.
// Grid: heads x nsplit workgroups of 256 threads.  Every workgroup of a head computes ALL scores of that head (t x D bytes of keys,
// bit-identical statistics in every split, no exchange), then the p.v sum over ITS 64-position blocks (b = split, split + nsplit,
// ...).  nsplit == 1: the workgroup finishes the head.  nsplit > 1: exact int64 partial sums go to `part` with write-through
// stores, a per-head ticket counts the splits, the last one adds the partials (integers: any order is THE sum) and finishes.
enum { AG_QK_A = 0, AG_QK_B = 1, AG_QK_OUT = 2, AG_PV_A = 3, AG_PV_B = 4, AG_PV_OUT = 5, AG_O_IN = 6, AG_COUNT = 7 };

// sum over aligned groups of N = 2 / 4 adjacent lanes: DPP quad permutes (a __shfl_xor is an LDS round trip, ~100 cycles each)
template <int N>
__device__ __forceinline__ int quad_sum(int v) {
  if (N == 1) return v;
  v += __builtin_amdgcn_update_dpp(0, v, 0xB1, 0xf, 0xf, true);                 // quad_perm [1,0,3,2]
  if (N == 4) v += __builtin_amdgcn_update_dpp(0, v, 0x4E, 0xf, 0xf, true);     // quad_perm [2,3,0,1]
  return v;
}

// D = head_dim (compile time: every index below is a shift).  Scores: LPP lanes per cached position (4; 2 at D = 32), each with CH
// 16-byte chunks of the key row (LPP * CH * 16 = D); 256 / LPP positions per pass, KB passes in flight.  p.v: a thread owns one dword (4 dims) of the
// value rows of G = 1024 / D position groups; a 64-position block gives each thread PPB = D / 16 positions.
template <int D>
__global__ void __launch_bounds__(256) decode_attention_kernel(const mq_decode_attention_args a, unsigned long long* stamps) {
  constexpr int LPP = D >= 64 ? 4 : 2, CH = D >= 64 ? D / 64 : 1, PPP = 256 / LPP, KB = 8 / CH;
  constexpr int DQ = D / 4, G = 256 / DQ, PPB = 64 / G, VB = 16;
  static_assert(PPB * G == 64 && VB % PPB == 0, "block mapping");
  extern __shared__ __attribute__((aligned(16))) char smem_raw[];
  float* s_sc = reinterpret_cast<float*>(smem_raw);              // [cache_len] scores -> exp -> (p index - zp) as int
  __shared__ __attribute__((aligned(16))) int8_t s_q8[D], s_k8[D], s_v8[D];
  __shared__ float s_redf[4];
  __shared__ int s_redq[4];
  __shared__ long long s_acc[1024];                              // [G][D] partial p.v sums
  __shared__ unsigned s_ticket;
  const int H = a.heads, rot = a.rot_dim, nsplit = a.nsplit;
  if ((int)blockIdx.y >= nsplit) {
    // ---- L2 prefetch role (grid rows behind the attention's): this launch keeps 32 .. 128 of 256 CUs busy and moves a few hundred
    // KB, so the memory fabric idles for its whole duration -- and the w1|w3 launch two steps down the chain is the step's biggest
    // stream (23 MB, 4 us at the fabric's ~5.5 TB/s).  Workgroup q of these rows reads exactly what workgroup (q + first) % n of that
    // launch will read, with default-policy loads, into the L2 of the XCD both run on under the observed (linear id % 8) placement:
    // speed only, never correctness.  Tried first as a kernel on a parallel graph branch: the cross-queue dependencies cost 3 - 14 us
    // per edge (1.28 ms / token instead of 0.68); rows of the SAME launch cost nothing.
    const int q = ((int)blockIdx.y - nsplit) * H + (int)blockIdx.x;
    if (q >= a.prefetch_wgs) return;
    // the attention workgroups' own requests (position -> cos / sin, keys, values: a dependent chain) go first: this stream starts
    // prefetch_delay x 10 ns into the launch (measured without it: the attention's first data arrived 2.3 us late)
    const unsigned long long t_go = __builtin_amdgcn_s_memrealtime() + (unsigned long long)a.prefetch_delay;
    while (__builtin_amdgcn_s_memrealtime() < t_go) __builtin_amdgcn_s_sleep(8);
    const int gwg = (q + nsplit * H) % a.prefetch_wgs;             // same linear id % 8 as this workgroup when prefetch_wgs % 8 == 0
    const size_t beg = (size_t)gwg * a.prefetch_stride;
    const size_t end = beg + a.prefetch_bytes_per_wg < a.prefetch_total ? beg + a.prefetch_bytes_per_wg : a.prefetch_total;
    const v4i* p = reinterpret_cast<const v4i*>(a.prefetch + beg);
    const size_t n = end > beg ? (end - beg) >> 4 : 0;
    v4i acc = {0, 0, 0, 0};
    for (size_t i = threadIdx.x; i < n; i += 256 * 8) {
      v4i b[8];
#pragma unroll
      for (int u = 0; u < 8; ++u) b[u] = p[i + (size_t)u * 256 < n ? i + (size_t)u * 256 : i];
#pragma unroll
      for (int u = 0; u < 8; ++u) acc |= b[u];
    }
    asm volatile("" ::"v"(acc));
    return;
  }
  DG_STAMP(0);
  const int h = blockIdx.x, c = blockIdx.y, kvh = h / (H / a.kv_heads);
  const int tid = threadIdx.x, lane = tid & 63, wv = tid >> 6;
  const float cv = a.consts[lane];
  const int pos = a.pos[0];
  // the new token's q / k / v rows (fp32 outputs of the q|k|v launch) and their RoPE partners: independent of the position
  const float* qp = a.qkv + (size_t)h * D;
  const float* kp = a.qkv + (size_t)H * D + (size_t)kvh * D;
  const float* vp = a.qkv + (size_t)(H + a.kv_heads) * D + (size_t)kvh * D;
  const int dd = tid < D ? tid : D - 1;
  const int half = rot >> 1;
  const int dpart = dd < rot ? (dd < half ? dd + half : dd - half) : dd;
  const float q_raw = qp[dd], q_par = qp[dpart], k_raw = kp[dd], k_par = kp[dpart], v_raw = vp[dd];
  if (pos < 0 || pos >= a.cache_len) return;                       // a step past the cache: nothing is read or written (the host raises first)
  const int T = pos + 1;
  const int dr = dd < rot ? dd : 0;
  const float cs = a.cos[(size_t)pos * rot + dr], sn = a.sin[(size_t)pos * rot + dr];
  const int8_t* kc = a.k_cache + (size_t)kvh * a.cache_len * D;
  const int8_t* vc = a.v_cache + (size_t)kvh * a.cache_len * D;
  // ---- key loads of the first batch ------------------------------------------------------------------------------------------------
  const int sub = tid & (LPP - 1), slot = tid / LPP;
  v4i kbuf[KB][CH];
  auto load_keys = [&](int t0) {
#pragma unroll
    for (int u = 0; u < KB; ++u) {
      const int t = t0 + u * PPP + slot;
      const int tc = (t < T && t != pos) ? t : 0;                   // position 0 stands in (always valid memory); masked below
#pragma unroll
      for (int ch = 0; ch < CH; ++ch) kbuf[u][ch] = *reinterpret_cast<const v4i*>(kc + (size_t)tc * D + (sub * CH + ch) * 16);
    }
  };
  load_keys(0);
  // ---- the value loads of this split's first blocks: thread (dq = dword of 4 dims, grp): positions 64 b + grp + G j ------------
  const int dq = tid & (DQ - 1), grp = tid / DQ;
  int vbuf[VB];
  auto item_pos = [&](int i) { return 64 * (c + nsplit * (i / PPB)) + grp + G * (i % PPB); };   // PPB: a power of two (shifts)
  auto load_values = [&](int i0) {                                   // cached positions t < pos only: the new one stays in registers
#pragma unroll
    for (int u = 0; u < VB; ++u) {
      const int t = item_pos(i0 + u);
      vbuf[u] = *reinterpret_cast<const int*>(vc + (size_t)(t < pos ? t : 0) * D + dq * 4);
    }
  };
  load_values(0);
  // ---- RoPE + the three input quantizers of the new token ---------------------------------------------------------------------------
  const Grid qa = const_grid(cv, AG_QK_A, a.qk_a), qb = const_grid(cv, AG_QK_B, a.qk_b), qo = const_grid(cv, AG_QK_OUT, a.qk_out);
  const Grid pa = const_grid(cv, AG_PV_A, a.pv_a), pb = const_grid(cv, AG_PV_B, a.pv_b), po = const_grid(cv, AG_PV_OUT, a.pv_out);
  const Grid oi = const_grid(cv, AG_O_IN, a.o_in);
  int qsum_part = 0;
  if (tid < D) {
    float qv = q_raw, kv = k_raw;
    if (tid < rot) {                                               // x * cos + rot(x) * sin, rot(x)[d] = d < rot/2 ? -x[d + rot/2] : x[d - rot/2]
      const float sg = tid < half ? -1.f : 1.f;                    // (-x) * sin == -(x * sin) exactly
      qv = __fadd_rn(__fmul_rn(q_raw, cs), __fmul_rn(sg * q_par, sn));
      kv = __fadd_rn(__fmul_rn(k_raw, cs), __fmul_rn(sg * k_par, sn));
    }
    const float iq = dq_index(qv, qa.s, qa.inv_s, qa.o, qa.qmin, qa.qmax), ik = dq_index(kv, qb.s, qb.inv_s, qb.o, qb.qmin, qb.qmax);
    const float iv = dq_index(v_raw, pb.s, pb.inv_s, pb.o, pb.qmin, pb.qmax);
    const int sq = (iq != iq ? 0 : (int)iq) - 128, sk = (ik != ik ? 0 : (int)ik) - 128, sv = (iv != iv ? 0 : (int)iv) - 128;
    s_q8[tid] = (int8_t)sq;
    s_k8[tid] = (int8_t)sk;
    s_v8[tid] = (int8_t)sv;
    qsum_part = sq;
    if (c == 0 && h % (H / a.kv_heads) == 0) {                     // the group's first head appends to the cache
      a.k_cache[((size_t)kvh * a.cache_len + pos) * D + tid] = (int8_t)sk;
      a.v_cache[((size_t)kvh * a.cache_len + pos) * D + tid] = (int8_t)sv;
    }
  }
  {
    const int w = wave_sum_dpp(qsum_part);
    if (lane == 0) s_redq[wv] = w;
  }
  __syncthreads();
  DG_STAMP(1);
  const int qsum = (s_redq[0] + s_redq[1]) + (s_redq[2] + s_redq[3]);
  const int zq = (int)qa.o - 128, zk = (int)qb.o - 128, zv = (int)pb.o - 128, zp = (int)pa.o;
  const float alpha_qk = __fmul_rn(qa.s, qb.s), alpha_pv = __fmul_rn(pa.s, pb.s);
  const int qconst = D * zq * zk - zk * qsum;                      // sum (iq - zq)(ik - zk) = sum sq sk - zk sum sq - zq sum sk + D zq zk
  constexpr bool pow2 = (D == 64 || D == 256);                     // sqrt(D) a power of two: the divide is an exact multiply
  const float sqrt_d = __fsqrt_rn((float)D), inv_sqrt_d = 1.0f / (D == 64 ? 8.0f : 16.0f);
  v4i qf[CH], kn[CH];                                              // this lane's share of the query / of the NEW key (never via memory)
#pragma unroll
  for (int ch = 0; ch < CH; ++ch) {
    qf[ch] = *reinterpret_cast<const v4i*>(s_q8 + (sub * CH + ch) * 16);
    kn[ch] = *reinterpret_cast<const v4i*>(s_k8 + (sub * CH + ch) * 16);
  }
  const v4i ones = {0x01010101, 0x01010101, 0x01010101, 0x01010101};
  // ---- scores ----------------------------------------------------------------------------------------------------------------------
  float lmax = -INFINITY;
  for (int t0 = 0; t0 < T; t0 += KB * PPP) {
    if (t0 > 0) load_keys(t0);
#pragma unroll
    for (int u = 0; u < KB; ++u) {
      if (t0 + u * PPP >= T) break;                                // (uniform) the rest of the batch lies beyond the sequence
      const int t = t0 + u * PPP + slot;
      int dot = 0, ks = 0;
#pragma unroll
      for (int ch = 0; ch < CH; ++ch) {
        const v4i kf = t == pos ? kn[ch] : kbuf[u][ch];
        dot = dot16(kf, qf[ch], dot);
        ks = dot16(kf, ones, ks);
      }
      dot = quad_sum<LPP>(dot);
      ks = quad_sum<LPP>(ks);
      if (t < T && sub == 0) {
        const int ti = dot - zq * ks + qconst;
        const float val = __fmul_rn((float)ti, alpha_qk);
        const float qv = qo.fq(val);
        const float sc = pow2 ? __fmul_rn(qv, inv_sqrt_d) : __fdiv_rn(qv, sqrt_d);     // qk_bmm(...) / sqrt(head_dim)  (hf_model.py:513)
        s_sc[t] = sc;
        lmax = fmaxf(lmax, sc);
      }
    }
  }
  lmax = wave_max_f(lmax);
  if (lane == 0) s_redf[wv] = lmax;
  __syncthreads();
  DG_STAMP(2);
  const float mx = fmaxf(fmaxf(s_redf[0], s_redf[1]), fmaxf(s_redf[2], s_redf[3]));
  __syncthreads();
  float lsum = 0.f;
  for (int t = tid; t < T; t += 256) {
    const float e = expf(s_sc[t] - mx);
    s_sc[t] = e;
    lsum += e;
  }
  lsum = wave_sum_f(lsum);
  if (lane == 0) s_redf[wv] = lsum;
  __syncthreads();
  const float tot_e = (s_redf[0] + s_redf[1]) + (s_redf[2] + s_redf[3]);
  // pv_bmm's input quantizer, once per position of THIS split's blocks: (index - zp) as int
  int* s_pi = reinterpret_cast<int*>(s_sc);
  for (int t = tid; t < T; t += 256) {
    if (nsplit == 1 || ((t >> 6) % nsplit) == c) {
      const float p = __fdiv_rn(s_sc[t], tot_e);
      const float ip = dq_index(p, pa.s, pa.inv_s, pa.o, pa.qmin, pa.qmax);
      s_pi[t] = (ip != ip ? 0 : (int)ip) - zp;
    }
  }
  __syncthreads();
  DG_STAMP(3);
  // ---- p.v over this split's blocks: exact integers ------------------------------------------------------------------------------
  // sum_t pi[t] (vs[t][d] - zv) = sum_t pi[t] vs[t][d] - zv sum_t pi[t]: v_bfe_i32 + v_mad_i32_i24 per element, one add per position
  long long acc[4] = {0, 0, 0, 0};
  long long psum = 0;
  const int nblk = (pos + 63) >> 6;                                  // blocks of CACHED positions 0 .. pos - 1
  const int my_blocks = c < nblk ? (nblk - 1 - c) / nsplit + 1 : 0;
  const int items = my_blocks * PPB;
  for (int i0 = 0; i0 < items; i0 += VB) {
    if (i0 > 0) load_values(i0);
    int a32[4] = {0, 0, 0, 0}, p32 = 0;                              // <= 16 positions x 65535 x 128 < 2^31
#pragma unroll
    for (int u = 0; u < VB; ++u) {
      const int t = item_pos(i0 + u);
      const bool ok = i0 + u < items && t < pos;
      const int pi = s_pi[ok ? t : 0];
      const int pim = ok ? pi : 0;
      p32 += pim;
#pragma unroll
      for (int e = 0; e < 4; ++e) a32[e] += (int)__builtin_amdgcn_sbfe(vbuf[u], 8 * e, 8) * pim;      // (the builtin returns unsigned)
    }
#pragma unroll
    for (int e = 0; e < 4; ++e) acc[e] += a32[e];
    psum += p32;
  }
  if (grp == 0 && ((pos >> 6) % nsplit) == c) {                      // the new position: its split's group 0 adds it from registers
    const int sv4 = *reinterpret_cast<const int*>(s_v8 + dq * 4);
    const int pi = s_pi[pos];
    psum += pi;
#pragma unroll
    for (int e = 0; e < 4; ++e) acc[e] += (long long)((int)__builtin_amdgcn_sbfe(sv4, 8 * e, 8) * pi);
  }
#pragma unroll
  for (int e = 0; e < 4; ++e) acc[e] -= (long long)zv * psum;
#pragma unroll
  for (int e = 0; e < 4; ++e) s_acc[grp * D + dq * 4 + e] = acc[e];
  __syncthreads();
  DG_STAMP(4);
  long long tot = 0;
  if (tid < D) {
#pragma unroll
    for (int gq = 0; gq < G; ++gq) tot += s_acc[gq * D + tid];
  }
  if (nsplit > 1) {
    // publish this split's exact partial sums write-through, count the head's splits; the last one adds them (cdna guide G16 R1)
    if (tid < D) __hip_atomic_store(reinterpret_cast<unsigned long long*>(a.part) + ((size_t)c * H + h) * D + tid, (unsigned long long)tot,
                                    __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    // RELEASE at agent scope: the partials above are already write-through (8-byte agent atomics, drained by the vmcnt wait
    // and ordered for the whole workgroup by the barrier), which is what makes this correct on gfx950; the release gives the formal
    // release -> acquire edge to the last split's fence below as well.  One L2 write-back per split workgroup, on the long-context path only.
    if (tid == 0) s_ticket = __hip_atomic_fetch_add(a.ticket + h, 1u, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_AGENT);
    __syncthreads();
    if (s_ticket != (unsigned)(nsplit - 1)) return;
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
    if (tid == 0) __hip_atomic_store(a.ticket + h, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);   // ready for the next launch
    if (tid < D) {
      tot = 0;
      for (int cc = 0; cc < nsplit; ++cc)
        tot += (long long)__hip_atomic_load(reinterpret_cast<unsigned long long*>(a.part) + ((size_t)cc * H + h) * D + tid, __ATOMIC_RELAXED,
                                            __HIP_MEMORY_SCOPE_AGENT);
    }
  }
  if (tid < D) {
    const float pre = (float)((double)tot * (double)alpha_pv);     // one rounding of the exact sum (as mq_attention.hip)
    const float y = po.fq(pre);
    if (a.out) a.out[(size_t)h * D + tid] = y;
    if (a.out_q) {
      const float qi = dq_index(y, oi.s, oi.inv_s, oi.o, oi.qmin, oi.qmax);
      a.out_q[(size_t)h * D + tid] = (int8_t)((qi != qi ? (int)oi.qmin : (int)qi) - 128);
    }
  }
#ifdef MQ_DECODE_STAMPS
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  DG_STAMP(5);
#endif
}

// ---- round 6: attention of one query token + o_proj's contraction in ONE launch ---------------------------------------------------
// Why.  A layer at M = 1 is a chain of all-to-all dependences; every link costs a kernel boundary (1.35 us) plus a launch's ramp, its
// dependent first load and its tail -- 4.8 us for the o_proj launch, whose own work (4 MB of weights) is 0.8 us.  o_proj is linear in its
// input, and its input is the concatenation of the heads' outputs: workgroup (h, c) -- head h, row range c -- holds head h's output
// and can contract it with the K-slice [h D, (h + 1) D) of ITS rows on the spot.  The partial sums are integers, so adding them across
// the 32 heads with device-scope atomics is exact and order free (tools/atomic_probe.cpp: 65 536 no-return int32 atomics on 2 048
// addresses add 0.6-0.7 us to a launch, profiles/r06/atomic_probe.log); o_proj's epilogue -- which needs the COMPLETE sums -- moves into
// the prologue of the next launch (OPRE above).  The price: all `slices` workgroups of a head repeat the head's attention (they all need
// the whole output vector and there is no cheaper way to share it than to recompute it: a cross-workgroup hand-off costs ~3 us on this
// chip): at M = 1 that arithmetic is free, the 256 CUs were idle in the old launch (32 workgroups).
// Grid: heads * slices workgroups of 256 threads (+ prefetch rows); workgroups that share a KV head sit on the same XCD(s) where the
// geometry allows (speed only).  RoPE / quantizers / scores / softmax / p.v: the expressions of decode_attention_kernel with nsplit = 1
// (every result bit is that kernel's).  What differs is the ORDER OF REQUESTS, because this launch is a chain of dependent round trips:
//   * nothing requested at the top depends on *pos (first-batch addresses are clamped by the cache length), and *pos itself comes
//     through the scalar cache behind them: as a vector load with an early return hipcc hoisted load + wait + branch in front of
//     every other request -- one more round trip at the head of the launch;
//   * TWO batches of values (512 positions) are in registers before the scores start, later batches are refilled a batch ahead;
//   * o_proj's 16 KB per workgroup are requested behind the scores (hipcc waits vmcnt(0) at the head of the score loop: weights from
//     HBM in front of it held the scores back by ~1.3 us) and are in registers long before the head's output exists.
// o_proj: thread (r = tid / tpr, sub = tid % tpr) holds cpt = D / 16 / tpr 16-byte chunks of row c R + r of head h's slice.
// Threads per workgroup.  256 (one wave per SIMD) is the measured optimum: the phases of this launch are bound by the instructions
// every WAVE executes (grids, addresses, reductions, barriers), not by per-position work, so 1024 threads quadruple that overhead on
// the same four SIMDs: 1 668 tok/s at 256 threads against 1 505 at 1 024 (profiles/r06/decode_stamps_L4_1024_threads.log).
constexpr int AO_THREADS = 256;
// (Round 6, tried and removed: a second instantiation for long caches that keeps 2 048 positions' keys / values in registers -- the
// keys requested behind *pos, the later value chunks behind the scores -- so that the sweeps are straight-line code without an exposed
// round trip per 512 positions.  Slower at every context: 1 500 / 1 408 / 1 159 tok/s at 512 / 1 024 / 2 048 against 1 615 / 1 455 /
// 1 205-1 240: 200 VGPRs of unconditional requests cost the CU's memory pipe more than three round trips.)
// NT_ = AO_THREADS (256) is the launch for short caches; NT_ = AO_THREADS_LONG (1024) the one DecodeEngine replays from
// DecodeEngine.LONG4_FROM cached positions on: there the per-position arithmetic (which every slice of a head repeats) outweighs the
// per-wave overhead, and four times the lanes take a quarter of the positions each.  No prefetch ROWS then (sixteen waves fill the CU):
// the workgroups carry the prefetch share themselves.
constexpr int AO_THREADS_LONG = 1024;
template <int D, int NT_>
__global__ void __launch_bounds__(NT_) decode_attention_oproj_kernel(const mq_decode_attention_oproj_args a, unsigned long long* stamps) {
  // Geometry (NT threads).  PPP positions per pass, KB passes requested at the top (512 positions at head_dim 64).  p.v: thread (dq = dword of 4 dims, grp) owns one position of every G-position stripe;
  // BLK positions per block, PPB stripes per block; VB stripes (<= 32 registers: 512 positions at head_dim 64) requested at the top.
  constexpr int NT = NT_, NW = NT / 64;
  // Scores: ONE lane per cached position at head_dim <= 64 (the whole key row: CH = 4 16-byte chunks per lane, LPP lanes per position
  // beyond): with four lanes per position (rounds 3-5) a sweep over 257 positions was five passes whose epilogue ran on a quarter of the
  // lanes -- 1.6 us of instruction issue; one lane per position is two passes and no cross-lane sum.
  constexpr int LPP = D / 64 >= 1 ? D / 64 : 1, CH = D / 16 < 4 ? D / 16 : 4, PPP = NT / LPP;
  constexpr int KB0 = (512 / PPP < 8 / CH ? 512 / PPP : 8 / CH) >= 1 ? (512 / PPP < 8 / CH ? 512 / PPP : 8 / CH) : 1, KB = KB0;
  // p.v: the VALUE cache is transposed ([kv head][dim][position]): thread (d = tid % D, g = tid / D) owns dimension d of the 16-position
  // chunks g, g + NG, ...; a chunk is ONE 16-byte request and 12 v_dot4_i32_i8 against the probabilities' byte digits (below).  VC chunks
  // per thread are requested at the top (512 positions at head_dim 64).
  constexpr int NG = NT / D >= 1 ? NT / D : 1, VC0 = 32 / NG > 8 ? 8 : (32 / NG >= 1 ? 32 / NG : 1), VC = VC0;
  constexpr int MAXC = D / 16 < 8 ? D / 16 : 8;                    // 16-byte chunks of an o_proj row slice per thread
  static_assert(NG * D == NT || D > NT, "p.v mapping");
  extern __shared__ __attribute__((aligned(16))) char smem_raw[];
  float* s_sc = reinterpret_cast<float*>(smem_raw);              // [cache_len] scores -> exp; behind it 3 x [cache_len] bytes: the probabilities' digits
  __shared__ __attribute__((aligned(16))) int8_t s_q8[D], s_k8[D], s_v8[D], s_a8[D];
  __shared__ float s_redf[NW];
  __shared__ int s_redq[NW];
  __shared__ long long s_acc[NG * D];                            // [NG][D] partial p.v sums
  __shared__ int s_pnew;                                         // the new position's (index - zero point)
  __shared__ int s_redp[NW];                                     // per wave: sum over its positions of (probability index - zero point)
  const int H = a.heads, W = H * a.slices, rot = a.rot_dim;
  if ((int)blockIdx.x >= W) {                                      // L2 prefetch role (see decode_attention_kernel)
    const int q = (int)blockIdx.x - W;
    if (q >= a.prefetch_wgs) return;
    const unsigned long long t_go = __builtin_amdgcn_s_memrealtime() + (unsigned long long)a.prefetch_delay;
    while (__builtin_amdgcn_s_memrealtime() < t_go) __builtin_amdgcn_s_sleep(8);
    const int gwg = (q + W) % a.prefetch_wgs;                      // same linear id % 8 as this workgroup when prefetch_wgs % 8 == 0
    const size_t beg = (size_t)gwg * a.prefetch_stride;
    const size_t end = beg + a.prefetch_bytes_per_wg < a.prefetch_total ? beg + a.prefetch_bytes_per_wg : a.prefetch_total;
    const v4i* p = reinterpret_cast<const v4i*>(a.prefetch + beg);
    const size_t n = end > beg ? (end - beg) >> 4 : 0;
    v4i acc = {0, 0, 0, 0};
    for (size_t i = threadIdx.x; i < n; i += NT * 8) {
      v4i b[8];
#pragma unroll
      for (int u = 0; u < 8; ++u) b[u] = p[i + (size_t)u * NT < n ? i + (size_t)u * NT : i];
#pragma unroll
      for (int u = 0; u < 8; ++u) acc |= b[u];
    }
    asm volatile("" ::"v"(acc));
    return;
  }
  DG_STAMP(0);
  // workgroup -> (head, row range).  With kv_heads dividing 8 and W a multiple of 8: XCD x = block % 8 serves KV head x / (8 / kv_heads),
  // so the slices * (heads / kv_heads) workgroups that sweep the same keys / values share an L2 (observed placement: speed only).
  // (heads per KV head, KV heads and slices are powers of two in every model of the family: the host passes their logarithms, so that
  // no integer division sits in front of this launch's first request -- eight of them cost ~0.3 us; lg_slices < 0: the generic mapping)
  int h, c, kvh;
  {
    const int b = blockIdx.x;
    if (a.lg_slices >= 0) {
      const int lgs = a.lg_slices, lgg = a.lg_group, lgk = a.lg_kv;             // slices, heads per KV head, KV heads
      if (lgk <= 3) {
        const int lgx = 3 - lgk, x = b & 7, i = b >> 3;                          // 2^lgx XCDs per KV head
        const int lg_per_x = lgs + lgg - lgx;                                     // workgroups of a KV head per XCD (W / 8 >= 1 by the host's check)
        const int j = ((x & ((1 << lgx) - 1)) << lg_per_x) + i;                   // 0 .. per_kv - 1 inside KV head x >> lgx
        kvh = x >> lgx;
        h = (kvh << lgg) + (j >> lgs);
        c = j & ((1 << lgs) - 1);
      } else {
        h = b >> lgs;
        c = b & ((1 << lgs) - 1);
        kvh = h >> lgg;
      }
    } else {
      h = b / a.slices;
      c = b % a.slices;
      kvh = h / (H / a.kv_heads);
    }
  }
  const int tid = threadIdx.x, lane = tid & 63, wv = tid >> 6;
  const float cv = a.consts[lane];
  // the new token's q / k / v rows (fp32 outputs of the q|k|v launch), their RoPE partners and the cos / sin row of this position
  // (rope_row: staged once per token by mq_decode_embed, so that no request of this launch waits for *pos)
  const float* qp = a.qkv + (size_t)h * D;
  const float* kp = a.qkv + (size_t)H * D + (size_t)kvh * D;
  const float* vp = a.qkv + (size_t)(H + a.kv_heads) * D + (size_t)kvh * D;
  const int dd = tid < D ? tid : D - 1;
  const int half = rot >> 1;
  const int dpart = dd < rot ? (dd < half ? dd + half : dd - half) : dd;
  const int dr = dd < rot ? dd : 0;
  const float q_raw = qp[dd], q_par = qp[dpart], k_raw = kp[dd], k_par = kp[dpart], v_raw = vp[dd];
  const float cs = a.rope_row[dr], sn = a.rope_row[rot + dr];
  const int CL = a.cache_len;
  const int8_t* kc = a.k_cache + (size_t)kvh * CL * D;
  const int8_t* vc = a.v_cache + (size_t)kvh * D * CL;             // [16-position chunk][dim][16 positions]
  // ---- keys and values of the first 512 positions: the first 256 before *pos is known (addresses clamped by the cache length, masked by
  // T below), the rest behind it clamped by the position (beyond it every lane reads position 0: one line) -- requesting all 512
  // unconditionally made every workgroup pull 64 KB through its L1 at the head of the launch, 0.5 us at context 256
  const int sub = tid & (LPP - 1), slot = tid / LPP;
  v4i kbuf[KB][CH];
  auto load_keys = [&](int u, int lim) {
    const int t = u * PPP + slot;
    const int tc = t < lim ? t : 0;                                 // position 0 stands in (always valid memory)
#pragma unroll
    for (int ch = 0; ch < CH; ++ch) {
#ifdef MQ_AO_WHATIF_NOKEYS
      kbuf[u][ch] = v4i{tc, t, u, ch};
#else
      kbuf[u][ch] = *reinterpret_cast<const v4i*>(kc + (size_t)tc * D + (sub * CH + ch) * 16);
#endif
    }
  };
  const int vd = tid & (D - 1), vg = D >= NT ? 0 : tid / D;          // this thread's dimension and chunk stripe
  // value cache: [kv head][16-position chunk][dim][position in the chunk] -- a chunk of all dimensions is D x 16 contiguous bytes, so a
  // wave's request for (chunk j, 64 dimensions) is ONE coalesced KiB (as [dim][position] rows it touched 64 cache lines per request:
  // the p.v sweep took 6 us at 2 048 positions)
  const int8_t* vrow = vc + (size_t)vd * 16;
  v4i vbuf[VC];
  auto load_chunk = [&](int kk, int kbase, int lim_chunks) {         // chunk j = vg + NG (kbase + kk) of 16 positions
    const int j = vg + NG * (kbase + kk);
#ifdef MQ_AO_WHATIF_NOVALUES
    vbuf[kk] = v4i{j, kk, kbase, lim_chunks};
#else
    vbuf[kk] = *reinterpret_cast<const v4i*>(vrow + (size_t)(j < lim_chunks ? j : 0) * (D * 16));
#endif
  };
#pragma unroll
  for (int u = 0; u < KB; ++u)
    if ((u + 1) * PPP <= 256 || u == 0) load_keys(u, CL);
#pragma unroll
  for (int kk = 0; kk < VC0; ++kk)
    if (16 * NG * (kk + 1) <= 256 || kk == 0) load_chunk(kk, 0, CL >> 4);
  int pos;
  asm volatile("s_load_dword %0, %1, 0x0\n\ts_waitcnt lgkmcnt(0)" : "=s"(pos) : "s"(a.pos) : "memory");
  const bool live = pos >= 0 && pos < CL;                           // a step past the cache: nothing is written (the host raises first)
  const int T = live ? pos + 1 : 0;
#pragma unroll
  for (int u = 0; u < KB; ++u)
    if (!((u + 1) * PPP <= 256 || u == 0)) load_keys(u, T);
  const int nchunk = live ? (pos + 15) >> 4 : 0;                     // 16-position chunks of CACHED positions 0 .. pos - 1
#pragma unroll
  for (int kk = 0; kk < VC0; ++kk)
    if (!(16 * NG * (kk + 1) <= 256 || kk == 0)) load_chunk(kk, 0, nchunk);
  // ---- RoPE + the three input quantizers of the new token ---------------------------------------------------------------------------
  const Grid qa = const_grid(cv, AG_QK_A, a.qk_a), qb = const_grid(cv, AG_QK_B, a.qk_b), qo = const_grid(cv, AG_QK_OUT, a.qk_out);
  const Grid pa = const_grid(cv, AG_PV_A, a.pv_a), pb = const_grid(cv, AG_PV_B, a.pv_b), po = const_grid(cv, AG_PV_OUT, a.pv_out);
  const Grid oi = const_grid(cv, AG_O_IN, a.o_in);
  int qsum_part = 0;
  if (tid < D) {
    float qv = q_raw, kv = k_raw;
    if (tid < rot) {                                               // x * cos + rot(x) * sin, rot(x)[d] = d < rot/2 ? -x[d + rot/2] : x[d - rot/2]
      const float sg = tid < half ? -1.f : 1.f;                    // (-x) * sin == -(x * sin) exactly
      qv = __fadd_rn(__fmul_rn(q_raw, cs), __fmul_rn(sg * q_par, sn));
      kv = __fadd_rn(__fmul_rn(k_raw, cs), __fmul_rn(sg * k_par, sn));
    }
    const float iq = dq_index(qv, qa.s, qa.inv_s, qa.o, qa.qmin, qa.qmax), ik = dq_index(kv, qb.s, qb.inv_s, qb.o, qb.qmin, qb.qmax);
    const float iv = dq_index(v_raw, pb.s, pb.inv_s, pb.o, pb.qmin, pb.qmax);
    const int sq = (iq != iq ? 0 : (int)iq) - 128, sk = (ik != ik ? 0 : (int)ik) - 128, sv = (iv != iv ? 0 : (int)iv) - 128;
    s_q8[tid] = (int8_t)sq;
    s_k8[tid] = (int8_t)sk;
    s_v8[tid] = (int8_t)sv;
    qsum_part = sq;
    if (live && c == 0 && h == kvh * (H / a.kv_heads)) {           // the group's first head (its first slice) appends to the cache
      a.k_cache[((size_t)kvh * CL + pos) * D + tid] = (int8_t)sk;
      a.v_cache[(size_t)kvh * D * CL + ((size_t)(pos >> 4) * D + tid) * 16 + (pos & 15)] = (int8_t)sv;  // (chunk-blocked transposed value cache)
    }
  }
  if (wv < (D + 63) / 64) {                                        // (the waves that hold the D query bytes)
    const int w = wave_sum_dpp(qsum_part);
    if (lane == 0) s_redq[wv] = w;
  }
  __syncthreads();
  DG_STAMP(1);
  int qsum = 0;
#pragma unroll
  for (int w = 0; w < (D + 63) / 64; ++w) qsum += s_redq[w];
  const int zq = (int)qa.o - 128, zk = (int)qb.o - 128, zv = (int)pb.o - 128, zp = (int)pa.o;
  const float alpha_qk = __fmul_rn(qa.s, qb.s), alpha_pv = __fmul_rn(pa.s, pb.s);
  const int qconst = D * zq * zk - zk * qsum;                      // sum (iq - zq)(ik - zk) = sum sq sk - zk sum sq - zq sum sk + D zq zk
  constexpr bool pow2 = (D == 64 || D == 256);
  const float sqrt_d = __fsqrt_rn((float)D), inv_sqrt_d = 1.0f / (D == 64 ? 8.0f : 16.0f);
  v4i qf[CH], kn[CH];                                              // this lane's share of the query / of the NEW key (never via memory)
#pragma unroll
  for (int ch = 0; ch < CH; ++ch) {
    qf[ch] = *reinterpret_cast<const v4i*>(s_q8 + (sub * CH + ch) * 16);
    kn[ch] = *reinterpret_cast<const v4i*>(s_k8 + (sub * CH + ch) * 16);
  }
  const v4i ones = {0x01010101, 0x01010101, 0x01010101, 0x01010101};
  // ---- scores ----------------------------------------------------------------------------------------------------------------------
  // The CACHED positions t < pos in passes of PPP; the first KB passes are straight-line code on the registers requested at the top (so
  // that hipcc waits for exactly the pass it needs: a loop header with a refill inside costs a vmcnt(0), i.e. the arrival of every
  // request in flight, incl. the passes behind *pos); the new position's score comes from the key in LDS.
  float lmax = -INFINITY;
  auto score_pass = [&](int t0, int u) {
    const int t = t0 + u * PPP + slot;
    int dot = 0, ks = 0;
#pragma unroll
    for (int ch = 0; ch < CH; ++ch) {
      dot = dot16(kbuf[u][ch], qf[ch], dot);
      ks = dot16(kbuf[u][ch], ones, ks);
    }
    dot = quad_sum<LPP>(dot);
    ks = quad_sum<LPP>(ks);
    if (t < pos && sub == 0) {
      const int ti = dot - zq * ks + qconst;
      const float val = __fmul_rn((float)ti, alpha_qk);
      const float qv = qo.fq(val);
      const float sc = pow2 ? __fmul_rn(qv, inv_sqrt_d) : __fdiv_rn(qv, sqrt_d);     // qk_bmm(...) / sqrt(head_dim)  (hf_model.py:513)
      s_sc[t] = sc;
      lmax = fmaxf(lmax, sc);
    }
  };
  if (live) {
#pragma unroll
    for (int u = 0; u < KB; ++u)
      if (u * PPP < pos) score_pass(0, u);                          // (uniform)
    for (int t0 = KB * PPP; t0 < pos; t0 += KB * PPP) {             // later batches (one exposed round trip each: long caches only)
#pragma unroll
      for (int u = 0; u < KB; ++u) {
        const int t = t0 + u * PPP + slot, tc = t < pos ? t : 0;
#pragma unroll
        for (int ch = 0; ch < CH; ++ch) kbuf[u][ch] = *reinterpret_cast<const v4i*>(kc + (size_t)tc * D + (sub * CH + ch) * 16);
      }
#pragma unroll
      for (int u = 0; u < KB; ++u)
        if (t0 + u * PPP < pos) score_pass(t0, u);
    }
    {                                                               // the new position (every LPP-lane group computes it; one lane stores it)
      int dot = 0, ks = 0;
#pragma unroll
      for (int ch = 0; ch < CH; ++ch) {
        dot = dot16(kn[ch], qf[ch], dot);
        ks = dot16(kn[ch], ones, ks);
      }
      dot = quad_sum<LPP>(dot);
      ks = quad_sum<LPP>(ks);
      const int ti = dot - zq * ks + qconst;
      const float val = __fmul_rn((float)ti, alpha_qk);
      const float qv = qo.fq(val);
      const float sc = pow2 ? __fmul_rn(qv, inv_sqrt_d) : __fdiv_rn(qv, sqrt_d);
      if (tid == 0) s_sc[pos] = sc;
      lmax = fmaxf(lmax, sc);
    }
  }
  lmax = wave_max_f(lmax);
  if (lane == 0) s_redf[wv] = lmax;
  // o_proj's weights are requested HERE, behind the scores (see the header comment); needed ~2 us from now
  const int tpr = a.tpr, lgt = tpr == 4 ? 2 : (tpr == 2 ? 1 : 0);
  const int R = a.lg_slices >= 0 ? a.N >> a.lg_slices : a.N / a.slices, cpt = (D / 16) >> lgt;
  const int orow = tid >> lgt, osub = tid & (tpr - 1);
  const bool o_ok = orow < R;
  const int n_out = c * R + (o_ok ? orow : 0);
  v4i wbuf[MAXC];
  {
    const v4i* wp = reinterpret_cast<const v4i*>(a.o_w + ((size_t)h * a.N + n_out) * D) + osub * cpt;
#pragma unroll
    for (int j = 0; j < MAXC; ++j) wbuf[j] = __builtin_nontemporal_load(wp + (j < cpt ? j : 0));
  }
  const int o_zp = a.o_wzp[n_out];
  // 1024 threads: no room for co-resident prefetch workgroups, so every workgroup pulls its own share of the later weight stream into the
  // L2 of the XCD it runs on (piece b = what workgroup b of that launch will read): three 16-byte requests per thread, never waited for
  // before the kernel's last instruction
  constexpr int PFN = NT > 256 ? 3 : 0;
  v4i pf[PFN > 0 ? PFN : 1];
  if constexpr (PFN > 0) {
    const int q = blockIdx.x;
    const size_t beg = (size_t)q * a.prefetch_stride;
    const size_t end = beg + a.prefetch_bytes_per_wg < a.prefetch_total ? beg + a.prefetch_bytes_per_wg : a.prefetch_total;
    const bool on = q < a.prefetch_wgs && end > beg;
    const v4i* p = on ? reinterpret_cast<const v4i*>(a.prefetch + beg) : reinterpret_cast<const v4i*>(a.o_w);
    const size_t n = on ? (end - beg) >> 4 : 1;
#pragma unroll
    for (int u = 0; u < PFN; ++u) {
      const size_t i = (size_t)tid + (size_t)u * NT;
      pf[u] = p[i < n ? i : 0];
    }
  }
  __syncthreads();
  DG_STAMP(2);
  float mx = s_redf[0];
#pragma unroll
  for (int w = 1; w < NW; ++w) mx = fmaxf(mx, s_redf[w]);
  __syncthreads();
  // The sum of the exponentials is a FLOAT sum: its order is part of the result -- thread j adds e[j], e[j + 256], ... in that order, DPP
  // tree per wave, (w0 + w1) + (w2 + w3), as decode_attention_kernel's 256 threads do (with more threads: the first four waves, from LDS)
  if constexpr (NT == 256) {
    float lsum = 0.f;
    for (int t = tid; t < T; t += 256) {
      const float e = expf(s_sc[t] - mx);
      s_sc[t] = e;
      lsum += e;
    }
    lsum = wave_sum_f(lsum);
    if (lane == 0) s_redf[wv] = lsum;
  } else {
    for (int t = tid; t < T; t += NT) s_sc[t] = expf(s_sc[t] - mx);
    __syncthreads();
    if (wv < 4) {
      float lsum = 0.f;
      for (int t = tid; t < T; t += 256) lsum += s_sc[t];
      lsum = wave_sum_f(lsum);
      if (lane == 0) s_redf[wv] = lsum;
    }
  }
  __syncthreads();
  const float tot_e = (s_redf[0] + s_redf[1]) + (s_redf[2] + s_redf[3]);
  // pv_bmm's input quantizer, once per position.  The index ip (0 .. 65535) leaves as THREE signed bytes per position so that the sweep
  // over the values is v_dot4_i32_i8 work:  ip = 256 (ph - 128) + (pl - 128) + 32896 m  with m = 1 for a cached position, and
  // ph = pl = m = 0 (no contribution) beyond the sequence and AT the new position, whose value is not in the cache yet (another
  // workgroup appends it in this very launch): it is added from registers below.
  int8_t* s_ph = reinterpret_cast<int8_t*>(s_sc + CL);
  int8_t* s_pl = s_ph + CL;
  int8_t* s_pm = s_pl + CL;
  int my_p = 0;                                                     // <= 128 positions per thread x 65535: int32 holds a wave's sum too
  for (int t = tid; t < (nchunk << 4) || t < T; t += NT) {
    int bh = 0, bl = 0, bm = 0;
    if (t < T) {
      const float p = __fdiv_rn(s_sc[t], tot_e);
      const float ipf = dq_index(p, pa.s, pa.inv_s, pa.o, pa.qmin, pa.qmax);
      const int ip = ipf != ipf ? 0 : (int)ipf;
      my_p += ip - zp;
      if (t == pos) s_pnew = ip - zp;
      else bh = (ip >> 8) - 128, bl = (ip & 255) - 128, bm = 1;
    }
    if (t < (nchunk << 4)) {
      s_ph[t] = (int8_t)bh;
      s_pl[t] = (int8_t)bl;
      s_pm[t] = (int8_t)bm;
    }
  }
  {
    const int w = wave_sum_dpp(my_p);                               // (an LDS atomic per thread was tried: 256 same-address atomics cost ~1 us)
    if (lane == 0) s_redp[wv] = w;
  }
  __syncthreads();
  DG_STAMP(3);
  // ---- p.v over the cached positions t < pos: exact integers -----------------------------------------------------------------------
  // sum_t (ip - zp)(vs - zv) = [256 Sh + Sl + (32896 - zp) Sm] - zv P,  Sh / Sl / Sm = sum_t digit[t] vs[t][d],  P = sum_t (ip - zp)
  int sh = 0, sl = 0, sm = 0;
  const int my_chunks = vg < nchunk ? (nchunk - vg + NG - 1) / NG : 0;
  auto pv_chunk = [&](int k0, int kk) {
    const int j = vg + NG * (k0 + kk);
    const v4i ph4 = *reinterpret_cast<const v4i*>(s_ph + 16 * j), pl4 = *reinterpret_cast<const v4i*>(s_pl + 16 * j);
    const v4i pm4 = *reinterpret_cast<const v4i*>(s_pm + 16 * j);
    sh = dot16(vbuf[kk], ph4, sh);
    sl = dot16(vbuf[kk], pl4, sl);
    sm = dot16(vbuf[kk], pm4, sm);
  };
#pragma unroll
  for (int kk = 0; kk < VC; ++kk)                                   // the chunks requested at the top: straight-line code (see the scores)
    if (kk < my_chunks) pv_chunk(0, kk);
  for (int k0 = VC; k0 < my_chunks; k0 += VC) {                     // later batches: one exposed round trip each (long caches only)
#pragma unroll
    for (int kk = 0; kk < VC; ++kk) load_chunk(kk, k0, nchunk);
#pragma unroll
    for (int kk = 0; kk < VC; ++kk)
      if (k0 + kk < my_chunks) pv_chunk(k0, kk);
  }
  {
    long long part = 256ll * sh + sl + (long long)(32896 - zp) * sm;
    if (vg == 0 && live) part += (long long)s_v8[vd] * (long long)s_pnew;      // the new position: stripe 0 adds it from LDS
    if (D <= NT || tid < D) s_acc[vg * D + vd] = part;
  }
  __syncthreads();
  DG_STAMP(4);
  int a_byte = 0;
  if (tid < D) {
    long long tot = 0;
#pragma unroll
    for (int g2 = 0; g2 < NG; ++g2) tot += s_acc[g2 * D + tid];
    long long psum = 0;
#pragma unroll
    for (int w = 0; w < NW; ++w) psum += s_redp[w];
    tot -= (long long)zv * psum;
    const float pre = (float)((double)tot * (double)alpha_pv);     // one rounding of the exact sum (as mq_attention.hip)
    const float y = po.fq(pre);
    const float qi = dq_index(y, oi.s, oi.inv_s, oi.o, oi.qmin, oi.qmax);
    a_byte = (qi != qi ? (int)oi.qmin : (int)qi) - 128;
    s_a8[tid] = (int8_t)a_byte;
    if (a.out_q && c == 0 && live) a.out_q[(size_t)h * D + tid] = (int8_t)a_byte;
  }
  {
    const int w = wave_sum_dpp(a_byte);                            // (threads that hold no output byte contribute 0)
    if (lane == 0) s_redq[wv] = w;
  }
  __syncthreads();
  // ---- o_proj: rows [c R, (c + 1) R) x K-slice [h D, (h + 1) D) ---------------------------------------------------------------------
  int rs_h = 0;
#pragma unroll
  for (int w = 0; w < NW; ++w) rs_h += s_redq[w];
  int part = 0;
#pragma unroll
  for (int j = 0; j < MAXC; ++j)
    if (j < cpt) part = dot16(wbuf[j], *reinterpret_cast<const v4i*>(s_a8 + (osub * cpt + j) * 16), part);
  if (tpr >= 2) part += __builtin_amdgcn_update_dpp(0, part, 0xB1, 0xf, 0xf, true);
  if (tpr == 4) part += __builtin_amdgcn_update_dpp(0, part, 0x4E, 0xf, 0xf, true);
  if (o_ok && osub == 0 && live) {
    const int v = (int)((unsigned)part - (unsigned)o_zp * (unsigned)rs_h);
    __hip_atomic_fetch_add(a.o_acc + n_out, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
  }
  if constexpr (PFN > 0) {
#pragma unroll
    for (int u = 0; u < PFN; ++u) asm volatile("" ::"v"(pf[u]));
  }
#ifdef MQ_DECODE_STAMPS
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  DG_STAMP(5);
#endif
}

// ---- token start: embedding row + the RoPE row of this position ---------------------------------------------------------------------
// x <- table[*tok] (embed_tokens; a scaled table carries Gemma's normalize_embed) and rope_row <- {cos[*pos][0 .. rot), sin[*pos][0 .. rot)}:
// the one place of a token that chases *pos, so that the 22 attention launches read a FIXED address instead of a pos -> cos / sin chain
// (two dependent round trips at the head of each: 0.8 us of 6).  A step past the table reads row 0 (the launches behind it store nothing).
__global__ void __launch_bounds__(256) decode_embed_kernel(const float* __restrict__ table, const long long* __restrict__ tok, int hidden, long long vocab,
                                                           const float* __restrict__ cosr, const float* __restrict__ sinr, const int* __restrict__ pos, int rot,
                                                           int max_pos, float* __restrict__ x, float* __restrict__ rope_row) {
  const long long tk = tok[0];
  const long long row = (tk >= 0 && tk < vocab) ? tk : 0;
  for (int i = blockIdx.x * 256 + threadIdx.x; i < hidden; i += gridDim.x * 256) x[i] = table[(size_t)row * hidden + i];
  if (blockIdx.x == 0 && rope_row) {
    const int p = pos[0], pc = (p >= 0 && p < max_pos) ? p : 0;
    for (int i = threadIdx.x; i < rot; i += 256) {
      rope_row[i] = cosr[(size_t)pc * rot + i];
      rope_row[rot + i] = sinr[(size_t)pc * rot + i];
    }
  }
}

// ---- final norm (floating point HFRMSNorm) + lm_head (fp32 weights) ----------------------------------------------------------------
__global__ void __launch_bounds__(256) decode_head_kernel(const float* __restrict__ x, const float* __restrict__ norm_w,
                                                          const float* __restrict__ norm_b, const int layernorm, float eps,
                                                          const float* __restrict__ w, const float* __restrict__ bias, int K, int V,
                                                          float* __restrict__ logits, unsigned long long* stamps) {
  DG_STAMP(0);
  extern __shared__ __attribute__((aligned(16))) char smem_raw[];
  float* s_x = reinterpret_cast<float*>(smem_raw);               // [K] normalised activation
  __shared__ float s_red[4];
  const int tid = threadIdx.x, lane = tid & 63, wv = tid >> 6;
  if (layernorm) {                                                 // final nn.LayerNorm (StableLM-2: hf_model.py:1440-1441), fp32, torch's expression
    float s1 = 0.f;
    for (int i = tid; i < K; i += 256) s1 += x[i];
    s1 = wave_sum_f(s1);
    if (lane == 0) s_red[wv] = s1;
    __syncthreads();
    const float mu = __fdiv_rn((s_red[0] + s_red[1]) + (s_red[2] + s_red[3]), (float)K);
    __syncthreads();
    float s2 = 0.f;
    for (int i = tid; i < K; i += 256) {
      const float d = x[i] - mu;
      s2 += d * d;
    }
    s2 = wave_sum_f(s2);
    if (lane == 0) s_red[wv] = s2;
    __syncthreads();
    const float var = __fdiv_rn((s_red[0] + s_red[1]) + (s_red[2] + s_red[3]), (float)K);
    const float r = __fdiv_rn(1.0f, __fsqrt_rn(__fadd_rn(var, eps))), sh = __fmul_rn(-r, mu);
    for (int i = tid; i < K; i += 256) {
      float y = __fadd_rn(__fmul_rn(x[i], r), sh);
      if (norm_w) y = __fmul_rn(y, norm_w[i]);
      if (norm_b) y = __fadd_rn(y, norm_b[i]);
      s_x[i] = y;
    }
  } else {
    float ss = 0.f;
    for (int i = tid; i < K; i += 256) ss += x[i] * x[i];
    ss = wave_sum_f(ss);
    if (lane == 0) s_red[wv] = ss;
    __syncthreads();
    const float mean = __fdiv_rn((s_red[0] + s_red[1]) + (s_red[2] + s_red[3]), (float)K);
    const float r = __fdiv_rn(1.0f, __fsqrt_rn(__fadd_rn(mean, eps)));      // hf_model.py:183 (x * rsqrt(mean + eps)), then weight *
    for (int i = tid; i < K; i += 256) s_x[i] = norm_w ? __fmul_rn(norm_w[i], __fmul_rn(x[i], r)) : x[i];
  }
  __syncthreads();
  // a wave per vocabulary row, float4 loads (16 B per lane)
  const int nvec = K >> 2;
  typedef float v4f __attribute__((ext_vector_type(4)));
  const int stride = gridDim.x * 4;
  if (K % 256 == 0 && K <= 2048) {
    // Round 6: the row's <= 8 requests per lane leave together, and the NEXT row's leave before this row's sum starts (two rows in
    // flight per wave): the stream used to issue one request per loop trip behind the previous trip's four dependent adds -- 5.8 TB/s.
    // The sum itself is unchanged (i ascending, four adds per request): the same bits.
    constexpr int NV = 8;
    const int nj = nvec >> 6;                                      // requests per lane and row (uniform)
    v4f cur[NV], nxt[NV];
    int row = blockIdx.x * 4 + wv;
    auto fetch = [&](v4f (&buf)[NV], int r) {
      const v4f* wr = reinterpret_cast<const v4f*>(w + (size_t)(r < V ? r : V - 1) * K);
#pragma unroll
      for (int j = 0; j < NV; ++j) buf[j] = __builtin_nontemporal_load(wr + lane + 64 * (j < nj ? j : 0));
    };
    auto reduce = [&](const v4f (&buf)[NV], int r) {
      float acc = 0.f;
#pragma unroll
      for (int j = 0; j < NV; ++j) {
        if (j < nj) {
          const float4 b = reinterpret_cast<const float4*>(s_x)[lane + 64 * j];
          acc += buf[j][0] * b.x;
          acc += buf[j][1] * b.y;
          acc += buf[j][2] * b.z;
          acc += buf[j][3] * b.w;
        }
      }
      acc = wave_sum_f(acc);
      if (lane == 0 && r < V) logits[r] = bias ? acc + bias[r] : acc;
    };
    fetch(cur, row);
    for (; row < V; row += 2 * stride) {
      fetch(nxt, row + stride);
      reduce(cur, row);
      fetch(cur, row + 2 * stride);
      reduce(nxt, row + stride);
    }
  } else {
    for (int row = blockIdx.x * 4 + wv; row < V; row += stride) {
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

static int decode_gemv_geometry(const mq_decode_gemv_args& g, int* rows_per_wg, unsigned* grid) {
  static std::atomic<int> cus_of[kMaxDevices];
  const int dev = current_device();
  int cus = cus_of[dev].load(std::memory_order_relaxed);
  if (!cus) {
    hipDeviceProp_t prop;
    cus = (hipGetDeviceProperties(&prop, dev) == hipSuccess && prop.multiProcessorCount > 0) ? prop.multiProcessorCount : 256;
    cus_of[dev].store(cus, std::memory_order_relaxed);
  }
  const int NL = g.gate_q ? g.N / 2 : g.N;
  int rpw = (NL + cus - 1) / cus;
  if (rpw > DG_STR * 64) rpw = DG_STR * 64;
  *rows_per_wg = rpw;
  *grid = (unsigned)((NL + rpw - 1) / rpw);
  return NL;
}

int mq_decode_gemv_geometry(const mq_decode_gemv_args* args, int64_t* workgroups, int64_t* bytes_per_workgroup, int64_t* total_bytes) {
  MQ_REQUIRE(args && workgroups && bytes_per_workgroup && total_bytes && args->K > 0 && args->N > 0, "mq_decode_gemv_geometry: null / empty argument");
  int rpw;
  unsigned grid;
  decode_gemv_geometry(*args, &rpw, &grid);
  const int64_t row = (int64_t)(args->gate_q ? 2 : 1) * (args->w4 ? args->K / 2 : args->K);
  *workgroups = grid;
  *bytes_per_workgroup = rpw * row;
  *total_bytes = (int64_t)args->N * (args->w4 ? args->K / 2 : args->K);
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
  const bool gate = g.gate_q != nullptr, opre = g.o_acc != nullptr;
  MQ_REQUIRE(gate || g.y, "mq_decode_gemv: no output");
  MQ_REQUIRE(!gate || (g.norm_w && !g.xq), "mq_decode_gemv: gate mode is served for the norm-fused prologue (fp32 x + norm_w)");
  MQ_REQUIRE(!gate || (g.N % 2 == 0 && g.gate_out.scale && g.out_grid[0].scale && g.out_grid[1].scale && (g.gate_act == 0 || g.gate_act == 1)),
             "mq_decode_gemv: gate mode needs an even N (interleaved w1 / w3 rows), both output grids and the w2 input grid");
  MQ_REQUIRE(!g.zero_acc || (g.zero_n > 0 && aligned(g.zero_acc, 4)), "mq_decode_gemv: zero_acc needs zero_n > 0");
  MQ_REQUIRE(!opre || (gate && g.o_alpha && g.o_ct && g.x_mid && g.K <= 2 * 4 * DG_PRO * 64 && aligned(g.o_acc, 16) && aligned(g.o_alpha, 16) &&
                       aligned(g.o_ct, 16) && aligned(g.x_mid, 16) && (!g.o_bias || aligned(g.o_bias, 16))),
             "mq_decode_gemv: o_acc (o_proj's epilogue as prologue) is served for the gate launch, K <= 4096, with o_alpha / o_ct / x_mid, 16-byte aligned");
  int rows_per_wg;
  unsigned grid;
  decode_gemv_geometry(g, &rows_per_wg, &grid);
  const size_t lds = (size_t)g.K + 64;
  hipStream_t st = as_stream(stream);
  unsigned long long* stamps = STAMP_SLOT(gate ? 1 : (g.norm_w ? 0 : (g.xq ? 3 : 2)), grid);
  const int xmode = g.xq ? XM_I8 : (g.norm_w ? (g.layernorm ? XM_LNORM : XM_NORM) : XM_F32);
  MQ_REQUIRE(!g.norm_bias || (g.layernorm && aligned(g.norm_bias, 16)), "mq_decode_gemv: norm_bias belongs to the LayerNorm prologue (layernorm = 1), 16-byte aligned");
#define MQ_DG_LAUNCH(XM, PR, W4, OP)                                                                                   \
  decode_gemv_kernel<XM, PR, W4, OP><<<grid, DG_THREADS, lds, st>>>(g, rows_per_wg, stamps)
#define MQ_DG_LAUNCH_W4(XM, PR, OP)                    \
  do {                                                 \
    if (g.w4) MQ_DG_LAUNCH(XM, PR, true, OP);          \
    else MQ_DG_LAUNCH(XM, PR, false, OP);              \
  } while (0)
  if (gate && opre) {
    if (xmode == XM_LNORM) MQ_DG_LAUNCH_W4(XM_LNORM, PAIR_GATE, true);
    else MQ_DG_LAUNCH_W4(XM_NORM, PAIR_GATE, true);
  } else if (gate) {
    if (xmode == XM_LNORM) MQ_DG_LAUNCH_W4(XM_LNORM, PAIR_GATE, false);
    else MQ_DG_LAUNCH_W4(XM_NORM, PAIR_GATE, false);
  } else if (xmode == XM_LNORM) {
    MQ_DG_LAUNCH_W4(XM_LNORM, PAIR_NONE, false);
  } else if (xmode == XM_NORM) {
    MQ_DG_LAUNCH_W4(XM_NORM, PAIR_NONE, false);
  } else if (xmode == XM_F32) {
    MQ_DG_LAUNCH_W4(XM_F32, PAIR_NONE, false);
  } else {
    MQ_DG_LAUNCH_W4(XM_I8, PAIR_NONE, false);
  }
#undef MQ_DG_LAUNCH_W4
#undef MQ_DG_LAUNCH
  MQ_LAUNCH_CHECK("mq_decode_gemv");
  return MQ_OK;
}

int mq_decode_attention(const mq_decode_attention_args* args, mq_stream_t stream) {
  MQ_REQUIRE(args != nullptr, "mq_decode_attention: null argument block");
  const mq_decode_attention_args& a = *args;
  MQ_REQUIRE(a.qkv && a.k_cache && a.v_cache && a.cos && a.sin && a.pos && a.consts && (a.out || a.out_q), "mq_decode_attention: null pointer");
  MQ_REQUIRE(a.heads > 0 && a.kv_heads > 0 && a.heads % a.kv_heads == 0 && (a.head_dim == 32 || a.head_dim == 64 || a.head_dim == 128 || a.head_dim == 256) &&
                 a.cache_len > 0 && a.cache_len <= 32768 && a.rot_dim > 0 && a.rot_dim <= a.head_dim && a.rot_dim % 2 == 0,
             "mq_decode_attention: heads=%d kv_heads=%d head_dim=%d (32 / 64 / 128 / 256) cache_len=%d (<= 32768) rot_dim=%d", a.heads, a.kv_heads, a.head_dim,
             a.cache_len, a.rot_dim);
  MQ_REQUIRE(a.nsplit >= 1 && a.nsplit <= 16 && (a.nsplit == 1 || (a.part && a.ticket)), "mq_decode_attention: nsplit=%d (1..16; > 1 needs part and ticket)", a.nsplit);
  MQ_REQUIRE(a.qk_a.scale && a.qk_b.scale && a.pv_a.scale && a.pv_b.scale && a.qk_a.qmin == 0.f && a.qk_a.qmax == 255.f && a.qk_b.qmin == 0.f &&
                 a.qk_b.qmax == 255.f && a.pv_b.qmin == 0.f && a.pv_b.qmax == 255.f && a.pv_a.qmin == 0.f && a.pv_a.qmax <= 65535.f,
             "mq_decode_attention: q / k / v need 8-bit unsigned grids, the probabilities an unsigned grid of at most 16 bits");
  MQ_REQUIRE(!a.out_q || (a.o_in.scale && a.o_in.qmin == 0.f && a.o_in.qmax == 255.f), "mq_decode_attention: the int8 output image needs the consumer's 8-bit unsigned grid (o_in)");
  MQ_REQUIRE(aligned(a.k_cache, 16) && aligned(a.v_cache, 16) && aligned(a.consts, 16) && aligned(a.qkv, 4), "mq_decode_attention: caches / consts must be 16-byte aligned");
  const size_t lds = (size_t)a.cache_len * sizeof(float);
  const void* fn = a.head_dim == 32 ? reinterpret_cast<const void*>(decode_attention_kernel<32>)
                   : a.head_dim == 64 ? reinterpret_cast<const void*>(decode_attention_kernel<64>)
                   : a.head_dim == 128 ? reinterpret_cast<const void*>(decode_attention_kernel<128>)
                                       : reinterpret_cast<const void*>(decode_attention_kernel<256>);
  static std::atomic<size_t> lds_set[kMaxDevices][4];
  const int dev = current_device(), ki = a.head_dim == 32 ? 0 : a.head_dim == 64 ? 1 : a.head_dim == 128 ? 2 : 3;
  if (lds > 32768 && lds_set[dev][ki].load(std::memory_order_relaxed) < lds) {
    MQ_REQUIRE(hipFuncSetAttribute(fn, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds) == hipSuccess,
               "mq_decode_attention: %zu bytes of dynamic LDS rejected", lds);
    lds_set[dev][ki].store(lds, std::memory_order_relaxed);
  }
  MQ_REQUIRE(a.prefetch_wgs == 0 || (a.prefetch && aligned(a.prefetch, 16) && a.prefetch_bytes_per_wg % 16 == 0 && a.prefetch_wgs > 0 && a.prefetch_wgs <= 4096),
             "mq_decode_attention: prefetch needs a 16-byte aligned range and 1..4096 workgroups");
  MQ_REQUIRE(a.prefetch_wgs == 0 || (a.prefetch_stride >= a.prefetch_bytes_per_wg && a.prefetch_stride % 16 == 0 && a.prefetch_delay >= 0 && a.prefetch_delay <= 100000),
             "mq_decode_attention: prefetch_stride >= prefetch_bytes_per_wg (multiples of 16), prefetch_delay in 0..100000 (10 ns units)");
  const dim3 grid((unsigned)a.heads, (unsigned)(a.nsplit + (a.prefetch_wgs + a.heads - 1) / a.heads));
  unsigned long long* stamps = STAMP_SLOT(4, (unsigned)(a.heads * a.nsplit));
  hipStream_t st = as_stream(stream);
  switch (a.head_dim) {
    case 32: decode_attention_kernel<32><<<grid, 256, lds, st>>>(a, stamps); break;
    case 64: decode_attention_kernel<64><<<grid, 256, lds, st>>>(a, stamps); break;
    case 128: decode_attention_kernel<128><<<grid, 256, lds, st>>>(a, stamps); break;
    default: decode_attention_kernel<256><<<grid, 256, lds, st>>>(a, stamps); break;
  }
  MQ_LAUNCH_CHECK("mq_decode_attention");
  return MQ_OK;
}

int mq_decode_attention_oproj(const mq_decode_attention_oproj_args* args, mq_stream_t stream) {
  MQ_REQUIRE(args != nullptr, "mq_decode_attention_oproj: null argument block");
  const mq_decode_attention_oproj_args& a = *args;
  MQ_REQUIRE(a.qkv && a.k_cache && a.v_cache && a.rope_row && a.pos && a.consts && a.o_w && a.o_wzp && a.o_acc, "mq_decode_attention_oproj: null pointer");
  MQ_REQUIRE(a.heads > 0 && a.kv_heads > 0 && a.heads % a.kv_heads == 0 && (a.head_dim == 32 || a.head_dim == 64 || a.head_dim == 128 || a.head_dim == 256) &&
                 a.cache_len > 0 && a.cache_len <= 32768 && a.rot_dim > 0 && a.rot_dim <= a.head_dim && a.rot_dim % 2 == 0,
             "mq_decode_attention_oproj: heads=%d kv_heads=%d head_dim=%d (32 / 64 / 128 / 256) cache_len=%d (<= 32768) rot_dim=%d", a.heads, a.kv_heads, a.head_dim,
             a.cache_len, a.rot_dim);
  MQ_REQUIRE(a.qk_a.scale && a.qk_b.scale && a.pv_a.scale && a.pv_b.scale && a.qk_a.qmin == 0.f && a.qk_a.qmax == 255.f && a.qk_b.qmin == 0.f &&
                 a.qk_b.qmax == 255.f && a.pv_b.qmin == 0.f && a.pv_b.qmax == 255.f && a.pv_a.qmin == 0.f && a.pv_a.qmax <= 65535.f,
             "mq_decode_attention_oproj: q / k / v need 8-bit unsigned grids, the probabilities an unsigned grid of at most 16 bits");
  MQ_REQUIRE(a.o_in.scale && a.o_in.qmin == 0.f && a.o_in.qmax == 255.f, "mq_decode_attention_oproj: o_proj needs an 8-bit unsigned input grid (o_in)");
  const int chunks = a.head_dim / 16;
  MQ_REQUIRE(a.N > 0 && a.slices > 0 && a.N % a.slices == 0 && (a.tpr == 1 || a.tpr == 2 || a.tpr == 4) && chunks % a.tpr == 0 && chunks / a.tpr <= 8 &&
                 (a.N / a.slices) * a.tpr <= AO_THREADS && (long long)a.heads * a.slices <= 65535 && (a.threads == 0 || a.threads == AO_THREADS || a.threads == AO_THREADS_LONG),
             "mq_decode_attention_oproj: N=%d slices=%d tpr=%d: N %% slices == 0, tpr in {1, 2, 4} dividing head_dim / 16 with <= 8 chunks per thread, "
             "N / slices * tpr <= %d", a.N, a.slices, a.tpr, AO_THREADS);
  MQ_REQUIRE(a.lg_slices < 0 || ((1 << a.lg_slices) == a.slices && (1 << a.lg_kv) == a.kv_heads && (a.kv_heads << a.lg_group) == a.heads &&
                                 (a.lg_kv > 3 || a.lg_slices + a.lg_group >= 3 - a.lg_kv)),
             "mq_decode_attention_oproj: lg_slices / lg_group / lg_kv must be the logarithms of slices, heads / kv_heads, kv_heads (or lg_slices < 0)");
  MQ_REQUIRE(aligned(a.k_cache, 16) && aligned(a.v_cache, 16) && aligned(a.consts, 16) && aligned(a.qkv, 4) && aligned(a.o_w, 16) && aligned(a.rope_row, 4),
             "mq_decode_attention_oproj: caches / consts / o_w must be 16-byte aligned");
  MQ_REQUIRE(a.prefetch_wgs == 0 || (a.prefetch && aligned(a.prefetch, 16) && a.prefetch_bytes_per_wg % 16 == 0 && a.prefetch_wgs > 0 && a.prefetch_wgs <= 4096 &&
                                    a.prefetch_stride >= a.prefetch_bytes_per_wg && a.prefetch_stride % 16 == 0 && a.prefetch_delay >= 0 && a.prefetch_delay <= 100000),
             "mq_decode_attention_oproj: prefetch needs a 16-byte aligned range, 1..4096 workgroups, stride >= bytes per workgroup, delay in 0..100000 (10 ns units)");
  MQ_REQUIRE(a.cache_len % 16 == 0, "mq_decode_attention_oproj: cache_len=%d must be a multiple of 16 (transposed value cache, 16-byte chunks)", a.cache_len);
  const size_t lds = (size_t)a.cache_len * (sizeof(float) + 3);
  const bool lng = a.threads == AO_THREADS_LONG;
#define MQ_AO_FN(DD) (lng ? reinterpret_cast<const void*>(decode_attention_oproj_kernel<DD, AO_THREADS_LONG>) : reinterpret_cast<const void*>(decode_attention_oproj_kernel<DD, AO_THREADS>))
  const void* fn = a.head_dim == 32 ? MQ_AO_FN(32) : a.head_dim == 64 ? MQ_AO_FN(64) : a.head_dim == 128 ? MQ_AO_FN(128) : MQ_AO_FN(256);
#undef MQ_AO_FN
  static std::atomic<size_t> lds_set[kMaxDevices][8];
  const int dev = current_device(), ki = (a.head_dim == 32 ? 0 : a.head_dim == 64 ? 1 : a.head_dim == 128 ? 2 : 3) + (lng ? 4 : 0);
  if (lds > 16384 && lds_set[dev][ki].load(std::memory_order_relaxed) < lds) {      // (the kernel holds ~33 KB of static LDS)
    MQ_REQUIRE(hipFuncSetAttribute(fn, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds) == hipSuccess,
               "mq_decode_attention_oproj: %zu bytes of dynamic LDS rejected", lds);
    lds_set[dev][ki].store(lds, std::memory_order_relaxed);
  }
  const unsigned grid = (unsigned)(a.heads * a.slices + (lng ? 0 : a.prefetch_wgs));      // (1024 threads: the prefetch share rides inside the workgroups)
  unsigned long long* stamps = STAMP_SLOT(4, (unsigned)(a.heads * a.slices));
  hipStream_t st = as_stream(stream);
#define MQ_AO_LAUNCH(DD)                                                                                              \
  do {                                                                                                                \
    if (lng) decode_attention_oproj_kernel<DD, AO_THREADS_LONG><<<grid, AO_THREADS_LONG, lds, st>>>(a, stamps);      \
    else decode_attention_oproj_kernel<DD, AO_THREADS><<<grid, AO_THREADS, lds, st>>>(a, stamps);                    \
  } while (0)
  switch (a.head_dim) {
    case 32: MQ_AO_LAUNCH(32); break;
    case 64: MQ_AO_LAUNCH(64); break;
    case 128: MQ_AO_LAUNCH(128); break;
    default: MQ_AO_LAUNCH(256); break;
  }
#undef MQ_AO_LAUNCH
  MQ_LAUNCH_CHECK("mq_decode_attention_oproj");
  return MQ_OK;
}

int mq_decode_embed(const float* table, const int64_t* tok, int64_t hidden, int64_t vocab, const float* cos, const float* sin, const int* pos,
                    int rot_dim, int max_pos, float* x, float* rope_row, mq_stream_t stream) {
  MQ_REQUIRE(table && tok && x && hidden > 0 && vocab > 0, "mq_decode_embed: null pointer / empty table");
  MQ_REQUIRE(!rope_row || (cos && sin && pos && rot_dim > 0 && max_pos > 0), "mq_decode_embed: rope_row needs cos / sin / pos, rot_dim and max_pos");
  const unsigned grid = (unsigned)((hidden + 1023) / 1024 < 64 ? (hidden + 1023) / 1024 : 64);
  decode_embed_kernel<<<grid, 256, 0, as_stream(stream)>>>(table, reinterpret_cast<const long long*>(tok), (int)hidden, (long long)vocab, cos, sin, pos, rot_dim,
                                                          max_pos, x, rope_row);
  MQ_LAUNCH_CHECK("mq_decode_embed");
  return MQ_OK;
}

int mq_decode_head(const float* x, const float* norm_weight, const float* norm_bias, int layernorm, float eps, const float* w, const float* bias,
                   int64_t K, int64_t V, float* logits, mq_stream_t stream) {
  MQ_REQUIRE(x && w && logits && K > 0 && K % 4 == 0 && K <= 12288 && V > 0, "mq_decode_head: bad arguments (K %% 4 == 0, K <= 12288)");
  MQ_REQUIRE(aligned(w, 16), "mq_decode_head: the weight must be 16-byte aligned");
  int64_t blocks = (V + 3) / 4;
  if (blocks > 256 * 8) blocks = 256 * 8;
  decode_head_kernel<<<(unsigned)blocks, 256, (size_t)K * sizeof(float), as_stream(stream)>>>(x, norm_weight, norm_bias, layernorm, eps, w, bias, (int)K, (int)V, logits,
                                                                                    STAMP_SLOT(5, (unsigned)blocks));
  MQ_LAUNCH_CHECK("mq_decode_head");
  return MQ_OK;
}

}  // extern "C"

/*
 * mobilequant_amd.h -- C ABI of libmobilequant_amd.so (MI355X / gfx950 only).
 *
 * The drop-in boundary of the MobileQuant hot path.  The reference has no
 * native boundary for this path: everything happens inside torch ops called from
 * mobilellm/quantization/qmodule.py and ptq/generate_act_range.py.  Each entry point below names
 * the reference code (file:line, relative to the reference checkout) whose arithmetic it replaces.
 * A maintainer binds these with ctypes; mobilequant_amd/_lib.py is that binding.
 *
 * Conventions
 *   - Every pointer is a DEVICE pointer (HBM) unless the name ends in _host.
 *   - `stream` is a hipStream_t passed as void* (NULL = the default stream).  Kernels are enqueued
 *     on it and the call returns without synchronising; no entry point reads results back.
 *   - Tensors are dense row-major.  "rows x cols" views follow the reference:
 *       per-tensor  : one (scale, offset) pair            (n_scale == 1)
 *       per-row     : one pair per row of a [rows, cols] view (weights [N,K]: per output channel,
 *                     qmodule.py:263-264; per-group: the caller views the tensor as [-1, g], :259-260)
 *   - Return value: MQ_OK or an mq_status error; mq_last_error() gives the message for the calling
 *     thread.  Nothing aborts the process.
 *   - Integer storage of 8-bit indices ("i8 storage"): the reference's index q (qmodule.py:286-287)
 *     minus `shift`, where shift = 128 for unsigned grids [0,255] and 0 for signed grids, so every
 *     stored byte is a signed int8 that the MFMA i8 instructions consume directly.  An integer image presupposes an
 *     INTEGRAL offset, which is what the reference produces (offset = -round(beta / scale), qmodule.py:60); with a
 *     fractional offset the index has no integer image and the stored byte is unspecified (rounded or truncated).
 */
#ifndef MOBILEQUANT_AMD_H
#define MOBILEQUANT_AMD_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define MQ_VERSION 301 /* 0.3.1: + mq_calib_norm, mq_calib_gated, mq_calib_rope(_qkv), mq_calib_attention_probs_causal.  0.3.0: ABI BREAK -- argument structs grew at their tails (mq_decode_gemv_args in 0.3.0, mq_attention_args in 0.2.x): a caller MUST compare mq_version() / 100 with MQ_VERSION / 100 of the header it was built against (mobilequant_amd/_lib.py does) before passing a struct; + mq_decode_attention_oproj.  0.2.1: + mq_qmatmul, mq_calib_attention_probs (0.2.0: chan_scale in mq_quantize / mq_quantize_tiled; tuning knobs moved to mobilequant_amd_tuning.h) */

typedef void* mq_stream_t;

typedef enum mq_status {
  MQ_OK = 0,
  MQ_EINVAL = 1,       /* bad argument: null pointer, misaligned pointer, bad shape/dtype */
  MQ_EHIP = 2,         /* a HIP runtime call or kernel launch failed */
  MQ_EUNSUPPORTED = 3  /* valid request this build has no kernel for */
} mq_status;

typedef enum mq_dtype {
  MQ_F32 = 0,
  MQ_F16 = 1,
  MQ_I8 = 2,
  MQ_U8 = 3,
  MQ_I16 = 4,
  MQ_U16 = 5,
  MQ_I32 = 6
} mq_dtype;

/* ---- library ------------------------------------------------------------------------------- */
int mq_version(void);
const char* mq_last_error(void);
/* Device facts used by launch heuristics and by bench.py's roofline (compute units, clock). */
int mq_device_info(int* cu_count, int* max_clock_khz, char* arch_name, size_t arch_name_len);

/* ---- a1: compute_scale_offset_from_min_max  (qmodule.py:40-61) ------------------------------ */
/* scale[i] = clamp(alpha/qmax, 1e-5, 1e6), offset[i] = -rint(beta/scale) for i < n; asymmetric:
 * alpha = max-min, beta = min; symmetric: alpha = max(|min|,|max|), beta = 0 (offset = -0.0f). */
int mq_scale_offset_from_minmax(const float* min_val, const float* max_val, int64_t n, int bitwidth,
                                int is_symmetric, float* scale, float* offset, mq_stream_t stream);

/* ---- a3 / a12 / a13: min/max reductions ----------------------------------------------------- */
/* All reductions ACCUMULATE into their outputs (running min / running max), which is the update
 * rule of update_act_range (ptq/generate_act_range.py:55-69).  Call mq_minmax_init first to start a
 * fresh statistic (min = +inf, max = -inf).  Results are exact (min/max are order independent);
 * -0.0 and +0.0 compare equal and either may be returned. */
int mq_minmax_init(float* min_out, float* max_out, int64_t n, mq_stream_t stream);
/* per-tensor: compute_min_max_from_tensor (qmodule.py:31-33); generate_act_range.py:65 */
int mq_minmax_tensor(const void* x, int dtype, int64_t numel, float* min_out, float* max_out,
                     mq_stream_t stream);
/* per-tensor, FRESH statistic: min_out[0] / max_out[0] are OVERWRITTEN with the tensor's min / max (what a dynamic
 * quantizer or a first-forward weight range needs, qmodule.py:262-277).  Two launches without atomics (per-workgroup
 * partials into `scratch`, >= 1024 floats of device memory the caller owns for the duration of the call on `stream`,
 * then a fold) instead of init + up to 1024 contended atomics.  numel == 0 gives (+inf, -inf). */
int mq_minmax_tensor_fresh(const void* x, int dtype, int64_t numel, float* min_out, float* max_out,
                           float* scratch, int64_t scratch_floats, mq_stream_t stream);
/* per-row of a [rows, cols] view -> min_out[rows], max_out[rows] (qmodule.py:27-30, :263-264) */
int mq_minmax_rows(const void* x, int dtype, int64_t rows, int64_t cols, float* min_out,
                   float* max_out, mq_stream_t stream);
/* per-column of a [rows, cols] view -> min_out[cols], max_out[cols]: per-channel activation
 * statistics (generate_act_range.py:57-63).  SmoothQuant's absmax (generate_act_scale_shift.py:47-53)
 * is max(|min|, |max|) of the same statistic. */
int mq_minmax_cols(const void* x, int dtype, int64_t rows, int64_t cols, float* min_out,
                   float* max_out, mq_stream_t stream);

/* ---- a5: Quantizer.forward (qmodule.py:286-295) --------------------------------------------- */
/* y = (clamp(rint(x / scale) + offset, qmin, qmax) - offset) * scale, IEEE fp32, bit-exact with the
 * reference's CPU path.  x, y: [rows, cols] of `dtype` (MQ_F32 or MQ_F16), may alias.
 * n_scale == 1: per-tensor; n_scale == rows: per-row.  For MQ_F16 the arithmetic follows torch's
 * promotion rule (SURVEY 8a' item 4): per-tensor math rounds to half after every op, per-row math
 * runs in fp32 and rounds once at the end. */
int mq_fake_quant(const void* x, void* y, int dtype, int64_t rows, int64_t cols, const float* scale,
                  const float* offset, int64_t n_scale, float qmin, float qmax, mq_stream_t stream);

/* Backward of the above for the PTQ training loops (mobilellm/quantization/algorithm.py:381, :587), fp32:
 * the gradients torch autograd derives for qmodule.py:286-290 with round_ste (qmodule.py:17-21) -- identity
 * through the rounding, clamp passes the gradient where qmin <= q <= qmax:
 *   grad_x = (g*s)/s inside, 0 clamped;  grad_scale += g*(r - x/s) inside, g*(clamp(q) - o) clamped;
 *   grad_offset += 0 inside, -g*s clamped     (r = rint(x/s), q = r + o)
 * grad_scale / grad_offset ([n_scale]) must be ZERO-INITIALISED by the caller; they are accumulated with float
 * atomics (summation order is not deterministic; tolerance 1e-5 relative). */
int mq_fake_quant_backward(const float* x, const float* grad_y, int64_t rows, int64_t cols,
                           const float* scale, const float* offset, int64_t n_scale, float qmin,
                           float qmax, float* grad_x, float* grad_scale, float* grad_offset,
                           mq_stream_t stream);

/* f3: learnable weight clipping + per-row fake-quant of a weight, one pass per direction (fp32, rows of up to 16384 columns, a
 * multiple of 4).  Replaces, per training forward of a weight quantizer in LWC mode (mobilellm/quantization/qmodule.py:133-185,
 * :259-295 under algorithm.py:381 / :587): torch.amin / amax per output row (:263-268), sigmoid(bound factor) * range (:271-273; the
 * caller passes sig_lo = sigmoid(lowbound_factor), sig_hi = sigmoid(upbound_factor), [rows]), compute_scale_offset_from_min_max
 * (:40-61) and the fake-quant (:286-290) -- the same fp32 operations in the same order, bit-identical values.  Writes the row ranges
 * and the grid ([rows] each) for the backward / for Quantizer.scale, .offset. */
int mq_lwc_fake_quant(const float* w, int64_t rows, int64_t cols, const float* sig_lo, const float* sig_hi, int bitwidth,
                      int is_symmetric, float* out, float* row_min, float* row_max, float* scale, float* offset, mq_stream_t stream);
/* Backward of the above, what torch autograd derives for that chain: grad_w = the straight-through gradient (clamp mask) + the range
 * gradients sent to each row's extreme element(s) (ties share evenly, as amin / amax do); grad_sig_lo / grad_sig_hi [rows].  The
 * offset is -round(beta / scale) and carries no gradient (torch.round has none). */
int mq_lwc_fake_quant_backward(const float* w, const float* grad_out, int64_t rows, int64_t cols, const float* sig_lo,
                               const float* sig_hi, const float* row_min, const float* row_max, int bitwidth, int is_symmetric,
                               float* grad_w, float* grad_sig_lo, float* grad_sig_hi, mq_stream_t stream);

/* f3: what a training-mode attention block does to the [rows = batch * heads * S, cols = keys] scores between its two matmuls
 * (mobilellm/model/hf_model.py:511-520 with the QMatMuls of qmodule.py:408-466, under algorithm.py:381 / :587), in one pass:
 *   out = Q2( softmax( Q1(raw) / sqrt_d + mask ) )
 * Q1 = qk_bmm's output quantizer (s1, o1, [qmin1, qmax1]), Q2 = pv_bmm's input quantizer, both static per-tensor grids read by
 * pointer (learnable ranges), fake-quant arithmetic of mq_fake_quant; softmax in fp32 as torch computes it (max, expf, sum, divide).
 * mask: additive fp32 [mask_rows, cols] (row r uses mask row r % mask_rows) or NULL.  fp32, cols % 4 == 0, cols <= 4096. */
int mq_attention_probs_train(const float* raw, int64_t rows, int64_t cols, const float* mask, int64_t mask_rows, const float* s1,
                             const float* o1, float qmin1, float qmax1, const float* s2, const float* o2, float qmin2, float qmax2,
                             float sqrt_d, float* out, mq_stream_t stream);
/* Backward of the above from the raw scores and the incoming gradient (nothing else is kept): grad_raw, and grad_grids[4] =
 * d s1, d o1, d s2, d o2 accumulated with float atomics into a ZERO-INITIALISED buffer -- the gradients torch autograd derives for
 * the module chain (straight-through rounding, clamp masks, softmax backward p * (g - sum(g p))). */
int mq_attention_probs_train_backward(const float* raw, const float* grad_out, int64_t rows, int64_t cols, const float* mask,
                                      int64_t mask_rows, const float* s1, const float* o1, float qmin1, float qmax1, const float* s2,
                                      const float* o2, float qmin2, float qmax2, float sqrt_d, float* grad_raw, float* grad_grids,
                                      mq_stream_t stream);

/* The integer index itself (qmodule.py:286-287) written as integers instead of being dequantised.
 * q_dtype MQ_I8: i8 storage (index - shift, see top);  MQ_U8 / MQ_I16 / MQ_U16 / MQ_I32: the plain
 * index.  row_sum (nullable, [rows] int32): sum over the row of the STORED values -- the
 * zero-point correction term of the integer GEMM (SURVEY 8a' item 9), produced in the same pass.
 * chan_scale (nullable, [cols] fp32; needs float32 x and a per-tensor grid): SmoothQuant per-channel scale fused in front
 * of the quantizer -- the index of x[m,k] / chan_scale[k] (IEEE divide, then the arithmetic above op for op).  This is the
 * run-time form of the scales the reference folds into the weights (`ln.weight /= s; fc.weight *= s`:
 * ptq/smoothquant.py:64-69, mobilellm/quantization/algorithm.py:47-68) for an activation whose producer cannot absorb
 * 1/s; the consumer's integer weights are then formed from weight * s. */
int mq_quantize(const void* x, int dtype, int64_t rows, int64_t cols, const float* scale,
                const float* offset, int64_t n_scale, float qmin, float qmax, int shift,
                const float* chan_scale, void* q, int q_dtype, int32_t* row_sum, mq_stream_t stream);

/* ---- a8: QLinear.forward as a real-int8 GEMM (qmodule.py:341-358; SURVEY 8a' item 9) --------- */
/* Epilogue vectors of one QLinear, computed on device from quantizer state (no host sync):
 *   alpha[n]    = a_scale * w_scale[n]
 *   w_zp[n]     = (int)w_offset[n] - w_shift                    stored-domain weight zero point
 *   col_term[n] = -za * w_colsum[n] + K * za * w_zp[n],  za = (int)a_offset - a_shift
 * a_scale/a_offset: 1 element.  w_scale/w_offset: n_wscale == 1 or N.  w_colsum[N] = row_sum
 * output of mq_quantize on the weight.  */
int mq_linear_epilogue_prepare(const float* a_scale, const float* a_offset, int a_shift,
                               const float* w_scale, const float* w_offset, int64_t n_wscale,
                               int w_shift, const int32_t* w_colsum, int64_t N, int64_t K,
                               float* alpha, int32_t* w_zp, int32_t* col_term, mq_stream_t stream);

/* out[m,n] = alpha[n] * (sum_k a[m,k]*w[n,k] - w_zp[n]*a_rowsum[m] + col_term[n]) + bias[n]
 * optionally followed by the output quantizer (qmodule.py:356-357).
 *   a [M,K] int8, w [N,K] int8 (i8 storage), K % 64 == 0, pointers 16-byte aligned.
 *   a_rowsum [M] may be NULL when every w_zp is 0 (symmetric weights); bias may be NULL.
 *   out_scale/out_offset (1 element each) NULL -> no output quantizer; out_dtype MQ_F32 / MQ_F16.
 *   With an output quantizer: MQ_F32 / MQ_F16 store the fake-quantised value (what the reference
 *   returns); MQ_U8 / MQ_I8 / MQ_U16 / MQ_I16 store the output index itself (MQ_I8 = index - 128 for an
 *   unsigned grid) so the next integer GEMM can consume it.
 * The int32 contraction is exact.  The output quantizer divides by multiplying with 1/out_scale
 * (<= 1.5 ulp): DESIGN.md states the resulting tolerance against the reference. */
int mq_w8a8_linear(const int8_t* a, const int8_t* w, int64_t M, int64_t N, int64_t K,
                   const int32_t* a_rowsum, const float* alpha, const int32_t* w_zp,
                   const int32_t* col_term, const float* bias, const float* out_scale,
                   const float* out_offset, float out_qmin, float out_qmax, void* out, int out_dtype,
                   mq_stream_t stream);

/* mq_w8a8_linear with fp32 output PLUS a residual: out[m, n] = resid[m, n] + Qout(linear)[m, n] (plain fp32 add, what
 * `x + self_attn(...)` / `x + mlp(...)` do in the decoder layer, hf_model.py:1127-1141) -- the add costs no launch and no extra
 * pass.  Row-major int8 activations, M > 8.  resid may alias nothing the kernel writes (out != resid). */
int mq_w8a8_linear_residual(const int8_t* a, const int8_t* w, int64_t M, int64_t N, int64_t K, const int32_t* a_rowsum,
                            const float* alpha, const int32_t* w_zp, const int32_t* col_term, const float* bias,
                            const float* out_scale, const float* out_offset, float out_qmin, float out_qmax,
                            const float* resid, float* out, mq_stream_t stream);

/* Fragment-blocked activations for the large FFN shapes.  mq_gemm_tiled_supported(M, N, K) != 0 (N a multiple of 176,
 * K a multiple of 256, at least 192 tiles of 256 x 176) means: quantise with mq_quantize_tiled instead of mq_quantize and
 * call mq_w8a8_linear_tiled instead of mq_w8a8_linear -- same arguments and results (bit-identical), ~7 % faster: the
 * GEMM's main loop (generated gfx950 ISA) loads every A fragment with one fully coalesced 1-KiB request straight into
 * registers instead of staging row-major rows through the LDS.
 * Layout of q_tiled (ceil(rows/16)*16 * cols bytes): 1-KiB blocks of 16 rows x 64 k ordered [row block][k block]; inside
 * a block byte offset 16 * ((row & 15) + 16 * ((k & 63) >> 4)) + (k & 15).  Rows past `rows` are padding.
 * mq_quantize_tiled: per-tensor grid (scale/offset: 1 element), int8 storage (index - shift), cols % 128 == 0; row_sum
 * (nullable) and chan_scale (nullable, [cols], 16-byte aligned, float32 x only) as in mq_quantize. */
int mq_gemm_tiled_supported(int64_t M, int64_t N, int64_t K);
int mq_quantize_tiled(const void* x, int dtype, int64_t rows, int64_t cols, const float* scale,
                      const float* offset, float qmin, float qmax, int shift, const float* chan_scale,
                      int8_t* q_tiled, int32_t* row_sum, mq_stream_t stream);
int mq_w8a8_linear_tiled(const int8_t* a_tiled, const int8_t* w, int64_t M, int64_t N, int64_t K,
                         const int32_t* a_rowsum, const float* alpha, const int32_t* w_zp,
                         const int32_t* col_term, const float* bias, const float* out_scale,
                         const float* out_offset, float out_qmin, float out_qmax, void* out,
                         int out_dtype, mq_stream_t stream);

/* QLinear.forward (mobilellm/quantization/qmodule.py:341-358) with PER-GROUP weight grids (Quantizer with group_size != -1,
 * qmodule.py:259-260, :292-293; --group_size of ptq/mobilequant.py:41, :157) on the int8 MFMA units:
 *   out[m, n] = sum_g alpha[g, n] * float( sum_{k in g} a_q[m, k] w_q[n, k] + cw[g, n] * a_gsum[g, m] + t[g, n] ) + bias[n]
 * a_q [M, K], w_q [N, K]: stored int8 values (index - shift); group g = input channels [g group_size, (g + 1) group_size);
 * a_gsum [G, M]: per-group sums of the stored activations; alpha [G, N] = s_a s_w; cw [G, N] = w_shift - o_w; t [G, N] =
 * c_a W_g + group_size c_a cw with c_a = a_shift - z_a and W_g the per-group sums of the stored weights (all exact integers: the
 * bracket is the reference's sum of (ia - z_a)(iw - o_w) over the group).  fp32 output (an output quantizer follows as its own
 * kernel).  group_size % 64 == 0, K % group_size == 0, N % 128 == 0. */
int mq_w8a8_linear_grouped(const int8_t* a_q, const int8_t* w_q, int64_t M, int64_t N, int64_t K, int64_t group_size,
                           const int32_t* a_gsum, const float* alpha, const int32_t* cw, const int32_t* t, const float* bias,
                           float* out, mq_stream_t stream);

/* The same fragment-blocked path for outputs that do not tile by 176 (N = 2048: o_proj / w2; N = 2560: q | k | v): generated gfx950
 * ISA on 128-column tiles.  mq_gemm_tiled128_supported(M, N, K) != 0: N % 128 == 0, K % 256 == 0, K >= 768.
 *   mq_w8a8_linear_tiled_residual  = mq_w8a8_linear_residual on fragment-blocked activations, for a 16-bit output grid
 *     (out_qmax - out_qmin > 255; QLinear.forward + `x + ...`, qmodule.py:341-358 / hf_model.py:1127-1141): same operations in
 *     the same order, bit-identical results.  128 x 128 tiles (one per CU at 2048 x 2048 outputs), 256 x 128 beyond 512 tiles.
 *   mq_w8a8_linear_tiled_segmented = mq_w8a8_linear_segmented on fragment-blocked activations (bit-identical indices). */
int mq_gemm_tiled128_supported(int64_t M, int64_t N, int64_t K);
int mq_w8a8_linear_tiled_residual(const int8_t* a_tiled, const int8_t* w, int64_t M, int64_t N, int64_t K, const int32_t* a_rowsum,
                                  const float* alpha, const int32_t* w_zp, const int32_t* col_term, const float* bias,
                                  const float* out_scale, const float* out_offset, float out_qmin, float out_qmax,
                                  const float* resid, float* out, mq_stream_t stream);
/* The same with PACKED 4-bit weights (mq_pack_w4's image, [N, K / 2] bytes): the packed pieces are expanded once per workgroup into the
 * int8 ring of the generated 128 x 128 program (tools/gen_fr_asm.py frw4x_128r).  One weight image for prefill and decode. */
int mq_w4a8_linear_tiled_residual(const int8_t* a_tiled, const uint8_t* w_packed, int64_t M, int64_t N, int64_t K, const int32_t* a_rowsum,
                                  const float* alpha, const int32_t* w_zp, const int32_t* col_term, const float* bias,
                                  const float* out_scale, const float* out_offset, float out_qmin, float out_qmax,
                                  const float* resid, float* out, mq_stream_t stream);

/* Two QLinears over the SAME activation in one launch -- w1 / w3 of a gated FFN receive the same tensor (hf_model.py:1057:
 * w2(act(w1(x)) * w3(x))).  Both problems have the shape M x N x K (mq_gemm_tiled_supported, K % 256 == 0, K >= 768), their
 * own weights, epilogue vectors and 8-bit unsigned output grid (out_qmin 0, out_qmax 255), and write the output INDICES
 * (out_dtype MQ_U8, or MQ_I8 = index - 128) the next integer kernel consumes.  Results are bit-identical to two
 * mq_w8a8_linear_tiled calls; the launch streams the activation panel once per tile pair and pays the kernel boundary once. */
int mq_w8a8_linear_tiled_pair(const int8_t* a_tiled, int64_t M, int64_t N, int64_t K, const int32_t* a_rowsum,
                              const int8_t* w0, const float* alpha0, const int32_t* w_zp0,
                              const int32_t* col_term0, const float* bias0, const float* out_scale0,
                              const float* out_offset0, void* out0, const int8_t* w1, const float* alpha1,
                              const int32_t* w_zp1, const int32_t* col_term1, const float* bias1,
                              const float* out_scale1, const float* out_offset1, void* out1, int out_dtype,
                              mq_stream_t stream);

/* The gated FFN's first three modules in TWO launches: w1 as mq_w8a8_linear_tiled (indices into idx_scratch [M, N] u8), then w3 on
 * the same program whose epilogue looks (w1 index, w3 index) up in `table` (mq_gated_table; 64 KiB, LDS-resident) and writes w2's int8
 * input image q_tiled (fragment-blocked [ceil16(M), N]) + row_sum [M] -- what mq_w8a8_linear_tiled_pair + mq_gated_lookup_tiled
 * produce in two launches and 35 MB more traffic; bit-identical image and row sums.  Shapes as the pair launch, N % 64 == 0. */
int mq_w8a8_linear_tiled_gated(const int8_t* a_tiled, int64_t M, int64_t N, int64_t K, const int32_t* a_rowsum,
                               const int8_t* w0, const float* alpha0, const int32_t* w_zp0, const int32_t* col_term0,
                               const float* bias0, const float* out_scale0, const float* out_offset0,
                               const int8_t* w1, const float* alpha1, const int32_t* w_zp1, const int32_t* col_term1,
                               const float* bias1, const float* out_scale1, const float* out_offset1,
                               const int8_t* table, uint8_t* idx_scratch, int8_t* q_tiled, int32_t* row_sum, mq_stream_t stream);
/* The same from PACKED 4-bit weights (two mq_pack_w4 images, [N, K / 2] bytes each): w1 on the packed index kernel, w3 on the packed
 * gate-epilogue kernel (tools/gen_fr_asm.py frw4x / frgw4x and their 128-column forms).  Identical bytes. */
int mq_w4a8_linear_tiled_gated(const int8_t* a_tiled, int64_t M, int64_t N, int64_t K, const int32_t* a_rowsum,
                               const uint8_t* w0, const float* alpha0, const int32_t* w_zp0, const int32_t* col_term0,
                               const float* bias0, const float* out_scale0, const float* out_offset0,
                               const uint8_t* w1, const float* alpha1, const int32_t* w_zp1, const int32_t* col_term1,
                               const float* bias1, const float* out_scale1, const float* out_offset1,
                               const int8_t* table, uint8_t* idx_scratch, int8_t* q_tiled, int32_t* row_sum, mq_stream_t stream);

/* Decode shapes (M <= 8 tokens, M*K < 64 KiB, K % 256 == 0): the activation quantizer (qmodule.py:349-351) fused
 * into the weight-streaming GEMV -- x is the fp32 [M,K] activation, quantised on the fly to its grid
 * (a_scale/a_offset: 1 element; a_shift as in mq_quantize) with the row sums reduced in LDS; alpha / w_zp /
 * col_term are the vectors mq_linear_epilogue_prepare made for that same grid.  Other shapes: MQ_EUNSUPPORTED
 * (use mq_quantize + mq_w8a8_linear; mq_w8a8_linear itself switches to the GEMV kernel for M <= 8). */
int mq_w8a8_linear_f32in(const float* x, const float* a_scale, const float* a_offset, float a_qmin,
                         float a_qmax, int a_shift, const int8_t* w, int64_t M, int64_t N, int64_t K,
                         const float* alpha, const int32_t* w_zp, const int32_t* col_term,
                         const float* bias, const float* out_scale, const float* out_offset,
                         float out_qmin, float out_qmax, void* out, int out_dtype, mq_stream_t stream);

/* W4A8: weights as packed 4-bit indices.  mq_pack_w4 packs an [N,K] tensor of UNSIGNED nibbles
 * (index - qmin, 0..15, one per byte) two per byte, K-interleaved in blocks of 32: byte j (0..15)
 * of each 16-byte group holds element j in its low nibble and element j+16 in its high nibble.
 * K % 64 == 0.  mq_w4a8_linear unpacks in registers and runs the same int8 MFMA contraction; its
 * w_zp / col_term are in the unsigned-nibble domain (w_shift = qmin). */
int mq_pack_w4(const uint8_t* nibbles, int64_t N, int64_t K, uint8_t* packed, mq_stream_t stream);
int mq_w4a8_linear(const int8_t* a, const uint8_t* w_packed, int64_t M, int64_t N, int64_t K,
                   const int32_t* a_rowsum, const float* alpha, const int32_t* w_zp,
                   const int32_t* col_term, const float* bias, const float* out_scale,
                   const float* out_offset, float out_qmin, float out_qmax, void* out, int out_dtype,
                   mq_stream_t stream);

/* W4A8 decode shapes: mq_w4a8_linear itself streams the packed nibbles through the GEMV kernel for M <= 8 (half the
 * weight bytes of W8 per token); mq_w4a8_linear_f32in is mq_w8a8_linear_f32in for packed 4-bit weights (same shape
 * limits, K % 256 == 0).  This is the reference's deployment mode (W4A8 LLaMA-1.1B) at decode time. */
int mq_w4a8_linear_f32in(const float* x, const float* a_scale, const float* a_offset, float a_qmin,
                         float a_qmax, int a_shift, const uint8_t* w_packed, int64_t M, int64_t N, int64_t K,
                         const float* alpha, const int32_t* w_zp, const int32_t* col_term,
                         const float* bias, const float* out_scale, const float* out_offset,
                         float out_qmin, float out_qmax, void* out, int out_dtype, mq_stream_t stream);

/* ---- a10: QRMSNorm.forward in one pass (qmodule.py:469-530 around hf_model.py:184-195) ---------------------- */
/* out = Qout( weight * (xi * 1/sqrt(mean(xi^2) + eps)) (+ bias) ),  xi = Qin(x); x, y [rows, cols] fp32, cols % 4 == 0.
 *   weight [cols] (+ bias [cols], nullable): ALREADY fake-quantised by the caller (weight quantizer, once per weight).
 *   in_scale/in_offset, out_scale/out_offset: 1 element each, NULL -> that quantizer is absent.
 *   y (nullable): the fp32 result the reference returns.  q_out (nullable, needs the output quantizer): the same
 *   result as int8 storage (index - q_shift) plus row_sum (nullable, [rows]) = sum of the stored values per row, i.e.
 *   exactly what mq_quantize(want row sums) would produce from y -- the consumer linears skip their quantize launch.
 *   q_tiled (nullable, cols % 64 == 0, ceil(rows/16)*16 * cols bytes): the same integer image in the fragment-blocked
 *   layout of mq_quantize_tiled, for consumers served by mq_w8a8_linear_tiled (w1 / w3).
 * Elementwise arithmetic is the reference's op for op; the sum of squares is reduced in another order than torch's,
 * so an output within ~1e-7 relative of a rounding boundary may land on the neighbouring grid point. */
int mq_rmsnorm_quant(const float* x, int64_t rows, int64_t cols, const float* weight, const float* bias,
                     float eps, const float* in_scale, const float* in_offset, float in_qmin,
                     float in_qmax, const float* out_scale, const float* out_offset, float out_qmin,
                     float out_qmax, float* y, int8_t* q_out, int8_t* q_tiled, int q_shift,
                     int32_t* row_sum, mq_stream_t stream);

/* QLayerNorm.forward (qmodule.py:624-640 around F.layer_norm): same contract as mq_rmsnorm_quant with the row's mean and
 * biased variance, y = (xi * rstd + (-rstd * mean)) * weight + bias (torch's CPU kernel expression); bias is the raw
 * (unquantised) LayerNorm bias, nullable. */
int mq_layernorm_quant(const float* x, int64_t rows, int64_t cols, const float* weight, const float* bias,
                       float eps, const float* in_scale, const float* in_offset, float in_qmin,
                       float in_qmax, const float* out_scale, const float* out_offset, float out_qmin,
                       float out_qmax, float* y, int8_t* q_out, int8_t* q_tiled, int q_shift,
                       int32_t* row_sum, mq_stream_t stream);

/* ---- a10: QSiLU / QGELU.forward in one pass (qmodule.py:739-754, :790-798) ------------------------------------ */
/* act 0 (SiLU): y = Qout( xi * Qmid(sigmoid(xi)) ), act 1 (GELU, erf form): y = Qout( gelu(xi) ), xi = Qin(x); fp32,
 * per-tensor grids (1 element each; NULL pair = quantizer absent; mid is ignored for GELU).  exp / erf come from the
 * device math library: after the output quantizer at most one LSB from the CPU reference on a vanishing fraction of
 * the elements. */
int mq_act_quant(const float* x, int64_t numel, int act, const float* in_scale, const float* in_offset,
                 float in_qmin, float in_qmax, const float* mid_scale, const float* mid_offset,
                 float mid_qmin, float mid_qmax, const float* out_scale, const float* out_offset,
                 float out_qmin, float out_qmax, float* y, mq_stream_t stream);

/* ---- f1: the gated FFN's act(w1(x)) * w3(x) -> integer input image of w2 (hf_model.py:1057; qmodule.py:739-753) -------- */
/* p = Qact(act(va)) * vb (QSiLU: act(va) = va * Qmid(sigmoid(va)); QGELU: gelu(va); the product itself is not quantised in
 * the reference), then w2's input quantizer: q_out = int8 storage (index - q_shift) of p on the output grid, row_sum as
 * mq_quantize would produce it, y (nullable) = p in fp32.  in_dtype MQ_F32: a / b are fp32 values.  in_dtype MQ_U8: a / b are
 * the 8-bit output INDICES written by the w1 / w3 GEMMs (mq_w8a8_linear_tiled_pair) on the grids (a_scale, a_offset) /
 * (b_scale, b_offset) -- the kernel dequantises (q - offset) * scale, bit-identical to the fp32 values of the fake-quant
 * path, and the fp32 intermediates never exist in memory.  Per-tensor grids (1 element; NULL pair = quantizer absent; mid is
 * ignored for GELU); cols % 16 == 0.  Same arithmetic and tolerance as mq_act_quant. */
int mq_gated_act_quant(const void* a, const void* b, int in_dtype, int64_t rows, int64_t cols, int act,
                       const float* a_scale, const float* a_offset, const float* b_scale,
                       const float* b_offset, const float* mid_scale, const float* mid_offset,
                       float mid_qmin, float mid_qmax, const float* act_scale, const float* act_offset,
                       float act_qmin, float act_qmax, const float* out_scale, const float* out_offset,
                       float out_qmin, float out_qmax, int q_shift, int8_t* q_out, int32_t* row_sum,
                       float* y, mq_stream_t stream);

/* The same map as a table: with index inputs and static grids, act(a) * b -> w2's input index is a function of the two 8-bit
 * indices.  mq_gated_table evaluates mq_gated_act_quant's per-element arithmetic for all 65 536 (ia, ib) pairs -> table[ia * 256 + ib]
 * (int8 storage, index - q_shift; build once per set of grids); mq_gated_lookup then maps index tensors [rows, cols] (cols % 8 == 0)
 * to the int8 image + row sums with one LDS byte read per element -- bit-identical to mq_gated_act_quant by construction. */
int mq_gated_table(int act, const float* a_scale, const float* a_offset, const float* b_scale, const float* b_offset,
                   const float* mid_scale, const float* mid_offset, float mid_qmin, float mid_qmax, const float* act_scale,
                   const float* act_offset, float act_qmin, float act_qmax, const float* out_scale, const float* out_offset,
                   float out_qmin, float out_qmax, int q_shift, int8_t* table, mq_stream_t stream);
int mq_gated_lookup(const uint8_t* a, const uint8_t* b, int64_t rows, int64_t cols, const int8_t* table, int8_t* q_out,
                    int32_t* row_sum, mq_stream_t stream);
/* the same lookup writing the fragment-blocked layout of mq_quantize_tiled (cols % 64 == 0, ceil(rows/16)*16 * cols bytes, 16-byte
 * aligned): the image mq_w8a8_linear_tiled_residual (w2) reads. */
int mq_gated_lookup_tiled(const uint8_t* a, const uint8_t* b, int64_t rows, int64_t cols, const int8_t* table, int8_t* q_tiled,
                          int32_t* row_sum, mq_stream_t stream);

/* ---- f2: single-token decode step (mobilellm/model/sim_model.py:160-221 on the quantized module graph) -------------------- */
/* A per-tensor quantizer grid on the device: scale / offset point at 1 float each; scale == NULL means "no quantizer here". */
typedef struct mq_grid {
  const float* scale;
  const float* offset;
  float qmin, qmax;
} mq_grid;

/* One fused weight-streaming phase of a decoder layer at M = 1 (DESIGN.md "Decode").  The activation is either fp32 x [K]
 * (quantised on a_grid -- 8-bit unsigned -- inside the kernel; with norm_w != NULL the QRMSNorm of qmodule.py:515-531 runs first:
 * norm_in = its input grid, norm_w = its fake-quantised weight vector, a_grid = its output grid) or a ready int8 image xq [K]
 * (index - 128).  Weights: int8 [N, K] (index - 128), per-row epilogue vectors of mq_linear_epilogue_prepare for a_grid.
 * Plain mode (gate_q == NULL): y[n] = (resid ? resid[n] : 0) + Qout_seg(n)(alpha[n] * (...) + bias[n]); rows [0, seg_end[0]) use
 *   out_grid[0], [seg_end[0], seg_end[1]) out_grid[1], the rest out_grid[2] (q | k | v in one stream).
 * Gate mode (gate_q != NULL): weight rows 2i / 2i+1 are row i of w1 / w3 (out_grid[0] / out_grid[1]); the epilogue applies
 *   QSiLU (gate_act 0; gate_mid = sigmoid grid) or QGELU (1) with output grid gate_actout, the product, and w2's input quantizer
 *   gate_out: gate_q[i] = int8 storage (index - 128); y (nullable) = the fp32 product. */
/* Launch constants of the decode kernels: n <= 16 static per-tensor grids -> consts[4k .. 4k+3] = {scale, offset, 1 / scale, 0}
 * (an absent grid, scale == NULL, packs as {1, 0, 1, 0}).  One tiny launch at engine-build time; the decode kernels then fetch all
 * their quantizer parameters with ONE load instead of two dependent pointer chases per grid (qmodule.py:279-283 keeps scale / offset
 * as device tensors, so they cannot travel as kernel arguments without a host read-back). */
#define MQ_DECODE_MAX_GRIDS 16
typedef struct mq_decode_grid_pack {
  mq_grid grids[MQ_DECODE_MAX_GRIDS];
  int n;
} mq_decode_grid_pack;
int mq_decode_pack_grids(const mq_grid* grids, int n, float* consts, mq_stream_t stream);

typedef struct mq_decode_gemv_args {
  const float* x;
  const int8_t* xq;
  int K, N;
  const float* norm_w;
  const float* norm_bias; /* layernorm = 1 only (nullable) */
  int layernorm;          /* 0: QRMSNorm (qmodule.py:515-531); 1: QLayerNorm (qmodule.py:624-640) in the fused prologue */
  mq_grid norm_in;
  float eps;
  mq_grid a_grid;
  const int8_t* w;
  const float* alpha;
  const int32_t* w_zp;
  const int32_t* col_term;
  const float* bias;
  int seg_end[2];
  mq_grid out_grid[3];
  const float* resid;
  float* y;
  int gate_act;
  mq_grid gate_mid, gate_actout, gate_out;
  int8_t* gate_q;
  int w4; /* 1: w holds packed unsigned nibbles [N, K/2] (mq_pack_w4; gate mode: rows 2i / 2i+1 interleaved alike), w_zp / col_term in
           * the unsigned-nibble domain as for mq_w4a8_linear -- the reference's W4A8 deployment mode */
  const float* consts; /* REQUIRED: mq_decode_pack_grids of {norm_in, a_grid, out_grid[0..2], gate_mid, gate_actout, gate_out} into
                        * a 64-float, 16-byte aligned block (unused tail zero): the kernel reads every grid from this one cache line; the mq_grid pointers above only say
                        * which grids are present (scale != NULL) and carry qmin / qmax */
  /* ---- round 6: four launches per layer.  Every field below may stay zero: the launch is then the one above. ----
   * zero_acc (nullable): the launch also clears zero_n int32 accumulators (the o_proj sums of mq_decode_attention_oproj, which comes
   * next in the chain). */
  int32_t* zero_acc;
  int zero_n;
  /* o_proj's epilogue as the prologue of the NEXT linear (o_acc != NULL; needs norm_w and fp32 x, K <= 4096): the activation row is
   * x[k] + Qo_out(o_alpha[k] * (o_acc[k] + o_ct[k]) + o_bias[k]) -- the residual add of the attention block on o_proj's integer sums
   * left by mq_decode_attention_oproj (o_out: slot 8 of consts) -- and is also stored to x_mid (each workgroup a share), the
   * residual input of the w2 launch. */
  const int32_t* o_acc;
  const float* o_alpha;
  const int32_t* o_ct;
  const float* o_bias;
  mq_grid o_out;
  float* x_mid;
} mq_decode_gemv_args;
int mq_decode_gemv(const mq_decode_gemv_args* args, mq_stream_t stream);
/* How mq_decode_gemv spreads a launch's weight rows over its workgroups: workgroup b reads bytes [b, b + 1) * bytes_per_workgroup of
 * the weight image (for mq_decode_attention's L2 prefetch rows). */
int mq_decode_gemv_geometry(const mq_decode_gemv_args* args, int64_t* workgroups, int64_t* bytes_per_workgroup, int64_t* total_bytes);

/* Attention of one query token over a static INTEGER KV cache (hf_model.py:486-534; QMatMul qk_bmm / pv_bmm of qmodule.py:453-466):
 * qkv = [heads*D | kv_heads*D | kv_heads*D] fp32 outputs of the q|k|v phase; RoPE (rotate-half over the first rot_dim dims; cos /
 * sin [max_pos, rot_dim]) at position *pos (device memory: one captured graph serves every step; *pos >= cache_len: the launch
 * does nothing).  k_cache / v_cache [kv_heads, cache_len, D] int8 hold the post-RoPE keys / the values of positions < *pos as
 * INDICES (index - 128) on their QMatMul input grids (qk_b / pv_b: the reference re-quantises the cached tensors at every step
 * with static grids, which is idempotent) and receive position *pos.  Both contractions are exact integer sums (as
 * mq_attention_quant); quantizers in their exact divide form; softmax in fp32.
 * Outputs: out (nullable) [heads*D] fp32 = pv_bmm's (quantised) output; out_q (nullable) [heads*D] int8 = its index (- 128) on the
 * consumer linear's input grid o_in (o_proj's int8 image: mq_decode_gemv with xq).
 * nsplit workgroups per head share the cached positions in 64-position blocks (nsplit > 1: part [nsplit, heads*D] int64 scratch and
 * ticket [heads] uint32, zeroed once by the caller, self-resetting).  consts: mq_decode_pack_grids of {qk_a, qk_b, qk_out, pv_a, pv_b,
 * pv_out, o_in} into a 64-float block. */
typedef struct mq_decode_attention_args {
  const float* qkv;
  int8_t* k_cache;
  int8_t* v_cache;
  const float* cos;
  const float* sin;
  const int* pos;
  int heads, kv_heads, head_dim, cache_len, rot_dim, nsplit;
  mq_grid qk_a, qk_b, qk_out, pv_a, pv_b, pv_out, o_in;
  const float* consts;
  float* out;
  int8_t* out_q;
  long long* part;
  unsigned* ticket;
  /* optional L2 prefetch rows (prefetch_wgs = 0: none): prefetch_wgs extra workgroups of this launch read the first
   * prefetch_bytes_per_wg bytes of every prefetch_stride-byte piece of prefetch[0, prefetch_total) -- the weights of a LATER
   * mq_decode_gemv launch, piece b = what its workgroup b reads (mq_decode_gemv_geometry) -- into the XCDs' L2s while the attention
   * leaves the memory fabric idle, starting prefetch_delay x 10 ns into the launch.  Never affects results. */
  const int8_t* prefetch;
  int64_t prefetch_bytes_per_wg, prefetch_stride, prefetch_total;
  int prefetch_wgs, prefetch_delay;
} mq_decode_attention_args;
int mq_decode_attention(const mq_decode_attention_args* args, mq_stream_t stream);

/* Round 6: the attention of one query token AND o_proj's contraction in one launch (hf_model.py:486-534 + the o_proj QLinear,
 * qmodule.py:341-358).  Inputs, cache semantics, grids and arithmetic are mq_decode_attention's (qkv fp32, RoPE at *pos, cache append,
 * exact integer qk / pv, quantizers in their divide form) EXCEPT the layout of v_cache: [kv_heads][cache_len / 16][head_dim][16] (16-position
 * chunks of all dimensions; cache_len % 16 == 0).  heads x slices workgroups: workgroup (h, c) computes head h's attention
 * (every slice repeats it: at M = 1 the arithmetic is free, a launch boundary is not; the head's first slice appends to the cache), puts
 * the head's output on o_proj's input grid (o_in) and adds ITS share of o_proj -- rows [c, c + 1) * N / slices of the K-slice
 * [h D, (h + 1) D): o_w [heads][N][D] int8 (index - 128; one byte per weight also for 4-bit weights) -- to the int32 accumulators
 * o_acc [N] with device-scope integer atomics (exact, order free): o_acc[n] += sum_d w[h][n][d] a8[h][d] - o_wzp[n] sum_d a8[h][d].
 * o_proj's epilogue (col_term, alpha, bias, output quantizer, residual add) runs in the next launch's prologue
 * (mq_decode_gemv_args.o_acc).  o_acc must be zero at launch (mq_decode_gemv_args.zero_acc).  consts: mq_decode_pack_grids of {qk_a,
 * qk_b, qk_out, pv_a, pv_b, pv_out, o_in}.  out_q (nullable): the heads' int8 outputs (what mq_decode_attention writes), for tests.
 * tpr: threads per o_proj row (1, 2 or 4; N / slices * tpr <= 256, head_dim / 16 / tpr <= 8 chunks per thread).  Prefetch rows as in
 * mq_decode_attention. */
typedef struct mq_decode_attention_oproj_args {
  const float* qkv;
  int8_t* k_cache;
  int8_t* v_cache;
  const float* rope_row; /* {cos[*pos][0 .. rot_dim), sin[*pos][0 .. rot_dim)}: staged once per token by mq_decode_embed */
  const int* pos;
  int heads, kv_heads, head_dim, cache_len, rot_dim;
  mq_grid qk_a, qk_b, qk_out, pv_a, pv_b, pv_out, o_in;
  const float* consts;
  const int8_t* o_w;
  const int32_t* o_wzp;
  int32_t* o_acc;
  int N, slices, tpr;
  int lg_slices, lg_group, lg_kv; /* log2 of slices, heads / kv_heads, kv_heads when all three are powers of two (the workgroup -> (head,
                                   * slice) mapping then needs no integer division in front of the launch's first request); lg_slices = -1: generic */
  int8_t* out_q;
  const int8_t* prefetch;
  int64_t prefetch_bytes_per_wg, prefetch_stride, prefetch_total;
  int prefetch_wgs, prefetch_delay;
  int threads; /* 0 / 256: the launch for short caches; 1024: four times the lanes per workgroup (same results) -- faster from a few hundred
                * cached positions on, slower below; the prefetch share (<= 48 KiB per workgroup) is then requested by the attention workgroups themselves */
} mq_decode_attention_oproj_args;
int mq_decode_attention_oproj(const mq_decode_attention_oproj_args* args, mq_stream_t stream);
/* Token start (sim_model.py:160-175: embed_tokens of the new token): x [hidden] <- table [vocab, hidden] row *tok, and (rope_row != NULL)
 * rope_row [2 * rot_dim] <- {cos[*pos], sin[*pos]} of the tables cos / sin [max_pos, rot_dim] -- the one launch of a token that chases
 * *pos, so that every layer's mq_decode_attention_oproj reads a fixed address. */
int mq_decode_embed(const float* table, const int64_t* tok, int64_t hidden, int64_t vocab, const float* cos, const float* sin, const int* pos,
                    int rot_dim, int max_pos, float* x, float* rope_row, mq_stream_t stream);

/* Final norm (floating point: the surgery skips it, qmodule.py:843) fused in front of the fp32 lm_head stream: logits[v] = sum_k
 * w[v,k] * norm(x)[k] (+ bias[v]).  layernorm = 0: HFRMSNorm (norm_weight NULL = no norm, norm_bias unused); layernorm = 1:
 * nn.LayerNorm with optional weight / bias (StableLM-2: hf_model.py:1440-1441). */
int mq_decode_head(const float* x, const float* norm_weight, const float* norm_bias, int layernorm, float eps, const float* w,
                   const float* bias, int64_t K, int64_t V, float* logits, mq_stream_t stream);

/* ---- a10: quantized causal attention at prefill (hf_model.py:486-534 with the two QMatMuls of qmodule.py:453-466) ------------ */
/* One sequence.  q [seq, heads*D], k / v [seq, kv_heads*D] fp32 = the q / k / v projection outputs BEFORE RoPE; cos / sin [seq, D]
 * (rotate-half); out [seq, heads*D] fp32 = pv_bmm's (quantised when pv_out.scale != NULL) output in o_proj's input layout.
 * Grids: qk_a / qk_b / pv_b 8-bit unsigned per tensor; qk_out (nullable scale = no output quantizer) and pv_a (<= 16 bit unsigned).
 * Causal mask only (the mask of hf_model.py:1180-1205 at prefill); the scores are divided by sqrt(D) AFTER qk_out, as the reference.
 * Scratch (caller-owned, overwritten): q_i8 [heads][seq][D], k_i8 [kv_heads][seq][D], vt_i8 [kv_heads][seq/64][D][64] (values
 * transposed, keys permuted inside each 64-block), q_rowsum [heads][seq], k_rowsum [kv_heads][seq] (the zero-point terms of the integer
 * q.k^T, derived from the row sums of the images).  q_i8 / q_rowsum may be left untouched: at head_dim 64 with a 16-bit score grid and
 * full rotary the attention workgroups prepare their own q rows in registers (mq_attention_set_fused_q in the tuning header).
 * Limits: head_dim 64, 128 or 256 (every "64" of a layout above reads head_dim; the int8 output image is [rows, heads*head_dim]),
 * seq % 64 == 0, seq <= 65536.  The integer contractions are exact; see DESIGN.md 4.5 for the rounding points. */
/* q | k | v (or any 1..3 linears reading one activation) as ONE int8 GEMM whose column segments carry their own 8-bit unsigned
 * output grids: weights / epilogue vectors concatenated along N, segment i = columns [seg_end[i-1], seg_end[i]) (seg_end[-1] = 0,
 * seg_end[n_segments-1] = N, multiples of 4) quantised on grids[i]; out = uint8 indices [M, N] -- per column exactly the index
 * mq_w8a8_linear writes for that linear alone.  Row-major int8 activations, M > 8. */
int mq_w8a8_linear_segmented(const int8_t* a, const int8_t* w, int64_t M, int64_t N, int64_t K, const int32_t* a_rowsum,
                             const float* alpha, const int32_t* w_zp, const int32_t* col_term, const float* bias, int n_segments,
                             const int64_t* seg_end, const mq_grid* grids, uint8_t* out, mq_stream_t stream);

/* the same for packed 4-bit weights (mq_pack_w4 layout, w_zp / col_term in the unsigned-nibble domain as for mq_w4a8_linear) */
int mq_w4a8_linear_segmented(const int8_t* a, const uint8_t* w_packed, int64_t M, int64_t N, int64_t K, const int32_t* a_rowsum,
                             const float* alpha, const int32_t* w_zp, const int32_t* col_term, const float* bias, int n_segments,
                             const int64_t* seg_end, const mq_grid* grids, uint8_t* out, mq_stream_t stream);
int mq_w8a8_linear_tiled_segmented(const int8_t* a_tiled, const int8_t* w, int64_t M, int64_t N, int64_t K, const int32_t* a_rowsum,
                                   const float* alpha, const int32_t* w_zp, const int32_t* col_term, const float* bias, int n_segments,
                                   const int64_t* seg_end, const mq_grid* grids, uint8_t* out, mq_stream_t stream);

/* Packed 4-bit weights on the PRODUCTION prefill path (round 4; BASELINE.json configs[3]: "packed 4-bit weights, in-register unpack ->
 * MFMA_I32_16x16x64_I8"): QLinear.forward with a 4-bit weight quantizer (qmodule.py:341-358 under the W4A8 recipes,
 * experiments/w4a8/main/e2e_gemma-s1024-ep60-sym.sh:17-23) on fragment-blocked activations (mq_quantize_tiled) and the mq_pack_w4 image
 * [N, K/2] -- the SAME image the decode kernels stream.  Whole-kernel generated gfx950 ISA (tools/gen_fr_asm.py: frw4 / frw4_128): the
 * packed rows go through the LDS ring by LDS-DMA, one ds_read_b128 per 16 output columns and K = 128 stage, nibbles split with v_and /
 * v_lshrrev under the MFMAs.  1..3 column segments with their own 8-bit unsigned output grids (n_segments = 1: seg_end may be NULL);
 * out = uint8 indices (MQ_U8) or index - 128 (MQ_I8), [M, N] row-major; w_zp / col_term in the unsigned-nibble domain as for
 * mq_w4a8_linear.  Shapes: mq_gemm_tiled_w4_supported (N % 176 == 0, or N % 128 == 0 -- required for more than one segment; K % 256 == 0,
 * K >= 768).  Indices are exactly those of mq_w4a8_linear / mq_w4a8_linear_segmented. */
int mq_gemm_tiled_w4_supported(int64_t M, int64_t N, int64_t K);
int mq_w4a8_linear_tiled(const int8_t* a_tiled, const uint8_t* w_packed, int64_t M, int64_t N, int64_t K, const int32_t* a_rowsum,
                         const float* alpha, const int32_t* w_zp, const int32_t* col_term, const float* bias, int n_segments,
                         const int64_t* seg_end, const mq_grid* grids, void* out, int out_dtype, mq_stream_t stream);

typedef struct mq_attention_args {
  const float* q;
  const float* k;
  const float* v;
  const float* cos;
  const float* sin;
  int seq, heads, kv_heads, head_dim;
  float inv_sqrt_d;
  mq_grid qk_a, qk_b, qk_out, pv_a, pv_b, pv_out;
  float* out;
  int8_t* q_i8;
  int8_t* k_i8;
  int8_t* vt_i8;
  int32_t* q_rowsum;
  int32_t* k_rowsum;
  /* optional second output for the consumer linear (o_proj): pv_out indices as its int8 input image (storage = index - out_shift)
   * + row sums: row-major [rows, heads*64] (out_i8_tiled = 0) or the fragment-blocked [ceil16(rows), heads*64] layout of
   * mq_quantize_tiled (1); this sequence owns rows out_row0 .. out_row0 + seq_real - 1 (seq_real <= seq: rows beyond it are
   * padding and are not written).  With out_i8 set, `out` may be NULL. */
  int8_t* out_i8;
  int32_t* out_rowsum;
  int64_t out_row0;
  int seq_real, out_shift, out_i8_tiled;
  /* alternative input: the uint8 output indices [seq, (heads + 2 kv_heads) * 64] of a fused q|k|v GEMM (mq_w8a8_linear_segmented)
   * with the three output grids; the prep kernel dequantises (index - offset) * scale -- the fp32 value the linear would have
   * written.  When qkv_idx is set, q / k / v are ignored (may be NULL). */
  const uint8_t* qkv_idx;
  mq_grid q_in, k_in, v_in;
  int rot_dim; /* partial rotary (hf_model.py:489-500): RoPE on the first rot_dim dims, cos / sin [seq, rot_dim]; 0 = head_dim */
  int32_t* v_prefix; /* head_dim 128 / 256: scratch [kv_heads][seq/64][head_dim] (running column sums of the stored v image) */
  /* cache continuation (chunked prefill; no counterpart in the reference, whose context encoding is one forward): with cache_seq > 0
   * k_i8 / vt_i8 / k_rowsum / v_prefix are caller-owned CACHES laid out for cache_seq rows ([kv_heads][cache_seq][D], ...) that already
   * hold positions 0 .. pos0 - 1 from earlier calls with the same grids; this call appends rows pos0 .. pos0 + seq - 1 and attends to
   * all of them.  q / k / v / cos / sin / out describe the chunk only (cos / sin rows of positions pos0 ...).  pos0 % 64 == 0,
   * cache_seq % 64 == 0, pos0 + seq <= cache_seq.  cache_seq = 0: the buffers are scratch of seq rows, pos0 = 0. */
  int pos0, cache_seq;
  /* head_dim 64 with a 16-bit score grid (the production configuration): optional fp16 images of the CENTRED indices (index - offset,
   * exact in fp16) -- q_f16 [heads][seq][64] scratch, k_f16 [kv_heads][cache_seq or seq][64] (a cache like k_i8).  With both set the
   * scores are contracted as v_mfma_f32_16x16x32_f16 (exact: integers below 2^24 in fp32), which yields sum (qi - zq)(ki - zk) as a
   * float directly -- no zero-point terms, no integer -> float conversion per score.  NULL: the int8 contraction. */
  uint16_t* q_f16;
  uint16_t* k_f16;
  /* batch > 1: that many sequences of `seq` (padded) rows in ONE launch pair -- q / k / v (or qkv_idx), out and every scratch buffer are
   * [batch][...] of the single-sequence shapes above, cos / sin are shared, sequence b owns rows out_row0 + b * seq_real ... of the
   * int8 image.  0 or 1: one sequence.  Not with cache continuation (cache_seq must be 0). */
  int batch;
} mq_attention_args;
int mq_attention_quant(const mq_attention_args* args, mq_stream_t stream);

/* ---- calibration: the score chain of an attention block with its two statistics ------------------ */
/* ptq/generate_act_range.py:55-69 hooks qk_bmm's output (raw scores) and pv_bmm's input (probabilities); the graph between them is
 * hf_model.py:513-530: att / sqrt(head_dim) [+ mask] -> softmax(dim = -1, fp32).  One pass per row: running [min, max] of the raw scores,
 * the probabilities written to `probs` (may alias `raw`), running [min, max] of the probabilities -- instead of two hook reductions and
 * ~five elementwise passes over [heads, S, S].  raw / probs [rows, cols] fp32, cols % 4 == 0, cols <= 4096 (else MQ_EUNSUPPORTED);
 * mask: NULL or additive fp32 [mask_rows, cols], row r uses mask row r % mask_rows; the four statistics are 1-float running values
 * as mq_minmax_tensor keeps them (initialise with mq_minmax_init; NaN is sticky). */
int mq_calib_attention_probs(const float* raw, float* probs, int64_t rows, int64_t cols, const float* mask, int64_t mask_rows, double sqrt_d,
                             float* raw_min, float* raw_max, float* probs_min, float* probs_max, mq_stream_t stream);

/* The same pass under the causal mask of a square block (rows = n * seq rows of seq columns; row r of a block masks the columns > r):
 * no mask is read, and with store_masked = 0 the quads wholly above the diagonal are not stored -- `probs` (!= raw) must already hold
 * zeros there, e.g. the buffer a previous call with the same shape wrote (a quarter of the pass's bytes).  Element for element the
 * arithmetic of mq_calib_attention_probs with the explicit -inf / 0 mask. */
int mq_calib_attention_probs_causal(const float* raw, float* probs, int64_t rows, int64_t seq, double sqrt_d, int store_masked, float* raw_min,
                                    float* raw_max, float* probs_min, float* probs_max, mq_stream_t stream);

/* ---- calibration: a decoder layer's glue with its statistics (round 6) --------------------------- */
/* The fp32 calibration forward (ptq/generate_act_range.py:49-122 over mobilellm/model/hf_model.py) hooks the input and the output of
 * every norm / activation / linear; between the linears it runs torch elementwise launches (an HFRMSNorm, hf_model.py:183-186, is
 * six of them) and every hooked tensor is read once more for its [min, max].  Two one-pass forms for the leaf graph of this package:
 *
 * mq_calib_norm: h = x (+ delta, the residual branch; h is written to h_out), running [min, max] of h (the norm's input hook),
 *   y = weight * (h * rsqrt(mean(h^2) + eps))  (layernorm = 0; hf_model.py:183-186)  or  LayerNorm(h) * weight + bias (layernorm = 1,
 *   biased variance), running [min, max] of y (its output hook).  x / delta / h_out / y_out [rows, cols] fp32, cols % 4 == 0,
 *   cols <= 8192, 16-byte aligned (else MQ_EUNSUPPORTED); delta, h_out, bias may be NULL (h_out is required with a delta);
 *   delta_min / delta_max (both or neither; NULL without a delta): running [min, max] of delta itself -- the output hook of the linear
 *   that produced the branch (o_proj, w2).
 * mq_calib_gated: out = act(a) * b (hf_model.py:1057: w2's input) with the running [min, max] of a (w1's output = the activation's
 *   input), act(a) (the activation's output), b (w3's output) and the product, in that order in stats[8] = {min, max} x 4.
 *   act 0 = SiLU, 1 = GELU (erf).  numel % 4 == 0.
 * Statistics are 1-float running values as mq_minmax_tensor keeps them (mq_minmax_init; NaN is sticky).  The row sums are taken in
 * another order than torch's: values agree with the module chain within a few ulp (tests bound the act_dict at 1e-5 relative). */
int mq_calib_norm(const float* x, const float* delta, float* h_out, float* y_out, int64_t rows, int64_t cols, const float* weight, const float* bias,
                  float eps, int layernorm, float* in_min, float* in_max, float* out_min, float* out_max, float* delta_min, float* delta_max,
                  mq_stream_t stream);
int mq_calib_gated(const float* a, const float* b, float* out, int64_t numel, int act, float* const* stats, mq_stream_t stream);
/* mq_calib_rope: rotary embedding of the q and k projections (hf_model.py:486-501, rotate-half; rot_dim < head_dim: partial rotary) with
 * the running [min, max] of q_proj's output, qk_bmm's input, k_proj's output and qk_bmm's input2 (stats[8] in that order): q_in
 * [batch, seq, heads * head_dim] and k_in [batch, seq, kv_heads * head_dim] as the linears wrote them -> q_out [batch, heads, seq,
 * head_dim], k_out [batch, kv_heads, seq, head_dim] contiguous.  cos / sin [seq, rot_dim] (the rows of the positions).  Two rounded
 * products and a rounded sum per element: the bits of the module chain. */
int mq_calib_rope(const float* q_in, const float* k_in, float* q_out, float* k_out, int64_t batch, int64_t seq, int heads, int kv_heads, int head_dim,
                  int rot_dim, const float* cos, const float* sin, float* const* stats, mq_stream_t stream);
/* mq_calib_rope_qkv: the same with v carried along (no rotation) and repeat_kv (hf_model.py:509-510) inside: k_out and v_out are
 * [batch, heads, seq, head_dim] with every kv head written heads / kv_heads times.  stats[12] = {min, max} of q_in, q_out, k_in, k_out,
 * v_in and (v_out: both NULL or both given -- v_out holds v_in's values).  */
int mq_calib_rope_qkv(const float* q_in, const float* k_in, const float* v_in, float* q_out, float* k_out, float* v_out, int64_t batch, int64_t seq, int heads,
                      int kv_heads, int head_dim, int rot_dim, const float* cos, const float* sin, float* const* stats, mq_stream_t stream);

/* ---- QMatMul as a module: quantized batched matmul of two activations ------------------------ */
/* Replaces QMatMul.forward (mobilellm/quantization/qmodule.py:453-466): out = Qout(matmul(Q1(x1), Q2(x2))) -- two fake-quant passes
 * per operand, an fp32 library bmm and two more passes over the product in the reference -- by ONE launch: both fp32 operands are
 * quantised on load (the exact index arithmetic of qmodule.py:286-287), contracted as int8 on the matrix pipe (a 9 ... 16-bit x1,
 * pv_bmm's probabilities, as two byte planes), corrected for the zero points in 64-bit integers, scaled once and passed through the
 * output quantizer (IEEE quotient).  x1 [batch, M, K] dense; x2 [batch, K, N] held either as [batch, N, K] (x2_k_contiguous = 1: what
 * hf_model.py:513's k.transpose(2, 3) view is in memory) or as [batch, K, N] (0: pv_bmm's v); out [batch, M, N] fp32, the
 * fake-quantised values the module returns.  grid1 / grid2: static per-tensor grids of at most 16 / 8 bits (integral offsets);
 * grid_out NULL or scale == NULL: no output quantizer.  Any M / N / K; with x2_k_contiguous == 0, N % 4 == 0 and a 16-byte aligned
 * x2 are required (MQ_EUNSUPPORTED / MQ_EINVAL otherwise).  16-byte aligned bases with K % 4 == 0 take the vector loads.  No mask or causality assumption (the fused causal attention is
 * mq_attention_quant).  Non-finite inputs saturate like every integer image (NaN -> qmin). */
int mq_qmatmul(const float* x1, const float* x2, float* out, int64_t batch, int64_t M, int64_t N, int64_t K, int x2_k_contiguous,
               const mq_grid* grid1, const mq_grid* grid2, const mq_grid* grid_out, mq_stream_t stream);

#ifdef __cplusplus
}
#endif
#endif /* MOBILEQUANT_AMD_H */

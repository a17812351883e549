/* mobilequant_amd_tuning.h -- tuning / profiling knobs of libmobilequant_amd.so.
 *
 * NOT part of the drop-in boundary (include/mobilequant_amd.h): nothing in the reference has a counterpart, and a
 * maintainer integrating the library never needs them.  They exist for tools/ (mq_probe, bench_shapes, A/B timing) and
 * for the parity tests that run one problem on every GEMM tile variant.  The settings are process-wide atomics: set them
 * from one thread while no GEMM call is in flight. */
#ifndef MOBILEQUANT_AMD_TUNING_H
#define MOBILEQUANT_AMD_TUNING_H
#ifdef __cplusplus
extern "C" {
#endif

/* Force a GEMM tile configuration (DESIGN.md "GEMM variants"); variant < 0 restores the built-in heuristic.
 * Returns the number of variants. */
int mq_gemm_set_variant(int variant);
const char* mq_gemm_variant_name(int variant);
/* Ablation switches of profiling builds (-DMQ_GEMM_ABLATE; results are WRONG when non-zero): 1 = no LDS-DMA after the
 * first stage, 2 = no MFMA loop, 4 = no epilogue, 16 = s_memtime stamps.  0 = normal operation; ignored by production
 * builds. */
int mq_gemm_set_debug(int flags);
/* Clock probe of the headline GEMM kernel (mq::gemm_i8_fr_kernel): with a device buffer of (workgroups x 8) x 8 bytes set, lane 0 of
 * every wave stores [uint32 shader-clock cycles (s_memtime), uint32 ticks of the constant 100-MHz counter (s_memrealtime)] spent inside
 * the generated program; sum(cycles) / sum(ticks) x 100 MHz is the clock the chip SUSTAINED under this launch (it clocks to its power
 * budget: random int8 operands ~1.75 GHz, zero-filled ~2.1 GHz of the nominal 2.4).  NULL (default) = no stores; the two counter reads
 * per wave stay in the kernel either way.  Pair launches share the buffer (second problem's workgroups write behind the first's). */
int mq_gemm_set_clock_probe(void* buf);
/* mq_w4a8_linear_tiled: 1 (default) = the packed pieces are expanded ONCE per workgroup into the int8 W ring (generated variants frw4x /
 * frw4x_128: the int8 kernel's loop), 0 = every wave splits the nibbles of its own fragments in registers (frw4 / frw4_128: one LDS read
 * per 16 columns and stage, 12 VALU per 4 MFMAs in every wave; measured 30 % slower).  Identical results.  Mode 0 is compiled into
 * experiment builds only (-DMQ_BUILD_EXPERIMENTS, `python -m mobilequant_amd.build --experiments`): the production library returns 1
 * (= not built) and stays in mode 1. */
int mq_gemm_set_w4_mode(int mode);
/* Tile order of the 128-column generated kernels (residual / segmented GEMMs): M-tiles per group of the grouped order each XCD walks
 * (0 = the built-in 4).  Traffic experiment of DESIGN.md 4.2.1 (L2 fetch bytes per XCD footprint); results do not depend on it. */
int mq_gemm_set_group_m(int group_m);
/* mq_w8a8_linear_tiled_pair: 0 (default) = one workgroup per tile and problem (2 x tiles workgroups), 1 = one workgroup per tile runs
 * problem 0 then problem 1 (persistent over the pair).  Identical results. */
int mq_gemm_set_pair_mode(int mode);
/* Tile height of mq_w8a8_linear_tiled_residual: 128 (four waves) / 256 (eight waves); 512 = 256-row tiles with the K loop split over two
 * workgroups that swap partial sums through a per-device scratch buffer (experimental, measured slower, one launch at a time per device;
 * falls back to the unsplit tile when both halves of every tile cannot be resident at once; experiment builds only -- the production
 * library returns 1 for 512 and keeps choosing by shape); anything else = by shape. */
int mq_gemm_set_residual_tile(int rows);
/* mq_w8a8_linear_tiled_segmented: 128 = always the 256 x 128 tile; anything else = 128 x 160 tiles where they fit one per CU. */
int mq_gemm_set_segmented_tile(int cols);
/* mq_quantize_tiled: 1 (default) = the LDS-staged eight-row kernel where it applies (fp32, 1024 <= cols <= 4096), 0 = the
 * lane-per-fragment kernel for every shape (A/B timing; identical images). */
int mq_quantize_tiled_set_staged(int on);
/* the staged kernel's rows per workgroup: 0 (default) = by shape (4 up to 2048 columns: two 512-thread workgroups per CU, 8 beyond),
 * 4 / 8 = forced (A/B timing; identical images). */
int mq_quantize_tiled_set_rows(int rows);
/* image-only tiled norm (mq_rmsnorm_quant / mq_layernorm_quant with only q_tiled): rows per workgroup -- 0 (default) = by shape (4 up to
 * 2048 columns, 8 beyond), 8 = 1024 threads, one workgroup per CU, 4 = 512 threads, two per CU whose load / arithmetic / store phases
 * overlap.  Identical images. */
int mq_norm_tiled_set_rows(int rows);
/* mq_attention_quant at head_dim 64: 1 = small exponential cache (two key blocks in the LDS, three waves per SIMD), anything else =
 * the deep cache (four blocks in the LDS + five in registers, two waves per SIMD; default).  Identical results. */
int mq_attention_set_cache(int mode);
/* mq_attention_quant at head_dim 64 (16-bit score grid, full rotary or rot_dim 16, deep cache): 1 (default) = every attention workgroup prepares its
 * own 64 query rows (RoPE + input quantizer in registers: no q image, the prep launch covers k / v only), 0 = the prep kernel writes the
 * q image as for the other shapes (A/B timing; identical results). */
int mq_attention_set_fused_q(int on);
/* 0 = int8 score contraction even when mq_attention_args carries the fp16 images (A/B timing); default 1 */
int mq_attention_set_f16(int on);
/* 1 = two heads of a KV group per eight-wave workgroup over one copy of the K / vT tiles (f16 form with in-kernel q rows, even heads per
 * KV group); default 0.  Identical results, measured slower: built only with -DMQ_BUILD_EXPERIMENTS, otherwise the call returns 1. */
int mq_attention_set_pair(int on);

#ifdef __cplusplus
}
#endif
#endif /* MOBILEQUANT_AMD_TUNING_H */

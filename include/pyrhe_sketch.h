/*
 * pyrhe_sketch.h -- C ABI of the MI355X (gfx950) streaming randomized-trace sketch engine.
 *
 * Drop-in boundary for ONE path of sriramlab/PyRHE: the per-jackknife-block sketch of
 * StreamingBase (2-bit PLINK .bed decode -> per-SNP standardize -> X^T R and X (X^T R) per
 * annotation bin).  The reference has no FFI; the Python seams these entry points replace are
 * cited per function (paths relative to the reference repository root).
 *
 * Conventions
 *   - every function returns 0 on success, a negative PYRHE_E* code or a positive hipError_t
 *     otherwise; pyrhe_last_error() returns a thread-local message.  No exceptions cross the ABI.
 *   - the caller owns every buffer; "device" pointers are HIP device memory of the engine's device.
 *   - kernels are enqueued on the caller's stream (void* = hipStream_t) and never synchronise,
 *     except where a function is documented as synchronous.
 *   - one engine per GPU, driven from one host thread.
 *
 * Device data layout
 *   packed genotypes : uint8 [n_rows][pitch], PLINK1 SNP-major rows (4 samples/byte, LSB first,
 *                      code 00 -> 0, 10 -> 1, 11 -> 2 copies of allele 2, 01 -> missing), each row padded
 *                      with zero bytes to `pitch` (a multiple of 256).  n_pad = 4 * pitch samples.
 *   right-hand sides : float [n_cols][n_pad], one contiguous row per column (probe vectors Z, U Z,
 *                      phenotype, e (.) Z ...); entries of dropped / padding samples must be 0.
 *   sketch output    : float [nbin][n_sketch][n_pad].
 */
#ifndef PYRHE_SKETCH_H
#define PYRHE_SKETCH_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define PYRHE_ABI_VERSION 1

/* error codes (negative; positive values are hipError_t) */
#define PYRHE_OK 0
#define PYRHE_EINVAL (-1)   /* bad argument (message says which) */
#define PYRHE_ENOMEM (-2)   /* workspace too small / allocation failed */
#define PYRHE_ESTATE (-3)   /* call order (e.g. sketch before set_rhs) */

/* precision modes: how the float32 operands R and W = D V are fed to the matrix cores */
#define PYRHE_PREC_F16 0    /* one fp16 rounding (11 significant bits), 1 MFMA pass            */
#define PYRHE_PREC_F16X2 1  /* hi + lo fp16 split (~22 significant bits), 2 MFMA passes        */
#define PYRHE_PREC_F16_EXACT 2 /* one fp16 rounding, genotypes unpacked to normal fp16 numbers (3 VALU ops per
                                  register pair instead of the 1-op subnormal unpack of PYRHE_PREC_F16)        */

#define PYRHE_PREC_I8X3 3   /* both contractions on the int8 matrix cores: every operand column as three balanced radix-256
                               int8 digits (all 24 bits of the float32), exact int32 accumulation, 2x K per MFMA, the
                               (digit, column) pairs packed densely into 16-row tiles.  GEMM 2 takes this form for up to
                               20 sketch columns per call of several lists (or one list of a large cohort); otherwise it
                               runs as PYRHE_PREC_F16X2.  The parity-safe default.                                      */

#define PYRHE_PREC_I8X3_F16 4 /* GEMM 1 as I8X3, GEMM 2 with ONE fp16 rounding of W (as PYRHE_PREC_F16).  Measured vs I8X3
                                 (M = 200k, jn = 100): sigma^2 3e-5 / 6e-6 / 2e-6 / 3e-7 and SE 5e-5 / 1.3e-5 / 9e-6 / 9e-6
                                 at N = 5k / 20k / 100k / 500k -- the SE error does not shrink with N (it is set by the
                                 SNPs per block), so this mode is NOT parity-safe; since the int8 GEMM 2 it is not faster
                                 than I8X3 either (kept for the precision experiments of DESIGN.md 3.1)                  */

/* column roles for pyrhe_engine_set_rhs */
#define PYRHE_COL_SKETCH 1  /* gets both contractions: out[:, col] = X (X^T r)   (a7/a9 in SURVEY 8a) */
#define PYRHE_COL_NORM 2    /* gets sum_s (x_s^T r)^2                            (a8)                 */

#define PYRHE_COL_CROSS 3   /* gets the first contraction only; the engine returns, per bin, the Gram matrix of these
                               columns' projections  sum_s (x_s^T r_i)(x_s^T r_j) = R_x^T X X^T R_x
                               (pyrhe_engine_sketch_block_cross).  With the covariates W as cross columns that is all the
                               normal equations need of XXUz = X X^T (U Z) (a9): it only enters through sum(XXUz * UXXz)
                               (base.py:589-591), UXXz lies in span(W), and W^T X X^T (U Z) = (W^T X X^T W)(Q W^T Z).     */

/* role[c] | PYRHE_COL_DIGITS(d), d = 1, 2 (3 = default): the int8 modes feed this NORM / CROSS column to the first
   contraction as its d most significant balanced radix-256 digits (8 d bits of its largest entry) instead of all three --
   fewer packed digit rows, fewer 16-row tiles of matrix work and operand traffic.  Digit counts must not grow with the column
   index; sketch columns always carry three.  For covariates as cross columns (their Gram matrix only enters the O(c / N)
   terms of the normal equations) one digit moves sigma^2 by 2e-7 and the jackknife SE by 2e-6 at N = 2 000 and by 1 / N^2
   beyond, two digits by 5e-8 at N = 500.  Ignored by the fp16 modes. */
#define PYRHE_COL_DIGITS(d) ((d) << 8)

#define PYRHE_MAX_COLS 46      /* sketch + norm + cross columns of the int8 precisions (a ones column is appended internally), as long as
                                  their packed digit rows fit six 16-row tiles: columns + 1 + those with >= 2 digits + those with 3 <= 96
                                  (20 sketch + 2 norm + 20 one-digit cross columns = 87: GENIE G x E with 10 covariates in ONE pass);
                                  at most 32 of them sketch columns */
#define PYRHE_MAX_COLS_FP16 30 /* the fp16 precisions: two 16-column operand tiles incl. the ones column */

typedef struct pyrhe_engine pyrhe_engine_t;

/* message of the last failing call on this thread */
const char* pyrhe_last_error(void);
int pyrhe_abi_version(void);

/*
 * Create an engine on `device` for cohorts of `n_keep` retained samples stored with row pitch
 * `pitch` bytes.  `max_entries` bounds the SNPs of one (block, bin) list; `sub_snps` is the number of
 * entries one kernel triple (GEMM 1, mid, GEMM 2) takes (0 = 16384: whole blocks per launch -- measured,
 * re-reading a just-streamed sub-block from the Infinity Cache gains less than short launches lose).
 * On failure nothing stays allocated.
 * Replaces: Base._init_device / to_tensor device placement (pyrhe/src/base/base.py:195-206,
 * pyrhe/src/util/mat_mul.py:4-15).   Synchronous (allocates).
 */
int pyrhe_engine_create(int device, int64_t n_keep, int64_t pitch, int64_t max_entries, int precision,
                        int64_t sub_snps, pyrhe_engine_t** out);
int pyrhe_engine_destroy(pyrhe_engine_t* e);

/*
 * Install the right-hand columns.  R: device float [n_cols][n_pad]; role[c] in {SKETCH, NORM, CROSS}, in that order;
 * keep: device float [n_pad] with 1 for retained samples and 0 for dropped / padding samples
 * (it becomes the internal "ones" column that yields the exact integer genotype sums).
 * Sketch columns must come first.  Builds fp16 MFMA-operand images, per-column power-of-two scales
 * and column sums.   Replaces: to_tensor(all_zb / all_Uzb / pheno) in Base._compute_XXz/_XXUz/_yXXy
 * (pyrhe/src/base/base.py:403-417; mat_mul.py:12 float32 cast).   Asynchronous on `stream`.
 */
int pyrhe_engine_set_rhs(pyrhe_engine_t* e, const float* R, const int32_t* role_host, int n_cols,
                         const float* keep, void* stream);

/*
 * Sketch one jackknife block.
 *   packed      device uint8 [n_rows][pitch]  -- the block's SNP rows
 *   entries     device int32 [offsets[nbin]]  -- row indices (0..n_rows-1) per bin, concatenated
 *   offsets     HOST   int64 [nbin+1]
 *   fill        device float [n_rows] or NULL -- value (0/1/2) substituted for missing calls of a SNP
 *                                                (NULL: the block has no missing calls)
 *   out_sketch  device float [nbin][n_sketch][n_pad]; accumulate != 0 adds to it, 0 overwrites
 *   out_norm    device double [nbin][n_norm]        ; same accumulate flag
 *   out_colsum  device int32 [offsets[nbin]] or NULL -- exact genotype sum per entry (after filling)
 * Per bin b with entry list L_b and X_b the standardized genotypes of those SNPs
 * (mu = mean, 1/sd = 1/sqrt(mu (1 - mu/2)), float32, pyrhe/src/base/base.py:291-296):
 *   out_sketch[b][c][:] (+)= X_b (X_b^T R[c])      for the sketch columns
 *   out_norm[b][c]      (+)= || X_b^T R[c] ||^2    for the norm columns
 * Replaces, for one block: open_bed.read + 0<->2 flip (base.py:338-356), partition_bins (:325-336),
 * standardize_geno (:291-296) and the k GEMV pairs of _compute_XXz/_compute_XXUz/_compute_yXXy
 * (:403-417) as called from StreamingRHE.pre_compute_jackknife_bin(_pass_2)
 * (pyrhe/src/models/rhe/streaming_rhe.py:13-43).   Asynchronous on `stream`.
 * Limits: n_rows * pitch < 2^32 and n_rows < 2^24 (32-bit row byte offsets; PYRHE_EINVAL beyond, "split it"), unless the
 * caller set PYRHE_OPT_LIST_SPAN = v -- then only every window of v consecutive lists (one jackknife block's bins) must stay
 * within that span and n_rows < 2^31 (several blocks of a large cohort in one call: the entry lists of consecutive
 * jackknife blocks are just more bins of the call).
 * A list of more than 15 104 entries is taken in pieces of that size, each piece its own launch sequence (the second contraction
 * keeps a piece's rows in LDS, and two pieces of a list may not share a launch: both add into the same output slab).  A caller that
 * wants ONE launch over a long block hands its rows over as several shorter lists with output slabs of their own and adds those
 * (pyrhe_amd/streaming.py: SketchRunner._windows; N = 500k, blocks of 16 667 ... 100 000 SNPs: 51 - 56 -> 46 - 49 ms per pass).
 */
int pyrhe_engine_sketch_block(pyrhe_engine_t* e, const uint8_t* packed, int64_t n_rows,
                              const int32_t* entries, const int64_t* offsets_host, int nbin,
                              const float* fill, float* out_sketch, double* out_norm,
                              int32_t* out_colsum, int accumulate, void* stream);

/*
 * As pyrhe_engine_sketch_block, plus the DOMINANCE sketches of RHE_DOM for the same entry lists
 * (pyrhe/src/models/rhe_dom/rhe_dom.py:15-41,56-68 and streaming_rhe_dom.py): with p = mean/2, the encoding
 * enc(0,1,2) = (0, 2p, 4p-2) and X_dom = (enc - mean(enc)) / (2p(1-p)),
 *   out_sketch_dom[b][c][:] (+)= X_dom (X_dom^T R[c]),   out_norm_dom[b][c] (+)= || X_dom^T R[c] ||^2.
 * Costs one extra GEMM 1 (view 2*[g==2]) and two extra GEMM 2 launches per entry list.   Asynchronous on `stream`.
 */
int pyrhe_engine_sketch_block_dom(pyrhe_engine_t* e, const uint8_t* packed, int64_t n_rows,
                                  const int32_t* entries, const int64_t* offsets_host, int nbin,
                                  const float* fill, float* out_sketch, double* out_norm,
                                  float* out_sketch_dom, double* out_norm_dom, int accumulate, void* stream);

/*
 * As pyrhe_engine_sketch_block / _dom for an engine whose right-hand columns include PYRHE_COL_CROSS columns (they come
 * last: sketch, norm, cross).  Additional outputs, same accumulate flag:
 *   out_cross      device double [nbin][n_cross][n_cross] (+)= sum_{s in L_b} (x_s^T R_i)(x_s^T R_j) over the cross columns
 *                  i, j (symmetric; the diagonal is what a norm column would give); ordered sums, bit-reproducible
 *   out_cross_dom  the same under the dominance encoding; NULL unless out_sketch_dom / out_norm_dom are given
 * out_sketch_dom, out_norm_dom, out_cross_dom all NULL: additive sketches only.  An engine may hold cross columns alone
 * (no sketch columns: the call is one GEMM 1 + the small reductions, no second read of the genotypes).
 * Replaces Base._compute_XXUz (pyrhe/src/base/base.py:410-412) as consumed by setup_lhs_rhs_jackknife (base.py:583-593):
 * with R_x = W (the covariates), W^T XXUz_b = out_cross (Q W^T z_b).   Asynchronous on `stream`.
 */
int pyrhe_engine_sketch_block_cross(pyrhe_engine_t* e, const uint8_t* packed, int64_t n_rows,
                                    const int32_t* entries, const int64_t* offsets_host, int nbin,
                                    const float* fill, float* out_sketch, double* out_norm, double* out_cross,
                                    float* out_sketch_dom, double* out_norm_dom, double* out_cross_dom,
                                    int accumulate, void* stream);

/*
 * Pipelining across calls.  Internally the engine enqueues on private streams with two workspace sets (optionally
 * GEMM 1 of the next sub-block beside GEMM 2 of the current one, pyrhe_engine_set_pipeline).  By default
 * every pyrhe_engine_sketch_block call ends by making the caller's stream wait for its results (plain stream
 * semantics).  pyrhe_engine_set_async(e, 1) defers that wait: results of all calls since the last join are ordered
 * into `stream` by pyrhe_engine_join(e, stream) -- the overlap then also spans consecutive blocks.  Output buffers of
 * un-joined calls must be distinct, inputs must stay unchanged until the join.
 */
int pyrhe_engine_set_async(pyrhe_engine_t* e, int enable);
/* enable = 1 puts GEMM 2 on a second internal stream so it overlaps GEMM 1 of the next sub-block (+3.5 % measured at
 * N=500k; per-kernel timings then include the contention).  Default 0: both on one internal stream, in order. */
int pyrhe_engine_set_pipeline(pyrhe_engine_t* e, int enable);
int pyrhe_engine_join(pyrhe_engine_t* e, void* stream);

/*
 * Per-SNP statistics of a block, for the host-side imputation replay (Base.impute_geno,
 * pyrhe/src/base/base.py:277-289): sum of observed genotypes and number of missing calls among the
 * retained samples.  out_sum / out_nmiss: device int32 [n_rows].   Asynchronous on `stream`.
 */
int pyrhe_engine_block_stats(pyrhe_engine_t* e, const uint8_t* packed, int64_t n_rows,
                             int32_t* out_sum, int32_t* out_nmiss, void* stream);

/*
 * out[i] = sum_{j < n_blocks} (double) blocks[j * block_stride + i], i < n: the float64 aggregate over per-block float32
 * sketches (StreamingBase.aggregate, pyrhe/src/base/base_streaming.py:21-32; Base.aggregate, base.py:468-471) as one
 * streaming pass.  blocks / out: device; n and block_stride multiples of 4.   Asynchronous on `stream`.
 */
int pyrhe_sum_blocks(const float* blocks, int n_blocks, int64_t block_stride, int64_t n, double* out, void* stream);
/* out[i] = sum_j weights[j] * blocks[j * block_stride + i] (weights: device float64 [n_blocks]).  With weights jn - j it is
 * the whole-data UXXz argument of the reference's streaming pass 1, which applies U to the RUNNING sum of XXz
 * (pyrhe/src/models/rhe/streaming_rhe.py:18-21): sum_j U(sum_{i<=j} XXz_i) = U(sum_i (jn - i) XXz_i).   Asynchronous. */
int pyrhe_sum_blocks_weighted(const float* blocks, int n_blocks, int64_t block_stride, int64_t n,
                              const double* weights, double* out, void* stream);
/* Both of the above from ONE read of the blocks: out = the plain sum, out_weighted = the weighted one (the same additions in
 * the same order as the two calls: same bits).  The reference-compatible streaming mode with covariates needs both, and the
 * per-block store is 16 GB at 8 bins and N = 500k.   Asynchronous. */
int pyrhe_sum_blocks_both(const float* blocks, int n_blocks, int64_t block_stride, int64_t n,
                          const double* weights, double* out, double* out_weighted, void* stream);

/*
 * Leave-one-out inner products for the normal equations: the Gram loops of Base.setup_lhs_rhs_jackknife
 * (pyrhe/src/base/base.py:568-628: <XXz_a, XXz_b>, <XXUz_a, UXXz_b>, <XXz_a, U z>) and GENIE.b_trace_calculation
 * (pyrhe/src/models/genie/genie.py:84-94: <XXz_a, z>), on the device, for a whole batch of jackknife blocks per call.
 * A row is described, not materialised:
 *     v[idx] = tot[idx] - blk[z * blk_batch_stride + idx] * mul[idx % period],     idx < n_sub * period
 * tot (device float64, the whole-data accumulator), blk (device float32, the stored sketch of batch element z; the
 * product with mul is rounded in float32 like the reference's elem_mul, streaming_genie.py:39-41) and mul (device
 * float32 [period]) may each be NULL (tot or blk must be given).  The arrays x / y themselves are HOST arrays.
 *     out[z][a][b] = sum_idx x_a[idx] * y_b[idx]        float64, device, [n_batch][nx][ny]
 * Sums are deterministic (fixed order).  period must be even, tot 16-byte and blk/mul 8-byte aligned.
 * Asynchronous on `stream`; the engine owns the scratch, so calls on one engine must be on one stream.
 */
typedef struct pyrhe_row {
    const double* tot;
    const float* blk;
    const float* mul;
    int64_t blk_batch_stride; /* elements between the blk rows of consecutive batch elements */
} pyrhe_row_t;
int pyrhe_engine_loo_dots(pyrhe_engine_t* e, const pyrhe_row_t* x, int nx, const pyrhe_row_t* y, int ny,
                          int64_t period, int n_sub, int n_batch, double* out, void* stream);

/*
 * One read of the stored block rows for all of their pass-2 products (the blk parts of the same Gram loops, base.py:568-628,
 * when covariates are present: base.py:583-593).  x: 1..16 block-only rows (tot NULL; blk 16-byte aligned), evaluated as
 * -blk * mul like in pyrhe_engine_loo_dots; y_sub: 1..16 tot-only rows of n_sub * period elements; y_flat: 0..16 tot-only
 * rows of `period` elements (the covariate columns W_m).
 *     out_sub [z][a][b]      = sum_{s, i} x_a[z][s * period + i] * y_sub_b[s * period + i]
 *     out_flat[z][a][s][m]   = sum_i      x_a[z][s * period + i] * y_flat_m[i]                (NULL with ny_flat = 0)
 *     out_self[z][a][b]      = sum_{s, i} blk_a * blk_b  (signs cancel)                       (NULL with want_self = 0)
 * float64, device; deterministic; period a multiple of 32.  The same numbers as three pyrhe_engine_loo_dots calls (block rows
 * x sub-strided rows, their nx * n_sub sub-rows x flat rows with n_sub = 1, block rows x themselves) for one pass over the
 * store instead of three.  Asynchronous on `stream`; one stream per engine (shared scratch).
 */
int pyrhe_engine_loo_dots_multi(pyrhe_engine_t* e, const pyrhe_row_t* x, int nx, const pyrhe_row_t* y_sub, int ny_sub,
                                const pyrhe_row_t* y_flat, int ny_flat, int want_self, int64_t period, int n_sub,
                                int n_batch, double* out_sub, double* out_flat, double* out_self, void* stream);

/*
 * out[r][:] = W (Q (W^T x[r][:])) for n_rows rows of n elements: the covariate projector U = W Q W^T of
 * Base._compute_UXXz (pyrhe/src/base/base.py:407-408), applied to the whole-data XXz rows in pass 1
 * (pyrhe/src/models/rhe/streaming_rhe.py:18-21).  Float32 semantics like the reference's mat_mul chain (operands cast to
 * float32, each of the three products rounded to float32 once; inner sums in float64, fixed order: deterministic).
 * x / out: device float64 [n_rows][n] (may not alias); Wt: device float [c][n] = W transposed, zero at dropped and padding
 * samples; Q: device float [c][c] row-major = pinv(W^T W); 1 <= c <= 64.   Asynchronous on `stream`; uses the engine's
 * scratch (one stream per engine, as pyrhe_engine_loo_dots).
 */
int pyrhe_engine_apply_projector(pyrhe_engine_t* e, const double* x, int64_t n_rows, int64_t n, const float* Wt,
                                 const float* Q, int c, double* out, void* stream);

/*
 * Genotype coding of the packed buffers.  PLINK (default): the .bed body as it is.  VALUE: every 2-bit field holds the
 * allele-2 count itself (00->0, 10->1, 11->2) and 3 for a missing call: the kernels then skip the three-instruction
 * code -> count conversion they otherwise pay per dword in both contractions of both passes (-8 % GEMM 2 time).
 * pyrhe_recode_rows converts a buffer IN PLACE (either direction, one streaming read + write: the decode contract of
 * base.py:338-356 is unchanged, only where it is paid); pyrhe_engine_set_coding tells an engine which coding the
 * `packed` arguments of its sketch / block_stats calls use.   Asynchronous on `stream` / immediate.
 */
#define PYRHE_CODING_PLINK 0
#define PYRHE_CODING_VALUE 1
int pyrhe_recode_rows(uint8_t* packed, int64_t n_rows, int64_t pitch, int to_coding, void* stream);
int pyrhe_engine_set_coding(pyrhe_engine_t* e, int coding);

/*
 * Decode PLINK-coded rows to dense float32 [n_rows][n_pad] (count of allele 2, NaN = missing): the bit-exact
 * decode contract of bed_reader.read + the reference's flip (base.py:338-356).  Test/diagnostic entry
 * point; independent of any engine.   Asynchronous on `stream`.
 */
int pyrhe_decode_rows(const uint8_t* packed, int64_t n_rows, int64_t pitch, float* out, void* stream);

/*
 * Fill `packed` [n_rows][pitch] with synthetic genotypes: SNP (row0 + r) has allele frequency
 * p = (13 + h(seed, row) % 116) / 256 and each sample two Bernoulli(p) draws from a counter-based
 * hash keyed (seed, row, sample) -- regenerable row by row on the CPU (oracle/synth.py).
 * Samples >= n_samples are padding (code 00).   Asynchronous on `stream`.
 */
int pyrhe_synth_genotypes(uint8_t* packed, int64_t n_rows, int64_t pitch, int64_t n_samples,
                          uint64_t seed, int64_t row0, void* stream);

/*
 * Plain streaming read of `bytes` bytes of device memory (16-byte loads, every byte read once; the xor of the data may be
 * added to *sink, a device word): the box's measured read-stream bandwidth that bench.py reports beside the nominal HBM peak
 * (no counterpart in the reference; measurement aid).   Asynchronous on `stream`.
 */
int pyrhe_stream_read(const void* buf, int64_t bytes, uint32_t* sink, void* stream);

/* engine introspection: 0 n_pad, 1 sub_snps, 2 n_tiles_in, 3 n_sketch, 4 n_norm, 5 workspace bytes, 10 n_cross,
 * 11 packed (digit, column) rows of the int8 GEMM 1's operand;
 * which kernels the LAST sketch call launched (so a parity test can assert the path it checked is the path it meant):
 *   6 GEMM 1: 100 + tiles for gemm1_i8_kernel<tiles>, 200 + tiles for gemm1_i8w_kernel<tiles>, tiles for gemm1_kernel<tiles>
 *             (fp16), 0 before any call
 *   7 GEMM 2: 11 / 12 gemm2_i8p_kernel<NG = 1 / 2>, 13 gemm2_i8p_kernel<2, GV = 2> (additive + dominance in one launch), 1 gemm2_kernel (fp16 hi+lo or single), 0 none (no sketch columns)
 *   8 flow  : 1 segmented (one GEMM 1 over the concatenated lists, GEMM 2 by segment), 0 one kernel triple per list
 *   9 merge : segments per workgroup of the last gemm2_i8p_kernel launch (> 1: short lists shared workgroups, PYRHE_OPT_G2_MERGE) */
int64_t pyrhe_engine_info(const pyrhe_engine_t* e, int what);

/*
 * Launch-shape knobs (explicit setters; the library reads no environment variables).  Every value computes the same
 * results; they exist for the measurements of DESIGN.md 3.4 and for tests that must pin a kernel path.
 */
#define PYRHE_OPT_G1_WORKGROUPS 1 /* target workgroups per GEMM 1 launch (sample-range splits follow); default 1536 */
#define PYRHE_OPT_G2_PACKED 2     /* 0: never take the limb-packed int8 GEMM 2 (fp16 hi+lo instead); default 1    */
#define PYRHE_OPT_G2_IW 3         /* fp16 GEMM 2: dwords per lane per row (1 / 2); 0 = automatic                    */
#define PYRHE_OPT_G2_SPLITS 4     /* fp16 GEMM 2, per-list flow: K splits over workgroups; 0 = automatic           */
#define PYRHE_OPT_DOTS_GENERIC 5  /* 1: pyrhe_engine_loo_dots always takes the register-tiled generic kernel        */
#define PYRHE_OPT_G1_SPLIT_MAJOR 6 /* GEMM 1 grid order.  0: row group fast; 1: sample split fast (co-resident workgroups
                                      share rows); 2 (int8 kernels): every XCD works through its own run of sample ranges, whose
                                      slice of the operand image then stays in that XCD's L2                         */
#define PYRHE_OPT_G1_WIDE 7        /* int8 GEMM 1 with 128 rows per wave (gemm1_i8w_kernel: half the R-image traffic): 0 never,
                                      1 (default) up to three tiles and >= 196 608 samples or enough rows per call to fill the chip, 2 whenever the registers allow (<= 4)  */
#define PYRHE_OPT_G2_MERGE 8       /* packed int8 GEMM 2: consecutive short segments (bins of a block) share a workgroup while the
                                      launch keeps this many workgroups; 0 = one segment per workgroup; default 1536             */
#define PYRHE_OPT_G1_ZERO_ROWS 9   /* int8 GEMM 1: lanes of the last tile's unused digit rows fetch one shared zero
                                      entry of the R image instead of their own zeros; default 1                               */
#define PYRHE_OPT_LIST_SPAN 10     /* v >= 1: the caller vouches that the rows of any v consecutive entry lists starting at a multiple
                                      of v (v = bins per jackknife block: the lists of ONE block) together span < 2^32 bytes and
                                      < 2^24 rows; a call may then address more than 2^32 bytes of rows (several blocks of a large
                                      cohort in one launch) on the int8 path with <= 20 sketch columns (other paths still refuse).
                                      The packed GEMM 2 addresses the lists that share a workgroup relative to their lowest row and
                                      in such a call only lets lists of one such window share one.  Default 0 (refuse)            */
#define PYRHE_OPT_G1_FILL 11       /* int8 GEMM 1: among the sample-split counts near the G1_WORKGROUPS target take the one whose
                                      workgroups fill whole rounds of the chip's resident slots best; 0 = the plain target; default 1 */
#define PYRHE_OPT_DOM_SEG 12       /* pyrhe_engine_sketch_block_dom / _cross with dominance outputs: 1 (default) all lists of a call share
                                      one GEMM 1 launch (both genotype views) and three GEMM 2 launches (the segmented flow); 0 = one
                                      kernel sequence per list (round 2's flow)                                                   */
#define PYRHE_OPT_DOTS_F32 13      /* pyrhe_engine_loo_dots_multi: 1 = the products on the float32 matrix instruction with float32 partial
                                      sums over 128 elements, added up in float64 (9.3 -> 7.3 ms for the 16 GB store of 8 bins at
                                      N = 500k; sigma^2 unchanged to 16 digits, jackknife SEs 5e-9); 0 = float64 products and sums
                                      throughout; 2 (default) = 1 for rows of >= 65 536 elements, 0 below                        */
#define PYRHE_OPT_DOTS_TILES 14    /* pyrhe_engine_loo_dots_multi: 16-row tiles of block rows per wave: 4 (a wave owns the 64 rows of its
                                      row group; one wave per SIMD), 2 (default: wave pairs share a chunk, two waves per SIMD) or 1
                                      (four waves per SIMD).  Pass 2 at 8 bins + covariates, N = 500k: 9.3 / 8.2 / 12.7 ms        */
#define PYRHE_OPT_G1_LINE 15       /* int8 GEMM 1 with more than three tiles (256-sample steps): 1 (default) a register stage holds both
                                      halves of a 128-byte line, loaded together; 0 = 64 bytes per row and step                  */
#define PYRHE_OPT_DOM_FUSE 16      /* dominance outputs in the segmented flow, <= 10 sketch columns: 1 (default) ONE second contraction
                                      for both estimates (one read of the genotypes: additive and dominance operand tiles side by
                                      side, the 2 [g == 2] view from the same unpacked bytes); 0 = three launches                  */
#define PYRHE_OPT_G1_COMPACT 17    /* 128-row int8 GEMM 1 whose last operand tile has at most two live digit rows (10 probes + 1 phenotype:
                                      34 rows): 1 (default) those rows are staged from a compact copy of the tile, one LDS-DMA per
                                      step instead of eight; 0 = the tile is staged like the others                               */
int pyrhe_engine_set_option(pyrhe_engine_t* e, int option, int64_t value);

/* name + average device time (ms, from hipEvents on the launch stream) of the kernels of the last
 * pyrhe_engine_sketch_block call made while profiling was enabled; see pyrhe_engine_profile(). */
int pyrhe_engine_profile(pyrhe_engine_t* e, int enable);
int pyrhe_engine_profile_read(pyrhe_engine_t* e, double* ms_gemm1, double* ms_mid, double* ms_gemm2,
                              int64_t* launches);

#ifdef __cplusplus
}
#endif
#endif /* PYRHE_SKETCH_H */

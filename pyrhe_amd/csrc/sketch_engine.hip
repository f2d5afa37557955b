// sketch_engine.hip -- MI355X (gfx950, CDNA4) streaming randomized-trace sketch engine.
//
// One jackknife block of PyRHE's StreamingBase hot loop, computed straight from the packed 2-bit
// PLINK stream (reference: pyrhe/src/base/base.py:291-296,338-356,403-417 and
// pyrhe/src/models/rhe/streaming_rhe.py:13-43; the C ABI is declared in include/pyrhe_sketch.h).
//
// The standardized matrix X is never materialised.  With G the raw allele counts (0/1/2),
// mu_s = mean, d_s = 1/sqrt(mu_s (1 - mu_s/2)):
//     X^T R = D (G^T R - mu (1^T R))          GEMM 1  (reduce over samples)   + rank-1 correction
//     X  V  = G (D V) - 1 (mu^T D V)          GEMM 2  (reduce over SNPs)      + rank-1 correction
// G is decoded in registers (0/1/2 are exact in int8 and fp16).  Default precision (i8x3): both contractions on
// v_mfma_i32_16x16x64_i8 with R and W = D V as three balanced radix-256 int8 digits per column and exact int32
// accumulation (gemm1_i8_kernel, gemm2_i8p_kernel: <= 20 sketch columns, segmented flow); the fp16 kernels
// (v_mfma_f32_16x16x32_f16, hi+lo pair or one rounding) serve f16x2 / f16 and the shapes the packed GEMM 2 does not take.
//
// Files: this one = engine state, launch logic, C ABI; sketch_common.h = decode / unpack helpers + segment table;
// kernels_prep.inc, kernels_gemm1.inc, kernels_mid.inc, kernels_gemm2.inc, kernels_util.inc, kernels_dots.inc = kernels.
// Kernels (a call = the lists of one or several jackknife blocks; all lists share ONE GEMM 1, see SegTab):
//   gemm1(_i8)_kernel : P[e][c] = sum_i G[i, row(e)] * R[i][c]      (+ exact column sums via a ones column)
//   mid1_kernel  : mu, d, V = d (P/s - mu csum), norm columns -> sum V^2, W = d V (fp32) + column max, per segment
//   mid2_kernel  : W -> fp16 MFMA operand image (per-column power-of-two scale), corr = mu^T Wh, per segment
//   gemm2_kernel : out[c][i] (+)= keep[i] * (sum_e G[i, row(e)] * Wh[e][c] - corr[c]) / s_c   (blockIdx.y = segment)
//   sum_blocks / loo_dots / blk_fixed_dots / blk_self_gram : aggregate and the leave-one-out normal equations
//   recode_rows  : PLINK code -> 2-bit value code, once per cohort (template flag VC in the GEMM kernels)
//
// MFMA operand maps used (guide: cdna_hip_programming.md section 3):
//   A: lane l holds A[row l&15][k = 8 (l>>4) + j], j = 0..7;  B: lane l holds B[k = 8 (l>>4) + j][col l&15];
//   D: lane l holds D[row 4 (l>>4) + r][col l&15], r = 0..3.
// The k <-> sample/SNP assignment inside a fragment is a free permutation as long as both operands
// agree, which is what makes a 1-instruction-per-register-pair unpack possible (see unpack()).

#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include <string.h>
#include <math.h>
#include <stdarg.h>
#include <stdlib.h>
#include <algorithm>
#include <vector>
#include <type_traits>

#include "../../include/pyrhe_sketch.h"
#include "sketch_common.h"

// ------------------------------------------------------------------------------------------------
// error plumbing
// ------------------------------------------------------------------------------------------------
static thread_local char g_err[512] = "";

static int fail(int code, const char* fmt, ...) {
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(g_err, sizeof(g_err), fmt, ap);
    va_end(ap);
    return code;
}
#define HIPCHK(expr)                                                                              \
    do {                                                                                          \
        hipError_t _e = (expr);                                                                   \
        if (_e != hipSuccess)                                                                     \
            return fail((int)_e, "%s failed: %s (%s:%d)", #expr, hipGetErrorString(_e), __FILE__, \
                        __LINE__);                                                                \
    } while (0)

extern "C" const char* pyrhe_last_error(void) { return g_err; }
extern "C" int pyrhe_abi_version(void) { return PYRHE_ABI_VERSION; }

// ---- kernels (one translation unit; see the files for the design notes of each) ----------------
#include "kernels_prep.inc"
#include "kernels_gemm1.inc"
#include "kernels_mid.inc"
#include "kernels_gemm2.inc"
#include "kernels_util.inc"
#include "kernels_dots.inc"

// ------------------------------------------------------------------------------------------------
// engine
// ------------------------------------------------------------------------------------------------
struct pyrhe_engine {
    int device = 0;
    int64_t n_keep = 0, pitch = 0, n_pad = 0, max_entries = 0, sub_snps = 0;
    int precision = 0, parts = 1;
    bool exact = false;   // exact (normal-number) genotype unpack instead of the 1-op subnormal unpack
    bool i8 = false;      // int8 digit kernels (precision i8x3: GEMM 1 always, GEMM 2 where the packed kernel applies)
    bool vc = false;      // the caller's packed buffers hold the 2-bit VALUE code (pyrhe_recode_rows), not PLINK codes
    int n_sketch = 0, n_norm = 0, n_cross = 0, n_user = 0, ntl = 0, nto = 0, nt8 = 0;
    int nd2 = 0, nd3 = 0, nslots = 0;   // int8 GEMM 1: columns with >= 2 / 3 digits, packed (digit, column) rows (i8_slot)
    bool rhs_set = false;
    // device buffers
    half8* rfrag = nullptr;
    size_t rfrag_bytes = 0;
    float* rscale = nullptr;      // [16*ntl_max]
    double* csum = nullptr;       // [16*ntl_max]
    uint32_t* colmax = nullptr;   // [16*ntl_max]
    uint32_t* keepmask = nullptr; // [n_pad/16]
    float* keep = nullptr;        // [n_pad] private copy
    // per sub-block workspace (one zeroed region first: P | wmax | corr)
    uint8_t* ws = nullptr;
    size_t ws_bytes = 0, zero_bytes = 0, zero_tail = 0;   // zero_tail: offset of the small per-segment slots behind P
    float* P = nullptr;
    uint32_t* wmax = nullptr;
    double* corr = nullptr;
    double* vsq = nullptr;        // [sub_snps][n_norm] squares of the norm columns, summed per segment by norm_reduce()
    float* vcross = nullptr;      // [sub_snps][n_cross] V of the cross columns, Gram per segment by cross_reduce_kernel
    float* mu = nullptr;
    float* Wtmp = nullptr;
    float* wscale = nullptr;
    half8* wfrag = nullptr;
    int32_t* rows32 = nullptr;
    float* partial = nullptr;     // [max 3][n_sketch<=32][n_pad] K-split partial sketches (allocated on first use)
    size_t partial_bytes = 0;
    double* dots_partial = nullptr;   // workgroup partials of pyrhe_engine_loo_dots (allocated on first use)
    size_t dots_partial_bytes = 0;
    const void* rimg_compact = nullptr;   // compact last tile of the int8 operand image (set_rhs), or NULL
    float* dots_y32 = nullptr;        // float32 copies of the fixed rows of pyrhe_engine_loo_dots_multi (float32 form)
    size_t dots_y32_bytes = 0;
    // Two workspace sets + two internal streams: GEMM 1 (+ mid) of sub-block g+1 runs on stream A while GEMM 2 of
    // sub-block g runs on stream B -- a DRAM-pattern-bound kernel beside an instruction-bound one.  The members
    // above alias the set in use (kernel arguments are captured by value at launch).
    struct WorkSet {
        uint8_t* base; float* P; uint32_t* wmax; double* corr; double* vsq; float* vcross; float* mu; float* Wtmp; float* wscale;
        half8* wfrag; int32_t* rows32;
    } sets[2];
    // dominance (RHE_DOM) extras, allocated on first use: one zeroed region (P2 | wmax2) + vsq2, W2, cmu, zeros
    uint8_t* dom_ws = nullptr;
    size_t dom_zero_bytes = 0, dom_wmax_off = 0;
    float *P2 = nullptr, *Wtmp2 = nullptr, *cmu = nullptr, *zcoef = nullptr, *vcross2 = nullptr, *W1d = nullptr;
    uint32_t *wmax2 = nullptr, *wmax1d = nullptr;
    double* vsq2 = nullptr;
    hipStream_t sA = nullptr, sB = nullptr;
    hipEvent_t evIn = nullptr, evA[2] = {nullptr, nullptr}, evB[2] = {nullptr, nullptr}, evDoneA = nullptr, evDoneB = nullptr;
    uint64_t sub_counter = 0;     // sub-blocks enqueued so far (set = counter & 1)
    bool async_join = false;      // true: the caller joins with pyrhe_engine_join instead of at the end of every call
    bool pipeline = false;        // false: GEMM 2 is enqueued on stream A too (serial; clean per-kernel timings)
    bool pending = false;         // work enqueued on the internal streams that the caller's stream has not joined yet
    void use_set(int i) {
        const WorkSet& w = sets[i];
        ws = w.base; P = w.P; wmax = w.wmax; corr = w.corr; vsq = w.vsq; vcross = w.vcross; mu = w.mu; Wtmp = w.Wtmp;
        wscale = w.wscale; wfrag = w.wfrag; rows32 = w.rows32;
    }
    // profiling
    bool profile = false;
    struct Ev { int kind; hipEvent_t a, b; };
    std::vector<Ev> evs;
    std::vector<hipEvent_t> pool;   // events are created once and reused across profiled passes
    size_t pool_used = 0;
    bool prof_failed = false;       // an event could not be created: the profile of this pass is void (profile_read says so)
    hipEvent_t take_event() {
        if (pool_used == pool.size()) {
            hipEvent_t ev = nullptr;
            if (hipEventCreate(&ev) != hipSuccess) { prof_failed = true; return nullptr; }
            pool.push_back(ev);
        }
        return pool[pool_used++];
    }
    size_t total_bytes = 0;
    // launch-shape knobs (pyrhe_engine_set_option); every value computes the same results
    int opt_g1_wgs = 1536, opt_g2_packed = 1, opt_g2_iw = 0, opt_g2_splits = 0, opt_dots_generic = 0, opt_g1_split_major = 0, opt_g1_wide = 1, opt_g2_merge = 1536, opt_g1_zero_rows = 1, opt_list_span = 0, opt_g1_fill = 1, opt_dom_seg = 1, opt_dots_f32 = 2, opt_dots_tiles = 2, opt_g1_line = 1, opt_dom_fuse = 1, opt_g1_compact = 1, n_cu = 0;
    // kernels of the last sketch call (pyrhe_engine_info 6..8)
    int last_g1 = 0, last_g2 = 0, last_flow = 0, last_g2_merged = 0;
#ifdef PYRHE_DIAG
    int diag_g1_plain_store = 0, diag_g1_no_barrier = 0;
    long long* diag_g1_stats = nullptr;   // device uint64[6]: cycles per loop phase + waves, summed over the launches    // diagnostic build only (tools/): never compiled into libpyrhe_sketch.so
#endif
};

static constexpr int kMaxTilesI8 = 6; // int8 GEMM 1: 16-row tiles of packed (digit, column) rows (gemm1_i8_kernel<6>, P's 96 slots per entry)
static constexpr int kMaxTiles = 2;  // 32 logical columns: fp16 operand tiles, sketch columns of GEMM 2 (all precisions)
static constexpr int kColCap = 48;   // per-column arrays (scale, sum, maximum): PYRHE_MAX_COLS + the ones column, rounded up
static_assert(PYRHE_MAX_COLS + 1 <= kColCap, "column arrays hold every user column and the ones column");
// Packed int8 GEMM 2: dynamic LDS of a workgroup = the row list of its segment group (128 B per 32-entry K block), the
// per-segment epilogue parameters (correction + 1 / scale of kSegCols columns), the keep mask of its <= 1024 samples and the
// epilogue's staging buffer (which lies over the row list when the group is one segment); 64 KB at most, so the longest
// segment is 472 K blocks = 15 104 entries.
static constexpr int kG2pSegCap = 472 * 32;
static inline size_t g2p_lds(int kb32, int nseg, int ng) {
    const size_t rows = (size_t)kb32 * 128, stage = (size_t)g2p_stage_floats(ng) * sizeof(float);
    return (nseg > 1 ? rows + stage : std::max(rows, stage)) + (size_t)nseg * 2 * kSegCols * sizeof(float) + 1024 * sizeof(float);
}
// Two-stream overlap of GEMM 1 (next sub-block) with GEMM 2 (current): measured +3.5 % at N=500k (the two kernels
// mostly time-share the CUs), at the price of per-kernel timings that include the contention.  Off by default;
// pyrhe_engine_set_pipeline(e, 1) turns it on.

// Allocation failures inside pyrhe_engine_create: release everything made so far (destroy takes a half-built engine).
#define HIPCHK_CREATE(expr)                                                                       \
    do {                                                                                          \
        hipError_t _e = (expr);                                                                   \
        if (_e != hipSuccess) {                                                                   \
            (void)pyrhe_engine_destroy(e);                                                        \
            return fail((int)_e, "%s failed: %s (%s:%d)", #expr, hipGetErrorString(_e), __FILE__, \
                        __LINE__);                                                                \
        }                                                                                         \
    } while (0)

extern "C" int pyrhe_engine_create(int device, int64_t n_keep, int64_t pitch, int64_t max_entries, int precision,
                                   int64_t sub_snps, pyrhe_engine_t** out) {
    if (!out) return fail(PYRHE_EINVAL, "out is NULL");
    if (pitch <= 0 || pitch % 256) return fail(PYRHE_EINVAL, "pitch (%lld) must be a positive multiple of 256", (long long)pitch);
    if (n_keep <= 0 || n_keep > 4 * pitch) return fail(PYRHE_EINVAL, "n_keep (%lld) must be in 1..4*pitch", (long long)n_keep);
    if (n_keep >= (1 << 23)) return fail(PYRHE_EINVAL, "n_keep (%lld) >= 2^23: genotype sums would not be exact in float32", (long long)n_keep);
    if (precision != PYRHE_PREC_F16 && precision != PYRHE_PREC_F16X2 && precision != PYRHE_PREC_F16_EXACT &&
        precision != PYRHE_PREC_I8X3 && precision != PYRHE_PREC_I8X3_F16)
        return fail(PYRHE_EINVAL, "unknown precision %d", precision);
    if (max_entries <= 0) return fail(PYRHE_EINVAL, "max_entries must be positive");
    HIPCHK(hipSetDevice(device));
    auto* e = new pyrhe_engine();
    e->sets[0] = e->sets[1] = pyrhe_engine::WorkSet{};
    e->device = device;
    e->pipeline = false;
    {
        int ncu = 0;
        if (hipDeviceGetAttribute(&ncu, hipDeviceAttributeMultiprocessorCount, device) == hipSuccess) e->n_cu = ncu;
    }
    e->n_keep = n_keep;
    e->pitch = pitch;
    e->n_pad = 4 * pitch;
    e->precision = precision;
    e->parts = (precision == PYRHE_PREC_F16X2 || precision == PYRHE_PREC_I8X3) ? 2 : 1;
    e->exact = precision != PYRHE_PREC_F16 && precision != PYRHE_PREC_I8X3_F16;
    e->i8 = precision == PYRHE_PREC_I8X3 || precision == PYRHE_PREC_I8X3_F16;
    // int8 GEMM 1 accumulates |digit| <= 128 times a genotype <= 3 (an unfilled missing code) over the samples in int32
    if (e->i8 && (int64_t)e->n_pad * 3 * 128 >= (int64_t)1 << 31) {
        const long long np = (long long)e->n_pad;
        delete e;
        return fail(PYRHE_EINVAL, "%lld samples overflow the int32 accumulators of the int8 modes; use PYRHE_PREC_F16X2", np);
    }
    if (sub_snps <= 0) {
        // Whole blocks per kernel pair.  Measured on MI355X (tools/microbench.hip E2): re-reading a sub-block that
        // was just streamed (Infinity Cache) is at most ~20 % faster than HBM, while short launches lose more than
        // that to fill/drain -- so sub-blocks only bound the workspace (P, W image: 16384 x 32 floats = 2 MiB).
        sub_snps = 16384;
    }
    sub_snps = (sub_snps + 31) / 32 * 32;
    if (sub_snps > max_entries) sub_snps = (max_entries + 31) / 32 * 32;
    e->sub_snps = sub_snps;
    e->max_entries = max_entries;
    const int ncolmax = 16 * kMaxTiles;
    e->rfrag_bytes = (size_t)e->n_pad * ncolmax * e->parts * sizeof(_Float16);
    // int8 engines stage the packed digit image here: 16 bytes per sample and tile, up to kMaxTilesI8 tiles (set_rhs), plus the
    // compact copy of a nearly empty last tile (2 bytes per sample).  PYRHE_PREC_I8X3_F16 has parts = 1: the fp16 size alone
    // (64 bytes per sample) would not hold five or six tiles.
    if (e->i8) e->rfrag_bytes = std::max(e->rfrag_bytes, (size_t)e->n_pad * (16 * kMaxTilesI8 + 2));
    HIPCHK_CREATE(hipMalloc(&e->rfrag, e->rfrag_bytes));
    HIPCHK_CREATE(hipMalloc(&e->rscale, kColCap * sizeof(float)));
    HIPCHK_CREATE(hipMalloc(&e->csum, kColCap * sizeof(double)));
    HIPCHK_CREATE(hipMalloc(&e->colmax, kColCap * sizeof(uint32_t)));
    HIPCHK_CREATE(hipMalloc(&e->keepmask, e->n_pad / 16 * sizeof(uint32_t)));
    HIPCHK_CREATE(hipMalloc(&e->keep, e->n_pad * sizeof(float)));
    // workspace
    const size_t S = (size_t)sub_snps;
    size_t off = 0;
    auto carve = [&](size_t bytes) { size_t o = off; off += (bytes + 255) / 256 * 256; return o; };
    const size_t oP = carve(S * ncolmax * sizeof(float) * (e->i8 ? 3 : 1));
    static_assert(kSegCols == 16 * kMaxTiles, "segment slots hold one value per logical column");
    const size_t oWmax = carve((size_t)kMaxSeg * kSegCols * sizeof(uint32_t));
    e->zero_tail = oWmax;
    const size_t oCorr = carve((size_t)kMaxSeg * kSegCols * sizeof(double));
    e->zero_bytes = off;
    const size_t oVsq = carve(S * PYRHE_MAX_COLS * sizeof(double));
    const size_t oVcr = carve(S * PYRHE_MAX_COLS * sizeof(float));
    const size_t oMu = carve(S * sizeof(float));
    const size_t oWtmp = carve(S * ncolmax * sizeof(float));
    const size_t oWscale = carve((size_t)kMaxSeg * kSegCols * sizeof(float));
    // every segment pads its tail to a 32-entry K block (64 for the int8 GEMM 2)
    const size_t oWfrag = carve((S / 32 + 4 * kMaxSeg) * e->parts * kMaxTiles * 64 * sizeof(half8));   // segments pad to 128 entries
    const size_t oRows = carve((S + 128 + 128 * kMaxSeg) * sizeof(int32_t));
    e->ws_bytes = off;
    for (int i = 0; i < 2; ++i) {
        uint8_t* base = nullptr;
        HIPCHK_CREATE(hipMalloc(&base, e->ws_bytes));
        pyrhe_engine::WorkSet& w = e->sets[i];
        w.base = base;                                  // owned by the engine from here on (destroy frees it if the memset fails)
        HIPCHK_CREATE(hipMemset(base, 0, e->ws_bytes));
        w.P = (float*)(base + oP);
        w.wmax = (uint32_t*)(base + oWmax);
        w.corr = (double*)(base + oCorr);
        w.vsq = (double*)(base + oVsq);
        w.vcross = (float*)(base + oVcr);
        w.mu = (float*)(base + oMu);
        w.Wtmp = (float*)(base + oWtmp);
        w.wscale = (float*)(base + oWscale);
        w.wfrag = (half8*)(base + oWfrag);
        w.rows32 = (int32_t*)(base + oRows);
        HIPCHK_CREATE(hipEventCreateWithFlags(&e->evA[i], hipEventDisableTiming));
        HIPCHK_CREATE(hipEventCreateWithFlags(&e->evB[i], hipEventDisableTiming));
    }
    e->use_set(0);
    HIPCHK_CREATE(hipStreamCreateWithFlags(&e->sA, hipStreamNonBlocking));
    HIPCHK_CREATE(hipStreamCreateWithFlags(&e->sB, hipStreamNonBlocking));
    HIPCHK_CREATE(hipEventCreateWithFlags(&e->evIn, hipEventDisableTiming));
    HIPCHK_CREATE(hipEventCreateWithFlags(&e->evDoneA, hipEventDisableTiming));
    HIPCHK_CREATE(hipEventCreateWithFlags(&e->evDoneB, hipEventDisableTiming));
    e->total_bytes = 2 * e->ws_bytes + e->rfrag_bytes + e->n_pad * 4 + e->n_pad / 4;
    *out = e;
    return PYRHE_OK;
}

extern "C" int pyrhe_engine_destroy(pyrhe_engine_t* e) {
    if (!e) return PYRHE_OK;
    hipSetDevice(e->device);
    for (auto& ev : e->pool) (void)hipEventDestroy(ev);
    hipFree(e->rfrag); hipFree(e->rscale); hipFree(e->csum); hipFree(e->colmax);
    if (e->sA) { (void)hipStreamSynchronize(e->sA); (void)hipStreamDestroy(e->sA); }
    if (e->sB) { (void)hipStreamSynchronize(e->sB); (void)hipStreamDestroy(e->sB); }
    for (int i = 0; i < 2; ++i) {
        if (e->evA[i]) (void)hipEventDestroy(e->evA[i]);
        if (e->evB[i]) (void)hipEventDestroy(e->evB[i]);
        hipFree(e->sets[i].base);
    }
    if (e->evIn) (void)hipEventDestroy(e->evIn);
    if (e->evDoneA) (void)hipEventDestroy(e->evDoneA);
    if (e->evDoneB) (void)hipEventDestroy(e->evDoneB);
    hipFree(e->keepmask); hipFree(e->keep); hipFree(e->partial); hipFree(e->dom_ws); hipFree(e->dots_partial); hipFree(e->dots_y32);
    delete e;
    return PYRHE_OK;
}

extern "C" int64_t pyrhe_engine_info(const pyrhe_engine_t* e, int what) {
    if (!e) return -1;
    switch (what) {
        case 0: return e->n_pad;
        case 1: return e->sub_snps;
        case 2: return e->ntl;
        case 3: return e->n_sketch;
        case 4: return e->n_norm;
        case 5: return (int64_t)e->total_bytes;
        case 6: return e->last_g1;
        case 7: return e->last_g2;
        case 8: return e->last_flow;
        case 9: return e->last_g2_merged;
        case 10: return e->n_cross;
        case 11: return e->nslots;
        default: return -1;
    }
}

extern "C" int pyrhe_engine_set_rhs(pyrhe_engine_t* e, const float* R, const int32_t* role_host, int n_cols,
                                    const float* keep, void* stream_) {
    if (!e || !R || !role_host || !keep) return fail(PYRHE_EINVAL, "NULL argument");
    const int max_cols = e && e->i8 ? PYRHE_MAX_COLS : PYRHE_MAX_COLS_FP16;
    if (n_cols < 1 || n_cols > max_cols) return fail(PYRHE_EINVAL, "n_cols %d outside 1..%d", n_cols, max_cols);
    int ns = 0, nn = 0, nx = 0, nd2 = 0, nd3 = 0, last_nd = 3;
    for (int c = 0; c < n_cols; ++c) {
        const int nd = (role_host[c] >> 8) & 3 ? (role_host[c] >> 8) & 3 : 3;
        if (nd > last_nd) return fail(PYRHE_EINVAL, "digit counts must not grow with the column index (column %d: %d after %d)", c, nd, last_nd);
        last_nd = nd;
        nd2 += nd >= 2;
        nd3 += nd >= 3;
        const int role = role_host[c] & 0xFF;
        if (role == PYRHE_COL_SKETCH && nd != 3) return fail(PYRHE_EINVAL, "sketch columns carry all three digits (column %d)", c);
        if (role == PYRHE_COL_SKETCH) {
            if (nn || nx) return fail(PYRHE_EINVAL, "sketch columns must precede norm and cross columns (column %d)", c);
            ++ns;
        } else if (role == PYRHE_COL_NORM) {
            if (nx) return fail(PYRHE_EINVAL, "norm columns must precede cross columns (column %d)", c);
            ++nn;
        } else if (role == PYRHE_COL_CROSS) {
            ++nx;
        } else {
            return fail(PYRHE_EINVAL, "role[%d] = %d is not SKETCH/NORM/CROSS", c, role_host[c]);
        }
    }
    // fp16 operand image: two 16-column tiles; int8 image: six 16-row tiles of packed (digit, column) rows (the accumulators of
    // gemm1_i8_kernel<6>, P's 96 slots per entry); the second contraction takes <= 32 sketch columns in every precision.
    // Checked before anything of the engine's state changes: a refused call leaves the installed columns as they were.
    const int nslots = n_cols + 1 + nd2 + nd3;
    if (!e->i8 && (n_cols + 1 + 15) / 16 > kMaxTiles) return fail(PYRHE_EINVAL, "too many columns");
    if (e->i8 && (nslots + 15) / 16 > kMaxTilesI8) return fail(PYRHE_EINVAL, "%d packed digit rows exceed the six tiles of the int8 GEMM 1 (fewer columns or digits)", nslots);
    if (e->i8 && (size_t)e->n_pad * 16 * ((nslots + 15) / 16) > e->rfrag_bytes)
        return fail(PYRHE_EINVAL, "the operand image of %d tiles (%zu bytes) does not fit the engine's buffer (%zu bytes)", (nslots + 15) / 16,
                    (size_t)e->n_pad * 16 * ((nslots + 15) / 16), e->rfrag_bytes);
    if (ns > 16 * kMaxTiles) return fail(PYRHE_EINVAL, "%d sketch columns exceed the %d of the second contraction", ns, 16 * kMaxTiles);
    hipStream_t stream = (hipStream_t)stream_;
    HIPCHK(hipSetDevice(e->device));
    if (e->pending) {      // earlier sketches on the internal streams still read the old operand image
        HIPCHK(hipStreamWaitEvent(stream, e->evDoneA, 0));
        HIPCHK(hipStreamWaitEvent(stream, e->evDoneB, 0));
    }
    e->n_sketch = ns;
    e->n_norm = nn;
    e->n_cross = nx;
    e->n_user = n_cols;
    e->ntl = (n_cols + 1 + 15) / 16;
    e->nd2 = nd2;
    e->nd3 = nd3;
    e->nslots = nslots;
    e->nt8 = (nslots + 15) / 16;                  // int8 GEMM 1: tiles of packed (digit, column) rows, see i8_slot()
    e->nto = ns ? (ns + 15) / 16 : 1;
    HIPCHK(hipMemcpyAsync(e->keep, keep, e->n_pad * sizeof(float), hipMemcpyDeviceToDevice, stream));
    HIPCHK(hipMemsetAsync(e->colmax, 0, kColCap * sizeof(uint32_t), stream));
    HIPCHK(hipMemsetAsync(e->csum, 0, kColCap * sizeof(double), stream));
    {
        dim3 grid((unsigned)std::min<int64_t>((e->n_pad + 255) / 256, 1024), n_cols);
        col_absmax_kernel<<<grid, 256, 0, stream>>>(R, e->n_pad, n_cols, e->colmax);
    }
    {
        const int64_t threads = (e->n_pad / 256) * 8 * e->parts * e->ntl * 64;
        const unsigned grid = (unsigned)((threads + 255) / 256);
        if (e->i8) {
            const int64_t th = (e->n_pad / 256) * 4 * e->nt8 * 64;
            prep_rimg_i8_kernel<<<(unsigned)((th + 255) / 256), 256, 0, stream>>>(R, e->keep, e->n_pad, n_cols, e->nd2, e->nd3, e->nt8, e->colmax, e->rscale,
                                                                              (int4v*)e->rfrag, e->csum);
            // at most two live rows in the last of two or three tiles (the headline's 34 digit rows): a compact copy of that tile
            // behind the image for gemm1_i8w_kernel<.., CL> (2 bytes per sample; the image buffer is sized for kMaxTiles tiles)
            e->rimg_compact = nullptr;
            const int live_last_rows = e->nslots - 16 * (e->nt8 - 1);
            if (e->nt8 >= 2 && e->nt8 <= 3 && live_last_rows <= 2 && (size_t)e->n_pad * (16 * e->nt8 + 2) <= e->rfrag_bytes) {
                int4v* comp = (int4v*)e->rfrag + (size_t)e->n_pad * e->nt8;          // the image holds n_pad * nt8 int4v
                const int64_t n_steps = e->n_pad / 512, thc = n_steps * 64;
                prep_rimg_i8_compact_kernel<<<(unsigned)((thc + 255) / 256), 256, 0, stream>>>((const int4v*)e->rfrag, e->nt8, n_steps, comp);
                e->rimg_compact = comp;
            }
        } else if (e->parts == 2)
            prep_rfrag_kernel<2, true><<<grid, 256, 0, stream>>>(R, e->keep, e->n_pad, n_cols, e->ntl, e->colmax, e->rscale, e->rfrag, e->csum);
        else if (e->exact)
            prep_rfrag_kernel<1, true><<<grid, 256, 0, stream>>>(R, e->keep, e->n_pad, n_cols, e->ntl, e->colmax, e->rscale, e->rfrag, e->csum);
        else
            prep_rfrag_kernel<1, false><<<grid, 256, 0, stream>>>(R, e->keep, e->n_pad, n_cols, e->ntl, e->colmax, e->rscale, e->rfrag, e->csum);
    }
    {
        const int64_t words = e->n_pad / 16;
        prep_keepmask_kernel<<<(unsigned)((words + 255) / 256), 256, 0, stream>>>(e->keep, e->n_pad, e->keepmask);
    }
    HIPCHK(hipGetLastError());
    e->rhs_set = true;
    return PYRHE_OK;
}

// ---- launch helpers --------------------------------------------------------------------------
struct ProfScope {
    pyrhe_engine* e;
    hipStream_t s;
    int kind;
    hipEvent_t a = nullptr, b = nullptr;
    ProfScope(pyrhe_engine* e_, hipStream_t s_, int kind_) : e(e_), s(s_), kind(kind_) {
        if (e->profile) {
            a = e->take_event();
            b = e->take_event();
            if (a && b) (void)hipEventRecord(a, s);
        }
    }
    ~ProfScope() {
        if (e->profile && a && b) {
            (void)hipEventRecord(b, s);
            e->evs.push_back({kind, a, b});
        }
    }
};

template <int NTL, int PARTS, bool EXACT, int GV = 0>
static void launch_gemm1(pyrhe_engine* e, const uint8_t* packed, const int32_t* rows, int n, const float* fill,
                         int n_steps, int sps, int splits, hipStream_t stream) {
    dim3 grid((n + 255) / 256, splits);
    const size_t lds = (size_t)2 * 2 * 8 * PARTS * NTL * 1024;   // [2 buffers][2 chunks][8 PARTS NTL fragments][1 KiB]
#define L1(MISS_, VC_) gemm1_kernel<NTL, PARTS, EXACT, MISS_, GV, VC_><<<grid, 256, lds, stream>>>(packed, e->pitch, rows, n, fill, e->rfrag, n_steps, sps, e->P, splits > 1)
    if (e->vc) { if (fill) L1(true, true); else L1(false, true); }
    else { if (fill) L1(true, false); else L1(false, false); }
#undef L1
}

template <int NTO, int PARTS, int IW, bool EXACT, int GV = 0>
static void launch_gemm2(pyrhe_engine* e, const uint8_t* packed, int n_kb, int kbps, int splits, const float* fill,
                         float* out, int mode, float* partial, hipStream_t stream, const SegTab& tab) {
    // segment mode (tab.nseg > 0): `splits` = number of segments, `kbps` = K blocks of the longest one (LDS row list)
    dim3 grid((unsigned)((e->n_pad + 1024 * IW - 1) / (1024 * IW)), splits);
#define L2(MISS_, VC_) gemm2_kernel<NTO, PARTS, IW, EXACT, MISS_, GV, VC_><<<grid, 256, (size_t)kbps * 128, stream>>>(packed, e->pitch, e->rows32, n_kb, kbps, fill, e->wfrag, e->corr, e->wscale, e->keep, e->n_sketch, e->n_pad, out, mode, partial, tab)
    if (e->vc) { if (fill) L2(true, true); else L2(false, true); }
    else { if (fill) L2(true, false); else L2(false, false); }
#undef L2
}
// One implementation behind pyrhe_engine_sketch_block (additive sketches) and pyrhe_engine_sketch_block_dom
// (additive + dominance sketches, RHE_DOM).
static int sketch_impl(pyrhe_engine_t* e, const uint8_t* packed, int64_t n_rows, const int32_t* entries,
                       const int64_t* offsets_host, int nbin, const float* fill, float* out_sketch, double* out_norm,
                       int32_t* out_colsum, int accumulate, float* out_dom, double* norm_dom, double* out_cross,
                       double* cross_dom, void* stream_) {
    if (!e || !packed || !offsets_host) return fail(PYRHE_EINVAL, "NULL argument");
    if (!e->rhs_set) return fail(PYRHE_ESTATE, "pyrhe_engine_set_rhs has not been called");
    if (nbin < 1) return fail(PYRHE_EINVAL, "nbin must be >= 1");
    if (n_rows < 0) return fail(PYRHE_EINVAL, "n_rows negative");
    if (e->n_sketch && !out_sketch) return fail(PYRHE_EINVAL, "out_sketch is NULL but there are sketch columns");
    if (e->n_norm && !out_norm) return fail(PYRHE_EINVAL, "out_norm is NULL but there are norm columns");
    if (e->n_cross && !out_cross) return fail(PYRHE_EINVAL, "out_cross is NULL but there are cross columns (pyrhe_engine_sketch_block_cross)");
    for (int b = 0; b < nbin; ++b) {
        if (offsets_host[b + 1] < offsets_host[b]) return fail(PYRHE_EINVAL, "offsets not monotone at bin %d", b);
        if (offsets_host[b + 1] - offsets_host[b] > e->max_entries)
            return fail(PYRHE_EINVAL, "bin %d has %lld entries > max_entries %lld", b,
                        (long long)(offsets_host[b + 1] - offsets_host[b]), (long long)e->max_entries);
    }
    if (offsets_host[nbin] > 0 && !entries) return fail(PYRHE_EINVAL, "entries is NULL");
    const bool wide_rows = n_rows >= (1 << 24) || (uint64_t)n_rows * (uint64_t)e->pitch >= (1ull << 32);
    if (n_rows >= (1ll << 31) || e->pitch >= (1 << 24) || (wide_rows && !e->opt_list_span))
        return fail(PYRHE_EINVAL, "block of %lld rows x %lld bytes exceeds the 32-bit row offsets of one call; split it",
                    (long long)n_rows, (long long)e->pitch);
    const bool dom = out_dom != nullptr || norm_dom != nullptr || cross_dom != nullptr;
    if (dom && ((e->n_sketch && !out_dom) || (e->n_norm && !norm_dom) || (e->n_cross && !cross_dom))) return fail(PYRHE_EINVAL, "dominance outputs incomplete");
    const int64_t cr_stride = (int64_t)e->n_cross * e->n_cross;
    hipStream_t caller = (hipStream_t)stream_;
    HIPCHK(hipSetDevice(e->device));
    // Serial mode (default): everything goes to the caller's stream, no events.  Pipelined mode: private streams A
    // (GEMM 1 + mid kernels) and B (GEMM 2), ordered against the caller's stream by events.
    const bool piped = e->pipeline && !dom;
    if (!piped && e->pending) {      // leftovers of earlier pipelined calls: order them before this call
        HIPCHK(hipStreamWaitEvent(caller, e->evDoneA, 0));
        HIPCHK(hipStreamWaitEvent(caller, e->evDoneB, 0));
        e->pending = false;
    }
    if (piped) {
        // inputs (and the output buffers) are ready on the caller's stream at this point of ITS order
        HIPCHK(hipEventRecord(e->evIn, caller));
        HIPCHK(hipStreamWaitEvent(e->sA, e->evIn, 0));
        HIPCHK(hipStreamWaitEvent(e->sB, e->evIn, 0));
    }
    hipStream_t stream = piped ? e->sA : caller;
    hipStream_t sB2 = piped ? e->sB : caller;
    const int n_chunks = (int)(e->pitch / 64);
    const int ncol_p = 16 * (e->i8 ? e->nt8 : e->ntl);       // values (fp16 form) / packed int32 slots (int8 form) per entry of P
    const int64_t sk_stride = (int64_t)e->n_sketch * e->n_pad;
    const float pscale = e->exact ? 1.0f : 16777216.0f;
    const int kstep = 32;
    // limb-packed int8 GEMM 2 (see gemm2_i8p_kernel): segment mode, <= 10 sketch columns, additive view
    const bool g2p_off = e->opt_g2_packed == 0;
    const bool g2p_ok = !g2p_off && e->i8 && e->parts == 2 && e->n_sketch >= 1 && e->n_sketch <= 2 * kI8pCols && !dom;
    const int g2p_ng = e->n_sketch <= kI8pCols ? 1 : 2;        // column groups of 10 = pairs of A tiles
    // Several short lists per call (the bins of a block): consecutive segments share a workgroup of the packed GEMM 2 as long
    // as the launch keeps opt_g2_merge workgroups; such segments are padded to whole ring turns (128 entries), others to 64.
    const int g2p_gx = (int)((e->n_pad + 1023) / 1024) * g2p_ng;
    const bool g2p_merge = g2p_ok && e->opt_g2_merge > 0 && (int64_t)nbin * g2p_gx / e->opt_g2_merge >= 2;
    // RHE_DOM (per-list flow, three GEMM 2 launches per list, the third under the [g == 2] view): the packed kernel as a
    // one-segment launch, for cohorts whose sample range alone fills the chip; lists go in pieces of <= 16384 entries
    const bool g2p_dom = !g2p_off && e->i8 && e->parts == 2 && dom && e->n_sketch >= 1 && e->n_sketch <= kI8pCols &&
                         e->n_pad >= 192 * 1024;

    if (dom && !e->dom_ws) {
        const size_t S = (size_t)e->sub_snps, nc = 16 * kMaxTiles;
        size_t off = 0;
        auto carve = [&](size_t bytes) { size_t o = off; off += (bytes + 255) / 256 * 256; return o; };
        const size_t oP2 = carve(S * nc * sizeof(float) * (e->i8 ? 3 : 1));
        const size_t oWm = carve((size_t)kMaxSeg * kSegCols * sizeof(uint32_t));       // per segment (segmented flow), slot 0 otherwise
        const size_t oWm1 = carve((size_t)kMaxSeg * kSegCols * sizeof(uint32_t));      // maxima of W1 when all three operands live at once (fused GEMM 2)
        e->dom_zero_bytes = off;
        e->dom_wmax_off = oWm;
        const size_t oNa = carve(S * PYRHE_MAX_COLS * sizeof(double));
        const size_t oW2 = carve(S * nc * sizeof(float));
        const size_t oW1 = carve(S * kI8pCols * sizeof(float));
        const size_t oCm = carve(S * sizeof(float));
        const size_t oZc = carve(S * sizeof(float));
        const size_t oVc2 = carve(S * PYRHE_MAX_COLS * sizeof(float));
        HIPCHK(hipMalloc(&e->dom_ws, off));
        HIPCHK(hipMemset(e->dom_ws, 0, off));
        e->P2 = (float*)(e->dom_ws + oP2);
        e->wmax2 = (uint32_t*)(e->dom_ws + oWm);
        e->vsq2 = (double*)(e->dom_ws + oNa);
        e->Wtmp2 = (float*)(e->dom_ws + oW2);
        e->W1d = (float*)(e->dom_ws + oW1);
        e->wmax1d = (uint32_t*)(e->dom_ws + oWm1);
        e->cmu = (float*)(e->dom_ws + oCm);
        e->zcoef = (float*)(e->dom_ws + oZc);
        e->vcross2 = (float*)(e->dom_ws + oVc2);
    }

    // zero P for the n entries of this launch (not the whole sub_snps-sized buffer: 25 MB per call at 65536 entries) and the
    // per-segment maxima / corrections behind it
    auto zero_ws = [&](int n) -> int {
        const size_t pbytes = std::min(e->zero_tail, (size_t)n * ncol_p * sizeof(float));
        HIPCHK(hipMemsetAsync(e->ws, 0, pbytes, stream));
        HIPCHK(hipMemsetAsync(e->ws + e->zero_tail, 0, e->zero_bytes - e->zero_tail, stream));
        return PYRHE_OK;
    };

    // ---- GEMM 1 of one entry list, genotype view gv, into Pbuf (zeroed by the caller)
    auto run_gemm1 = [&](int gv, const int32_t* rows, int n, float* Pbuf, float* Pbuf2 = nullptr) {
        // waves per workgroup (they share one staged R slice).  6 and 12 were built and measured in round 2 (template parameter
        // WAVES of gemm1_i8_kernel: a third of the R traffic at 12) and are slower -- 31.7 / 28.9 ms per pass against 27.6 at
        // the headline shape: the barrier of a 12-wave workgroup costs 32 % of its waves' time.
        constexpr int g1_waves = 4;
        // "wide" kernel (gemm1_i8w_kernel): 128 rows per wave, 512 per workgroup, half the R traffic, two waves per SIMD
        // Measured (round 2): three tiles at N = 500k 25.3 vs 27.6 ms per pass; four tiles (236 registers) 37.6 vs 31.7; N = 50k
        // 1.70 vs 1.65 -- so: up to three tiles and cohorts whose sample range alone gives every workgroup a long loop
        // (opt_g1_wide: 0 never, 1 by that rule, 2 whenever the registers allow)
        // Smaller cohorts take it too once a launch has rows enough to fill the chip with 512-row workgroups (N = 50k with 26
        // blocks = 130 000 rows per call: 1.46 vs 1.56 ms per pass; with 3 000 rows per call the 64-row kernel stays).
        const int64_t wide_par = (int64_t)((n + 511) / 512) * std::max(1, n_chunks / 8);      // row groups x largest split count
        const bool wide = e->i8 && gv == 0 && ((e->opt_g1_wide == 1 && e->nt8 <= 3 && (e->n_pad >= 192 * 1024 || wide_par >= 1024)) ||
                                               (e->opt_g1_wide == 2 && e->nt8 <= 4));
        const int gx = wide ? (n + 511) / 512 : (n + 64 * g1_waves - 1) / (64 * g1_waves);
        const int g1_ch = (e->i8 && e->nt8 > 3) ? 1 : 2;  // 256-sample chunks per step (see gemm1_i8_kernel)
        const int n_steps = n_chunks / g1_ch;             // pitch is a multiple of 256 B = 4 chunks
        const int g1_wgs = std::max(1, e->opt_g1_wgs * 4 / g1_waves);     // target workgroups per launch: 2 rounds of 3 per CU, or (wide: 2 per CU) 3 rounds -- measured 1024 / 1536 / 3072 wide workgroups: 25.3 / 24.5 / 26.5 ms
        int splits = (g1_wgs + gx - 1) / gx;
        // Every workgroup pays a fixed price -- the prologue and its 128 (64) rows x 16 NT slots of integer atomics per wave --
        // that a short loop does not amortise: where the row groups alone already fill the chip, no sample range gets fewer than 32
        // steps (N = 50k, 26 blocks per call: three splits of 32 steps instead of six of 16: GEMM 1 1.44 -> 1.35 ms per pass,
        // tools/ab.sh [c2_knobs]; N = 500k has 70 steps per split either way)
        int smax = n_steps / 4 > 0 ? n_steps / 4 : 1;
        if (e->i8 && e->n_cu > 0 && n_steps / 32 >= 1 && (int64_t)gx * (n_steps / 32) >= (int64_t)e->n_cu * 2) smax = std::min(smax, n_steps / 32);
        splits = std::max(1, std::min(splits, smax));
        // full-line loads of the 256-sample-step kernels (gemm1_i8_kernel<.., LINE>): splits of whole line PAIRS (4 chunks)
        const bool line = e->i8 && !wide && g1_ch == 1 && e->opt_g1_line && gv != 2;      // both views at once: no registers left
        const int cmul = line ? 4 : 2;
        auto round_cps = [&](int c) { return (c + cmul - 1) / cmul * cmul; };
        int cps = round_cps((n_steps + splits - 1) / splits);        // steps per split, even (n_steps is even; a multiple of 4 chunks)
        splits = (n_steps + cps - 1) / cps;
        if (e->i8 && e->opt_g1_fill && e->n_cu > 0) {
            // Among the split counts that give 0.65 .. 1.7 x the target, take the one whose workgroups fill whole rounds of the
            // resident slots best: a launch of 1564 workgroups on 512 slots (3.05 rounds) runs a nearly empty fourth round --
            // measured at N = 500k with 10 / 20 / 40 blocks per call: 25.5 / 26.0 / 25.9 ms per pass against 23.0 - 23.3 at
            // 13 / 26 blocks (1778 / 2032 workgroups).  Ties go to fewer splits (fewer atomics).
            const int slots = e->n_cu * (wide ? 2 : (e->nt8 <= 3 ? 3 : 2));
            double best = -1.0;
            for (int s = 1; s <= smax && (int64_t)gx * s <= (int64_t)g1_wgs * 2; ++s) {
                const int c = round_cps((n_steps + s - 1) / s);
                const int s2 = (n_steps + c - 1) / c;
                const int64_t wgs = (int64_t)gx * s2;
                if (wgs * 100 < (int64_t)g1_wgs * 65 || wgs * 10 > (int64_t)g1_wgs * 17) continue;
                const double r = (double)wgs / slots, fill = r / std::ceil(r);
                if (fill > best + 1e-9) { best = fill; splits = s2; cps = c; }
            }
            // Short rows (<= 128 steps: cohorts up to 65 536 samples) whose row groups alone fill ONE round of the resident slots --
            // a full call of such a cohort: 262 144 entries = 512 groups of 512 rows -- take no sample split at all: a third of the
            // prologues, plain stores instead of atomics (N = 50k: GEMM 1 1.29-1.36 -> 1.25 ms per pass, tools/ab.sh [s28])
            if (wide && n_steps <= 128 && gx <= slots && (int64_t)gx * 100 >= (int64_t)slots * 85) {
                splits = 1;
                cps = round_cps(n_steps);
            }
        }
        float* saveP = e->P;
        e->P = Pbuf;
        ProfScope ps(e, stream, 0);
        // grid order (g1_grid_index): 0 row group fast, 1 sample split fast, 2 every XCD works through its own run of sample ranges
        // (int8 kernels only; the mode word then carries the row-group count)
        const int smode = e->i8 ? e->opt_g1_split_major : (e->opt_g1_split_major == 1);
        const int sm = smode == 2 ? (2 | (gx << 2)) : smode;
        dim3 grid(smode == 1 ? splits : gx, smode == 1 ? gx : splits);
        if (smode == 2) grid = dim3((unsigned)(((int64_t)gx * splits + 7) / 8 * 8), 1);
        int ua = splits > 1, g1_diag = 0;
#ifdef PYRHE_DIAG
        if (ua && e->diag_g1_plain_store) ua = 2;          // WRONG results: plain stores instead of atomics, to price the atomics
        g1_diag = e->diag_g1_no_barrier;                    // WRONG results: bit 0 no step barrier, bit 1 no wait for the R image
        if (e->diag_g1_stats && gv != 2) { g1_diag |= 4; Pbuf2 = reinterpret_cast<float*>(e->diag_g1_stats); }   // bit 2: phase cycle account
#endif
        e->last_g1 = e->i8 ? (wide ? 200 : 100) + e->nt8 : e->ntl;
        const int live_last = e->opt_g1_zero_rows ? e->nslots - 16 * (e->nt8 - 1) : 16;     // live digit rows of the last int8 tile
        const bool compact = wide && e->opt_g1_compact && e->rimg_compact && e->nslots - 16 * (e->nt8 - 1) <= 2;   // gemm1_i8w_kernel<.., CL>
        if (e->i8) {
            const size_t lds = (size_t)2 * g1_ch * 4 * e->nt8 * 1024;
#define G1A (packed, e->pitch, rows, n, fill, (const int4v*)e->rfrag, n_steps, cps, (int*)Pbuf, ua, (int*)Pbuf2, sm, g1_diag, live_last)
#define G1I(NT_, MISS_, GV_) do { if (NT_ > 3 && line) { if (e->vc) gemm1_i8_kernel<NT_, MISS_, GV_, true, 4, (NT_ > 3 && GV_ != 2)><<<grid, 256, lds, stream>>>G1A; \
                                                        else gemm1_i8_kernel<NT_, MISS_, GV_, false, 4, (NT_ > 3 && GV_ != 2)><<<grid, 256, lds, stream>>>G1A; } \
                                  else if (e->vc) gemm1_i8_kernel<NT_, MISS_, GV_, true><<<grid, 256, lds, stream>>>G1A; \
                                  else gemm1_i8_kernel<NT_, MISS_, GV_, false><<<grid, 256, lds, stream>>>G1A; } while (0)
#define G1XA (packed, e->pitch, rows, n, fill, (const int4v*)e->rfrag, n_steps, cps, (int*)Pbuf, ua, sm, live_last, (const int4v*)e->rimg_compact)
#define G1X(NT_, MISS_) do { if (compact && NT_ >= 2 && NT_ <= 3) { if (e->vc) gemm1_i8w_kernel<NT_, MISS_, true, 4, (NT_ >= 2 && NT_ <= 3)><<<grid, 256, lds + 4096, stream>>>G1XA; \
                                                                      else gemm1_i8w_kernel<NT_, MISS_, false, 4, (NT_ >= 2 && NT_ <= 3)><<<grid, 256, lds + 4096, stream>>>G1XA; } \
                             else if (e->vc) gemm1_i8w_kernel<NT_, MISS_, true><<<grid, 256, lds, stream>>>G1XA; \
                             else gemm1_i8w_kernel<NT_, MISS_, false><<<grid, 256, lds, stream>>>G1XA; } while (0)
#define G1J(NT_) do { if (wide) { if (fill) G1X(NT_, true); else G1X(NT_, false); break; } if (gv == 0) { if (fill) G1I(NT_, true, 0); else G1I(NT_, false, 0); } \
                      else if (gv == 1) { if (fill) G1I(NT_, true, 1); else G1I(NT_, false, 1); } \
                      else { if (fill) G1I(NT_, true, 2); else G1I(NT_, false, 2); } } while (0)
            switch (e->nt8) {
                case 1: G1J(1); break;
                case 2: G1J(2); break;
                case 3: G1J(3); break;
                case 4: G1J(4); break;
                case 5: G1J(5); break;
                default: G1J(6); break;
            }
#undef G1J
#undef G1X
#undef G1XA
#undef G1I
#undef G1A
        } else {
#define G1(NTL_, PARTS_, EX_, GV_) launch_gemm1<NTL_, PARTS_, EX_, GV_>(e, packed, rows, n, fill, n_steps, cps, splits, stream)
            if (gv == 0) {
                if (e->parts == 2) { if (e->ntl == 1) G1(1, 2, true, 0); else G1(2, 2, true, 0); }
                else if (e->exact) { if (e->ntl == 1) G1(1, 1, true, 0); else G1(2, 1, true, 0); }
                else { if (e->ntl == 1) G1(1, 1, false, 0); else G1(2, 1, false, 0); }
            } else {
                if (e->parts == 2) { if (e->ntl == 1) G1(1, 2, true, 1); else G1(2, 2, true, 1); }
                else if (e->exact) { if (e->ntl == 1) G1(1, 1, true, 1); else G1(2, 1, true, 1); }
                else { if (e->ntl == 1) G1(1, 1, false, 1); else G1(2, 1, false, 1); }
            }
#undef G1
        }
        e->P = saveP;
    };

    // ---- W (float) -> operand image + rank-1 correction, then GEMM 2 with genotype view gv into outp
    // `tab` describes the chunk's segments for the mid kernel; segmode = GEMM 2 too runs all segments in one launch
    // (outp / ndst are then the caller's [nbin] arrays, indexed by tab.bin), otherwise tab has the one segment of the
    // list and GEMM 2 may split K over blockIdx.y.
    static const SegTab no_tab = {};
    auto run_gemm2 = [&](int gv, const float* Wsrc, const float* coef, const uint32_t* wmaxp, int n, int n_kb, float* outp,
                         bool acc_now, int set, const int32_t* rows, const double* nsrc, double* ndst, const SegTab& tab,
                         bool segmode) -> int {
        {
            ProfScope ps(e, stream, 1);
            const int64_t threads = (int64_t)n_kb * e->parts * e->nto * 64;
            const unsigned grid = (unsigned)std::max<int64_t>((threads + 255) / 256, (int64_t)tab.nseg * e->n_norm);
            if ((g2p_ok && segmode && gv == 0) || (g2p_dom && segmode)) {
                const unsigned gridp = (unsigned)std::max<int64_t>(((int64_t)n_kb * 64 * g2p_ng + 255) / 256, (int64_t)tab.nseg * e->n_norm);
                mid2_i8p_kernel<<<gridp, 256, 0, stream>>>(Wsrc, coef, e->n_sketch, 2 * g2p_ng, wmaxp, (int4v*)e->wfrag, e->corr, e->wscale, rows, e->rows32,
                                                           nsrc, ndst, e->n_norm, tab);
            } else if (e->parts == 2) mid2_kernel<2><<<grid, 256, 0, stream>>>(Wsrc, coef, e->n_sketch, e->nto, wmaxp, e->wfrag, e->corr, e->wscale, rows, e->rows32, nsrc, ndst, e->n_norm, tab);
            else mid2_kernel<1><<<grid, 256, 0, stream>>>(Wsrc, coef, e->n_sketch, e->nto, wmaxp, e->wfrag, e->corr, e->wscale, rows, e->rows32, nsrc, ndst, e->n_norm, tab);
        }
        if (piped) {      // GEMM 2 goes to stream B, behind this sub-block's mid kernels on stream A
            HIPCHK(hipEventRecord(e->evA[set], e->sA));
            HIPCHK(hipStreamWaitEvent(sB2, e->evA[set], 0));
        }
        // 512 samples per wave (8-byte loads, one 128-B line per row per 16 lanes) when the accumulators fit; measured at
        // N=500k: the single-rounding pass is fastest with 512-sample waves even at 1 wave/SIMD, the hi+lo pass with
        // 256-sample waves at 2 waves/SIMD
        int iw = (e->nto == 1 && e->parts == 1 && e->n_pad >= 2048 * 160) ? 2 : 1;
        if (e->opt_g2_iw) iw = e->opt_g2_iw == 2 && e->nto == 1 ? 2 : 1;
        const int gx = (int)((e->n_pad + 1024 * iw - 1) / (1024 * iw));
        int splits = gx >= 192 ? 1 : std::min(16, (512 + gx - 1) / gx);
        if (e->opt_g2_splits > 0) splits = e->opt_g2_splits;
        splits = std::max(splits, (n_kb + 511) / 512);   // row list of a K range lives in LDS: <= 64 KiB
        splits = std::max(1, std::min(splits, n_kb));
        int kbps = (n_kb + splits - 1) / splits;
        splits = (n_kb + kbps - 1) / kbps;
        if (segmode) {
            splits = tab.nseg;
            kbps = 0;
            for (int g = 0; g < tab.nseg; ++g) kbps = std::max(kbps, tab.kb0[g + 1] - tab.kb0[g]);
        }
        const SegTab& gtab = segmode ? tab : no_tab;
        int mode;
        float* partial = nullptr;
        if (segmode) {
            mode = 0;
        } else if (splits > 1 && splits <= 16) {
            // K splits write private partials, summed by one streaming kernel (no atomics)
            const size_t need = (size_t)(splits - 1) * sk_stride * sizeof(float);
            if (e->partial_bytes < need) {
                if (e->partial) HIPCHK(hipFree(e->partial));
                HIPCHK(hipMalloc(&e->partial, need));
                e->partial_bytes = need;
            }
            partial = e->partial;
            mode = acc_now ? 1 : 0;
        } else if (splits > 1) {
            if (!acc_now) HIPCHK(hipMemsetAsync(outp, 0, sk_stride * sizeof(float), sB2));
            mode = 2;
        } else {
            mode = acc_now ? 1 : 0;
        }
        ProfScope ps(e, sB2, 2);
        e->last_g2 = ((g2p_dom && segmode) || (g2p_ok && segmode && gv == 0)) ? (g2p_dom && segmode ? 11 : 10 + g2p_ng) : 1;
#define G2(NTO_, PARTS_, IW_, EX_, GV_) launch_gemm2<NTO_, PARTS_, IW_, EX_, GV_>(e, packed, n_kb, kbps, splits, fill, outp, mode, partial, sB2, gtab)
        // Packed int8 GEMM 2: consecutive segments share a workgroup (one row list in LDS, one pass of the load ring, an
        // epilogue per segment) as long as the list fits the LDS budget (64 KB) and the grid keeps opt_g2_merge (1536 = three resident rounds) workgroups
        // -- 8 bins x 1250 SNPs per block (20 ring steps per segment, a prologue + epilogue each) run as 12 segments per
        // workgroup; long segments and small cohorts stay one per workgroup.
        // Groups of ONE segment and groups of several go to separate launches (plain / MERGE kernel): a long list alone may fill
        // the LDS budget with its rows (the staging buffer of its one epilogue lies over them), a merged group needs rows and
        // staging side by side.
        SegTab ptab1 = tab, ptabm = tab;                      // single-segment groups / merged groups
        int ngrp1 = 0, ngrpm = 0, segmax = 1;
        size_t lds1 = 0, ldsm = 0;                               // dynamic LDS of the two launches: the largest group's need
        const int lds_ng = (g2p_ok && gv == 0) ? g2p_ng : 1;
        if (segmode) {
            const int want = (g2p_merge && g2p_ok && gv == 0) ? std::max(1, (int)((int64_t)tab.nseg * g2p_gx / e->opt_g2_merge)) : 1;   // segments per group
            for (int g = 0; g < tab.nseg;) {
                int kbs = tab.kb0[g + 1] - tab.kb0[g], n = 1;
                // A merged group is addressed relative to the lowest row of ALL its lists (32-bit offsets inside): in a call
                // beyond the 32-bit row span only lists the caller vouched for TOGETHER may share one -- the opt_list_span
                // consecutive lists from a multiple of it (the bins of one jackknife block), never the last bins of one block
                // with the first of the next.
                while (g + n < tab.nseg && n < want && (!wide_rows || tab.bin[g + n] / e->opt_list_span == tab.bin[g] / e->opt_list_span) &&
                       g2p_lds(kbs + tab.kb0[g + n + 1] - tab.kb0[g + n], n + 1, lds_ng) <= 65536) {
                    kbs += tab.kb0[g + n + 1] - tab.kb0[g + n];
                    ++n;
                }
                if (n == 1) {
                    ptab1.grp0[ngrp1] = (unsigned char)g;
                    ptab1.grp1[ngrp1++] = (unsigned char)(g + 1);
                    lds1 = std::max(lds1, g2p_lds(kbs, 1, lds_ng));
                } else {
                    ptabm.grp0[ngrpm] = (unsigned char)g;
                    ptabm.grp1[ngrpm++] = (unsigned char)(g + n);
                    ldsm = std::max(ldsm, g2p_lds(kbs, n, lds_ng));
                    segmax = std::max(segmax, n);
                }
                g += n;
            }
            e->last_g2_merged = segmax;
        }
        if (g2p_dom && segmode) {
            dim3 grid((unsigned)((e->n_pad + 1023) / 1024), ngrp1);      // per-list flow: one segment, never merged
            const size_t lds = lds1;
#define L2D(MISS_, VC_, GV_) gemm2_i8p_kernel<1, MISS_, VC_, GV_><<<grid, 256, lds, sB2>>>(packed, e->pitch, e->rows32, fill, (const int4v*)e->wfrag, e->corr, e->wscale, e->keep, e->n_sketch, e->n_pad, outp, ptab1)
#define L2E(GV_) do { if (e->vc) { if (fill) L2D(true, true, GV_); else L2D(false, true, GV_); } \
                      else { if (fill) L2D(true, false, GV_); else L2D(false, false, GV_); } } while (0)
            if (gv == 0) L2E(0); else L2E(1);
#undef L2E
#undef L2D
        } else if (g2p_ok && segmode && gv == 0) {
            const unsigned gxp = (unsigned)((e->n_pad + 1023) / 1024 * g2p_ng);       // NG = 2: a workgroup covers 512 samples
#define L2P(NG_, MISS_, VC_) gemm2_i8p_kernel<NG_, MISS_, VC_, 0, MERGE_><<<dim3(gxp, NGRP_), 256, LDS_, sB2>>>(packed, e->pitch, e->rows32, fill, (const int4v*)e->wfrag, e->corr, e->wscale, e->keep, e->n_sketch, e->n_pad, outp, PTAB_)
#define L2Q(NG_) do { if (e->vc) { if (fill) L2P(NG_, true, true); else L2P(NG_, false, true); } \
                      else { if (fill) L2P(NG_, true, false); else L2P(NG_, false, false); } } while (0)
#define MERGE_ true
#define NGRP_ ngrpm
#define LDS_ ldsm
#define PTAB_ ptabm
            if (ngrpm) { if (g2p_ng == 1) L2Q(1); else L2Q(2); }
#undef MERGE_
#undef NGRP_
#undef LDS_
#undef PTAB_
#define MERGE_ false
#define NGRP_ ngrp1
#define LDS_ lds1
#define PTAB_ ptab1
            if (ngrp1) { if (g2p_ng == 1) L2Q(1); else L2Q(2); }
#undef MERGE_
#undef NGRP_
#undef LDS_
#undef PTAB_
#undef L2Q
#undef L2P
        } else if (gv == 0) {
            if (e->parts == 2) { if (iw == 2) G2(1, 2, 2, true, 0); else if (e->nto == 1) G2(1, 2, 1, true, 0); else G2(2, 2, 1, true, 0); }
            else if (e->exact) { if (iw == 2) G2(1, 1, 2, true, 0); else if (e->nto == 1) G2(1, 1, 1, true, 0); else G2(2, 1, 1, true, 0); }
            else { if (iw == 2) G2(1, 1, 2, false, 0); else if (e->nto == 1) G2(1, 1, 1, false, 0); else G2(2, 1, 1, false, 0); }
        } else {
            if (e->parts == 2) { if (iw == 2) G2(1, 2, 2, true, 1); else if (e->nto == 1) G2(1, 2, 1, true, 1); else G2(2, 2, 1, true, 1); }
            else if (e->exact) { if (iw == 2) G2(1, 1, 2, true, 1); else if (e->nto == 1) G2(1, 1, 1, true, 1); else G2(2, 1, 1, true, 1); }
            else { if (iw == 2) G2(1, 1, 2, false, 1); else if (e->nto == 1) G2(1, 1, 1, false, 1); else G2(2, 1, 1, false, 1); }
        }
#undef G2
        if (partial) {
            const int64_t n4 = sk_stride / 4;
            add_partials_kernel<<<2048, 256, 0, sB2>>>((float4*)outp, (const float4*)partial, splits - 1, n4);
        }
        return PYRHE_OK;
    };

    // ---- several bins, additive sketch only: GEMM 1 once per chunk of the concatenated lists, per-bin work by segment
    // (a single list takes this flow too when the packed int8 GEMM 2 applies and the sample range alone fills the chip;
    // small cohorts keep the per-list flow, whose GEMM 2 can split K over workgroups)
    // RHE_DOM takes the segmented flow too (round 3) when both genotype views fit one GEMM 1 launch: one GEMM 1 over the
    // concatenated lists of several blocks, then THREE GEMM 2 launches over all segments (additive; dominance under the G view;
    // dominance under the 2 [g == 2] view, accumulating) instead of three per list.
    const bool dom_seg = dom && e->opt_dom_seg && e->i8 && e->nt8 <= 4 && e->n_sketch >= 1 && (nbin > 1 || g2p_dom);
    const bool segflow = dom ? dom_seg : (nbin > 1 || (g2p_ok && e->n_pad >= 192 * 1024));
    const bool g2p_seg = g2p_ok || (g2p_dom && dom_seg);        // segments padded to the packed GEMM 2's 64-entry K blocks
    e->last_flow = segflow ? 1 : 0;
    // rows beyond the 32-bit offsets of one call: only the packed int8 GEMM 2 addresses a list relative to its lowest row
    // (PYRHE_OPT_LIST_SPAN: the caller vouches for every list spanning < 2^32 bytes and < 2^24 rows)
    if (wide_rows && !(segflow && g2p_seg))
        return fail(PYRHE_EINVAL, "block of %lld rows x %lld bytes exceeds the 32-bit row offsets of one call on this kernel path; split it",
                    (long long)n_rows, (long long)e->pitch);
    e->last_g2 = 0;
    if (segflow) {
        std::vector<char> touched(nbin, 0);
        for (int b = 0; b < nbin; ++b)
            if (offsets_host[b + 1] == offsets_host[b] && !accumulate) {
                if (out_sketch) HIPCHK(hipMemsetAsync(out_sketch + b * sk_stride, 0, sk_stride * sizeof(float), sB2));
                if (out_norm && e->n_norm) HIPCHK(hipMemsetAsync(out_norm + (int64_t)b * e->n_norm, 0, e->n_norm * sizeof(double), stream));
                if (e->n_cross) HIPCHK(hipMemsetAsync(out_cross + b * cr_stride, 0, cr_stride * sizeof(double), stream));
                if (out_dom) HIPCHK(hipMemsetAsync(out_dom + b * sk_stride, 0, sk_stride * sizeof(float), sB2));
                if (norm_dom && e->n_norm) HIPCHK(hipMemsetAsync(norm_dom + (int64_t)b * e->n_norm, 0, e->n_norm * sizeof(double), stream));
                if (cross_dom && e->n_cross) HIPCHK(hipMemsetAsync(cross_dom + b * cr_stride, 0, cr_stride * sizeof(double), stream));
            }
        const int64_t total = offsets_host[nbin];
        const int64_t seg_cap = std::min<int64_t>(e->sub_snps, kG2pSegCap);   // GEMM 2 keeps a segment's row list in LDS
        // Plan first: the chunks of the call -- what ONE segment table (kMaxSeg lists) and one pass of the mid kernels / GEMM 2 take.
        struct Chunk { SegTab tab; int64_t chunk0; int n, kb; };
        std::vector<Chunk> chunks;
        int b = 0;
        for (int64_t pos = 0; pos < total;) {
            chunks.emplace_back();
            SegTab& tab = chunks.back().tab;
            tab = SegTab{};
            const int64_t chunk0 = pos;
            int n = 0, kb = 0;
            while (pos < total && n < e->sub_snps && tab.nseg < kMaxSeg) {
                while (offsets_host[b + 1] <= pos) ++b;
                const int take = (int)std::min<int64_t>(std::min<int64_t>(offsets_host[b + 1] - pos, e->sub_snps - n), seg_cap);
                const int g = tab.nseg++;
                tab.start[g] = n;
                tab.kb0[g] = kb;
                tab.bin[g] = b;
                tab.acc[g] = (accumulate || touched[b]) ? 1 : 0;
                touched[b] = 1;
                n += take;
                kb += g2p_seg ? (g2p_merge ? 4 * ((take + 127) / 128) : 2 * ((take + 63) / 64)) : (take + 31) / 32;   // the packed int8 GEMM 2 walks 64-entry K blocks
                pos += take;
                // A list cut here (longer than a segment may be, or than the chunk has room for) continues in the NEXT
                // chunk: its pieces add into the same output slab and norm slot, and only launches are ordered -- the
                // segments of one launch run concurrently.
                if (pos < offsets_host[b + 1]) break;
            }
            tab.start[tab.nseg] = n;
            tab.kb0[tab.nseg] = kb;
            chunks.back().chunk0 = chunk0;
            chunks.back().n = n;
            chunks.back().kb = kb;
        }
        // The FIRST contraction does not need the segment table: consecutive chunks whose entries fit the P buffer (sub_snps) share
        // ONE GEMM 1 launch (a "super-chunk"), the mid kernels and GEMM 2 then go chunk by chunk over their slice of P.  A call of
        // many short lists (8 bins x 8 blocks fill a table at 80 000 rows of N = 500k; 32 bins at 20 000) thereby keeps full-size
        // GEMM 1 launches.  (The private two-stream pipeline keeps one launch per chunk: its chunks alternate work sets.)
        size_t super_end = 0;
        int64_t super0 = 0;
        int set = 0;
        for (size_t ci = 0; ci < chunks.size(); ++ci) {
            const SegTab& tab = chunks[ci].tab;
            const int64_t chunk0 = chunks[ci].chunk0;
            const int n = chunks[ci].n, kb = chunks[ci].kb;
            const int32_t* rows = entries + chunk0;
            if (ci == super_end) {
                int64_t n_super = n;
                super_end = ci + 1;
                if (!piped)
                    while (super_end < chunks.size() && n_super + chunks[super_end].n <= e->sub_snps) n_super += chunks[super_end++].n;
                super0 = chunk0;
                set = (int)(e->sub_counter & 1);
                e->use_set(set);
                if (piped && e->sub_counter >= 2) HIPCHK(hipStreamWaitEvent(e->sA, e->evB[set], 0));
                e->sub_counter++;
                HIPCHK(hipMemsetAsync(e->ws, 0, std::min(e->zero_tail, (size_t)n_super * ncol_p * sizeof(float)), stream));      // P of the super-chunk
                if (dom) {      // both genotype views in one launch (one read)
                    HIPCHK(hipMemsetAsync(e->dom_ws, 0, std::min(e->dom_wmax_off, (size_t)n_super * ncol_p * sizeof(float)), stream));
                    run_gemm1(2, rows, (int)n_super, e->P, e->P2);
                } else {
                    run_gemm1(0, rows, (int)n_super, e->P);
                }
            }
            // this chunk's slice of P (P2), its per-segment maxima / corrections
            float* const Pc = e->P + (chunk0 - super0) * ncol_p;
            float* const P2c = dom ? e->P2 + (chunk0 - super0) * ncol_p : nullptr;
            HIPCHK(hipMemsetAsync(e->ws + e->zero_tail, 0, e->zero_bytes - e->zero_tail, stream));
            if (dom) HIPCHK(hipMemsetAsync(e->dom_ws + e->dom_wmax_off, 0, e->dom_zero_bytes - e->dom_wmax_off, stream));
            {
                ProfScope ps(e, stream, 1);
                mid1_kernel<<<(n + 63) / 64, 256, 0, stream>>>(Pc, pscale, e->i8 ? 1 : 0, n, ncol_p, e->n_sketch, e->n_norm, e->rscale, e->csum,
                                                                 (float)e->n_keep, e->mu, e->Wtmp, e->wmax, e->vsq,
                                                                 out_colsum ? out_colsum + chunk0 : nullptr, e->n_cross, e->vcross, e->nd2, e->nd3, tab);
                if (e->n_cross)
                    cross_reduce_kernel<<<tab.nseg * (e->n_cross * (e->n_cross + 1) / 2), 256, 0, stream>>>(e->vcross, e->n_cross, out_cross, tab);
            }
            // RHE_DOM: additive and dominance sketch from ONE read of the genotypes in the second contraction too (round 3:
            // gemm2_i8p_kernel<2, .., GV = 2> over the six-tile image [Wa | W1 | W2] instead of three launches, the third
            // read-modify-write)
            const bool dom_fuse = dom && g2p_dom && e->opt_dom_fuse && e->n_sketch >= 1;
            if (dom_fuse) {
                if (e->n_norm) norm_reduce_kernel<<<tab.nseg * e->n_norm, 256, 0, stream>>>(e->vsq, e->n_norm, out_norm, tab);
            } else if (e->n_sketch) {
                const int rc = run_gemm2(0, e->Wtmp, e->mu, e->wmax, n, kb, out_sketch, false, set, rows, e->n_norm ? e->vsq : nullptr,
                                         e->n_norm ? out_norm : nullptr, tab, true);
                if (rc) return rc;
            } else {
                if (e->n_norm)
                    norm_reduce_kernel<<<tab.nseg * e->n_norm, 256, 0, stream>>>(e->vsq, e->n_norm, out_norm, tab);
                if (piped) {
                    HIPCHK(hipEventRecord(e->evA[set], e->sA));
                    HIPCHK(hipStreamWaitEvent(sB2, e->evA[set], 0));
                }
            }
            if (dom_fuse) {
                {
                    ProfScope ps(e, stream, 1);
                    mid_dom1_kernel<<<(n + 255) / 256, 256, 0, stream>>>(Pc, P2c, pscale, 1, n, ncol_p, e->n_sketch, e->n_norm, e->rscale,
                                                                         e->csum, (float)e->n_keep, e->cmu, e->W1d, e->Wtmp2, e->wmax1d, e->wmax2,
                                                                         e->vsq2, e->n_cross, e->vcross2, e->nd2, e->nd3, tab);
                    if (e->n_cross)
                        cross_reduce_kernel<<<tab.nseg * (e->n_cross * (e->n_cross + 1) / 2), 256, 0, stream>>>(e->vcross2, e->n_cross, cross_dom, tab);
                    if (e->n_norm) norm_reduce_kernel<<<tab.nseg * e->n_norm, 256, 0, stream>>>(e->vsq2, e->n_norm, norm_dom, tab);
                    const unsigned gridm = (unsigned)(((int64_t)(kb / 2) * 6 * 64 + 255) / 256);
                    mid2_i8p_dom_kernel<<<gridm, 256, 0, stream>>>(e->Wtmp, e->W1d, e->Wtmp2, e->mu, e->cmu, e->n_sketch, e->wmax, e->wmax1d,
                                                                  e->wmax2, (int4v*)e->wfrag, e->corr, e->wscale, rows, e->rows32, tab);
                }
                SegTab ptab = tab;
                size_t lds = 0;
                for (int g = 0; g < tab.nseg; ++g) {
                    ptab.grp0[g] = (unsigned char)g;
                    ptab.grp1[g] = (unsigned char)(g + 1);
                    lds = std::max(lds, g2p_lds(tab.kb0[g + 1] - tab.kb0[g], 1, 2));
                }
                e->last_g2 = 13;
                e->last_g2_merged = 1;
                ProfScope ps(e, sB2, 2);
                dim3 grid((unsigned)((e->n_pad + 1023) / 1024 * 2), tab.nseg);
#define L2F(MISS_, VC_) gemm2_i8p_kernel<2, MISS_, VC_, 2, false><<<grid, 256, lds, sB2>>>(packed, e->pitch, e->rows32, fill, (const int4v*)e->wfrag, e->corr, e->wscale, e->keep, e->n_sketch, e->n_pad, out_sketch, ptab, out_dom)
                if (e->vc) { if (fill) L2F(true, true); else L2F(false, true); }
                else { if (fill) L2F(true, false); else L2F(false, false); }
#undef L2F
            } else if (dom) {
                // dominance (rhe_dom.py:15-41; see mid_dom1_kernel): W1 for the G view, W2 for the 2 [g == 2] view, per segment
                HIPCHK(hipMemsetAsync(e->ws + e->zero_tail, 0, e->zero_bytes - e->zero_tail, stream));      // wmax | corr of every segment
                {
                    ProfScope ps(e, stream, 1);
                    mid_dom1_kernel<<<(n + 255) / 256, 256, 0, stream>>>(Pc, P2c, pscale, 1, n, ncol_p, e->n_sketch, e->n_norm, e->rscale,
                                                                         e->csum, (float)e->n_keep, e->cmu, e->Wtmp, e->Wtmp2, e->wmax, e->wmax2,
                                                                         e->vsq2, e->n_cross, e->vcross2, e->nd2, e->nd3, tab);
                    if (e->n_cross)
                        cross_reduce_kernel<<<tab.nseg * (e->n_cross * (e->n_cross + 1) / 2), 256, 0, stream>>>(e->vcross2, e->n_cross, cross_dom, tab);
                }
                int rc = run_gemm2(0, e->Wtmp, e->cmu, e->wmax, n, kb, out_dom, false, set, rows, e->n_norm ? e->vsq2 : nullptr,
                                   e->n_norm ? norm_dom : nullptr, tab, true);
                if (rc) return rc;
                HIPCHK(hipMemsetAsync(e->corr, 0, (size_t)kMaxSeg * kSegCols * sizeof(double), stream));
                SegTab tab2 = tab;
                for (int g = 0; g < tab2.nseg; ++g) tab2.acc[g] = 1;
                rc = run_gemm2(1, e->Wtmp2, e->zcoef, e->wmax2, n, kb, out_dom, true, set, rows, nullptr, nullptr, tab2, true);
                if (rc) return rc;
            }
            if (piped) HIPCHK(hipEventRecord(e->evB[set], sB2));
        }
    }
    for (int b = 0; b < nbin && !segflow; ++b) {
        const int64_t n_b = offsets_host[b + 1] - offsets_host[b];
        float* outb = out_sketch ? out_sketch + b * sk_stride : nullptr;
        double* normb = out_norm ? out_norm + (int64_t)b * e->n_norm : nullptr;
        float* outd = out_dom ? out_dom + b * sk_stride : nullptr;
        double* normd = norm_dom ? norm_dom + (int64_t)b * e->n_norm : nullptr;
        double* crossb = out_cross ? out_cross + b * cr_stride : nullptr;
        double* crossd = cross_dom ? cross_dom + b * cr_stride : nullptr;
        if (n_b == 0) {
            if (!accumulate) {
                if (outb) HIPCHK(hipMemsetAsync(outb, 0, sk_stride * sizeof(float), sB2));
                if (normb && e->n_norm) HIPCHK(hipMemsetAsync(normb, 0, e->n_norm * sizeof(double), stream));
                if (outd) HIPCHK(hipMemsetAsync(outd, 0, sk_stride * sizeof(float), sB2));
                if (normd && e->n_norm) HIPCHK(hipMemsetAsync(normd, 0, e->n_norm * sizeof(double), stream));
                if (crossb && e->n_cross) HIPCHK(hipMemsetAsync(crossb, 0, cr_stride * sizeof(double), stream));
                if (crossd && e->n_cross) HIPCHK(hipMemsetAsync(crossd, 0, cr_stride * sizeof(double), stream));
            }
            continue;
        }
        bool first = true;
        const int64_t list_step = g2p_dom ? std::min<int64_t>(e->sub_snps, kG2pSegCap) : e->sub_snps;   // packed GEMM 2: row list in LDS
        for (int64_t s0 = 0; s0 < n_b; s0 += list_step) {
            const int n = (int)std::min<int64_t>(list_step, n_b - s0);
            const int32_t* rows = entries + offsets_host[b] + s0;
            const int n32 = (n + kstep - 1) / kstep * kstep;
            const int n_kb = g2p_dom ? 2 * ((n + 63) / 64) : n32 / kstep;      // packed GEMM 2: 64-entry K blocks, counted in 32s
            const int set = (int)(e->sub_counter & 1);
            e->use_set(set);
            if (piped && e->sub_counter >= 2) HIPCHK(hipStreamWaitEvent(e->sA, e->evB[set], 0));   // GEMM 2 of sub-block g-2 is done with this set
            e->sub_counter++;
            const bool acc_now = accumulate || !first;
            SegTab tab = {};
            tab.nseg = 1;
            tab.start[1] = n;
            tab.kb0[1] = n_kb;
            tab.acc[0] = acc_now ? 1 : 0;
            { const int rcz = zero_ws(n); if (rcz) return rcz; }
            // int8 engine: one GEMM 1 launch (one read) for both genotype views; beyond four tiles twice the accumulators spill
            const bool both_views = dom && e->i8 && e->nt8 <= 4;
            if (both_views) {
                HIPCHK(hipMemsetAsync(e->dom_ws, 0, e->dom_zero_bytes, stream));
                run_gemm1(2, rows, n, e->P, e->P2);
            } else {
                run_gemm1(0, rows, n, e->P);
            }
            {
                ProfScope ps(e, stream, 1);
                mid1_kernel<<<(n + 63) / 64, 256, 0, stream>>>(e->P, pscale, e->i8 ? 1 : 0, n, ncol_p, e->n_sketch, e->n_norm, e->rscale, e->csum,
                                                                 (float)e->n_keep, e->mu, e->Wtmp, e->wmax, e->vsq,
                                                                 out_colsum ? out_colsum + offsets_host[b] + s0 : nullptr, e->n_cross, e->vcross, e->nd2, e->nd3, tab);
                if (e->n_norm && !e->n_sketch) norm_reduce_kernel<<<e->n_norm, 256, 0, stream>>>(e->vsq, e->n_norm, normb, tab);
                if (e->n_cross) cross_reduce_kernel<<<e->n_cross * (e->n_cross + 1) / 2, 256, 0, stream>>>(e->vcross, e->n_cross, crossb, tab);
            }
            if (!e->n_sketch && piped) {
                HIPCHK(hipEventRecord(e->evA[set], e->sA));
                HIPCHK(hipStreamWaitEvent(sB2, e->evA[set], 0));
            }
            if (e->n_sketch) {
                const int rc = run_gemm2(0, e->Wtmp, e->mu, e->wmax, n, n_kb, outb, acc_now, set, rows, e->n_norm ? e->vsq : nullptr,
                                         e->n_norm ? normb : nullptr, tab, g2p_dom);
                if (rc) return rc;
            }
            if (dom) {
                // ---- dominance: second GEMM 1 with the 2*[g==2] view, then two GEMM 2 launches into out_dom
                if (!both_views) {
                    HIPCHK(hipMemsetAsync(e->dom_ws, 0, e->dom_zero_bytes, stream));
                    run_gemm1(1, rows, n, e->P2);
                }
                HIPCHK(hipMemsetAsync(e->corr, 0, 16 * kMaxTiles * sizeof(double), stream));
                HIPCHK(hipMemsetAsync(e->wmax, 0, 16 * kMaxTiles * sizeof(uint32_t), stream));
                {
                    ProfScope ps(e, stream, 1);
                    mid_dom1_kernel<<<(n + 255) / 256, 256, 0, stream>>>(e->P, e->P2, pscale, e->i8 ? 1 : 0, n, ncol_p, e->n_sketch, e->n_norm, e->rscale,
                                                                         e->csum, (float)e->n_keep, e->cmu, e->Wtmp, e->Wtmp2, e->wmax, e->wmax2,
                                                                         e->vsq2, e->n_cross, e->vcross2, e->nd2, e->nd3, tab);
                    if (e->n_norm && !e->n_sketch) norm_reduce_kernel<<<e->n_norm, 256, 0, stream>>>(e->vsq2, e->n_norm, normd, tab);
                    if (e->n_cross) cross_reduce_kernel<<<e->n_cross * (e->n_cross + 1) / 2, 256, 0, stream>>>(e->vcross2, e->n_cross, crossd, tab);
                }
                if (e->n_sketch) {
                    int rc = run_gemm2(0, e->Wtmp, e->cmu, e->wmax, n, n_kb, outd, acc_now, set, rows, e->n_norm ? e->vsq2 : nullptr,
                                       e->n_norm ? normd : nullptr, tab, g2p_dom);
                    if (rc) return rc;
                    HIPCHK(hipMemsetAsync(e->corr, 0, 16 * kMaxTiles * sizeof(double), stream));
                    tab.acc[0] = 1;
                    rc = run_gemm2(1, e->Wtmp2, e->zcoef, e->wmax2, n, n_kb, outd, true, set, rows, nullptr, nullptr, tab, g2p_dom);
                    if (rc) return rc;
                }
            }
            if (piped) HIPCHK(hipEventRecord(e->evB[set], sB2));
            first = false;
        }
    }
    HIPCHK(hipGetLastError());
    if (piped) {
        HIPCHK(hipEventRecord(e->evDoneA, e->sA));
        HIPCHK(hipEventRecord(e->evDoneB, sB2));
        e->pending = true;
        if (!e->async_join) {
            HIPCHK(hipStreamWaitEvent(caller, e->evDoneA, 0));
            HIPCHK(hipStreamWaitEvent(caller, e->evDoneB, 0));
            e->pending = false;
        }
    }
    return PYRHE_OK;
}

extern "C" int pyrhe_engine_sketch_block(pyrhe_engine_t* e, const uint8_t* packed, int64_t n_rows,
                                         const int32_t* entries, const int64_t* offsets_host, int nbin,
                                         const float* fill, float* out_sketch, double* out_norm,
                                         int32_t* out_colsum, int accumulate, void* stream_) {
    return sketch_impl(e, packed, n_rows, entries, offsets_host, nbin, fill, out_sketch, out_norm, out_colsum, accumulate,
                       nullptr, nullptr, nullptr, nullptr, stream_);
}

extern "C" int pyrhe_engine_sketch_block_dom(pyrhe_engine_t* e, const uint8_t* packed, int64_t n_rows,
                                             const int32_t* entries, const int64_t* offsets_host, int nbin,
                                             const float* fill, float* out_sketch, double* out_norm,
                                             float* out_sketch_dom, double* out_norm_dom, int accumulate,
                                             void* stream_) {
    if (!out_sketch_dom && !out_norm_dom) return fail(PYRHE_EINVAL, "dominance outputs are NULL");
    return sketch_impl(e, packed, n_rows, entries, offsets_host, nbin, fill, out_sketch, out_norm, nullptr, accumulate,
                       out_sketch_dom, out_norm_dom, nullptr, nullptr, stream_);
}

extern "C" int pyrhe_engine_sketch_block_cross(pyrhe_engine_t* e, const uint8_t* packed, int64_t n_rows,
                                               const int32_t* entries, const int64_t* offsets_host, int nbin,
                                               const float* fill, float* out_sketch, double* out_norm, double* out_cross,
                                               float* out_sketch_dom, double* out_norm_dom, double* out_cross_dom,
                                               int accumulate, void* stream_) {
    if (e && !e->n_cross) return fail(PYRHE_ESTATE, "no PYRHE_COL_CROSS columns installed (pyrhe_engine_set_rhs)");
    if (!out_cross) return fail(PYRHE_EINVAL, "out_cross is NULL");
    return sketch_impl(e, packed, n_rows, entries, offsets_host, nbin, fill, out_sketch, out_norm, nullptr, accumulate,
                       out_sketch_dom, out_norm_dom, out_cross, out_cross_dom, stream_);
}

extern "C" int pyrhe_engine_set_async(pyrhe_engine_t* e, int enable) {
    if (!e) return fail(PYRHE_EINVAL, "NULL engine");
    e->async_join = enable != 0;
    return PYRHE_OK;
}

extern "C" int pyrhe_engine_set_option(pyrhe_engine_t* e, int option, int64_t value) {
    if (!e) return fail(PYRHE_EINVAL, "NULL engine");
    switch (option) {
        case PYRHE_OPT_G1_WORKGROUPS:
            if (value < 1 || value > (1 << 20)) return fail(PYRHE_EINVAL, "G1_WORKGROUPS %lld outside 1..2^20", (long long)value);
            e->opt_g1_wgs = (int)value; break;
        case PYRHE_OPT_G2_PACKED: e->opt_g2_packed = value != 0; break;
        case PYRHE_OPT_G2_IW:
            if (value < 0 || value > 2) return fail(PYRHE_EINVAL, "G2_IW takes 0 (automatic), 1 or 2");
            e->opt_g2_iw = (int)value; break;
        case PYRHE_OPT_G2_SPLITS:
            if (value < 0 || value > 4096) return fail(PYRHE_EINVAL, "G2_SPLITS %lld outside 0..4096", (long long)value);
            e->opt_g2_splits = (int)value; break;
        case PYRHE_OPT_DOTS_GENERIC: e->opt_dots_generic = value != 0; break;
        case PYRHE_OPT_G1_SPLIT_MAJOR:
            if (value < 0 || value > 2) return fail(PYRHE_EINVAL, "G1_SPLIT_MAJOR takes 0 (row group fast), 1 (sample split fast) or 2 (sample ranges per XCD)");
            e->opt_g1_split_major = (int)value; break;
        case PYRHE_OPT_G1_WIDE:
            if (value < 0 || value > 2) return fail(PYRHE_EINVAL, "G1_WIDE takes 0 (never), 1 (automatic) or 2 (whenever possible)");
            e->opt_g1_wide = (int)value; break;
        case PYRHE_OPT_G1_ZERO_ROWS: e->opt_g1_zero_rows = value != 0; break;
        case PYRHE_OPT_G1_FILL: e->opt_g1_fill = value != 0; break;
        case PYRHE_OPT_G1_LINE: e->opt_g1_line = value != 0; break;
        case PYRHE_OPT_DOM_SEG: e->opt_dom_seg = value != 0; break;
        case PYRHE_OPT_DOM_FUSE: e->opt_dom_fuse = value != 0; break;
        case PYRHE_OPT_G1_COMPACT: e->opt_g1_compact = value != 0; break;
        case PYRHE_OPT_DOTS_TILES:
            if (value != 1 && value != 2 && value != 4) return fail(PYRHE_EINVAL, "DOTS_TILES takes 1, 2 or 4");
            e->opt_dots_tiles = (int)value; break;
        case PYRHE_OPT_DOTS_F32:
            if (value < 0 || value > 2) return fail(PYRHE_EINVAL, "DOTS_F32 takes 0 (float64), 1 (float32 products) or 2 (automatic)");
            e->opt_dots_f32 = (int)value; break;
        case PYRHE_OPT_LIST_SPAN:
            if (value < 0 || value > (1 << 20)) return fail(PYRHE_EINVAL, "LIST_SPAN %lld outside 0..2^20", (long long)value);
            e->opt_list_span = (int)value; break;
        case PYRHE_OPT_G2_MERGE:
            if (value < 0 || value > (1 << 20)) return fail(PYRHE_EINVAL, "G2_MERGE %lld outside 0..2^20", (long long)value);
            e->opt_g2_merge = (int)value; break;
#ifdef PYRHE_DIAG
        case 100: e->diag_g1_plain_store = value != 0; break;
        case 101: e->diag_g1_no_barrier = (int)value; break;
        case 102: e->diag_g1_stats = reinterpret_cast<long long*>(value); break;
#endif
        default: return fail(PYRHE_EINVAL, "unknown option %d", option);
    }
    return PYRHE_OK;
}

extern "C" int pyrhe_engine_set_pipeline(pyrhe_engine_t* e, int enable) {
    if (!e) return fail(PYRHE_EINVAL, "NULL engine");
    e->pipeline = enable != 0;
    return PYRHE_OK;
}

extern "C" int pyrhe_engine_join(pyrhe_engine_t* e, void* stream_) {
    if (!e) return fail(PYRHE_EINVAL, "NULL engine");
    if (e->pending) {
        HIPCHK(hipSetDevice(e->device));
        HIPCHK(hipStreamWaitEvent((hipStream_t)stream_, e->evDoneA, 0));
        HIPCHK(hipStreamWaitEvent((hipStream_t)stream_, e->evDoneB, 0));
        e->pending = false;
    }
    return PYRHE_OK;
}


extern "C" int pyrhe_engine_block_stats(pyrhe_engine_t* e, const uint8_t* packed, int64_t n_rows,
                                        int32_t* out_sum, int32_t* out_nmiss, void* stream_) {
    if (!e) return fail(PYRHE_EINVAL, "NULL engine");
    if (n_rows <= 0) return PYRHE_OK;            // a rank without local blocks: nothing to do, whatever the pointers are
    if (!packed || !out_sum || !out_nmiss) return fail(PYRHE_EINVAL, "NULL argument");
    if (!e->rhs_set) return fail(PYRHE_ESTATE, "pyrhe_engine_set_rhs has not been called (keep mask unknown)");
    HIPCHK(hipSetDevice(e->device));
    if (e->vc) block_stats_kernel<true><<<(unsigned)((n_rows + 3) / 4), 256, 0, (hipStream_t)stream_>>>(packed, e->pitch, n_rows, e->keepmask, out_sum, out_nmiss);
    else block_stats_kernel<false><<<(unsigned)((n_rows + 3) / 4), 256, 0, (hipStream_t)stream_>>>(packed, e->pitch, n_rows, e->keepmask, out_sum, out_nmiss);
    HIPCHK(hipGetLastError());
    return PYRHE_OK;
}

extern "C" int pyrhe_sum_blocks(const float* blocks, int n_blocks, int64_t block_stride, int64_t n, double* out,
                                void* stream_) {
    if (!blocks || !out) return fail(PYRHE_EINVAL, "NULL argument");
    if (n <= 0 || n % 4 || block_stride % 4) return fail(PYRHE_EINVAL, "n and block_stride must be positive multiples of 4");
    if (n_blocks < 0) return fail(PYRHE_EINVAL, "n_blocks negative");
    sum_blocks_kernel<<<4096, 256, 0, (hipStream_t)stream_>>>(blocks, n_blocks, block_stride, n / 4, out, nullptr);
    HIPCHK(hipGetLastError());
    return PYRHE_OK;
}

extern "C" int pyrhe_sum_blocks_weighted(const float* blocks, int n_blocks, int64_t block_stride, int64_t n,
                                         const double* weights, double* out, void* stream_) {
    if (!blocks || !out || !weights) return fail(PYRHE_EINVAL, "NULL argument");
    if (n <= 0 || n % 4 || block_stride % 4) return fail(PYRHE_EINVAL, "n and block_stride must be positive multiples of 4");
    if (n_blocks < 0) return fail(PYRHE_EINVAL, "n_blocks negative");
    sum_blocks_kernel<<<4096, 256, 0, (hipStream_t)stream_>>>(blocks, n_blocks, block_stride, n / 4, out, weights);
    HIPCHK(hipGetLastError());
    return PYRHE_OK;
}

extern "C" int pyrhe_sum_blocks_both(const float* blocks, int n_blocks, int64_t block_stride, int64_t n,
                                     const double* weights, double* out, double* out_weighted, void* stream_) {
    if (!blocks || !out || !weights || !out_weighted) return fail(PYRHE_EINVAL, "NULL argument");
    if (n <= 0 || n % 4 || block_stride % 4) return fail(PYRHE_EINVAL, "n and block_stride must be positive multiples of 4");
    if (n_blocks < 0) return fail(PYRHE_EINVAL, "n_blocks negative");
    sum_blocks_both_kernel<<<4096, 256, 0, (hipStream_t)stream_>>>(blocks, n_blocks, block_stride, n / 4, out, weights, out_weighted);
    HIPCHK(hipGetLastError());
    return PYRHE_OK;
}

extern "C" int pyrhe_engine_loo_dots(pyrhe_engine_t* e, const pyrhe_row_t* x, int nx, const pyrhe_row_t* y, int ny,
                                     int64_t period, int n_sub, int n_batch, double* out, void* stream_) {
    if (!e || !x || !y || !out) return fail(PYRHE_EINVAL, "NULL argument");
    if (nx < 1 || ny < 1) return fail(PYRHE_EINVAL, "nx and ny must be >= 1");
    if (period <= 0 || period % 2) return fail(PYRHE_EINVAL, "period (%lld) must be positive and even", (long long)period);
    if (n_sub < 1 || n_sub > 65535 || n_batch < 1 || n_batch > 65535) return fail(PYRHE_EINVAL, "n_sub / n_batch outside 1..65535");
    for (int pass = 0; pass < 2; ++pass) {
        const pyrhe_row_t* r = pass ? y : x;
        for (int a = 0; a < (pass ? ny : nx); ++a) {
            if (!r[a].tot && !r[a].blk) return fail(PYRHE_EINVAL, "%c row %d has neither tot nor blk", pass ? 'y' : 'x', a);
            if (r[a].mul && !r[a].blk) return fail(PYRHE_EINVAL, "%c row %d: mul without blk", pass ? 'y' : 'x', a);
            if (((uintptr_t)r[a].tot & 15) || ((uintptr_t)r[a].blk & 7) || ((uintptr_t)r[a].mul & 7) || (r[a].blk_batch_stride & 1))
                return fail(PYRHE_EINVAL, "%c row %d is not 16-byte (tot) / 8-byte (blk, mul, stride) aligned", pass ? 'y' : 'x', a);
        }
    }
    hipStream_t stream = (hipStream_t)stream_;
    HIPCHK(hipSetDevice(e->device));
    auto ensure_partial = [&](size_t need) -> int {
        if (e->dots_partial_bytes < need) {
            if (e->dots_partial) {
                HIPCHK(hipDeviceSynchronize());       // earlier launches may still read the old scratch
                HIPCHK(hipFree(e->dots_partial));
            }
            HIPCHK(hipMalloc(&e->dots_partial, need));
            e->dots_partial_bytes = need;
        }
        return PYRHE_OK;
    };
    // ---- block rows (float32) against fixed rows (float64): the matrix-core path
    bool fast = period % 32 == 0 && (int64_t)nx * n_batch < (1 << 30);
    for (int a = 0; a < nx && fast; ++a) fast = !x[a].tot && x[a].blk && (((uintptr_t)x[a].blk | (uintptr_t)x[a].mul) & 15) == 0 && x[a].blk_batch_stride % 4 == 0;
    for (int b = 0; b < ny && fast; ++b) fast = y[b].tot && !y[b].blk;
    // ---- fixed rows against fixed rows (whole-data sketches, covariates, phenotypes), one batch element
    bool fixed = period % 32 == 0 && n_batch == 1;
    for (int a = 0; a < nx && fixed; ++a) fixed = x[a].tot && !x[a].blk;
    for (int b = 0; b < ny && fixed; ++b) fixed = y[b].tot && !y[b].blk;
    const bool no_fast = e->opt_dots_generic != 0;
    // ---- block rows against themselves (x and y are the same block-only rows): the self-Gram path
    bool self = period % 32 == 0 && nx == ny && nx <= 16;
    for (int a = 0; a < nx && self; ++a)
        self = !x[a].tot && x[a].blk && x[a].blk == y[a].blk && x[a].mul == y[a].mul && !y[a].tot &&
               x[a].blk_batch_stride == y[a].blk_batch_stride && (((uintptr_t)x[a].blk | (uintptr_t)x[a].mul) & 15) == 0 &&
               x[a].blk_batch_stride % 4 == 0;
    auto table = [&](int a0, int n) {
        BlkRowTable t;
        for (int a = 0; a < kDescMax; ++a)
            t.r[a] = a >= n ? BlkRowDev{nullptr, nullptr, 0}
                     : fixed ? BlkRowDev{reinterpret_cast<const float*>(x[a0 + a].tot), nullptr, 0}
                             : BlkRowDev{x[a0 + a].blk, x[a0 + a].mul, (long long)x[a0 + a].blk_batch_stride};
        return t;
    };
    if (self && !no_fast) {
        const BlkRowTable tab = table(0, nx);
        const int zt = 16 / nx, tiles = (n_batch + zt - 1) / zt, tgs = (tiles + 3) / 4;
        int64_t n_chunks = std::max<int64_t>(1, 4096 / ((int64_t)tgs * n_sub));
        n_chunks = std::min<int64_t>(n_chunks, std::max<int64_t>(1, period / 256));
        const int chunk = (int)(((period + n_chunks - 1) / n_chunks + 31) / 32 * 32);
        n_chunks = (period + chunk - 1) / chunk;
        const int n_part = (int)n_chunks * n_sub;
        const int rc = ensure_partial((size_t)tgs * n_part * 1024 * sizeof(double));
        if (rc) return rc;
        dim3 grid((unsigned)((n_chunks + 3) / 4), n_sub, tgs);
        blk_self_gram_kernel<<<grid, 256, 0, stream>>>(tab, nx, n_batch, period, chunk, (int)n_chunks, e->dots_partial);
        const int64_t n_out = (int64_t)n_batch * nx * nx;
        int g_log2 = 0;
        while (g_log2 < 6 && (n_part >> g_log2) > 8 && (n_out << g_log2) < (1 << 18)) ++g_log2;
        blk_self_gram_reduce_kernel<<<(unsigned)(((n_out << g_log2) + 255) / 256), 256, 0, stream>>>(e->dots_partial, n_part, nx, n_batch, g_log2, out);
        HIPCHK(hipGetLastError());
        return PYRHE_OK;
    }
    if ((fast || fixed) && !no_fast) {
        for (int a0 = 0; a0 < nx; a0 += kDescMax) {
            const int nxc = std::min(kDescMax, nx - a0);
            const BlkRowTable tab = table(a0, nxc);
            const int n_rows = nxc * n_batch;
            const int rgs = (n_rows + 63) / 64;
            int64_t n_chunks = std::max<int64_t>(1, 4096 / ((int64_t)rgs * n_sub));
            n_chunks = std::min<int64_t>(n_chunks, std::max<int64_t>(1, period / 256));
            const int chunk = (int)(((period + n_chunks - 1) / n_chunks + 31) / 32 * 32);
            n_chunks = (period + chunk - 1) / chunk;
            const int n_part = (int)n_chunks * n_sub;
            const int rc = ensure_partial((size_t)rgs * n_part * 1024 * sizeof(double));
            if (rc) return rc;
            for (int b0 = 0; b0 < ny; b0 += 16) {
                const int nyt = std::min(16, ny - b0);
                FixedRows Y;
                for (int c = 0; c < 16; ++c) Y.p[c] = c < nyt ? y[b0 + c].tot : nullptr;
                dim3 grid((unsigned)((n_chunks + 3) / 4), n_sub, rgs);
                if (fixed) blk_fixed_dots_kernel<true><<<grid, 256, 0, stream>>>(tab, nxc, n_rows, Y, nyt, period, chunk, (int)n_chunks, e->dots_partial);
                else blk_fixed_dots_kernel<false><<<grid, 256, 0, stream>>>(tab, nxc, n_rows, Y, nyt, period, chunk, (int)n_chunks, e->dots_partial);
                int g_log2 = 0;                       // lanes per output: few rows with many partials share the work
                while (g_log2 < 6 && (n_part >> g_log2) > 8 && ((int64_t)n_rows * 16 << g_log2) < (1 << 18)) ++g_log2;
                blk_fixed_reduce_kernel<<<(unsigned)((((int64_t)n_rows * 16 << g_log2) + 255) / 256), 256, 0, stream>>>(
                    e->dots_partial, n_part, n_rows, nxc, a0, nx, nyt, b0, ny, fixed ? 1.0 : -1.0, g_log2, out);
            }
        }
        HIPCHK(hipGetLastError());
        return PYRHE_OK;
    }
    // enough workgroups to fill the chip, each with >= 4 iterations where the data allows
    const int64_t want = (2048 + (int64_t)n_sub * n_batch - 1) / ((int64_t)n_sub * n_batch);
    const int gx = (int)std::max<int64_t>(1, std::min<int64_t>(std::max<int64_t>(want, 4), (period + 2047) / 2048));
    const int n_part = gx * n_sub;
    {
        const int rc = ensure_partial((size_t)n_part * n_batch * 64 * sizeof(double));
        if (rc) return rc;
    }
    auto tile = [](const pyrhe_row_t* r, int n0, int n, DotRows& d) {
        for (int a = 0; a < 8; ++a) {
            const bool on = a < n;
            d.tot[a] = on ? r[n0 + a].tot : nullptr;
            d.blk[a] = on ? r[n0 + a].blk : nullptr;
            d.mul[a] = on ? r[n0 + a].mul : nullptr;
            d.stride[a] = on ? r[n0 + a].blk_batch_stride : 0;
        }
    };
    for (int a0 = 0; a0 < nx; a0 += 8)
        for (int b0 = 0; b0 < ny; b0 += 8) {
            const int nxt = std::min(8, nx - a0), nyt = std::min(8, ny - b0);
            DotRows X, Y;
            tile(x, a0, nxt, X);
            tile(y, b0, nyt, Y);
            dim3 grid(gx, n_sub, n_batch);
            loo_dots_kernel<<<grid, 256, 0, stream>>>(X, nxt, Y, nyt, period, e->dots_partial);
            loo_dots_reduce_kernel<<<(n_batch * 64 + 255) / 256, 256, 0, stream>>>(e->dots_partial, n_part, n_batch, nxt, nyt, a0, b0, nx, ny, out);
        }
    HIPCHK(hipGetLastError());
    return PYRHE_OK;
}

extern "C" int pyrhe_engine_loo_dots_multi(pyrhe_engine_t* e, const pyrhe_row_t* x, int nx, const pyrhe_row_t* y_sub, int ny_sub,
                                           const pyrhe_row_t* y_flat, int ny_flat, int want_self, int64_t period, int n_sub,
                                           int n_batch, double* out_sub, double* out_flat, double* out_self, void* stream_) {
    if (!e || !x || !y_sub || !out_sub) return fail(PYRHE_EINVAL, "NULL argument");
    if (nx < 1 || nx > 16 || ny_sub < 1 || ny_sub > 16 || ny_flat < 0 || ny_flat > 16)
        return fail(PYRHE_EINVAL, "loo_dots_multi takes 1..16 block rows, 1..16 sub-strided and 0..16 flat fixed rows");
    if ((ny_flat > 0) != (y_flat != nullptr && out_flat != nullptr)) return fail(PYRHE_EINVAL, "flat rows and out_flat go together");
    if ((want_self != 0) != (out_self != nullptr)) return fail(PYRHE_EINVAL, "want_self and out_self go together");
    if (period <= 0 || period % 32) return fail(PYRHE_EINVAL, "period (%lld) must be a positive multiple of 32", (long long)period);
    if (n_sub < 1 || n_sub > 65535 || n_batch < 1 || (int64_t)n_batch * 16 >= (1 << 30)) return fail(PYRHE_EINVAL, "n_sub / n_batch out of range");
    for (int a = 0; a < nx; ++a)
        if (x[a].tot || !x[a].blk || (((uintptr_t)x[a].blk | (uintptr_t)x[a].mul) & 15) || x[a].blk_batch_stride % 4)
            return fail(PYRHE_EINVAL, "x row %d must be a 16-byte aligned block-only row", a);
    for (int pass = 0; pass < 2; ++pass) {
        const pyrhe_row_t* r = pass ? y_flat : y_sub;
        for (int b = 0; b < (pass ? ny_flat : ny_sub); ++b)
            if (!r[b].tot || r[b].blk || ((uintptr_t)r[b].tot & 15)) return fail(PYRHE_EINVAL, "fixed row %d must be a 16-byte aligned tot-only row", b);
    }
    hipStream_t stream = (hipStream_t)stream_;
    HIPCHK(hipSetDevice(e->device));
    int nxp = 1;
    while (nxp < nx) nxp *= 2;
    const int64_t n_rows = (int64_t)n_batch * nxp;
    const int rgs = (int)((n_rows + 63) / 64);
    int64_t n_chunks = std::max<int64_t>(1, 4096 / ((int64_t)rgs * n_sub));
    n_chunks = std::min<int64_t>(n_chunks, std::max<int64_t>(1, period / 256));
    const int chunk = (int)(((period + n_chunks - 1) / n_chunks + 31) / 32 * 32);
    n_chunks = (period + chunk - 1) / chunk;
    const size_t one = (size_t)rgs * n_sub * n_chunks * 1024 * sizeof(double);
    const int n_sets = 1 + (ny_flat > 0) + (want_self != 0);
    if (e->dots_partial_bytes < one * n_sets) {
        if (e->dots_partial) {
            HIPCHK(hipDeviceSynchronize());
            HIPCHK(hipFree(e->dots_partial));
        }
        HIPCHK(hipMalloc(&e->dots_partial, one * n_sets));
        e->dots_partial_bytes = one * n_sets;
    }
    double* ps = e->dots_partial;
    double* pf = ny_flat > 0 ? ps + one / sizeof(double) : nullptr;
    double* pg = want_self ? ps + (size_t)(1 + (ny_flat > 0)) * (one / sizeof(double)) : nullptr;
    BlkRowTable tab;
    for (int a = 0; a < kDescMax; ++a)
        tab.r[a] = a < nx ? BlkRowDev{x[a].blk, x[a].mul, (long long)x[a].blk_batch_stride} : BlkRowDev{nullptr, nullptr, 0};
    FixedRows Ys, Yf;
    for (int c = 0; c < 16; ++c) {
        Ys.p[c] = c < ny_sub ? y_sub[c].tot : nullptr;
        Yf.p[c] = c < ny_flat ? y_flat[c].tot : nullptr;
    }
    const bool dots_f32 = e->opt_dots_f32 == 1 || (e->opt_dots_f32 == 2 && period >= 65536);
    if (dots_f32) {
        // the float32 form reads float32 COPIES of the fixed rows (made here, once per call: the rows change between calls)
        FixedRows32Src src;
        int64_t total = 0, longest = 0;
        for (int c = 0; c < ny_sub + ny_flat; ++c) {
            const int64_t len = c < ny_sub ? (int64_t)n_sub * period : period;
            src.p[c] = c < ny_sub ? y_sub[c].tot : y_flat[c - ny_sub].tot;
            src.len[c] = len;
            src.off[c] = total;
            total += len;
            longest = std::max(longest, len);
        }
        if (e->dots_y32_bytes < (size_t)total * sizeof(float)) {
            // sized ONCE for every row count this entry point takes at these (n_sub, period) -- 16 sub-strided + 16 flat rows --,
            // so the later calls of a pass 2 (other row counts, same shapes) never free a buffer that is in use: the
            // synchronising path below is left to a caller that comes back with longer rows
            const size_t cap = std::max((size_t)total, (size_t)16 * ((size_t)n_sub + 1) * (size_t)period) * sizeof(float);
            if (e->dots_y32) {
                HIPCHK(hipDeviceSynchronize());
                HIPCHK(hipFree(e->dots_y32));
                e->dots_y32 = nullptr;
                e->dots_y32_bytes = 0;
            }
            HIPCHK(hipMalloc(&e->dots_y32, cap));
            e->dots_y32_bytes = cap;
        }
        fixed_rows_f32_kernel<<<dim3((unsigned)std::min<int64_t>(1024, (longest / 4 + 255) / 256), (unsigned)(ny_sub + ny_flat)), 256, 0, stream>>>(src, e->dots_y32);
        for (int c = 0; c < ny_sub; ++c) Ys.p[c] = reinterpret_cast<const double*>(e->dots_y32 + src.off[c]);
        for (int c = 0; c < ny_flat; ++c) Yf.p[c] = reinterpret_cast<const double*>(e->dots_y32 + src.off[ny_sub + c]);
    }
    // two tiles (32 block rows) per wave, two waves per SIMD (PYRHE_OPT_DOTS_TILES = 2, the default) or four tiles per wave
    const int tpw = e->opt_dots_tiles, cpw = tpw;      // tiles per wave = chunks per workgroup (1, 2 or 4)
    const int64_t ncgs = ((n_chunks + cpw - 1) / cpw) * (int64_t)n_sub;      // (chunk group, sub-row) pieces of the fixed rows
    dim3 grid((unsigned)((ncgs + 7) / 8 * 8 * rgs));
    // float32 matrix instruction with 128-element float32 partial sums (PYRHE_OPT_DOTS_F32): automatic from 65 536 samples on,
    // where a row holds >= 512 such partial sums and the float64 sum over them averages their rounding (measured at N = 500k:
    // sigma^2 unchanged to 16 digits, jackknife SEs 5e-9; tools/ab.sh [dots_f32]); small cohorts, which gain nothing, stay float64
#define LMK(F_, S_, X_, T_) blk_multi_dots_kernel<F_, S_, X_, T_><<<grid, 256, 0, stream>>>(tab, nx, nxp, n_batch, Ys, ny_sub, Yf, ny_flat, period, chunk, (int)n_chunks, n_sub, rgs, ps, pf, pg)
#define LM(F_, S_) do { if (dots_f32) { if (tpw == 1) LMK(F_, S_, true, 1); else if (tpw == 2) LMK(F_, S_, true, 2); else LMK(F_, S_, true, 4); } \
                        else { if (tpw == 1) LMK(F_, S_, false, 1); else if (tpw == 2) LMK(F_, S_, false, 2); else LMK(F_, S_, false, 4); } } while (0)
    if (ny_flat > 0) { if (want_self) LM(true, true); else LM(true, false); }
    else { if (want_self) LM(false, true); else LM(false, false); }
#undef LM
#undef LMK
    auto reduce = [&](const double* part, int kind, int ny, double* out) {
        const int64_t n_out = (int64_t)n_batch * nx * (kind == 1 ? n_sub : 1) * ny;
        const int n_part = (int)(kind == 1 ? n_chunks : n_chunks * n_sub);
        int g_log2 = 0;
        while (g_log2 < 6 && (n_part >> g_log2) > 8 && (n_out << g_log2) < (1 << 18)) ++g_log2;
        blk_multi_reduce_kernel<<<(unsigned)(((n_out << g_log2) + 255) / 256), 256, 0, stream>>>(part, kind, (int)n_chunks, n_sub, nx, nxp, n_batch,
                                                                                               ny, g_log2, out);
    };
    reduce(ps, 0, ny_sub, out_sub);
    if (ny_flat > 0) reduce(pf, 1, ny_flat, out_flat);
    if (want_self) reduce(pg, 2, nx, out_self);
    HIPCHK(hipGetLastError());
    return PYRHE_OK;
}

extern "C" int pyrhe_engine_apply_projector(pyrhe_engine_t* e, const double* x, int64_t n_rows, int64_t n, const float* Wt,
                                            const float* Q, int c, double* out, void* stream_) {
    if (!e) return fail(PYRHE_EINVAL, "NULL engine");
    if (n_rows <= 0) return PYRHE_OK;
    if (!x || !Wt || !Q || !out) return fail(PYRHE_EINVAL, "NULL argument");
    if (c < 1 || c > kProjMaxCov) return fail(PYRHE_EINVAL, "%d covariates outside 1..%d", c, kProjMaxCov);
    if (n <= 0 || n_rows > 65535) return fail(PYRHE_EINVAL, "n must be positive and n_rows <= 65535");
    hipStream_t stream = (hipStream_t)stream_;
    HIPCHK(hipSetDevice(e->device));
    const int n_chunks = (int)((n + kProjChunk - 1) / kProjChunk);
    const size_t need = ((size_t)n_rows * n_chunks * c) * sizeof(double) + (size_t)n_rows * c * sizeof(float);
    if (e->dots_partial_bytes < need) {
        if (e->dots_partial) {
            HIPCHK(hipDeviceSynchronize());       // earlier launches may still read the old scratch
            HIPCHK(hipFree(e->dots_partial));
            e->dots_partial = nullptr;
            e->dots_partial_bytes = 0;
        }
        HIPCHK(hipMalloc(&e->dots_partial, need));
        e->dots_partial_bytes = need;
    }
    double* partial = e->dots_partial;
    float* u = reinterpret_cast<float*>(partial + (size_t)n_rows * n_chunks * c);
    proj_dots_kernel<<<dim3(n_chunks, (unsigned)n_rows), 256, 0, stream>>>(x, n, Wt, c, n_chunks, partial);
    proj_coef_kernel<<<(unsigned)n_rows, kProjMaxCov, 0, stream>>>(partial, n_chunks, Q, c, u);
    proj_expand_kernel<<<dim3((unsigned)((n + 255) / 256), (unsigned)n_rows), 256, 0, stream>>>(u, Wt, c, n, out);
    HIPCHK(hipGetLastError());
    return PYRHE_OK;
}

extern "C" int pyrhe_recode_rows(uint8_t* packed, int64_t n_rows, int64_t pitch, int to_coding, void* stream_) {
    if (pitch <= 0 || pitch % 4) return fail(PYRHE_EINVAL, "pitch must be a positive multiple of 4");
    if (to_coding != PYRHE_CODING_PLINK && to_coding != PYRHE_CODING_VALUE) return fail(PYRHE_EINVAL, "unknown coding %d", to_coding);
    if (n_rows <= 0) return PYRHE_OK;            // an empty buffer (a rank without local blocks) has no address
    if (!packed) return fail(PYRHE_EINVAL, "NULL argument");
    recode_rows_kernel<<<8192, 256, 0, (hipStream_t)stream_>>>(reinterpret_cast<uint32_t*>(packed), n_rows * (pitch / 4), to_coding == PYRHE_CODING_VALUE);
    HIPCHK(hipGetLastError());
    return PYRHE_OK;
}

extern "C" int pyrhe_engine_set_coding(pyrhe_engine_t* e, int coding) {
    if (!e) return fail(PYRHE_EINVAL, "NULL engine");
    if (coding != PYRHE_CODING_PLINK && coding != PYRHE_CODING_VALUE) return fail(PYRHE_EINVAL, "unknown coding %d", coding);
    e->vc = coding == PYRHE_CODING_VALUE;
    return PYRHE_OK;
}

extern "C" int pyrhe_decode_rows(const uint8_t* packed, int64_t n_rows, int64_t pitch, float* out, void* stream_) {
    if (!packed || !out) return fail(PYRHE_EINVAL, "NULL argument");
    if (n_rows <= 0) return PYRHE_OK;
    const int64_t threads = n_rows * pitch;
    decode_rows_kernel<<<(unsigned)((threads + 255) / 256), 256, 0, (hipStream_t)stream_>>>(packed, n_rows, pitch, out);
    HIPCHK(hipGetLastError());
    return PYRHE_OK;
}

extern "C" int pyrhe_synth_genotypes(uint8_t* packed, int64_t n_rows, int64_t pitch, int64_t n_samples,
                                     uint64_t seed, int64_t row0, void* stream_) {
    if (!packed) return fail(PYRHE_EINVAL, "NULL argument");
    if (pitch <= 0 || pitch % 4) return fail(PYRHE_EINVAL, "pitch must be a positive multiple of 4");
    if (n_samples < 0 || n_samples > 4 * pitch) return fail(PYRHE_EINVAL, "n_samples outside 0..4*pitch");
    if (n_rows <= 0) return PYRHE_OK;
    const int64_t threads = n_rows * (pitch / 4);
    if ((threads + 255) / 256 > 0x7fffffffll) return fail(PYRHE_EINVAL, "too many rows for one launch; call per block");
    synth_kernel<<<(unsigned)((threads + 255) / 256), 256, 0, (hipStream_t)stream_>>>(packed, n_rows, pitch, n_samples, seed, row0);
    HIPCHK(hipGetLastError());
    return PYRHE_OK;
}

extern "C" int pyrhe_stream_read(const void* buf, int64_t bytes, uint32_t* sink, void* stream_) {
    if (!buf || !sink) return fail(PYRHE_EINVAL, "NULL argument");
    if (bytes <= 0 || bytes % 16 || ((uintptr_t)buf & 15)) return fail(PYRHE_EINVAL, "buffer must be 16-byte aligned and a multiple of 16 bytes");
    stream_read_kernel<<<4096, 256, 0, (hipStream_t)stream_>>>(reinterpret_cast<const uint4*>(buf), bytes / 16, sink);
    HIPCHK(hipGetLastError());
    return PYRHE_OK;
}

extern "C" int pyrhe_engine_profile(pyrhe_engine_t* e, int enable) {
    if (!e) return fail(PYRHE_EINVAL, "NULL engine");
    e->evs.clear();
    e->pool_used = 0;
    e->prof_failed = false;
    e->profile = enable != 0;
    return PYRHE_OK;
}

extern "C" int pyrhe_engine_profile_read(pyrhe_engine_t* e, double* ms_gemm1, double* ms_mid, double* ms_gemm2,
                                         int64_t* launches) {
    if (!e) return fail(PYRHE_EINVAL, "NULL engine");
    if (e->prof_failed) return fail(PYRHE_ENOMEM, "hipEventCreate failed while profiling: the timings of this pass are incomplete");
    double t[3] = {0, 0, 0};
    int64_t n = 0;
    for (auto& ev : e->evs) {
        HIPCHK(hipEventSynchronize(ev.b));
        float ms = 0;
        HIPCHK(hipEventElapsedTime(&ms, ev.a, ev.b));
        t[ev.kind] += ms;
        if (ev.kind == 0) ++n;
    }
    if (ms_gemm1) *ms_gemm1 = t[0];
    if (ms_mid) *ms_mid = t[1];
    if (ms_gemm2) *ms_gemm2 = t[2];
    if (launches) *launches = n;
    return PYRHE_OK;
}

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
// G is decoded in registers (0/1/2 are exact in int8 and fp16).  Default precision (i8x3): GEMM 1 on
// v_mfma_i32_16x16x64_i8 with R as three int8 limbs and exact int32 accumulation, GEMM 2 on
// v_mfma_f32_16x16x32_f16 with W = D V as a hi+lo fp16 pair; the fp16 forms of GEMM 1 serve f16x2 / f16.
//
// Kernels (a call = one jackknife block; several bins share ONE GEMM 1, see SegTab):
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

#include "../../include/pyrhe_sketch.h"

typedef _Float16 half8 __attribute__((ext_vector_type(8)));
typedef _Float16 half2v __attribute__((ext_vector_type(2)));
typedef float float4v __attribute__((ext_vector_type(4)));
typedef int int4v __attribute__((ext_vector_type(4)));

union Frag {
    half8 h;
    uint32_t u[4];
};

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

// ------------------------------------------------------------------------------------------------
// device helpers
// ------------------------------------------------------------------------------------------------
// PLINK code -> allele-2 count, 16 genotypes at once: 00->0, 10->1, 11->2 (and 01 "missing" -> 1):
// value = code - (code >> 1).  No borrow crosses a crumb because code >= code >> 1.
__device__ __forceinline__ uint32_t value_code(uint32_t w) { return w - ((w >> 1) & 0x55555555u); }
// crumbs equal to 01 (missing call) -> 01 at that crumb, else 00
__device__ __forceinline__ uint32_t missing_mask(uint32_t w) { return (~(w >> 1)) & w & 0x55555555u; }
// substitute `fill` (0/1/2) for missing calls in a value-coded dword (missing currently decodes to 1)
__device__ __forceinline__ uint32_t apply_fill(uint32_t v, uint32_t miss, int fill) {
    return fill == 0 ? v - miss : (fill == 2 ? v + miss : v);
}

// 16 genotypes of a packed dword -> allele-2 counts in the 2-bit value code.  VC = the buffer was recoded once by
// pyrhe_recode_rows (00->0, 10->1, 11->2, 01 "missing"->3): the three VALU ops of value_code() are not paid per kernel
// per pass.  MISSING: substitute `fill` (0/1/2) for missing calls.
template <bool VC, bool MISSING>
__device__ __forceinline__ uint32_t geno_values(uint32_t w, int fill) {
    if (!VC) {
        uint32_t v = value_code(w);
        if (MISSING) v = apply_fill(v, missing_mask(w), fill);
        return v;
    }
    if (!MISSING) return w;
    const uint32_t m = w & (w >> 1) & 0x55555555u;          // 01 at the crumbs that hold 3
    return fill == 0 ? w - m - (m << 1) : (fill == 1 ? w - (m << 1) : w - m);
}

// Two genotypes -> one register of two fp16 values.  Two variants:
//  EXACT = false (precision f16): ONE v_and_b32.  For crumb position t = 0..3 of each 16-bit half of x,
//     (x & (3 << 2t)) read as fp16 is the SUBNORMAL number c * 4^t * 2^-24 (exponent field 0).  The matrix
//     cores multiply fp16 subnormals (tools/microbench.hip E1: exact against small integers), so the constant
//     4^t * 2^-24 is folded into the other operand (GEMM 1: R image pre-divided by 4^t) or into the epilogue
//     (GEMM 2: one accumulator per crumb position), never paid per genotype.  Measured caveat (E1b): with a
//     subnormal operand the MFMA keeps ~19 significant bits of each product -- invisible next to the single
//     fp16 rounding of this mode, too coarse for the hi+lo mode.
//  EXACT = true (precision f16x2): (x & mask) | 0x6400 is the NORMAL fp16 number 1024 + c * 4^t; one packed fma
//     h * 4^-t - 1024 * 4^-t returns c exactly (3 VALU ops per register instead of 1).
// GV (genotype view): 0 = allele count c; 1 = 2 * [c == 2] (bit 1 of the value code), the second operand of the
// dominance encoding of RHE_DOM (rhe_dom.py:23-41: enc = 2p c - 2 [c == 2]).
template <int T, bool EXACT, int GV = 0>
__device__ __forceinline__ uint32_t unpack(uint32_t x) {
    constexpr uint32_t mask = (GV ? 0x00020002u : 0x00030003u) << (2 * T);
    if (!EXACT) return x & mask;
    constexpr float sc = T == 0 ? 1.0f : (T == 1 ? 0.25f : (T == 2 ? 0.0625f : 0.015625f));
    // one v_and_or_b32 (VOP3 takes no literals on gfx9: the mask must sit in a VGPR, the magic in an SGPR; hipcc
    // otherwise emits v_and + v_or with two literals)
    uint32_t vmask = mask;
    asm("" : "+v"(vmask));            // opaque VGPR: lets instruction selection pick v_and_or_b32 (mask VGPR, magic SGPR)
    const uint32_t m = (x & vmask) | 0x64006400u;
    const half2v hx = __builtin_bit_cast(half2v, m);
    const half2v s = {(_Float16)sc, (_Float16)sc};
    const half2v b = {(_Float16)(-1024.0f * sc), (_Float16)(-1024.0f * sc)};
    return __builtin_bit_cast(uint32_t, __builtin_elementwise_fma(hx, s, b));
}
// 8 genotypes (crumbs 0..3 of both 16-bit halves of x) -> MFMA operand fragment.
// k-slot j of the fragment = crumb (j>>1) of half (j&1) (times 4^(j>>1) * 2^-24 when !EXACT).
template <bool EXACT, int GV = 0>
__device__ __forceinline__ half8 unpack4(uint32_t x) {
    Frag f;
    f.u[0] = unpack<0, EXACT, GV>(x);
    f.u[1] = unpack<1, EXACT, GV>(x);
    f.u[2] = unpack<2, EXACT, GV>(x);
    f.u[3] = unpack<3, EXACT, GV>(x);
    return f.h;
}

__device__ __forceinline__ float pow2_scale(float maxabs) {
    // largest power of two s with maxabs * s <= 8192 (fp16 max 65504); 1 for an all-zero column
    if (!(maxabs > 0.0f) || !isfinite(maxabs)) return 1.0f;
    int e;
    frexpf(maxabs, &e);  // maxabs = f * 2^e, f in [0.5, 1)
    return ldexpf(1.0f, 13 - e);
}

// ------------------------------------------------------------------------------------------------
// right-hand-side preparation
// ------------------------------------------------------------------------------------------------
__global__ void col_absmax_kernel(const float* __restrict__ R, int64_t n_pad, int n_cols,
                                  uint32_t* __restrict__ colmax_bits) {
    const int c = blockIdx.y;
    float m = 0.0f;
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n_pad; i += (int64_t)gridDim.x * blockDim.x)
        m = fmaxf(m, fabsf(R[(int64_t)c * n_pad + i]));
    for (int off = 32; off; off >>= 1) m = fmaxf(m, __shfl_xor(m, off));
    if ((threadIdx.x & 63) == 0) atomicMax(&colmax_bits[c], __float_as_uint(m));
}

// sample index of k-slot j of lane-group q in fragment (chunk, d, h) of GEMM 1
__device__ __forceinline__ int64_t g1_sample(int64_t chunk, int d, int h, int q, int j) {
    return chunk * 256 + q * 64 + d * 16 + (j & 1) * 8 + h * 4 + (j >> 1);
}

// Rfrag layout: [chunk][d 4][h 2][part][tile][lane 64][8 halves]; lane = q*16 + c
template <int PARTS, bool EXACT>
__global__ void prep_rfrag_kernel(const float* __restrict__ R, const float* __restrict__ keep,
                                  int64_t n_pad, int n_user_cols, int ntl,
                                  const uint32_t* __restrict__ colmax_bits, float* __restrict__ scale,
                                  half8* __restrict__ rfrag, double* __restrict__ csum) {
    const int64_t gid = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    const int lane = gid & 63;
    int64_t rest = gid >> 6;
    const int tile = rest % ntl;
    rest /= ntl;
    const int part = rest % PARTS;
    rest /= PARTS;
    const int h = rest & 1;
    const int d = (rest >> 1) & 3;
    const int64_t chunk = rest >> 3;
    if (chunk * 256 >= n_pad) return;
    const int c = tile * 16 + (lane & 15);
    const int q = lane >> 4;
    const bool is_ones = (c == n_user_cols);
    float s = 1.0f;
    if (c < n_user_cols) s = pow2_scale(__uint_as_float(colmax_bits[c]));
    Frag f;
    double acc = 0.0;
#pragma unroll
    for (int jj = 0; jj < 4; ++jj) {
        _Float16 v2[2];
#pragma unroll
        for (int b = 0; b < 2; ++b) {
            const int j = jj * 2 + b;
            const int64_t i = g1_sample(chunk, d, h, q, j);
            float x = 0.0f;
            if (c < n_user_cols) x = R[(int64_t)c * n_pad + i] * s;
            else if (is_ones) x = keep[i];
            // the subnormal unpack (!EXACT): compensate the 4^(j>>1) carried by the genotype of k-slot j
            const float slot = EXACT ? 1.0f : (jj == 0 ? 1.0f : (jj == 1 ? 0.25f : (jj == 2 ? 0.0625f : 0.015625f)));
            x *= slot;
            _Float16 hi = (_Float16)x;
            _Float16 val = hi;
            if (PARTS == 2 && part == 1) val = (_Float16)(x - (float)hi);
            v2[b] = val;
            acc += (double)(float)val / (double)slot;
        }
        half2v hv = {v2[0], v2[1]};
        f.u[jj] = __builtin_bit_cast(uint32_t, hv);
    }
    rfrag[gid] = f.h;
    // column sum of exactly the values the matrix cores will multiply (both parts summed)
    acc += __shfl_xor(acc, 16);
    acc += __shfl_xor(acc, 32);
    if (q == 0 && c <= n_user_cols) atomicAdd(&csum[c], acc);
    if (chunk == 0 && d == 0 && h == 0 && part == 0 && q == 0) scale[c] = s;
}

// 2-bit keep mask (11 = retained sample) for the statistics kernel
__global__ void prep_keepmask_kernel(const float* __restrict__ keep, int64_t n_pad, uint32_t* __restrict__ mask) {
    const int64_t w = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (w * 16 >= n_pad) return;
    uint32_t m = 0;
    for (int t = 0; t < 16; ++t)
        if (keep[w * 16 + t] != 0.0f) m |= 3u << (2 * t);
    mask[w] = m;
}

// ------------------------------------------------------------------------------------------------
// GEMM 1:  P[e][c] = (2^-24) * sum_i G[i, row(e)] * Rh[i][c]
// workgroup = 4 waves in lockstep over one sample range; each wave owns 4 tiles of 16 entries (64 SNP
// rows).  One step = 512 samples = one full 128-byte line of every row (two 64-byte halves per lane
// group), so no line is ever fetched for half its bytes; every R fragment is reused 4x from registers.
// The R operand image of the step (16 fragments x NTL x PARTS KiB) is staged once per workgroup through
// a double-buffered LDS ring; the packed genotype words of step s+1 are in flight while step s is
// decoded and multiplied (register prefetch, one barrier per step).
// ------------------------------------------------------------------------------------------------
template <int NTL, int PARTS, bool EXACT, bool MISSING, int GV = 0, bool VC = false>
__global__ __launch_bounds__(256) void gemm1_kernel(const uint8_t* __restrict__ packed, int64_t pitch,
                                                    const int32_t* __restrict__ rows, int n_entries,
                                                    const float* __restrict__ fill,
                                                    const half8* __restrict__ rfrag, int n_steps,
                                                    int steps_per_split, float* __restrict__ P,
                                                    int use_atomic) {
    constexpr int CH = 2;                        // 256-sample chunks per step
    constexpr int FR = 8 * PARTS * NTL;          // fragments (1 KiB each) per chunk
    constexpr int LD = CH * FR / 4;              // 16-byte pieces each thread stages per step
    extern __shared__ __attribute__((aligned(16))) half8 rbuf[];   // [2][CH * FR * 64]
    const int lane = threadIdx.x & 63;
    const int wave = threadIdx.x >> 6;
    const int r = lane & 15, q = lane >> 4;
    const int ebase = (blockIdx.x * 4 + wave) * 64;
    const int s0 = blockIdx.y * steps_per_split;
    const int s1 = min(s0 + steps_per_split, n_steps);
    if (s0 >= s1) return;

    const uint8_t* rowp[4];
    int fillv[4];
#pragma unroll
    for (int t = 0; t < 4; ++t) {
        int e = ebase + 16 * t + r;
        e = e < n_entries ? e : n_entries - 1;
        const int row = rows[e];
        rowp[t] = packed + (int64_t)row * pitch + 16 * q;
        fillv[t] = MISSING ? (int)fill[row] : 0;
    }
    float4v acc[4][NTL];
#pragma unroll
    for (int t = 0; t < 4; ++t)
#pragma unroll
        for (int n = 0; n < NTL; ++n) acc[t][n] = (float4v){0.f, 0.f, 0.f, 0.f};

    uint4 g[2][4][CH];
    auto issue_g = [&](int s, uint4 (&gg)[4][CH]) {
#pragma unroll
        for (int t = 0; t < 4; ++t)
#pragma unroll
            for (int ch = 0; ch < CH; ++ch)
                gg[t][ch] = *reinterpret_cast<const uint4*>(rowp[t] + (int64_t)s * (64 * CH) + 64 * ch);
    };
    // R image of a step -> LDS by LDS-DMA (see gemm1_i8_kernel): one contiguous 1-KiB fragment per wave-instruction
    auto issue_r = [&](int s, int buf) {
        const half8* src = rfrag + (int64_t)s * (CH * FR * 64) + threadIdx.x;
        half8* dst = rbuf + buf * (CH * FR * 64) + wave * 64;
#pragma unroll
        for (int i = 0; i < LD; ++i)
            __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(src + i * 256),
                                             (__attribute__((address_space(3))) void*)(dst + i * 256), 16, 0, 0);
    };
    auto compute = [&](const uint4 (&gg)[4][CH], int buf) {
#pragma unroll
        for (int ch = 0; ch < CH; ++ch) {
            const half8* rf = rbuf + buf * (CH * FR * 64) + ch * (FR * 64) + lane;
#pragma unroll
            for (int d = 0; d < 4; ++d) {
                half8 b[2][PARTS][NTL];
#pragma unroll
                for (int h = 0; h < 2; ++h)
#pragma unroll
                    for (int p = 0; p < PARTS; ++p)
#pragma unroll
                        for (int n = 0; n < NTL; ++n) b[h][p][n] = rf[(((d * 2 + h) * PARTS + p) * NTL + n) * 64];
                uint32_t v[4];
#pragma unroll
                for (int t = 0; t < 4; ++t) {
                    const uint32_t w = d == 0 ? gg[t][ch].x : (d == 1 ? gg[t][ch].y : (d == 2 ? gg[t][ch].z : gg[t][ch].w));
                    v[t] = geno_values<VC, MISSING>(w, fillv[t]);
                }
#pragma unroll
                for (int h = 0; h < 2; ++h) {
#pragma unroll
                    for (int t = 0; t < 4; ++t) {
                        const half8 a = unpack4<EXACT, GV>(h ? (v[t] >> 8) : v[t]);
#pragma unroll
                        for (int p = 0; p < PARTS; ++p)
#pragma unroll
                            for (int n = 0; n < NTL; ++n)
                                acc[t][n] = __builtin_amdgcn_mfma_f32_16x16x32_f16(a, b[h][p][n], acc[t][n], 0, 0, 0);
                    }
                }
            }
        }
    };
    auto touch = [&](const uint4 (&gg)[4][CH]) {
#pragma unroll
        for (int t = 0; t < 4; ++t)
#pragma unroll
            for (int ch = 0; ch < CH; ++ch)
                asm volatile("" ::"v"(gg[t][ch].x), "v"(gg[t][ch].y), "v"(gg[t][ch].z), "v"(gg[t][ch].w));
    };

    // The host guarantees an even number of steps per split, so the ring needs no tail; the prefetch of the step
    // after the last one is clamped (a harmless re-read) rather than branched around.  Order inside a step: touch
    // the packed words of this step (the compiler's wait lands before the new loads), issue the LDS-DMA of the next R
    // slice and the next packed words, compute, wait for the LDS-DMA only (counted vmcnt), raw barrier.
    issue_r(s0, 0);
    issue_g(s0, g[0]);
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();
    for (int s = s0; s < s1; s += 2) {
#pragma unroll
        for (int k = 0; k < 2; ++k) {
            const int nxt = min(s + k + 1, s1 - 1);
            touch(g[k]);
            __builtin_amdgcn_sched_barrier(0);
            issue_r(nxt, k ^ 1);
            __builtin_amdgcn_sched_barrier(0);      // the counted vmcnt below relies on: LDS-DMAs first, packed words after
            issue_g(nxt, g[k ^ 1]);
            __builtin_amdgcn_sched_barrier(0);
            compute(g[k], k);
            __builtin_amdgcn_sched_barrier(0);
            asm volatile("s_waitcnt vmcnt(%0)" ::"n"(4 * CH) : "memory");
            __builtin_amdgcn_s_barrier();
            asm volatile("" ::: "memory");
        }
    }
    // D: lane holds rows 4q + reg (entry), col r
#pragma unroll
    for (int t = 0; t < 4; ++t)
#pragma unroll
        for (int n = 0; n < NTL; ++n)
#pragma unroll
            for (int reg = 0; reg < 4; ++reg) {
                const int e = ebase + 16 * t + 4 * q + reg;
                if (e < n_entries) {
                    float* dst = P + (int64_t)e * (16 * NTL) + n * 16 + r;
                    if (use_atomic) atomicAdd(dst, acc[t][n][reg]);
                    else *dst = acc[t][n][reg];
                }
            }
}

// ------------------------------------------------------------------------------------------------
// int8 variant of GEMM 1 (precision i8x3): R is split into three int8 limbs, x = R * s (|x| <= 127, s a power of
// two), x ~ a1 + a2/128 + a3/16384 (22 bits relative to the column maximum), genotypes are unpacked to bytes
// ((v >> 2t) & 0x03030303: 7 VALU ops per 16 genotypes) and multiplied on v_mfma_i32_16x16x64_i8 -- twice the K
// per instruction of the fp16 form at the same cycles, EXACT int32 accumulation, integer atomics (deterministic).
// Operand map (tools/probe_i8.hip): lane l holds bytes k = 16 (l>>4) + j of row / column l&15.
// Byte 4t+b of a lane's fragment = crumb t + 4b of its dword.
// Rimg layout: [chunk][dword d 4][limb 3][tile][lane 64][16 bytes].
// ------------------------------------------------------------------------------------------------
__device__ __forceinline__ float i8_scale(float maxabs) {
    if (!(maxabs > 0.0f) || !isfinite(maxabs)) return 1.0f;
    int e;
    frexpf(maxabs, &e);                 // maxabs < 2^e  ->  maxabs * 2^(6-e) < 64 <= 127
    return ldexpf(1.0f, 6 - e);
}

__global__ void prep_rimg_i8_kernel(const float* __restrict__ R, const float* __restrict__ keep, int64_t n_pad,
                                    int n_user_cols, int ntl, const uint32_t* __restrict__ colmax_bits,
                                    float* __restrict__ scale, int4v* __restrict__ rimg, double* __restrict__ csum) {
    const int64_t gid = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    const int lane = gid & 63;
    int64_t rest = gid >> 6;
    const int tile = rest % ntl;
    rest /= ntl;
    const int limb = rest % 3;
    rest /= 3;
    const int d = rest & 3;
    const int64_t chunk = rest >> 2;
    if (chunk * 256 >= n_pad) return;
    const int c = tile * 16 + (lane & 15);
    const int q = lane >> 4;
    float s = 1.0f;
    if (c < n_user_cols) s = i8_scale(__uint_as_float(colmax_bits[c]));
    uint32_t w[4];
    double acc = 0.0;
#pragma unroll
    for (int t = 0; t < 4; ++t) {
        uint32_t word = 0;
#pragma unroll
        for (int b = 0; b < 4; ++b) {
            const int64_t i = chunk * 256 + q * 64 + d * 16 + t + 4 * b;
            float x = 0.0f;
            if (c < n_user_cols) x = R[(int64_t)c * n_pad + i] * s;
            else if (c == n_user_cols) x = keep[i];
            const float a1 = rintf(x);
            const float r1 = (x - a1) * 128.0f;
            const float a2 = rintf(r1);
            const float a3 = rintf((r1 - a2) * 128.0f);
            const float a = limb == 0 ? a1 : (limb == 1 ? a2 : a3);
            acc += (double)a * (limb == 0 ? 1.0 : (limb == 1 ? 1.0 / 128.0 : 1.0 / 16384.0));
            word |= ((uint32_t)(int)a & 0xFFu) << (8 * b);
        }
        w[t] = word;
    }
    rimg[gid] = (int4v){(int)w[0], (int)w[1], (int)w[2], (int)w[3]};
    acc += __shfl_xor(acc, 16);
    acc += __shfl_xor(acc, 32);
    if (q == 0 && c <= n_user_cols) atomicAdd(&csum[c], acc);
    if (chunk == 0 && d == 0 && limb == 0 && q == 0) scale[c] = s;
}

template <int NTL, bool MISSING, int GV = 0, bool VC = false>
__global__ __launch_bounds__(256, NTL == 1 ? 3 : 2) void gemm1_i8_kernel(const uint8_t* __restrict__ packed, int64_t pitch,
                                                       const int32_t* __restrict__ rows, int n_entries,
                                                       const float* __restrict__ fill,
                                                       const int4v* __restrict__ rimg, int n_steps,
                                                       int steps_per_split, int* __restrict__ P, int use_atomic) {
    // 256-sample chunks per step.  One column tile: 512-sample steps (a full 128-B line per row) at 154 registers = 3
    // waves per SIMD.  Two tiles: 256-sample steps keep the kernel at 178 registers = 2 waves per SIMD (with 512-sample
    // steps it needs 322 and runs one wave per SIMD: 56 -> 41 ms per pass on the 8-bin + covariates configuration).
    constexpr int CH = NTL == 1 ? 2 : 1;
    constexpr int FR = 12 * NTL;                 // 1-KiB fragments per chunk: 4 dwords x 3 limbs x NTL
    constexpr int LD = CH * FR / 4;
    extern __shared__ __attribute__((aligned(16))) int4v ibuf[];   // [2][CH * FR * 64]
    const int lane = threadIdx.x & 63;
    const int wave = threadIdx.x >> 6;
    const int r = lane & 15, q = lane >> 4;
    const int ebase = (blockIdx.x * 4 + wave) * 64;
    const int s0 = blockIdx.y * steps_per_split;
    const int s1 = min(s0 + steps_per_split, n_steps);
    if (s0 >= s1) return;
    const uint8_t* rowp[4];
    int fillv[4];
#pragma unroll
    for (int t = 0; t < 4; ++t) {
        int e = ebase + 16 * t + r;
        e = e < n_entries ? e : n_entries - 1;
        const int row = rows[e];
        rowp[t] = packed + (int64_t)row * pitch + 16 * q;
        fillv[t] = MISSING ? (int)fill[row] : 0;
    }
    int4v acc[4][3][NTL];
#pragma unroll
    for (int t = 0; t < 4; ++t)
#pragma unroll
        for (int l = 0; l < 3; ++l)
#pragma unroll
            for (int n = 0; n < NTL; ++n) acc[t][l][n] = (int4v){0, 0, 0, 0};
    uint4 g[2][4][CH];
    auto issue_g = [&](int s, uint4 (&gg)[4][CH]) {
#pragma unroll
        for (int t = 0; t < 4; ++t)
#pragma unroll
            for (int ch = 0; ch < CH; ++ch)
                gg[t][ch] = *reinterpret_cast<const uint4*>(rowp[t] + (int64_t)s * (64 * CH) + 64 * ch);
    };
    // R image of a step -> LDS by LDS-DMA (global_load_lds_dwordx4: no VGPR round trip, no ds_write): the image is
    // stored in fragment order, so every wave-instruction copies one contiguous 1-KiB fragment (lane-linear in LDS).
    auto issue_r = [&](int s, int buf) {
        const int4v* src = rimg + (int64_t)s * (CH * FR * 64) + threadIdx.x;
        int4v* dst = ibuf + buf * (CH * FR * 64) + wave * 64;            // wave-uniform base; lane l lands at +l
#pragma unroll
        for (int i = 0; i < LD; ++i)
            __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(src + i * 256),
                                             (__attribute__((address_space(3))) void*)(dst + i * 256), 16, 0, 0);
    };
    auto compute = [&](const uint4 (&gg)[4][CH], int buf) {
#pragma unroll
        for (int ch = 0; ch < CH; ++ch) {
            const int4v* rf = ibuf + buf * (CH * FR * 64) + ch * (FR * 64) + lane;
#pragma unroll
            for (int d = 0; d < 4; ++d) {
                int4v b[3][NTL];
#pragma unroll
                for (int l = 0; l < 3; ++l)
#pragma unroll
                    for (int n = 0; n < NTL; ++n) b[l][n] = rf[((d * 3 + l) * NTL + n) * 64];
#pragma unroll
                for (int t = 0; t < 4; ++t) {
                    const uint32_t w = d == 0 ? gg[t][ch].x : (d == 1 ? gg[t][ch].y : (d == 2 ? gg[t][ch].z : gg[t][ch].w));
                    const uint32_t v = geno_values<VC, MISSING>(w, fillv[t]);
                    constexpr uint32_t bm = GV ? 0x02020202u : 0x03030303u;
                    const int4v a = {(int)(v & bm), (int)((v >> 2) & bm), (int)((v >> 4) & bm), (int)((v >> 6) & bm)};
#pragma unroll
                    for (int l = 0; l < 3; ++l)
#pragma unroll
                        for (int n = 0; n < NTL; ++n)
                            acc[t][l][n] = __builtin_amdgcn_mfma_i32_16x16x64_i8(a, b[l][n], acc[t][l][n], 0, 0, 0);
                }
            }
        }
    };
    // Touch every register of a packed-word stage: the compiler's wait for those loads lands HERE, before this step's
    // new loads are issued (with an LDS-DMA in flight hipcc waits vmcnt(0) at the first use of an ordinary load).
    auto touch = [&](const uint4 (&gg)[4][CH]) {
#pragma unroll
        for (int t = 0; t < 4; ++t)
#pragma unroll
            for (int ch = 0; ch < CH; ++ch)
                asm volatile("" ::"v"(gg[t][ch].x), "v"(gg[t][ch].y), "v"(gg[t][ch].z), "v"(gg[t][ch].w));
    };
    issue_r(s0, 0);
    issue_g(s0, g[0]);
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();
    for (int s = s0; s < s1; s += 2) {
#pragma unroll
        for (int k = 0; k < 2; ++k) {
            const int nxt = min(s + k + 1, s1 - 1);
            touch(g[k]);
            __builtin_amdgcn_sched_barrier(0);
            issue_r(nxt, k ^ 1);
            __builtin_amdgcn_sched_barrier(0);      // the counted vmcnt below relies on: LDS-DMAs first, packed words after
            issue_g(nxt, g[k ^ 1]);
            __builtin_amdgcn_sched_barrier(0);
            compute(g[k], k);
            __builtin_amdgcn_sched_barrier(0);
            // the LD LDS-DMAs are older than the 4*CH packed-word loads behind them: leave those in flight
            asm volatile("s_waitcnt vmcnt(%0)" ::"n"(4 * CH) : "memory");
            __builtin_amdgcn_s_barrier();
            asm volatile("" ::: "memory");
        }
    }
    // P layout: [entry][limb][16 NTL columns] int32
#pragma unroll
    for (int t = 0; t < 4; ++t)
#pragma unroll
        for (int l = 0; l < 3; ++l)
#pragma unroll
            for (int n = 0; n < NTL; ++n)
#pragma unroll
                for (int reg = 0; reg < 4; ++reg) {
                    const int e = ebase + 16 * t + 4 * q + reg;
                    if (e < n_entries) {
                        int* dst = P + ((int64_t)e * 3 + l) * (16 * NTL) + n * 16 + r;
                        if (use_atomic) atomicAdd(dst, acc[t][l][n][reg]);
                        else *dst = acc[t][l][n][reg];
                    }
                }
}

// ------------------------------------------------------------------------------------------------
// Segment table of one chunk of entries, passed to the kernels BY VALUE (kernel argument: no upload, no sync).
// A chunk is a run of the caller's concatenated per-bin entry lists; segment g = the part of bin `bin[g]` inside the
// chunk = chunk entries start[g] .. start[g+1]-1.  GEMM 1 does not care about bins (X^T R is per SNP) and runs once
// over the whole chunk; everything that is per bin -- the column maxima / scales of W, the rank-1 corrections, the
// norm sums, GEMM 2's K range and its output slab -- is indexed by segment (stride kSegCols columns).  GEMM 2 walks
// K blocks of 32 entries, so each segment owns K blocks kb0[g] .. kb0[g+1]-1 (its tail padded with zero-weight rows).
// ------------------------------------------------------------------------------------------------
constexpr int kMaxSeg = 64;
constexpr int kSegCols = 32;
struct SegTab {
    int nseg;
    int start[kMaxSeg + 1];
    int kb0[kMaxSeg + 1];
    int bin[kMaxSeg];
    unsigned char acc[kMaxSeg];   // 1: add to the caller's outputs of that bin, 0: overwrite
};
__device__ __forceinline__ int seg_of_entry(const SegTab& t, int e) {
    int g = 0;
    while (g + 1 < t.nseg && e >= t.start[g + 1]) ++g;
    return g;
}
__device__ __forceinline__ int seg_of_kb(const SegTab& t, int kb) {
    int g = 0;
    while (g + 1 < t.nseg && kb >= t.kb0[g + 1]) ++g;
    return g;
}

// Ordered per-segment sum of the squares written by mid 1 / mid (dominance): WORKGROUP w = (segment g, norm column i)
// adds vsq[e][i] over the segment's entries -- thread-strided (4 loads in flight per thread), a fixed butterfly per
// wave, the four wave sums in a fixed order -- and hands the sum to the caller's [nbin][n_norm] array (row tab.bin[g],
// overwrite or add per tab.acc[g]).  Same bits on every run.  Call with all 256 threads of the workgroup.
__device__ __forceinline__ void norm_reduce(int wg, const double* __restrict__ vsq, int n_norm, const SegTab& tab,
                                            double* __restrict__ norm_dst, double* s_part /* shared [4] */) {
    if (wg >= tab.nseg * n_norm) return;                       // uniform per workgroup
    const int g = wg / n_norm, i = wg - g * n_norm;
    double s0 = 0.0, s1 = 0.0, s2 = 0.0, s3 = 0.0;
    const int e1 = tab.start[g + 1];
    int e = tab.start[g] + threadIdx.x;
    for (; e + 768 < e1; e += 1024) {
        s0 += vsq[(int64_t)e * n_norm + i];
        s1 += vsq[(int64_t)(e + 256) * n_norm + i];
        s2 += vsq[(int64_t)(e + 512) * n_norm + i];
        s3 += vsq[(int64_t)(e + 768) * n_norm + i];
    }
    for (; e < e1; e += 256) s0 += vsq[(int64_t)e * n_norm + i];
    double s = (s0 + s1) + (s2 + s3);
#pragma unroll
    for (int off = 32; off; off >>= 1) {
        int lo = __double2loint(s), hi = __double2hiint(s);
        lo = __shfl_xor(lo, off);
        hi = __shfl_xor(hi, off);
        s += __hiloint2double(hi, lo);
    }
    if ((threadIdx.x & 63) == 0) s_part[threadIdx.x >> 6] = s;
    __syncthreads();
    if (threadIdx.x == 0) {
        const double tot = (s_part[0] + s_part[1]) + (s_part[2] + s_part[3]);
        double* d = norm_dst + (int64_t)tab.bin[g] * n_norm + i;
        *d = tab.acc[g] ? *d + tot : tot;
    }
}

__global__ __launch_bounds__(256) void norm_reduce_kernel(const double* __restrict__ vsq, int n_norm, double* __restrict__ norm_dst,
                                                          const SegTab tab) {
    __shared__ double s_part[4];
    norm_reduce(blockIdx.x, vsq, n_norm, tab, norm_dst, s_part);
}

// ------------------------------------------------------------------------------------------------
// mid 1: per entry statistics, V = X^T R, norm squares, W = D V (float32) + column maxima
// ------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void mid1_kernel(const float* __restrict__ P, float pscale, int i8, int n_entries, int ncol_p,
                                                   int n_sketch, int n_norm, const float* __restrict__ scale,
                                                   const double* __restrict__ csum, float n_keep, float* __restrict__ mu_out,
                                                   float* __restrict__ Wtmp, uint32_t* __restrict__ wmax_bits,
                                                   double* __restrict__ vsq, int32_t* __restrict__ colsum_out,
                                                   const SegTab tab) {
    // workgroup = 64 entries; thread (c16, el): column c16 of a 16-column tile, entries el, el+16, el+32, el+48.
    // Consecutive lanes read consecutive columns of one entry (coalesced); column maxima are kept in registers over
    // the 4 entries, folded over the wave by 2 shuffles and over the 4 waves through LDS.  Norm columns: the squares
    // V^2 go to vsq[entry][norm column]; norm_reduce() adds them per segment in a FIXED order (no float64 atomics:
    // yXXy is reproducible bit for bit).
    __shared__ float s_mu[64], s_d[64];
    __shared__ float s_max[4][16];
    const int c16 = threadIdx.x & 15, el = threadIdx.x >> 4, wave = threadIdx.x >> 6;
    const int ones = n_sketch + n_norm;
    const int ebase = blockIdx.x * 64;
    // segment of this workgroup's entries; a workgroup that straddles a boundary (at most nseg-1 of them) skips the
    // in-register reduction and sends every entry's contribution straight to its segment's slot
    const int g0 = seg_of_entry(tab, ebase);
    const bool mixed = seg_of_entry(tab, min(ebase + 63, n_entries - 1)) != g0;
    wmax_bits += g0 * kSegCols;
    if (threadIdx.x < 64) {
        const int e = ebase + threadIdx.x;
        float mu = 0.f, d = 0.f;
        if (e < n_entries) {
            // exact integer (< 2^24); float P carries 1/pscale (2^-24 with the subnormal unpack); int8 mode: limb 0 as int32
            const float sum = i8 ? (float)reinterpret_cast<const int*>(P)[(int64_t)e * 3 * ncol_p + ones]
                                 : P[(int64_t)e * ncol_p + ones] * pscale;
            // float32, the reference's operation order (base.py:292-293): mean; 1/sqrt(mean*(1-0.5*mean))
            mu = sum / n_keep;
            d = 1.0f / sqrtf(mu * (1.0f - 0.5f * mu));
            mu_out[e] = mu;
            if (colsum_out) colsum_out[e] = (int32_t)sum;
        }
        s_mu[threadIdx.x] = mu;
        s_d[threadIdx.x] = d;
    }
    __syncthreads();
    for (int tile = 0; tile * 16 < ones; ++tile) {
        const int c = tile * 16 + c16;
        const bool colok = c < ones;
        float mx = 0.f;
        if (colok) {
            const double inv_s = 1.0 / (double)scale[c];
            const double cs = csum[c] * inv_s;
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                const int lo = i * 16 + el, e = ebase + lo;
                if (e < n_entries) {
                    double pc;
                    if (i8) {
                        const int* Pi = reinterpret_cast<const int*>(P) + (int64_t)e * 3 * ncol_p + c;
                        pc = ((double)Pi[0] + (double)Pi[ncol_p] * (1.0 / 128.0) + (double)Pi[2 * ncol_p] * (1.0 / 16384.0)) * inv_s;
                    } else {
                        pc = (double)P[(int64_t)e * ncol_p + c] * (double)pscale * inv_s;
                    }
                    const float d = s_d[lo];
                    const float v = (float)((double)d * (pc - (double)s_mu[lo] * cs));
                    if (c < n_sketch) {
                        const float w = d * v;
                        Wtmp[(int64_t)e * n_sketch + c] = w;
                        if (mixed) atomicMax(&wmax_bits[(seg_of_entry(tab, e) - g0) * kSegCols + c], __float_as_uint(fabsf(w)));
                        else mx = fmaxf(mx, fabsf(w));
                    } else {
                        vsq[(int64_t)e * n_norm + (c - n_sketch)] = (double)v * (double)v;
                    }
                }
            }
        }
        mx = fmaxf(mx, __shfl_xor(mx, 16));
        mx = fmaxf(mx, __shfl_xor(mx, 32));
        if ((threadIdx.x & 63) < 16) s_max[wave][c16] = mx;
        __syncthreads();
        if (threadIdx.x < 16 && colok && c < n_sketch) {
            const float m = fmaxf(fmaxf(s_max[0][c16], s_max[1][c16]), fmaxf(s_max[2][c16], s_max[3][c16]));
            atomicMax(&wmax_bits[c], __float_as_uint(m));        // a maximum: order-independent
        }
        __syncthreads();
    }
}

// ------------------------------------------------------------------------------------------------
// mid (dominance): RHE_DOM's second estimate per bin (rhe_dom.py:15-41).  With H = [g == 1], T = [g == 2]:
//   enc = a H + b T,  a = 2p, b = 4p - 2, p = mean(g)/2 (float32);  X_dom = (enc - m) t,  m = mean(enc),
//   t = 1/(2p(1-p)) (float32).  Since H = G - 2T:   X_dom^T r = t (a G^T r - 2 T^T r - m 1^T r)
//                                                   X_dom v   = G (t a v) + 2T (-t v) - 1 (m t v)
// PG is GEMM 1's output for the allele-count view, PT for the 2*[g==2] view (so PT = 2 T^T r, its ones column
// 2 n2).  Outputs: W1 = t a V (for the G view), W2 = -t V (for the 2T view), cmu = m / a (rank-1 coefficient
// relative to W1), sum V^2 for the norm columns.
// ------------------------------------------------------------------------------------------------
__global__ void mid_dom1_kernel(const float* __restrict__ PG, const float* __restrict__ PT, float pscale, int i8,
                                int n_entries, int ncol_p, int n_sketch, int n_norm, const float* __restrict__ scale,
                                const double* __restrict__ csum, float n_keep, float* __restrict__ cmu,
                                float* __restrict__ W1, float* __restrict__ W2, uint32_t* __restrict__ wmax1,
                                uint32_t* __restrict__ wmax2, double* __restrict__ vsq) {
    const int e = blockIdx.x * blockDim.x + threadIdx.x;
    const int ones = n_sketch + n_norm;
    const bool live = e < n_entries;
    auto rd = [&](const float* P, int c) -> double {
        if (i8) {
            const int* Pi = reinterpret_cast<const int*>(P) + (int64_t)e * 3 * ncol_p + c;
            return (double)Pi[0] + (double)Pi[ncol_p] * (1.0 / 128.0) + (double)Pi[2 * ncol_p] * (1.0 / 16384.0);
        }
        return (double)P[(int64_t)e * ncol_p + c] * (double)pscale;
    };
    float a = 0.f, t = 0.f;
    double m = 0.0, b = 0.0;
    if (live) {
        const double sumG = rd(PG, ones), n2 = 0.5 * rd(PT, ones);     // exact integers
        const double n1 = sumG - 2.0 * n2;
        const float maf = ((float)sumG / n_keep) / 2.0f;                // np.mean(geno) / 2, float32
        a = 2.0f * maf;
        const float bf = 4.0f * maf - 2.0f;
        b = (double)bf;
        t = 1.0f / (2.0f * maf * (1.0f - maf));
        m = (n1 * (double)a + n2 * b) / (double)n_keep;
        cmu[e] = a != 0.f ? (float)(m / (double)a) : 0.f;
    }
    for (int c = 0; c < n_sketch + n_norm; ++c) {
        float v = 0.f;
        if (live) {
            const double inv_s = 1.0 / (double)scale[c];
            const double g = rd(PG, c) * inv_s, tt = 0.5 * rd(PT, c) * inv_s;
            v = (float)((double)t * ((double)a * (g - 2.0 * tt) + b * tt - m * (csum[c] * inv_s)));
        }
        if (c < n_sketch) {
            const float w1 = t * a * v, w2 = -t * v;
            if (live) {
                W1[(int64_t)e * n_sketch + c] = w1;
                W2[(int64_t)e * n_sketch + c] = w2;
            }
            float m1 = live ? fabsf(w1) : 0.f, m2 = live ? fabsf(w2) : 0.f;
            for (int off = 32; off; off >>= 1) {
                m1 = fmaxf(m1, __shfl_xor(m1, off));
                m2 = fmaxf(m2, __shfl_xor(m2, off));
            }
            if ((threadIdx.x & 63) == 0) {
                atomicMax(&wmax1[c], __float_as_uint(m1));
                atomicMax(&wmax2[c], __float_as_uint(m2));
            }
        } else if (live) {
            vsq[(int64_t)e * n_norm + (c - n_sketch)] = (double)v * (double)v;
        }
    }
}

// ------------------------------------------------------------------------------------------------
// mid 2: W -> fp16 operand image for GEMM 2 + rank-1 correction  corr[c] = sum_e mu_e * Wh[e][c]
// Wfrag layout: [kb][part][tile][lane 64][8 halves]; lane = q*16 + c; element j <-> entry 32 kb + 8 q + j
// ------------------------------------------------------------------------------------------------
// Also (folded in to save two launches per entry list): the row list padded to a multiple of 32 for GEMM 2
// (pad rows read rows[0]; their W is zero) and the hand-over of the norm sums of mid 1 to the caller's buffer.
// Per segment (see SegTab): scales, corrections and norm sums live at [g * kSegCols + c]; norm_dst is the caller's
// [nbin][n_norm] array, segment g lands in row tab.bin[g].
template <int PARTS>
__global__ void mid2_kernel(const float* __restrict__ Wtmp, const float* __restrict__ mu,
                            int n_sketch, int nto, const uint32_t* __restrict__ wmax_bits,
                            half8* __restrict__ wfrag, double* __restrict__ corr, float* __restrict__ wscale,
                            const int32_t* __restrict__ rows, int32_t* __restrict__ rows32,
                            const double* __restrict__ vsq, double* __restrict__ norm_dst, int n_norm,
                            const SegTab tab) {
    const int64_t gid = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    const int n_kb = tab.kb0[tab.nseg];
    if (rows32 && gid < (int64_t)n_kb * 32) {
        const int g = seg_of_kb(tab, (int)(gid >> 5));
        const int idx = (int)gid - tab.kb0[g] * 32;
        rows32[gid] = rows[tab.start[g] + (idx < tab.start[g + 1] - tab.start[g] ? idx : 0)];
    }
    __shared__ double s_part[4];
    if (norm_dst) norm_reduce(blockIdx.x, vsq, n_norm, tab, norm_dst, s_part);
    const int lane = gid & 63;
    int64_t rest = gid >> 6;
    const int tile = rest % nto;
    rest /= nto;
    const int part = rest % PARTS;
    const int64_t kb = rest / PARTS;
    if (kb >= n_kb) return;
    const int g = seg_of_kb(tab, (int)kb);
    const int kbl = (int)kb - tab.kb0[g], ebase = tab.start[g], n_entries = tab.start[g + 1];
    wmax_bits += g * kSegCols;
    corr += g * kSegCols;
    wscale += g * kSegCols;
    const int c = tile * 16 + (lane & 15);
    const int q = lane >> 4;
    const float s = c < n_sketch ? pow2_scale(__uint_as_float(wmax_bits[c])) : 1.0f;
    Frag f;
    double acc = 0.0;
#pragma unroll
    for (int jj = 0; jj < 4; ++jj) {
        _Float16 v2[2];
#pragma unroll
        for (int b = 0; b < 2; ++b) {
            const int e = ebase + kbl * 32 + q * 8 + jj * 2 + b;
            float x = 0.f, m = 0.f;
            if (e < n_entries && c < n_sketch) {
                x = Wtmp[(int64_t)e * n_sketch + c] * s;
                m = mu[e];
            }
            const _Float16 hi = (_Float16)x;
            _Float16 val = hi;
            if (PARTS == 2 && part == 1) val = (_Float16)(x - (float)hi);
            v2[b] = val;
            acc += (double)m * (double)(float)val;
        }
        half2v hv = {v2[0], v2[1]};
        f.u[jj] = __builtin_bit_cast(uint32_t, hv);
    }
    wfrag[gid] = f.h;
    acc += __shfl_xor(acc, 16);
    acc += __shfl_xor(acc, 32);
    if (q == 0 && c < n_sketch) atomicAdd(&corr[c], acc);
    if (kbl == 0 && part == 0 && q == 0) wscale[c] = s;
}

// ------------------------------------------------------------------------------------------------
// GEMM 2:  out[c][i] (+)= keep[i] * ( sum_e G[i, row(e)] * Wh[e][c] - corr[c] ) / s_c
// workgroup = 4 waves; a wave owns 256*IW samples (IW dwords per lane per SNP row) and walks the entry
// list 32 SNPs at a time: lane (n, q) reads IW dwords of rows 8q..8q+7 (16 lanes x 4*IW B contiguous
// per row).  The packed words and W fragments of the next NS-1 steps are in flight in a register ring
// while a step is decoded and multiplied (HBM latency x bandwidth needs ~64-128 KiB in flight per CU);
// the row indices of the workgroup's K range sit in LDS, so no dependent global load ever drains the
// in-order vector-memory queue.  MFMA u consumes crumb u of every dword: with the subnormal unpack the
// accumulator of MFMA u carries the constant 4^(u&3) * 2^-24, removed once in the epilogue.
// ------------------------------------------------------------------------------------------------
template <int NTO, int PARTS, int IW, bool EXACT, bool MISSING, int GV = 0, bool VC = false>
__global__ __launch_bounds__(256, 2) void gemm2_kernel(const uint8_t* __restrict__ packed, int64_t pitch,
                                                    const int32_t* __restrict__ rows32, int n_kb,
                                                    int kb_per_split, const float* __restrict__ fill,
                                                    const half8* __restrict__ wfrag,
                                                    const double* __restrict__ corr,
                                                    const float* __restrict__ wscale,
                                                    const float* __restrict__ keep, int n_sketch,
                                                    int64_t n_pad, float* __restrict__ out, int mode,
                                                    float* __restrict__ partial, const SegTab tab) {
    // mode: 0 store, 1 read-modify-write, 2 atomicAdd.  With `partial` != NULL, K split y > 0 stores its share
    // into partial[y-1] (summed into `out` by add_partials_kernel) instead of touching `out`.
    // Segment mode (tab.nseg > 0): blockIdx.y is a SEGMENT instead of a K split -- all bins of a chunk in one launch;
    // the workgroup takes its K range, scales, correction, output slab and store mode from the table.
    constexpr int NU = 16 * IW;                  // MFMAs (= samples per lane) per step
    constexpr int NS = 3;                        // register ring depth
    extern __shared__ __attribute__((aligned(16))) int32_t s_rows[];
    const int lane = threadIdx.x & 63;
    const int wave = threadIdx.x >> 6;
    const int n = lane & 15, q = lane >> 4;
    const int64_t grp = (int64_t)blockIdx.x * 4 + wave;
    const bool segmode = tab.nseg > 0;
    int kb0 = blockIdx.y * kb_per_split;
    int kb1 = min(kb0 + kb_per_split, n_kb);
    int ysplit = blockIdx.y;
    if (segmode) {
        const int g = blockIdx.y;
        kb0 = tab.kb0[g];
        kb1 = tab.kb0[g + 1];
        corr += g * kSegCols;
        wscale += g * kSegCols;
        out += (int64_t)tab.bin[g] * n_sketch * n_pad;
        mode = tab.acc[g] ? 1 : 0;
        ysplit = 0;
    }
    if (kb0 >= kb1) return;
    for (int i = threadIdx.x; i < (kb1 - kb0) * 32; i += 256) s_rows[i] = rows32[(int64_t)kb0 * 32 + i];
    __syncthreads();
    if (grp * (256 * IW) >= n_pad) return;
    // 32-bit byte offsets: row * pitch with the full-rate 24-bit multiply (rows, pitch < 2^24 and rows * pitch < 2^32
    // are checked on the host); a 64-bit multiply-add per row is four quarter-rate VALU ops per load otherwise
    const uint32_t coloff = (uint32_t)(grp * (64 * IW) + n * (4 * IW));
    const uint32_t pitch24 = (uint32_t)pitch;

    float4v acc[NTO][NU];
#pragma unroll
    for (int t = 0; t < NTO; ++t)
#pragma unroll
        for (int u = 0; u < NU; ++u) acc[t][u] = (float4v){0.f, 0.f, 0.f, 0.f};

    uint32_t w[NS][8][IW];
    half8 a[NS][PARTS][NTO];
    int fl[NS][8];

    auto issue = [&](int kb, uint32_t (&ww)[8][IW], half8 (&aa)[PARTS][NTO], int (&ff)[8]) {
        const int4* rp = reinterpret_cast<const int4*>(s_rows + (kb - kb0) * 32 + q * 8);
        const int4 r0 = rp[0], r1 = rp[1];
        const int rr[8] = {r0.x, r0.y, r0.z, r0.w, r1.x, r1.y, r1.z, r1.w};
#pragma unroll
        for (int j = 0; j < 8; ++j) {
            const uint8_t* p = packed + (size_t)(__umul24((uint32_t)rr[j], pitch24) + coloff);
            if (IW == 1) {
                ww[j][0] = *reinterpret_cast<const uint32_t*>(p);
            } else {
                const uint2 t2 = *reinterpret_cast<const uint2*>(p);
                ww[j][0] = t2.x;
                ww[j][IW - 1] = t2.y;
            }
            if (MISSING) ff[j] = (int)fill[rr[j]];
        }
        const half8* wf = wfrag + (int64_t)kb * (PARTS * NTO * 64) + lane;
#pragma unroll
        for (int p = 0; p < PARTS; ++p)
#pragma unroll
            for (int t = 0; t < NTO; ++t) aa[p][t] = wf[(p * NTO + t) * 64];
    };

    auto compute = [&](const uint32_t (&ww)[8][IW], const half8 (&aa)[PARTS][NTO], const int (&ff)[8]) {
#pragma unroll
        for (int iw = 0; iw < IW; ++iw) {
            uint32_t v[8];
#pragma unroll
            for (int j = 0; j < 8; ++j) {
                v[j] = geno_values<VC, MISSING>(ww[j][iw], MISSING ? ff[j] : 0);
            }
            // x[p]: low halves of rows 2p (lo 16 bits) and 2p+1 (hi 16 bits) = samples 0..7 of the dword;
            // y[p]: high halves = samples 8..15.
            uint32_t x[4], y[4];
#pragma unroll
            for (int p = 0; p < 4; ++p) {
                x[p] = __builtin_amdgcn_perm(v[2 * p + 1], v[2 * p], 0x05040100u);
                y[p] = __builtin_amdgcn_perm(v[2 * p + 1], v[2 * p], 0x07060302u);
            }
            // fragments in groups of 4 (crumbs 0..3 of one source register set): unpack all four, then issue their MFMAs
            // round-robin so consecutive matrix instructions never touch the same accumulator or operand registers
#pragma unroll
            for (int ug = 0; ug < 4; ++ug) {
                Frag f[4];
#pragma unroll
                for (int p = 0; p < 4; ++p) {
                    uint32_t src = (ug & 2) ? y[p] : x[p];
                    if (ug & 1) src >>= 8;
                    f[0].u[p] = unpack<0, EXACT, GV>(src);
                    f[1].u[p] = unpack<1, EXACT, GV>(src);
                    f[2].u[p] = unpack<2, EXACT, GV>(src);
                    f[3].u[p] = unpack<3, EXACT, GV>(src);
                }
#pragma unroll
                for (int p = 0; p < PARTS; ++p)
#pragma unroll
                    for (int t = 0; t < NTO; ++t)
#pragma unroll
                        for (int uu = 0; uu < 4; ++uu)
                            acc[t][iw * 16 + ug * 4 + uu] = __builtin_amdgcn_mfma_f32_16x16x32_f16(aa[p][t], f[uu].h, acc[t][iw * 16 + ug * 4 + uu], 0, 0, 0);
            }
        }
    };

    // ring prologue / steady state / tail.  Prefetches past the end are clamped to the last step (harmless re-reads)
    // instead of branched around: a conditional load makes the compiler wait for the NEW loads on the hot path.
#pragma unroll
    for (int s = 0; s < NS; ++s) issue(min(kb0 + s, kb1 - 1), w[s], a[s], fl[s]);
    int kb = kb0;
    for (; kb + NS <= kb1; kb += NS) {
#pragma unroll
        for (int s = 0; s < NS; ++s) {
            compute(w[s], a[s], fl[s]);
            __builtin_amdgcn_sched_barrier(0);      // refill the stage right here, not wherever the scheduler sinks it
            issue(min(kb + s + NS, kb1 - 1), w[s], a[s], fl[s]);
            __builtin_amdgcn_sched_barrier(0);
        }
    }
#pragma unroll
    for (int s = 0; s < NS - 1; ++s)
        if (kb + s < kb1) compute(w[s], a[s], fl[s]);
    // D[row = column c = 4q + reg][col = n]; for MFMA (iw, u) the sample is (256 IW) grp + (16 IW) n + 16 iw + u:
    // a lane owns 16*IW consecutive samples per (tile, reg).
    const int64_t i0 = grp * (256 * IW) + n * NU;
    float kp[NU];
#pragma unroll
    for (int u4 = 0; u4 < NU / 4; ++u4) {
        const float4 k4 = *reinterpret_cast<const float4*>(keep + i0 + 4 * u4);
        kp[4 * u4] = k4.x; kp[4 * u4 + 1] = k4.y; kp[4 * u4 + 2] = k4.z; kp[4 * u4 + 3] = k4.w;
    }
    // undo the unpack constant: 2^24 * 4^-(u&3)
    const float usc[4] = {EXACT ? 1.0f : 16777216.0f, EXACT ? 1.0f : 4194304.0f,
                          EXACT ? 1.0f : 1048576.0f, EXACT ? 1.0f : 262144.0f};
#pragma unroll
    for (int t = 0; t < NTO; ++t)
#pragma unroll
        for (int reg = 0; reg < 4; ++reg) {
            const int c = t * 16 + 4 * q + reg;
            if (c >= n_sketch) continue;
            const float cr = ysplit == 0 ? (float)corr[c] : 0.f;
            const float inv = 1.0f / wscale[c];
            float* dst = out + (int64_t)c * n_pad + i0;
            if (partial && ysplit > 0) {
                dst = partial + ((int64_t)(ysplit - 1) * n_sketch + c) * n_pad + i0;
                mode = 0;
            }
#pragma unroll
            for (int u4 = 0; u4 < NU / 4; ++u4) {
                float4 val;
                val.x = (acc[t][4 * u4 + 0][reg] * usc[0] - cr) * inv * kp[4 * u4 + 0];
                val.y = (acc[t][4 * u4 + 1][reg] * usc[1] - cr) * inv * kp[4 * u4 + 1];
                val.z = (acc[t][4 * u4 + 2][reg] * usc[2] - cr) * inv * kp[4 * u4 + 2];
                val.w = (acc[t][4 * u4 + 3][reg] * usc[3] - cr) * inv * kp[4 * u4 + 3];
                float4* d4 = reinterpret_cast<float4*>(dst + 4 * u4);
                if (mode == 0) {
                    *d4 = val;
                } else if (mode == 1) {
                    float4 o = *d4;
                    o.x += val.x; o.y += val.y; o.z += val.z; o.w += val.w;
                    *d4 = o;
                } else {
                    atomicAdd(dst + 4 * u4 + 0, val.x);
                    atomicAdd(dst + 4 * u4 + 1, val.y);
                    atomicAdd(dst + 4 * u4 + 2, val.z);
                    atomicAdd(dst + 4 * u4 + 3, val.w);
                }
            }
        }
}

// ------------------------------------------------------------------------------------------------
// int8 variant of GEMM 2 (precision i8x3, one tile of <= 16 sketch columns): W = D V as three int8 limbs (per-column
// power-of-two scale, |x| < 64), one int32 accumulator set per limb, K = 64 SNPs per step.
// Lane (n, q) reads dword n of rows 16q..16q+15; MFMA u needs, per lane, the crumb u of those 16 rows as 16 bytes:
// rows are spread to bytes ((v >> 2t), masked after the transpose) and 4x4 byte blocks are transposed across row
// quads with v_perm_b32 (8 per block).  Wimg layout: [kb64][limb 3][lane 64][16 bytes], lane = q*16 + c,
// byte j <-> entry 64 kb + 16 q + j.
// ------------------------------------------------------------------------------------------------
__global__ void mid2_i8_kernel(const float* __restrict__ Wtmp, const float* __restrict__ mu, int n_entries, int n_kb,
                               int n_sketch, const uint32_t* __restrict__ wmax_bits, int4v* __restrict__ wimg,
                               double* __restrict__ corr, float* __restrict__ wscale) {
    const int64_t gid = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    const int lane = gid & 63;
    int64_t rest = gid >> 6;
    const int limb = rest % 3;
    const int64_t kb = rest / 3;
    if (kb >= n_kb) return;
    const int c = lane & 15, q = lane >> 4;
    const float s = c < n_sketch ? i8_scale(__uint_as_float(wmax_bits[c])) : 1.0f;
    uint32_t w[4];
    double acc = 0.0;
#pragma unroll
    for (int p = 0; p < 4; ++p) {
        uint32_t word = 0;
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            const int e = kb * 64 + q * 16 + p * 4 + r;
            float x = 0.f, m = 0.f;
            if (e < n_entries && c < n_sketch) {
                x = Wtmp[(int64_t)e * n_sketch + c] * s;
                m = mu[e];
            }
            const float a1 = rintf(x);
            const float r1 = (x - a1) * 128.0f;
            const float a2 = rintf(r1);
            const float a3 = rintf((r1 - a2) * 128.0f);
            const float a = limb == 0 ? a1 : (limb == 1 ? a2 : a3);
            acc += (double)m * (double)a * (limb == 0 ? 1.0 : (limb == 1 ? 1.0 / 128.0 : 1.0 / 16384.0));
            word |= ((uint32_t)(int)a & 0xFFu) << (8 * r);
        }
        w[p] = word;
    }
    wimg[gid] = (int4v){(int)w[0], (int)w[1], (int)w[2], (int)w[3]};
    acc += __shfl_xor(acc, 16);
    acc += __shfl_xor(acc, 32);
    if (q == 0 && c < n_sketch) atomicAdd(&corr[c], acc);
    if (kb == 0 && limb == 0 && q == 0) wscale[c] = s;
}

template <bool MISSING, bool VC = false>
__global__ __launch_bounds__(256, 1) void gemm2_i8_kernel(const uint8_t* __restrict__ packed, int64_t pitch,
                                                          const int32_t* __restrict__ rows64, int n_kb,
                                                          int kb_per_split, const float* __restrict__ fill,
                                                          const int4v* __restrict__ wimg,
                                                          const double* __restrict__ corr,
                                                          const float* __restrict__ wscale,
                                                          const float* __restrict__ keep, int n_sketch,
                                                          int64_t n_pad, float* __restrict__ out, int mode,
                                                          float* __restrict__ partial) {
    constexpr int NS = 2;
    extern __shared__ __attribute__((aligned(16))) int32_t s_rows[];
    const int lane = threadIdx.x & 63;
    const int wave = threadIdx.x >> 6;
    const int n = lane & 15, q = lane >> 4;
    const int64_t grp = (int64_t)blockIdx.x * 4 + wave;
    const int kb0 = blockIdx.y * kb_per_split;
    const int kb1 = min(kb0 + kb_per_split, n_kb);
    if (kb0 >= kb1) return;
    for (int i = threadIdx.x; i < (kb1 - kb0) * 64; i += 256) s_rows[i] = rows64[(int64_t)kb0 * 64 + i];
    __syncthreads();
    if (grp * 256 >= n_pad) return;
    const uint32_t coloff = (uint32_t)(grp * 64 + n * 4);
    const uint32_t pitch24 = (uint32_t)pitch;

    int4v acc[3][16];
#pragma unroll
    for (int l = 0; l < 3; ++l)
#pragma unroll
        for (int u = 0; u < 16; ++u) acc[l][u] = (int4v){0, 0, 0, 0};

    uint32_t w[NS][16];
    int4v a[NS][3];
    int fl[NS][16];
    auto issue = [&](int kb, uint32_t (&ww)[16], int4v (&aa)[3], int (&ff)[16]) {
        const int4* rp = reinterpret_cast<const int4*>(s_rows + (kb - kb0) * 64 + q * 16);
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const int4 r4 = rp[i];
            const int rr[4] = {r4.x, r4.y, r4.z, r4.w};
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                ww[4 * i + r] = *reinterpret_cast<const uint32_t*>(packed + (size_t)(__umul24((uint32_t)rr[r], pitch24) + coloff));
                if (MISSING) ff[4 * i + r] = (int)fill[rr[r]];
            }
        }
        const int4v* wf = wimg + (int64_t)kb * (3 * 64) + lane;
#pragma unroll
        for (int l = 0; l < 3; ++l) aa[l] = wf[l * 64];
    };
    auto compute = [&](const uint32_t (&ww)[16], const int4v (&aa)[3], const int (&ff)[16]) {
        int4v opnd[16];      // opnd[u][p] = bytes (row 4p .. 4p+3) of crumb u
#pragma unroll
        for (int p = 0; p < 4; ++p) {
            uint32_t v[4];
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                v[r] = geno_values<VC, MISSING>(ww[4 * p + r], MISSING ? ff[4 * p + r] : 0);
            }
#pragma unroll
            for (int t = 0; t < 4; ++t) {
                // bytes of (v >> 2t): crumbs t, t+4, t+8, t+12 in their low 2 bits (masked after the transpose)
                const uint32_t x0 = v[0] >> (2 * t), x1 = v[1] >> (2 * t), x2 = v[2] >> (2 * t), x3 = v[3] >> (2 * t);
                const uint32_t l01 = __builtin_amdgcn_perm(x1, x0, 0x05010400u);   // [x0.b0, x1.b0, x0.b1, x1.b1]
                const uint32_t h01 = __builtin_amdgcn_perm(x1, x0, 0x07030602u);   // [x0.b2, x1.b2, x0.b3, x1.b3]
                const uint32_t l23 = __builtin_amdgcn_perm(x3, x2, 0x05010400u);
                const uint32_t h23 = __builtin_amdgcn_perm(x3, x2, 0x07030602u);
                opnd[t + 0][p] = (int)(__builtin_amdgcn_perm(l23, l01, 0x05040100u) & 0x03030303u);   // byte 0 of every row
                opnd[t + 4][p] = (int)(__builtin_amdgcn_perm(l23, l01, 0x07060302u) & 0x03030303u);   // byte 1
                opnd[t + 8][p] = (int)(__builtin_amdgcn_perm(h23, h01, 0x05040100u) & 0x03030303u);   // byte 2
                opnd[t + 12][p] = (int)(__builtin_amdgcn_perm(h23, h01, 0x07060302u) & 0x03030303u);  // byte 3
            }
        }
#pragma unroll
        for (int l = 0; l < 3; ++l)
#pragma unroll
            for (int u = 0; u < 16; ++u)
                acc[l][u] = __builtin_amdgcn_mfma_i32_16x16x64_i8(aa[l], opnd[u], acc[l][u], 0, 0, 0);
    };
#pragma unroll
    for (int s = 0; s < NS; ++s) issue(min(kb0 + s, kb1 - 1), w[s], a[s], fl[s]);
    int kb = kb0;
    for (; kb + NS <= kb1; kb += NS) {
#pragma unroll
        for (int s = 0; s < NS; ++s) {
            compute(w[s], a[s], fl[s]);
            __builtin_amdgcn_sched_barrier(0);
            issue(min(kb + s + NS, kb1 - 1), w[s], a[s], fl[s]);
            __builtin_amdgcn_sched_barrier(0);
        }
    }
    if (kb < kb1) compute(w[0], a[0], fl[0]);
    // epilogue: D[row = column c = 4q + reg][col = n]; MFMA u <-> sample 256 grp + 16 n + u
    const int64_t i0 = grp * 256 + n * 16;
    float kp[16];
#pragma unroll
    for (int u4 = 0; u4 < 4; ++u4) {
        const float4 k4 = *reinterpret_cast<const float4*>(keep + i0 + 4 * u4);
        kp[4 * u4] = k4.x; kp[4 * u4 + 1] = k4.y; kp[4 * u4 + 2] = k4.z; kp[4 * u4 + 3] = k4.w;
    }
#pragma unroll
    for (int reg = 0; reg < 4; ++reg) {
        const int c = 4 * q + reg;
        if (c >= n_sketch) continue;
        const float cr = blockIdx.y == 0 ? (float)corr[c] : 0.f;
        const float inv = 1.0f / wscale[c];
        float* dst = out + (int64_t)c * n_pad + i0;
        int md = mode;
        if (partial && blockIdx.y > 0) {
            dst = partial + ((int64_t)(blockIdx.y - 1) * n_sketch + c) * n_pad + i0;
            md = 0;
        }
#pragma unroll
        for (int u4 = 0; u4 < 4; ++u4) {
            float vv[4];
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                const int u = 4 * u4 + i;
                const float x = (float)acc[0][u][reg] + (float)acc[1][u][reg] * (1.0f / 128.0f) + (float)acc[2][u][reg] * (1.0f / 16384.0f);
                vv[i] = (x - cr) * inv * kp[u];
            }
            float4* d4 = reinterpret_cast<float4*>(dst + 4 * u4);
            if (md == 0) {
                *d4 = make_float4(vv[0], vv[1], vv[2], vv[3]);
            } else if (md == 1) {
                float4 o = *d4;
                o.x += vv[0]; o.y += vv[1]; o.z += vv[2]; o.w += vv[3];
                *d4 = o;
            } else {
                atomicAdd(dst + 4 * u4 + 0, vv[0]);
                atomicAdd(dst + 4 * u4 + 1, vv[1]);
                atomicAdd(dst + 4 * u4 + 2, vv[2]);
                atomicAdd(dst + 4 * u4 + 3, vv[3]);
            }
        }
    }
}

// ------------------------------------------------------------------------------------------------
// small utility kernels
// ------------------------------------------------------------------------------------------------
// pad an entry sub-list to a multiple of 32 (pad rows read row rows[0]; their W is zero)
__global__ void pad_rows_kernel(const int32_t* __restrict__ rows, int n, int n_pad32, int32_t* __restrict__ out) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n_pad32) out[i] = i < n ? rows[i] : rows[0];
}

template <bool VC>
__global__ void block_stats_kernel(const uint8_t* __restrict__ packed, int64_t pitch, int64_t n_rows,
                                   const uint32_t* __restrict__ keepmask, int32_t* __restrict__ out_sum,
                                   int32_t* __restrict__ out_nmiss) {
    // one wave per row
    const int64_t row = (int64_t)blockIdx.x * (blockDim.x >> 6) + (threadIdx.x >> 6);
    if (row >= n_rows) return;
    const int lane = threadIdx.x & 63;
    const uint32_t* rp = reinterpret_cast<const uint32_t*>(packed + row * pitch);
    int sum = 0, nm = 0;
    for (int64_t w = lane; w < pitch / 4; w += 64) {
        const uint32_t km = keepmask[w];
        const uint32_t x = rp[w] & km;
        const uint32_t miss = VC ? (x & (x >> 1) & 0x55555555u) : missing_mask(x);
        const uint32_t v = VC ? x - miss - (miss << 1) : value_code(x) - miss;   // missing -> 0
        sum += __popc(v & 0x55555555u) + 2 * __popc(v & 0xAAAAAAAAu);
        nm += __popc(miss);
    }
    for (int off = 32; off; off >>= 1) {
        sum += __shfl_xor(sum, off);
        nm += __shfl_xor(nm, off);
    }
    if (lane == 0) {
        out_sum[row] = sum;
        out_nmiss[row] = nm;
    }
}

// in-place recode of packed rows: PLINK code -> value code (00->0, 10->1, 11->2, 01->3) or back
__global__ void recode_rows_kernel(uint32_t* __restrict__ p, int64_t n_words, int to_value) {
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n_words; i += (int64_t)gridDim.x * blockDim.x) {
        const uint32_t w = p[i];
        uint32_t o;
        if (to_value) {
            o = value_code(w) | (missing_mask(w) << 1);          // missing (01) -> 01 | 10 = 11
        } else {
            const uint32_t hi = (w >> 1) & 0x55555555u, lo = w & 0x55555555u;
            o = ((hi ^ lo) << 1) | hi;                           // 0->00, 1->10, 2->11, 3->01
        }
        p[i] = o;
    }
}

__global__ void decode_rows_kernel(const uint8_t* __restrict__ packed, int64_t n_rows, int64_t pitch,
                                   float* __restrict__ out) {
    const int64_t gid = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;  // one thread per byte
    if (gid >= n_rows * pitch) return;
    const uint32_t b = packed[gid];
    float4 o;
    float* po = &o.x;
#pragma unroll
    for (int t = 0; t < 4; ++t) {
        const uint32_t code = (b >> (2 * t)) & 3u;
        po[t] = code == 1 ? __builtin_nanf("") : (float)(code - (code >> 1));
    }
    reinterpret_cast<float4*>(out)[gid] = o;
}

__device__ __forceinline__ uint64_t mix64(uint64_t x) {
    x = (x ^ (x >> 30)) * 0xBF58476D1CE4E5B9ull;
    x = (x ^ (x >> 27)) * 0x94D049BB133111EBull;
    return x ^ (x >> 31);
}

// one thread per dword (16 samples); see oracle/synth.py for the CPU twin
__global__ void synth_kernel(uint8_t* __restrict__ packed, int64_t n_rows, int64_t pitch, int64_t n_samples,
                             uint64_t seed, int64_t row0) {
    const int64_t words = pitch / 4;
    const int64_t gid = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (gid >= n_rows * words) return;
    const int64_t r = gid / words, w = gid % words;
    const uint64_t row = (uint64_t)(row0 + r);
    const uint32_t thr = 13u + (uint32_t)(mix64(seed * 0x9E3779B97F4A7C15ull + row * 0xD1B54A32D192ED03ull) % 116ull);
    const uint64_t base = seed * 0x9E3779B97F4A7C15ull + row * 0xD1B54A32D192ED03ull + ((uint64_t)w + 1) * 0x8CB92BA72F3D8DD7ull;
    uint32_t out = 0;
#pragma unroll
    for (int k = 0; k < 4; ++k) {
        const uint64_t h = mix64(base + (uint64_t)(k + 1) * 0x9E3779B97F4A7C15ull);
#pragma unroll
        for (int t = 0; t < 4; ++t) {
            const uint32_t a = (uint32_t)(h >> (16 * t)) & 0xFFu;
            const uint32_t b = (uint32_t)(h >> (16 * t + 8)) & 0xFFu;
            const uint32_t g = (a < thr) + (b < thr);
            const int64_t smp = w * 16 + k * 4 + t;
            const uint32_t code = smp < n_samples ? (g == 0 ? 0u : (g == 1 ? 2u : 3u)) : 0u;
            out |= code << (2 * (k * 4 + t));
        }
    }
    reinterpret_cast<uint32_t*>(packed)[gid] = out;
}

// ------------------------------------------------------------------------------------------------
// engine
// ------------------------------------------------------------------------------------------------
struct pyrhe_engine {
    int device = 0;
    int64_t n_keep = 0, pitch = 0, n_pad = 0, max_entries = 0, sub_snps = 0;
    int precision = 0, parts = 1;
    bool exact = false;   // exact (normal-number) genotype unpack instead of the 1-op subnormal unpack
    bool i8 = false;      // GEMM 1 on int8 limbs (precision i8x3); GEMM 2 then runs the hi+lo fp16 form
    bool vc = false;      // the caller's packed buffers hold the 2-bit VALUE code (pyrhe_recode_rows), not PLINK codes
    int n_sketch = 0, n_norm = 0, n_user = 0, ntl = 0, nto = 0;
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
    size_t ws_bytes = 0, zero_bytes = 0;
    float* P = nullptr;
    uint32_t* wmax = nullptr;
    double* corr = nullptr;
    double* vsq = nullptr;        // [sub_snps][n_norm] squares of the norm columns, summed per segment by norm_reduce()
    float* mu = nullptr;
    float* Wtmp = nullptr;
    float* wscale = nullptr;
    half8* wfrag = nullptr;
    int32_t* rows32 = nullptr;
    float* partial = nullptr;     // [max 3][n_sketch<=32][n_pad] K-split partial sketches (allocated on first use)
    size_t partial_bytes = 0;
    double* dots_partial = nullptr;   // workgroup partials of pyrhe_engine_loo_dots (allocated on first use)
    size_t dots_partial_bytes = 0;
    // Two workspace sets + two internal streams: GEMM 1 (+ mid) of sub-block g+1 runs on stream A while GEMM 2 of
    // sub-block g runs on stream B -- a DRAM-pattern-bound kernel beside an instruction-bound one.  The members
    // above alias the set in use (kernel arguments are captured by value at launch).
    struct WorkSet {
        uint8_t* base; float* P; uint32_t* wmax; double* corr; double* vsq; float* mu; float* Wtmp; float* wscale;
        half8* wfrag; int32_t* rows32;
    } sets[2];
    // dominance (RHE_DOM) extras, allocated on first use: one zeroed region (P2 | wmax2) + vsq2, W2, cmu, zeros
    uint8_t* dom_ws = nullptr;
    size_t dom_zero_bytes = 0;
    float *P2 = nullptr, *Wtmp2 = nullptr, *cmu = nullptr, *zcoef = nullptr;
    uint32_t* wmax2 = nullptr;
    double* vsq2 = nullptr;
    hipStream_t sA = nullptr, sB = nullptr;
    hipEvent_t evIn = nullptr, evA[2] = {nullptr, nullptr}, evB[2] = {nullptr, nullptr}, evDoneA = nullptr, evDoneB = nullptr;
    uint64_t sub_counter = 0;     // sub-blocks enqueued so far (set = counter & 1)
    bool async_join = false;      // true: the caller joins with pyrhe_engine_join instead of at the end of every call
    bool pipeline = false;        // false: GEMM 2 is enqueued on stream A too (serial; clean per-kernel timings)
    bool pending = false;         // work enqueued on the internal streams that the caller's stream has not joined yet
    void use_set(int i) {
        const WorkSet& w = sets[i];
        ws = w.base; P = w.P; wmax = w.wmax; corr = w.corr; vsq = w.vsq; mu = w.mu; Wtmp = w.Wtmp;
        wscale = w.wscale; wfrag = w.wfrag; rows32 = w.rows32;
    }
    // profiling
    bool profile = false;
    struct Ev { int kind; hipEvent_t a, b; };
    std::vector<Ev> evs;
    std::vector<hipEvent_t> pool;   // events are created once and reused across profiled passes
    size_t pool_used = 0;
    hipEvent_t take_event() {
        if (pool_used == pool.size()) {
            hipEvent_t ev;
            (void)hipEventCreate(&ev);
            pool.push_back(ev);
        }
        return pool[pool_used++];
    }
    size_t total_bytes = 0;
};

static constexpr int kMaxTiles = 2;  // 32 logical columns
// Two-stream overlap of GEMM 1 (next sub-block) with GEMM 2 (current): measured +3.5 % at N=500k (the two kernels
// mostly time-share the CUs), at the price of per-kernel timings that include the contention.  Off by default;
// PYRHE_PIPELINE=1 or pyrhe_engine_set_pipeline(e, 1) turns it on.
static bool e_pipeline_default = false;

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
    if (const char* ev = getenv("PYRHE_PIPELINE")) e_pipeline_default = atoi(ev) != 0;
    e->device = device;
    e->pipeline = e_pipeline_default;
    e->n_keep = n_keep;
    e->pitch = pitch;
    e->n_pad = 4 * pitch;
    e->precision = precision;
    e->parts = (precision == PYRHE_PREC_F16X2 || precision == PYRHE_PREC_I8X3) ? 2 : 1;
    e->exact = precision != PYRHE_PREC_F16 && precision != PYRHE_PREC_I8X3_F16;
    e->i8 = precision == PYRHE_PREC_I8X3 || precision == PYRHE_PREC_I8X3_F16;
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
    HIPCHK(hipMalloc(&e->rfrag, e->rfrag_bytes));
    HIPCHK(hipMalloc(&e->rscale, ncolmax * sizeof(float)));
    HIPCHK(hipMalloc(&e->csum, ncolmax * sizeof(double)));
    HIPCHK(hipMalloc(&e->colmax, ncolmax * sizeof(uint32_t)));
    HIPCHK(hipMalloc(&e->keepmask, e->n_pad / 16 * sizeof(uint32_t)));
    HIPCHK(hipMalloc(&e->keep, e->n_pad * sizeof(float)));
    // workspace
    const size_t S = (size_t)sub_snps;
    size_t off = 0;
    auto carve = [&](size_t bytes) { size_t o = off; off += (bytes + 255) / 256 * 256; return o; };
    const size_t oP = carve(S * ncolmax * sizeof(float) * (e->i8 ? 3 : 1));
    static_assert(kSegCols == 16 * kMaxTiles, "segment slots hold one value per logical column");
    const size_t oWmax = carve((size_t)kMaxSeg * kSegCols * sizeof(uint32_t));
    const size_t oCorr = carve((size_t)kMaxSeg * kSegCols * sizeof(double));
    e->zero_bytes = off;
    const size_t oVsq = carve(S * PYRHE_MAX_COLS * sizeof(double));
    const size_t oMu = carve(S * sizeof(float));
    const size_t oWtmp = carve(S * ncolmax * sizeof(float));
    const size_t oWscale = carve((size_t)kMaxSeg * kSegCols * sizeof(float));
    // every segment pads its tail to a 32-entry K block (64 for the int8 GEMM 2)
    const size_t oWfrag = carve((S / 32 + kMaxSeg) * e->parts * kMaxTiles * 64 * sizeof(half8));
    const size_t oRows = carve((S + 64 + 32 * kMaxSeg) * sizeof(int32_t));
    e->ws_bytes = off;
    for (int i = 0; i < 2; ++i) {
        uint8_t* base = nullptr;
        HIPCHK(hipMalloc(&base, e->ws_bytes));
        HIPCHK(hipMemset(base, 0, e->ws_bytes));
        pyrhe_engine::WorkSet& w = e->sets[i];
        w.base = base;
        w.P = (float*)(base + oP);
        w.wmax = (uint32_t*)(base + oWmax);
        w.corr = (double*)(base + oCorr);
        w.vsq = (double*)(base + oVsq);
        w.mu = (float*)(base + oMu);
        w.Wtmp = (float*)(base + oWtmp);
        w.wscale = (float*)(base + oWscale);
        w.wfrag = (half8*)(base + oWfrag);
        w.rows32 = (int32_t*)(base + oRows);
        HIPCHK(hipEventCreateWithFlags(&e->evA[i], hipEventDisableTiming));
        HIPCHK(hipEventCreateWithFlags(&e->evB[i], hipEventDisableTiming));
    }
    e->use_set(0);
    HIPCHK(hipStreamCreateWithFlags(&e->sA, hipStreamNonBlocking));
    HIPCHK(hipStreamCreateWithFlags(&e->sB, hipStreamNonBlocking));
    HIPCHK(hipEventCreateWithFlags(&e->evIn, hipEventDisableTiming));
    HIPCHK(hipEventCreateWithFlags(&e->evDoneA, hipEventDisableTiming));
    HIPCHK(hipEventCreateWithFlags(&e->evDoneB, hipEventDisableTiming));
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
    hipFree(e->keepmask); hipFree(e->keep); hipFree(e->partial); hipFree(e->dom_ws); hipFree(e->dots_partial);
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
        default: return -1;
    }
}

extern "C" int pyrhe_engine_set_rhs(pyrhe_engine_t* e, const float* R, const int32_t* role_host, int n_cols,
                                    const float* keep, void* stream_) {
    if (!e || !R || !role_host || !keep) return fail(PYRHE_EINVAL, "NULL argument");
    if (n_cols < 1 || n_cols > PYRHE_MAX_COLS) return fail(PYRHE_EINVAL, "n_cols %d outside 1..%d", n_cols, PYRHE_MAX_COLS);
    int ns = 0, nn = 0;
    for (int c = 0; c < n_cols; ++c) {
        if (role_host[c] == PYRHE_COL_SKETCH) {
            if (nn) return fail(PYRHE_EINVAL, "sketch columns must precede norm columns (column %d)", c);
            ++ns;
        } else if (role_host[c] == PYRHE_COL_NORM) {
            ++nn;
        } else {
            return fail(PYRHE_EINVAL, "role[%d] = %d is not SKETCH/NORM", c, role_host[c]);
        }
    }
    hipStream_t stream = (hipStream_t)stream_;
    HIPCHK(hipSetDevice(e->device));
    if (e->pending) {      // earlier sketches on the internal streams still read the old operand image
        HIPCHK(hipStreamWaitEvent(stream, e->evDoneA, 0));
        HIPCHK(hipStreamWaitEvent(stream, e->evDoneB, 0));
    }
    e->n_sketch = ns;
    e->n_norm = nn;
    e->n_user = n_cols;
    e->ntl = (n_cols + 1 + 15) / 16;
    e->nto = ns ? (ns + 15) / 16 : 1;
    if (e->ntl > kMaxTiles) return fail(PYRHE_EINVAL, "too many columns");
    HIPCHK(hipMemcpyAsync(e->keep, keep, e->n_pad * sizeof(float), hipMemcpyDeviceToDevice, stream));
    HIPCHK(hipMemsetAsync(e->colmax, 0, 16 * kMaxTiles * sizeof(uint32_t), stream));
    HIPCHK(hipMemsetAsync(e->csum, 0, 16 * kMaxTiles * sizeof(double), stream));
    {
        dim3 grid((unsigned)std::min<int64_t>((e->n_pad + 255) / 256, 1024), n_cols);
        col_absmax_kernel<<<grid, 256, 0, stream>>>(R, e->n_pad, n_cols, e->colmax);
    }
    {
        const int64_t threads = (e->n_pad / 256) * 8 * e->parts * e->ntl * 64;
        const unsigned grid = (unsigned)((threads + 255) / 256);
        if (e->i8) {
            const int64_t th = (e->n_pad / 256) * 4 * 3 * e->ntl * 64;
            prep_rimg_i8_kernel<<<(unsigned)((th + 255) / 256), 256, 0, stream>>>(R, e->keep, e->n_pad, n_cols, e->ntl, e->colmax, e->rscale,
                                                                              (int4v*)e->rfrag, e->csum);
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
            (void)hipEventRecord(a, s);
        }
    }
    ~ProfScope() {
        if (e->profile) {
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

// out[i] += sum_s partial[s][i]   (K-split partial sketches; float4 granularity)
__global__ void add_partials_kernel(float4* __restrict__ out, const float4* __restrict__ partial, int n_parts,
                                    int64_t n4) {
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n4; i += (int64_t)gridDim.x * blockDim.x) {
        float4 o = out[i];
        for (int s = 0; s < n_parts; ++s) {
            const float4 p = partial[(int64_t)s * n4 + i];
            o.x += p.x; o.y += p.y; o.z += p.z; o.w += p.w;
        }
        out[i] = o;
    }
}

// out[i] = sum_j (double) blocks[j * stride + i]: the float64 "aggregate" over per-block float32 sketches
// (base_streaming.py:21-32 / base.py:468-471) in one streaming pass, blocks added in order j = 0, 1, ...
__global__ void sum_blocks_kernel(const float* __restrict__ blocks, int n_blocks, int64_t stride, int64_t n4,
                                  double* __restrict__ out, const double* __restrict__ weights) {
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n4; i += (int64_t)gridDim.x * blockDim.x) {
        double a0 = 0, a1 = 0, a2 = 0, a3 = 0;
        for (int j = 0; j < n_blocks; ++j) {
            const float4 v = reinterpret_cast<const float4*>(blocks + (int64_t)j * stride)[i];
            const double w = weights ? weights[j] : 1.0;
            a0 += w * (double)v.x; a1 += w * (double)v.y; a2 += w * (double)v.z; a3 += w * (double)v.w;
        }
        double* o = out + 4 * i;
        o[0] = a0; o[1] = a1; o[2] = a2; o[3] = a3;
    }
}

// ------------------------------------------------------------------------------------------------
// Leave-one-out inner products (the Gram loops of Base.setup_lhs_rhs_jackknife, base.py:568-628, and GENIE's stochastic
// trace, genie.py:84-94).  A logical row is  v[idx] = tot[idx] - blk[z * stride + idx] * mul[idx % period]  (tot float64,
// blk float32 = one stored block sketch of batch element z, product with mul rounded in float32 like the reference's
// elem_mul; any of the three may be absent).  out[z][a][b] = sum_idx x_a[idx] * y_b[idx] in float64 for a tile of up to
// 8 x 8 rows.  HBM-bound streaming: every thread owns two consecutive idx (16-byte loads), 64 float64 accumulators;
// blockIdx.y = sub-row (idx = y * period + i, no modulo), blockIdx.z = batch element.  Deterministic: the wave folds its
// 64 x 64 accumulators with a transposing butterfly (63 exchanges instead of 384), waves meet in LDS, workgroups write
// partials that loo_dots_reduce_kernel adds in a fixed order.
// ------------------------------------------------------------------------------------------------
struct DotRows {
    const double* tot[8];
    const float* blk[8];
    const float* mul[8];
    long long stride[8];
};

__device__ __forceinline__ double shfl_xor_f64(double v, int mask) {
    int lo = __double2loint(v), hi = __double2hiint(v);
    lo = __shfl_xor(lo, mask);
    hi = __shfl_xor(hi, mask);
    return __hiloint2double(hi, lo);
}

__device__ __forceinline__ void dot_rows_load(const DotRows& R, int n, int64_t idx, int64_t i, int z, double (&v0)[8], double (&v1)[8]) {
#pragma unroll
    for (int a = 0; a < 8; ++a) {
        double x0 = 0.0, x1 = 0.0;
        if (a < n) {
            if (R.tot[a]) {
                const double2 t = *reinterpret_cast<const double2*>(R.tot[a] + idx);
                x0 = t.x; x1 = t.y;
            }
            if (R.blk[a]) {
                float2 b = *reinterpret_cast<const float2*>(R.blk[a] + (int64_t)z * R.stride[a] + idx);
                if (R.mul[a]) {
                    const float2 m = *reinterpret_cast<const float2*>(R.mul[a] + i);
                    b.x *= m.x; b.y *= m.y;
                }
                x0 -= (double)b.x; x1 -= (double)b.y;
            }
        }
        v0[a] = x0; v1[a] = x1;
    }
}

__global__ __launch_bounds__(256) void loo_dots_kernel(const DotRows X, int nx, const DotRows Y, int ny, int64_t period,
                                                       double* __restrict__ partial) {
    __shared__ double s_red[4][64];
    const int z = blockIdx.z;
    const int64_t base = (int64_t)blockIdx.y * period;
    double acc[64];
#pragma unroll
    for (int t = 0; t < 64; ++t) acc[t] = 0.0;
    for (int64_t i = ((int64_t)blockIdx.x * 256 + threadIdx.x) * 2; i < period; i += (int64_t)gridDim.x * 512) {
        double x0[8], x1[8], y0[8], y1[8];
        dot_rows_load(X, nx, base + i, i, z, x0, x1);
        dot_rows_load(Y, ny, base + i, i, z, y0, y1);
#pragma unroll
        for (int a = 0; a < 8; ++a)
#pragma unroll
            for (int b = 0; b < 8; ++b) acc[a * 8 + b] = fma(x1[a], y1[b], fma(x0[a], y0[b], acc[a * 8 + b]));
    }
    // transposing butterfly: after the step with mask m a lane keeps the half of its accumulators selected by its bit m
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
#pragma unroll
    for (int m = 32; m >= 1; m >>= 1) {
        const bool up = (lane & m) != 0;
#pragma unroll
        for (int t = 0; t < m; ++t) {
            const double keep = up ? acc[t + m] : acc[t];
            const double send = up ? acc[t] : acc[t + m];
            acc[t] = keep + shfl_xor_f64(send, m);
        }
    }
    s_red[wave][lane] = acc[0];            // lane L holds accumulator index L = a * 8 + b
    __syncthreads();
    if (threadIdx.x < 64) {
        const double v = (s_red[0][threadIdx.x] + s_red[1][threadIdx.x]) + (s_red[2][threadIdx.x] + s_red[3][threadIdx.x]);
        const int64_t wg = ((int64_t)z * gridDim.y + blockIdx.y) * gridDim.x + blockIdx.x;
        partial[wg * 64 + threadIdx.x] = v;
    }
}

// out[z][a0 + a][b0 + b] = sum over the n_part workgroup partials of batch element z (fixed order)
__global__ void loo_dots_reduce_kernel(const double* __restrict__ partial, int n_part, int n_batch, int nx_t, int ny_t,
                                       int a0, int b0, int nx, int ny, double* __restrict__ out) {
    const int gid = blockIdx.x * blockDim.x + threadIdx.x;
    const int z = gid >> 6, t = gid & 63, a = t >> 3, b = t & 7;
    if (z >= n_batch || a >= nx_t || b >= ny_t) return;
    const double* p = partial + (int64_t)z * n_part * 64 + t;
    double s = 0.0;
    for (int w = 0; w < n_part; ++w) s += p[(int64_t)w * 64];
    out[((int64_t)z * nx + a0 + a) * ny + b0 + b] = s;
}

// ------------------------------------------------------------------------------------------------
// Fast path of the leave-one-out inner products: MANY float32 block rows against a FEW fixed float64 rows
// (projections of the stored block sketches onto the covariates, <block XXz, whole-data XXz>, <block XXUz, UXXz>,
// <block XXz, z>): a tall-skinny GEMM on the float64 matrix cores.  Row r = (batch element z, descriptor a) is the A
// operand of v_mfma_f64_16x16x4_f64 (lane (n, q): row n of the tile, k = q), the fixed rows are the B operand (lane
// (n, q): column n, k = q); the contraction index is free to permute, so a lane reads 8 consecutive elements
// idx = i0 + 8 q + j and MFMA j contracts element j of every lane group.  A wave keeps 4 row tiles (64 rows) x 16
// columns in 32 accumulator registers, so the fixed rows are re-read once per 64 block rows instead of once per 8 as
// in the register-tiled generic kernel.  Rows evaluate to -blk * mul (the block part of a leave-one-out row); the sign
// is restored by the reduction.  Rows / columns past the end compute garbage that the reduction never reads.
// ------------------------------------------------------------------------------------------------
struct BlkRowDev {
    const float* blk;
    const float* mul;
    long long stride;
};
struct FixedRows {
    const double* p[16];
};
constexpr int kDescMax = 128;          // block-row descriptors per launch, passed by value (3 KiB of kernel arguments:
struct BlkRowTable {                   // no upload, nothing for the host to keep alive)
    BlkRowDev r[kDescMax];
};
typedef double double4v __attribute__((ext_vector_type(4)));

__global__ __launch_bounds__(256) void blk_fixed_dots_kernel(const BlkRowTable tab, int nx, int n_rows,
                                                             const FixedRows Y, int ny, int64_t period, int chunk,
                                                             int n_chunks, double* __restrict__ partial) {
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int n = lane & 15, q = lane >> 4;
    const int cx = blockIdx.x * 4 + wave;
    if (cx >= n_chunks) return;                       // no barriers below
    const int sub = blockIdx.y, rg = blockIdx.z;
    const int64_t i_begin = (int64_t)cx * chunk;
    const int64_t i_end = min(i_begin + (int64_t)chunk, period);
    const float* ap[4];
    const float* mp[4];
#pragma unroll
    for (int t = 0; t < 4; ++t) {
        int r = rg * 64 + t * 16 + n;
        if (r >= n_rows) r = 0;
        const int z = r / nx, a = r - z * nx;
        const BlkRowDev d = tab.r[a];
        ap[t] = d.blk + (int64_t)z * d.stride + (int64_t)sub * period;
        mp[t] = d.mul;
    }
    const double* bp = Y.p[0];
#pragma unroll
    for (int c = 1; c < 16; ++c)
        if (n == c && c < ny) bp = Y.p[c];
    bp += (int64_t)sub * period;
    double4v acc[4];
#pragma unroll
    for (int t = 0; t < 4; ++t) acc[t] = (double4v){0.0, 0.0, 0.0, 0.0};
    for (int64_t i0 = i_begin; i0 < i_end; i0 += 32) {
        const int64_t off = i0 + 8 * q;
        double b[8];
#pragma unroll
        for (int h = 0; h < 4; ++h) {
            const double2 v = *reinterpret_cast<const double2*>(bp + off + 2 * h);
            b[2 * h] = v.x; b[2 * h + 1] = v.y;
        }
        float a[4][8];
#pragma unroll
        for (int t = 0; t < 4; ++t) {
            const float4 v0 = *reinterpret_cast<const float4*>(ap[t] + off);
            const float4 v1 = *reinterpret_cast<const float4*>(ap[t] + off + 4);
            float4 m0 = {1.f, 1.f, 1.f, 1.f}, m1 = {1.f, 1.f, 1.f, 1.f};
            if (mp[t]) {
                m0 = *reinterpret_cast<const float4*>(mp[t] + off);
                m1 = *reinterpret_cast<const float4*>(mp[t] + off + 4);
            }
            a[t][0] = v0.x * m0.x; a[t][1] = v0.y * m0.y; a[t][2] = v0.z * m0.z; a[t][3] = v0.w * m0.w;
            a[t][4] = v1.x * m1.x; a[t][5] = v1.y * m1.y; a[t][6] = v1.z * m1.z; a[t][7] = v1.w * m1.w;
        }
#pragma unroll
        for (int j = 0; j < 8; ++j)
#pragma unroll
            for (int t = 0; t < 4; ++t)
                acc[t] = __builtin_amdgcn_mfma_f64_16x16x4f64((double)a[t][j], b[j], acc[t], 0, 0, 0);
    }
    // D: col = n (fixed row), row = q + 4 reg (row of the tile)
    double* dst = partial + (((int64_t)rg * gridDim.y + sub) * n_chunks + cx) * 1024;
#pragma unroll
    for (int t = 0; t < 4; ++t)
#pragma unroll
        for (int reg = 0; reg < 4; ++reg) dst[(t * 16 + q + 4 * reg) * 16 + n] = acc[t][reg];
}

// out[z][a][b0 + b] = - sum over (sub-row, chunk) partials of row r = z * nx + a
__global__ void blk_fixed_reduce_kernel(const double* __restrict__ partial, int n_part, int n_rows, int nxc, int a0, int nx,
                                        int nyt, int b0, int ny, double* __restrict__ out) {
    const int gid = blockIdx.x * blockDim.x + threadIdx.x;
    const int r = gid >> 4, b = gid & 15;            // r = z * nxc + a over the nxc descriptors of this launch
    if (r >= n_rows || b >= nyt) return;
    const double* p = partial + ((int64_t)(r >> 6) * n_part) * 1024 + (r & 63) * 16 + b;
    double s = 0.0;
    for (int w = 0; w < n_part; ++w) s += p[(int64_t)w * 1024];
    const int z = r / nxc, a = r - z * nxc;
    out[((int64_t)z * nx + a0 + a) * ny + b0 + b] = -s;
}

// Self Gram of the block rows of every batch element, out[z][a][b] = <blk_a, blk_b> (the bb term of the expanded
// leave-one-out Gram): with the float64 16x16x4 matrix instruction the A operand (lane (n, q): row n, k = q) and the B
// operand (column n, k = q) of a tile against itself are the SAME register, so every element is loaded once.  A tile
// holds floor(16 / nx) batch elements x nx rows; the products across different batch elements of a tile are computed
// and never read.
__global__ __launch_bounds__(256) void blk_self_gram_kernel(const BlkRowTable tab, int nx, int n_batch,
                                                            int64_t period, int chunk, int n_chunks,
                                                            double* __restrict__ partial) {
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int n = lane & 15, q = lane >> 4;
    const int cx = blockIdx.x * 4 + wave;
    if (cx >= n_chunks) return;
    const int sub = blockIdx.y, tg = blockIdx.z;
    const int zt = 16 / nx;                          // batch elements per tile
    const int64_t i_begin = (int64_t)cx * chunk;
    const int64_t i_end = min(i_begin + (int64_t)chunk, period);
    const float* ap[4];
    const float* mp[4];
#pragma unroll
    for (int t = 0; t < 4; ++t) {
        int z = (tg * 4 + t) * zt + n / nx, a = n % nx;
        if (n >= zt * nx || z >= n_batch) { z = 0; a = 0; }
        const BlkRowDev d = tab.r[a];
        ap[t] = d.blk + (int64_t)z * d.stride + (int64_t)sub * period;
        mp[t] = d.mul;
    }
    double4v acc[4];
#pragma unroll
    for (int t = 0; t < 4; ++t) acc[t] = (double4v){0.0, 0.0, 0.0, 0.0};
    for (int64_t i0 = i_begin; i0 < i_end; i0 += 32) {
        const int64_t off = i0 + 8 * q;
        float a[4][8];
#pragma unroll
        for (int t = 0; t < 4; ++t) {
            const float4 v0 = *reinterpret_cast<const float4*>(ap[t] + off);
            const float4 v1 = *reinterpret_cast<const float4*>(ap[t] + off + 4);
            float4 m0 = {1.f, 1.f, 1.f, 1.f}, m1 = {1.f, 1.f, 1.f, 1.f};
            if (mp[t]) {
                m0 = *reinterpret_cast<const float4*>(mp[t] + off);
                m1 = *reinterpret_cast<const float4*>(mp[t] + off + 4);
            }
            a[t][0] = v0.x * m0.x; a[t][1] = v0.y * m0.y; a[t][2] = v0.z * m0.z; a[t][3] = v0.w * m0.w;
            a[t][4] = v1.x * m1.x; a[t][5] = v1.y * m1.y; a[t][6] = v1.z * m1.z; a[t][7] = v1.w * m1.w;
        }
#pragma unroll
        for (int j = 0; j < 8; ++j)
#pragma unroll
            for (int t = 0; t < 4; ++t) {
                const double v = (double)a[t][j];
                acc[t] = __builtin_amdgcn_mfma_f64_16x16x4f64(v, v, acc[t], 0, 0, 0);
            }
    }
    double* dst = partial + (((int64_t)tg * gridDim.y + sub) * n_chunks + cx) * 1024;
#pragma unroll
    for (int t = 0; t < 4; ++t)
#pragma unroll
        for (int reg = 0; reg < 4; ++reg) dst[(t * 16 + q + 4 * reg) * 16 + n] = acc[t][reg];
}

__global__ void blk_self_gram_reduce_kernel(const double* __restrict__ partial, int n_part, int nx, int n_batch,
                                            double* __restrict__ out) {
    const int gid = blockIdx.x * blockDim.x + threadIdx.x;
    const int z = gid / (nx * nx), ab = gid % (nx * nx), a = ab / nx, b = ab % nx;
    if (z >= n_batch) return;
    const int zt = 16 / nx, tile = z / zt, zl = z % zt;
    const int tg = tile >> 2, t = tile & 3;
    const double* p = partial + ((int64_t)tg * n_part) * 1024 + (t * 16 + zl * nx + a) * 16 + zl * nx + b;
    double s = 0.0;
    for (int w = 0; w < n_part; ++w) s += p[(int64_t)w * 1024];
    out[gid] = s;
}


// One implementation behind pyrhe_engine_sketch_block (additive sketches) and pyrhe_engine_sketch_block_dom
// (additive + dominance sketches, RHE_DOM).
static int sketch_impl(pyrhe_engine_t* e, const uint8_t* packed, int64_t n_rows, const int32_t* entries,
                       const int64_t* offsets_host, int nbin, const float* fill, float* out_sketch, double* out_norm,
                       int32_t* out_colsum, int accumulate, float* out_dom, double* norm_dom, void* stream_) {
    if (!e || !packed || !offsets_host) return fail(PYRHE_EINVAL, "NULL argument");
    if (!e->rhs_set) return fail(PYRHE_ESTATE, "pyrhe_engine_set_rhs has not been called");
    if (nbin < 1) return fail(PYRHE_EINVAL, "nbin must be >= 1");
    if (n_rows < 0) return fail(PYRHE_EINVAL, "n_rows negative");
    if (e->n_sketch && !out_sketch) return fail(PYRHE_EINVAL, "out_sketch is NULL but there are sketch columns");
    if (e->n_norm && !out_norm) return fail(PYRHE_EINVAL, "out_norm is NULL but there are norm columns");
    for (int b = 0; b < nbin; ++b) {
        if (offsets_host[b + 1] < offsets_host[b]) return fail(PYRHE_EINVAL, "offsets not monotone at bin %d", b);
        if (offsets_host[b + 1] - offsets_host[b] > e->max_entries)
            return fail(PYRHE_EINVAL, "bin %d has %lld entries > max_entries %lld", b,
                        (long long)(offsets_host[b + 1] - offsets_host[b]), (long long)e->max_entries);
    }
    if (offsets_host[nbin] > 0 && !entries) return fail(PYRHE_EINVAL, "entries is NULL");
    if (n_rows >= (1 << 24) || e->pitch >= (1 << 24) || (uint64_t)n_rows * (uint64_t)e->pitch >= (1ull << 32))
        return fail(PYRHE_EINVAL, "block of %lld rows x %lld bytes exceeds the 32-bit row offsets of one call; split it",
                    (long long)n_rows, (long long)e->pitch);
    const bool dom = out_dom != nullptr || norm_dom != nullptr;
    if (dom && ((e->n_sketch && !out_dom) || (e->n_norm && !norm_dom))) return fail(PYRHE_EINVAL, "dominance outputs incomplete");
    hipStream_t caller = (hipStream_t)stream_;
    HIPCHK(hipSetDevice(e->device));
    // int8 GEMM 2 (K = 64 SNPs per step, three int32 accumulator sets): correct (agrees with the fp16 hi+lo form to
    // 4e-10 on sigma^2) but measured SLOWER at N=500k -- 40.4 ms vs 32.9 ms per pass: 468 registers pin it to one wave
    // per SIMD and the 16x16 crumb transpose costs 240 v_perm/v_and per step.  Kept opt-in (PYRHE_G2_I8=1).
    static const bool g2i8_env = getenv("PYRHE_G2_I8") && atoi(getenv("PYRHE_G2_I8")) != 0;
    const bool g2i8 = g2i8_env && e->i8 && e->nto == 1 && e->n_sketch > 0 && !dom;
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
    const int ncol_p = 16 * e->ntl;
    const int64_t sk_stride = (int64_t)e->n_sketch * e->n_pad;
    const float pscale = e->exact ? 1.0f : 16777216.0f;
    const int kstep = g2i8 ? 64 : 32;

    if (dom && !e->dom_ws) {
        const size_t S = (size_t)e->sub_snps, nc = 16 * kMaxTiles;
        size_t off = 0;
        auto carve = [&](size_t bytes) { size_t o = off; off += (bytes + 255) / 256 * 256; return o; };
        const size_t oP2 = carve(S * nc * sizeof(float) * (e->i8 ? 3 : 1));
        const size_t oWm = carve(nc * sizeof(uint32_t));
        e->dom_zero_bytes = off;
        const size_t oNa = carve(S * PYRHE_MAX_COLS * sizeof(double));
        const size_t oW2 = carve(S * nc * sizeof(float));
        const size_t oCm = carve(S * sizeof(float));
        const size_t oZc = carve(S * sizeof(float));
        HIPCHK(hipMalloc(&e->dom_ws, off));
        HIPCHK(hipMemset(e->dom_ws, 0, off));
        e->P2 = (float*)(e->dom_ws + oP2);
        e->wmax2 = (uint32_t*)(e->dom_ws + oWm);
        e->vsq2 = (double*)(e->dom_ws + oNa);
        e->Wtmp2 = (float*)(e->dom_ws + oW2);
        e->cmu = (float*)(e->dom_ws + oCm);
        e->zcoef = (float*)(e->dom_ws + oZc);
    }

    // ---- GEMM 1 of one entry list, genotype view gv, into Pbuf (zeroed by the caller)
    auto run_gemm1 = [&](int gv, const int32_t* rows, int n, float* Pbuf) {
        const int gx = (n + 255) / 256;
        const int g1_ch = (e->i8 && e->ntl > 1) ? 1 : 2;  // 256-sample chunks per step (see gemm1_i8_kernel)
        const int n_steps = n_chunks / g1_ch;             // pitch is a multiple of 256 B = 4 chunks
        static const int g1_wgs = getenv("PYRHE_G1_WGS") ? atoi(getenv("PYRHE_G1_WGS")) : 1536;   // target workgroups per launch
        int splits = (g1_wgs + gx - 1) / gx;
        splits = std::max(1, std::min(splits, n_steps / 4 > 0 ? n_steps / 4 : 1));
        int cps = (n_steps + splits - 1) / splits;        // steps per split, even (n_steps is even)
        cps += cps & 1;
        splits = (n_steps + cps - 1) / cps;
        float* saveP = e->P;
        e->P = Pbuf;
        ProfScope ps(e, stream, 0);
        dim3 grid(gx, splits);
        const int ua = splits > 1;
        if (e->i8) {
            const size_t lds = (size_t)2 * g1_ch * 12 * e->ntl * 1024;
#define G1I(NTL_, MISS_, GV_) do { if (e->vc) gemm1_i8_kernel<NTL_, MISS_, GV_, true><<<grid, 256, lds, stream>>>(packed, e->pitch, rows, n, fill, (const int4v*)e->rfrag, n_steps, cps, (int*)Pbuf, ua); \
                                   else gemm1_i8_kernel<NTL_, MISS_, GV_, false><<<grid, 256, lds, stream>>>(packed, e->pitch, rows, n, fill, (const int4v*)e->rfrag, n_steps, cps, (int*)Pbuf, ua); } while (0)
            if (gv == 0) { if (e->ntl == 1) { if (fill) G1I(1, true, 0); else G1I(1, false, 0); } else { if (fill) G1I(2, true, 0); else G1I(2, false, 0); } }
            else         { if (e->ntl == 1) { if (fill) G1I(1, true, 1); else G1I(1, false, 1); } else { if (fill) G1I(2, true, 1); else G1I(2, false, 1); } }
#undef G1I
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
            if (g2i8) {
                const int64_t th = (int64_t)n_kb * 3 * 64;
                mid2_i8_kernel<<<(unsigned)((th + 255) / 256), 256, 0, stream>>>(Wsrc, coef, n, n_kb, e->n_sketch, wmaxp, (int4v*)e->wfrag, e->corr, e->wscale);
                pad_rows_kernel<<<(n_kb * 64 + 255) / 256, 256, 0, stream>>>(rows, n, n_kb * 64, e->rows32);
                if (ndst) norm_reduce_kernel<<<tab.nseg * e->n_norm, 256, 0, stream>>>(nsrc, e->n_norm, ndst, tab);
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
        if (g2i8) iw = 1;
        if (const char* ev = getenv("PYRHE_G2_IW")) iw = atoi(ev) == 2 && e->nto == 1 ? 2 : 1;
        const int gx = (int)((e->n_pad + 1024 * iw - 1) / (1024 * iw));
        int splits = gx >= 192 ? 1 : std::min(16, (512 + gx - 1) / gx);
        if (const char* ev = getenv("PYRHE_G2_SPLITS")) splits = atoi(ev);
        splits = std::max(splits, (n_kb + (g2i8 ? 255 : 511)) / (g2i8 ? 256 : 512));   // row list of a K range lives in LDS: <= 64 KiB
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
#define G2(NTO_, PARTS_, IW_, EX_, GV_) launch_gemm2<NTO_, PARTS_, IW_, EX_, GV_>(e, packed, n_kb, kbps, splits, fill, outp, mode, partial, sB2, gtab)
        if (g2i8) {
            dim3 grid((unsigned)((e->n_pad + 1023) / 1024), splits);
            const size_t lds = (size_t)kbps * 256;
#define L2I(MISS_, VC_) gemm2_i8_kernel<MISS_, VC_><<<grid, 256, lds, sB2>>>(packed, e->pitch, e->rows32, n_kb, kbps, fill, (const int4v*)e->wfrag, e->corr, e->wscale, e->keep, e->n_sketch, e->n_pad, outp, mode, partial)
            if (e->vc) { if (fill) L2I(true, true); else L2I(false, true); }
            else { if (fill) L2I(true, false); else L2I(false, false); }
#undef L2I
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
    const bool segflow = nbin > 1 && !dom && !g2i8;
    if (segflow) {
        std::vector<char> touched(nbin, 0);
        for (int b = 0; b < nbin; ++b)
            if (offsets_host[b + 1] == offsets_host[b] && !accumulate) {
                if (out_sketch) HIPCHK(hipMemsetAsync(out_sketch + b * sk_stride, 0, sk_stride * sizeof(float), sB2));
                if (out_norm && e->n_norm) HIPCHK(hipMemsetAsync(out_norm + (int64_t)b * e->n_norm, 0, e->n_norm * sizeof(double), stream));
            }
        const int64_t total = offsets_host[nbin];
        const int64_t seg_cap = std::min<int64_t>(e->sub_snps, 512 * 32);   // GEMM 2 keeps a segment's row list in LDS
        int b = 0;
        for (int64_t pos = 0; pos < total;) {
            SegTab tab = {};
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
                kb += (take + 31) / 32;
                pos += take;
            }
            tab.start[tab.nseg] = n;
            tab.kb0[tab.nseg] = kb;
            const int32_t* rows = entries + chunk0;
            const int set = (int)(e->sub_counter & 1);
            e->use_set(set);
            if (piped && e->sub_counter >= 2) HIPCHK(hipStreamWaitEvent(e->sA, e->evB[set], 0));
            e->sub_counter++;
            HIPCHK(hipMemsetAsync(e->ws, 0, e->zero_bytes, stream));
            run_gemm1(0, rows, n, e->P);
            {
                ProfScope ps(e, stream, 1);
                mid1_kernel<<<(n + 63) / 64, 256, 0, stream>>>(e->P, pscale, e->i8 ? 1 : 0, n, ncol_p, e->n_sketch, e->n_norm, e->rscale, e->csum,
                                                                 (float)e->n_keep, e->mu, e->Wtmp, e->wmax, e->vsq,
                                                                 out_colsum ? out_colsum + chunk0 : nullptr, tab);
            }
            if (e->n_sketch) {
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
            if (piped) HIPCHK(hipEventRecord(e->evB[set], sB2));
        }
    }
    for (int b = 0; b < nbin && !segflow; ++b) {
        const int64_t n_b = offsets_host[b + 1] - offsets_host[b];
        float* outb = out_sketch ? out_sketch + b * sk_stride : nullptr;
        double* normb = out_norm ? out_norm + (int64_t)b * e->n_norm : nullptr;
        float* outd = out_dom ? out_dom + b * sk_stride : nullptr;
        double* normd = norm_dom ? norm_dom + (int64_t)b * e->n_norm : nullptr;
        if (n_b == 0) {
            if (!accumulate) {
                if (outb) HIPCHK(hipMemsetAsync(outb, 0, sk_stride * sizeof(float), sB2));
                if (normb && e->n_norm) HIPCHK(hipMemsetAsync(normb, 0, e->n_norm * sizeof(double), stream));
                if (outd) HIPCHK(hipMemsetAsync(outd, 0, sk_stride * sizeof(float), sB2));
                if (normd && e->n_norm) HIPCHK(hipMemsetAsync(normd, 0, e->n_norm * sizeof(double), stream));
            }
            continue;
        }
        bool first = true;
        for (int64_t s0 = 0; s0 < n_b; s0 += e->sub_snps) {
            const int n = (int)std::min<int64_t>(e->sub_snps, n_b - s0);
            const int32_t* rows = entries + offsets_host[b] + s0;
            const int n32 = (n + kstep - 1) / kstep * kstep;
            const int n_kb = n32 / kstep;
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
            HIPCHK(hipMemsetAsync(e->ws, 0, e->zero_bytes, stream));
            run_gemm1(0, rows, n, e->P);
            {
                ProfScope ps(e, stream, 1);
                mid1_kernel<<<(n + 63) / 64, 256, 0, stream>>>(e->P, pscale, e->i8 ? 1 : 0, n, ncol_p, e->n_sketch, e->n_norm, e->rscale, e->csum,
                                                                 (float)e->n_keep, e->mu, e->Wtmp, e->wmax, e->vsq,
                                                                 out_colsum ? out_colsum + offsets_host[b] + s0 : nullptr, tab);
                if (e->n_norm && !e->n_sketch) norm_reduce_kernel<<<e->n_norm, 256, 0, stream>>>(e->vsq, e->n_norm, normb, tab);
            }
            if (!e->n_sketch && piped) {
                HIPCHK(hipEventRecord(e->evA[set], e->sA));
                HIPCHK(hipStreamWaitEvent(sB2, e->evA[set], 0));
            }
            if (e->n_sketch) {
                const int rc = run_gemm2(0, e->Wtmp, e->mu, e->wmax, n, n_kb, outb, acc_now, set, rows, e->n_norm ? e->vsq : nullptr,
                                         e->n_norm ? normb : nullptr, tab, false);
                if (rc) return rc;
            }
            if (dom) {
                // ---- dominance: second GEMM 1 with the 2*[g==2] view, then two GEMM 2 launches into out_dom
                HIPCHK(hipMemsetAsync(e->dom_ws, 0, e->dom_zero_bytes, stream));
                run_gemm1(1, rows, n, e->P2);
                HIPCHK(hipMemsetAsync(e->corr, 0, 16 * kMaxTiles * sizeof(double), stream));
                HIPCHK(hipMemsetAsync(e->wmax, 0, 16 * kMaxTiles * sizeof(uint32_t), stream));
                {
                    ProfScope ps(e, stream, 1);
                    mid_dom1_kernel<<<(n + 255) / 256, 256, 0, stream>>>(e->P, e->P2, pscale, e->i8 ? 1 : 0, n, ncol_p, e->n_sketch, e->n_norm, e->rscale,
                                                                         e->csum, (float)e->n_keep, e->cmu, e->Wtmp, e->Wtmp2, e->wmax, e->wmax2,
                                                                         e->vsq2);
                    if (e->n_norm && !e->n_sketch) norm_reduce_kernel<<<e->n_norm, 256, 0, stream>>>(e->vsq2, e->n_norm, normd, tab);
                }
                if (e->n_sketch) {
                    int rc = run_gemm2(0, e->Wtmp, e->cmu, e->wmax, n, n_kb, outd, acc_now, set, rows, e->n_norm ? e->vsq2 : nullptr,
                                       e->n_norm ? normd : nullptr, tab, false);
                    if (rc) return rc;
                    HIPCHK(hipMemsetAsync(e->corr, 0, 16 * kMaxTiles * sizeof(double), stream));
                    tab.acc[0] = 1;
                    rc = run_gemm2(1, e->Wtmp2, e->zcoef, e->wmax2, n, n_kb, outd, true, set, rows, nullptr, nullptr, tab, false);
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
                       nullptr, nullptr, stream_);
}

extern "C" int pyrhe_engine_sketch_block_dom(pyrhe_engine_t* e, const uint8_t* packed, int64_t n_rows,
                                             const int32_t* entries, const int64_t* offsets_host, int nbin,
                                             const float* fill, float* out_sketch, double* out_norm,
                                             float* out_sketch_dom, double* out_norm_dom, int accumulate,
                                             void* stream_) {
    if (!out_sketch_dom && !out_norm_dom) return fail(PYRHE_EINVAL, "dominance outputs are NULL");
    return sketch_impl(e, packed, n_rows, entries, offsets_host, nbin, fill, out_sketch, out_norm, nullptr, accumulate,
                       out_sketch_dom, out_norm_dom, stream_);
}

extern "C" int pyrhe_engine_set_async(pyrhe_engine_t* e, int enable) {
    if (!e) return fail(PYRHE_EINVAL, "NULL engine");
    e->async_join = enable != 0;
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
    if (!e || !packed || !out_sum || !out_nmiss) return fail(PYRHE_EINVAL, "NULL argument");
    if (!e->rhs_set) return fail(PYRHE_ESTATE, "pyrhe_engine_set_rhs has not been called (keep mask unknown)");
    if (n_rows <= 0) return PYRHE_OK;
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
    static const bool no_fast = getenv("PYRHE_DOTS_GENERIC") && atoi(getenv("PYRHE_DOTS_GENERIC")) != 0;
    // ---- block rows against themselves (x and y are the same block-only rows): the self-Gram path
    bool self = period % 32 == 0 && nx == ny && nx <= 16;
    for (int a = 0; a < nx && self; ++a)
        self = !x[a].tot && x[a].blk && x[a].blk == y[a].blk && x[a].mul == y[a].mul && !y[a].tot &&
               x[a].blk_batch_stride == y[a].blk_batch_stride && (((uintptr_t)x[a].blk | (uintptr_t)x[a].mul) & 15) == 0 &&
               x[a].blk_batch_stride % 4 == 0;
    auto table = [&](int a0, int n) {
        BlkRowTable t;
        for (int a = 0; a < kDescMax; ++a)
            t.r[a] = a < n ? BlkRowDev{x[a0 + a].blk, x[a0 + a].mul, (long long)x[a0 + a].blk_batch_stride} : BlkRowDev{nullptr, nullptr, 0};
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
        blk_self_gram_reduce_kernel<<<(n_batch * nx * nx + 255) / 256, 256, 0, stream>>>(e->dots_partial, n_part, nx, n_batch, out);
        HIPCHK(hipGetLastError());
        return PYRHE_OK;
    }
    if (fast && !no_fast) {
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
                blk_fixed_dots_kernel<<<grid, 256, 0, stream>>>(tab, nxc, n_rows, Y, nyt, period, chunk, (int)n_chunks, e->dots_partial);
                blk_fixed_reduce_kernel<<<(n_rows * 16 + 255) / 256, 256, 0, stream>>>(e->dots_partial, n_part, n_rows, nxc, a0, nx, nyt, b0, ny, out);
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

extern "C" int pyrhe_recode_rows(uint8_t* packed, int64_t n_rows, int64_t pitch, int to_coding, void* stream_) {
    if (!packed) return fail(PYRHE_EINVAL, "NULL argument");
    if (pitch <= 0 || pitch % 4) return fail(PYRHE_EINVAL, "pitch must be a positive multiple of 4");
    if (to_coding != PYRHE_CODING_PLINK && to_coding != PYRHE_CODING_VALUE) return fail(PYRHE_EINVAL, "unknown coding %d", to_coding);
    if (n_rows <= 0) return PYRHE_OK;
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

extern "C" int pyrhe_engine_profile(pyrhe_engine_t* e, int enable) {
    if (!e) return fail(PYRHE_EINVAL, "NULL engine");
    e->evs.clear();
    e->pool_used = 0;
    e->profile = enable != 0;
    return PYRHE_OK;
}

extern "C" int pyrhe_engine_profile_read(pyrhe_engine_t* e, double* ms_gemm1, double* ms_mid, double* ms_gemm2,
                                         int64_t* launches) {
    if (!e) return fail(PYRHE_EINVAL, "NULL engine");
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

// sketch_common.h -- part of the MI355X sketch engine (included by sketch_engine.hip, one translation unit).
// Vector types, the genotype decode / unpack helpers shared by both contractions, and the segment table.
#pragma once

typedef _Float16 half8 __attribute__((ext_vector_type(8)));
typedef _Float16 half2v __attribute__((ext_vector_type(2)));
typedef float float4v __attribute__((ext_vector_type(4)));
typedef int int4v __attribute__((ext_vector_type(4)));

union Frag {
    half8 h;
    uint32_t u[4];
};

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
    // 3 (binary 11) -> fill: clear the low bit unless fill is 1, the high bit unless fill is 2.  The two masks depend on
    // the row only (GEMM 1 hoists them out of its loop): shift, and, and, and, shift, xor3 per dword.
    const uint32_t kl = fill == 1 ? 0u : 0x55555555u;
    const uint32_t kh = fill == 2 ? 0u : 0x55555555u;
    const uint32_t t = w & (w >> 1);                        // low bit of a crumb: it holds 3 (other bits: masked below)
    return w ^ (t & kl) ^ ((t & kh) << 1);
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
// int8 digits.  Balanced radix 256: x = a1 + a2 / 256 + a3 / 65536 with every digit in [-128, 127] and |x| <= 126 -- all
// 24 bits of a float32.  Every step is exact in float32: x - rint(x), the multiplications by 256, the carries.
// ------------------------------------------------------------------------------------------------
__device__ __forceinline__ float i8p_scale(float maxabs) {
    if (!(maxabs > 0.0f) || !isfinite(maxabs)) return 1.0f;
    return 126.0f / maxabs;
}
// the first `nd` digits of x rounded to nearest (nd = 3: all of x): x ~ a1 + a2 / 256 [+ a3 / 65536]
__device__ __forceinline__ void i8p_digits_n(float x, int nd, float& a1, float& a2, float& a3) {
    a1 = rintf(x);
    a2 = a3 = 0.0f;
    if (nd == 1) return;
    const float r1 = (x - a1) * 256.0f;               // [-128, 128]
    a2 = rintf(r1);
    if (nd == 3) {
        a3 = rintf((r1 - a2) * 256.0f);               // [-128, 128]
        if (a3 > 127.0f) { a3 -= 256.0f; a2 += 1.0f; }
    }
    if (a2 > 127.0f) { a2 -= 256.0f; a1 += 1.0f; }     // |a1| <= 127 since |x| <= 126
}
__device__ __forceinline__ void i8p_digits(float x, float& a1, float& a2, float& a3) {
    a1 = rintf(x);
    const float r1 = (x - a1) * 256.0f;               // [-128, 128]
    a2 = rintf(r1);
    a3 = rintf((r1 - a2) * 256.0f);                   // [-128, 128]
    if (a3 > 127.0f) { a3 -= 256.0f; a2 += 1.0f; }
    if (a2 > 127.0f) { a2 -= 256.0f; a1 += 1.0f; }     // |a1| <= 127 since |x| <= 126
}
// Rows ("slots") of the int8 GEMM 1's B operand: the (digit, column) pairs packed densely into 16-row tiles instead of
// one tile set per digit.  The `ones` user columns carry up to three digits -- the first nd3 of them three, the first
// nd2 >= nd3 at least two, the rest one (the digit counts do not grow with the column index: sketch columns and
// phenotypes keep all 24 bits, cross columns may carry fewer, PYRHE_COL_DIGITS) -- and the ones column (index `ones`,
// values 0 / 1) one: slots 0..ones = digit 0 of every column, then digit 1 of columns 0..nd2-1, then digit 2 of columns
// 0..nd3-1; ones + 1 + nd2 + nd3 slots in all (k = 10 probes + phenotype: 34 -> 3 tiles; with G x E, 22 columns: 67 -> 5 tiles
// instead of 6; with 10 covariates as one-digit cross columns: 44 -> 3 tiles).
__device__ __host__ __forceinline__ int i8_slot(int digit, int c, int ones, int nd2) {
    return digit == 0 ? c : (digit == 1 ? ones + 1 + c : ones + 1 + nd2 + c);
}
__device__ __forceinline__ double i8_value(const int* __restrict__ Prow, int c, int ones, int nd2, int nd3) {
    double v = (double)Prow[c];
    if (c < nd2) v += (double)Prow[ones + 1 + c] * (1.0 / 256.0);
    if (c < nd3) v += (double)Prow[ones + 1 + nd2 + c] * (1.0 / 65536.0);
    return v;
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
    // the packed int8 GEMM 2 only: workgroup row y of its grid walks segments grp0[y] .. grp1[y]-1 (filled in by the launch)
    unsigned char grp0[kMaxSeg], grp1[kMaxSeg];
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

// pattern3.hip -- read-pattern skeleton for pass 2's one-read kernel (pyrhe_engine_loo_dots_multi), pure reads, no matrix work:
//   A: what blk_multi_dots_kernel asks for -- a wave reads 32 rows x 128 bytes per step (lane (n, q): 2 x 16 B of row n at 32 q)
//   B: the same bytes as 1 KB contiguous per row and instruction (lane l: 16 B at 16 l), 32 rows per wave, then the next 1 KB
//   C: A + the kernel's reads of its fixed rows (two sets of 16 rows x 128 bytes per step: lane (n, q) 2 x 16 B of fixed row n), which
//      every row group re-reads through L1 / L2
//   D: C with ONE set of fixed rows
//   E / F: the fix on trial -- all four waves of a workgroup on ONE chunk (128 rows), both sets of fixed rows of 4 / 8 steps staged ONCE
//      per workgroup in LDS by LDS-DMA (fragment order, 1 KB per instruction), double buffered, one barrier per 4 / 8 steps
//      (a barrier per step -- 16 KB of block rows per workgroup -- runs at 2.3 TB/s)
// over a store of [rows][period] floats (rows 2 MB apart at N = 500k), with workgroups laid out as the kernel's (4 waves: two
// chunks x two 32-row halves of a 64-row group).  Build: hipcc --offload-arch=gfx950 -O3 -o tools/bin/pattern3 tools/pattern3.hip
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdint.h>
#include <vector>
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e_)); return 1; } } while (0)

template <int SUP>
__global__ __launch_bounds__(256, 2) void read_lds_kernel(const float* __restrict__ store, const float* __restrict__ fixed, int64_t period, int n_rows,
                                                          int chunk, int n_chunks, unsigned* __restrict__ sink) {
    extern __shared__ __attribute__((aligned(16))) uint4 fbuf[];         // [slot 2][sub-step SUP][set * 2 + half 4][lane 64]
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int n = lane & 15, q = lane >> 4;
    const int sgs = (n_rows + 127) / 128;                                 // 128-row super groups
    const int sg = blockIdx.x % sgs, cx = blockIdx.x / sgs;
    if (cx >= n_chunks) return;
    const int64_t i_begin = (int64_t)cx * chunk, i_end = min(i_begin + (int64_t)chunk, period);      // chunk: a multiple of 32 * SUP
    const int row0 = sg * 128 + 32 * wave;
    const float* a0 = store + (int64_t)min(row0 + n, n_rows - 1) * period;
    const float* a1 = store + (int64_t)min(row0 + 16 + n, n_rows - 1) * period;
    // this wave's LDS-DMAs of a super-step: per sub-step j the instruction (set = wave >> 1, half = wave & 1)
    const float* fsrc = fixed + (int64_t)((wave >> 1) * 16 + n) * period + 8 * q + 4 * (wave & 1);
    auto issue_f = [&](int64_t i0, int slot) {
#pragma unroll
        for (int j = 0; j < SUP; ++j)
            __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(fsrc + min(i0 + 32 * j, period - 32)),
                                             (__attribute__((address_space(3))) void*)(fbuf + ((slot * SUP + j) * 4 + wave) * 64), 16, 0, 0);
    };
    unsigned acc = 0;
    uint4 g[2][4];
    auto issue_g = [&](int64_t i0, uint4 (&gg)[4]) {
        const int64_t off = i0 + 8 * q;
        gg[0] = *reinterpret_cast<const uint4*>(a0 + off); gg[1] = *reinterpret_cast<const uint4*>(a0 + off + 4);
        gg[2] = *reinterpret_cast<const uint4*>(a1 + off); gg[3] = *reinterpret_cast<const uint4*>(a1 + off + 4);
    };
    const int64_t i_last = i_end - 32;
    issue_f(i_begin, 0);
    issue_g(i_begin, g[0]);
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();
    int slot = 0;
    for (int64_t I = i_begin; I < i_end; I += 32 * SUP, slot ^= 1) {
        issue_f(min(I + 32 * SUP, i_last), slot ^ 1);                     // the next super-step's fixed rows (clamped: a harmless re-read at the end)
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int j = 0; j < SUP; ++j) {
            const int k = j & 1;
            const int64_t i0 = I + 32 * j;
            asm volatile("" ::"v"(g[k][0].x), "v"(g[k][1].x), "v"(g[k][2].x), "v"(g[k][3].x));
            __builtin_amdgcn_sched_barrier(0);
            issue_g(min(i0 + 32, i_last), g[k ^ 1]);
            __builtin_amdgcn_sched_barrier(0);
            const uint4* fb = fbuf + (slot * SUP + j) * 256 + lane;
            const uint4 c0 = fb[0], c1 = fb[64], d0 = fb[128], d1 = fb[192];
            acc += g[k][0].x ^ g[k][1].y ^ g[k][2].z ^ g[k][3].w ^ c0.x ^ c1.y ^ d0.z ^ d1.w;
            __builtin_amdgcn_sched_barrier(0);
        }
        asm volatile("s_waitcnt vmcnt(4)" ::: "memory");                 // the LDS-DMAs of the next super-step have landed; the last four loads may fly
        __builtin_amdgcn_s_barrier();
        asm volatile("" ::: "memory");
    }
    if (acc == 0x12345678u) sink[0] = acc;
}

template <int MODE>
__global__ __launch_bounds__(256, 2) void read_kernel(const float* __restrict__ store, const float* __restrict__ fixed, int64_t period, int n_rows,
                                                      int chunk, int n_chunks, unsigned* __restrict__ sink) {
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int rgs = (n_rows + 63) / 64;
    const int rg = blockIdx.x % rgs, cgp = blockIdx.x / rgs;
    const int cx = cgp * 2 + (wave >> 1);
    if (cx >= n_chunks) return;
    const int64_t i_begin = (int64_t)cx * chunk, i_end = min(i_begin + (int64_t)chunk, period);
    const int row0 = rg * 64 + 32 * (wave & 1);
    unsigned acc = 0;
    if (MODE != 1) {
        const int n = lane & 15, q = lane >> 4;
        const float* a0 = store + (int64_t)min(row0 + n, n_rows - 1) * period;
        const float* a1 = store + (int64_t)min(row0 + 16 + n, n_rows - 1) * period;
        const float* f0 = fixed + (int64_t)n * period;                     // fixed rows: [32][period] floats
        const float* f1 = fixed + (int64_t)(16 + n) * period;
        for (int64_t i0 = i_begin; i0 < i_end; i0 += 32) {
            const int64_t off = i0 + 8 * q;
            const uint4 x0 = *reinterpret_cast<const uint4*>(a0 + off), x1 = *reinterpret_cast<const uint4*>(a0 + off + 4);
            const uint4 y0 = *reinterpret_cast<const uint4*>(a1 + off), y1 = *reinterpret_cast<const uint4*>(a1 + off + 4);
            acc += x0.x ^ x1.y ^ y0.z ^ y1.w;
            if (MODE >= 2) {
                const uint4 c0 = *reinterpret_cast<const uint4*>(f0 + off), c1 = *reinterpret_cast<const uint4*>(f0 + off + 4);
                acc += c0.x ^ c1.y;
                if (MODE == 2) {
                    const uint4 d0 = *reinterpret_cast<const uint4*>(f1 + off), d1 = *reinterpret_cast<const uint4*>(f1 + off + 4);
                    acc += d0.z ^ d1.w;
                }
            }
        }
    } else {
        for (int64_t i0 = i_begin; i0 < i_end; i0 += 256) {
#pragma unroll 4
            for (int r = 0; r < 32; ++r) {
                const float* a = store + (int64_t)min(row0 + r, n_rows - 1) * period + i0;
                const int64_t o = 4 * lane;
                if (i0 + o < i_end) { const uint4 x = *reinterpret_cast<const uint4*>(a + o); acc += x.x ^ x.w; }
            }
        }
    }
    if (acc == 0x12345678u) sink[0] = acc;
}

int main() {
    const int64_t period = 500736;
    const int n_rows = 2048;                                   // 4 GB of float32 rows
    const size_t bytes = (size_t)n_rows * period * 4;
    float *store, *fixed; unsigned* sink;
    CK(hipMalloc(&store, bytes)); CK(hipMalloc(&sink, 4)); CK(hipMemset(store, 1, bytes));
    CK(hipMalloc(&fixed, (size_t)32 * period * 4)); CK(hipMemset(fixed, 2, (size_t)32 * period * 4));
    hipEvent_t a, b; CK(hipEventCreate(&a)); CK(hipEventCreate(&b));
    const int rgs = (n_rows + 63) / 64;
    for (int n_chunks : {31, 62, 124}) {
        const int chunk = (int)(((period + n_chunks - 1) / n_chunks + 511) / 512 * 512);
        const int nc = (int)((period + chunk - 1) / chunk);
        const dim3 grid((unsigned)(rgs * ((nc + 1) / 2)));
        for (int mode = 0; mode < 6; ++mode) {
            float best = 1e9f;
            for (int rep = 0; rep < 4; ++rep) {
                CK(hipEventRecord(a));
                if (mode == 0) read_kernel<0><<<grid, 256>>>(store, fixed, period, n_rows, chunk, nc, sink);
                else if (mode == 1) read_kernel<1><<<grid, 256>>>(store, fixed, period, n_rows, chunk, nc, sink);
                else if (mode == 2) read_kernel<2><<<grid, 256>>>(store, fixed, period, n_rows, chunk, nc, sink);
                else if (mode == 3) read_kernel<3><<<grid, 256>>>(store, fixed, period, n_rows, chunk, nc, sink);
                else if (mode == 4) read_lds_kernel<4><<<dim3((unsigned)(((n_rows + 127) / 128) * nc)), 256, 2 * 4 * 4096>>>(store, fixed, period, n_rows, chunk, nc, sink);
                else read_lds_kernel<8><<<dim3((unsigned)(((n_rows + 127) / 128) * nc)), 256, 2 * 8 * 4096>>>(store, fixed, period, n_rows, chunk, nc, sink);
                CK(hipEventRecord(b)); CK(hipEventSynchronize(b));
                float ms; CK(hipEventElapsedTime(&ms, a, b));
                if (rep) best = ms < best ? ms : best;
            }
            printf("chunks per row %3d (%6d floats) pattern %c: %7.3f ms  %5.2f TB/s\n", nc, chunk, "ABCDEF"[mode], best, bytes / (best * 1e-3) / 1e12);
        }
    }
    return 0;
}

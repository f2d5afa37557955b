// pattern2.hip -- pure-read ceilings of the two contractions' access patterns under two genotype LAYOUTS (round 2):
//   row-major   [row][pitch]              : the .bed body as it is (today)
//   panel-major [panel][row][128 B]       : the matrix cut into column panels of 512 samples, each panel stored row after row
// GEMM 1 (a wave owns 64 rows and sweeps the samples, 128 B of every row per step) reads, panel-major, ONE CONTIGUOUS 8 KB piece
// per wave-step (32 KB per workgroup-step) instead of 64 pieces of 128 B that lie 125 KB apart; GEMM 2 (a wave owns 64 B of
// every row and sweeps the rows) reads contiguous 8 KB pieces per wave pair either way, but the chip as a whole touches 978
// panels at once instead of one 8 MB row range.  No arithmetic; cold (a 2.5 GB flush read in between).
// build: hipcc --offload-arch=gfx950 -O3 -o tools/bin/pattern2 tools/pattern2.hip
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdint.h>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e), __LINE__); return 1; } } while (0)

// GEMM 1: workgroup = 4 waves x 64 rows; step s = 512 samples = 128 B of every row.  row_stride / step_stride in bytes:
// row-major (pitch, 128), panel-major (128, n_rows * 128).
__global__ __launch_bounds__(256) void g1_kernel(const uint8_t* p, int64_t row_stride, int64_t step_stride, int n_rows, int steps_total,
                                                 int steps_per_split, uint32_t* sink) {
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, r = lane & 15, q = lane >> 4;
    const int ebase = (blockIdx.x * 4 + wave) * 64;
    if (ebase >= n_rows) return;
    const int s0 = blockIdx.y * steps_per_split, s1 = min(s0 + steps_per_split, steps_total);
    uint32_t acc = 0;
    const uint8_t* rp[4];
    for (int t = 0; t < 4; ++t) rp[t] = p + (int64_t)min(ebase + 16 * t + r, n_rows - 1) * row_stride + q * 16;
    for (int s = s0; s < s1; ++s)
#pragma unroll
        for (int t = 0; t < 4; ++t)
#pragma unroll
            for (int ch = 0; ch < 2; ++ch) {
                const uint4 g = *(const uint4*)(rp[t] + (int64_t)s * step_stride + ch * 64);
                acc += g.x ^ g.y ^ g.z ^ g.w;
            }
    if (acc == 0x12345678u) sink[0] = acc;
}

// GEMM 2 as the packed int8 kernel runs it: a wave owns 64 B of every row (half a 128-B line; waves w, w^1 share the lines),
// 64 rows per step (lane (n, q): dword n of rows 16q..16q+15), a workgroup barrier every 2 steps.
// row-major: wave g reads bytes [64 g, 64 g + 64) of each row;  panel-major: panel g / 2, half g % 2.
__global__ __launch_bounds__(256) void g2_kernel(const uint8_t* p, int64_t row_stride, int64_t panel_stride, int64_t n_groups, int n_rows,
                                                 int rows_per_split, uint32_t* sink) {
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, n = lane & 15, q = lane >> 4;
    const int64_t grp = (int64_t)blockIdx.x * 4 + wave;
    const bool live = grp < n_groups;
    const int64_t g = live ? grp : 0;
    const uint8_t* cp = p + (panel_stride ? (g >> 1) * panel_stride + (g & 1) * 64 : g * 64) + n * 4;
    const int r0 = blockIdx.y * rows_per_split, r1 = min(r0 + rows_per_split, n_rows);
    uint32_t acc = 0;
    for (int kb = r0 / 64; kb < r1 / 64; ++kb) {
        if ((kb & 1) == 0) __syncthreads();
        if (!live) continue;
#pragma unroll
        for (int j = 0; j < 16; ++j) acc += *(const uint32_t*)(cp + (int64_t)(kb * 64 + q * 16 + j) * row_stride);
    }
    if (acc == 0x12345678u) sink[0] = acc;
}

__global__ void linear_kernel(const uint4* p, int64_t n16, uint32_t* sink) {
    uint32_t acc = 0;
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n16; i += (int64_t)gridDim.x * blockDim.x) { uint4 g = p[i]; acc += g.x ^ g.y ^ g.z ^ g.w; }
    if (acc == 0x12345678u) sink[0] = acc;
}

int main() {
    const int64_t pitch = 125184; const int n_rows = 30048, n_panels = (int)(pitch / 128);     // three 10 016-row blocks, 978 panels
    const int64_t bytes = (int64_t)n_rows * pitch, flush_bytes = (int64_t)20000 * pitch;
    uint8_t* arena; CK(hipMalloc(&arena, bytes + flush_bytes)); CK(hipMemset(arena, 0x5a, bytes + flush_bytes));
    uint32_t* sink; CK(hipMalloc(&sink, 4));
    hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    const double gb = (double)bytes / 1e9;
    auto flush = [&]() { linear_kernel<<<4096, 256>>>((const uint4*)(arena + bytes), flush_bytes / 16, sink); };
    auto timeit = [&](const char* name, auto&& launch) {
        float best = 1e9f;
        for (int rep = 0; rep < 3; ++rep) {
            flush(); (void)hipDeviceSynchronize();
            (void)hipEventRecord(e0); launch(); (void)hipEventRecord(e1); (void)hipEventSynchronize(e1);
            float ms; (void)hipEventElapsedTime(&ms, e0, e1); best = ms < best ? ms : best;
        }
        printf("%-86s %7.1f us  %5.2f TB/s\n", name, best * 1e3, gb / best);
    };
    timeit("linear 16 B/lane grid-stride", [&]() { linear_kernel<<<4096, 256>>>((const uint4*)arena, bytes / 16, sink); });
    for (int wgs : {768, 1536, 3072}) {
        const int gx = (n_rows + 255) / 256;
        int splits = (wgs + gx - 1) / gx;
        int sps = (n_panels + splits - 1) / splits; splits = (n_panels + sps - 1) / sps;
        char nm[128];
        snprintf(nm, sizeof nm, "GEMM 1 pattern, row-major   (64 x 128 B pieces, 125 KB apart, per wave-step), %d WGs", gx * splits);
        timeit(nm, [&]() { g1_kernel<<<dim3(gx, splits), 256>>>(arena, pitch, 128, n_rows, n_panels, sps, sink); });
        snprintf(nm, sizeof nm, "GEMM 1 pattern, panel-major (one contiguous 8 KB piece per wave-step),        %d WGs", gx * splits);
        timeit(nm, [&]() { g1_kernel<<<dim3(gx, splits), 256>>>(arena, 128, (int64_t)n_rows * 128, n_rows, n_panels, sps, sink); });
    }
    for (int ysplit : {1, 3}) {
        const int64_t groups = pitch / 64;
        const int grid = (int)((groups + 3) / 4);
        const int rps = (n_rows / 64 / ysplit) * 64;
        char nm[128];
        snprintf(nm, sizeof nm, "GEMM 2 pattern, row-major   (chip sweeps the same rows), %d row range(s)", ysplit);
        timeit(nm, [&]() { g2_kernel<<<dim3(grid, ysplit), 256>>>(arena, pitch, 0, groups, rps * ysplit, rps, sink); });
        snprintf(nm, sizeof nm, "GEMM 2 pattern, panel-major (978 panels at once, 8 KB per wave pair and step), %d row range(s)", ysplit);
        timeit(nm, [&]() { g2_kernel<<<dim3(grid, ysplit), 256>>>(arena, 128, (int64_t)n_rows * 128, groups, rps * ysplit, rps, sink); });
    }
    return 0;
}

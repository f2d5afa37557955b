// pattern.hip -- pure-read ceilings of the two genotype access patterns on one 10 016-row x 125 184-B block (1.25 GB),
// cold (a 2.5 GB flush read in between), as a function of how many CONTIGUOUS bytes of a row a workgroup touches at
// about the same time.  No arithmetic: what the memory system gives each pattern.
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdint.h>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e), __LINE__); return 1; } } while (0)

// GEMM 1 pattern: a wave owns 64 rows (4 tiles x 16), per step every lane reads LB bytes of its row in each tile
// (4 lanes per row -> 4*LB contiguous bytes per row per wave-step); HW waves of the workgroup take adjacent column
// halves of the same rows (HW = 1: today's kernel), the other waves take other rows.  Grid y splits the row length.
template <int LB, int HW>
__global__ void rows_kernel(const uint8_t* p, int64_t pitch, int n_rows, int steps_total, int steps_per_split, uint32_t* sink) {
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, r = lane & 15, q = lane >> 4;
    const int waves = blockDim.x >> 6, row_groups = waves / HW;
    const int rg = wave % row_groups, half = wave / row_groups;
    const int ebase = (blockIdx.x * row_groups + rg) * 64;
    if (ebase >= n_rows) return;
    const int s0 = blockIdx.y * steps_per_split, s1 = min(s0 + steps_per_split, steps_total);
    uint32_t acc = 0;
    const uint8_t* rp[4];
    for (int t = 0; t < 4; ++t) rp[t] = p + (int64_t)min(ebase + 16 * t + r, n_rows - 1) * pitch + half * (4 * LB) + q * LB;
    for (int s = s0; s < s1; ++s)
#pragma unroll
        for (int t = 0; t < 4; ++t)
#pragma unroll
            for (int b = 0; b < LB / 16; ++b) {
                const uint4 g = *(const uint4*)(rp[t] + (int64_t)s * (4 * LB * HW) + b * 16);
                acc += g.x ^ g.y ^ g.z ^ g.w;
            }
    if (acc == 0x12345678u) sink[0] = acc;
}

// GEMM 2 pattern: lane (n, q) reads W bytes of rows 8q+j (j < 8) per step; a wave covers 16*W contiguous bytes of a
// row, the workgroup's waves adjacent 16*W-byte pieces.
template <int W>
__global__ void cols_kernel(const uint8_t* p, int64_t pitch, int n_rows, uint32_t* sink) {
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, n = lane & 15, q = lane >> 4;
    const int64_t grp = (int64_t)blockIdx.x * (blockDim.x >> 6) + wave;
    if (grp * 16 * W >= pitch) return;
    const uint8_t* cp = p + grp * 16 * W + n * W;
    uint32_t acc = 0;
    for (int kb = 0; kb < n_rows / 32; ++kb)
#pragma unroll
        for (int j = 0; j < 8; ++j) {
            const uint8_t* a = cp + (int64_t)(kb * 32 + q * 8 + j) * pitch;
            if (W == 4) acc += *(const uint32_t*)a;
            else if (W == 8) { uint2 g = *(const uint2*)a; acc += g.x ^ g.y; }
            else { uint4 g = *(const uint4*)a; acc += g.x ^ g.y ^ g.z ^ g.w; }
        }
    if (acc == 0x12345678u) sink[0] = acc;
}

// GEMM 2 pattern as the packed int8 kernel runs it: 16 rows per step (lane (n, q) reads W bytes of rows 16q+j), a
// workgroup barrier every 2 steps (BAR) so that sibling waves stay on the same 128-byte lines; PAIR: waves w, w^1 read the
// SAME 16*W bytes (each would use half of them) -- useful bytes are counted once.
template <int W, bool BAR, bool PAIR>
__global__ void cols16_kernel(const uint8_t* p, int64_t pitch, int n_rows, uint32_t* sink) {
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, n = lane & 15, q = lane >> 4;
    const int wv = blockIdx.x * (blockDim.x >> 6) + wave;
    const int64_t grp = PAIR ? wv >> 1 : wv;
    const bool live = grp * 16 * W < pitch;
    const uint8_t* cp = p + (live ? grp : 0) * 16 * W + n * W;
    uint32_t acc = 0;
    for (int kb = 0; kb < n_rows / 64; ++kb) {
        if (BAR && (kb & 1) == 0) __syncthreads();
        if (!live) continue;
#pragma unroll
        for (int j = 0; j < 16; ++j) {
            const uint8_t* a = cp + (int64_t)(kb * 64 + q * 16 + j) * pitch;
            if (W == 4) acc += *(const uint32_t*)a;
            else { uint2 g = *(const uint2*)a; acc += g.x ^ g.y; }
        }
    }
    if (acc == 0x12345678u) sink[0] = acc;
}

__global__ void linear_kernel(const uint4* p, int64_t n16, uint32_t* sink) {
    uint32_t acc = 0;
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n16; i += (int64_t)gridDim.x * blockDim.x) { uint4 g = p[i]; acc += g.x ^ g.y ^ g.z ^ g.w; }
    if (acc == 0x12345678u) sink[0] = acc;
}

int main() {
    const int64_t pitch = 125184; const int n_rows = 10016;
    const int64_t total_rows = 3 * (int64_t)n_rows;
    uint8_t* arena; CK(hipMalloc(&arena, total_rows * pitch)); CK(hipMemset(arena, 0x5a, total_rows * pitch));
    uint32_t* sink; CK(hipMalloc(&sink, 4));
    hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    const double gb = n_rows * (double)pitch / 1e9;
    auto flush = [&]() { linear_kernel<<<4096, 256>>>((const uint4*)(arena + n_rows * pitch), 2 * n_rows * pitch / 16, sink); };
    auto timeit = [&](const char* name, auto&& launch) {
        float best = 1e9f;
        for (int rep = 0; rep < 3; ++rep) {
            flush(); (void)hipDeviceSynchronize();
            (void)hipEventRecord(e0); launch(); (void)hipEventRecord(e1); (void)hipEventSynchronize(e1);
            float ms; (void)hipEventElapsedTime(&ms, e0, e1); best = ms < best ? ms : best;
        }
        printf("%-64s %7.1f us  %5.2f TB/s\n", name, best * 1e3, gb / best);
    };
    timeit("linear 16 B/lane grid-stride", [&]() { linear_kernel<<<4096, 256>>>((const uint4*)arena, n_rows * pitch / 16, sink); });
    // GEMM 1 pattern
    {
        auto run = [&](auto kern, int lb, int hw, int waves, int wgs_target, const char* name) {
            const int row_groups = waves / hw;
            const int gx = (n_rows + 64 * row_groups - 1) / (64 * row_groups);
            const int steps_total = (int)(pitch / (4 * lb * hw));
            int splits = (wgs_target + gx - 1) / gx; if (splits < 1) splits = 1;
            int sps = (steps_total + splits - 1) / splits; splits = (steps_total + sps - 1) / sps;
            timeit(name, [&]() { kern<<<dim3(gx, splits), 64 * waves>>>(arena, pitch, n_rows, steps_total, sps, sink); });
        };
        run(rows_kernel<32, 1>, 32, 1, 4, 1536, "rows: 4 waves x 64 rows, 128 B/row/step (today), 1536 WGs");
        run(rows_kernel<32, 1>, 32, 1, 4, 3072, "rows: 4 waves x 64 rows, 128 B/row/step, 3072 WGs");
        run(rows_kernel<32, 1>, 32, 1, 4, 768, "rows: 4 waves x 64 rows, 128 B/row/step, 768 WGs");
        run(rows_kernel<32, 2>, 32, 2, 8, 768, "rows: 8 waves = 2 halves x 4 x 64 rows, 256 B/row/WG-step, 768 WGs");
        run(rows_kernel<32, 2>, 32, 2, 8, 1536, "rows: 8 waves = 2 halves x 4 x 64 rows, 256 B/row/WG-step, 1536 WGs");
        run(rows_kernel<32, 4>, 32, 4, 16, 512, "rows: 16 waves = 4 quarters x 4 x 64 rows, 512 B/row/WG-step, 512 WGs");
        run(rows_kernel<64, 1>, 64, 1, 4, 1536, "rows: 4 waves x 64 rows, 256 B/row/wave-step, 1536 WGs");
        run(rows_kernel<64, 2>, 64, 2, 8, 768, "rows: 8 waves, 2 halves, 512 B/row/WG-step, 768 WGs");
        run(rows_kernel<16, 1>, 16, 1, 4, 1536, "rows: 4 waves x 64 rows, 64 B/row/step, 1536 WGs");
    }
    // GEMM 2 pattern
    {
        auto run = [&](auto kern, int w, int waves, const char* name) {
            const int grid = (int)((pitch / (16 * w) + waves - 1) / waves);
            timeit(name, [&]() { kern<<<grid, 64 * waves>>>(arena, pitch, n_rows, sink); });
        };
        run(cols_kernel<4>, 4, 4, "cols: W=4 (64 B/row/wave), 4 waves = 256 B/row/WG (today i8x3)");
        run(cols_kernel<4>, 4, 8, "cols: W=4, 8 waves = 512 B/row/WG");
        run(cols_kernel<4>, 4, 16, "cols: W=4, 16 waves = 1 KB/row/WG");
        run(cols_kernel<8>, 8, 4, "cols: W=8 (128 B/row/wave), 4 waves = 512 B/row/WG (today f16)");
        run(cols_kernel<8>, 8, 8, "cols: W=8, 8 waves = 1 KB/row/WG");
        run(cols_kernel<8>, 8, 16, "cols: W=8, 16 waves = 2 KB/row/WG");
        auto run16 = [&](auto kern, int w, int waves, bool pair, const char* name) {
            const int64_t groups = pitch / (16 * w);
            const int grid = (int)((groups * (pair ? 2 : 1) + waves - 1) / waves);
            timeit(name, [&]() { kern<<<grid, 64 * waves>>>(arena, pitch, n_rows, sink); });
        };
        run16(cols16_kernel<4, false, false>, 4, 4, false, "cols16: W=4, 16 rows/step, no barrier");
        run16(cols16_kernel<4, true, false>, 4, 4, false, "cols16: W=4, 16 rows/step, barrier every 2 steps (packed int8 kernel)");
        run16(cols16_kernel<4, true, true>, 4, 4, true, "cols16: W=4, wave PAIRS read the same 64 B, barrier (useful bytes once)");
        run16(cols16_kernel<4, true, true>, 4, 8, true, "cols16: W=4, wave PAIRS, 8 waves/WG, barrier");
        run16(cols16_kernel<8, true, false>, 8, 4, false, "cols16: W=8, barrier every 2 steps, distinct data per wave");
        run16(cols16_kernel<8, true, true>, 8, 4, true, "cols16: W=8, wave PAIRS read the same 128 B, barrier (useful bytes once)");
        run16(cols16_kernel<8, false, true>, 8, 4, true, "cols16: W=8, wave PAIRS read the same 128 B, no barrier");
        run(cols_kernel<16>, 16, 4, "cols: W=16 (256 B/row/wave), 4 waves = 1 KB/row/WG");
        run(cols_kernel<16>, 16, 8, "cols: W=16, 8 waves = 2 KB/row/WG");
    }
    return 0;
}

// coissue_i8.hip -- issue model of the GEMM 1 (int8) step: 96 v_mfma_i32_16x16x64_i8 (12 independent accumulators)
// and 320 plain 32-bit VALU instructions (and / shift / sub), alone and interleaved as in the kernel (per 12 MFMAs:
// 40 VALU), at 1 and 2 waves per SIMD.
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdint.h>
typedef int int4v __attribute__((ext_vector_type(4)));
template <int MODE>   // 0 MFMA only, 1 VALU only, 2 interleaved
__global__ __launch_bounds__(512) void k(int* out, int iters) {
    int4v a = {(int)threadIdx.x, 1, 2, 3}, b = {1, 1, 1, 1};
    int4v acc[12];
    for (int i = 0; i < 12; ++i) acc[i] = (int4v){0, 0, 0, 0};
    uint32_t v[8];
    for (int i = 0; i < 8; ++i) v[i] = threadIdx.x * 2654435761u + i;
    uint32_t m = 0x03030303u + (threadIdx.x & 1);
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int g = 0; g < 8; ++g) {
            if (MODE == 1 || MODE == 2) {
#pragma unroll
                for (int u = 0; u < 5; ++u)
                    asm volatile("v_lshrrev_b32 %0, 2, %0\n v_and_b32 %1, %1, %4\n v_lshrrev_b32 %2, 4, %2\n v_and_b32 %3, %3, %4\n"
                                 "v_sub_u32 %0, %0, %1\n v_and_b32 %2, %2, %4\n v_lshrrev_b32 %3, 6, %3\n v_and_b32 %1, %1, %4"
                                 : "+v"(v[(u * 4) & 7]), "+v"(v[(u * 4 + 1) & 7]), "+v"(v[(u * 4 + 2) & 7]), "+v"(v[(u * 4 + 3) & 7]), "+v"(m));
            }
            if (MODE == 0 || MODE == 2) {
#pragma unroll
                for (int u = 0; u < 12; ++u) acc[u] = __builtin_amdgcn_mfma_i32_16x16x64_i8(a, b, acc[u], 0, 0, 0);
            }
        }
    }
    int s = 0;
    for (int i = 0; i < 12; ++i) s += acc[i][0] + acc[i][3];
    for (int i = 0; i < 8; ++i) s += (int)v[i];
    out[blockIdx.x * blockDim.x + threadIdx.x] = s;
}
int main() {
    int* o; (void)hipMalloc(&o, 256 * 512 * 4);
    hipEvent_t e0, e1; (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
    const char* names[] = {"96 MFMA i8 16x16x64", "320 VALU (and/shift/sub)", "96 MFMA + 320 VALU interleaved"};
    for (int waves = 1; waves <= 2; ++waves) for (int mode = 0; mode < 3; ++mode) {
        const int iters = 2000;
        auto run = [&]() {
            const int th = 256 * waves;
            if (mode == 0) k<0><<<256, th>>>(o, iters); if (mode == 1) k<1><<<256, th>>>(o, iters); if (mode == 2) k<2><<<256, th>>>(o, iters);
        };
        run(); (void)hipDeviceSynchronize();
        (void)hipEventRecord(e0); run(); (void)hipEventRecord(e1); (void)hipEventSynchronize(e1);
        float ms; (void)hipEventElapsedTime(&ms, e0, e1);
        printf("%d wave(s)/SIMD  %-34s %.1f ns per loop body per wave -> %.1f ns per body per SIMD\n", waves, names[mode], ms * 1e6 / iters, ms * 1e6 / iters / waves);
    }
    return 0;
}

// coissue.hip -- do VALU unpack instructions and MFMAs overlap on a CDNA4 SIMD?  Times a loop body of
// NM matrix instructions (v_mfma_f32_16x16x32_f16, independent accumulators) and NV packed VALU instructions, alone and
// interleaved, at 1 and 2 waves per SIMD.  Wall time by hipEvents over a full-chip grid.
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdint.h>
typedef _Float16 half8 __attribute__((ext_vector_type(8)));
typedef float float4v __attribute__((ext_vector_type(4)));
template <int MODE>   // 0: MFMA only, 1: VALU only, 2: interleaved (4 MFMA then 20 VALU), 3: interleaved fine (1 MFMA, 5 VALU)
__global__ __launch_bounds__(1024) void k(float* out, int iters) {
    half8 a, b;
    for (int i = 0; i < 8; ++i) { a[i] = (_Float16)(threadIdx.x & 3); b[i] = (_Float16)1; }
    float4v acc[8];
    for (int i = 0; i < 8; ++i) acc[i] = (float4v){0, 0, 0, 0};
    uint32_t v[8];
    for (int i = 0; i < 8; ++i) v[i] = threadIdx.x * 2654435761u + i;
    uint32_t m = 0x00030003u + (threadIdx.x & 1);
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int g = 0; g < 8; ++g) {
            if (MODE == 0 || MODE == 2) {
#pragma unroll
                for (int u = 0; u < 4; ++u) acc[(g * 4 + u) & 7] = __builtin_amdgcn_mfma_f32_16x16x32_f16(a, b, acc[(g * 4 + u) & 7], 0, 0, 0);
            }
            if (MODE == 1 || MODE == 2) {
#pragma unroll
                for (int u = 0; u < 5; ++u)
                    asm volatile("v_and_or_b32 %0, %0, %4, %4\n v_pk_fma_f16 %1, %1, %4, %4\n v_and_or_b32 %2, %2, %4, %4\n v_pk_fma_f16 %3, %3, %4, %4"
                                 : "+v"(v[(u * 4) & 7]), "+v"(v[(u * 4 + 1) & 7]), "+v"(v[(u * 4 + 2) & 7]), "+v"(v[(u * 4 + 3) & 7]), "+v"(m));
            }
            if (MODE == 3) {
#pragma unroll
                for (int u = 0; u < 4; ++u) {
                    acc[(g * 4 + u) & 7] = __builtin_amdgcn_mfma_f32_16x16x32_f16(a, b, acc[(g * 4 + u) & 7], 0, 0, 0);
                    asm volatile("v_and_or_b32 %0, %0, %4, %4\n v_pk_fma_f16 %1, %1, %4, %4\n v_and_or_b32 %2, %2, %4, %4\n v_pk_fma_f16 %3, %3, %4, %4\n v_and_or_b32 %0, %0, %4, %4"
                                 : "+v"(v[(u * 4) & 7]), "+v"(v[(u * 4 + 1) & 7]), "+v"(v[(u * 4 + 2) & 7]), "+v"(v[(u * 4 + 3) & 7]), "+v"(m));
                }
            }
        }
    }
    float s = 0;
    for (int i = 0; i < 8; ++i) s += acc[i][0] + acc[i][1] + (float)v[i];
    out[blockIdx.x * blockDim.x + threadIdx.x] = s;
}
int main() {
    float* o; hipMalloc(&o, 256 * 512 * 4 * 4);
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    const char* names[] = {"32 MFMA", "160 VALU", "32 MFMA + 160 VALU (4/20 groups)", "32 MFMA + 160 VALU (1/5 fine)"};
    for (int waves = 1; waves <= 4; ++waves) for (int mode = 0; mode < 4; ++mode) {
        const int iters = 4000;
        auto run = [&]() {
            const int th = 256 * waves;
            if (mode == 0) k<0><<<256, th>>>(o, iters); if (mode == 1) k<1><<<256, th>>>(o, iters);
            if (mode == 2) k<2><<<256, th>>>(o, iters); if (mode == 3) k<3><<<256, th>>>(o, iters);
        };
        run(); hipDeviceSynchronize();
        hipEventRecord(e0); run(); hipEventRecord(e1); hipEventSynchronize(e1);
        float ms; hipEventElapsedTime(&ms, e0, e1);
        printf("%d wave(s)/SIMD  %-36s %.1f ns per loop body per wave -> %.1f ns per body per SIMD\n", waves, names[mode], ms * 1e6 / iters, ms * 1e6 / iters / waves);
    }
    return 0;
}

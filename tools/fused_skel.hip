// fused_skel.hip -- data-movement + synchronisation skeleton of the single-read ("fused") sketch kernel.
// Question: can 245 persistent workgroups (one per CU, each owning a 2048-sample column slice) stream every packed
// row ONCE through an LDS-DMA ring while (a) adding a 64x16 float partial per 64-row tile into a global buffer with
// device-scope atomics and (b) waiting, two tiles later, on a per-tile arrival counter + re-reading the reduced
// 4 KiB -- at >= 5 TB/s?   No MFMA work here: this bounds the fused design from above.
// build: hipcc --offload-arch=gfx950 -O3 -o tools/fused_skel tools/fused_skel.hip
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdint.h>
#include <vector>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e), __LINE__); return 1; } } while (0)

constexpr int TS = 64;            // rows per tile
constexpr int SLICE = 512;        // bytes per row per CU (2048 samples)
constexpr int RING = 4;

__device__ __forceinline__ void dma16(const void* gsrc, void* lds_dst_wave_base) {
    // 64 lanes x 16 B -> 1 KiB contiguous at the wave-uniform LDS base
    __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)gsrc,
                                     (__attribute__((address_space(3))) void*)lds_dst_wave_base, 16, 0, 0);
}

template <int MODE>   // 0: stream only, 1: + atomics, 2: + atomics + counter wait + P read (lag 2)
__global__ __launch_bounds__(256) void skel(const uint8_t* __restrict__ packed, int64_t pitch, int n_tiles,
                                            float* __restrict__ Pbuf, unsigned* __restrict__ counters,
                                            unsigned* __restrict__ fail, float* __restrict__ sink) {
    extern __shared__ __attribute__((aligned(16))) uint8_t lds[];     // RING x TS x SLICE
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int ncu = gridDim.x;
    const uint8_t* base = packed + (int64_t)blockIdx.x * SLICE;
    auto issue = [&](int t) {
        // tile t: 64 rows x 512 B = 32 KiB = 32 DMA instructions (1 KiB = 2 rows each); 8 per wave
        uint8_t* slot = lds + (t % RING) * (TS * SLICE);
#pragma unroll
        for (int i = 0; i < 8; ++i) {
            const int piece = wave * 8 + i;              // rows 2*piece, 2*piece+1
            const int row = 2 * piece + (lane >> 5);
            const int chunk = (lane & 31) ^ (row & 15);  // source swizzle (reader undoes it)
            dma16(base + ((int64_t)t * TS + row) * pitch + chunk * 16, slot + piece * 1024);
        }
    };
    float acc = 0.f;
    for (int t = 0; t < RING - 1 && t < n_tiles; ++t) issue(t);
    for (int t = 0; t < n_tiles; ++t) {
        if (t + RING - 1 < n_tiles) issue(t + RING - 1);
        // wait until tile t has landed: at most (RING-1)*8 younger DMA instructions may stay in flight
        if (t + RING - 1 < n_tiles) asm volatile("s_waitcnt vmcnt(24)" ::: "memory");
        else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __builtin_amdgcn_s_barrier();
        // "GEMM 1": touch the tile
        const uint8_t* slot = lds + (t % RING) * (TS * SLICE);
        const uint4 v = *reinterpret_cast<const uint4*>(slot + threadIdx.x * 128);
        acc += __uint_as_float((v.x ^ v.y ^ v.z ^ v.w) & 0x007fffffu);
        if (MODE >= 1) {
            // 64x16 floats per tile per CU: 1024 atomics -> 4 per thread, 256 contiguous bytes per wave-instruction
            float* p = Pbuf + (int64_t)t * 1024 + threadIdx.x;
#pragma unroll
            for (int i = 0; i < 4; ++i) atomicAdd(p + i * 256, 1.0f);
        }
        if (MODE >= 2) {
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");     // my atomics have been issued & acknowledged
            __builtin_amdgcn_s_barrier();
            if (threadIdx.x == 0) __hip_atomic_fetch_add(&counters[t], 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            if (t >= 2) {
                // consume tile t-2's reduced partial: poll its counter (bounded), then read the 4 KiB with sc1 loads
                if (threadIdx.x == 0) {
                    unsigned spins = 0;
                    while (__hip_atomic_load(&counters[t - 2], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) < (unsigned)ncu) {
                        __builtin_amdgcn_s_sleep(2);
                        if (++spins > 2000000u) { atomicAdd(fail, 1u); break; }
                    }
                }
                __builtin_amdgcn_s_barrier();
                const float* q = Pbuf + (int64_t)(t - 2) * 1024 + threadIdx.x;
#pragma unroll
                for (int i = 0; i < 4; ++i) acc += __hip_atomic_load(q + i * 256, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            }
        }
        __builtin_amdgcn_s_barrier();    // everybody is done with slot t before it is refilled (issue of t+RING)
    }
    if (acc == 123.456f) sink[0] = acc;
}

int main() {
    const int64_t pitch = 125184;
    const int n_rows = 10240, n_tiles = n_rows / TS;       // one 1.28 GB block
    const int ncu = (int)(pitch / SLICE);                   // 244 slices
    uint8_t* g; CK(hipMalloc(&g, (size_t)n_rows * pitch)); CK(hipMemset(g, 0x5a, (size_t)n_rows * pitch));
    float* P; CK(hipMalloc(&P, (size_t)n_tiles * 1024 * 4));
    unsigned *cnt, *fail; CK(hipMalloc(&cnt, n_tiles * 4)); CK(hipMalloc(&fail, 4));
    float* sink; CK(hipMalloc(&sink, 4));
    hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    const size_t lds = RING * TS * SLICE;
    CK(hipFuncSetAttribute((const void*)skel<0>, hipFuncAttributeMaxDynamicSharedMemorySize, lds));
    CK(hipFuncSetAttribute((const void*)skel<1>, hipFuncAttributeMaxDynamicSharedMemorySize, lds));
    CK(hipFuncSetAttribute((const void*)skel<2>, hipFuncAttributeMaxDynamicSharedMemorySize, lds));
    for (int mode = 0; mode < 3; ++mode) {
        for (int rep = 0; rep < 3; ++rep) {
            CK(hipMemset(P, 0, (size_t)n_tiles * 1024 * 4)); CK(hipMemset(cnt, 0, n_tiles * 4)); CK(hipMemset(fail, 0, 4));
            CK(hipDeviceSynchronize());
            hipEventRecord(e0);
            if (mode == 0) skel<0><<<ncu, 256, lds>>>(g, pitch, n_tiles, P, cnt, fail, sink);
            if (mode == 1) skel<1><<<ncu, 256, lds>>>(g, pitch, n_tiles, P, cnt, fail, sink);
            if (mode == 2) skel<2><<<ncu, 256, lds>>>(g, pitch, n_tiles, P, cnt, fail, sink);
            hipEventRecord(e1); CK(hipDeviceSynchronize());
            float ms; hipEventElapsedTime(&ms, e0, e1);
            unsigned f; CK(hipMemcpy(&f, fail, 4, hipMemcpyDeviceToHost));
            float p0; CK(hipMemcpy(&p0, P, 4, hipMemcpyDeviceToHost));
            const double gb = (double)n_rows * ncu * SLICE / 1e9;
            printf("mode %d rep %d: %.1f us  %.2f TB/s  (timeouts %u, P[0]=%.0f)\n", mode, rep, ms * 1e3, gb / ms, f, p0);
        }
    }
    return 0;
}

// ic_pipeline.hip -- round 3 gate for the Infinity-Cache pipeline: can the SECOND read of the genotypes be
// served by the 256 MiB Infinity Cache WHILE the first read of the next chunk streams from HBM -- i.e. is the aggregate of the two
// contractions' read streams above what HBM alone gives (5.9 / 6.4 TB/s for the patterns alone, tools/pattern2.hip)?
//
// Pure reads, no arithmetic.  A persistent grid (every workgroup resident) walks PHASES separated by a grid-wide barrier; the
// genotype rows ([row][pitch], the .bed layout at N = 500k) are cut into chunks of `chunk_rows` rows (32 .. 128 MB).
//   role A ("GEMM 1"): a workgroup item = 256 rows x 4 steps of 128 B (a wave owns 64 rows, 128 B of every row per step)
//   role B ("GEMM 2"): a workgroup item = 256 B of every row of the chunk (a wave owns 64 B of every row, 64 rows per step, a
//                      workgroup barrier every two steps like gemm2_i8p_kernel)
// modes:  pipe   in phase p half the workgroups stream chunk p (A, from HBM), the other half re-read chunk p - 1 (B)
//         serial in phase 2c all workgroups run A on chunk c, in phase 2c + 1 all run B on chunk c (the re-read alone)
//         a-only / b-only  all workgroups run A (B) on chunk p: the patterns alone, cold
// build: hipcc --offload-arch=gfx950 -O3 -o tools/bin/ic_pipeline tools/ic_pipeline.hip
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdint.h>
#include <stdlib.h>
#include <string.h>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e), __LINE__); return 1; } } while (0)

__device__ __forceinline__ void grid_barrier(unsigned* counter, unsigned target) {
    __syncthreads();
    if (threadIdx.x == 0) {
        __hip_atomic_fetch_add(counter, 1u, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_AGENT);
        // bounded poll (about a second): a grid that is not fully resident ends with wrong timings, never with a hung GPU
        for (long it = 0; it < (1l << 24) && __hip_atomic_load(counter, __ATOMIC_ACQUIRE, __HIP_MEMORY_SCOPE_AGENT) < target; ++it)
            __builtin_amdgcn_s_sleep(2);
    }
    __syncthreads();
}

// role A over chunk rows [r0, r0 + nrows): items = (row block of 256, 4-step range); `me` of `nwg` workgroups
__device__ __forceinline__ uint32_t role_a(const uint8_t* p, int64_t pitch, int r0, int nrows, int steps_total, int me, int nwg) {
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, r = lane & 15, q = lane >> 4;
    const int nrb = (nrows + 255) / 256, nsr = (steps_total + 3) / 4;
    uint32_t acc = 0;
    for (int item = me; item < nrb * nsr; item += nwg) {
        const int rb = item % nrb, sr = item / nrb;
        const int ebase = r0 + rb * 256 + wave * 64;
        const int s0 = sr * 4, s1 = min(s0 + 4, steps_total);
        const uint8_t* rp[4];
#pragma unroll
        for (int t = 0; t < 4; ++t) rp[t] = p + (int64_t)min(ebase + 16 * t + r, r0 + nrows - 1) * pitch + q * 16;
        for (int s = s0; s < s1; ++s)
#pragma unroll
            for (int t = 0; t < 4; ++t)
#pragma unroll
                for (int ch = 0; ch < 2; ++ch) {
                    const uint4 g = *(const uint4*)(rp[t] + (int64_t)s * 128 + ch * 64);
                    acc += g.x ^ g.y ^ g.z ^ g.w;
                }
    }
    return acc;
}

// role B over chunk rows [r0, r0 + nrows) (nrows a multiple of 128): items = column group of 256 B (4 waves x 64 B)
__device__ __forceinline__ uint32_t role_b(const uint8_t* p, int64_t pitch, int r0, int nrows, int me, int nwg) {
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, n = lane & 15, q = lane >> 4;
    const int ncg = (int)(pitch / 256);
    uint32_t acc = 0;
    for (int item = me; item < ncg; item += nwg) {
        const uint8_t* cp = p + (int64_t)r0 * pitch + (int64_t)item * 256 + wave * 64 + n * 4;
        for (int kb = 0; kb < nrows / 64; ++kb) {
            if ((kb & 1) == 0) __syncthreads();
#pragma unroll
            for (int j = 0; j < 16; ++j) acc += *(const uint32_t*)(cp + (int64_t)(kb * 64 + q * 16 + j) * pitch);
        }
    }
    return acc;
}

// mode 0 pipe, 1 serial, 2 a-only, 3 b-only.  split_a: of every 8 consecutive workgroups (one per XCD) groups, how the roles
// alternate: workgroup w takes role A when ((w / 8) % den) < num  (num / den of the grid streams, the rest re-reads).
__global__ __launch_bounds__(256) void pipe_kernel(const uint8_t* p, int64_t pitch, int n_chunks, int chunk_rows, int steps_total, int mode,
                                                   int num, int den, unsigned* counter, uint32_t* sink) {
    const int nwg = gridDim.x, w = blockIdx.x;
    uint32_t acc = 0;
    unsigned phase = 0;
    if (mode == 0) {
        const int grp = w / 8, xcd = w % 8;
        const bool is_a = (grp % den) < num;
        const int per = nwg / 8 / den;                       // full periods of `den` groups
        const int n_a = per * num * 8, n_b = per * (den - num) * 8;
        const int me = is_a ? ((grp / den) * num + grp % den) * 8 + xcd : ((grp / den) * (den - num) + (grp % den - num)) * 8 + xcd;
        for (int ph = 0; ph <= n_chunks; ++ph) {
            if (is_a) { if (ph < n_chunks) acc += role_a(p, pitch, ph * chunk_rows, chunk_rows, steps_total, me, n_a); }
            else if (ph >= 1) acc += role_b(p, pitch, (ph - 1) * chunk_rows, chunk_rows, me, n_b);
            grid_barrier(counter, ++phase * nwg);
        }
    } else if (mode == 1) {
        for (int c = 0; c < n_chunks; ++c) {
            acc += role_a(p, pitch, c * chunk_rows, chunk_rows, steps_total, w, nwg);
            grid_barrier(counter, ++phase * nwg);
            acc += role_b(p, pitch, c * chunk_rows, chunk_rows, w, nwg);
            grid_barrier(counter, ++phase * nwg);
        }
    } else {
        for (int c = 0; c < n_chunks; ++c) {
            acc += mode == 2 ? role_a(p, pitch, c * chunk_rows, chunk_rows, steps_total, w, nwg) : role_b(p, pitch, c * chunk_rows, chunk_rows, w, nwg);
            grid_barrier(counter, ++phase * nwg);
        }
    }
    if (acc == 0x12345678u) sink[0] = acc;
}

__global__ void linear_kernel(const uint4* p, int64_t n16, uint32_t* sink) {
    uint32_t acc = 0;
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n16; i += (int64_t)gridDim.x * blockDim.x) { uint4 g = p[i]; acc += g.x ^ g.y ^ g.z ^ g.w; }
    if (acc == 0x12345678u) sink[0] = acc;
}

int main(int argc, char** argv) {
    const int64_t pitch = 125184;                       // N = 500k
    const int steps_total = (int)(pitch / 128);         // 978
    const int64_t total_rows = 30720;                   // 3.85 GB of genotypes: far beyond the 256 MiB cache
    const int64_t bytes = total_rows * pitch, flush_bytes = (int64_t)8192 * pitch;
    uint8_t* arena; CK(hipMalloc(&arena, bytes + flush_bytes)); CK(hipMemset(arena, 0x5a, bytes + flush_bytes));
    uint32_t* sink; CK(hipMalloc(&sink, 4));
    unsigned* counter; CK(hipMalloc(&counter, 4));
    hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    int ncu = 0; CK(hipDeviceGetAttribute(&ncu, hipDeviceAttributeMultiprocessorCount, 0));
    printf("ic_pipeline: %d CUs, pitch %lld B, %lld rows = %.2f GB\n", ncu, (long long)pitch, (long long)total_rows, bytes / 1e9);
    auto run = [&](const char* name, int mode, int chunk_rows, int wg_per_cu, int num, int den, double reads) -> int {
        const int n_chunks = (int)(total_rows / chunk_rows);
        const int nwg = ncu * wg_per_cu;
        float best = 1e9f;
        for (int rep = 0; rep < 3; ++rep) {
            linear_kernel<<<4096, 256>>>((const uint4*)(arena + bytes), flush_bytes / 16, sink);
            CK(hipMemset(counter, 0, 4));
            CK(hipDeviceSynchronize());
            CK(hipEventRecord(e0));
            pipe_kernel<<<nwg, 256>>>(arena, pitch, n_chunks, chunk_rows, steps_total, mode, num, den, counter, sink);
            CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1));
            float ms; CK(hipEventElapsedTime(&ms, e0, e1)); best = ms < best ? ms : best;
        }
        const double gb = (double)n_chunks * chunk_rows * pitch * reads / 1e9;
        printf("%-34s chunk %4d rows = %5.1f MB, %d WG/CU, A:B = %d:%d  %8.1f us  aggregate %5.2f TB/s  (one-read equivalent %5.2f TB/s)\n", name, chunk_rows,
               chunk_rows * pitch / 1e6, wg_per_cu, num, den - num, best * 1e3, gb / best, gb / reads / best);
        return 0;
    };
    for (int wpc : {2, 4}) {
        if (run("a-only (GEMM 1 pattern, HBM)", 2, 1024, wpc, 1, 2, 1.0)) return 1;
        if (run("b-only (GEMM 2 pattern, HBM)", 3, 1024, wpc, 1, 2, 1.0)) return 1;
    }
    for (int rows : {256, 512, 768, 1024, 1536}) {
        for (int wpc : {2, 4}) {
            if (run("serial (A then B on the chunk)", 1, rows, wpc, 1, 2, 2.0)) return 1;
            if (run("pipe   (A on c beside B on c-1)", 0, rows, wpc, 1, 2, 2.0)) return 1;
        }
    }
    // uneven role splits at the two chunk sizes SURVEY 7.3(3) named
    for (int rows : {512, 1024}) {
        if (run("pipe, 3 of 4 workgroups stream", 0, rows, 4, 3, 4, 2.0)) return 1;
        if (run("pipe, 1 of 4 workgroups stream", 0, rows, 4, 1, 4, 2.0)) return 1;
    }
    return 0;
}

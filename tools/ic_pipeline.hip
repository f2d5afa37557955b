// ic_pipeline.hip -- round 3 gate for the Infinity-Cache pipeline: can the SECOND read of the genotypes be
// served by the 256 MiB Infinity Cache WHILE the first read of the next chunk streams from HBM -- i.e. is the aggregate of the two
// contractions' read streams above what HBM alone gives (5.9 / 6.4 TB/s for the patterns alone, tools/pattern2.hip)?
//
// Pure reads, no arithmetic.  ONE persistent launch (every workgroup resident); the genotype rows ([row][pitch], the .bed layout
// at N = 500k) are cut into chunks of `chunk_rows` rows (32 .. 192 MB) that both roles walk in order:
//   role A ("GEMM 1"): item = 256 rows x 4 steps of 128 B (a wave owns 64 rows, 128 B of every row per step), from HBM
//   role B ("GEMM 2"): item = 256 B of every row of the chunk (a wave owns 64 B of every row, 64 rows per step, a workgroup
//                      barrier every two steps like gemm2_i8p_kernel) -- the chunk role A has just streamed
// Pacing by ONE progress word per role, written by that role's workgroup 0 only ("I have finished chunk c": the workgroups of a
// role advance at the same rate) and polled by everybody else: role B starts chunk c when role A's leader is past it, role A
// starts chunk c when role B's leader is past chunk c - 1 - LAG, so the two read fronts stay within LAG + 1 chunks of each other
// -- inside the cache if (LAG + 2) x chunk < 256 MiB.  (Two earlier protocols measured themselves instead of the memory system:
// a grid-wide barrier per chunk, 60 - 150 us per phase of 1024 polling workgroups; and exact per-chunk arrival counters -- 512
// same-address atomics per flag, two flags per chunk: 150 us per chunk against 45 us of reads.  A real pipeline would need the
// exact dependency -- GEMM 2 of a chunk needs its finished X^T R -- through per-XCD sharded counters; this gate only asks what
// the memory system gives when it is paced for free.)
// modes: pipe (half / half or other splits), a-only, b-only (all workgroups in one role, no flags: the patterns alone, cold).
// build: hipcc --offload-arch=gfx950 -O3 -o tools/bin/ic_pipeline tools/ic_pipeline.hip
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdint.h>
#include <stdlib.h>
#include <string.h>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e), __LINE__); return 1; } } while (0)

// one lane polls (bounded: about a second -- a protocol error ends with wrong timings, never with a hung GPU), the workgroup follows
__device__ __forceinline__ void wait_flag(const unsigned* flag, unsigned target) {
    if (threadIdx.x == 0)
        for (long it = 0; it < (1l << 20) && __hip_atomic_load(flag, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) < target; ++it)
            __builtin_amdgcn_s_sleep(32);
    __syncthreads();
}
__device__ __forceinline__ void publish(unsigned* flag, unsigned value, bool leader) {
    __syncthreads();
    if (leader && threadIdx.x == 0) __hip_atomic_store(flag, value, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
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

// mode 0 pipe, 2 a-only, 3 b-only.  Role split of the pipe: workgroup w (XCD w % 8) takes role A when ((w / 8) % den) < num.
__global__ __launch_bounds__(256, 8) void pipe_kernel(const uint8_t* p, int64_t pitch, int n_chunks, int chunk_rows, int steps_total, int mode,
                                                   int num, int den, int lag, unsigned* doneA, unsigned* doneB, uint32_t* sink) {
    const int nwg = gridDim.x, w = blockIdx.x;
    uint32_t acc = 0;
    if (mode == 0) {
        const int grp = w / 8, xcd = w % 8;
        const bool is_a = (grp % den) < num;
        const int per = nwg / 8 / den;                       // full periods of `den` groups
        const int n_a = per * num * 8, n_b = per * (den - num) * 8;
        const int me = is_a ? ((grp / den) * num + grp % den) * 8 + xcd : ((grp / den) * (den - num) + (grp % den - num)) * 8 + xcd;
        for (int c = 0; c < n_chunks; ++c) {
            if (is_a) {
                if (c >= 2 + lag) wait_flag(doneB, (unsigned)(c - 1 - lag));  // B's leader has finished chunk c - 2 - lag: A on c runs beside B on c - 1 - lag .. c - 1
                acc += role_a(p, pitch, c * chunk_rows, chunk_rows, steps_total, me, n_a);
                publish(doneA, (unsigned)(c + 1), me == 0);
            } else {
                wait_flag(doneA, (unsigned)(c + 1));                          // A's leader has finished chunk c
                acc += role_b(p, pitch, c * chunk_rows, chunk_rows, me, n_b);
                publish(doneB, (unsigned)(c + 1), me == 0);
            }
        }
    } else {
        for (int c = 0; c < n_chunks; ++c)
            acc += mode == 2 ? role_a(p, pitch, c * chunk_rows, chunk_rows, steps_total, w, nwg) : role_b(p, pitch, c * chunk_rows, chunk_rows, w, nwg);
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
    unsigned* flags; CK(hipMalloc(&flags, 2 * 4096 * sizeof(unsigned)));
    hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    int ncu = 0; CK(hipDeviceGetAttribute(&ncu, hipDeviceAttributeMultiprocessorCount, 0));
    printf("ic_pipeline: %d CUs, pitch %lld B, %lld rows = %.2f GB; rates are bytes READ per second (the pipe reads every byte twice)\n", ncu,
           (long long)pitch, (long long)total_rows, bytes / 1e9);
    auto run = [&](const char* name, int mode, int chunk_rows, int wg_per_cu, int num, int den, int lag, double reads) -> int {
        const int n_chunks = (int)(total_rows / chunk_rows);
        const int nwg = ncu * wg_per_cu;
        float best = 1e9f;
        for (int rep = 0; rep < 3; ++rep) {
            linear_kernel<<<4096, 256>>>((const uint4*)(arena + bytes), flush_bytes / 16, sink);
            CK(hipMemset(flags, 0, 2 * 4096 * sizeof(unsigned)));
            CK(hipDeviceSynchronize());
            CK(hipEventRecord(e0));
            pipe_kernel<<<nwg, 256>>>(arena, pitch, n_chunks, chunk_rows, steps_total, mode, num, den, lag, flags, flags + 4096, sink);     // two progress words, 16 KB apart
            CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1));
            float ms; CK(hipEventElapsedTime(&ms, e0, e1)); best = ms < best ? ms : best;
        }
        const double gb = (double)n_chunks * chunk_rows * pitch * reads / 1e9;
        printf("%-30s chunk %4d rows = %5.1f MB, lag %d, %d WG/CU, A:B = %d:%d  %8.1f us  aggregate %5.2f TB/s  = %5.2f TB/s per read of the genotypes\n", name,
               chunk_rows, chunk_rows * pitch / 1e6, lag, wg_per_cu, num, den - num, best * 1e3, gb / best, gb / reads / best);
        return 0;
    };
    for (int wpc : {4, 8}) {
        if (run("a-only (GEMM 1 pattern, HBM)", 2, 1024, wpc, 1, 2, 0, 1.0)) return 1;
        if (run("b-only (GEMM 2 pattern, HBM)", 3, 1024, wpc, 1, 2, 0, 1.0)) return 1;
    }
    for (int rows : {256, 512, 768, 1024, 1536})
        for (int wpc : {4, 8})
            for (int lag : {0, 1}) {
                if ((int64_t)(lag + 2) * rows * pitch > (int64_t)420 << 20) continue;        // fronts further apart than the cache could ever hold
                if (run("pipe (A on c, B on c - lag..c)", 0, rows, wpc, 1, 2, lag, 2.0)) return 1;
            }
    for (int rows : {512, 1024}) {
        if (run("pipe, 3 of 4 workgroups stream", 0, rows, 8, 3, 4, 0, 2.0)) return 1;
        if (run("pipe, 1 of 4 workgroups stream", 0, rows, 8, 1, 4, 0, 2.0)) return 1;
    }
    return 0;
}

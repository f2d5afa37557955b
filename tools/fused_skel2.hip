// fused_skel2.hip -- round-2 skeleton of the single-read ("fused") sketch kernel: data movement + the cross-CU reduction and
// hand-off protocol only, no matrix work.  It bounds the fused design from above.
//
// Shape of the fused kernel it stands for: 245 persistent workgroups (one per CU), each owning a 2048-sample slice (512 B of
// every row) for the whole launch -- the only scale at which the right-hand operand (24 MB at N = 500k) is resident on chip
// (48 registers per wave) AND the second contraction's accumulators stay in registers.  Per tile of TS = 16 SNP rows a CU
//   (a) streams its 8 KB slice into an LDS ring (LDS-DMA, PF tiles ahead),
//   (b) adds its partial G^T R (16 rows x 48 int32 slots = 3 KB) into the tile's global slot: memory-side atomics,
//   (c) arrives on the tile's counter, and LAG tiles later polls that counter and reads the reduced operand back
//       (512 B: the int8 W image of 16 rows; reading the whole 3 KB P would be 6x more) before it may free the LDS slot.
// Scopes:  device -- one counter per tile, 245 arrivals;  xcd -- arrivals sharded per XCD (<= 32 each) and the LAST arriver of
// an XCD forwards to the tile's top counter (8 arrivals): the barrier-xcd pattern of MI355X_MICROARCH.md's price list.
// Modes: 0 stream only | 1 + atomics | 2 + atomics + protocol (scope device) | 3 + atomics + protocol (scope xcd).
// Waves: 0, 1 loaders (LDS-DMA only: counted vmcnt stays meaningful), 2 touches the tile (stand-in for compute), 3 reducer
// (atomics, arrive, poll, read).  One workgroup barrier per tile.
// build: hipcc --offload-arch=gfx950 -O3 -o tools/bin/fused_skel2 tools/fused_skel2.hip ;  run: tools/bin/fused_skel2
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdint.h>
#include <stdlib.h>
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e_), __LINE__); return 1; } } while (0)

constexpr int TS = 16;            // rows per tile
constexpr int SLICE = 512;        // bytes per row per CU (2048 samples)
constexpr int PF = 3;             // tiles in flight ahead of the one being consumed by phase 1
constexpr int PSLOTS = TS * 48;   // int32 partial sums per tile per CU

__device__ __forceinline__ void dma16(const void* gsrc, void* lds_dst_wave_base) {
    __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)gsrc,
                                     (__attribute__((address_space(3))) void*)lds_dst_wave_base, 16, 0, 0);
}
__device__ __forceinline__ unsigned xcc_id() {
    unsigned v;
    asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID, 0, 4)" : "=s"(v));
    return v & 7u;
}

template <int MODE>
__global__ __launch_bounds__(256) void skel(const uint8_t* __restrict__ packed, int64_t pitch, int n_tiles, int lag, int ring,
                                            int* __restrict__ Pbuf, unsigned* __restrict__ cnt_top, unsigned* __restrict__ cnt_xcd,
                                            const unsigned* __restrict__ xcd_members, const uint2* __restrict__ Wpub,
                                            unsigned* __restrict__ fail, unsigned long long* __restrict__ sink) {
    extern __shared__ __attribute__((aligned(16))) uint8_t lds[];     // ring x TS x SLICE
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int ncu = gridDim.x;
    const uint8_t* base = packed + (int64_t)blockIdx.x * SLICE;
    const unsigned xcc = MODE == 3 ? xcc_id() : 0u;
    auto issue = [&](int t) {      // loader waves only: tile t = 16 rows x 512 B = 8 DMA instructions (2 rows each), 4 per loader wave
        uint8_t* slot = lds + (t % ring) * (TS * SLICE);
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const int piece = wave * 4 + i;
            const int row = 2 * piece + (lane >> 5);
            dma16(base + ((int64_t)t * TS + row) * pitch + (lane & 31) * 16, slot + piece * 1024);
        }
    };
    unsigned long long acc = 0;
    if (wave < 2)
        for (int t = 0; t < PF && t < n_tiles; ++t) issue(t);
    for (int t = 0; t < n_tiles; ++t) {
        if (wave < 2) {
            if (t + PF < n_tiles) {
                issue(t + PF);
                asm volatile("s_waitcnt vmcnt(%0)" ::"n"(4 * PF) : "memory");      // tile t has landed; PF younger tiles stay in flight
            } else {
                asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            }
        }
        __builtin_amdgcn_s_barrier();                                               // tile t is in LDS for every wave
        const uint8_t* slot = lds + (t % ring) * (TS * SLICE);
        if (wave == 2) {                                                            // "phase 1 compute": touch the tile
            const uint4 v = *reinterpret_cast<const uint4*>(slot + lane * 128);
            acc += v.x ^ v.y ^ v.z ^ v.w;
        }
        if (wave == 3 && MODE >= 1) {
            // (b) partial sums of this CU for tile t: 768 int32 = 12 per lane, 256 contiguous bytes per wave-instruction
            int* p = Pbuf + (int64_t)t * PSLOTS + lane;
#pragma unroll
            for (int i = 0; i < PSLOTS / 64; ++i) atomicAdd(p + i * 64, 1);
            if (MODE >= 2) {
                asm volatile("s_waitcnt vmcnt(0)" ::: "memory");                    // my adds are acknowledged
                if (lane == 0) {
                    if (MODE == 2) {
                        __hip_atomic_fetch_add(&cnt_top[t], 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                    } else {
                        const unsigned old = __hip_atomic_fetch_add(&cnt_xcd[(int64_t)xcc * n_tiles + t], 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                        if (old + 1 == xcd_members[xcc])                            // last arriver of this XCD forwards
                            __hip_atomic_fetch_add(&cnt_top[t], xcd_members[xcc], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                    }
                }
                if (t >= lag) {
                    // (c) tile t - lag: every CU has arrived? then read the reduced operand (512 B, sc1 loads) -- only then may
                    // the tile's LDS slot be reused (phase 2 of the real kernel consumes it here)
                    const int tc = t - lag;
                    if (lane == 0) {
                        unsigned spins = 0;
                        while (__hip_atomic_load(&cnt_top[tc], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) < (unsigned)ncu) {
                            __builtin_amdgcn_s_sleep(1);
                            if (++spins > 4000000u) { atomicAdd(fail, 1u); break; }
                        }
                    }
                    __builtin_amdgcn_wave_barrier();
                    const uint2* q = Wpub + (int64_t)tc * 64 + lane;
                    const unsigned long long w8 = __hip_atomic_load(reinterpret_cast<const unsigned long long*>(q), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                    acc += w8;
                }
            }
        }
        // the slot of tile t + PF - ring + ... is refilled by the next iteration's issue: the barrier at its top orders it
        // behind this iteration's consumers (wave 3 included: a reducer stuck polling holds the ring, as the real kernel would)
    }
    if (acc == 0x123456789abcdefull) sink[0] = acc;
}

int main(int argc, char** argv) {
    const int64_t pitch = 125184;
    const int n_rows = 10240 * 2, n_tiles = n_rows / TS;      // two 1.28 GB blocks
    const int ncu = 240;                                       // of the 244 slices of a row: 30 workgroups per XCD under round-robin dispatch
    uint8_t* g; CK(hipMalloc(&g, (size_t)n_rows * pitch)); CK(hipMemset(g, 0x5a, (size_t)n_rows * pitch));
    int* P; CK(hipMalloc(&P, (size_t)n_tiles * PSLOTS * 4));
    unsigned *ctop, *cxcd, *fail, *members; CK(hipMalloc(&ctop, n_tiles * 4)); CK(hipMalloc(&cxcd, 8 * n_tiles * 4));
    CK(hipMalloc(&fail, 4)); CK(hipMalloc(&members, 8 * 4));
    uint2* W; CK(hipMalloc(&W, (size_t)n_tiles * 64 * 8)); CK(hipMemset(W, 1, (size_t)n_tiles * 64 * 8));
    unsigned long long* sink; CK(hipMalloc(&sink, 8));
    hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    // census: how many of the ncu workgroups land on each XCD (placement is not contractual: measure it, every launch shape)
    {
        CK(hipMemset(cxcd, 0, 8 * n_tiles * 4));
        const size_t lds0 = (size_t)(PF + 2) * TS * SLICE;
        CK(hipFuncSetAttribute((const void*)skel<3>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
        CK(hipMemset(members, 0xff, 32));                      // nobody forwards during the census
        CK(hipMemset(ctop, 0, n_tiles * 4)); CK(hipMemset(fail, 0, 4));
        skel<3><<<ncu, 256, lds0>>>(g, pitch, 1, 1 << 30, PF + 2, P, ctop, cxcd, members, W, fail, sink);
        CK(hipDeviceSynchronize());
        unsigned h[8];
        for (int x = 0; x < 8; ++x) CK(hipMemcpy(&h[x], cxcd + x, 4, hipMemcpyDeviceToHost));          // the census ran with n_tiles = 1
        printf("workgroups per XCD:");
        unsigned tot = 0;
        for (int x = 0; x < 8; ++x) { printf(" %u", h[x]); tot += h[x]; }
        printf("  (total %u of %d)\n", tot, ncu);
        CK(hipMemcpy(members, h, 32, hipMemcpyHostToDevice));
        for (int x = 0; x < 8; ++x)
            if (h[x] != (unsigned)ncu / 8) printf("WARNING: uneven placement -- the xcd-scope rows below may time out\n");
    }
    const int lags[] = {2, 6, 12, 16};
    for (int mode = 0; mode < 4; ++mode) {
        for (int li = 0; li < (mode < 2 ? 1 : 4); ++li) {
            const int lag = mode < 2 ? 2 : lags[li];
            const int ring = lag + PF + 1;
            const size_t lds = (size_t)ring * TS * SLICE;
            if (lds > 160 * 1024) continue;
            for (int rep = 0; rep < 3; ++rep) {
                CK(hipMemset(P, 0, (size_t)n_tiles * PSLOTS * 4)); CK(hipMemset(ctop, 0, n_tiles * 4));
                CK(hipMemset(cxcd, 0, 8 * n_tiles * 4)); CK(hipMemset(fail, 0, 4));
                CK(hipDeviceSynchronize());
                hipEventRecord(e0);
#define LAUNCH(M_) do { CK(hipFuncSetAttribute((const void*)skel<M_>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024)); \
                        skel<M_><<<ncu, 256, lds>>>(g, pitch, n_tiles, lag, ring, P, ctop, cxcd, members, W, fail, sink); } while (0)
                if (mode == 0) LAUNCH(0); else if (mode == 1) LAUNCH(1); else if (mode == 2) LAUNCH(2); else LAUNCH(3);
                hipEventRecord(e1); CK(hipDeviceSynchronize());
                float ms; hipEventElapsedTime(&ms, e0, e1);
                unsigned f; CK(hipMemcpy(&f, fail, 4, hipMemcpyDeviceToHost));
                int p0; CK(hipMemcpy(&p0, P, 4, hipMemcpyDeviceToHost));
                const double gb = (double)n_rows * ncu * SLICE / 1e9;
                printf("mode %d (%s) lag %2d ring %2d (%3zu KB LDS) rep %d: %8.1f us  %.2f TB/s  (poll timeouts %u, P[0]=%d)\n", mode,
                       mode == 0 ? "stream" : mode == 1 ? "stream+atomics" : mode == 2 ? "protocol, device scope" : "protocol, xcd scope",
                       lag, ring, lds / 1024, rep, ms * 1e3, gb / ms, f, p0);
            }
        }
    }
    return 0;
}

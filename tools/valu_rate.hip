// valu_rate.hip -- issue cost of the unpack instructions (cycles per wave64 instruction, 1 and 2 waves per SIMD)
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdint.h>
template <int OP>
__global__ void k(uint32_t* out, unsigned long long* cyc, int iters) {
    uint32_t a = threadIdx.x * 2654435761u, b = a ^ 0x9e3779b9u, c = a + 77, d = b + 99;
    uint32_t m = 0x00030003u + (threadIdx.x & 1);
    unsigned long long t0 = __builtin_amdgcn_s_memtime();
    for (int i = 0; i < iters; ++i) {
#pragma unroll
        for (int j = 0; j < 16; ++j) {
            if (OP == 0) { asm volatile("v_and_b32 %0, %0, %1\n v_and_b32 %2, %2, %1\n v_and_b32 %3, %3, %1\n v_and_b32 %4, %4, %1" : "+v"(a), "+v"(m), "+v"(b), "+v"(c), "+v"(d)); }
            if (OP == 1) { asm volatile("v_pk_fma_f16 %0, %0, %1, %1\n v_pk_fma_f16 %2, %2, %1, %1\n v_pk_fma_f16 %3, %3, %1, %1\n v_pk_fma_f16 %4, %4, %1, %1" : "+v"(a), "+v"(m), "+v"(b), "+v"(c), "+v"(d)); }
            if (OP == 2) { asm volatile("v_perm_b32 %0, %0, %1, %1\n v_perm_b32 %2, %2, %1, %1\n v_perm_b32 %3, %3, %1, %1\n v_perm_b32 %4, %4, %1, %1" : "+v"(a), "+v"(m), "+v"(b), "+v"(c), "+v"(d)); }
            if (OP == 3) { asm volatile("v_and_or_b32 %0, %0, %1, %1\n v_and_or_b32 %2, %2, %1, %1\n v_and_or_b32 %3, %3, %1, %1\n v_and_or_b32 %4, %4, %1, %1" : "+v"(a), "+v"(m), "+v"(b), "+v"(c), "+v"(d)); }
            if (OP == 4) { asm volatile("v_pk_add_f16 %0, %0, %1\n v_pk_add_f16 %2, %2, %1\n v_pk_add_f16 %3, %3, %1\n v_pk_add_f16 %4, %4, %1" : "+v"(a), "+v"(m), "+v"(b), "+v"(c), "+v"(d)); }
            if (OP == 5) { asm volatile("v_lshrrev_b32 %0, 1, %0\n v_lshrrev_b32 %2, 1, %2\n v_lshrrev_b32 %3, 1, %3\n v_lshrrev_b32 %4, 1, %4" : "+v"(a), "+v"(m), "+v"(b), "+v"(c), "+v"(d)); }
        }
    }
    unsigned long long t1 = __builtin_amdgcn_s_memtime();
    out[blockIdx.x * blockDim.x + threadIdx.x] = a ^ b ^ c ^ d;
    if (threadIdx.x == 0 && blockIdx.x == 0) cyc[0] = t1 - t0;
}
int main() {
    uint32_t* o; unsigned long long* c; hipMalloc(&o, 1 << 22); hipMalloc(&c, 8);
    const char* names[] = {"v_and_b32", "v_pk_fma_f16", "v_perm_b32", "v_and_or_b32", "v_pk_add_f16", "v_lshrrev_b32"};
    for (int waves = 1; waves <= 2; ++waves) for (int op = 0; op < 6; ++op) {
        const int iters = 2000; unsigned long long cy;
        auto run = [&](int threads) {
            if (op == 0) k<0><<<256, threads>>>(o, c, iters); if (op == 1) k<1><<<256, threads>>>(o, c, iters);
            if (op == 2) k<2><<<256, threads>>>(o, c, iters); if (op == 3) k<3><<<256, threads>>>(o, c, iters);
            if (op == 4) k<4><<<256, threads>>>(o, c, iters); if (op == 5) k<5><<<256, threads>>>(o, c, iters);
        };
        run(256 * waves); hipDeviceSynchronize(); run(256 * waves); hipDeviceSynchronize();
        hipMemcpy(&cy, c, 8, hipMemcpyDeviceToHost);
        printf("%d wave(s)/SIMD %-14s %.2f cycles per wave-instruction (per wave)\n", waves, names[op], (double)cy / (iters * 64.0));
    }
    return 0;
}

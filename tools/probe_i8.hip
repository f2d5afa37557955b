// probe_i8.hip -- operand layout of v_mfma_i32_16x16x64_i8 on gfx950, checked with random int8 data.
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdint.h>
#include <vector>
typedef int v4i __attribute__((ext_vector_type(4)));
__global__ void k(const v4i* a, const v4i* b, v4i* out) {
    v4i acc = {0, 0, 0, 0};
    acc = __builtin_amdgcn_mfma_i32_16x16x64_i8(a[threadIdx.x], b[threadIdx.x], acc, 0, 0, 0);
    out[threadIdx.x] = acc;
}
int main() {
    std::vector<int8_t> A(16 * 64), B(64 * 16);          // A[row][k], B[k][col]
    uint32_t rng = 1;
    auto nx = [&]() { rng = rng * 1664525u + 1013904223u; return (int)((rng >> 16) % 255) - 127; };
    for (auto& v : A) v = (int8_t)nx();
    for (auto& v : B) v = (int8_t)nx();
    // hypothesis: lane l holds A[l&15][16*(l>>4) + j] and B[16*(l>>4) + j][l&15], j = 0..15 in byte order
    std::vector<int8_t> ha(64 * 16), hb(64 * 16);
    for (int l = 0; l < 64; ++l) for (int j = 0; j < 16; ++j) {
        ha[l * 16 + j] = A[(l & 15) * 64 + 16 * (l >> 4) + j];
        hb[l * 16 + j] = B[(16 * (l >> 4) + j) * 16 + (l & 15)];
    }
    int8_t *da, *db; int* dout;
    hipMalloc(&da, 1024); hipMalloc(&db, 1024); hipMalloc(&dout, 1024);
    hipMemcpy(da, ha.data(), 1024, hipMemcpyHostToDevice); hipMemcpy(db, hb.data(), 1024, hipMemcpyHostToDevice);
    k<<<1, 64>>>((const v4i*)da, (const v4i*)db, (v4i*)dout);
    std::vector<int> O(256);
    hipMemcpy(O.data(), dout, 1024, hipMemcpyDeviceToHost);
    int bad = 0;
    for (int l = 0; l < 64; ++l) for (int r = 0; r < 4; ++r) {
        const int row = 4 * (l >> 4) + r, col = l & 15;
        int ref = 0; for (int kk = 0; kk < 64; ++kk) ref += (int)A[row * 64 + kk] * (int)B[kk * 16 + col];
        bad += ref != O[l * 4 + r];
    }
    printf("i8 16x16x64 layout hypothesis (k = 16*(lane>>4) + byte j): %s (%d mismatches)\n", bad ? "WRONG" : "CONFIRMED", bad);
    return 0;
}

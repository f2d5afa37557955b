// g2t_proto.hip -- speed prototype (no numerics check): GEMM 2 on v_mfma_i32_16x16x64_i8 reading a TRANSPOSED copy of
// the genotypes (one dword = 16 consecutive SNPs of ONE sample, K-group rows of n_pad dwords), so a dword unpacks
// straight into one MFMA operand (7 VALU ops) -- against today's fp16 hi+lo kernel (152 VALU + 32 MFMA per 2 KB).
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdint.h>
typedef int int4v __attribute__((ext_vector_type(4)));
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e), __LINE__); return 1; } } while (0)

template <int SPL>   // SPL = 16-byte loads per lane per step (4*SPL samples per lane, 64*SPL per wave)
__global__ __launch_bounds__(256) void g2t(const uint32_t* __restrict__ T, int64_t n_pad, int n_kg, const int4v* __restrict__ Wimg,
                                           float* __restrict__ out) {
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, n = lane & 15, kq = lane >> 4;
    const int64_t s0 = ((int64_t)blockIdx.x * 4 + wave) * (64 * SPL) + 4 * n;
    if (s0 >= n_pad) return;
    int4v acc[SPL][4][3];
    for (int a = 0; a < SPL; ++a) for (int u = 0; u < 4; ++u) for (int l = 0; l < 3; ++l) acc[a][u][l] = (int4v){0, 0, 0, 0};
    const int n_steps = n_kg / 4;
    uint4 g[2][SPL];
    auto issue = [&](int st, uint4 (&gg)[SPL]) {
        const uint32_t* p = T + (int64_t)(st * 4 + kq) * n_pad + s0;
        for (int a = 0; a < SPL; ++a) gg[a] = *reinterpret_cast<const uint4*>(p + 64 * a);
    };
    issue(0, g[0]);
    for (int st = 0; st < n_steps; st += 2) {
#pragma unroll
        for (int k = 0; k < 2; ++k) {
            issue(min(st + k + 1, n_steps - 1), g[k ^ 1]);
            const int4v* wf = Wimg + (int64_t)(st + k) * (3 * 64) + lane;
            const int4v b0 = wf[0], b1 = wf[64], b2 = wf[128];
#pragma unroll
            for (int a = 0; a < SPL; ++a)
#pragma unroll
                for (int u = 0; u < 4; ++u) {
                    const uint32_t v = u == 0 ? g[k][a].x : (u == 1 ? g[k][a].y : (u == 2 ? g[k][a].z : g[k][a].w));
                    const int4v av = {(int)(v & 0x03030303u), (int)((v >> 2) & 0x03030303u), (int)((v >> 4) & 0x03030303u), (int)((v >> 6) & 0x03030303u)};
                    acc[a][u][0] = __builtin_amdgcn_mfma_i32_16x16x64_i8(av, b0, acc[a][u][0], 0, 0, 0);
                    acc[a][u][1] = __builtin_amdgcn_mfma_i32_16x16x64_i8(av, b1, acc[a][u][1], 0, 0, 0);
                    acc[a][u][2] = __builtin_amdgcn_mfma_i32_16x16x64_i8(av, b2, acc[a][u][2], 0, 0, 0);
                }
        }
    }
    // epilogue: combine limbs, store [c][sample]
    for (int a = 0; a < SPL; ++a) for (int u = 0; u < 4; ++u) for (int reg = 0; reg < 4; ++reg) {
        const float v = (float)acc[a][u][0][reg] + (float)acc[a][u][1][reg] * (1.f / 128.f) + (float)acc[a][u][2][reg] * (1.f / 16384.f);
        const int64_t samp = ((int64_t)blockIdx.x * 4 + wave) * (64 * SPL) + 64 * a + 4 * (4 * kq + reg) + u;   // row 4kq+reg of tile u
        if (n < 10) out[(int64_t)n * n_pad + samp] = v;
    }
}

int main() {
    const int64_t n_pad = 500736; const int n_kg = 628;          // 10 048 SNPs
    uint32_t* T; CK(hipMalloc(&T, (size_t)n_kg * n_pad * 4)); CK(hipMemset(T, 0x15, (size_t)n_kg * n_pad * 4));
    int4v* W; CK(hipMalloc(&W, (size_t)(n_kg / 4) * 3 * 64 * 16)); CK(hipMemset(W, 1, (size_t)(n_kg / 4) * 3 * 64 * 16));
    float* out; CK(hipMalloc(&out, (size_t)10 * n_pad * 4));
    uint8_t* flush; CK(hipMalloc(&flush, (size_t)2 << 30)); 
    hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    const double gb = (double)n_kg * n_pad * 4 / 1e9;
    for (int spl = 1; spl <= 2; ++spl) {
        float best = 1e9f;
        for (int rep = 0; rep < 4; ++rep) {
            CK(hipMemsetAsync(flush, rep, (size_t)2 << 30, 0));
            const int grid = (int)((n_pad + 256 * spl - 1) / (256 * spl));
            CK(hipEventRecord(e0));
            if (spl == 1) g2t<1><<<grid, 256>>>(T, n_pad, n_kg, W, out); else g2t<2><<<grid, 256>>>(T, n_pad, n_kg, W, out);
            CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1));
            float ms; CK(hipEventElapsedTime(&ms, e0, e1)); best = ms < best ? ms : best;
        }
        printf("g2t SPL=%d (%d samples per wave): %.1f us for %.2f GB -> %.2f TB/s\n", spl, 64 * spl, best * 1e3, gb, gb / best);
    }
    return 0;
}

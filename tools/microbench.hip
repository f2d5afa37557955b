// microbench.hip -- hardware questions behind the sketch-kernel design (run on the MI355X box):
//   E1  does v_mfma_f32_16x16x32_f16 honour fp16 SUBNORMAL inputs exactly?  (1-op unpack trick)
//   E2  read bandwidth of a packed sub-block: first touch (HBM) vs second touch (Infinity Cache),
//       for the row-streaming (GEMM 1) and column-panel (GEMM 2, 64/128/256 B per row) patterns.
// build: hipcc --offload-arch=gfx950 -O3 -o tools/microbench tools/microbench.hip
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdint.h>
#include <vector>
typedef _Float16 half8 __attribute__((ext_vector_type(8)));
typedef float float4v __attribute__((ext_vector_type(4)));
union Frag { half8 h; uint32_t u[4]; uint16_t s[8]; };
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e), __LINE__); return 1; } } while (0)

__global__ void e1_kernel(const uint32_t* a_bits, const uint32_t* b_bits, float* out) {
    Frag a, b;
    for (int i = 0; i < 4; ++i) { a.u[i] = a_bits[threadIdx.x * 4 + i]; b.u[i] = b_bits[threadIdx.x * 4 + i]; }
    float4v acc = {0, 0, 0, 0};
    acc = __builtin_amdgcn_mfma_f32_16x16x32_f16(a.h, b.h, acc, 0, 0, 0);
    for (int i = 0; i < 4; ++i) out[threadIdx.x * 4 + i] = acc[i];
}

// row streaming: each wave sweeps `rows_per_wave` rows, 16 rows x 64 B per wave-instruction like gemm1
__global__ void rowstream_kernel(const uint8_t* p, int64_t pitch, int n_rows, int chunks, int cps, uint32_t* sink) {
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, r = lane & 15, q = lane >> 4;
    const int ebase = (blockIdx.x * 4 + wave) * 64;
    if (ebase >= n_rows) return;
    const int c0 = blockIdx.y * cps, c1 = min(c0 + cps, chunks);
    uint32_t acc = 0;
    const uint8_t* rp[4];
    for (int t = 0; t < 4; ++t) rp[t] = p + (int64_t)min(ebase + 16 * t + r, n_rows - 1) * pitch + 16 * q;
    for (int c = c0; c < c1; ++c)
        for (int t = 0; t < 4; ++t) { uint4 g = *(const uint4*)(rp[t] + (int64_t)c * 64); acc += g.x ^ g.y ^ g.z ^ g.w; }
    if (acc == 0x12345678u) sink[0] = acc;
}

// column panel: lane (n, q) reads W bytes of rows 8q+j at column group grp; W = 4/8/16
template <int W>
__global__ void colpanel_kernel(const uint8_t* p, int64_t pitch, int n_rows, uint32_t* sink) {
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, n = lane & 15, q = lane >> 4;
    const int64_t grp = (int64_t)blockIdx.x * 4 + wave;
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

static float time_ms(hipEvent_t a, hipEvent_t b) { float ms; hipEventElapsedTime(&ms, a, b); return ms; }

int main() {
    // ---------------- E1
    {
        std::vector<uint32_t> A(256), B(256);
        // A[row r][k]: fp16 subnormal bit patterns c << (2t): value c * 4^t * 2^-24 ; B[k][col] = small integers
        std::vector<double> a_val(16 * 32), b_val(32 * 16);
        for (int l = 0; l < 64; ++l) {
            Frag fa, fb;
            for (int j = 0; j < 8; ++j) {
                const int r = l & 15, k = 8 * (l >> 4) + j;
                const int c = (r + 3 * k) % 3, t = (k + r) % 4;
                const uint16_t bits = (uint16_t)(c << (2 * t));
                fa.s[j] = bits;
                a_val[r * 32 + k] = c * (double)(1 << (2 * t)) / 16777216.0;
                const int col = l & 15;
                const float bv = (float)(((k * 7 + col * 3) % 11) - 5);
                _Float16 hb = (_Float16)bv; fb.s[j] = *(uint16_t*)&hb;
                b_val[k * 16 + col] = bv;
            }
            for (int i = 0; i < 4; ++i) { A[l * 4 + i] = fa.u[i]; B[l * 4 + i] = fb.u[i]; }
        }
        uint32_t *dA, *dB; float* dO;
        CK(hipMalloc(&dA, 1024)); CK(hipMalloc(&dB, 1024)); CK(hipMalloc(&dO, 1024));
        CK(hipMemcpy(dA, A.data(), 1024, hipMemcpyHostToDevice)); CK(hipMemcpy(dB, B.data(), 1024, hipMemcpyHostToDevice));
        e1_kernel<<<1, 64>>>(dA, dB, dO);
        std::vector<float> O(256);
        CK(hipMemcpy(O.data(), dO, 1024, hipMemcpyDeviceToHost));
        double maxerr = 0, maxref = 0; int nz = 0;
        for (int l = 0; l < 64; ++l) for (int reg = 0; reg < 4; ++reg) {
            const int row = 4 * (l >> 4) + reg, col = l & 15;
            double ref = 0; for (int k = 0; k < 32; ++k) ref += a_val[row * 32 + k] * b_val[k * 16 + col];
            maxerr = fmax(maxerr, fabs(O[l * 4 + reg] - ref)); maxref = fmax(maxref, fabs(ref)); nz += O[l * 4 + reg] != 0;
        }
        printf("E1 mfma fp16 subnormal inputs: max|err| = %.3e (max|ref| = %.3e, nonzero outputs %d/256) -> %s\n",
               maxerr, maxref, nz, maxerr == 0 ? "EXACT (denormals honoured)" : "NOT exact (flushed?)");
    }
    // ---------------- E1b: subnormal A x {large normal, small normal, subnormal} B, chained accumulation
    {
        for (int test = 0; test < 3; ++test) {
            std::vector<uint32_t> A(256), B(256);
            std::vector<double> a_val(16 * 32), b_val(32 * 16);
            uint32_t rng = 12345u + test;
            auto nextf = [&]() { rng = rng * 1664525u + 1013904223u; return (rng >> 8) * (1.0f / 16777216.0f); };
            for (int l = 0; l < 64; ++l) {
                Frag fa, fb;
                for (int j = 0; j < 8; ++j) {
                    const int r = l & 15, k = 8 * (l >> 4) + j, p = j >> 1;
                    const int c = (int)(nextf() * 3.0f);
                    fa.s[j] = (uint16_t)(c << (2 * p));
                    a_val[r * 32 + k] = c * (double)(1 << (2 * p)) / 16777216.0;
                    float bv = (nextf() - 0.5f) * (test == 0 ? 8192.f : (test == 1 ? 0.05f : 1e-4f)) / (float)(1 << (2 * p));
                    _Float16 hb = (_Float16)bv; fb.s[j] = *(uint16_t*)&hb;
                    b_val[k * 16 + (l & 15)] = (double)(float)hb;
                }
                for (int i = 0; i < 4; ++i) { A[l * 4 + i] = fa.u[i]; B[l * 4 + i] = fb.u[i]; }
            }
            uint32_t *dA, *dB; float* dO;
            CK(hipMalloc(&dA, 1024)); CK(hipMalloc(&dB, 1024)); CK(hipMalloc(&dO, 1024));
            CK(hipMemcpy(dA, A.data(), 1024, hipMemcpyHostToDevice)); CK(hipMemcpy(dB, B.data(), 1024, hipMemcpyHostToDevice));
            e1_kernel<<<1, 64>>>(dA, dB, dO);
            std::vector<float> O(256);
            CK(hipMemcpy(O.data(), dO, 1024, hipMemcpyDeviceToHost));
            double maxerr = 0, maxref = 0;
            for (int l = 0; l < 64; ++l) for (int reg = 0; reg < 4; ++reg) {
                const int row = 4 * (l >> 4) + reg, col = l & 15;
                double ref = 0; for (int k = 0; k < 32; ++k) ref += a_val[row * 32 + k] * b_val[k * 16 + col];
                maxerr = fmax(maxerr, fabs(O[l * 4 + reg] - ref)); maxref = fmax(maxref, fabs(ref));
            }
            printf("E1b test %d (B scale %s): max|err| = %.3e  max|ref| = %.3e  rel = %.2e\n", test,
                   test == 0 ? "8192" : (test == 1 ? "0.05" : "1e-4 (subnormal)"), maxerr, maxref, maxerr / maxref);
        }
    }
    // ---------------- E2
    hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    const int64_t pitch = 125184;
    const int64_t big_rows = 16384;                       // 2 GB arena to flush caches
    uint8_t* arena; CK(hipMalloc(&arena, big_rows * pitch)); CK(hipMemset(arena, 0x5a, big_rows * pitch));
    uint32_t* sink; CK(hipMalloc(&sink, 4));
    const int chunks = pitch / 64;
    for (int n_rows : {1024}) {
        const double mb = n_rows * (double)pitch / 1e6;
        // flush: touch the far end of the arena
        auto flush = [&]() { rowstream_kernel<<<dim3(16, 64), 256>>>(arena + (big_rows - 4096) * pitch, pitch, 4096, chunks, (chunks + 63) / 64, sink); };
        const int gx = (n_rows + 255) / 256; int splits = (768 + gx - 1) / gx; const int cps = (chunks + splits - 1) / splits; splits = (chunks + cps - 1) / cps;
        float t_first, t_second, t_col[3], t_col_first[3];
        flush(); CK(hipDeviceSynchronize());
        hipEventRecord(e0); rowstream_kernel<<<dim3(gx, splits), 256>>>(arena, pitch, n_rows, chunks, cps, sink); hipEventRecord(e1); CK(hipDeviceSynchronize()); t_first = time_ms(e0, e1);
        hipEventRecord(e0); rowstream_kernel<<<dim3(gx, splits), 256>>>(arena, pitch, n_rows, chunks, cps, sink); hipEventRecord(e1); CK(hipDeviceSynchronize()); t_second = time_ms(e0, e1);
        for (int wi = 0; wi < 3; ++wi) {
            const int W = 4 << wi; const int grid = (int)((pitch / (16 * W) + 3) / 4);
            auto run = [&]() { if (W == 4) colpanel_kernel<4><<<grid, 256>>>(arena, pitch, n_rows, sink); else if (W == 8) colpanel_kernel<8><<<grid, 256>>>(arena, pitch, n_rows, sink); else colpanel_kernel<16><<<grid, 256>>>(arena, pitch, n_rows, sink); };
            flush(); CK(hipDeviceSynchronize());
            hipEventRecord(e0); run(); hipEventRecord(e1); CK(hipDeviceSynchronize()); t_col_first[wi] = time_ms(e0, e1);
            // warm: row-stream first (as gemm1 would), then column panel
            flush(); CK(hipDeviceSynchronize());
            rowstream_kernel<<<dim3(gx, splits), 256>>>(arena, pitch, n_rows, chunks, cps, sink);
            hipEventRecord(e0); run(); hipEventRecord(e1); CK(hipDeviceSynchronize()); t_col[wi] = time_ms(e0, e1);
        }
        printf("E2 rows=%4d (%.0f MB): rowstream cold %.1f us (%.2f TB/s) warm %.1f us (%.2f TB/s) | colpanel after rowstream W=4: %.1f us (%.2f TB/s) W=8: %.1f (%.2f) W=16: %.1f (%.2f) | colpanel cold W=4 %.2f W=8 %.2f W=16 %.2f TB/s\n",
               n_rows, mb, t_first * 1e3, mb / t_first / 1e3, t_second * 1e3, mb / t_second / 1e3,
               t_col[0] * 1e3, mb / t_col[0] / 1e3, t_col[1] * 1e3, mb / t_col[1] / 1e3, t_col[2] * 1e3, mb / t_col[2] / 1e3,
               mb / t_col_first[0] / 1e3, mb / t_col_first[1] / 1e3, mb / t_col_first[2] / 1e3);
    }
    return 0;
}

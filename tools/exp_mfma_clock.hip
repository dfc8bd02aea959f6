// exp_mfma_clock.hip — sustained shader clock under a saturated bf16 matrix pipe: 16x16x32 vs 32x32x16 MFMAs
// (same FLOP per cycle; the 32x32 form reads half the operand bytes per FLOP).  One wave per SIMD, every CU busy.
// build: hipcc --offload-arch=gfx950 -O3 -std=c++17 tools/exp_mfma_clock.hip -o tools/exp_mfma_clock
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef float f32x16 __attribute__((ext_vector_type(16)));
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s at %s:%d\n", hipGetErrorString(e), __FILE__, __LINE__); exit(1); } } while (0)

__device__ unsigned long long g_st[4];

template <int KIND>
__global__ __launch_bounds__(256, 1) void spin(float* out, int iters) {
    bf16x8 a[4], b[4];
    for (int i = 0; i < 4; ++i)
        for (int e = 0; e < 8; ++e) { a[i][e] = (__bf16)(0.001f * (threadIdx.x + i + e)); b[i][e] = (__bf16)(0.002f * (threadIdx.x + 3 * i + e)); }
    if (blockIdx.x == 0 && threadIdx.x == 0) { g_st[0] = __builtin_amdgcn_s_memrealtime(); g_st[1] = __builtin_amdgcn_s_memtime(); }
    float r = 0.f;
    if (KIND == 0) {
        f32x4 c[16];
        for (int i = 0; i < 16; ++i) c[i] = f32x4{0.f, 0.f, 0.f, 0.f};
        for (int it = 0; it < iters; ++it) {
#pragma unroll
            for (int i = 0; i < 16; ++i) c[i] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a[i & 3], b[i >> 2], c[i], 0, 0, 0);
        }
        for (int i = 0; i < 16; ++i) r += c[i][0] + c[i][3];
    } else {
        f32x16 c[8];
        for (int i = 0; i < 8; ++i) for (int e = 0; e < 16; ++e) c[i][e] = 0.f;
        for (int it = 0; it < iters; ++it) {
#pragma unroll
            for (int i = 0; i < 8; ++i) c[i] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[i & 3], b[i >> 1], c[i], 0, 0, 0);
        }
        for (int i = 0; i < 8; ++i) r += c[i][0] + c[i][15];
    }
    if (blockIdx.x == 0 && threadIdx.x == 0) { g_st[2] = __builtin_amdgcn_s_memrealtime(); g_st[3] = __builtin_amdgcn_s_memtime(); }
    out[blockIdx.x * 256 + threadIdx.x] = r;
}

int main() {
    float* out;
    CK(hipMalloc(&out, 256 * 256 * 4));
    for (int kind = 0; kind < 2; ++kind) {
        for (int rep = 0; rep < 3; ++rep) {
            const int iters = 400000;  // 16 x 16 cycles (or 8 x 32) per trip = 256 cycles -> ~ 50 ms
            if (kind == 0) spin<0><<<256, 256>>>(out, iters); else spin<1><<<256, 256>>>(out, iters);
            CK(hipDeviceSynchronize());
            unsigned long long st[4];
            CK(hipMemcpyFromSymbol(st, HIP_SYMBOL(g_st), sizeof(st)));
            const double us = (st[2] - st[0]) / 100.0, cyc = (double)(st[3] - st[1]);
            printf("%s: %.1f ms, %.0f MHz, pipe busy %.3f, %.0f TFLOP/s chip-wide\n", kind ? "32x32x16" : "16x16x32", us / 1e3, cyc / us,
                   (double)iters * 256 / cyc, (double)iters * 16 * 16384 * 1024 / (us * 1e-6) / 1e12);
        }
    }
    return 0;
}

// mfma_peak.hip — what fraction of the 157.3 TFLOP/s fp32-matrix figure can a loop of NOTHING but
// v_mfma_f32_16x16x4_f32 reach on this chip?  (context for roofline.frac; not product code)
// build: hipcc --offload-arch=gfx950 -O3 -std=c++17 tools/mfma_peak.hip -o tools/mfma_peak
#include <hip/hip_runtime.h>
#include <cstdio>
typedef float f32x4 __attribute__((ext_vector_type(4)));

// the same loop with NV independent VALU FMAs per 8 * NACC MFMAs: do they hide under the MFMAs?
template <int NACC, int NV>
__global__ __launch_bounds__(256) void kv(float* out, int iters, float a0, float b0) {
    f32x4 acc[NACC];
#pragma unroll
    for (int i = 0; i < NACC; ++i) acc[i] = f32x4{0.f, 0.f, 0.f, 0.f};
    float v[NV > 0 ? NV : 1];
#pragma unroll
    for (int i = 0; i < NV; ++i) v[i] = a0 + i;
    float a = a0 + threadIdx.x, b = b0 - threadIdx.x;
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int r = 0; r < 8; ++r)
#pragma unroll
            for (int i = 0; i < NACC; ++i) acc[i] = __builtin_amdgcn_mfma_f32_16x16x4f32(a, b, acc[i], 0, 0, 0);
#pragma unroll
        for (int i = 0; i < NV; ++i) v[i] = fmaf(v[i], a, b);
    }
    f32x4 s = acc[0];
#pragma unroll
    for (int i = 1; i < NACC; ++i) s += acc[i];
    float t = 0.f;
#pragma unroll
    for (int i = 0; i < NV; ++i) t += v[i];
    if (s[0] + t == 123.f) out[threadIdx.x] = s[1];
}

template <int NACC, int NV>
void runv(int waves_per_simd) {
    float* out; hipMalloc(&out, 4096);
    const int iters = 4000;
    dim3 grid(256 * waves_per_simd), block(256);
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    hipLaunchKernelGGL((kv<NACC, NV>), grid, block, 0, 0, out, iters, 1.f, 2.f);
    hipDeviceSynchronize();
    hipEventRecord(e0);
    for (int rep = 0; rep < 5; ++rep) hipLaunchKernelGGL((kv<NACC, NV>), grid, block, 0, 0, out, iters, 1.f, 2.f);
    hipEventRecord(e1); hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1); ms /= 5;
    const double flops = (double)grid.x * 4 * iters * 8 * NACC * 2048.0;
    printf("%d MFMAs + %2d VALU FMAs per iteration, waves/SIMD %d: %.3f ms  %.1f TFLOP/s  frac %.3f\n", 8 * NACC, NV,
           waves_per_simd, ms, flops / (ms * 1e-3) / 1e12, flops / (ms * 1e-3) / 1e12 / 157.3);
}

template <int NACC>
__global__ __launch_bounds__(256) void k(float* out, int iters, float a0, float b0) {
    f32x4 acc[NACC];
#pragma unroll
    for (int i = 0; i < NACC; ++i) acc[i] = f32x4{0.f, 0.f, 0.f, 0.f};
    float a = a0 + threadIdx.x, b = b0 - threadIdx.x;
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int r = 0; r < 8; ++r)
#pragma unroll
            for (int i = 0; i < NACC; ++i) acc[i] = __builtin_amdgcn_mfma_f32_16x16x4f32(a, b, acc[i], 0, 0, 0);
    }
    f32x4 s = acc[0];
#pragma unroll
    for (int i = 1; i < NACC; ++i) s += acc[i];
    if (s[0] == 123.f) out[threadIdx.x] = s[1];
}

template <int NACC>
void run(int waves_per_simd, const char* name) {
    float* out; hipMalloc(&out, 4096);
    const int iters = 4000;
    dim3 grid(256 * waves_per_simd), block(256);  // 4 waves per block = 1 per SIMD per resident block
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    hipLaunchKernelGGL(k<NACC>, grid, block, 0, 0, out, iters, 1.f, 2.f);
    hipDeviceSynchronize();
    hipEventRecord(e0);
    for (int rep = 0; rep < 5; ++rep) hipLaunchKernelGGL(k<NACC>, grid, block, 0, 0, out, iters, 1.f, 2.f);
    hipEventRecord(e1); hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1); ms /= 5;
    const double flops = (double)grid.x * 4 * iters * 8 * NACC * 2048.0;
    printf("%-28s waves/SIMD %d  %.3f ms  %.1f TFLOP/s  frac of 157.3 = %.3f\n", name, waves_per_simd, ms,
           flops / (ms * 1e-3) / 1e12, flops / (ms * 1e-3) / 1e12 / 157.3);
}

int main() {
    run<4>(1, "4 independent acc"); run<4>(2, "4 independent acc"); run<4>(4, "4 independent acc");
    run<8>(1, "8 independent acc"); run<8>(2, "8 independent acc");
    run<16>(1, "16 independent acc"); run<16>(2, "16 independent acc");
    runv<8, 0>(2); runv<8, 16>(2); runv<8, 32>(2); runv<8, 64>(2); runv<8, 32>(1); runv<8, 32>(4);
    return 0;
}

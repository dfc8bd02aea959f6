// exp_valu_phase.hip — do VALU bursts hide under the PARTNER wave's MFMAs when the two waves of a SIMD are kept out of
// phase?  (not product code)   build: hipcc --offload-arch=gfx950 -O3 -std=c++17 tools/exp_valu_phase.hip -o tools/exp_valu_phase
// Each wave: per iteration NM MFMAs (8 accumulators) and NV VALU FMAs in BURSTS of BL.  PH 0: every wave runs the same
// stream (bursts after MFMA groups); PH 1: waves 4-7 of the 8-wave block (the SIMD partners of waves 0-3) run their bursts
// half a period later; PH 2: odd waves shifted instead.
#include <hip/hip_runtime.h>
#include <cstdio>
typedef float f32x4 __attribute__((ext_vector_type(4)));

template <int NV, int BL, int PH>
__global__ __launch_bounds__(512, 1) void k(float* out, int iters, float a0, float b0) {
    constexpr int NM = 64;
    constexpr int NBURST = NV / BL;            // bursts per iteration
    constexpr int GAP = NM / NBURST;           // MFMAs between bursts
    f32x4 acc[8];
#pragma unroll
    for (int i = 0; i < 8; ++i) acc[i] = f32x4{0.f, 0.f, 0.f, 0.f};
    float v[8];
#pragma unroll
    for (int i = 0; i < 8; ++i) v[i] = a0 + i;
    float a = a0 + threadIdx.x, b = b0 - threadIdx.x;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const bool shifted = PH == 1 ? (wave >= 4) : (PH == 2 ? (wave & 1) : false);
    auto mf = [&](int n) {
#pragma unroll
        for (int m = 0; m < n; ++m) asm volatile("v_mfma_f32_16x16x4_f32 %0, %1, %2, %0" : "+v"(acc[m & 7]) : "v"(a), "v"(b));
    };
    auto burst = [&]() {
#pragma unroll
        for (int m = 0; m < BL; ++m) asm volatile("v_fmac_f32 %0, %1, %2" : "+v"(v[m & 7]) : "v"(a), "v"(b));
    };
    if (shifted) mf(GAP / 2);  // half a period ahead of the partner, once
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int q = 0; q < NBURST; ++q) {
            mf(GAP);
            burst();
        }
    }
    f32x4 s = acc[0];
#pragma unroll
    for (int i = 1; i < 8; ++i) s += acc[i];
    float t = 0.f;
#pragma unroll
    for (int i = 0; i < 8; ++i) t += v[i];
    if (s[0] + t == 123.f) out[threadIdx.x] = s[1];
}

template <int NV, int BL, int PH>
void run() {
    float* out; hipMalloc(&out, 4096);
    const int iters = 4000;
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    hipLaunchKernelGGL((k<NV, BL, PH>), dim3(256), dim3(512), 0, 0, out, iters, 1.f, 2.f);
    hipDeviceSynchronize();
    hipEventRecord(e0);
    for (int rep = 0; rep < 5; ++rep) hipLaunchKernelGGL((k<NV, BL, PH>), dim3(256), dim3(512), 0, 0, out, iters, 1.f, 2.f);
    hipEventRecord(e1); hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1); ms /= 5;
    const double flops = 256.0 * 8 * iters * 64 * 2048.0;
    printf("64 MFMAs + %2d VALU in bursts of %2d, phase mode %d: %.3f ms  frac %.3f\n", NV, BL, PH, ms, flops / (ms * 1e-3) / 1e12 / 157.3);
}

int main() {
    run<64, 64, 0>(); run<64, 64, 1>(); run<64, 64, 2>();
    run<64, 32, 0>(); run<64, 32, 1>(); run<64, 32, 2>();
    run<64, 16, 0>(); run<64, 16, 1>();
    run<64, 8, 0>(); run<64, 8, 1>();
    run<64, 2, 0>(); run<64, 2, 1>();
    run<32, 32, 0>(); run<32, 32, 1>();
    run<32, 8, 0>(); run<32, 8, 1>();
    return 0;
}

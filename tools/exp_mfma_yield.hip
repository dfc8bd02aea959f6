// exp_mfma_yield.hip — when does the PARTNER wave of a SIMD get VALU issue slots beside an fp32 MFMA stream?  (not product code)
//   build: hipcc --offload-arch=gfx950 -O3 -std=c++17 tools/exp_mfma_yield.hip -o tools/exp_mfma_yield
// 256 blocks x 8 waves: waves 0-3 (one per SIMD) issue N MFMAs in pattern Y, waves 4-7 (their SIMD partners) issue N x KV
// v_fmac_f32 on eight independent registers.  Each wave stamps its own start / end (s_memtime): if the partner's VALUs
// issue beside the MFMAs it finishes WITH the MFMA wave; if they are starved it finishes KV x N x ~5 cycles AFTER it.
#include <hip/hip_runtime.h>
#include <cstdio>
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef float f32x16 __attribute__((ext_vector_type(16)));

template <int Y, int KV, int PRIO>
__global__ __launch_bounds__(512, 1) void k(unsigned long long* stamps, float* out, int iters, float a0, float b0) {
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    float a = a0 + threadIdx.x, b = b0 - threadIdx.x;
    unsigned long long t0 = 0, t1 = 0;
    float res = 0.f;
    __syncthreads();
    if (wave < 4) {
        if (PRIO == 1) __builtin_amdgcn_s_setprio(0);
        f32x4 acc[8];
        f32x16 big[2];
#pragma unroll
        for (int i = 0; i < 8; ++i) acc[i] = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int i = 0; i < 2; ++i)
#pragma unroll
            for (int j = 0; j < 16; ++j) big[i][j] = 0.f;
        const f32x4 a4 = {a, b, a + 1.f, b + 1.f}, b4 = {b, a, b - 1.f, a - 1.f};  // (eight bf16 each for the bf16 forms)
        t0 = __builtin_readcyclecounter();
        for (int it = 0; it < iters; ++it) {
#pragma unroll
            for (int m = 0; m < 16; ++m) {
                if (Y == 6) {  // 32x32x2 (16 passes, 64 cycles): half as many for the same FLOPs
                    if (m & 1) continue;
                    asm volatile("v_mfma_f32_32x32x2_f32 %0, %1, %2, %0" : "+v"(big[(m >> 1) & 1]) : "v"(a), "v"(b));
                    continue;
                }
                if (Y == 10) {  // bf16 16x16x32 (16 cycles): the split form of the cohort GEMM (round 6, last session)
                    asm volatile("v_mfma_f32_16x16x32_bf16 %0, %1, %2, %0" : "+v"(acc[m & 7]) : "v"(a4), "v"(b4));
                    continue;
                }
                if (Y == 11) {  // bf16 32x32x16 (32 cycles)
                    if (m & 1) continue;
                    asm volatile("v_mfma_f32_32x32x16_bf16 %0, %1, %2, %0" : "+v"(big[(m >> 1) & 1]) : "v"(a4), "v"(b4));
                    continue;
                }
                if (Y == 7) {  // every MFMA depends on the previous one (one accumulator)
                    asm volatile("v_mfma_f32_16x16x4_f32 %0, %1, %2, %0" : "+v"(acc[0]) : "v"(a), "v"(b));
                    continue;
                }
                asm volatile("v_mfma_f32_16x16x4_f32 %0, %1, %2, %0" : "+v"(acc[m & 7]) : "v"(a), "v"(b));
                if (Y == 1) asm volatile("s_nop 7\n\ts_nop 7\n\ts_nop 7");
                if (Y == 2) asm volatile("s_nop 15");
                if (Y == 3) asm volatile("s_nop 15\n\ts_nop 7");
                if (Y == 4 && (m & 1)) asm volatile("s_nop 15\n\ts_nop 15\n\ts_nop 15");
                if (Y == 5) asm volatile("s_nop 3");
                if (Y == 8) asm volatile("s_sleep 0");
                if (Y == 9 && (m & 3) == 3) asm volatile("s_nop 15\n\ts_nop 15\n\ts_nop 15\n\ts_nop 15\n\ts_nop 15\n\ts_nop 15");
            }
        }
        t1 = __builtin_readcyclecounter();
        f32x4 s = acc[0];
#pragma unroll
        for (int i = 1; i < 8; ++i) s += acc[i];
        res = s[0] + big[0][0] + big[1][0];
    } else {
        if (PRIO == 1) __builtin_amdgcn_s_setprio(3);
        float v[8];
#pragma unroll
        for (int i = 0; i < 8; ++i) v[i] = a0 + i;
        t0 = __builtin_readcyclecounter();
        for (int it = 0; it < iters; ++it) {
#pragma unroll
            for (int m = 0; m < 16 * KV; ++m) asm volatile("v_fmac_f32 %0, %1, %2" : "+v"(v[m & 7]) : "v"(a), "v"(b));
        }
        t1 = __builtin_readcyclecounter();
#pragma unroll
        for (int i = 0; i < 8; ++i) res += v[i];
    }
    if (res == 123.f) out[threadIdx.x] = res;
    if (blockIdx.x == 100 && (threadIdx.x & 63) == 0) { stamps[2 * wave] = t0; stamps[2 * wave + 1] = t1; }
}

template <int Y, int KV, int PRIO>
void run(const char* what) {
    unsigned long long* st; float* out;
    hipMalloc(&st, 16 * 8); hipMalloc(&out, 4096);
    const int iters = 2000;
    for (int rep = 0; rep < 2; ++rep) hipLaunchKernelGGL((k<Y, KV, PRIO>), dim3(256), dim3(512), 0, 0, st, out, iters, 1.f, 2.f);
    hipDeviceSynchronize();
    unsigned long long h[16];
    hipMemcpy(h, st, sizeof(h), hipMemcpyDeviceToHost);
    const double n = 16.0 * iters;
    const double mf = (double)(h[1] - h[0]) / n, pa = (double)(h[9] - h[8]) / n, lag = (double)((long long)h[9] - (long long)h[1]) / n;
    printf("%-58s KV %d prio %d: MFMA wave %6.1f cyc/MFMA, partner %6.1f cyc per MFMA-slot (%5.1f per VALU), partner ends %+7.1f cyc/slot after\n",
           what, KV, PRIO, mf, pa, pa / KV, lag);
    hipFree(st); hipFree(out);
}

int main() {
    run<0, 4, 0>("back-to-back 16x16x4, 8 accumulators");
    run<0, 8, 0>("back-to-back 16x16x4, 8 accumulators");
    run<0, 4, 1>("back-to-back, MFMA wave prio 0 / partner prio 3");
    run<7, 4, 0>("every MFMA dependent on the previous (1 accumulator)");
    run<5, 4, 0>("MFMA + s_nop 3");
    run<2, 4, 0>("MFMA + s_nop 15");
    run<3, 4, 0>("MFMA + s_nop 15 + s_nop 7");
    run<1, 4, 0>("MFMA + 3 x s_nop 7");
    run<1, 6, 0>("MFMA + 3 x s_nop 7");
    run<4, 4, 0>("2 MFMAs + 3 x s_nop 15");
    run<9, 4, 0>("4 MFMAs + 6 x s_nop 15");
    run<8, 4, 0>("MFMA + s_sleep 0");
    run<6, 4, 0>("32x32x2 back-to-back (per 2 slots)");
    run<6, 8, 0>("32x32x2 back-to-back (per 2 slots)");
    run<10, 1, 0>("bf16 16x16x32 back-to-back, 8 accumulators");
    run<10, 2, 0>("bf16 16x16x32 back-to-back, 8 accumulators");
    run<10, 4, 0>("bf16 16x16x32 back-to-back, 8 accumulators");
    run<10, 2, 1>("bf16 16x16x32, MFMA wave prio 0 / partner prio 3");
    run<11, 2, 0>("bf16 32x32x16 back-to-back (per 2 slots)");
    run<11, 4, 0>("bf16 32x32x16 back-to-back (per 2 slots)");
    return 0;
}

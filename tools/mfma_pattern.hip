// mfma_pattern.hip — does the OPERAND PATTERN of the cohort kernel's K loop change the fp32-MFMA issue rate?
// (context for the cohort kernel's roofline; not product code)
// build: hipcc --offload-arch=gfx950 -O3 -std=c++17 tools/mfma_pattern.hip -o tools/mfma_pattern
#include <hip/hip_runtime.h>
#include <cstdio>
typedef float f32x4 __attribute__((ext_vector_type(4)));

// MODE 0: one a, one b for every MFMA (tools/mfma_peak.hip).  MODE 1: 4 A fragments x 2 B rows, 8 accumulators, the
// cohort kernel's order (kk major, then c, then g).  MODE 2: same with 16 accumulators (4 c x 4 g).  MODE 3: MODE 1 with
// the accumulators revisited in a different order (g major inside c).
template <int MODE, int WAVES>
__global__ __launch_bounds__(64 * WAVES) void k(const float* in, float* out, int iters) {
    f32x4 af[4], br[4];
#pragma unroll
    for (int c = 0; c < 4; ++c) {
        af[c] = *reinterpret_cast<const f32x4*>(in + 16 * c + 4 * (threadIdx.x & 3));
        br[c] = *reinterpret_cast<const f32x4*>(in + 64 + 16 * c + 4 * (threadIdx.x & 3));
    }
    f32x4 acc[16];
#pragma unroll
    for (int i = 0; i < 16; ++i) acc[i] = f32x4{0.f, 0.f, 0.f, 0.f};
    for (int it = 0; it < iters; ++it) {
        if (MODE == 0) {
#pragma unroll
            for (int r = 0; r < 4; ++r)
#pragma unroll
                for (int i = 0; i < 8; ++i) acc[i] = __builtin_amdgcn_mfma_f32_16x16x4f32(af[0][0], br[0][0], acc[i], 0, 0, 0);
        } else if (MODE == 1) {
#pragma unroll
            for (int kk = 0; kk < 4; ++kk)
#pragma unroll
                for (int c = 0; c < 4; ++c)
#pragma unroll
                    for (int g = 0; g < 2; ++g)
                        acc[2 * c + g] = __builtin_amdgcn_mfma_f32_16x16x4f32(af[c][kk], br[g][kk], acc[2 * c + g], 0, 0, 0);
        } else if (MODE == 2) {
#pragma unroll
            for (int kk = 0; kk < 4; ++kk)
#pragma unroll
                for (int c = 0; c < 4; ++c)
#pragma unroll
                    for (int g = 0; g < 4; ++g)
                        acc[4 * c + g] = __builtin_amdgcn_mfma_f32_16x16x4f32(af[c][kk], br[g][kk], acc[4 * c + g], 0, 0, 0);
        } else {
#pragma unroll
            for (int kk = 0; kk < 4; ++kk)
#pragma unroll
                for (int g = 0; g < 2; ++g)
#pragma unroll
                    for (int c = 0; c < 4; ++c)
                        acc[2 * c + g] = __builtin_amdgcn_mfma_f32_16x16x4f32(af[c][kk], br[g][kk], acc[2 * c + g], 0, 0, 0);
        }
        // keep the operands "new" for the compiler without touching them
#pragma unroll
        for (int c = 0; c < 4; ++c) { asm volatile("" : "+v"(af[c])); asm volatile("" : "+v"(br[c])); }
    }
    f32x4 s = acc[0];
#pragma unroll
    for (int i = 1; i < 16; ++i) s += acc[i];
    if (s[0] == 123.f) out[threadIdx.x] = s[1];
}

// the cohort kernel's K loop: A fragments re-read from LDS (4 x ds_read_b128 per 32 MFMAs, mid-step, pinned), 2 B rows
template <int NB, int WAVES, int VARIANT>
__global__ __launch_bounds__(64 * WAVES) void kl(const float* in, float* out, int iters) {
    __shared__ f32x4 lds[4 * NB * 64];
    for (int i = threadIdx.x; i < 4 * NB * 64; i += 64 * WAVES) lds[i] = *reinterpret_cast<const f32x4*>(in + 4 * (i & 15));
    __syncthreads();
    f32x4 br[2][NB];
#pragma unroll
    for (int g = 0; g < 2; ++g)
#pragma unroll
        for (int ks = 0; ks < NB; ++ks) br[g][ks] = *reinterpret_cast<const f32x4*>(in + 64 + 4 * ((threadIdx.x + ks + g) & 7));
    f32x4 acc[2][4];
#pragma unroll
    for (int g = 0; g < 2; ++g)
#pragma unroll
        for (int c = 0; c < 4; ++c) acc[g][c] = f32x4{0.f, 0.f, 0.f, 0.f};
    const f32x4* tb = lds + (threadIdx.x & 63);
    for (int it = 0; it < iters; ++it) {
        f32x4 af[2][4];
#pragma unroll
        for (int c = 0; c < 4; ++c) af[0][c] = tb[c * 64];
#pragma unroll
        for (int ks = 0; ks < NB; ++ks) {
            if (VARIANT == 0) {  // reads in the middle of the step
#pragma unroll
                for (int kk = 0; kk < 2; ++kk)
#pragma unroll
                    for (int c = 0; c < 4; ++c) {
                        acc[0][c] = __builtin_amdgcn_mfma_f32_16x16x4f32(af[ks & 1][c][kk], br[0][ks][kk], acc[0][c], 0, 0, 0);
                        acc[1][c] = __builtin_amdgcn_mfma_f32_16x16x4f32(af[ks & 1][c][kk], br[1][ks][kk], acc[1][c], 0, 0, 0);
                    }
                __builtin_amdgcn_sched_barrier(0);
                if (ks + 1 < NB) {
#pragma unroll
                    for (int c = 0; c < 4; ++c) af[(ks + 1) & 1][c] = tb[((ks + 1) * 4 + c) * 64];
                }
                __builtin_amdgcn_sched_barrier(0);
#pragma unroll
                for (int kk = 2; kk < 4; ++kk)
#pragma unroll
                    for (int c = 0; c < 4; ++c) {
                        acc[0][c] = __builtin_amdgcn_mfma_f32_16x16x4f32(af[ks & 1][c][kk], br[0][ks][kk], acc[0][c], 0, 0, 0);
                        acc[1][c] = __builtin_amdgcn_mfma_f32_16x16x4f32(af[ks & 1][c][kk], br[1][ks][kk], acc[1][c], 0, 0, 0);
                    }
            } else {  // one read between every four MFMA pairs (spread out)
#pragma unroll
                for (int kk = 0; kk < 4; ++kk) {
#pragma unroll
                    for (int c = 0; c < 4; ++c) {
                        acc[0][c] = __builtin_amdgcn_mfma_f32_16x16x4f32(af[ks & 1][c][kk], br[0][ks][kk], acc[0][c], 0, 0, 0);
                        acc[1][c] = __builtin_amdgcn_mfma_f32_16x16x4f32(af[ks & 1][c][kk], br[1][ks][kk], acc[1][c], 0, 0, 0);
                    }
                    __builtin_amdgcn_sched_barrier(0);
                    if (ks + 1 < NB) af[(ks + 1) & 1][kk] = tb[((ks + 1) * 4 + kk) * 64];
                    __builtin_amdgcn_sched_barrier(0);
                }
            }
        }
    }
    f32x4 s = acc[0][0];
#pragma unroll
    for (int c = 1; c < 4; ++c) s += acc[0][c] + acc[1][c];
    if (s[0] == 123.f) out[threadIdx.x] = s[1];
}

template <int NB, int WAVES, int VARIANT>
void runl(const char* name) {
    float *in, *out;
    hipMalloc(&in, 4096); hipMalloc(&out, 4096);
    hipMemset(in, 0, 4096);
    const int iters = 400;
    dim3 grid(256), block(64 * WAVES);
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    hipLaunchKernelGGL((kl<NB, WAVES, VARIANT>), grid, block, 0, 0, in, out, iters);
    hipDeviceSynchronize();
    hipEventRecord(e0);
    for (int rep = 0; rep < 5; ++rep) hipLaunchKernelGGL((kl<NB, WAVES, VARIANT>), grid, block, 0, 0, in, out, iters);
    hipEventRecord(e1); hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1); ms /= 5;
    const double flops = (double)grid.x * WAVES * iters * NB * 32 * 2048.0;
    printf("%-44s waves/SIMD %d: %.3f ms  %.1f TFLOP/s  frac of 157.3 = %.3f\n", name, WAVES / 4, ms,
           flops / (ms * 1e-3) / 1e12, flops / (ms * 1e-3) / 1e12 / 157.3);
}

template <int MODE, int WAVES>
void run(int blocks_per_cu, const char* name) {
    float *in, *out;
    hipMalloc(&in, 4096); hipMalloc(&out, 4096);
    hipMemset(in, 0, 4096);
    const int iters = 4000;
    const int per_it = MODE == 2 ? 64 : 32;
    dim3 grid(256 * blocks_per_cu), block(64 * WAVES);
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    hipLaunchKernelGGL((k<MODE, WAVES>), grid, block, 0, 0, in, out, iters);
    hipDeviceSynchronize();
    hipEventRecord(e0);
    for (int rep = 0; rep < 5; ++rep) hipLaunchKernelGGL((k<MODE, WAVES>), grid, block, 0, 0, in, out, iters);
    hipEventRecord(e1); hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1); ms /= 5;
    const double flops = (double)grid.x * WAVES * iters * per_it * 2048.0;
    printf("%-44s waves/SIMD %d: %.3f ms  %.1f TFLOP/s  frac of 157.3 = %.3f\n", name, blocks_per_cu * WAVES / 4, ms,
           flops / (ms * 1e-3) / 1e12, flops / (ms * 1e-3) / 1e12 / 157.3);
}

int main() {
    run<0, 4>(1, "same a, b; 8 accumulators");
    run<0, 8>(1, "same a, b; 8 accumulators");
    run<1, 4>(1, "4 A x 2 B, 8 acc (cohort order)");
    run<1, 8>(1, "4 A x 2 B, 8 acc (cohort order)");
    run<1, 4>(3, "4 A x 2 B, 8 acc (cohort order)");
    run<3, 8>(1, "4 A x 2 B, 8 acc (g major)");
    run<2, 4>(1, "4 A x 4 B, 16 acc");
    run<2, 8>(1, "4 A x 4 B, 16 acc");
    runl<11, 4, 0>("K loop, A from LDS mid-step");
    runl<11, 8, 0>("K loop, A from LDS mid-step");
    runl<11, 4, 1>("K loop, A from LDS spread");
    runl<11, 8, 1>("K loop, A from LDS spread");
    return 0;
}

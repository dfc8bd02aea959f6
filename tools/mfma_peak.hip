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


// Does the DATA matter?  The same loop with full-entropy operands (random mantissas) against the small integers
// above, and the shader clock actually held during the run (s_memtime ticks per 100 MHz s_memrealtime tick).
template <int NACC>
__global__ __launch_bounds__(256) void kd(const float* in, float* out, unsigned long long* clk, int iters) {
    f32x4 acc[NACC];
#pragma unroll
    for (int i = 0; i < NACC; ++i) acc[i] = f32x4{0.f, 0.f, 0.f, 0.f};
    const float a = in[threadIdx.x], b = in[256 + threadIdx.x];
    const unsigned long long c0 = __builtin_readcyclecounter(), r0 = __builtin_amdgcn_s_memrealtime();
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int r = 0; r < 8; ++r)
#pragma unroll
            for (int i = 0; i < NACC; ++i) acc[i] = __builtin_amdgcn_mfma_f32_16x16x4f32(a, b, acc[i], 0, 0, 0);
    }
    f32x4 s = acc[0];
#pragma unroll
    for (int i = 1; i < NACC; ++i) s += acc[i];
    const unsigned long long c1 = __builtin_readcyclecounter(), r1 = __builtin_amdgcn_s_memrealtime();
    if (s[0] == 123.f) out[threadIdx.x] = s[1];
    if (threadIdx.x == 0) { clk[2 * blockIdx.x] = c1 - c0; clk[2 * blockIdx.x + 1] = r1 - r0; }
}

template <int NACC>
void rund(int waves_per_simd, bool random, int iters, int reps) {
    float *in, *out; unsigned long long* clk;
    dim3 grid(256 * waves_per_simd), block(256);
    hipMalloc(&in, 512 * 4); hipMalloc(&out, 4096); hipMalloc(&clk, grid.x * 16);
    float h[512];
    unsigned x = 12345u;
    for (int i = 0; i < 512; ++i) {
        x = x * 1664525u + 1013904223u;
        h[i] = random ? ((float)(x >> 8) / 16777216.0f - 0.5f) : (float)(i % 7 - 3);
    }
    hipMemcpy(in, h, sizeof h, hipMemcpyHostToDevice);
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    hipLaunchKernelGGL(kd<NACC>, grid, block, 0, 0, in, out, clk, iters);
    hipDeviceSynchronize();
    hipEventRecord(e0);
    for (int rep = 0; rep < reps; ++rep) hipLaunchKernelGGL(kd<NACC>, grid, block, 0, 0, in, out, clk, iters);
    hipEventRecord(e1); hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1); ms /= reps;
    unsigned long long hc[2]; hipMemcpy(hc, clk, 16, hipMemcpyDeviceToHost);
    const double flops = (double)grid.x * 4 * iters * 8 * NACC * 2048.0;
    printf("%-8s operands, waves/SIMD %d, %6.2f ms/launch x %d: %.1f TFLOP/s  frac %.3f   cycle counter %.1f MHz\n",
           random ? "random" : "integer", waves_per_simd, ms, reps, flops / (ms * 1e-3) / 1e12,
           flops / (ms * 1e-3) / 1e12 / 157.3, (double)hc[0] / ((double)hc[1] / 100.0));
}

// The tile loop of the GEMM kernels without any memory: 4 x 4 accumulators, 4 + 4 float4 fragments, the MFMA order of
// the product kernels (kk outer, then ca, cb).  ORDER 0: kk, ca, cb;  1: ca, cb, kk (4 dependent in a row);  2: kk, cb, ca
template <int ORDER>
__global__ __launch_bounds__(256) void kt(const float* in, float* out, int iters) {
    f32x4 acc[4][4], fa[4], fb[4];
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        fa[i] = *reinterpret_cast<const f32x4*>(in + 4 * ((threadIdx.x + 17 * i) & 127));
        fb[i] = *reinterpret_cast<const f32x4*>(in + 4 * ((threadIdx.x + 29 * i + 5) & 127));
#pragma unroll
        for (int j = 0; j < 4; ++j) acc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};
    }
    for (int it = 0; it < iters; ++it) {
        if (ORDER == 0) {
#pragma unroll
            for (int kk = 0; kk < 4; ++kk)
#pragma unroll
                for (int ca = 0; ca < 4; ++ca)
#pragma unroll
                    for (int cb = 0; cb < 4; ++cb)
                        acc[ca][cb] = __builtin_amdgcn_mfma_f32_16x16x4f32(fb[cb][kk], fa[ca][kk], acc[ca][cb], 0, 0, 0);
        } else if (ORDER == 2) {
#pragma unroll
            for (int kk = 0; kk < 4; ++kk)
#pragma unroll
                for (int cb = 0; cb < 4; ++cb)
#pragma unroll
                    for (int ca = 0; ca < 4; ++ca)
                        acc[ca][cb] = __builtin_amdgcn_mfma_f32_16x16x4f32(fb[cb][kk], fa[ca][kk], acc[ca][cb], 0, 0, 0);
        } else {
#pragma unroll
            for (int ca = 0; ca < 4; ++ca)
#pragma unroll
                for (int cb = 0; cb < 4; ++cb)
#pragma unroll
                    for (int kk = 0; kk < 4; ++kk)
                        acc[ca][cb] = __builtin_amdgcn_mfma_f32_16x16x4f32(fb[cb][kk], fa[ca][kk], acc[ca][cb], 0, 0, 0);
        }
        // keep the fragments "changing" so that nothing is hoisted (opaque, no instructions)
#pragma unroll
        for (int i = 0; i < 4; ++i) asm volatile("" : "+v"(fa[i]), "+v"(fb[i]));
    }
    f32x4 s = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int j = 0; j < 4; ++j) s += acc[i][j];
    if (s[0] == 123.f) out[threadIdx.x] = s[1];
}

template <int ORDER>
void runt(int waves_per_simd) {
    float *in, *out;
    hipMalloc(&in, 512 * 4); hipMalloc(&out, 4096);
    float h[512];
    unsigned x = 12345u;
    for (int i = 0; i < 512; ++i) { x = x * 1664525u + 1013904223u; h[i] = (float)(x >> 8) / 16777216.0f - 0.5f; }
    hipMemcpy(in, h, sizeof h, hipMemcpyHostToDevice);
    const int iters = 4000;
    dim3 grid(256 * waves_per_simd), block(256);
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    hipLaunchKernelGGL(kt<ORDER>, grid, block, 0, 0, in, out, iters);
    hipDeviceSynchronize();
    hipEventRecord(e0);
    for (int rep = 0; rep < 5; ++rep) hipLaunchKernelGGL(kt<ORDER>, grid, block, 0, 0, in, out, iters);
    hipEventRecord(e1); hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1); ms /= 5;
    const double flops = (double)grid.x * 4 * iters * 64 * 2048.0;
    printf("tile loop order %d, waves/SIMD %d: %.3f ms  %.1f TFLOP/s  frac %.3f\n", ORDER, waves_per_simd, ms,
           flops / (ms * 1e-3) / 1e12, flops / (ms * 1e-3) / 1e12 / 157.3);
}

// The same tile loop with its 8 fragments re-read from LDS every iteration (read-only image, no barrier): what do
// ds_read_b128 -> MFMA operand hand-offs cost?  MODE 0: all 8 reads up front; 1: reads for the NEXT iteration issued
// before this iteration's MFMAs (software pipelined, 64 fragment registers).
template <int MODE>
__global__ __launch_bounds__(256) void kl(const float* in, float* out, int iters) {
    __shared__ f32x4 img[2][512];
    for (int i = threadIdx.x; i < 1024; i += 256) (&img[0][0])[i] = *reinterpret_cast<const f32x4*>(in + 4 * (i & 127));
    __syncthreads();
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const f32x4* fra = &img[0][(wave >> 1) * 256 + lane];
    const f32x4* frb = &img[1][(wave & 1) * 256 + lane];
    f32x4 acc[4][4], fa[4], fb[4], na[4], nb[4];
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        fa[i] = fra[64 * i]; fb[i] = frb[64 * i];
#pragma unroll
        for (int j = 0; j < 4; ++j) acc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};
    }
    for (int it = 0; it < iters; ++it) {
        if (MODE == 0) {
#pragma unroll
            for (int i = 0; i < 4; ++i) { fa[i] = fra[64 * i]; fb[i] = frb[64 * i]; }
        } else {
#pragma unroll
            for (int i = 0; i < 4; ++i) { na[i] = fra[64 * i]; nb[i] = frb[64 * i]; }
        }
        asm volatile("" ::: "memory");
#pragma unroll
        for (int kk = 0; kk < 4; ++kk)
#pragma unroll
            for (int ca = 0; ca < 4; ++ca)
#pragma unroll
                for (int cb = 0; cb < 4; ++cb)
                    acc[ca][cb] = __builtin_amdgcn_mfma_f32_16x16x4f32(fb[cb][kk], fa[ca][kk], acc[ca][cb], 0, 0, 0);
        if (MODE == 1) {
#pragma unroll
            for (int i = 0; i < 4; ++i) { fa[i] = na[i]; fb[i] = nb[i]; }
        }
        asm volatile("" ::: "memory");
    }
    f32x4 s = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int j = 0; j < 4; ++j) s += acc[i][j];
    if (s[0] == 123.f) out[threadIdx.x] = s[1];
}

template <int MODE>
void runl(int waves_per_simd) {
    float *in, *out;
    hipMalloc(&in, 512 * 4); hipMalloc(&out, 4096);
    float h[512];
    unsigned x = 12345u;
    for (int i = 0; i < 512; ++i) { x = x * 1664525u + 1013904223u; h[i] = (float)(x >> 8) / 16777216.0f - 0.5f; }
    hipMemcpy(in, h, sizeof h, hipMemcpyHostToDevice);
    const int iters = 4000;
    dim3 grid(256 * waves_per_simd), block(256);
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    hipLaunchKernelGGL(kl<MODE>, grid, block, 0, 0, in, out, iters);
    hipDeviceSynchronize();
    hipEventRecord(e0);
    for (int rep = 0; rep < 5; ++rep) hipLaunchKernelGGL(kl<MODE>, grid, block, 0, 0, in, out, iters);
    hipEventRecord(e1); hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1); ms /= 5;
    const double flops = (double)grid.x * 4 * iters * 64 * 2048.0;
    printf("tile loop + 8 ds_read_b128 per 64 MFMAs (%s), waves/SIMD %d: %.3f ms  %.1f TFLOP/s  frac %.3f\n",
           MODE ? "pipelined" : "read then multiply", waves_per_simd, ms, flops / (ms * 1e-3) / 1e12,
           flops / (ms * 1e-3) / 1e12 / 157.3);
}

int main() {
    runl<0>(1); runl<0>(2); runl<0>(3); runl<1>(2); runl<1>(3);
    runt<0>(1); runt<0>(2); runt<0>(3); runt<2>(2); runt<1>(2);
    rund<8>(2, false, 4000, 5); rund<8>(2, true, 4000, 5); rund<8>(2, false, 40000, 20); rund<8>(2, true, 40000, 20);
    rund<16>(3, true, 20000, 20);

    run<4>(1, "4 independent acc"); run<4>(2, "4 independent acc"); run<4>(4, "4 independent acc");
    run<8>(1, "8 independent acc"); run<8>(2, "8 independent acc");
    run<16>(1, "16 independent acc"); run<16>(2, "16 independent acc");
    runv<8, 0>(2); runv<8, 16>(2); runv<8, 32>(2); runv<8, 64>(2); runv<8, 32>(1); runv<8, 32>(4);
    return 0;
}

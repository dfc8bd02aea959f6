// exp_issue_cost.hip — what does ONE instruction of another kind cost when it sits between the MFMAs of a wave that is alone on
// its SIMD?  (context for the rooflines in DESIGN.md: the forward kernel issues 960 MFMAs and ~630 other instructions per tile,
// the AS-norm statistics kernel ~400 per 320; not product code)
//
// One block of 256 threads per CU (one wave per SIMD), a loop of 32 v_mfma_f32_16x16x4_f32 over 8 independent accumulators,
// written as assembly so that nothing moves; after every MFMA (or every second / fourth one) NF filler instructions of one kind.
// Cycles by s_memtime around the loop, wave 0 of block 0.  Prints cycles per MFMA and the cost per filler against the empty loop.
// build: hipcc --offload-arch=gfx950 -O3 -std=c++17 tools/exp_issue_cost.hip -o tools/exp_issue_cost
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>
typedef float f32x4 __attribute__((ext_vector_type(4)));

enum Fill {
    F_NONE, F_SNOP, F_WAIT_LGKM, F_WAIT_VM, F_VMOV, F_FMAC_INDEP, F_FMAC_CHAIN, F_PKADD, F_DSREAD, F_SMOV, F_SMOV_EXEC, F_VCMP_SGPR,
    F_VCMPX, F_ACCREAD, F_ACCWRITE, F_STORE, F_LOAD, F_READLANE, F_COUNT
};
static const char* kNames[F_COUNT] = {"(none)", "s_nop 0", "s_waitcnt lgkmcnt(0), nothing pending", "s_waitcnt vmcnt(0), nothing pending",
                                      "v_mov_b32", "v_fmac_f32, independent", "v_fmac_f32, one dependent chain", "v_pk_add_f32",
                                      "ds_read_b128 (waited for at the end of the 32)", "s_mov_b64 sgpr, exec", "s_mov_b64 exec, -1",
                                      "v_cmp_le_f32 -> sgpr pair", "v_cmpx_le_f32 (writes exec) + s_mov_b64 exec, -1",
                                      "v_accvgpr_read_b32 (an AGPR no MFMA touches)", "v_accvgpr_write_b32", "global_store_dword (own slot, L2-resident)",
                                      "global_load_dword (own slot; waited for at the end of the 32)", "v_readlane_b32"};

template <int FILL, int EVERY, int NF>
__global__ __launch_bounds__(256, 1) void k(const float* in, float* out, unsigned long long* cyc, int iters) {
    __shared__ f32x4 lds[256];
    const int tid = threadIdx.x;
    lds[tid] = f32x4{1.f, 2.f, 3.f, 4.f};
    __syncthreads();
    float a = in[tid & 63], b = in[64 + (tid & 63)];
    f32x4 acc[8];
#pragma unroll
    for (int i = 0; i < 8; ++i) acc[i] = f32x4{0.f, 0.f, 0.f, 0.f};
    float f[8] = {1.f, 2.f, 3.f, 4.f, 5.f, 6.f, 7.f, 8.f};
    float x = in[128 + tid], y = 0.5f, z = 0.25f;
    f32x4 dsv = {0.f, 0.f, 0.f, 0.f};
    float ag = 0.f, ldv = 0.f;
    asm volatile("v_accvgpr_write_b32 %0, %1" : "=a"(ag) : "v"(x));
    float* slot = out + 4096 + blockIdx.x * 256 + tid;
    const f32x4* lp = &lds[tid];
    unsigned long long sg = 0;
    int rl = 0;
    const unsigned long long t0 = __builtin_readcyclecounter();
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int m = 0; m < 32; ++m) {
            asm volatile("v_mfma_f32_16x16x4_f32 %0, %1, %2, %0" : "+v"(acc[m & 7]) : "v"(a), "v"(b));
            if (m % EVERY == EVERY - 1) {
#pragma unroll
                for (int n = 0; n < NF; ++n) {
                    const int j = (m + n) & 7;
                    if (FILL == F_SNOP) asm volatile("s_nop 0");
                    if (FILL == F_WAIT_LGKM) asm volatile("s_waitcnt lgkmcnt(0)");
                    if (FILL == F_WAIT_VM) asm volatile("s_waitcnt vmcnt(0)");
                    if (FILL == F_VMOV) asm volatile("v_mov_b32 %0, %1" : "=v"(f[j]) : "v"(y));
                    if (FILL == F_FMAC_INDEP) asm volatile("v_fmac_f32 %0, %1, %2" : "+v"(f[j]) : "v"(y), "v"(z));
                    if (FILL == F_FMAC_CHAIN) asm volatile("v_fmac_f32 %0, %1, %0" : "+v"(f[0]) : "v"(y));
                    if (FILL == F_PKADD) asm volatile("v_pk_add_f32 %0, %0, %1" : "+v"(*reinterpret_cast<double*>(&f[2 * (j & 3)])) : "v"(*reinterpret_cast<double*>(&dsv)));
                    if (FILL == F_DSREAD) asm volatile("ds_read_b128 %0, %1" : "=v"(dsv) : "v"((unsigned)(size_t)lp));
                    if (FILL == F_SMOV) asm volatile("s_mov_b64 %0, exec" : "=s"(sg));
                    if (FILL == F_SMOV_EXEC) asm volatile("s_mov_b64 exec, -1");
                    if (FILL == F_VCMP_SGPR) asm volatile("v_cmp_le_f32 %0, %1, %2" : "=s"(sg) : "v"(x), "v"(y));
                    if (FILL == F_VCMPX) asm volatile("v_cmpx_le_f32 vcc, %0, %1\n\ts_mov_b64 exec, -1" ::"v"(y), "v"(x) : "vcc");
                    if (FILL == F_ACCREAD) asm volatile("v_accvgpr_read_b32 %0, %1" : "=v"(f[j]) : "a"(ag));
                    if (FILL == F_ACCWRITE) asm volatile("v_accvgpr_write_b32 %0, %1" : "=a"(ag) : "v"(y));
                    if (FILL == F_STORE) asm volatile("global_store_dword %0, %1, off" ::"v"(slot), "v"(y) : "memory");
                    if (FILL == F_LOAD) asm volatile("global_load_dword %0, %1, off" : "=v"(ldv) : "v"(slot) : "memory");
                    if (FILL == F_READLANE) asm volatile("v_readlane_b32 %0, %1, 3" : "=s"(rl) : "v"(x));
                }
            }
        }
        if (FILL == F_DSREAD) asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        if (FILL == F_LOAD || FILL == F_STORE) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    }
    asm volatile("s_nop 15\n\ts_nop 7");
    const unsigned long long t1 = __builtin_readcyclecounter();
    float s = dsv[0] + ag + ldv + (float)sg + (float)rl;
#pragma unroll
    for (int i = 0; i < 8; ++i) s += acc[i][0] + acc[i][1] + acc[i][2] + acc[i][3] + f[i];
    out[blockIdx.x * 256 + tid] = s;
    if (blockIdx.x == 0 && tid == 0) cyc[0] = t1 - t0;
}

template <int FILL, int EVERY, int NF>
static double run(const float* in, float* out, unsigned long long* cyc, int iters) {
    hipLaunchKernelGGL((k<FILL, EVERY, NF>), dim3(256), dim3(256), 0, 0, in, out, cyc, iters);  // warm
    hipLaunchKernelGGL((k<FILL, EVERY, NF>), dim3(256), dim3(256), 0, 0, in, out, cyc, iters);
    hipDeviceSynchronize();
    unsigned long long h = 0;
    hipMemcpy(&h, cyc, 8, hipMemcpyDeviceToHost);
    return (double)h / ((double)iters * 32.0);
}

template <int FILL>
static void row(const float* in, float* out, unsigned long long* cyc, int iters, double base) {
    const double c1 = run<FILL, 1, 1>(in, out, cyc, iters), c2 = run<FILL, 2, 1>(in, out, cyc, iters), c4 = run<FILL, 4, 1>(in, out, cyc, iters),
                 c12 = run<FILL, 1, 2>(in, out, cyc, iters), c14 = run<FILL, 1, 4>(in, out, cyc, iters);
    printf("%-62s  %6.1f %6.1f %6.1f %6.1f %6.1f   | %5.1f %5.1f %5.1f %5.1f %5.1f\n", kNames[FILL], c4, c2, c1, c12, c14, (c4 - base) * 4, (c2 - base) * 2,
           c1 - base, (c12 - base) / 2, (c14 - base) / 4);
}

// Two waves per SIMD: waves 0 - 3 issue nothing but MFMAs, waves 4 - 7 nothing but VPER independent VALU instructions per MFMA
// of the partner (or exec-masked stores).  Does the partner's stream cost the MFMA wave anything?
template <int VPER, int KIND>
__global__ __launch_bounds__(512, 1) void k2(const float* in, float* out, unsigned long long* cyc, int iters) {
    const int tid = threadIdx.x, wave = tid >> 6;
    float a = in[tid & 63], b = in[64 + (tid & 63)];
    f32x4 acc[8];
#pragma unroll
    for (int i = 0; i < 8; ++i) acc[i] = f32x4{0.f, 0.f, 0.f, 0.f};
    float f[8] = {1.f, 2.f, 3.f, 4.f, 5.f, 6.f, 7.f, 8.f};
    float y = 0.5f, z = 0.25f;
    float* slot = out + 4096 + blockIdx.x * 512 + tid;
    __syncthreads();
    const unsigned long long t0 = __builtin_readcyclecounter();
    if (wave < 4) {
        for (int it = 0; it < iters; ++it)
#pragma unroll
            for (int m = 0; m < 32; ++m) asm volatile("v_mfma_f32_16x16x4_f32 %0, %1, %2, %0" : "+v"(acc[m & 7]) : "v"(a), "v"(b));
        asm volatile("s_nop 15\n\ts_nop 7");
    } else {
        for (int it = 0; it < iters; ++it)
#pragma unroll
            for (int m = 0; m < 32 * VPER; ++m) {
                if (KIND == 0) asm volatile("v_fmac_f32 %0, %1, %2" : "+v"(f[m & 7]) : "v"(y), "v"(z));
                if (KIND == 1) asm volatile("global_store_dword %0, %1, off" ::"v"(slot), "v"(y) : "memory");
                if (KIND == 2) asm volatile("v_cmpx_le_f32 vcc, %0, %1\n\ts_mov_b64 exec, -1" ::"v"(y), "v"(z) : "vcc");
            }
    }
    const unsigned long long t1 = __builtin_readcyclecounter();
    float s = 0.f;
#pragma unroll
    for (int i = 0; i < 8; ++i) s += acc[i][0] + acc[i][1] + acc[i][2] + acc[i][3] + f[i];
    out[blockIdx.x * 512 + tid] = s;
    if (blockIdx.x == 0 && (tid == 0 || tid == 256)) cyc[wave >> 2] = t1 - t0;
}

template <int VPER, int KIND>
static void row2(const char* what, const float* in, float* out, unsigned long long* cyc, int iters) {
    hipLaunchKernelGGL((k2<VPER, KIND>), dim3(256), dim3(512), 0, 0, in, out, cyc, iters);
    hipLaunchKernelGGL((k2<VPER, KIND>), dim3(256), dim3(512), 0, 0, in, out, cyc, iters);
    hipDeviceSynchronize();
    unsigned long long h[2];
    hipMemcpy(h, cyc, 16, hipMemcpyDeviceToHost);
    const double n = (double)iters * 32.0;
    printf("partner issues %d x %-28s per MFMA: MFMA wave %6.1f cycles per MFMA, partner done after %6.1f cycles per MFMA (%5.1f per instruction)\n", VPER, what,
           (double)h[0] / n, (double)h[1] / n, (double)h[1] / n / VPER);
}

int main() {
    float *in, *out;
    unsigned long long* cyc;
    hipMalloc(&in, 4096 * 4);
    hipMalloc(&out, (4096 + 256 * 512) * 4);
    hipMalloc(&cyc, 64);
    std::vector<float> h(4096, 0.001f);
    hipMemcpy(in, h.data(), 4096 * 4, hipMemcpyHostToDevice);
    const int iters = 2000;
    const double base = run<F_NONE, 1, 1>(in, out, cyc, iters);
    printf("one wave per SIMD, v_mfma_f32_16x16x4_f32 back to back: %.2f cycles per MFMA (s_memtime)\n", base);
    printf("%-62s  cycles per MFMA with 1/4, 1/2, 1, 2, 4 fillers per MFMA  | cycles per filler\n", "filler");
    row<F_SNOP>(in, out, cyc, iters, base);
    row<F_WAIT_LGKM>(in, out, cyc, iters, base);
    row<F_WAIT_VM>(in, out, cyc, iters, base);
    row<F_VMOV>(in, out, cyc, iters, base);
    row<F_FMAC_INDEP>(in, out, cyc, iters, base);
    row<F_FMAC_CHAIN>(in, out, cyc, iters, base);
    row<F_PKADD>(in, out, cyc, iters, base);
    row<F_DSREAD>(in, out, cyc, iters, base);
    row<F_SMOV>(in, out, cyc, iters, base);
    row<F_SMOV_EXEC>(in, out, cyc, iters, base);
    row<F_VCMP_SGPR>(in, out, cyc, iters, base);
    row<F_VCMPX>(in, out, cyc, iters, base);
    row<F_ACCREAD>(in, out, cyc, iters, base);
    row<F_ACCWRITE>(in, out, cyc, iters, base);
    row<F_STORE>(in, out, cyc, iters, base);
    row<F_LOAD>(in, out, cyc, iters, base);
    row<F_READLANE>(in, out, cyc, iters, base);
    printf("\ntwo waves per SIMD\n");
    row2<1, 0>("v_fmac_f32", in, out, cyc, iters);
    row2<2, 0>("v_fmac_f32", in, out, cyc, iters);
    row2<4, 0>("v_fmac_f32", in, out, cyc, iters);
    row2<8, 0>("v_fmac_f32", in, out, cyc, iters);
    row2<1, 1>("global_store_dword", in, out, cyc, iters);
    row2<1, 2>("v_cmpx + s_mov exec", in, out, cyc, iters);
    row2<2, 2>("v_cmpx + s_mov exec", in, out, cyc, iters);
    return 0;
}

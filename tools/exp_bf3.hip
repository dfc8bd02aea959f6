// exp_bf3.hip — timing, fence-wait accounting and ablations of the experimental 64-rows-per-wave split-bf16 kernel
// (tools/exp_bf3w_kernel.h), checked against the shipped kernel's scores: build once per NPLDA_BF3W_ABL mask.
// build: hipcc --offload-arch=gfx950 -O3 -std=c++17 -fno-slp-vectorize -DNPLDA_BF3W_ABL=0 tools/exp_bf3.hip -o tools/exp_bf3_0
// run:   tools/exp_bf3_0 [D=150] [B=1048576]      (masks != 0 give wrong scores: timing only)
#define NPLDA_BF3W_STAMPS 1
#include <cstdio>
#include <cstdlib>
#include <cmath>
#include <vector>
#include "exp_bf3w_kernel.h"

using namespace nplda;

#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s at %s:%d\n", hipGetErrorString(e), __FILE__, __LINE__); exit(1); } } while (0)

__global__ void fill_rand(float* p, size_t n, unsigned seed) {
    size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    const size_t stride = (size_t)gridDim.x * blockDim.x;
    for (; i < n; i += stride) {
        unsigned long long h = (i + 1) * 0x9E3779B97F4A7C15ull + seed * 0xD1B54A32D192ED03ull;
        h ^= h >> 32; h *= 0xD6E8FEB86659FD93ull; h ^= h >> 32; h *= 0xD6E8FEB86659FD93ull; h ^= h >> 32;
        float u = 0.f;
        for (int k = 0; k < 4; ++k) u += (float)((h >> (16 * k)) & 0xffff) / 65536.0f;
        p[i] = (u - 2.0f) * 1.7320508f;
    }
}

template <int NB>
void run(const Bf3wArgs& a, int tiles, int cus) {
    hipLaunchKernelGGL((nplda_fwd_bf3w_kernel<NB, MODE_PAIR>), dim3(cus), dim3(256), 0, 0, a, tiles);
}

int main(int argc, char** argv) {
    const int D = argc > 1 ? atoi(argv[1]) : 150;
    const long long B = argc > 2 ? atoll(argv[2]) : 1 << 20;
    const int D0 = 512;
    const Bf3wLayout L = bf3w_layout(D0, D, D);
    float *x1, *x2, *s, *packed, *W1, *b1, *W2, *b2, *Ps, *Q;
    CK(hipMalloc(&x1, B * D0 * 4)); CK(hipMalloc(&x2, B * D0 * 4)); CK(hipMalloc(&s, B * 4));
    CK(hipMalloc(&packed, L.total * 4));
    CK(hipMalloc(&W1, D * D0 * 4)); CK(hipMalloc(&b1, D * 4)); CK(hipMalloc(&W2, D * D * 4));
    CK(hipMalloc(&b2, D * 4)); CK(hipMalloc(&Ps, D * 4)); CK(hipMalloc(&Q, D * 4));
    fill_rand<<<4096, 256>>>(x1, (size_t)B * D0, 1); fill_rand<<<4096, 256>>>(x2, (size_t)B * D0, 2);
    fill_rand<<<64, 256>>>(W1, (size_t)D * D0, 3); fill_rand<<<1, 256>>>(b1, D, 4);
    fill_rand<<<64, 256>>>(W2, (size_t)D * D, 5); fill_rand<<<1, 256>>>(b2, D, 6);
    fill_rand<<<1, 256>>>(Ps, D, 7); fill_rand<<<1, 256>>>(Q, D, 8);
    const size_t nthreads = L.ob1 / 4 + (L.total - L.ob1);
    bf3w_pack_kernel<<<(unsigned)((nthreads + 255) / 256), 256>>>(W1, b1, W2, b2, Ps, Q, L, packed);
    CK(hipDeviceSynchronize());
    Bf3wArgs a = {};
    a.xa = x1; a.xb = x2; a.n = B; a.ldx = D0; a.img = packed; a.out_s = s;
    a.D0 = L.D0; a.KC1 = L.KC1; a.ob1 = L.ob1; a.ob2 = L.ob2; a.oQ = L.oQ; a.oP = L.oP; a.oW1c = L.oW1c; a.oW2c = L.oW2c;
    int cus = 256;
    CK(hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, 0));
    const int tiles = (int)((B + 127) / 128);
    auto go = [&]() {
        if (L.NB == 10) run<10>(a, tiles, cus);
        else if (L.NB == 11) run<11>(a, tiles, cus);
        else { printf("NB %d not built here\n", L.NB); exit(1); }
    };
    if (NPLDA_BF3W_ABL == 0) {  // scores against the shipped 16-pair kernel on the same inputs
        const Bf3Layout L0 = bf3_layout(D0, D, D);
        float *packed0, *s0;
        CK(hipMalloc(&packed0, L0.total * 4)); CK(hipMalloc(&s0, B * 4));
        const size_t nt0 = L0.ob1 / 4 + (L0.total - L0.ob1);
        nplda_pack_bf16x3_kernel<<<(unsigned)((nt0 + 255) / 256), 256>>>(W1, b1, W2, b2, Ps, Q, L0, packed0);
        Bf3Args a0 = {};
        a0.xa = x1; a0.xb = x2; a0.n = B; a0.ldx = D0; a0.img = packed0; a0.out_s = s0;
        a0.D0 = L0.D0; a0.KC1 = L0.KC1; a0.oW2 = L0.oW2; a0.ob1 = L0.ob1; a0.ob2 = L0.ob2; a0.oQ = L0.oQ; a0.oP = L0.oP;
        const unsigned g0 = (unsigned)((B + 127) / 128);
        if (L.NB == 10) hipLaunchKernelGGL((nplda_fwd_bf16x3_kernel<10, MODE_PAIR, 8, 2>), dim3(g0), dim3(512), 0, 0, a0);
        else hipLaunchKernelGGL((nplda_fwd_bf16x3_kernel<11, MODE_PAIR, 8, 2>), dim3(g0), dim3(512), 0, 0, a0);
        go();
        CK(hipDeviceSynchronize());
        std::vector<float> h0(B), h1(B);
        CK(hipMemcpy(h0.data(), s0, B * 4, hipMemcpyDeviceToHost)); CK(hipMemcpy(h1.data(), s, B * 4, hipMemcpyDeviceToHost));
        double md = 0, ms0 = 0;
        for (long long i = 0; i < B; ++i) { md = fmax(md, fabs((double)h0[i] - h1[i])); ms0 = fmax(ms0, fabs((double)h0[i])); }
        printf("max |s - s(shipped kernel)| = %.3g over %lld pairs (max |s| = %.3g)\n", md, B, ms0);
    }
    hipEvent_t e0, e1;
    CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    for (int k = 0; k < 30; ++k) go();
    CK(hipDeviceSynchronize());
    const int reps = 20;
    CK(hipEventRecord(e0));
    for (int k = 0; k < reps; ++k) go();
    CK(hipEventRecord(e1));
    CK(hipDeviceSynchronize());
    float ms = 0;
    CK(hipEventElapsedTime(&ms, e0, e1));
    ms /= reps;
    unsigned long long st[4];
    CK(hipMemcpyFromSymbol(st, HIP_SYMBOL(g_bf3w_stamps), sizeof(st)));
    const double us_blk = (st[2] - st[0]) / 100.0, mhz = (double)(st[3] - st[1]) / us_blk;
    const double mfma = (double)B / 128 * 4 * (L.KC1 + L.NB2) * (L.NB2 * 24);  // 32x32x16 MFMAs issued (all waves)
    printf("ABL=%d D=%d B=%lld: %.1f us  %.3e pairs/s  block 0: %.1f us at %.0f MHz  MFMA pipe %.3f of the cycles, %.0f TFLOP/s issued\n",
           NPLDA_BF3W_ABL, D, B, ms * 1e3, B / (ms * 1e-3), us_blk, mhz,
           mfma * 32 / (cus * 4) / (us_blk * mhz), mfma * 32768 / (ms * 1e-3) / 1e12);
    {   // time spent in the fences of block 0 (shader cycles; each figure includes ~2 s_memtime round trips per fence)
        unsigned long long ss[8];
        CK(hipMemcpyFromSymbol(ss, HIP_SYMBOL(g_bf3w_steps), sizeof(ss)));
        const double tot = (double)(st[3] - st[1]);
        for (int w = 0; w < 4; ++w)
            printf("  wave %d: waiting for its loads %.3f of the kernel, at the barrier %.3f\n", w, ss[2 * w] / tot, ss[2 * w + 1] / tot);
    }
    return 0;
}

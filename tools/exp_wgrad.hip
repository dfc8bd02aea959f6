// exp_wgrad.hip — the full-M weight-gradient kernel (nplda_wgrad_fm.h) on random operands: slab sums against a
// plain fp64 reference, time per launch, phase stamps of one wave (not product code).
// build: hipcc --offload-arch=gfx950 -O3 -std=c++17 tools/exp_wgrad.hip -o tools/exp_wgrad
// run:   tools/exp_wgrad [D=150]
#ifndef NPLDA_FM_STAMPS
#define NPLDA_FM_STAMPS 0
#endif
#include <cstdio>
#include <cstdlib>
#include <cmath>
#include <vector>
#include "../neuralplda_amd/csrc/nplda_wgrad_fm.h"

using namespace nplda;

#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s at %s:%d\n", hipGetErrorString(e), __FILE__, __LINE__); exit(1); } } while (0)

__global__ void fill_rand(float* p, size_t n, unsigned seed, float scale) {
    size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    const size_t stride = (size_t)gridDim.x * blockDim.x;
    for (; i < n; i += stride) {
        unsigned long long h = (i + 1) * 0x9E3779B97F4A7C15ull + seed * 0xD1B54A32D192ED03ull;
        h ^= h >> 32; h *= 0xD6E8FEB86659FD93ull; h ^= h >> 32;
        float u = 0.f;
        for (int k = 0; k < 4; ++k) u += (float)((h >> (16 * k)) & 0xffff) / 65536.0f;
        p[i] = (u - 2.0f) * 1.7320508f * scale;
    }
}

// reference: C[m][n] = sum_k A[k][m] B[k][n] in fp64, one thread per element
__global__ void ref_atb(const float* A, long long lda, const float* B0, const float* B1, long long ldb, long long K,
                        long long nsplit, int M, int N, double* C) {
    const int n = blockIdx.x * blockDim.x + threadIdx.x, m = blockIdx.y;
    if (n >= N || m >= M) return;
    double s = 0.0;
    for (long long k = 0; k < K; ++k) {
        const float* b = k < nsplit ? B0 + k * ldb : B1 + (k - nsplit) * ldb;
        s += (double)A[k * lda + m] * (double)b[n];
    }
    C[(size_t)m * N + n] = s;
}

template <int NB>
static void launch(const WgradFmArgs& fa, hipStream_t st) {
    hipLaunchKernelGGL(wgrad_fm_kernel<NB>, dim3((unsigned)((fa.nt0 + fa.nt1) * fa.w.ksplit)), dim3(kFmWaves * 64), 0, st, fa);
}

int main(int argc, char** argv) {
    const int D = argc > 1 ? atoi(argv[1]) : 150;
    const int D0 = 512;
    const int NB = (D + 15) / 16, Mp = 16 * NB;
    const long long Bmax = 8192;
    float *du, *dz, *x, *y, *slab1, *slab2, *ext, *pq;
    double* ref;
    CK(hipMalloc(&du, 2 * Bmax * Mp * 4)); CK(hipMalloc(&dz, 2 * Bmax * Mp * 4)); CK(hipMalloc(&y, 2 * Bmax * Mp * 4));
    CK(hipMalloc(&x, 2 * Bmax * D0 * 4));
    CK(hipMalloc(&slab1, (size_t)16 * Mp * D0 * 4)); CK(hipMalloc(&slab2, (size_t)16 * Mp * Mp * 4));
    CK(hipMalloc(&ext, 16 * 4 * Mp * 4)); CK(hipMalloc(&pq, (Bmax / 16) * 2 * Mp * 4));
    CK(hipMalloc(&ref, (size_t)Mp * D0 * 8));
    hipLaunchKernelGGL(fill_rand, dim3(1024), dim3(256), 0, 0, du, (size_t)2 * Bmax * Mp, 1u, 0.1f);
    hipLaunchKernelGGL(fill_rand, dim3(1024), dim3(256), 0, 0, dz, (size_t)2 * Bmax * Mp, 2u, 0.1f);
    hipLaunchKernelGGL(fill_rand, dim3(1024), dim3(256), 0, 0, y, (size_t)2 * Bmax * Mp, 3u, 0.1f);
    hipLaunchKernelGGL(fill_rand, dim3(1024), dim3(256), 0, 0, x, (size_t)2 * Bmax * D0, 4u, 1.0f);
    hipLaunchKernelGGL(fill_rand, dim3(1024), dim3(256), 0, 0, pq, (size_t)(Bmax / 16) * 2 * Mp, 5u, 1.0f);
    CK(hipDeviceSynchronize());
    hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    const long long sizes[] = {128, 2048, 4096, 8192};
    for (long long B : sizes) {
        const long long K = 2 * B;
        WgradFmArgs fa = {};
        WgradArgs& wa = fa.w;
        fa.nt0 = (D0 + 31) / 32; fa.nt1 = (Mp + 31) / 32;
        const int tiles = fa.nt0 + fa.nt1;
        long long ks = 256 / tiles;
        if (ks > 16) ks = 16;
        if (ks > (K / 4 + kFmWaves - 1) / kFmWaves) ks = (K / 4 + kFmWaves - 1) / kFmWaves;
        wa.K = K; wa.nsplit = B; wa.ksplit = (int)ks; wa.ldz = Mp; wa.ext = ext; wa.Mp = Mp;
        wa.pq = pq; wa.nblk = (int)(B / 16);
        fa.ps_cols = (2 * Mp + tiles - 1) / tiles;
        WgradProblem& p1 = wa.p[0];
        p1.A = du; p1.lda = Mp; p1.B0 = x; p1.B1 = x + (size_t)Bmax * D0; p1.ldb = D0; p1.M = Mp; p1.N = D0;
        p1.slab = slab1; p1.Mp = Mp; p1.Np = D0; p1.extras = 1;
        WgradProblem& p2 = wa.p[1];
        p2.A = dz; p2.lda = Mp; p2.B0 = y; p2.B1 = y + (size_t)B * Mp; p2.ldb = Mp; p2.M = Mp; p2.N = Mp;
        p2.slab = slab2; p2.Mp = Mp; p2.Np = Mp; p2.extras = 4;
        auto go = [&]() {
            switch (NB) {
                case 10: launch<10>(fa, 0); break;
                case 11: launch<11>(fa, 0); break;
                case 12: launch<12>(fa, 0); break;
                default: printf("NB must be 10..12\n"); exit(1);
            }
        };
        for (int i = 0; i < 5; ++i) go();
        CK(hipDeviceSynchronize());
        const int reps = 50;
        CK(hipEventRecord(e0, 0));
        for (int i = 0; i < reps; ++i) go();
        CK(hipEventRecord(e1, 0));
        CK(hipEventSynchronize(e1));
        float ms; CK(hipEventElapsedTime(&ms, e0, e1));
        unsigned long long st[16];
        CK(hipMemcpyFromSymbol(st, HIP_SYMBOL(g_fm_stamps), sizeof(st)));
        // check both products
        double worst = 0.0;
        for (int pi = 0; pi < 2; ++pi) {
            const WgradProblem& P = wa.p[pi];
            hipLaunchKernelGGL(ref_atb, dim3((P.N + 255) / 256, Mp), dim3(256), 0, 0, P.A, P.lda, P.B0, P.B1, P.ldb, K, B, Mp, P.N, ref);
            std::vector<double> hr((size_t)Mp * P.N);
            std::vector<float> hs((size_t)ks * Mp * P.Np);
            CK(hipMemcpy(hr.data(), ref, hr.size() * 8, hipMemcpyDeviceToHost));
            CK(hipMemcpy(hs.data(), P.slab, hs.size() * 4, hipMemcpyDeviceToHost));
            double scale = 0.0;
            for (double v : hr) scale = std::fmax(scale, std::fabs(v));
            for (int m = 0; m < Mp; ++m)
                for (int n = 0; n < P.N; ++n) {
                    double s = 0.0;
                    for (int k = 0; k < ks; ++k) s += hs[((size_t)k * Mp + m) * P.Np + n];
                    worst = std::fmax(worst, std::fabs(s - hr[(size_t)m * P.N + n]) / scale);
                }
        }
        // column sums (ext rows 3 = sum du, 2 = sum dz) and pair sums (rows 0, 1)
        std::vector<float> he((size_t)ks * 4 * Mp), hdu((size_t)K * Mp), hpq((size_t)(B / 16) * 2 * Mp);
        CK(hipMemcpy(he.data(), ext, he.size() * 4, hipMemcpyDeviceToHost));
        CK(hipMemcpy(hdu.data(), du, hdu.size() * 4, hipMemcpyDeviceToHost));
        CK(hipMemcpy(hpq.data(), pq, hpq.size() * 4, hipMemcpyDeviceToHost));
        double worst_e = 0.0;
        for (int f = 0; f < Mp; ++f) {
            double s = 0.0, r = 0.0, ps = 0.0, pr = 0.0;
            for (int k = 0; k < ks; ++k) { s += he[((size_t)k * 4 + 3) * Mp + f]; ps += he[((size_t)k * 4 + 1) * Mp + f]; }
            for (long long k = 0; k < K; ++k) r += hdu[(size_t)k * Mp + f];
            for (long long b = 0; b < B / 16; ++b) pr += hpq[((size_t)b * 2 + 1) * Mp + f];
            worst_e = std::fmax(worst_e, std::fmax(std::fabs(s - r), std::fabs(ps - pr)));
        }
        auto us = [&](int i) { return (st[i] - st[0]) / 100.0; };
        printf("D=%d B=%5lld ksplit %2lld blocks %3lld: %.2f us/launch | max err / max |C| %.2e, column / pair sums abs err %.2e | "
               "stamps thread %d: loop start %.2f | loop end %.2f | round 0: sync %.2f stored %.2f | round 1: sync %.2f stored %.2f | end %.2f\n",
               D, B, ks, ks * tiles, ms * 1e3 / reps, worst, worst_e, NPLDA_FM_STAMPS, us(1), us(2), us(5), us(6), us(7), us(8), us(4));
    }
    return 0;
}

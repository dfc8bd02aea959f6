// exp_fb.hip — phase time stamps of train_fb_small_kernel (forward + loss + data gradients of the fused training step)
// at several batch sizes: where the ~30 us of one 16-pair tile go (not product code).
// build: hipcc --offload-arch=gfx950 -O3 -std=c++17 tools/exp_fb.hip -o tools/exp_fb
// run:   tools/exp_fb [D=170] [stamp thread=0 (wave = thread / 64)]
#ifndef NPLDA_FB_STAMPS
#define NPLDA_FB_STAMPS 0
#endif
#include <cstdio>
#include <cstdlib>
#include <vector>
#include "../neuralplda_amd/csrc/nplda_train_fb_small.h"

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
__global__ void fill_targets(float* t, size_t n) {
    size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) t[i] = (i % 7) == 0 ? 1.f : 0.f;
}

template <int NB>
static void launch(const TrainFbArgs& fb, long long B, hipStream_t st) {
    hipLaunchKernelGGL((train_fb_small_kernel<NB, 32, false>), dim3((unsigned)((B + 15) / 16)), dim3(256), 0, st, fb);
}

int main(int argc, char** argv) {
    const int D = argc > 1 ? atoi(argv[1]) : 170;
    const int D0 = 512;
    const NpldaLayout L = nplda_layout(D0, D, D);
    const long long Bmax = 16384;
    const long long ldz = 16 * L.NB;
    float *packed, *x, *t, *y, *dz, *du, *pq, *s, *th;
    double* partial;
    CK(hipMalloc(&packed, L.total * 4));
    CK(hipMalloc(&x, 2 * Bmax * D0 * 4));
    CK(hipMalloc(&t, Bmax * 4));
    CK(hipMalloc(&y, 2 * Bmax * ldz * 4)); CK(hipMalloc(&dz, 2 * Bmax * ldz * 4)); CK(hipMalloc(&du, 2 * Bmax * ldz * 4));
    CK(hipMalloc(&pq, (Bmax / 16) * 2 * ldz * 4)); CK(hipMalloc(&s, Bmax * 4)); CK(hipMalloc(&th, 16));
    CK(hipMalloc(&partial, (Bmax / 16) * kLossNS * 8));
    hipLaunchKernelGGL(fill_rand, dim3(1024), dim3(256), 0, 0, packed, L.total, 1u, 0.05f);
    hipLaunchKernelGGL(fill_rand, dim3(1024), dim3(256), 0, 0, x, (size_t)2 * Bmax * D0, 2u, 1.0f);
    hipLaunchKernelGGL(fill_targets, dim3((unsigned)(Bmax / 256)), dim3(256), 0, 0, t, (size_t)Bmax);
    const float thh[4] = {-0.4f, -0.2f, 0.f, 0.f};
    CK(hipMemcpy(th, thh, 16, hipMemcpyHostToDevice));
    CK(hipDeviceSynchronize());

    hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    const long long sizes[] = {128, 1024, 2048, 4096, 8192};
    for (long long B : sizes) {
        TrainFbArgs fb = {};
        fb.xa = x; fb.xb = x + (size_t)Bmax * D0; fb.n = B; fb.ldx = D0; fb.packed = packed; fb.D0 = D0; fb.KS1 = L.KS1;
        fb.oW2 = L.oW2; fb.oW2T = L.oW2T; fb.ob1 = L.ob1; fb.ob2 = L.ob2; fb.oQ = L.oQ; fb.oP = L.oP;
        fb.out_s = s; fb.out_y = y; fb.dz = dz; fb.du = du; fb.ldz = ldz; fb.pq = pq;
        fb.ls.t = t; fb.ls.K = 2; fb.ls.kind = 0; fb.ls.alpha = 15.f; fb.ls.B = B; fb.ls.partial = partial;
        fb.ls.th.p[0] = th; fb.ls.th.p[1] = th + 1; fb.ls.beta.b[0] = 99.f; fb.ls.beta.b[1] = 199.f;
        auto go = [&]() {
            switch (L.NB) {
                case 10: launch<10>(fb, B, 0); break;
                case 11: launch<11>(fb, B, 0); break;
                default: printf("D must give NB 10 or 11\n"); exit(1);
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
        unsigned long long st[32];
        CK(hipMemcpyFromSymbol(st, HIP_SYMBOL(g_fb_stamps), sizeof(st)));
        auto us = [&](int i) { return (st[i] - st[0]) / 100.0; };
        printf("B=%5lld  %.2f us/launch (back to back) | stamps of thread %d, us since entry: prefetch issued %.2f | counts %.2f | layer 1 end %.2f | "
               "y published %.2f | layer 2 end %.2f | scores %.2f | dz published %.2f | dy end %.2f | end %.2f | "
               "layer-1 clock %.0f MHz\n",
               B, ms * 1e3 / reps, NPLDA_FB_STAMPS, us(1), us(2), us(3), us(4), us(5), us(6), us(7), us(8), us(9),
               (double)(st[16 + 3] - st[16 + 2]) / ((st[3] - st[2]) / 100.0));
    }
    return 0;
}

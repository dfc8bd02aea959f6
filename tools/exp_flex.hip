// exp_flex.hip — nplda_fwd_flex_kernel against the product dispatch (small / mid / streaming kernels): results and time
// build: hipcc --offload-arch=gfx950 -O3 -std=c++17 tools/exp_flex.hip -o tools/exp_flex      run: tools/exp_flex [D=150]
#include <cstdio>
#include <cstdlib>
#include <vector>
#include <cmath>
#include "../neuralplda_amd/csrc/nplda_fwd_dispatch.h"
#include "exp_flex_kernel.h"
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

static int launch_flex(FwdArgs a, const NpldaLayout& L, hipStream_t st) {
    a.D0 = L.D0; a.KS1 = L.KS1; a.oW2 = L.oW2; a.ob1 = L.ob1; a.ob2 = L.ob2; a.oQ = L.oQ; a.oP = L.oP; a.total = L.total;
    return L.NB == 10 ? launch_fwd_flex_nb<10>(a, 256, st) : launch_fwd_flex_nb<11>(a, 256, st);
}

int main(int argc, char** argv) {
    const int D = argc > 1 ? atoi(argv[1]) : 150;
    const long long BMAX = 1 << 17;
    const int D0 = 512;
    const NpldaLayout L = nplda_layout(D0, D, D);
    float *x1, *x2, *s, *s2, *packed, *W1, *b1, *W2, *b2, *Ps, *Q;
    CK(hipMalloc(&x1, BMAX * D0 * 4)); CK(hipMalloc(&x2, BMAX * D0 * 4)); CK(hipMalloc(&s, BMAX * 4)); CK(hipMalloc(&s2, BMAX * 4));
    CK(hipMalloc(&packed, L.total * 4));
    CK(hipMalloc(&W1, D * D0 * 4)); CK(hipMalloc(&b1, D * 4)); CK(hipMalloc(&W2, D * D * 4));
    CK(hipMalloc(&b2, D * 4)); CK(hipMalloc(&Ps, D * 4)); CK(hipMalloc(&Q, D * 4));
    fill_rand<<<4096, 256>>>(x1, (size_t)BMAX * D0, 1); fill_rand<<<4096, 256>>>(x2, (size_t)BMAX * D0, 2);
    fill_rand<<<64, 256>>>(W1, (size_t)D * D0, 3); fill_rand<<<1, 256>>>(b1, D, 4);
    fill_rand<<<64, 256>>>(W2, (size_t)D * D, 5); fill_rand<<<1, 256>>>(b2, D, 6);
    fill_rand<<<1, 256>>>(Ps, D, 7); fill_rand<<<1, 256>>>(Q, D, 8);
    nplda_pack_kernel<<<(unsigned)((L.total + 255) / 256), 256>>>(W1, b1, W2, b2, Ps, Q, L, packed);
    CK(hipDeviceSynchronize());
    const double flop_alg = 2.0 * (2.0 * D0 * D + 2.0 * D * D) + 8.0 * D;
    hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    const long long sizes_all[] = {9, 2048, 4096, 4097, 6000, 8192, 10240, 12288, 16384, 16385, 20000, 20480, 24577, 32768, 40000, 49152, 65536, 100000, 131072};
    const long long sizes_abl[] = {16384, 131072};
    std::vector<long long> sizes;
    if (NPLDA_FLEX_ABL) sizes.assign(sizes_abl, sizes_abl + 2); else sizes.assign(sizes_all, sizes_all + sizeof(sizes_all) / sizeof(sizes_all[0]));
    if (NPLDA_FLEX_ABL) printf("ABLATION %d (wrong results expected)\n", NPLDA_FLEX_ABL);
    for (long long B : sizes) {
        FwdArgs a = {};
        a.xa = x1; a.xb = x2; a.n = B; a.ldx = D0; a.packed = packed; a.out_s = s;
        FwdArgs f = a; f.out_s = s2;
        CK(hipMemset(s, 0, B * 4)); CK(hipMemset(s2, 0xff, B * 4));
        if (launch_fwd<MODE_PAIR>(a, L, 0) != 0) { printf("ref launch failed\n"); return 1; }
        if (launch_flex(f, L, 0) != 0) { printf("flex launch failed\n"); return 1; }
        CK(hipDeviceSynchronize());
        std::vector<float> h1(B), h2(B);
        CK(hipMemcpy(h1.data(), s, B * 4, hipMemcpyDeviceToHost)); CK(hipMemcpy(h2.data(), s2, B * 4, hipMemcpyDeviceToHost));
        double md = 0, mx = 0; long long bad = 0;
        for (long long i = 0; i < B; ++i) {
            const double d = fabs((double)h1[i] - h2[i]);
            if (!(d <= 2e-5 + 1e-5 * fabs(h1[i]))) ++bad;
            md = fmax(md, d); mx = fmax(mx, fabs(h1[i]));
        }
        float t[2];
        for (int v = 0; v < 2; ++v) {
            const int reps = 20;
            for (int i = 0; i < 3; ++i) { if (v == 0) launch_fwd<MODE_PAIR>(a, L, 0); else launch_flex(f, L, 0); }
            CK(hipEventRecord(e0, 0));
            for (int i = 0; i < reps; ++i) { if (v == 0) launch_fwd<MODE_PAIR>(a, L, 0); else launch_flex(f, L, 0); }
            CK(hipEventRecord(e1, 0)); CK(hipEventSynchronize(e1));
            CK(hipEventElapsedTime(&t[v], e0, e1)); t[v] = t[v] / reps * 1000.f;
        }
        printf("D=%d B=%7lld  product %7.1f us (frac %.3f)  flex %7.1f us (frac %.3f)  max|d| %.2e (max|s| %.1f)  out of tolerance: %lld\n", D, B,
               t[0], B * flop_alg / (t[0] * 1e-6) / 157.3e12, t[1], B * flop_alg / (t[1] * 1e-6) / 157.3e12, md, mx, bad);
    }
    return 0;
}

// exp_mid.hip — A/B of the mid-regime pair-scoring kernel (nplda_fwd_mid.h) against the product dispatch, over batch sizes.
// build: hipcc --offload-arch=gfx950 -O3 -std=c++17 tools/exp_mid.hip -o tools/exp_mid
// run:   tools/exp_mid [D=150] [rounds=2]
#define NPLDA_MID_STAMPS 1
#include <cstdio>
#include <cstdlib>
#include <vector>
#include <cmath>
#include "../neuralplda_amd/csrc/nplda_fwd_dispatch.h"

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

int main(int argc, char** argv) {
    const int D = argc > 1 ? atoi(argv[1]) : 150;
    const int rounds = argc > 2 ? atoi(argv[2]) : 2;
    const int D0 = 512;
    const long long BMAX = 1 << 18;
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
    if (argc > 3) {  // phase stamps of one block: tools/exp_mid D rounds B
        const long long B = atoll(argv[3]);
        const bool embed = argc > 4;  // tools/exp_mid D rounds N embed [noq]: the EMBED form on N rows (N / 2 pairs' worth of work)
        FwdArgs m = {};
        m.xa = x1; m.xb = x2; m.n = B; m.ldx = D0; m.packed = packed; m.out_s = s2;
        m.D0 = L.D0; m.KS1 = L.KS1; m.oW2 = L.oW2; m.ob1 = L.ob1; m.ob2 = L.ob2; m.oQ = L.oQ; m.oP = L.oP; m.total = L.total;
        float* zout = nullptr;
        if (embed) {
            CK(hipMalloc(&zout, (size_t)B * 16 * L.NB * 4));
            m.out_z = zout; m.ldz = 16 * L.NB; m.out_q = argc > 5 ? nullptr : s2; m.out_s = nullptr;
        }
        hipEvent_t f0, f1; CK(hipEventCreate(&f0)); CK(hipEventCreate(&f1));
        for (int rep = 0; rep < 4; ++rep) {
            for (int k = 0; k < 3; ++k) embed ? launch_fwd_mid<true>(m, L, 0) : launch_fwd_mid<false>(m, L, 0);
            CK(hipEventRecord(f0, 0));
            for (int k = 0; k < 10; ++k) embed ? launch_fwd_mid<true>(m, L, 0) : launch_fwd_mid<false>(m, L, 0);
            CK(hipEventRecord(f1, 0));
            CK(hipDeviceSynchronize());
            float tms = 0; CK(hipEventElapsedTime(&tms, f0, f1));
            printf("%s: %.1f us per launch\n", embed ? "EMBED" : "PAIR", tms * 100.0);
            unsigned long long st[32];
            CK(hipMemcpyFromSymbol(st, HIP_SYMBOL(g_mid_stamps), sizeof(st)));
            const char* names[] = {"entry", "prologue loads issued", "step 0 done", "K loop end", "exported", "barrier 1", "reduced + ss",
                                   "y published", "barrier 3", "layer 2 end", "end"};
            printf("B=%lld last group of the middle block (us since group entry / cycles):\n", B);
            for (int i = 1; i <= 10; ++i)
                printf("  %-22s %7.2f us  %8llu cyc  (+%6llu)\n", names[i], (st[i] - st[0]) / 100.0, st[16 + i] - st[16], st[16 + i] - st[16 + i - 1]);
            printf("  addr A %llu, addr next %llu (cycles since entry); z stores issued +%llu after layer 2\n", st[16 + 11] - st[16], st[16 + 12] - st[16], st[16 + 13] - st[16 + 9]);
            printf("  clock over the group: %.0f MHz\n", (double)(st[26] - st[16]) / ((st[10] - st[0]) / 100.0));
        }
        return 0;
    }
    const double flop_alg = 2.0 * (2.0 * D0 * D + 2.0 * D * D) + 8.0 * D;
    hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    const long long sizes[] = {16, 100, 1000, 2048, 4096, 4097, 6000, 8192, 12288, 16384, 16385, 20000, 20480, 24577, 32768, 40000, 49152,
                               65536, 100000, 131072, 262144};
    std::vector<float> h1(BMAX), h2(BMAX);
    for (int rd = 0; rd < rounds; ++rd) {
        for (long long B : sizes) {
            FwdArgs a = {};
            a.xa = x1; a.xb = x2; a.n = B; a.ldx = D0; a.packed = packed; a.out_s = s;
            FwdArgs m = a; m.out_s = s2;
            m.D0 = L.D0; m.KS1 = L.KS1; m.oW2 = L.oW2; m.ob1 = L.ob1; m.ob2 = L.ob2; m.oQ = L.oQ; m.oP = L.oP; m.total = L.total;
            float t[3] = {0, 0, 0};
            for (int v = 0; v < 2; ++v) {
                auto go = [&]() { return v == 0 ? launch_fwd_old<MODE_PAIR>(a, L, 0) : launch_fwd_mid<false>(m, L, 0); };
                CK(hipMemset(v == 0 ? s : s2, 0xff, B * 4));
                int rc = go(); rc |= go();
                if (rc) { printf("launch rc %d\n", rc); return 1; }
                CK(hipDeviceSynchronize());
                const int reps = 20;
                CK(hipEventRecord(e0, 0));
                for (int i = 0; i < reps; ++i) go();
                CK(hipEventRecord(e1, 0));
                CK(hipEventSynchronize(e1));
                CK(hipEventElapsedTime(&t[v], e0, e1)); t[v] /= reps;
            }
            CK(hipMemcpy(h1.data(), s, B * 4, hipMemcpyDeviceToHost));
            CK(hipMemcpy(h2.data(), s2, B * 4, hipMemcpyDeviceToHost));
            double md = 0, mx = 0; long long bad = 0;
            for (long long i = 0; i < B; ++i) {
                const double d = fabs((double)h1[i] - h2[i]);
                if (!(d <= 2e-5 + 1e-5 * fabs(h1[i]))) ++bad;
                md = fmax(md, d); mx = fmax(mx, fabs(h1[i]));
            }
            printf("D=%d B=%7lld  old %8.1f us frac %.3f | mid %8.1f us frac %.3f | mid1 %8.1f us frac %.3f | max|d| %.2e (max|s| %.1f) bad %lld\n", D, B,
                   t[0] * 1e3, B * flop_alg / (t[0] * 1e-3) / 1e12 / 157.3, t[1] * 1e3, B * flop_alg / (t[1] * 1e-3) / 1e12 / 157.3,
                   t[2] * 1e3, B * flop_alg / (t[2] * 1e-3) / 1e12 / 157.3, md, mx, bad);
            fflush(stdout);
        }
    }
    return 0;
}

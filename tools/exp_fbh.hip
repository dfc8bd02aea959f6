// exp_fbh.hip — train_fb_half_kernel (8-pair tiles, two blocks per CU) against train_fb_small_kernel (16-pair tiles): launch
// time back to back under every skew mode, the largest difference of what the two leave (y, dz, du, s, pair sums, loss sums),
// and phase stamps of one block (not product code).
// build: hipcc --offload-arch=gfx950 -O3 -std=c++17 tools/exp_fbh.hip -o tools/exp_fbh      run: tools/exp_fbh [D=150] [B=4096]
#ifndef NPLDA_FBH_STAMPS
#define NPLDA_FBH_STAMPS 300
#endif
#ifndef NPLDA_FBD_STAMPS
#define NPLDA_FBD_STAMPS 100
#endif
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <vector>
#ifndef NPLDA_DUO_JOINT
#define NPLDA_DUO_SPLIT 1
#endif
#include "exp_fb_duo_kernel.h"

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

struct Bufs { float *y, *dz, *du, *pq, *s; double* partial; };

static double maxdiff(const float* a, const float* b, size_t n, double* ref = nullptr) {
    std::vector<float> ha(n), hb(n);
    CK(hipMemcpy(ha.data(), a, n * 4, hipMemcpyDeviceToHost));
    CK(hipMemcpy(hb.data(), b, n * 4, hipMemcpyDeviceToHost));
    double m = 0, r = 0;
    for (size_t i = 0; i < n; ++i) { m = std::fmax(m, std::fabs((double)ha[i] - hb[i])); r = std::fmax(r, std::fabs((double)hb[i])); }
    if (ref) *ref = r;
    return m;
}

int main(int argc, char** argv) {
    const int D = argc > 1 ? atoi(argv[1]) : 150;
    const long long B = argc > 2 ? atoll(argv[2]) : 4096;
    const int D0 = 512;
    const NpldaLayout L = nplda_layout(D0, D, D);
    const long long ldz = 16 * L.NB;
    float *packed, *x, *t, *th;
    CK(hipMalloc(&packed, L.total * 4));
    CK(hipMalloc(&x, 2 * B * D0 * 4));
    CK(hipMalloc(&t, (B + 4) * 4));
    CK(hipMalloc(&th, 16));
    Bufs bo, bn;
    for (Bufs* b : {&bo, &bn}) {
        CK(hipMalloc(&b->y, 2 * B * ldz * 4)); CK(hipMalloc(&b->dz, 2 * B * ldz * 4)); CK(hipMalloc(&b->du, 2 * B * ldz * 4));
        CK(hipMalloc(&b->pq, ((B + 7) / 8) * 2 * ldz * 4)); CK(hipMalloc(&b->s, B * 4));
        CK(hipMalloc(&b->partial, ((B + 7) / 8) * kLossNS * 8));
        CK(hipMemset(b->y, 0, 2 * B * ldz * 4)); CK(hipMemset(b->dz, 0, 2 * B * ldz * 4)); CK(hipMemset(b->du, 0, 2 * B * ldz * 4));
    }
    hipLaunchKernelGGL(fill_rand, dim3(1024), dim3(256), 0, 0, packed, L.total, 1u, 0.05f);
    hipLaunchKernelGGL(fill_rand, dim3(1024), dim3(256), 0, 0, x, (size_t)2 * B * D0, 2u, 1.0f);
    hipLaunchKernelGGL(fill_targets, dim3((unsigned)((B + 255) / 256)), dim3(256), 0, 0, t, (size_t)B);
    // zero the image's padding as the pack kernel would (columns >= D of both layers): b1, b2, Q, P padding stay random-small, harmless
    const float thh[4] = {-0.4f, -0.2f, 0.f, 0.f};
    CK(hipMemcpy(th, thh, 16, hipMemcpyHostToDevice));
    CK(hipDeviceSynchronize());

    auto args = [&](const Bufs& b) {
        TrainFbArgs fb = {};
        fb.xa = x; fb.xb = x + (size_t)B * D0; fb.n = B; fb.ldx = D0; fb.packed = packed; fb.D0 = D0; fb.KS1 = L.KS1;
        fb.oW2 = L.oW2; fb.oW2T = L.oW2T; fb.ob1 = L.ob1; fb.ob2 = L.ob2; fb.oQ = L.oQ; fb.oP = L.oP;
        fb.out_s = b.s; fb.out_y = b.y; fb.dz = b.dz; fb.du = b.du; fb.ldz = ldz; fb.pq = b.pq;
        fb.ls.t = t; fb.ls.K = 2; fb.ls.kind = 0; fb.ls.alpha = 15.f; fb.ls.B = B; fb.ls.partial = b.partial;
        fb.ls.th.p[0] = th; fb.ls.th.p[1] = th + 1; fb.ls.beta.b[0] = 99.f; fb.ls.beta.b[1] = 199.f;
        return fb;
    };
    const TrainFbArgs fo = args(bo), fn = args(bn);
    auto go_old = [&]() {
        if (L.NB == 10) hipLaunchKernelGGL((train_fb_small_kernel<10, 32, false>), dim3((unsigned)((B + 15) / 16)), dim3(256), 0, 0, fo);
        else hipLaunchKernelGGL((train_fb_small_kernel<11, 32, false>), dim3((unsigned)((B + 15) / 16)), dim3(256), 0, 0, fo);
    };
    auto go_new = [&](HalfSkew sk) {
        if (L.NB == 10) hipLaunchKernelGGL((train_fb_half_kernel<10, false>), dim3((unsigned)((B + 7) / 8)), dim3(256), 0, 0, fn, sk);
        else hipLaunchKernelGGL((train_fb_half_kernel<11, false>), dim3((unsigned)((B + 7) / 8)), dim3(256), 0, 0, fn, sk);
    };
    Bufs bd;
    {
        Bufs* b = &bd;
        CK(hipMalloc(&b->y, 2 * B * ldz * 4)); CK(hipMalloc(&b->dz, 2 * B * ldz * 4)); CK(hipMalloc(&b->du, 2 * B * ldz * 4));
        CK(hipMalloc(&b->pq, ((B + 7) / 8 + 2) * 2 * ldz * 4)); CK(hipMalloc(&b->s, B * 4));
        CK(hipMalloc(&b->partial, ((B + 7) / 8 + 2) * kLossNS * 8));
        CK(hipMemset(b->y, 0, 2 * B * ldz * 4)); CK(hipMemset(b->dz, 0, 2 * B * ldz * 4)); CK(hipMemset(b->du, 0, 2 * B * ldz * 4));
    }
    const TrainFbArgs fd = args(bd);
    auto go_duo = [&]() {
        if (L.NB == 10) hipLaunchKernelGGL((train_fb_duo_kernel<10, false>), dim3((unsigned)((B + 15) / 16)), dim3(512), 0, 0, fd);
        else hipLaunchKernelGGL((train_fb_duo_kernel<11, false>), dim3((unsigned)((B + 15) / 16)), dim3(512), 0, 0, fd);
    };
    go_old();
    go_new(HalfSkew{1, 0});
    go_duo();
    CK(hipDeviceSynchronize());
    {
        double r2;
        printf("D=%d B=%lld  max|duo - old|: y %.3g", D, B, maxdiff(bd.y, bo.y, 2 * B * ldz, &r2)); printf(" (max %.3g)", r2);
        printf("  dz %.3g", maxdiff(bd.dz, bo.dz, 2 * B * ldz, &r2)); printf(" (max %.3g)", r2);
        printf("  du %.3g", maxdiff(bd.du, bo.du, 2 * B * ldz, &r2)); printf(" (max %.3g)", r2);
        printf("  s %.3g", maxdiff(bd.s, bo.s, B, &r2)); printf(" (max %.3g)\n", r2);
        const size_t no = ((B + 15) / 16), nn = 2 * ((B + 15) / 16);
        std::vector<float> po(no * 2 * ldz), pn(nn * 2 * ldz);
        std::vector<double> lo(no * kLossNS), ln(nn * kLossNS);
        CK(hipMemcpy(po.data(), bo.pq, po.size() * 4, hipMemcpyDeviceToHost));
        CK(hipMemcpy(pn.data(), bd.pq, pn.size() * 4, hipMemcpyDeviceToHost));
        CK(hipMemcpy(lo.data(), bo.partial, lo.size() * 8, hipMemcpyDeviceToHost));
        CK(hipMemcpy(ln.data(), bd.partial, ln.size() * 8, hipMemcpyDeviceToHost));
        double md = 0, mr = 0;
        for (int c = 0; c < 2 * ldz; ++c) {
            if ((c % ldz) >= D) continue;
            double so = 0, sn = 0;
            for (size_t b = 0; b < no; ++b) so += po[b * 2 * ldz + c];
            for (size_t b = 0; b < nn; ++b) sn += pn[b * 2 * ldz + c];
            md = std::fmax(md, std::fabs(so - sn)); mr = std::fmax(mr, std::fabs(so));
        }
        printf("  duo pair sums: max diff %.3g of %.3g;", md, mr);
        md = 0; mr = 0;
        for (int c = 0; c < kLossNS; ++c) {
            double so = 0, sn = 0;
            for (size_t b = 0; b < no; ++b) so += lo[b * kLossNS + c];
            for (size_t b = 0; b < nn; ++b) sn += ln[b * kLossNS + c];
            md = std::fmax(md, std::fabs(so - sn)); mr = std::fmax(mr, std::fabs(so));
        }
        printf("  loss sums: max diff %.3g of %.3g\n", md, mr);
    }
    double r;
    printf("D=%d B=%lld  max|new - old|: y %.3g", D, B, maxdiff(bn.y, bo.y, 2 * B * ldz, &r)); printf(" (max %.3g)", r);
    printf("  dz %.3g", maxdiff(bn.dz, bo.dz, 2 * B * ldz, &r)); printf(" (max %.3g)", r);
    printf("  du %.3g", maxdiff(bn.du, bo.du, 2 * B * ldz, &r)); printf(" (max %.3g)", r);
    printf("  s %.3g", maxdiff(bn.s, bo.s, B, &r)); printf(" (max %.3g)\n", r);
    {   // pair sums and loss sums: totals over the blocks
        const size_t no = ((B + 15) / 16), nn = ((B + 7) / 8);
        std::vector<float> po(no * 2 * ldz), pn(nn * 2 * ldz);
        std::vector<double> lo(no * kLossNS), ln(nn * kLossNS);
        CK(hipMemcpy(po.data(), bo.pq, po.size() * 4, hipMemcpyDeviceToHost));
        CK(hipMemcpy(pn.data(), bn.pq, pn.size() * 4, hipMemcpyDeviceToHost));
        CK(hipMemcpy(lo.data(), bo.partial, lo.size() * 8, hipMemcpyDeviceToHost));
        CK(hipMemcpy(ln.data(), bn.partial, ln.size() * 8, hipMemcpyDeviceToHost));
        double md = 0, mr = 0;
        for (int c = 0; c < 2 * ldz; ++c) {
            if ((c % ldz) >= D) continue;
            double so = 0, sn = 0;
            for (size_t b = 0; b < no; ++b) so += po[b * 2 * ldz + c];
            for (size_t b = 0; b < nn; ++b) sn += pn[b * 2 * ldz + c];
            md = std::fmax(md, std::fabs(so - sn)); mr = std::fmax(mr, std::fabs(so));
        }
        printf("  pair sums (dQ / dP columns): max diff %.3g of %.3g;", md, mr);
        md = 0; mr = 0;
        for (int c = 0; c < kLossNS; ++c) {
            double so = 0, sn = 0;
            for (size_t b = 0; b < no; ++b) so += lo[b * kLossNS + c];
            for (size_t b = 0; b < nn; ++b) sn += ln[b * kLossNS + c];
            md = std::fmax(md, std::fabs(so - sn)); mr = std::fmax(mr, std::fabs(so));
        }
        printf("  loss sums: max diff %.3g of %.3g\n", md, mr);
    }
    hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    auto timeit = [&](auto&& fn_) {
        for (int i = 0; i < 20; ++i) fn_();
        CK(hipDeviceSynchronize());
        const int reps = 200;
        float best = 1e9f;
        for (int rr = 0; rr < 3; ++rr) {
            CK(hipEventRecord(e0, 0));
            for (int i = 0; i < reps; ++i) fn_();
            CK(hipEventRecord(e1, 0));
            CK(hipEventSynchronize(e1));
            float ms; CK(hipEventElapsedTime(&ms, e0, e1));
            best = std::fmin(best, ms * 1e3f / reps);
        }
        return best;
    };
    printf("16-pair kernel: %.2f us / launch\n", timeit(go_old));
    {
        const float us = timeit(go_duo);
        unsigned long long st[64];
        CK(hipMemcpyFromSymbol(st, HIP_SYMBOL(g_fbd_stamps), sizeof(st)));
        const unsigned long long t00 = st[0] < st[32] ? st[0] : st[32];
        printf("16-pair tile by eight waves (duo): %.2f us / launch\n", us);
        for (int o = 0; o <= 32; o += 32) {
            auto u = [&](int i) { return (double)(st[o + i] - t00) / 100.0; };
            printf("      block %3d wave %d: entry %.2f | loads issued %.2f | L1 end %.2f | exchange %.2f | y %.2f | L2 end %.2f | scores %.2f | dz %.2f | "
                   "dy end %.2f | end %.2f us\n", NPLDA_FBD_STAMPS, o ? 4 : 0, u(0), o ? u(0) : u(1), u(2), u(3), u(4), u(5), u(6), u(7), u(8), u(9));
        }
        printf("16-pair kernel again: %.2f us / launch\n", timeit(go_old));
    }
    const HalfSkew modes[] = {{0, 0}};
    for (const HalfSkew& sk : modes) {
        const float us = timeit([&]() { go_new(sk); });
        unsigned long long st[64];
        CK(hipMemcpyFromSymbol(st, HIP_SYMBOL(g_fbh_stamps), sizeof(st)));
        const unsigned long long t00 = st[0] < st[32] ? st[0] : st[32];
        printf("half tiles, skew %d,%-3d: %.2f us / launch\n", sk.mode, sk.arg, us);
        for (int o = 32; o >= 0; o -= 32) {
            auto u = [&](int i) { return (double)(st[o + i] - t00) / 100.0; };
            printf("      block %3d: entry %.2f | loads issued %.2f | L1 end %.2f | exchange %.2f | y %.2f | L2 end %.2f | scores %.2f | dz %.2f | dy end %.2f | "
                   "end %.2f us | L1 clock %.0f MHz\n", NPLDA_FBH_STAMPS - (o ? 256 : 0), u(0), u(1), u(2), u(3), u(4), u(5), u(6), u(7), u(8), u(9),
                   (double)(st[o + 16 + 2] - st[o + 16 + 1]) / ((st[o + 2] - st[o + 1]) / 100.0));
        }
    }
    return 0;
}

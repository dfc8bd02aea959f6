// exp_fwd.hip — within-process interleaved A/B of forward-kernel variants (not product code).
// build: hipcc --offload-arch=gfx950 -O3 -std=c++17 tools/exp_fwd.hip -o tools/exp_fwd
// run:   tools/exp_fwd [log2_pairs=20] [rounds=3]
#define NPLDA_SMALL_STAMPS 1
#include <cstdio>
#include <cstdlib>
#include <vector>
#include <cmath>
#include "../neuralplda_amd/csrc/nplda_fwd_small.h"
#include "../neuralplda_amd/csrc/nplda_fwd_v2.h"
#include "../neuralplda_amd/csrc/nplda_fwd_v3.h"
#include "../neuralplda_amd/csrc/nplda_fwd_v5.h"
#include "../neuralplda_amd/csrc/nplda_fwd_v6.h"
#include "../neuralplda_amd/csrc/nplda_fwd_bf16x3.h"

using namespace nplda;

#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s at %s:%d\n", hipGetErrorString(e), __FILE__, __LINE__); exit(1); } } while (0)

__global__ void fill_rand(float* p, size_t n, unsigned seed) {
    size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    const size_t stride = (size_t)gridDim.x * blockDim.x;
    for (; i < n; i += stride) {
        unsigned long long h = (i + 1) * 0x9E3779B97F4A7C15ull + seed * 0xD1B54A32D192ED03ull;
        h ^= h >> 32; h *= 0xD6E8FEB86659FD93ull; h ^= h >> 32; h *= 0xD6E8FEB86659FD93ull; h ^= h >> 32;
        // approx normal: sum of 4 uniforms
        float u = 0.f;
        for (int k = 0; k < 4; ++k) u += (float)((h >> (16 * k)) & 0xffff) / 65536.0f;
        p[i] = (u - 2.0f) * 1.7320508f;
    }
}

// FETCH_SIZE calibration: exactly the x access pattern of the forward kernels (one float4 per lane per side per
// k16-step: 16 rows x 64 B per wave instruction), nothing else.  Known traffic = 2 * B * D0 * 4 bytes.
template <bool NT>
__global__ __launch_bounds__(512) void xload_only(const float* xa, const float* xb, long long n, int D0, float* out) {
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, j = lane & 15, g = lane >> 4;
    const long long row = ((long long)blockIdx.x * 8 + wave) * 16 + j;
    if (row >= n) return;
    const float* pa = xa + row * D0 + 4 * g;
    const float* pb = xb + row * D0 + 4 * g;
    f32x4 acc = {0.f, 0.f, 0.f, 0.f};
    for (int ks = 0; ks < D0 / 16; ++ks) {
        acc += load_x4<NT>(pa + 16 * ks, true);
        acc += load_x4<NT>(pb + 16 * ks, true);
    }
    const float v = acc[0] + acc[1] + acc[2] + acc[3];
    if (v == 123.456f) out[row] = v;  // keep the loads alive without writing
}
template <bool NT>
void launch_x(const FwdArgs& a, long long B, hipStream_t st) {
    hipLaunchKernelGGL((xload_only<NT>), dim3((unsigned)((B + 127) / 128)), dim3(512), 0, st, a.xa, a.xb, B, a.D0, a.out_s);
}

static Bf3Args g_b3;
template <int NB, int WAVES, int KPB, bool EARLY = false>
void launch_b3(const FwdArgs& a, long long B, hipStream_t st) {
    const long long per_block = 16 * WAVES;
    dim3 grid((unsigned)((B + per_block - 1) / per_block)), block(WAVES * 64);
    Bf3Args b = g_b3;
    b.out_s = a.out_s;
    hipLaunchKernelGGL((nplda_fwd_bf16x3_kernel<NB, MODE_PAIR, WAVES, KPB, EARLY>), grid, block, 0, st, b);
}

struct Variant { const char* name; void (*launch)(const FwdArgs&, long long, hipStream_t); };

template <int NB, int WAVES, bool NT, int KPB>
void launch_v(const FwdArgs& a, long long B, hipStream_t st) {
    const long long per_block = 16 * WAVES;
    dim3 grid((unsigned)((B + per_block - 1) / per_block)), block(WAVES * 64);
    hipLaunchKernelGGL((nplda_fwd_kernel<NB, MODE_PAIR, WAVES, NT, KPB>), grid, block, 0, st, a);
}

template <int NB, int WAVES, bool NT, int KPB, int ABL>
void launch_a(const FwdArgs& a, long long B, hipStream_t st) {
    const long long per_block = 16 * WAVES;
    dim3 grid((unsigned)((B + per_block - 1) / per_block)), block(WAVES * 64);
    hipLaunchKernelGGL((nplda_fwd_kernel<NB, MODE_PAIR, WAVES, NT, KPB, ABL>), grid, block, 0, st, a);
}

template <int NB, int WAVES, bool NT, int KPB, int G = 4>
void launch_2(const FwdArgs& a, long long B, hipStream_t st) {
    const long long per_block = 16 * WAVES;
    dim3 grid((unsigned)((B + per_block - 1) / per_block)), block(WAVES * 64);
    hipLaunchKernelGGL((nplda_fwd_v2_kernel<NB, MODE_PAIR, WAVES, NT, KPB, G>), grid, block, 0, st, a);
}

template <int NB, int WAVES, bool NT, int KPB, int BPC, int G = 4>
void launch_3(const FwdArgs& a, long long B, hipStream_t st) {
    const long long per_block = 16 * WAVES;
    const int ntiles = (int)((B + per_block - 1) / per_block);
    int grid = 256 * BPC;
    if (grid > ntiles) grid = ntiles;
    hipLaunchKernelGGL((nplda_fwd_v3_kernel<NB, MODE_PAIR, WAVES, NT, KPB, G>), dim3(grid), dim3(WAVES * 64), 0, st, a, ntiles);
}

template <int NB, int WAVES, int KPB, int G = 4, int H = 2>
void launch_5(const FwdArgs& a, long long B, hipStream_t st) {
    const long long per_block = 16 * WAVES;
    const int ntiles = (int)((B + per_block - 1) / per_block);
    int grid = 256;
    if (grid > ntiles) grid = ntiles;
    hipLaunchKernelGGL((nplda_fwd_v5_kernel<NB, WAVES, KPB, G, H>), dim3(grid), dim3(WAVES * 64), 0, st, a, ntiles);
}

template <int NB, int TF, int WAVES, int KPB, int G1, int G2, int ABL = 0, int NCH = 2>
void launch_6(const FwdArgs& a, long long B, hipStream_t st) {
    const long long per_block = 16 * WAVES;
    const int ntiles = (int)((B + per_block - 1) / per_block);
    int grid = 256;
    if (grid > ntiles) grid = ntiles;
    hipLaunchKernelGGL((nplda_fwd_v6_kernel<NB, TF, WAVES, KPB, G1, G2, 0, ABL, NCH>), dim3(grid), dim3(WAVES * 64), 0, st, a, ntiles);
}

int main(int argc, char** argv) {
    const int lg = argc > 1 ? atoi(argv[1]) : 20;
    const int rounds = argc > 2 ? atoi(argv[2]) : 3;
    const int D = argc > 3 ? atoi(argv[3]) : 150;
    const long long B = 1ll << lg;
    const int D0 = 512;
    const NpldaLayout L = nplda_layout(D0, D, D);
    float *x1, *x2, *s, *packed, *W1, *b1, *W2, *b2, *Ps, *Q;
    CK(hipMalloc(&x1, B * D0 * 4)); CK(hipMalloc(&x2, B * D0 * 4)); CK(hipMalloc(&s, B * 4));
    CK(hipMalloc(&packed, L.total * 4));
    CK(hipMalloc(&W1, D * D0 * 4)); CK(hipMalloc(&b1, D * 4)); CK(hipMalloc(&W2, D * D * 4));
    CK(hipMalloc(&b2, D * 4)); CK(hipMalloc(&Ps, D * 4)); CK(hipMalloc(&Q, D * 4));
    fill_rand<<<4096, 256>>>(x1, (size_t)B * D0, 1); fill_rand<<<4096, 256>>>(x2, (size_t)B * D0, 2);
    fill_rand<<<64, 256>>>(W1, (size_t)D * D0, 3); fill_rand<<<1, 256>>>(b1, D, 4);
    fill_rand<<<64, 256>>>(W2, (size_t)D * D, 5); fill_rand<<<1, 256>>>(b2, D, 6);
    fill_rand<<<1, 256>>>(Ps, D, 7); fill_rand<<<1, 256>>>(Q, D, 8);
    nplda_pack_kernel<<<(unsigned)((L.total + 255) / 256), 256>>>(W1, b1, W2, b2, Ps, Q, L, packed);
    CK(hipDeviceSynchronize());

    FwdArgs a = {};
    a.xa = x1; a.xb = x2; a.n = B; a.ldx = D0; a.packed = packed; a.D0 = D0; a.KS1 = L.KS1;
    a.oW2 = L.oW2; a.ob1 = L.ob1; a.ob2 = L.ob2; a.oQ = L.oQ; a.oP = L.oP; a.total = L.total; a.out_s = s;

    // bf16x3 image
    const Bf3Layout L3 = bf3_layout(D0, D, D);
    float* img3; CK(hipMalloc(&img3, L3.total * 4));
    {
        const size_t nthreads = L3.ob1 / 4 + (L3.total - L3.ob1);
        nplda_pack_bf16x3_kernel<<<(unsigned)((nthreads + 255) / 256), 256>>>(W1, b1, W2, b2, Ps, Q, L3, img3);
        CK(hipDeviceSynchronize());
    }
    g_b3 = Bf3Args{};
    g_b3.xa = x1; g_b3.xb = x2; g_b3.n = B; g_b3.ldx = D0; g_b3.img = img3; g_b3.D0 = D0; g_b3.KC1 = L3.KC1;
    g_b3.oW2 = L3.oW2; g_b3.ob1 = L3.ob1; g_b3.ob2 = L3.ob2; g_b3.oQ = L3.oQ; g_b3.oP = L3.oP;
    std::vector<float> sref(1 << 16);
    std::vector<Variant> vs;
    if (L.NB == 10) {
        vs = { {"v3 w8 kpb2", launch_3<10, 8, false, 2, 1>}, {"v5 w8 kpb2 h1", launch_5<10, 8, 2, 4, 1>},
               {"v6 g5/3 2 chains", launch_6<10, 6, 8, 2, 5, 3>}, {"v6 g5/3 4 chains", launch_6<10, 6, 8, 2, 5, 3, 0, 4>},
               {"v6 g5/3 1 chain", launch_6<10, 6, 8, 2, 5, 3, 0, 1>}, {"v6 ABL1 g5/3", launch_6<10, 6, 8, 2, 5, 3, 1>} };
    } else if (L.NB == 11) {
        vs = { {"v5 kpb2 g4", launch_5<11, 8, 2, 4, 1>}, {"v5 kpb2 g3", launch_5<11, 8, 2, 3, 1>},
               {"v5 kpb2 g5", launch_5<11, 8, 2, 5, 1>}, {"v5 kpb2 g6", launch_5<11, 8, 2, 6, 1>},
               {"v5 kpb4 g4", launch_5<11, 8, 4, 4, 1>}, {"v6 4x4 g4/4", launch_6<11, 10, 8, 2, 4, 4>} };
    } else if (L.NB == 8) {
        vs = { {"v3 w8 kpb2 nb8", launch_3<8, 8, false, 2, 1>}, {"v5 w8 kpb2 nb8", launch_5<8, 8, 2>} };
    } else {
        vs = { {"v2 w8 kpb2", launch_2<12, 8, false, 2>}, {"v5 w8 kpb2", launch_5<12, 8, 2>},
               {"v5 w8 kpb4 h1", launch_5<12, 8, 4, 4, 1>}, {"v5 w8 kpb2 h1", launch_5<12, 8, 2, 4, 1>} };
    }
    if (argc > 4) {  // phase stamps of the small-batch kernel (training mode) at 2^lg pairs: tools/exp_fwd 12 1 150 stamps
        float *y, *z, *rn;
        CK(hipMalloc(&y, 2 * B * 16 * L.NB * 4)); CK(hipMalloc(&z, 2 * B * 16 * L.NB * 4)); CK(hipMalloc(&rn, 2 * B * 4));
        FwdArgs t = a;
        t.out_y = y; t.out_z = z; t.out_rn = rn; t.ldz = 16 * L.NB;
        for (int rep = 0; rep < 4; ++rep) {
            if (L.NB == 10) hipLaunchKernelGGL((nplda_fwd_small_kernel<10, MODE_TRAIN, 32>), dim3((unsigned)(B / 16)), dim3(256), 0, 0, t);
            else hipLaunchKernelGGL((nplda_fwd_small_kernel<11, MODE_TRAIN, 32>), dim3((unsigned)(B / 16)), dim3(256), 0, 0, t);
            CK(hipDeviceSynchronize());
            unsigned long long st[16];
            CK(hipMemcpyFromSymbol(st, HIP_SYMBOL(g_small_stamps), sizeof(st)));
            printf("small-kernel stamps (us since kernel entry): loads issued %.2f | first 8 steps %.2f | K loop end %.2f | y published %.2f | barrier %.2f | layer 2 end %.2f | end %.2f\n",
                   (st[1] - st[0]) / 100.0, (st[2] - st[0]) / 100.0, (st[3] - st[0]) / 100.0, (st[4] - st[0]) / 100.0,
                   (st[5] - st[0]) / 100.0, (st[6] - st[0]) / 100.0, (st[7] - st[0]) / 100.0);
            printf("   shader clock over steps 8..KS1: %.0f MHz (%llu cycles / %.2f us)\n", (double)(st[11] - st[10]) / ((st[3] - st[2]) / 100.0),
                   st[11] - st[10], (st[3] - st[2]) / 100.0);
            printf("   layer 1 (stamp 1 -> 3): %.2f us, %llu shader cycles = %.0f MHz; layer 2 (5 -> 6): %.2f us, %llu cycles\n",
                   (st[3] - st[1]) / 100.0, st[11] - st[9], (double)(st[11] - st[9]) / ((st[3] - st[1]) / 100.0),
                   (st[6] - st[5]) / 100.0, st[14] - st[13]);
        }
        return 0;
    }
    const double flop_alg = 2.0 * (2.0 * D0 * D + 2.0 * D * D) + 8.0 * D;
    hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    std::vector<float> hs(4096);
    for (int r = 0; r < rounds; ++r) {
        for (auto& v : vs) {
            v.launch(a, B, 0); v.launch(a, B, 0);
            CK(hipDeviceSynchronize());
            const int reps = 8;
            CK(hipEventRecord(e0, 0));
            for (int i = 0; i < reps; ++i) v.launch(a, B, 0);
            CK(hipEventRecord(e1, 0));
            CK(hipEventSynchronize(e1));
            float ms; CK(hipEventElapsedTime(&ms, e0, e1)); ms /= reps;
            CK(hipMemcpy(hs.data(), s, 4096 * 4, hipMemcpyDeviceToHost));
            double cs = 0; for (float f : hs) cs += f;
            if (&v == &vs[0]) sref.assign(hs.begin(), hs.end());
            double md = 0, mx = 0; for (int i = 0; i < 4096; ++i) { md = fmax(md, fabs((double)hs[i] - sref[i])); mx = fmax(mx, fabs(sref[i])); }
            md /= mx > 0 ? mx : 1;  // relative to the largest reference score
            printf("round %d  %-14s  %.3f ms  %.3e pairs/s  %.1f TF(alg)  frac %.3f  checksum %.6f  max|d vs first| %.2e\n", r, v.name, ms,
                   B / (ms * 1e-3), B * flop_alg / (ms * 1e-3) / 1e12, B * flop_alg / (ms * 1e-3) / 1e12 / 157.3, cs, md);
        }
    }
    return 0;
}

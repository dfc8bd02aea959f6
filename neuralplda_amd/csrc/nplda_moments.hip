// Weighted first / second moments of paired-embedding rows on gfx950:
//     cnt[c] = sum_k w_c[k],   sum[c][i] = sum_k w_c[k] x[k][i],   sq[c][i][j] = sum_k w_c[k] x[k][i] x[k][j]
// for up to two weight vectors (c = 0, 1) in ONE pass over x.  Two callers:
//   * the Gaussian-backend "training" pass, xvector_GaussianBackend_pytorch.py:30-56 of the reference: per-class
//     sum x and sum x x^T of x = forward_getpaired(x1, x2) with w_0 = [t > 0.5], w_1 = [t < 0.5];
//   * the gradient of DPlda's linear unit (utils/models.py:484-490): with w_0 = dL/ds the feature-weight gradient
//     sum_k g_k [y1 y2^T + y2 y1^T, y1 y1^T + y2 y2^T, y1 + y2] is a fold of sq / sum (see ops.dplda_fold_grad).
// Split-K exact-fp32 MFMA "A^T B" GEMM (A = w .* x, B = x), upper-triangular 64x64 tiles only (the result is
// symmetric), float4-strided operand loads, in-block LDS reduction over the 4 k-quarters in a fixed order, then an
// fp64 reduction over the k-groups -> bitwise deterministic.  Bound: MFMA at large B (2 n^2 flop per row and
// class, halved by symmetry), launch latency at training-batch sizes.
#include <hip/hip_runtime.h>
#include <cstdlib>

#include "nplda_adam_math.h"
#include "nplda_common.h"
#include "nplda_loss_single.h"

namespace {

constexpr int kMaxN = 2 * NPLDA_MAX_DIM;  // paired rows: [y1; y2]
constexpr int kPFm = 6;                   // k4-steps of operand prefetch per wave

struct MomArgs {
    const float* x;
    long long ldx, B;
    int n;                 // valid columns (multiple of 4)
    const float* w0;
    const float* w1;       // unused when NC == 1
    int T;                 // 64-wide tiles per side
    int ntile;             // T (T + 1) / 2 upper-triangular tiles
    int kgroups;
    long long rows_per_group;  // multiple of 16
    int Np;                // 64 T
    float* slab;           // [class][kgroup][Np][Np]   (upper tiles written)
    float* ext;            // [class][kgroup][Np + 4]   column sums, then the weight sum
    float* step_bump;      // optional: Adam's step counter, counted here (one thread) for the update kernel of this step
    // LOSS != 0 (nplda_dplda_update_loss_f32): the weights are dL/ds_i of the loss, formed HERE from (score, target) by the
    // device functions of the loss kernels — and one more block at the end of the grid is the loss kernel itself (sums, loss,
    // dL/dtheta; nplda_loss_single.h): DPlda's recipe step needs no loss launch between its forward and this one
    const float* ls; const float* lt;      // scores, targets (B each, 16-byte aligned, B <= kSingleBlockMax)
    nplda_loss::ThetaPtrs lth; nplda_loss::BetaVals lbeta; float lalpha;
    double* lsums; float* lloss; float* ldtheta; float* lg;  // outputs of the loss block (lg: optional dL/ds)
};

__device__ __forceinline__ void tile_of(int t, int T, int& mt, int& nt) {
    // row-major enumeration of the upper triangle: (0,0) (0,1) ... (0,T-1) (1,1) ...
    mt = 0;
    int rowlen = T;
    while (t >= rowlen) { t -= rowlen; --rowlen; ++mt; }
    nt = mt + t;
}

// LOSS: 0 = weights from w0 / w1; 1 = BCE; 1 + K = SoftCdet with K thresholds (see MomArgs)
template <int NC, int LOSS = 0>
__global__ __launch_bounds__(256, 2) void moments_kernel(const MomArgs a) {
    static_assert(LOSS == 0 || NC == 1, "inline loss weights: one class");
    constexpr int LK = LOSS > 1 ? LOSS - 1 : 1;
    float ltheta[LK], lcn[LK], lct = 0.f, linvN = 0.f;
    if constexpr (LOSS != 0) {
        if (blockIdx.x == gridDim.x - 1) {  // the loss block (block-uniform)
            if constexpr (LOSS == 1) nplda_loss::loss_fused_bce_body(a.ls, a.lt, a.B, a.lth, a.lsums, a.lloss, a.lg, a.ldtheta);
            else nplda_loss::loss_fused_softcdet_body<LK>(a.ls, a.lt, a.B, a.lth, a.lbeta, a.lalpha, a.lsums, a.lloss, a.lg, a.ldtheta);
            return;
        }
        // the batch constants of dL/ds: the label counts by the loss block's own summation pattern (the same per-thread
        // partial sums in the same order: the same N_t, N_n to the last bit, soft labels included)
        __shared__ double ltot[2];
        double lacc[2] = {0.0, 0.0};
        nplda_loss::sums_single_block(a.ls, a.lt, a.B, [&](float, float ti) { lacc[0] += ti; lacc[1] += 1.0f - ti; });
        nplda_loss::block_totals<2>(lacc, ltot, nullptr);
#pragma unroll
        for (int k = 0; k < LK; ++k) ltheta[k] = a.lth.p[k][0];
        if constexpr (LOSS == 1) linvN = (float)(1.0 / (ltot[0] + ltot[1]));
        else nplda_loss::softcdet_consts<LK>(ltot[0], ltot[1], a.lbeta, a.lalpha, lcn, lct);
    }
    __shared__ f32x4 red[4][16 * 64];  // [wave][(ca*4+cb)*64 + lane]  (64 KB), reused per class
    __shared__ f32x4 rede[4][NC][16];
    __shared__ float redc[4][NC];
    const int lane = threadIdx.x & 63, wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int i16 = lane & 15, g4 = lane >> 4;
    const int kg = blockIdx.x / a.ntile;
    if (a.step_bump != nullptr && blockIdx.x == 0 && threadIdx.x == 0) a.step_bump[0] += 1.0f;
    int mt, nt;
    tile_of(blockIdx.x % a.ntile, a.T, mt, nt);
    const int m0 = mt * 64, n0 = nt * 64;
    const bool diag = mt == nt;
    const float* __restrict__ X = a.x;
    const float* __restrict__ W0 = a.w0;
    const float* __restrict__ W1 = a.w1;
    const long long ldx = a.ldx, B = a.B;
    const long long quarter = a.rows_per_group / 4;
    const long long k0 = (long long)kg * a.rows_per_group + wave * quarter;
    long long k1 = k0 + quarter;
    if (k1 > B) k1 = B;
    const bool mval = m0 + 4 * i16 < a.n;
    const bool nval = n0 + 4 * i16 < a.n;
    const int mcol = mval ? m0 + 4 * i16 : 0;
    const int ncol = nval ? n0 + 4 * i16 : 0;
    const f32x4 zero4 = {0.f, 0.f, 0.f, 0.f};

    f32x4 acc[NC][4][4];
#pragma unroll
    for (int c = 0; c < NC; ++c)
#pragma unroll
        for (int ca = 0; ca < 4; ++ca)
#pragma unroll
            for (int cb = 0; cb < 4; ++cb) acc[c][ca][cb] = zero4;
    f32x4 es[NC];
    float ec[NC];
#pragma unroll
    for (int c = 0; c < NC; ++c) { es[c] = zero4; ec[c] = 0.f; }

    // branch-free loads: clamped row / column + select (a predicated load costs an s_waitcnt vmcnt(0) per use)
    struct Frag { f32x4 xa, xb; float w[NC]; };
    auto load = [&](long long row) -> Frag {
        Frag f;
        const bool ok = row < k1;
        const long long rc = row < B ? row : B - 1;
        const f32x4 va = *reinterpret_cast<const f32x4*>(X + rc * ldx + mcol);
        const f32x4 vb = *reinterpret_cast<const f32x4*>(X + rc * ldx + ncol);
        f.xa = mval ? va : zero4;
        f.xb = nval ? vb : zero4;
        // (LOSS != 0: dL/ds_i formed HERE, at load time, as in the first form of this kernel — formed at the point of use, with
        // the batch constants computed behind the first operand loads, the step was 1 us shorter and SoftCdet's weights differed
        // from the loss kernel's in their last bit: the same expression compiled in another context)
        float w0;
        if constexpr (LOSS == 0) w0 = W0[rc];
        else if constexpr (LOSS == 1) w0 = nplda_loss::bce_gi(a.ls[rc], a.lt[rc], ltheta[0], linvN);
        else w0 = nplda_loss::softcdet_gi<LK>(a.ls[rc], a.lt[rc], ltheta, lcn, lct, a.lalpha);
        f.w[0] = ok ? w0 : 0.f;
        if (NC == 2) {
            const float w1 = W1[rc];
            f.w[NC - 1] = ok ? w1 : 0.f;
        }
        return f;
    };

    Frag ring[kPFm];
#pragma unroll
    for (int s = 0; s < kPFm; ++s) ring[s] = load(k0 + 4 * s + g4);
    for (long long kk = k0; kk < k1; kk += 4 * kPFm) {
#pragma unroll
        for (int s = 0; s < kPFm; ++s) {
            const Frag f = ring[s];
            ring[s] = load(kk + 4 * s + g4 + 4 * kPFm);
#pragma unroll
            for (int c = 0; c < NC; ++c) {
                const f32x4 av = f.w[c] * f.xa;
                if (diag) {  // block-uniform
                    es[c] += av;
                    ec[c] += f.w[c];
                }
#pragma unroll
                for (int ca = 0; ca < 4; ++ca)
#pragma unroll
                    for (int cb = 0; cb < 4; ++cb)
                        acc[c][ca][cb] = __builtin_amdgcn_mfma_f32_16x16x4f32(av[ca], f.xb[cb], acc[c][ca][cb], 0, 0, 0);
            }
        }
    }

    if (diag) {
#pragma unroll
        for (int c = 0; c < NC; ++c) {
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                es[c][j] = wave_xor_add(es[c][j], 16);
                es[c][j] = wave_xor_add(es[c][j], 32);
            }
            ec[c] = wave_xor_add(ec[c], 16);
            ec[c] = wave_xor_add(ec[c], 32);
            if (g4 == 0) rede[wave][c][i16] = es[c];
            if (lane == 0) redc[wave][c] = ec[c];
        }
    }
    // ---- in-block reduction over the 4 k-quarters (fixed order), one class at a time through the same LDS ----
#pragma unroll
    for (int c = 0; c < NC; ++c) {
        if (c) __syncthreads();
#pragma unroll
        for (int ca = 0; ca < 4; ++ca)
#pragma unroll
            for (int cb = 0; cb < 4; ++cb) red[wave][(ca * 4 + cb) * 64 + lane] = acc[c][ca][cb];
        __syncthreads();
        // wave `ca` finishes block-row ca: D[i][j] of block (ca, cb) is C[m0 + 4 i + ca][n0 + 4 j + cb];
        // lane (j = i16, g4) holds i = 4 g4 + r
        const int ca = wave;
        f32x4 sum[4];
#pragma unroll
        for (int cb = 0; cb < 4; ++cb) {
            const int idx = (ca * 4 + cb) * 64 + lane;
            sum[cb] = ((red[0][idx] + red[1][idx]) + red[2][idx]) + red[3][idx];
        }
        float* slab = a.slab + ((size_t)c * a.kgroups + kg) * a.Np * a.Np;
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            const int m = m0 + 4 * (4 * g4 + r) + ca;
            const f32x4 v = {sum[0][r], sum[1][r], sum[2][r], sum[3][r]};
            *reinterpret_cast<f32x4*>(slab + (size_t)m * a.Np + n0 + 4 * i16) = v;
        }
    }
    if (diag && wave == 0) {
#pragma unroll
        for (int c = 0; c < NC; ++c) {
            float* eb = a.ext + ((size_t)c * a.kgroups + kg) * (a.Np + 4);
            if (g4 == 0)
                *reinterpret_cast<f32x4*>(eb + m0 + 4 * i16) =
                    ((rede[0][c][i16] + rede[1][c][i16]) + rede[2][c][i16]) + rede[3][c][i16];
            if (lane == 0 && mt == 0) eb[a.Np] = ((redc[0][c] + redc[1][c]) + redc[2][c]) + redc[3][c];
        }
    }
}

struct MomReduceArgs {
    const float* slab;
    const float* ext;
    int nc, n, Np, kgroups, accumulate;
    double* cnt;   // [nc]
    double* sum;   // [nc][n]
    double* sq;    // [nc][n][n]
};

// fp64 sum of one slab entry over the k-groups, in k order; the loads of 40 k-groups are in flight together (the kernel
// is latency-bound: one output per thread, about one block per CU)
__device__ __forceinline__ double slab_sum1(const float* p, size_t stride, int kgroups) {
    constexpr int KB = 40;
    double s = 0.0;
    for (int k0 = 0; k0 < kgroups; k0 += KB) {
        float v[KB];
#pragma unroll
        for (int u = 0; u < KB; ++u) v[u] = p[(size_t)(k0 + u < kgroups ? k0 + u : kgroups - 1) * stride];
#pragma unroll
        for (int u = 0; u < KB; ++u)
            if (k0 + u < kgroups) s += (double)v[u];
    }
    return s;
}

// One block per 16 x 16 tile of sq (per class): a tile on or above the diagonal is summed as it lies in the slabs, a
// tile below it from its mirror image (the lower triangle is not computed) — 16 consecutive floats per row either way, the
// transposition taken through LDS.  (One output per thread with the mirror read element-wise made half of the loads one
// cache line per lane: 22 us at n = 300, now 8.)  The last blocks do the n column sums and the weight sum.
__global__ __launch_bounds__(256) void moments_reduce_kernel(const MomReduceArgs a) {
    const int TB = (a.n + 15) / 16;
    const int tid = threadIdx.x;
    const int ntile = TB * TB;
    const int b = (int)blockIdx.x;
    if (b >= a.nc * ntile) {
        const size_t e = (size_t)(b - a.nc * ntile) * 256 + tid;  // (class, column): n columns, then the weight sum
        if (e >= (size_t)a.nc * (a.n + 1)) return;
        const int c = (int)(e / (a.n + 1)), col = (int)(e % (a.n + 1));
        double* dst = col < a.n ? a.sum + (size_t)c * a.n + col : a.cnt + c;
        const float* p = a.ext + (size_t)c * a.kgroups * (a.Np + 4) + (col < a.n ? col : a.Np);
        const double s = slab_sum1(p, (size_t)a.Np + 4, a.kgroups);
        *dst = a.accumulate ? *dst + s : s;
        return;
    }
    __shared__ double tl[16][17];
    const int c = b / ntile, I = (b % ntile) / TB, J = b % TB;
    const int r = tid >> 4, cc = tid & 15;
    const bool tr = I > J;  // below the diagonal: the mirror tile, transposed
    const int rb = tr ? J : I, cb = tr ? I : J;
    {
        const int rr = 16 * rb + r, col = 16 * cb + cc;
        const float* p = a.slab + (size_t)c * a.kgroups * a.Np * a.Np + (size_t)rr * a.Np + col;
        tl[r][cc] = (rr < a.n && col < a.n) ? slab_sum1(p, (size_t)a.Np * a.Np, a.kgroups) : 0.0;
    }
    __syncthreads();
    const int i = 16 * I + r, j = 16 * J + cc;
    if (i < a.n && j < a.n) {
        // inside a diagonal tile both triangles exist in the slabs ((w x_i) x_j and (w x_j) x_i round differently): the
        // upper one is the value, so that sq is exactly symmetric
        const double s = (tr || (I == J && r > cc)) ? tl[cc][r] : tl[r][cc];
        double* dst = a.sq + (size_t)c * a.n * a.n + (size_t)i * a.n + j;
        *dst = a.accumulate ? *dst + s : s;
    }
}

// DPlda's gradient straight from the slabs (utils/models.py:484-490: one linear unit over [y1 y2^T + y2 y1^T,
// y1 y1^T + y2 y2^T, y1 + y2]): with G = sum_k g_k x_k x_k^T of the paired rows x = [y1 | y2] in blocks G11 G12 / G21 G22,
//   d wlr = [G12 + G21 | G11 + G22 | s1 + s2],  d bias = sum g.
// Every G element is the fp64 sum of its slab entries in k-group order — the value moments_reduce_kernel leaves in `sq` —
// and the two are added in fp64 before the one rounding to fp32: the bits of ops.dplda_fold_grad on that kernel's output,
// without the 0.7 MB fp64 matrix, its reduction launch and the five torch kernels of the fold.
struct DpldaFoldArgs {
    const float* slab;
    const float* ext;
    int kgroups, Np, D1;
    float* dw;   // 2 D1^2 + D1   (UPDATE: optional copy of the applied gradient)
    float* db;   // 1
    // UPDATE form (nplda_dplda_update_f32): the thread that has formed a gradient element goes on to torch.optim.Adam's update
    // of that parameter (csrc/nplda_adam_math.h: the arithmetic of nplda_adam_step_f32) and stores the new value into the
    // parameter AND into the quadratic-form image the next forward scores with (gb_pack_dplda_f32's layout) — the fold, the
    // optimiser and the two pack launches of the recipe step in one
    float* wlr; float* blr;          // logistic_regres.weight (2 D1^2 + D1), .bias (1)
    float* m; float* v;              // exp_avg / exp_avg_sq: [wlr | blr | thresholds]
    const float* step;               // step[0] already counts this step (the moments kernel bumped it)
    float lr, beta1, beta2, eps, wd;
    float* image;                    // the GaussianBackend-layout image (may be null)
    long long oG, ov, oc; int NB;    // its offsets (floats): G fragments, v, c
    float* theta[4]; const float* dtheta; int K;  // SoftCdet thresholds trained with the unit (K = 0: none)
    const float* loss; double* loss_sum;          // optional: loss_sum[0] += loss[0] (the training log's running sum; by the bias thread)
};

// One block per 16 x 16 tile of an output block.  Half of the entries a tile needs sit mirrored in the slabs (G21 = G12^T;
// the lower triangle of G11 / G22 is not computed): read element-wise they are 64 different cache lines per wave-load and
// the kernel took 21 us.  Here a thread sums ONE entry of the tile as it lies in memory (16 consecutive floats per row)
// and the mirror is taken through LDS.
template <bool UPDATE>
__global__ __launch_bounds__(256) void dplda_fold_kernel(const DpldaFoldArgs a) {
    const int D1 = a.D1, TB = (D1 + 15) / 16;
    const size_t n2 = (size_t)D1 * D1, sst = (size_t)a.Np * a.Np;
    const int tid = threadIdx.x;
    const int b = (int)blockIdx.x;
    nplda_adam::Consts ac = {};
    if constexpr (UPDATE) ac = nplda_adam::consts_for(a.step[0], a.lr, a.beta1, a.beta2, a.eps, a.wd);
    // gradient element e of [wlr | blr] -> Adam -> the parameter; returns the new value
    auto apply = [&](size_t e, float grad, float* prm) {
        float m = a.m[e], v = a.v[e];
        const float pn = nplda_adam::update(prm[0], grad, m, v, ac);
        prm[0] = pn;
        a.m[e] = m;
        a.v[e] = v;
        return pn;
    };
    if (b >= 2 * TB * TB) {  // s1 + s2 and sum g: one output per thread, consecutive columns
        const size_t c = (size_t)(b - 2 * TB * TB) * 256 + tid;
        const size_t est = (size_t)a.Np + 4;
        if (c < (size_t)D1) {
            const float gr = (float)(slab_sum1(a.ext + c, est, a.kgroups) + slab_sum1(a.ext + D1 + c, est, a.kgroups));
            if (a.dw) a.dw[2 * n2 + c] = gr;
            if constexpr (UPDATE) {
                const float pn = apply(2 * n2 + c, gr, a.wlr + 2 * n2 + c);
                if (a.image) {  // v = [ws; ws] (gb_vc_dplda_kernel)
                    a.image[a.ov + c] = pn;
                    a.image[a.ov + 16 * a.NB + c] = pn;
                }
            }
        } else if (c == (size_t)D1) {
            const float gr = (float)slab_sum1(a.ext + a.Np, est, a.kgroups);
            if (a.db) a.db[0] = gr;
            if constexpr (UPDATE) {
                const float pn = apply(2 * n2 + D1, gr, a.blr);
                if (a.image) a.image[a.oc] = pn;  // c = the bias
                if (a.loss_sum && a.loss) a.loss_sum[0] += (double)a.loss[0];
            }
        } else if (UPDATE && c > (size_t)D1 && c <= (size_t)D1 + a.K) {  // the thresholds' own Adam step
            const int k = (int)(c - D1 - 1);
            apply(2 * n2 + D1 + 1 + k, a.dtheta[k], a.theta[k]);
        }
        return;
    }
    __shared__ double t1[16][17], t2[16][17];
    const int which = b / (TB * TB);  // 0: G12 + G21, 1: G11 + G22
    const int I = (b % (TB * TB)) / TB, J = b % TB;
    const int r = tid >> 4, c = tid & 15;
    // the element this thread will update: its parameter and moments are asked for NOW, next to the slab loads — fetched
    // inside apply(), behind the slab sums and the barrier, they were a second memory round trip of an 8.5 us kernel
    float pre_p = 0.f, pre_m = 0.f, pre_v = 0.f;
    {
        const int i0 = 16 * I + r, j0 = 16 * J + c;
        if (UPDATE && i0 < D1 && j0 < D1) {
            const size_t e0 = (size_t)which * n2 + (size_t)i0 * D1 + j0;
            pre_p = a.wlr[e0];
            pre_m = a.m[e0];
            pre_v = a.v[e0];
        }
    }
    // source tiles (row block, column block in units of 16 inside a D1 x D1 block; row / column offsets of that block)
    const bool tr1 = which == 1 && I > J;        // G11 tile below the diagonal: its mirror image, transposed
    const int rb1 = tr1 ? J : I, cb1 = tr1 ? I : J;
    const int ro1 = 0, co1 = which == 0 ? D1 : 0;
    const bool tr2 = which == 0 || I > J;        // G21 = G12^T always; G22 like G11
    const int rb2 = tr2 ? J : I, cb2 = tr2 ? I : J;
    const int ro2 = which == 0 ? 0 : D1, co2 = D1;
    {
        const int rr = 16 * rb1 + r, cc = 16 * cb1 + c;
        t1[r][c] = (rr < D1 && cc < D1) ? slab_sum1(a.slab + (size_t)(ro1 + rr) * a.Np + co1 + cc, sst, a.kgroups) : 0.0;
    }
    {
        const int rr = 16 * rb2 + r, cc = 16 * cb2 + c;
        t2[r][c] = (rr < D1 && cc < D1) ? slab_sum1(a.slab + (size_t)(ro2 + rr) * a.Np + co2 + cc, sst, a.kgroups) : 0.0;
    }
    __syncthreads();
    const int i = 16 * I + r, j = 16 * J + c;
    if (i < D1 && j < D1) {
        // a diagonal tile of G11 / G22 holds both triangles; the upper one is the value (moments_reduce_kernel)
        const bool diag = which == 1 && I == J;
        const double v1 = (tr1 || (diag && r > c)) ? t1[c][r] : t1[r][c];
        const double v2 = (tr2 || (diag && r > c)) ? t2[c][r] : t2[r][c];
        const size_t e = (size_t)which * n2 + (size_t)i * D1 + j;
        const float gr = (float)(v1 + v2);
        if (a.dw) a.dw[e] = gr;
        if constexpr (UPDATE) {
            float m = pre_m, v = pre_v;  // (apply() with the three loads taken at the top)
            const float pn = nplda_adam::update(pre_p, gr, m, v, ac);
            a.wlr[e] = pn;
            a.m[e] = m;
            a.v[e] = v;
            if (a.image) {
                // gb_pack_kernel (dplda): block (h_out, h_in) of G at [f][k] is Ww[f][k] on the diagonal (which == 1), Wb[f][k]
                // off it (which == 0); fragment element (kb, nb, lane, e4) holds f = 16 nb + (lane & 15), k = 16 kb + 4 (lane >> 4) + e4
                const int nb = i >> 4, kb = j >> 4;
                const size_t in_blk = ((size_t)(kb * a.NB + nb) * 64 + (size_t)((i & 15) + 16 * ((j & 15) >> 2))) * 4 + (j & 3);
                const size_t hsz = (size_t)a.NB * a.NB * 256;
                const int hh0 = which == 1 ? 0 : 1, hh1 = which == 1 ? 3 : 2;  // 2 h_out + h_in
                a.image[a.oG + hh0 * hsz + in_blk] = pn;
                a.image[a.oG + hh1 * hsz + in_blk] = pn;
            }
        }
    }
}

struct MomPlan {
    int T, ntile, kgroups, Np;
    long long rows_per_group;
    size_t slab_floats, ext_floats;
};

MomPlan mom_plan(long long B, int n) {
    MomPlan p;
    p.T = (n + 63) / 64;
    p.Np = 64 * p.T;
    p.ntile = p.T * (p.T + 1) / 2;
    long long kg = 512 / p.ntile;  // ~2 blocks per CU in one wave of blocks
    if (kg < 1) kg = 1;
    // at least 64 rows per k-group (16 per wave: four k4-steps) — NPLDA_MOM_MIN_ROWS for A/B runs.  With 16 (one k4-step per
    // wave) a 256-row batch was 16 k-groups x 21 tiles = 336 blocks that each staged 64 KB through LDS and wrote a 16 KB slab
    // tile for four MFMAs' worth of rows, and the fold kernel summed 16 slabs per element (round 6: 13.4 + 8.9 us)
    static const long long min_rows = getenv("NPLDA_MOM_MIN_ROWS") ? atoll(getenv("NPLDA_MOM_MIN_ROWS")) : 64;
    const long long mr = min_rows < 16 ? 16 : min_rows;
    const long long maxkg = (B + mr - 1) / mr;
    if (kg > maxkg) kg = maxkg < 1 ? 1 : maxkg;
    long long rpg = (B + kg - 1) / kg;
    rpg = (rpg + 15) / 16 * 16;
    if (rpg < 16) rpg = 16;
    p.rows_per_group = rpg;
    p.kgroups = (int)((B + rpg - 1) / rpg);
    if (p.kgroups < 1) p.kgroups = 1;
    p.slab_floats = (size_t)2 * p.kgroups * p.Np * p.Np;
    p.ext_floats = (size_t)2 * p.kgroups * (p.Np + 4);
    return p;
}

}  // namespace

// kind < 0: the weights are `g` (nplda_dplda_update_f32); kind 0 / 1 (SoftCdet / BCE): they are formed inside the moments launch
// from (ls, lt) and the loss block rides in it (nplda_dplda_update_loss_f32)
struct DpldaLossIn {
    int kind, K;
    const float* s; const float* t; const float* const* theta; const float* beta; float alpha;
    double* sums; float* loss; float* dtheta; float* g_out;
};
static int dplda_update_impl(const float* paired, int64_t B, int64_t ld, int D1, const float* g, float* wlr, float* blr,
                             float* exp_avg, float* exp_avg_sq, float* const* thetas, const float* dtheta, int K, float* step,
                             float lr, float beta1, float beta2, float eps, float weight_decay, void* image, int D0,
                             float* grad_out, const float* loss, double* loss_sum, void* workspace, size_t workspace_bytes,
                             const DpldaLossIn* li, nplda_stream_t stream) {
    const int n = 2 * D1;
    if (B <= 0 || D1 <= 0 || K < 0 || K > 4) return NPLDA_EINVAL;
    if (loss_sum && !loss) return NPLDA_EINVAL;
    if (n > kMaxN || (n & 3)) return NPLDA_EUNSUPPORTED;
    if (!paired || (!g && !li) || !wlr || !blr || !exp_avg || !exp_avg_sq || !step || !workspace) return NPLDA_EINVAL;
    if (K > 0 && (!thetas || !dtheta)) return NPLDA_EINVAL;
    if (ld < n || (ld & 3) || !nplda_aligned16(paired) || !nplda_aligned16(workspace)) return NPLDA_EINVAL;
    const int NB = nplda_kernel_nb(D1, D1);
    if (image && (NB == 0 || D0 <= 0 || (D0 % 4) != 0 || !nplda_aligned16(image))) return NPLDA_EINVAL;
    const MomPlan p = mom_plan(B, n);
    if (workspace_bytes < (p.slab_floats + p.ext_floats) * sizeof(float)) return NPLDA_ENOSPC;
    hipStream_t st = (hipStream_t)stream;
    MomArgs a = {};
    a.x = paired; a.ldx = ld; a.B = B; a.n = n; a.w0 = g; a.w1 = nullptr;
    a.T = p.T; a.ntile = p.ntile; a.kgroups = p.kgroups; a.rows_per_group = p.rows_per_group; a.Np = p.Np;
    a.slab = (float*)workspace;
    a.ext = a.slab + p.slab_floats;
    a.step_bump = step;  // the step is counted here: the update kernel below reads t = step[0]
    const unsigned mgrid = (unsigned)(p.ntile * p.kgroups);
    if (!li) {
        hipLaunchKernelGGL(moments_kernel<1>, dim3(mgrid), dim3(256), 0, st, a);
    } else {
        const int LK = li->kind == 1 ? 1 : li->K;
        if ((li->kind != 0 && li->kind != 1) || LK < 1 || LK > nplda_loss::kMaxK) return NPLDA_EINVAL;
        if (!li->s || !li->t || !li->theta || !li->sums || !li->loss || !li->dtheta || (li->kind == 0 && !li->beta)) return NPLDA_EINVAL;
        if (B > nplda_loss::kSingleBlockMax || !nplda_aligned16(li->s) || !nplda_aligned16(li->t) ||
            (li->g_out && !nplda_aligned16(li->g_out)))
            return NPLDA_EUNSUPPORTED;  // (the caller runs nplda_loss_fwd_bwd_f32 + nplda_dplda_update_f32 instead)
        a.ls = li->s; a.lt = li->t; a.lalpha = li->alpha;
        a.lsums = li->sums; a.lloss = li->loss; a.ldtheta = li->dtheta; a.lg = li->g_out;
        for (int k = 0; k < LK; ++k) {
            if (!li->theta[k]) return NPLDA_EINVAL;
            a.lth.p[k] = li->theta[k];
            if (li->kind == 0) a.lbeta.b[k] = li->beta[k];  // beta is a HOST array (config constants)
        }
        const dim3 grid(mgrid + 1), block(256);  // + the loss block
        if (li->kind == 1) hipLaunchKernelGGL((moments_kernel<1, 1>), grid, block, 0, st, a);
        else switch (LK) {
            case 1: hipLaunchKernelGGL((moments_kernel<1, 2>), grid, block, 0, st, a); break;
            case 2: hipLaunchKernelGGL((moments_kernel<1, 3>), grid, block, 0, st, a); break;
            case 3: hipLaunchKernelGGL((moments_kernel<1, 4>), grid, block, 0, st, a); break;
            default: hipLaunchKernelGGL((moments_kernel<1, 5>), grid, block, 0, st, a); break;
        }
    }
    if (int rc = nplda_launch_status()) return rc;
    DpldaFoldArgs f = {};
    f.slab = a.slab; f.ext = a.ext; f.kgroups = p.kgroups; f.Np = p.Np; f.D1 = D1;
    f.dw = grad_out; f.db = grad_out ? grad_out + (size_t)2 * D1 * D1 + D1 : nullptr;
    f.wlr = wlr; f.blr = blr; f.m = exp_avg; f.v = exp_avg_sq; f.step = step;
    f.lr = lr; f.beta1 = beta1; f.beta2 = beta2; f.eps = eps; f.wd = weight_decay;
    f.loss = loss; f.loss_sum = loss_sum;
    f.image = (float*)image;
    if (image) {  // gb_layout(D0, D1) of nplda_gb.hip: [W1 fragments | G | b1 | v | c]
        const long long KS1 = (D0 + 15) / 16;
        f.NB = NB;
        f.oG = KS1 * NB * 256;
        f.ov = f.oG + 4LL * NB * NB * 256 + (long long)NB * 16;
        f.oc = f.ov + 2LL * NB * 16;
    }
    for (int k = 0; k < K; ++k) {
        if (!thetas[k]) return NPLDA_EINVAL;
        f.theta[k] = thetas[k];
    }
    f.dtheta = dtheta; f.K = K;
    const int TB = (D1 + 15) / 16;
    hipLaunchKernelGGL(dplda_fold_kernel<true>, dim3((unsigned)(2 * TB * TB + (D1 + 1 + K + 255) / 256)), dim3(256), 0, st, f);
    return nplda_launch_status();
}

extern "C" {

size_t nplda_moments_workspace_bytes(int64_t B, int n) {
    if (B < 0 || n <= 0 || n > kMaxN || (n & 3)) return 0;
    const MomPlan p = mom_plan(B, n);
    return (p.slab_floats + p.ext_floats) * sizeof(float);
}

int nplda_weighted_moments_f32(const float* x, int64_t B, int64_t ldx, int n, const float* w0, const float* w1,
                               double* cnt, double* sum, double* sq, int accumulate, void* workspace,
                               size_t workspace_bytes, nplda_stream_t stream) {
    if (B < 0 || n <= 0) return NPLDA_EINVAL;
    if (n > kMaxN || (n & 3)) return NPLDA_EUNSUPPORTED;
    if (!cnt || !sum || !sq) return NPLDA_EINVAL;
    const int nc = w1 ? 2 : 1;
    hipStream_t st = (hipStream_t)stream;
    if (B == 0) {
        if (!accumulate) {
            if (hipError_t e = hipMemsetAsync(cnt, 0, sizeof(double) * nc, st)) return (int)e;
            if (hipError_t e = hipMemsetAsync(sum, 0, sizeof(double) * nc * n, st)) return (int)e;
            if (hipError_t e = hipMemsetAsync(sq, 0, sizeof(double) * nc * n * n, st)) return (int)e;
        }
        return NPLDA_OK;
    }
    if (!x || !w0 || !workspace) return NPLDA_EINVAL;
    if (ldx < n || (ldx & 3) || !nplda_aligned16(x) || !nplda_aligned16(workspace)) return NPLDA_EINVAL;
    const MomPlan p = mom_plan(B, n);
    if (workspace_bytes < (p.slab_floats + p.ext_floats) * sizeof(float)) return NPLDA_ENOSPC;
    MomArgs a;
    a.x = x; a.ldx = ldx; a.B = B; a.n = n; a.w0 = w0; a.w1 = w1;
    a.T = p.T; a.ntile = p.ntile; a.kgroups = p.kgroups; a.rows_per_group = p.rows_per_group; a.Np = p.Np;
    a.slab = (float*)workspace;
    a.ext = a.slab + p.slab_floats;
    a.step_bump = nullptr;
    const dim3 grid((unsigned)(p.ntile * p.kgroups));
    if (nc == 2) hipLaunchKernelGGL(moments_kernel<2>, grid, dim3(256), 0, st, a);
    else hipLaunchKernelGGL(moments_kernel<1>, grid, dim3(256), 0, st, a);
    if (int rc = nplda_launch_status()) return rc;
    MomReduceArgs r;
    r.slab = a.slab; r.ext = a.ext; r.nc = nc; r.n = n; r.Np = p.Np; r.kgroups = p.kgroups; r.accumulate = accumulate;
    r.cnt = cnt; r.sum = sum; r.sq = sq;
    const int TBr = (n + 15) / 16;
    hipLaunchKernelGGL(moments_reduce_kernel, dim3((unsigned)(nc * TBr * TBr + (nc * (n + 1) + 255) / 256)), dim3(256), 0, st, r);
    return nplda_launch_status();
}

int nplda_dplda_grad_f32(const float* paired, int64_t B, int64_t ld, int D1, const float* g, float* dw, float* db,
                         void* workspace, size_t workspace_bytes, nplda_stream_t stream) {
    const int n = 2 * D1;
    if (B <= 0 || D1 <= 0) return NPLDA_EINVAL;
    if (n > kMaxN || (n & 3)) return NPLDA_EUNSUPPORTED;
    if (!paired || !g || !dw || !db || !workspace) return NPLDA_EINVAL;
    if (ld < n || (ld & 3) || !nplda_aligned16(paired) || !nplda_aligned16(workspace)) return NPLDA_EINVAL;
    const MomPlan p = mom_plan(B, n);
    if (workspace_bytes < (p.slab_floats + p.ext_floats) * sizeof(float)) return NPLDA_ENOSPC;
    hipStream_t st = (hipStream_t)stream;
    MomArgs a;
    a.x = paired; a.ldx = ld; a.B = B; a.n = n; a.w0 = g; a.w1 = nullptr;
    a.T = p.T; a.ntile = p.ntile; a.kgroups = p.kgroups; a.rows_per_group = p.rows_per_group; a.Np = p.Np;
    a.slab = (float*)workspace;
    a.ext = a.slab + p.slab_floats;
    a.step_bump = nullptr;
    hipLaunchKernelGGL(moments_kernel<1>, dim3((unsigned)(p.ntile * p.kgroups)), dim3(256), 0, st, a);
    if (int rc = nplda_launch_status()) return rc;
    DpldaFoldArgs f = {};
    f.slab = a.slab; f.ext = a.ext; f.kgroups = p.kgroups; f.Np = p.Np; f.D1 = D1; f.dw = dw; f.db = db;
    const int TB = (D1 + 15) / 16;
    hipLaunchKernelGGL(dplda_fold_kernel<false>, dim3((unsigned)(2 * TB * TB + (D1 + 1 + 255) / 256)), dim3(256), 0, st, f);
    return nplda_launch_status();
}

int nplda_dplda_update_f32(const float* paired, int64_t B, int64_t ld, int D1, const float* g, float* wlr, float* blr,
                           float* exp_avg, float* exp_avg_sq, float* const* thetas, const float* dtheta, int K, float* step,
                           float lr, float beta1, float beta2, float eps, float weight_decay, void* image, int D0,
                           float* grad_out, const float* loss, double* loss_sum, void* workspace, size_t workspace_bytes,
                           nplda_stream_t stream) {
    if (!g) return NPLDA_EINVAL;
    return dplda_update_impl(paired, B, ld, D1, g, wlr, blr, exp_avg, exp_avg_sq, thetas, dtheta, K, step, lr, beta1, beta2, eps,
                             weight_decay, image, D0, grad_out, loss, loss_sum, workspace, workspace_bytes, nullptr, stream);
}

int nplda_dplda_update_loss_f32(const float* paired, int64_t B, int64_t ld, int D1, const float* s, const float* t, int kind,
                                const float* const* loss_theta, const float* beta, int loss_K, float alpha, double* sums,
                                float* loss, float* dtheta, float* g_out, float* wlr, float* blr, float* exp_avg,
                                float* exp_avg_sq, float* const* thetas, int K, float* step, float lr, float beta1, float beta2,
                                float eps, float weight_decay, void* image, int D0, float* grad_out, double* loss_sum,
                                void* workspace, size_t workspace_bytes, nplda_stream_t stream) {
    const DpldaLossIn li = {kind, loss_K, s, t, loss_theta, beta, alpha, sums, loss, dtheta, g_out};
    return dplda_update_impl(paired, B, ld, D1, nullptr, wlr, blr, exp_avg, exp_avg_sq, thetas, dtheta, K, step, lr, beta1, beta2,
                             eps, weight_decay, image, D0, grad_out, loss, loss_sum, workspace, workspace_bytes, &li, stream);
}

}  // extern "C"

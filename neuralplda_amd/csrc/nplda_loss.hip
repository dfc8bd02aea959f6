// nplda_loss.hip — SoftCdet / BCE losses with their hand-derived backward (gfx950).
//
// Replaces utils/models.py:384-399 (softcdet, crossentropy, loss) and the autograd graph behind
// them (~40 tiny launches forward, ~60 backward) by two launches:
//   1. nplda_loss_sums_f32   : one pass over (s, t) accumulating every batch-global sum the loss
//                              and its gradient need, in fp64, into a small device vector
//      sums = [N_t, N_n, {S_miss_k, S_fa_k, D_t_k, D_n_k} for k < K]          (SoftCdet)
//      sums = [N_t, N_n, sum of BCE terms, sum (p - t)]                        (BCE)
//      Every entry is ADDITIVE over shards of the batch, so under data parallelism this vector
//      (<= 18 doubles) is what gets all-reduced — the counts N_t, N_n are batch-global in the
//      reference's formula (utils/models.py:386).
//   2. nplda_loss_finish_f32 : loss, dL/dtheta_k and g_i = dL/ds_i from the (global) sums.
// sigma'(v) is evaluated as e/(1+e)^2, e = exp(-|v|) (no 1 - sigma cancellation: the reference's
// fp32 autograd loses ~1e-3 relative accuracy there, see tests/test_oracle_golden.py).
#include "nplda_loss_math.h"
#include "nplda_loss_single.h"

namespace {

using namespace nplda_loss;

// `single`: the launch is ONE block that covers the whole batch — it stores its sums (no zero-fill of `out` before the
// launch, no atomics: one graph node less in the training step and a fixed summation order).
template <int NS>
__device__ __forceinline__ void block_reduce_add(double (&acc)[NS], double* out, int single) {
    __shared__ double red[kThreads / 64][NS];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
#pragma unroll
    for (int i = 0; i < NS; ++i) {
        const double v = wave_sum_d(acc[i]);
        if (lane == 0) red[wave][i] = v;
    }
    __syncthreads();
    if (threadIdx.x < NS) {
        double v = 0.0;
#pragma unroll
        for (int w = 0; w < kThreads / 64; ++w) v += red[w][threadIdx.x];
        if (single) out[threadIdx.x] = v;
        else atomicAdd(out + threadIdx.x, v);
    }
}

// kind 0 = SoftCdet, 1 = BCE, 2 = hard Cdet (utils/models.py:401-404: step functions, strict < / >)
template <int K, bool HARD>
__global__ __launch_bounds__(kThreads) void loss_sums_softcdet(const float* __restrict__ s,
                                                               const float* __restrict__ t, long long B,
                                                               ThetaPtrs th, float alpha, double* sums, int single) {
    constexpr int NS = 2 + 4 * K;
    double acc[NS];
#pragma unroll
    for (int i = 0; i < NS; ++i) acc[i] = 0.0;
    float theta[K];
#pragma unroll
    for (int k = 0; k < K; ++k) theta[k] = th.p[k][0];
    auto accumulate = [&](float si, float ti) { softcdet_accumulate<K, HARD>(si, ti, theta, alpha, acc); };
    if (single) {
        sums_single_block(s, t, B, accumulate);
    } else {
        const long long stride = (long long)gridDim.x * kThreads;
        for (long long i = (long long)blockIdx.x * kThreads + threadIdx.x; i < B; i += stride) accumulate(s[i], t[i]);
    }
    block_reduce_add<NS>(acc, sums, single);
}

__global__ __launch_bounds__(kThreads) void loss_sums_bce(const float* __restrict__ s, const float* __restrict__ t,
                                                          long long B, ThetaPtrs th, double* sums, int single) {
    double acc[4] = {0.0, 0.0, 0.0, 0.0};
    const float theta = th.p[0][0];
    auto accumulate = [&](float si, float ti) { bce_accumulate(si, ti, theta, acc); };
    if (single) {
        sums_single_block(s, t, B, accumulate);
    } else {
        const long long stride = (long long)gridDim.x * kThreads;
        for (long long i = (long long)blockIdx.x * kThreads + threadIdx.x; i < B; i += stride) accumulate(s[i], t[i]);
    }
    block_reduce_add<4>(acc, sums, single);
}

template <int K>
__global__ __launch_bounds__(kThreads) void loss_finish_softcdet(const float* __restrict__ s,
                                                                 const float* __restrict__ t, long long B,
                                                                 ThetaPtrs th, BetaVals beta, float alpha,
                                                                 const double* __restrict__ sums, float* loss,
                                                                 float* __restrict__ g, float* dtheta) {
    const double Nt = sums[0], Nn = sums[1];
    if (blockIdx.x == 0 && threadIdx.x == 0) softcdet_scalars<K>(sums, beta, alpha, loss, dtheta);
    if (g == nullptr) return;
    float theta[K], cn[K], ct;
#pragma unroll
    for (int k = 0; k < K; ++k) theta[k] = th.p[k][0];
    softcdet_consts<K>(Nt, Nn, beta, alpha, cn, ct);
    const long long stride = (long long)gridDim.x * kThreads;
    for (long long i = (long long)blockIdx.x * kThreads + threadIdx.x; i < B; i += stride) {
        g[i] = softcdet_gi<K>(s[i], t[i], theta, cn, ct, alpha);
    }
}

__global__ __launch_bounds__(kThreads) void loss_finish_bce(const float* __restrict__ s, const float* __restrict__ t,
                                                            long long B, ThetaPtrs th,
                                                            const double* __restrict__ sums, float* loss,
                                                            float* __restrict__ g, float* dtheta) {
    const double N = sums[0] + sums[1];
    if (blockIdx.x == 0 && threadIdx.x == 0) bce_scalars(sums, loss, dtheta);
    if (g == nullptr) return;
    const float theta = th.p[0][0];
    const float invN = (float)(1.0 / N);
    const long long stride = (long long)gridDim.x * kThreads;
    for (long long i = (long long)blockIdx.x * kThreads + threadIdx.x; i < B; i += stride) {
        g[i] = bce_gi(s[i], t[i], theta, invN);
    }
}

// ---- fused small-batch loss: sums, loss, dL/ds and dL/dtheta in ONE single-block launch (B <= kSingleBlockMax) -----------
// Same arithmetic, in the same order, as loss_sums (single-block form) followed by loss_finish: the block's totals stay
// in LDS, the batch stays in registers (four 16-byte groups per thread), so the gradient pass needs no second launch.
template <int K>
__global__ __launch_bounds__(kThreads) void loss_fused_softcdet(const float* __restrict__ s, const float* __restrict__ t,
                                                                long long B, ThetaPtrs th, BetaVals beta, float alpha,
                                                                double* sums, float* loss, float* __restrict__ g,
                                                                float* dtheta) {
    loss_fused_softcdet_body<K>(s, t, B, th, beta, alpha, sums, loss, g, dtheta);
}

__global__ __launch_bounds__(kThreads) void loss_fused_bce(const float* __restrict__ s, const float* __restrict__ t,
                                                           long long B, ThetaPtrs th, double* sums, float* loss,
                                                           float* __restrict__ g, float* dtheta) {
    loss_fused_bce_body(s, t, B, th, sums, loss, g, dtheta);
}

unsigned grid_for(long long B) {
    long long b = (B + kThreads - 1) / kThreads;
    if (b < 1) b = 1;
    if (b > 1024) b = 1024;
    return (unsigned)b;
}

}  // namespace

extern "C" {

int nplda_loss_nsums(int K, int kind) {
    if (kind == 1) return 4;
    if ((kind != 0 && kind != 2) || K < 1 || K > kMaxK) return 0;
    return nsums(K, kind);
}

int nplda_loss_sums_f32(const float* s, const float* t, int64_t B, const float* const* theta, int K, float alpha,
                        int kind, double* sums, nplda_stream_t stream) {
    const int ns = nplda_loss_nsums(K, kind);
    if (ns == 0) return (kind >= 0 && kind <= 2) ? NPLDA_EUNSUPPORTED : NPLDA_EINVAL;
    if (B < 0 || !sums || !theta) return NPLDA_EINVAL;
    if (B > 0 && (!s || !t)) return NPLDA_EINVAL;
    ThetaPtrs th = {};
    for (int k = 0; k < (kind == 1 ? 1 : K); ++k) {
        if (!theta[k]) return NPLDA_EINVAL;
        th.p[k] = theta[k];
    }
    hipStream_t st = (hipStream_t)stream;
    const int single = (B > 0 && B <= kSingleBlockMax && nplda_aligned16(s) && nplda_aligned16(t)) ? 1 : 0;
    if (!single) {
        hipError_t e = hipMemsetAsync(sums, 0, sizeof(double) * ns, st);
        if (e != hipSuccess) return (int)e;
    }
    if (B == 0) return NPLDA_OK;
    const dim3 grid(single ? 1u : grid_for(B)), block(kThreads);
    if (kind == 1) {
        hipLaunchKernelGGL(loss_sums_bce, grid, block, 0, st, s, t, (long long)B, th, sums, single);
    } else if (kind == 2) {
        switch (K) {
            case 1: hipLaunchKernelGGL((loss_sums_softcdet<1, true>), grid, block, 0, st, s, t, (long long)B, th, alpha, sums, single); break;
            case 2: hipLaunchKernelGGL((loss_sums_softcdet<2, true>), grid, block, 0, st, s, t, (long long)B, th, alpha, sums, single); break;
            case 3: hipLaunchKernelGGL((loss_sums_softcdet<3, true>), grid, block, 0, st, s, t, (long long)B, th, alpha, sums, single); break;
            default: hipLaunchKernelGGL((loss_sums_softcdet<4, true>), grid, block, 0, st, s, t, (long long)B, th, alpha, sums, single); break;
        }
    } else {
        switch (K) {
            case 1: hipLaunchKernelGGL((loss_sums_softcdet<1, false>), grid, block, 0, st, s, t, (long long)B, th, alpha, sums, single); break;
            case 2: hipLaunchKernelGGL((loss_sums_softcdet<2, false>), grid, block, 0, st, s, t, (long long)B, th, alpha, sums, single); break;
            case 3: hipLaunchKernelGGL((loss_sums_softcdet<3, false>), grid, block, 0, st, s, t, (long long)B, th, alpha, sums, single); break;
            default: hipLaunchKernelGGL((loss_sums_softcdet<4, false>), grid, block, 0, st, s, t, (long long)B, th, alpha, sums, single); break;
        }
    }
    return nplda_launch_status();
}

int nplda_loss_finish_f32(const float* s, const float* t, int64_t B, const float* const* theta, const float* beta,
                          int K, float alpha, int kind, const double* sums, float* loss, float* g, float* dtheta,
                          nplda_stream_t stream) {
    const int ns = nplda_loss_nsums(K, kind);
    if (ns == 0) return (kind >= 0 && kind <= 2) ? NPLDA_EUNSUPPORTED : NPLDA_EINVAL;
    if (B < 0 || !sums || !theta) return NPLDA_EINVAL;
    if (B > 0 && g && (!s || !t)) return NPLDA_EINVAL;
    if (kind != 1 && !beta) return NPLDA_EINVAL;
    if (kind == 2 && (g || dtheta)) return NPLDA_EINVAL;  // the hard cost has no gradient
    ThetaPtrs th = {};
    BetaVals bv = {};
    for (int k = 0; k < (kind == 1 ? 1 : K); ++k) {
        if (!theta[k]) return NPLDA_EINVAL;
        th.p[k] = theta[k];
        if (kind != 1) bv.b[k] = beta[k];  // beta is a HOST array (config constants, utils/NpldaConf.py:38)
    }
    hipStream_t st = (hipStream_t)stream;
    const dim3 grid(g ? grid_for(B) : 1), block(kThreads);
    if (kind == 1) {
        hipLaunchKernelGGL(loss_finish_bce, grid, block, 0, st, s, t, (long long)B, th, sums, loss, g, dtheta);
    } else {
        switch (K) {
            case 1: hipLaunchKernelGGL(loss_finish_softcdet<1>, grid, block, 0, st, s, t, (long long)B, th, bv, alpha, sums, loss, g, dtheta); break;
            case 2: hipLaunchKernelGGL(loss_finish_softcdet<2>, grid, block, 0, st, s, t, (long long)B, th, bv, alpha, sums, loss, g, dtheta); break;
            case 3: hipLaunchKernelGGL(loss_finish_softcdet<3>, grid, block, 0, st, s, t, (long long)B, th, bv, alpha, sums, loss, g, dtheta); break;
            default: hipLaunchKernelGGL(loss_finish_softcdet<4>, grid, block, 0, st, s, t, (long long)B, th, bv, alpha, sums, loss, g, dtheta); break;
        }
    }
    return nplda_launch_status();
}

int nplda_loss_fwd_bwd_f32(const float* s, const float* t, int64_t B, const float* const* theta, const float* beta,
                           int K, float alpha, int kind, double* sums, float* loss, float* g, float* dtheta,
                           nplda_stream_t stream) {
    const int ns = nplda_loss_nsums(K, kind);
    if (ns == 0) return (kind >= 0 && kind <= 2) ? NPLDA_EUNSUPPORTED : NPLDA_EINVAL;
    if (kind == 2) return NPLDA_EINVAL;  // the hard cost has no gradient: use the two passes
    if (B < 0 || !sums || !theta || !loss || !g || !dtheta) return NPLDA_EINVAL;
    if (B > 0 && (!s || !t)) return NPLDA_EINVAL;
    if (kind != 1 && !beta) return NPLDA_EINVAL;
    const bool fused = B > 0 && B <= kSingleBlockMax && nplda_aligned16(s) && nplda_aligned16(t) && nplda_aligned16(g);
    if (!fused) {
        if (int rc = nplda_loss_sums_f32(s, t, B, theta, K, alpha, kind, sums, stream)) return rc;
        return nplda_loss_finish_f32(s, t, B, theta, beta, K, alpha, kind, sums, loss, g, dtheta, stream);
    }
    ThetaPtrs th = {};
    BetaVals bv = {};
    for (int k = 0; k < (kind == 1 ? 1 : K); ++k) {
        if (!theta[k]) return NPLDA_EINVAL;
        th.p[k] = theta[k];
        if (kind != 1) bv.b[k] = beta[k];
    }
    hipStream_t st = (hipStream_t)stream;
    const dim3 grid(1), block(kThreads);
    if (kind == 1) {
        hipLaunchKernelGGL(loss_fused_bce, grid, block, 0, st, s, t, (long long)B, th, sums, loss, g, dtheta);
    } else {
        switch (K) {
            case 1: hipLaunchKernelGGL(loss_fused_softcdet<1>, grid, block, 0, st, s, t, (long long)B, th, bv, alpha, sums, loss, g, dtheta); break;
            case 2: hipLaunchKernelGGL(loss_fused_softcdet<2>, grid, block, 0, st, s, t, (long long)B, th, bv, alpha, sums, loss, g, dtheta); break;
            case 3: hipLaunchKernelGGL(loss_fused_softcdet<3>, grid, block, 0, st, s, t, (long long)B, th, bv, alpha, sums, loss, g, dtheta); break;
            default: hipLaunchKernelGGL(loss_fused_softcdet<4>, grid, block, 0, st, s, t, (long long)B, th, bv, alpha, sums, loss, g, dtheta); break;
        }
    }
    return nplda_launch_status();
}

}  // extern "C"

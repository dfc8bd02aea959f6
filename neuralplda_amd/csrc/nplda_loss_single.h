// nplda_loss_single.h — the ONE-BLOCK form of the loss (batches of <= kSingleBlockMax pairs, 16-byte aligned): sums, loss,
// dL/dtheta and dL/ds by one block of 256 threads in a fixed order.  Shared by nplda_loss.hip (nplda_loss_fwd_bwd_f32: the
// block is the launch) and nplda_moments.hip (nplda_dplda_update_loss_f32: the block rides in the weighted-moments launch as
// an extra block, and the moments blocks form dL/ds_i inline from the same device functions) — one definition, the same bits.
#pragma once
#include "nplda_loss_math.h"

namespace nplda_loss {

constexpr int kThreads = 256;
constexpr long long kSingleBlockMax = 4096;

__device__ __forceinline__ double wave_sum_d(double v) {
#pragma unroll
    for (int m = 32; m >= 1; m >>= 1) v += __shfl_xor(v, m, 64);
    return v;
}

// One block over a whole (16-byte aligned) batch: float4 groups, four groups per thread loaded as one batch with clamped
// indices (a scalar one-load-per-iteration loop is waited out load by load), then the < 4 tail elements.
template <class F>
__device__ __forceinline__ void sums_single_block(const float* __restrict__ s, const float* __restrict__ t, long long B,
                                                  F accumulate) {
    const int nv = (int)(B / 4);
    const f32x4* s4 = reinterpret_cast<const f32x4*>(s);
    const f32x4* t4 = reinterpret_cast<const f32x4*>(t);
    for (int base = threadIdx.x; base < nv; base += kThreads * 4) {
        f32x4 sv[4], tv[4];
#pragma unroll
        for (int u = 0; u < 4; ++u) {
            const int gidx = base + kThreads * u;
            sv[u] = s4[gidx < nv ? gidx : nv - 1];
            tv[u] = t4[gidx < nv ? gidx : nv - 1];
        }
#pragma unroll
        for (int u = 0; u < 4; ++u) {
            if (base + kThreads * u < nv) {
#pragma unroll
                for (int e = 0; e < 4; ++e) accumulate(sv[u][e], tv[u][e]);
            }
        }
    }
    for (long long i = 4LL * nv + threadIdx.x; i < B; i += kThreads) accumulate(s[i], t[i]);
}

template <int NS>
__device__ __forceinline__ void block_totals(double (&acc)[NS], double* tot, double* sums_out) {
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
        tot[threadIdx.x] = v;
        if (sums_out) sums_out[threadIdx.x] = v;
    }
    __syncthreads();
}

template <class F>
__device__ __forceinline__ void grad_single_block(const float* __restrict__ s, const float* __restrict__ t,
                                                  float* __restrict__ g, long long B, F gi) {
    const int nv = (int)(B / 4);
    const f32x4* s4 = reinterpret_cast<const f32x4*>(s);
    const f32x4* t4 = reinterpret_cast<const f32x4*>(t);
    f32x4* g4 = reinterpret_cast<f32x4*>(g);
    for (int base = threadIdx.x; base < nv; base += kThreads * 4) {
        f32x4 sv[4], tv[4];
#pragma unroll
        for (int u = 0; u < 4; ++u) {
            const int gidx = base + kThreads * u;
            sv[u] = s4[gidx < nv ? gidx : nv - 1];
            tv[u] = t4[gidx < nv ? gidx : nv - 1];
        }
#pragma unroll
        for (int u = 0; u < 4; ++u) {
            const int gidx = base + kThreads * u;
            if (gidx < nv) {
                f32x4 o;
#pragma unroll
                for (int e = 0; e < 4; ++e) o[e] = gi(sv[u][e], tv[u][e]);
                g4[gidx] = o;
            }
        }
    }
    for (long long i = 4LL * nv + threadIdx.x; i < B; i += kThreads) g[i] = gi(s[i], t[i]);
}

template <int K>
__device__ __forceinline__ void loss_fused_softcdet_body(const float* __restrict__ s, const float* __restrict__ t,
                                                                long long B, ThetaPtrs th, BetaVals beta, float alpha,
                                                                double* sums, float* loss, float* __restrict__ g,
                                                                float* dtheta) {
    constexpr int NS = 2 + 4 * K;
    __shared__ double tot[NS];
    double acc[NS];
#pragma unroll
    for (int i = 0; i < NS; ++i) acc[i] = 0.0;
    float theta[K];
#pragma unroll
    for (int k = 0; k < K; ++k) theta[k] = th.p[k][0];
    sums_single_block(s, t, B, [&](float si, float ti) { softcdet_accumulate<K, false>(si, ti, theta, alpha, acc); });
    block_totals<NS>(acc, tot, sums);
    if (threadIdx.x == 0) softcdet_scalars<K>(tot, beta, alpha, loss, dtheta);
    const double Nt = tot[0], Nn = tot[1];
    float cn[K], ct;
    softcdet_consts<K>(Nt, Nn, beta, alpha, cn, ct);
    if (g) grad_single_block(s, t, g, B, [&](float si, float ti) { return softcdet_gi<K>(si, ti, theta, cn, ct, alpha); });
}

__device__ __forceinline__ void loss_fused_bce_body(const float* __restrict__ s, const float* __restrict__ t,
                                                           long long B, ThetaPtrs th, double* sums, float* loss,
                                                           float* __restrict__ g, float* dtheta) {
    __shared__ double tot[4];
    double acc[4] = {0.0, 0.0, 0.0, 0.0};
    const float theta = th.p[0][0];
    sums_single_block(s, t, B, [&](float si, float ti) { bce_accumulate(si, ti, theta, acc); });
    block_totals<4>(acc, tot, sums);
    const double N = tot[0] + tot[1];
    if (threadIdx.x == 0) bce_scalars(tot, loss, dtheta);
    const float invN = (float)(1.0 / N);
    if (g) grad_single_block(s, t, g, B, [&](float si, float ti) { return bce_gi(si, ti, theta, invN); });
}


}  // namespace nplda_loss

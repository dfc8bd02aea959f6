// nplda_loss_tail.h — the scalar tail of the one-call training step: the per-block loss partials of K-A -> the loss, dL/dtheta
// and torch.optim.Adam's update of the thresholds.  One block's work that depends on K-A only: it rides in the weight-gradient
// launch as an extra block (nplda_wgrad_fm.h) — on a CU of its own, off the critical path — or, where that kernel does not
// apply, in the last block of the update kernel (nplda_backward.hip).
#pragma once
#include "nplda_adam_math.h"
#include "nplda_bwd_loss.h"
#include "nplda_loss_math.h"

namespace nplda {

struct LossTail {
    const double* partial;    // [nblk][kLossNS] loss sums of K-A's blocks of 16 pairs
    int nblk, K, kind;
    nplda_loss::BetaVals beta;
    float alpha;
    float* theta[nplda_loss::kMaxK];
    float* loss;              // the step's loss
    double* loss_sum;         // optional fp64: loss_sum[0] += loss (interval means of the training log)
    float* m;                 // the thresholds' exp_avg / exp_avg_sq (K floats each)
    float* v;
    float* gout;              // optional: dL/dtheta (K floats)
    const float* step;        // Adam's step counter
    int bumped;               // step[0] already counts this step (train_fb_small_kernel)
    float lr, beta1, beta2, eps, wd;
    // data parallel (nplda_train_step_grad_f32 / _apply_f32): the fp64 loss sums cross the ranks inside the ONE fp32
    // all-reduce of the flat gradient, as kLossLimbs fixed-point limbs of 16 bits each (loss_limbs_of): a limb is an
    // integer below 2^16, so the fp32 sum of up to 256 ranks' limbs is EXACT, and the limbs of the sum put the fp64 sum
    // back together to 2^-41 absolute (|sum| < 2^24: the sums are counts and sums of sigmoids / log terms over the batch).
    // (A (hi, lo) float pair does not do it: the ranks' `hi`s are rounded to 24 bits again when fp32 adds them.)
    float* sums_out;          // grad phase: write the block-summed loss sums here [kLossLimbs][kLossNS] and stop
    const float* sums_in;     // apply phase: take the (all-reduced) sums from here instead of `partial`
};

constexpr int kLossLimbs = 4;
__device__ __forceinline__ void loss_limbs_of(double x, float (&l)[kLossLimbs]) {
    const double sg = x < 0.0 ? -1.0 : 1.0;
    double r = fabs(x);
    const double a = floor(r * 0x1p-8);  r -= a * 0x1p8;
    const double b = floor(r * 0x1p8);   r -= b * 0x1p-8;
    const double c = floor(r * 0x1p24);  r -= c * 0x1p-24;
    const double d = rint(r * 0x1p40);
    l[0] = (float)(sg * a); l[1] = (float)(sg * b); l[2] = (float)(sg * c); l[3] = (float)(sg * d);
}
__device__ __forceinline__ double loss_limbs_sum(const float* p, int stride) {
    return (double)p[0] * 0x1p8 + (double)p[stride] * 0x1p-8 + (double)p[2 * stride] * 0x1p-24 + (double)p[3 * stride] * 0x1p-40;
}

constexpr int kLossTailSmem = (256 * (kLossNS + 1) + kLossNS * 8 + kLossNS) * 8 + nplda_loss::kMaxK * 4;  // bytes

// Called by EVERY thread of a block of >= 256 threads (barriers inside); the first 256 do the work.
// Fixed order: 256 partial rows at a time through LDS, 8 interleaved chains per sum, then the chains in order.
__device__ __forceinline__ void loss_tail_block(const LossTail& a, void* smem) {
    double (*tile)[kLossNS + 1] = reinterpret_cast<double (*)[kLossNS + 1]>(smem);
    double (*chain)[8] = reinterpret_cast<double (*)[8]>(tile + 256);
    double* sums = reinterpret_cast<double*>(chain + kLossNS);
    float* dth = reinterpret_cast<float*>(sums + kLossNS);
    const int tid = threadIdx.x;
    const bool on = tid < 256;
    const int ns = nplda_loss::nsums(a.K, a.kind);
    const int i = tid >> 3, cc = tid & 7;
    double vv = 0.0;
    for (int base = 0; base < (a.sums_in ? 0 : a.nblk); base += 256) {
        const int b = base + tid;
        if (on) {
#pragma unroll
            for (int q = 0; q < kLossNS; ++q) tile[tid][q] = b < a.nblk ? a.partial[(size_t)b * kLossNS + q] : 0.0;
        }
        __syncthreads();
        if (on && i < ns)
            for (int r = cc; r < 256; r += 8) vv += tile[r][i];
        __syncthreads();
    }
    if (on && i < ns) chain[i][cc] = vv;
    __syncthreads();
    if (tid < ns) {
        double w = 0.0;
#pragma unroll
        for (int c8 = 0; c8 < 8; ++c8) w += chain[tid][c8];
        if (a.sums_in) w = loss_limbs_sum(a.sums_in + tid, kLossNS);
        sums[tid] = w;
        if (a.sums_out) {
            float l[kLossLimbs];
            loss_limbs_of(w, l);
#pragma unroll
            for (int q = 0; q < kLossLimbs; ++q) a.sums_out[q * kLossNS + tid] = l[q];
        }
    }
    if (a.sums_out) {  // (unused slots stay zero for the all-reduce)
        if (tid >= ns && tid < kLossNS) {
#pragma unroll
            for (int q = 0; q < kLossLimbs; ++q) a.sums_out[q * kLossNS + tid] = 0.f;
        }
        return;
    }
    __syncthreads();
    if (tid == 0) {
        if (a.kind == 1) nplda_loss::bce_scalars(sums, a.loss, dth);
        else if (a.K == 1) nplda_loss::softcdet_scalars<1>(sums, a.beta, a.alpha, a.loss, dth);
        else if (a.K == 2) nplda_loss::softcdet_scalars<2>(sums, a.beta, a.alpha, a.loss, dth);
        else if (a.K == 3) nplda_loss::softcdet_scalars<3>(sums, a.beta, a.alpha, a.loss, dth);
        else nplda_loss::softcdet_scalars<4>(sums, a.beta, a.alpha, a.loss, dth);
        if (a.loss_sum) a.loss_sum[0] += (double)a.loss[0];
    }
    __syncthreads();
    const int nth = a.kind == 1 ? 1 : a.K;
    if (tid < nth) {
        const float t = a.bumped ? a.step[0] : a.step[0] + 1.0f;
        const nplda_adam::Consts c = nplda_adam::consts_for(t, a.lr, a.beta1, a.beta2, a.eps, a.wd);
        const float g = dth[tid];
        if (a.gout) a.gout[tid] = g;
        float m = a.m[tid], v = a.v[tid];
        a.theta[tid][0] = nplda_adam::update(a.theta[tid][0], g, m, v, c);
        a.m[tid] = m;
        a.v[tid] = v;
    }
}

}  // namespace nplda

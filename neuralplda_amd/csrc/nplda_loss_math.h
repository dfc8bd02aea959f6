// nplda_loss_math.h — per-pair terms of the SoftCdet / BCE losses (utils/models.py:384-399), shared by nplda_loss.hip
// (the loss entry points) and nplda_train_step.hip (the loss folded into the backward data kernel).  One definition, so
// that the fused training step and the separate launches produce the same bits.
#pragma once
#include "nplda_common.h"

namespace nplda_loss {

constexpr int kMaxK = 4;

struct ThetaPtrs { const float* p[kMaxK]; };
struct BetaVals { float b[kMaxK]; };

// sums layout: [N_t, N_n, {S_miss_k, S_fa_k, D_t_k, D_n_k} for k < K] (SoftCdet), [N_t, N_n, sum BCE, sum (p - t)] (BCE)
__host__ __device__ inline int nsums(int K, int kind) { return kind == 1 ? 4 : 2 + 4 * K; }

// one pair's contribution to the SoftCdet sums; sigma'(v) = e / (1 + e)^2 with e = exp(-|v|) (no 1 - sigma cancellation)
template <int K, bool HARD>
__device__ __forceinline__ void softcdet_accumulate(float si, float ti, const float (&theta)[K], float alpha,
                                                    double (&acc)[2 + 4 * K]) {
    const float ni = 1.0f - ti;
    acc[0] += ti;
    acc[1] += ni;
#pragma unroll
    for (int k = 0; k < K; ++k) {
        float sm, sf, d;
        if (HARD) {
            sm = si < theta[k] ? 1.0f : 0.0f;
            sf = si > theta[k] ? 1.0f : 0.0f;
            d = 0.0f;
        } else {
            const float v = alpha * (theta[k] - si);
            const float e = __expf(-fabsf(v));
            const float inv = 1.0f / (1.0f + e);
            const float sg_pos = inv;        // sigma(|v|)
            const float sg_neg = e * inv;    // sigma(-|v|)
            sm = v >= 0.f ? sg_pos : sg_neg;  // sigma(alpha (theta - s))   (miss)
            sf = v >= 0.f ? sg_neg : sg_pos;  // sigma(alpha (s - theta))   (false alarm)
            d = e * inv * inv;                // sigma'(v)
        }
        acc[2 + 4 * k + 0] += sm * ti;
        acc[2 + 4 * k + 1] += sf * ni;
        acc[2 + 4 * k + 2] += d * ti;
        acc[2 + 4 * k + 3] += d * ni;
    }
}

// F.binary_cross_entropy(sigmoid(s - theta), t): log terms clamped at -100 (utils/models.py:390-393)
__device__ __forceinline__ void bce_accumulate(float si, float ti, float theta, double (&acc)[4]) {
    const float p = 1.0f / (1.0f + expf(-(si - theta)));
    const float lp = fmaxf(logf(p), -100.0f);
    const float lq = fmaxf(logf(1.0f - p), -100.0f);
    acc[0] += ti;
    acc[1] += 1.0f - ti;
    acc[2] += -(ti * lp + (1.0f - ti) * lq);
    acc[3] += p - ti;
}

// batch constants of dL/ds: cn[k] = beta_k alpha / (N_n K), ct = -alpha / (N_t K)
template <int K>
__device__ __forceinline__ void softcdet_consts(double Nt, double Nn, const BetaVals& beta, float alpha, float (&cn)[K],
                                                float& ct) {
#pragma unroll
    for (int k = 0; k < K; ++k) cn[k] = (float)((double)beta.b[k] * alpha / (Nn * K));
    ct = (float)(-(double)alpha / (Nt * K));
}

// dL/ds_i of SoftCdet given the batch constants
template <int K>
__device__ __forceinline__ float softcdet_gi(float si, float ti, const float (&theta)[K], const float (&cn)[K], float ct,
                                             float alpha) {
    const float ni = 1.0f - ti;
    float gi = 0.f;
#pragma unroll
    for (int k = 0; k < K; ++k) {
        const float v = alpha * (theta[k] - si);
        const float e = __expf(-fabsf(v));
        const float inv = 1.0f / (1.0f + e);
        const float d = e * inv * inv;
        gi = fmaf(d, fmaf(ct, ti, cn[k] * ni), gi);  // explicit contraction: the same bits wherever this is inlined
    }
    return gi;
}

__device__ __forceinline__ float bce_gi(float si, float ti, float theta, float invN) {
    const float p = 1.0f / (1.0f + expf(-(si - theta)));
    return (p - ti) * invN;
}

// loss and dL/dtheta from the sums (one thread)
template <int K>
__device__ __forceinline__ void softcdet_scalars(const double* sums, const BetaVals& beta, float alpha, float* loss,
                                                 float* dtheta) {
    const double Nt = sums[0], Nn = sums[1];
    double L = 0.0;
#pragma unroll
    for (int k = 0; k < K; ++k) {
        L += sums[2 + 4 * k] / Nt + (double)beta.b[k] * sums[2 + 4 * k + 1] / Nn;
        if (dtheta)
            dtheta[k] = (float)(((double)alpha * sums[2 + 4 * k + 2] / Nt -
                                 (double)beta.b[k] * alpha * sums[2 + 4 * k + 3] / Nn) / K);
    }
    if (loss) loss[0] = (float)(L / K);
}

__device__ __forceinline__ void bce_scalars(const double* sums, float* loss, float* dtheta) {
    const double N = sums[0] + sums[1];
    if (loss) loss[0] = (float)(sums[2] / N);
    if (dtheta) dtheta[0] = (float)(-sums[3] / N);
}

}  // namespace nplda_loss

// nplda_bwd_loss.h — the loss as the data-gradient kernels see it (nplda_backward.hip, nplda_train_fb_small.h).
#pragma once
#include "nplda_loss_math.h"

namespace nplda {

// Loss folded into the data-gradient kernel (the fused training step, nplda_train_step_f32): dL/ds of a pair depends on
// the pair's own score and target and on the batch counts N_t, N_n only (utils/models.py:384-399), so the kernel that
// needs g forms it itself and leaves the batch sums of the loss as one fp64 partial per block.
constexpr int kLossNS = 2 + 4 * nplda_loss::kMaxK;
struct BwdLoss {
    const float* s;       // (B) scores of this step's forward
    const float* t;       // (B) targets, 16-byte aligned
    nplda_loss::ThetaPtrs th;
    nplda_loss::BetaVals beta;
    int K, kind;          // kind 0 = SoftCdet, 1 = BCE
    float alpha;
    long long B;
    float* g_out;         // (B) dL/ds, for the weight-gradient kernel's dQ / dP sums
    double* partial;      // [blocks][kLossNS] loss sums of the block's 16 pairs
    const double* gcount; // optional: the GLOBAL batch's [N_t, N_n] (data parallel: dL/ds_i needs the pair's own score and
                          // target and these two counts only, so a rank given them forms its shard's gradient with no
                          // collective in front — train_fb_small_kernel; null: counted from `t`)
};


// What a pair's loss step needs besides its score and target, in three parts so that a kernel can place each where it
// is free: the thresholds (global loads — early), the batch constants of dL/ds (fp64 divisions — once the counts are
// known), the pair itself.
// (theta / cn are VECTOR values, not arrays: the K = 1 .. 4 branches of loss_consts_counts write cn[0 .. K - 1], hipcc merged
// their tails into one store at a run-time offset and with that the struct lived in scratch — a scratch store, four scratch
// loads and an s_waitcnt vmcnt(0) in front of the counts' way to LDS, in every training kernel; round 6.)
static_assert(nplda_loss::kMaxK == 4, "PairLossConsts holds the thresholds' constants in float4 values");
struct PairLossConsts {
    f32x4 theta;
    f32x4 cn;   // SoftCdet: beta_k alpha / (N_n K);  BCE: cn[0] = 1 / N
    float ct;   // SoftCdet: -alpha / (N_t K)
};

__device__ __forceinline__ void loss_consts_theta(const BwdLoss& L, PairLossConsts& c) {
    const int nth = L.kind == 1 ? 1 : L.K;
#pragma unroll
    for (int k = 0; k < nplda_loss::kMaxK; ++k) {
        c.theta[k] = (k < nth ? L.th.p[k] : L.th.p[0])[0];  // (static indices: a dynamic one copies the pointer array to scratch)
        c.cn[k] = 0.f;
    }
    c.ct = 0.f;
}

template <int K>
__device__ __forceinline__ void loss_consts_counts_k(const BwdLoss& L, double Nt, double Nn, PairLossConsts& c) {
    float cn[K];
    nplda_loss::softcdet_consts<K>(Nt, Nn, L.beta, L.alpha, cn, c.ct);
#pragma unroll
    for (int k = 0; k < K; ++k) c.cn[k] = cn[k];
}

__device__ __forceinline__ void loss_consts_counts(const BwdLoss& L, double Nt, double Nn, PairLossConsts& c) {
    if (L.kind == 1) c.cn[0] = (float)(1.0 / (Nt + Nn));
    else if (L.K == 1) loss_consts_counts_k<1>(L, Nt, Nn, c);
    else if (L.K == 2) loss_consts_counts_k<2>(L, Nt, Nn, c);
    else if (L.K == 3) loss_consts_counts_k<3>(L, Nt, Nn, c);
    else loss_consts_counts_k<4>(L, Nt, Nn, c);
}

// g_i and the pair's contribution to the loss sums (SoftCdet with K thresholds)
template <int K>
__device__ __forceinline__ float loss_pair_softcdet(const BwdLoss& L, const PairLossConsts& c, float si, float ti,
                                                    double (&acc)[kLossNS]) {
    float theta[K], cn[K];
#pragma unroll
    for (int k = 0; k < K; ++k) {
        theta[k] = c.theta[k];
        cn[k] = c.cn[k];
    }
    double a2[2 + 4 * K];
#pragma unroll
    for (int i = 0; i < 2 + 4 * K; ++i) a2[i] = 0.0;
    nplda_loss::softcdet_accumulate<K, false>(si, ti, theta, L.alpha, a2);
#pragma unroll
    for (int i = 0; i < 2 + 4 * K; ++i) acc[i] = a2[i];
    return nplda_loss::softcdet_gi<K>(si, ti, theta, cn, c.ct, L.alpha);
}

// The per-pair loss step shared by the kernels: returns g_i = dL/ds_i, fills acc with the pair's terms of the loss sums.
__device__ __forceinline__ float loss_pair(const BwdLoss& L, const PairLossConsts& c, float si, float ti,
                                           double (&acc)[kLossNS]) {
#pragma unroll
    for (int i = 0; i < kLossNS; ++i) acc[i] = 0.0;
    if (L.kind == 1) {
        double a4[4] = {0.0, 0.0, 0.0, 0.0};
        nplda_loss::bce_accumulate(si, ti, c.theta[0], a4);
#pragma unroll
        for (int i = 0; i < 4; ++i) acc[i] = a4[i];
        return nplda_loss::bce_gi(si, ti, c.theta[0], c.cn[0]);
    }
    if (L.K == 1) return loss_pair_softcdet<1>(L, c, si, ti, acc);
    if (L.K == 2) return loss_pair_softcdet<2>(L, c, si, ti, acc);
    if (L.K == 3) return loss_pair_softcdet<3>(L, c, si, ti, acc);
    return loss_pair_softcdet<4>(L, c, si, ti, acc);
}

// all three parts at once
__device__ __forceinline__ float loss_pair(const BwdLoss& L, double Nt, double Nn, float si, float ti, double (&acc)[kLossNS]) {
    PairLossConsts c;
    loss_consts_theta(L, c);
    loss_consts_counts(L, Nt, Nn, c);
    return loss_pair(L, c, si, ti, acc);
}

// The element formulas of the data gradients with every contraction spelled out, shared by the kernels that must agree
// bit for bit (left to fp-contract=fast, `q z + p z'` fuses differently depending on where z comes from).
__device__ __forceinline__ f32x4 dz_of(float tg, f32x4 q, f32x4 p, f32x4 z, f32x4 zo) {  // 2 g (Q z + P z')
    f32x4 r;
#pragma unroll
    for (int c = 0; c < 4; ++c) r[c] = tg * fmaf(q[c], z[c], p[c] * zo[c]);
    return r;
}
__device__ __forceinline__ f32x4 du_of(f32x4 dy, f32x4 y, float dot, float rn) {  // (dy - y (y . dy)) / max(||u||, eps)
    f32x4 r;
#pragma unroll
    for (int c = 0; c < 4; ++c) r[c] = fmaf(-y[c], dot, dy[c]) * rn;
    return r;
}
__device__ __forceinline__ void pair_sum_terms(float gh, f32x4 z1, f32x4 z2, f32x4& eq, f32x4& ep) {  // g (z1^2 + z2^2), g z1 z2
#pragma unroll
    for (int c = 0; c < 4; ++c) {
        eq[c] = gh * fmaf(z1[c], z1[c], z2[c] * z2[c]);
        ep[c] = gh * (z1[c] * z2[c]);
    }
}

// N_t of the batch, summed by the whole 256-thread block (<= 16 float4 per thread out of L2); cnt_s: 4 floats of LDS.
// Contains one __syncthreads().
__device__ __forceinline__ double block_target_count(const BwdLoss& L, float* cnt_s) {
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    float cnt = 0.f;
    const int nv = (int)(L.B / 4);
    const f32x4* t4 = reinterpret_cast<const f32x4*>(L.t);
    for (int i = tid; i < nv; i += 256) {
        const f32x4 v = t4[i];
        cnt += (v[0] + v[1]) + (v[2] + v[3]);
    }
    for (long long i = 4LL * nv + tid; i < L.B; i += 256) cnt += L.t[i];
#pragma unroll
    for (int msk = 1; msk < 64; msk <<= 1) cnt = wave_xor_add(cnt, msk);
    if (lane == 0) cnt_s[wave] = cnt;
    __syncthreads();
    return (double)((cnt_s[0] + cnt_s[1]) + (cnt_s[2] + cnt_s[3]));  // exact: a count below 2^24
}

// The same count in two parts, for a kernel that has better things to do while the targets are on their way
// (train_fb_small_kernel): target_count_issue only LOADS — the first 4096 targets, four float4 per thread, the whole batch
// at the recipe sizes — and target_count_wave, called once those have long landed, sums them (and synchronously whatever
// a larger batch has left) over the wave.  The caller adds the four wave sums (a count below 2^24 is exact in any order).
// (64 lanes of ds_add_f32 on one LDS word instead of the cross-lane steps: 1.5 us slower.)
struct TargetEarly { f32x4 v[4]; };

__device__ __forceinline__ void target_count_issue(const BwdLoss& L, TargetEarly& e) {
    const int nv = (int)(L.B / 4);
    const f32x4* t4 = reinterpret_cast<const f32x4*>(L.t);
#pragma unroll
    for (int q = 0; q < 4; ++q) {
        const int i = (int)threadIdx.x + 256 * q;
        e.v[q] = t4[i < nv ? i : 0];  // (B >= 4 rows are always readable: the tail below covers B < 4)
    }
}

__device__ __forceinline__ float target_count_wave(const BwdLoss& L, const TargetEarly& e) {
    const int tid = threadIdx.x;
    const int nv = (int)(L.B / 4);
    const f32x4* t4 = reinterpret_cast<const f32x4*>(L.t);
    float cnt = 0.f;
#pragma unroll
    for (int q = 0; q < 4; ++q) {
        const f32x4 v = e.v[q];
        cnt += tid + 256 * q < nv ? (v[0] + v[1]) + (v[2] + v[3]) : 0.f;
    }
    for (int i = tid + 1024; i < nv; i += 256) {
        const f32x4 v = t4[i];
        cnt += (v[0] + v[1]) + (v[2] + v[3]);
    }
    for (long long i = 4LL * nv + tid; i < L.B; i += 256) cnt += L.t[i];
    cnt = row16_sum(cnt);
    cnt = wave_xor_add(cnt, 16);
    return wave_xor_add(cnt, 32);
}

}  // namespace nplda

// nplda_cohort_qz.h — first moments of the cohort for the fused AS-norm path (nplda_cohort_fused.hip), as a block-level
// device function: the blocks that run it ride in the launch of the cohort's Gram matrix (wgrad_kernel in
// nplda_backward.hip takes them as extra work items) — the two read the same table, do not depend on each other, and a
// launch of their own cost 10 us of a 0.8 ms call.
#pragma once
#include "nplda_common.h"

namespace nplda {

struct QzArgs {
    const float* zc;   // (M, ldz) cohort embeddings
    const float* qc;   // (M) self terms
    long long M, ldz;
    int Mp, nblocks;   // padded row width; blocks sharing the rows
    float* part;       // [nblocks][Mp + 2]: sum q_m z_m, sum q, sum q^2 (fp32: they only feed the threshold proposal)
    double* part64;    // [nblocks][Mp + 1]: sum z_m, sum q in fp64 — the row means the call RETURNS are formed analytically
                       // from these (mean_r = q_r + mean q + 2 (P z_r) . mean z)
};

constexpr size_t kQzSmemBytes = 4 * (NPLDA_MAX_DIM + 2) * sizeof(float) + 4 * (NPLDA_MAX_DIM + 1) * sizeof(double);

// block `b` of a.nblocks (256 threads); smem: kQzSmemBytes, 8-byte aligned
__device__ __forceinline__ void cohort_qz_block(const QzArgs& a, int b, void* smem) {
    double (*red64)[NPLDA_MAX_DIM + 1] = reinterpret_cast<double (*)[NPLDA_MAX_DIM + 1]>(smem);
    float (*red)[NPLDA_MAX_DIM + 2] = reinterpret_cast<float (*)[NPLDA_MAX_DIM + 2]>(red64 + 4);
    const int Mp = a.Mp;
    const int lane = threadIdx.x & 63, wy = threadIdx.x >> 6;
    const long long per = (a.M + a.nblocks - 1) / a.nblocks;
    const long long lo = (long long)b * per, hi = lo + per < a.M ? lo + per : a.M;
    float acc[3] = {0.f, 0.f, 0.f}, sq = 0.f, sqq = 0.f;
    double sz[3] = {0.0, 0.0, 0.0}, sq64 = 0.0;
    // four rows in flight per wave (independent loads), accumulated in row order
    for (long long m0 = lo + wy; m0 < hi; m0 += 16) {
        float q[4], z[4][3];
#pragma unroll
        for (int u = 0; u < 4; ++u) {
            const long long m = m0 + 4 * u;
            const long long mc = m < hi ? m : hi - 1;
            q[u] = m < hi ? a.qc[mc] : 0.f;
#pragma unroll
            for (int c = 0; c < 3; ++c) {
                const int f = lane + 64 * c;
                z[u][c] = (f < Mp && m < hi) ? a.zc[mc * a.ldz + f] : 0.f;
            }
        }
#pragma unroll
        for (int u = 0; u < 4; ++u) {
#pragma unroll
            for (int c = 0; c < 3; ++c) {
                acc[c] = fmaf(q[u], z[u][c], acc[c]);
                sz[c] += (double)z[u][c];
            }
            sq += q[u];
            sq64 += (double)q[u];
            sqq = fmaf(q[u], q[u], sqq);
        }
    }
#pragma unroll
    for (int c = 0; c < 3; ++c) {
        if (lane + 64 * c < Mp) {
            red[wy][lane + 64 * c] = acc[c];
            red64[wy][lane + 64 * c] = sz[c];
        }
    }
    if (lane == 0) {  // (every lane of a wave holds the same q sums)
        red[wy][Mp] = sq;
        red[wy][Mp + 1] = sqq;
        red64[wy][Mp] = sq64;
    }
    __syncthreads();
    for (int i = threadIdx.x; i < Mp + 2; i += 256)
        a.part[(size_t)b * (Mp + 2) + i] = ((red[0][i] + red[1][i]) + red[2][i]) + red[3][i];
    for (int i = threadIdx.x; i < Mp + 1; i += 256)
        a.part64[(size_t)b * (Mp + 1) + i] = ((red64[0][i] + red64[1][i]) + red64[2][i]) + red64[3][i];
}

}  // namespace nplda

// nplda_indexed.hip — the HBM/L2-bound half of the scoring path (gfx950): gather + diagonal score.
//
//  * nplda_gather_rows_f32     replaces the per-pair Python dict look-ups of
//                              utils/sv_trials_loaders.py:418-426 / :429-437 (1.8e4 pairs/s on the host)
//                              by a device index-select from a resident (N_utt, D0) x-vector matrix.
//  * nplda_score_indexed_f32   utils/models.py:372-376 on rows gathered from a pre-embedded table:
//                              s = q[i1] + q[i2] + 2 sum_d P_d z[i1,d] z[i2,d]   (q from nplda_embed_f32)
//                              Algorithmic bytes/pair = 2*4*D2 + 2*4 + 2*8 + 4 (1 228 B at D2 = 150).
//  * nplda_score_embeddings_f32  the same formula on two explicit (B, D2) tensors
//                              (NeuralPlda.forward_from_plda_embeddings).
// These are pure streaming kernels: float4 row loads, 8 lanes per pair (128 B contiguous per load),
// three DPP/shuffle adds per pair, nothing staged through LDS (there is no reuse to exploit).
#include "nplda_common.h"

namespace {

constexpr int kThreads = 256;
constexpr int kLPP = 8;  // lanes per pair

// 8 lanes per pair, KM float4 columns per lane (sub + 8 k).  What round 5 changed, measured on the 1.2 M-row table
// (profiles/r05r_indexed.txt):
//  * a pair's three dependent round trips (its indices -> its rows -> q[i1], q[i2] by lane 0 at the very end) are one now: the
//    NEXT pair's indices are loaded under this pair's rows, the self terms come with the rows — a wave had its row bytes in
//    flight for a third of an iteration;
//  * SELF: q[i] = sum_d Q_d z[i, d]^2 is formed from the row the lane holds anyway instead of being gathered — two scattered
//    4-byte reads (a 64-byte sector each) per 1 228-byte pair less;
//  * 2 P and Q are loop invariant: in registers, not re-read per pair; the 8-lane sum on the DPP path (three ds_bpermute
//    round trips per pair before).
template <bool SELF, int KM>
__global__ __launch_bounds__(kThreads) void score_indexed_kernel(const float* __restrict__ z, long long ldz,
                                                                 const float* __restrict__ q, long long N,
                                                                 const long long* __restrict__ i1,
                                                                 const long long* __restrict__ i2, long long B,
                                                                 const float* __restrict__ P, const float* __restrict__ Q,
                                                                 int ncol4, float* __restrict__ s) {
    const int sub = threadIdx.x & (kLPP - 1);
    const long long stride = (long long)gridDim.x * (kThreads / kLPP);
    long long p = (long long)blockIdx.x * (kThreads / kLPP) + threadIdx.x / kLPP;
    int col[KM];
    f32x4 p2[KM], qq[KM];
#pragma unroll
    for (int k = 0; k < KM; ++k) {
        const int c = sub + kLPP * k;
        const bool valid = c < ncol4;
        col[k] = valid ? c : ncol4 - 1;  // (always-valid addresses; the weights of a column past the row are zero)
        const f32x4 pv = reinterpret_cast<const f32x4*>(P)[col[k]], qv = reinterpret_cast<const f32x4*>(Q)[col[k]];
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            p2[k][r] = valid ? 2.0f * pv[r] : 0.f;
            qq[k][r] = valid && SELF ? qv[r] : 0.f;
        }
    }
    long long a = -1, b = -1;
    if (p < B) {
        a = i1[p];
        b = i2[p];
    }
    while (p < B) {
        const long long pn = p + stride;
        long long an = -1, bn = -1;
        if (pn < B) {  // the next pair's indices travel under this pair's rows
            an = i1[pn];
            bn = i2[pn];
        }
        const bool ok = a >= 0 && a < N && b >= 0 && b < N;
        const long long ac = ok ? a : 0, bc = ok ? b : 0;
        const f32x4* za = reinterpret_cast<const f32x4*>(z + ac * ldz);
        const f32x4* zb = reinterpret_cast<const f32x4*>(z + bc * ldz);
        float acc = 0.f;
        if (!SELF) acc = sub == 0 ? q[ac] : (sub == 1 ? q[bc] : 0.f);
        f32x4 va[KM], vb[KM];
#pragma unroll
        for (int k = 0; k < KM; ++k) {
            va[k] = za[col[k]];
            vb[k] = zb[col[k]];
        }
#pragma unroll
        for (int k = 0; k < KM; ++k)
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                acc = fmaf(p2[k][r] * va[k][r], vb[k][r], acc);
                if (SELF) acc = fmaf(qq[k][r], fmaf(va[k][r], va[k][r], vb[k][r] * vb[k][r]), acc);
            }
        acc += dpp_f32<0xB1>(acc);   // quad_perm [1, 0, 3, 2]
        acc += dpp_f32<0x4E>(acc);   // quad_perm [2, 3, 0, 1]
        acc += dpp_f32<0x141>(acc);  // row_half_mirror: the other quad of the pair's eight lanes
        if (sub == 0) s[p] = ok ? acc : __builtin_nanf("");
        p = pn;
        a = an;
        b = bn;
    }
}

// 16 lanes per pair, scalar (any-stride, any-alignment) loads.
__global__ __launch_bounds__(kThreads) void score_embeddings_kernel(const float* __restrict__ z1, long long ld1,
                                                                    const float* __restrict__ z2, long long ld2,
                                                                    long long B, int D2,
                                                                    const float* __restrict__ P_sqrt,
                                                                    const float* __restrict__ Q,
                                                                    float* __restrict__ s) {
    const int sub = threadIdx.x & 15;
    const long long stride = (long long)gridDim.x * (kThreads / 16);
    for (long long p = (long long)blockIdx.x * (kThreads / 16) + threadIdx.x / 16; p < B; p += stride) {
        const float* a = z1 + p * ld1;
        const float* b = z2 + p * ld2;
        float acc = 0.f;
        for (int d = sub; d < D2; d += 16) {
            const float va = a[d], vb = b[d], ps = P_sqrt[d];
            acc = fmaf(Q[d], fmaf(va, va, vb * vb), acc);
            acc = fmaf(2.0f * ps * ps, va * vb, acc);
        }
        acc += __shfl_xor(acc, 1, 64);
        acc += __shfl_xor(acc, 2, 64);
        acc += __shfl_xor(acc, 4, 64);
        acc += __shfl_xor(acc, 8, 64);
        if (sub == 0) s[p] = acc;
    }
}

// One wave per gathered row: D0/4 float4 per row, lanes stride the row.
__global__ __launch_bounds__(kThreads) void gather_rows_kernel(const float* __restrict__ table, long long ldt,
                                                               long long N, const long long* __restrict__ idx,
                                                               long long B, int ncol4, float* __restrict__ out,
                                                               long long ldo) {
    const int lane = threadIdx.x & 63;
    const long long stride = (long long)gridDim.x * (kThreads / 64);
    for (long long r = (long long)blockIdx.x * (kThreads / 64) + threadIdx.x / 64; r < B; r += stride) {
        const long long i = idx[r];
        const bool ok = i >= 0 && i < N;
        const f32x4* src = reinterpret_cast<const f32x4*>(table + (ok ? i : 0) * ldt);
        f32x4* dst = reinterpret_cast<f32x4*>(out + r * ldo);
        for (int c = lane; c < ncol4; c += 64) {
            f32x4 v = src[c];
            if (!ok) v = f32x4{__builtin_nanf(""), __builtin_nanf(""), __builtin_nanf(""), __builtin_nanf("")};
            __builtin_nontemporal_store(v, dst + c);
        }
    }
}

// load_xvec_trials_from_numbatch in one launch: both sides' rows through the trial-number -> table-row map.  Row r of the
// 2 B outputs (x1 rows, then x2 rows); a number outside the map or mapped to no row gives a NaN row and raises the flag.
__global__ __launch_bounds__(kThreads) void gather_pairs_mapped_kernel(const float* __restrict__ table, long long ldt,
                                                                       long long N, const long long* __restrict__ map,
                                                                       long long nmap, const long long* __restrict__ num1,
                                                                       const long long* __restrict__ num2, long long B,
                                                                       int ncol4, float* __restrict__ out1,
                                                                       float* __restrict__ out2, long long ldo,
                                                                       int* __restrict__ bad) {
    const int lane = threadIdx.x & 63;
    const long long stride = (long long)gridDim.x * (kThreads / 64);
    for (long long r = (long long)blockIdx.x * (kThreads / 64) + threadIdx.x / 64; r < 2 * B; r += stride) {
        const bool second = r >= B;
        const long long p = second ? r - B : r;
        const long long num = second ? num2[p] : num1[p];
        const bool in_map = num >= 0 && num < nmap;
        const long long i = in_map ? map[num] : -1;
        const bool ok = i >= 0 && i < N;
        // (system scope: the word may live in pinned host memory, where the caller reads it without a copy)
        if (!ok && lane == 0) __hip_atomic_fetch_or(bad, in_map ? 2 : 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
        const f32x4* src = reinterpret_cast<const f32x4*>(table + (ok ? i : 0) * ldt);
        f32x4* dst = reinterpret_cast<f32x4*>((second ? out2 : out1) + p * ldo);
        for (int c = lane; c < ncol4; c += 64) {
            f32x4 v = src[c];
            if (!ok) v = f32x4{__builtin_nanf(""), __builtin_nanf(""), __builtin_nanf(""), __builtin_nanf("")};
            dst[c] = v;  // (plain stores: the rows are read back at once by the forward kernel, out of L2)
        }
    }
}

unsigned grid_for(long long items, int per_block) {
    long long b = (items + per_block - 1) / per_block;
    if (b < 1) b = 1;
    if (b > 256 * 32) b = 256 * 32;
    return (unsigned)b;
}

}  // namespace

extern "C" {

int nplda_score_indexed_f32(const float* z, int64_t ldz, const float* q, int64_t N, const int64_t* i1,
                            const int64_t* i2, int64_t B, const void* packed, int D0, int D1, int D2, float* s,
                            nplda_stream_t stream) {
    if (B < 0 || N < 0) return NPLDA_EINVAL;
    if (D0 <= 0 || D1 <= 0 || D2 <= 0 || (D0 % 4) != 0) return NPLDA_EINVAL;
    if (!nplda_dims_ok(D0, D1, D2)) return NPLDA_EUNSUPPORTED;
    if (B == 0) return NPLDA_OK;
    const NpldaLayout L = nplda_layout(D0, D1, D2);
    if (!z || !i1 || !i2 || !packed || !s) return NPLDA_EINVAL;  // (q may be null: the self terms then come from z)
    if (ldz < 16 * L.NB || (ldz % 4) != 0 || !nplda_aligned16(z) || !nplda_aligned16(packed)) return NPLDA_EINVAL;
    const float* P = (const float*)packed + L.oP;
    const float* Q = (const float*)packed + L.oQ;
    const int ncol4 = (D2 + 3) / 4, km = (ncol4 + kLPP - 1) / kLPP;
    const dim3 grid(grid_for(B, kThreads / kLPP)), block(kThreads);
#define NPLDA_LAUNCH(SELFV, KMV)                                                                                        \
    hipLaunchKernelGGL((score_indexed_kernel<SELFV, KMV>), grid, block, 0, (hipStream_t)stream, z, (long long)ldz, q, \
                       (long long)N, (const long long*)i1, (const long long*)i2, (long long)B, P, Q, ncol4, s)
#define NPLDA_PICK(SELFV)                        \
    switch (km) {                                \
        case 1: NPLDA_LAUNCH(SELFV, 1); break;   \
        case 2: NPLDA_LAUNCH(SELFV, 2); break;   \
        case 3: NPLDA_LAUNCH(SELFV, 3); break;   \
        case 4: NPLDA_LAUNCH(SELFV, 4); break;   \
        case 5: NPLDA_LAUNCH(SELFV, 5); break;   \
        case 6: NPLDA_LAUNCH(SELFV, 6); break;   \
        default: return NPLDA_EUNSUPPORTED;      \
    }
    if (q == nullptr) { NPLDA_PICK(true) } else { NPLDA_PICK(false) }
#undef NPLDA_PICK
#undef NPLDA_LAUNCH
    return nplda_launch_status();
}

int nplda_score_embeddings_f32(const float* z1, int64_t ld1, const float* z2, int64_t ld2, int64_t B, int D2,
                               const float* P_sqrt, const float* Q, float* s, nplda_stream_t stream) {
    if (B < 0 || D2 <= 0) return NPLDA_EINVAL;
    if (B == 0) return NPLDA_OK;
    if (!z1 || !z2 || !P_sqrt || !Q || !s || ld1 < D2 || ld2 < D2) return NPLDA_EINVAL;
    hipLaunchKernelGGL(score_embeddings_kernel, dim3(grid_for(B, kThreads / 16)), dim3(kThreads), 0,
                       (hipStream_t)stream, z1, (long long)ld1, z2, (long long)ld2, (long long)B, D2, P_sqrt, Q, s);
    return nplda_launch_status();
}

int nplda_gather_rows_f32(const float* table, int64_t ldt, int64_t N, const int64_t* idx, int64_t B, int D0,
                          float* out, int64_t ldo, nplda_stream_t stream) {
    if (B < 0 || N < 0 || D0 <= 0 || (D0 % 4) != 0) return NPLDA_EINVAL;
    if (B == 0) return NPLDA_OK;
    if (!table || !idx || !out || N == 0) return NPLDA_EINVAL;
    if (ldt < D0 || ldo < D0 || (ldt % 4) != 0 || (ldo % 4) != 0 || !nplda_aligned16(table) || !nplda_aligned16(out))
        return NPLDA_EINVAL;
    hipLaunchKernelGGL(gather_rows_kernel, dim3(grid_for(B, kThreads / 64)), dim3(kThreads), 0, (hipStream_t)stream,
                       table, (long long)ldt, (long long)N, (const long long*)idx, (long long)B, D0 / 4, out,
                       (long long)ldo);
    return nplda_launch_status();
}

int nplda_gather_pairs_mapped_f32(const float* table, int64_t ldt, int64_t N, const int64_t* map, int64_t nmap,
                                  const int64_t* num1, const int64_t* num2, int64_t B, int D0, float* out1, float* out2,
                                  int64_t ldo, int32_t* bad, nplda_stream_t stream) {
    if (B < 0 || N < 0 || nmap < 0 || D0 <= 0 || (D0 % 4) != 0) return NPLDA_EINVAL;
    if (B == 0) return NPLDA_OK;
    if (!table || !map || !num1 || !num2 || !out1 || !out2 || !bad || N == 0) return NPLDA_EINVAL;
    if (ldt < D0 || ldo < D0 || (ldt % 4) != 0 || (ldo % 4) != 0 || !nplda_aligned16(table) || !nplda_aligned16(out1) ||
        !nplda_aligned16(out2))
        return NPLDA_EINVAL;
    hipLaunchKernelGGL(gather_pairs_mapped_kernel, dim3(grid_for(2 * B, kThreads / 64)), dim3(kThreads), 0,
                       (hipStream_t)stream, table, (long long)ldt, (long long)N, (const long long*)map, (long long)nmap,
                       (const long long*)num1, (const long long*)num2, (long long)B, D0 / 4, out1, out2, (long long)ldo,
                       (int*)bad);
    return nplda_launch_status();
}

}  // extern "C"

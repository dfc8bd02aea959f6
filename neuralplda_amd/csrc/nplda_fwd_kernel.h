// nplda_fwd_kernel.h — the fused Neural-PLDA forward kernel template (gfx950, fp32 MFMA).
// Included by nplda_forward.hip (product instantiations) and by tools/exp_fwd.hip (A/B harness).
//
// Replaces utils/models.py:366-382 of the reference (two nn.Linear, F.normalize, the diagonal
// quadratic score: 25 ATen launches with every intermediate round-tripping through memory) by
// ONE kernel that reads x once from HBM and writes only s (pair mode) or z,q (embed mode).
//
// Design (not a translation of anything in the reference, which has no device code):
//  * everything is computed TRANSPOSED:  u^T = W1 x^T,  z^T = W2 y^T.  The weight matrix is the
//    MFMA A operand (16 features x 4 k) and the data are the B operand (4 k x 16 batch rows), so
//    the accumulator of v_mfma_f32_16x16x4_f32 holds, in lane (j = lane&15, g = lane>>4), the
//    features {16 nb + 4 g + r} of batch row j.  That is exactly the B-operand shape of the NEXT
//    GEMM with the k-permutation "step (kb, r) covers k = 16 kb + 4 g + r" — matched by how the
//    weights were packed — so the normalised layer-1 output feeds layer 2 straight from
//    registers: no LDS round trip, no layout change.
//  * one wave owns 16 trial pairs: group A = the 16 x1 rows, group B = the 16 x2 rows of the SAME
//    pairs.  z1 and z2 then sit in identical lanes/registers and the score is an elementwise
//    epilogue + two cross-lane adds.  (Embed mode: groups A/B are 32 consecutive rows.)
//  * x is streamed HBM -> VGPR as one float4 per lane per group per k16-step (each x element is
//    used by exactly one wave, so LDS staging would only add traffic); the weights (0.4-0.5 MB,
//    L2-resident) are streamed L2 -> LDS in fragment order, KPB k16-steps per barrier,
//    double-buffered, and read back as conflict-free linear ds_read_b128.
//  * fp32-input MFMA is exact fp32 (a k-ordered fmaf chain), so results match the reference's
//    fp32 GEMMs to rounding: tolerance |ds| <= 2e-5 + 1e-5 |s| (tests/).
//
// Roofline: MFMA-bound.  Padded work per pair = 2 * 2 * (16 KS1 + 16 NB) * (16 NB) FLOP
// (430 080 at 512->150->150 vs 398 400 algorithmic); fp32 MFMA peak 157.3 TFLOP/s.
#pragma once
#include "nplda_common.h"

namespace nplda {

enum { MODE_PAIR = 0, MODE_EMBED = 1, MODE_TRAIN = 2, MODE_GB = 3 };

// Loads are UNCONDITIONAL (out-of-range slots re-read the chunk's last float4; the packed image carries a chunk
// of slack): a load under a runtime predicate becomes a branch, and hipcc then falls back to s_waitcnt vmcnt(0)
// at the next consumer, which serialises the weight stream behind the in-flight x prefetches.
template <int CH, int THREADS, int NSLOT>
__device__ __forceinline__ void chunk_load(const f32x4* __restrict__ src, long long /*avail*/,
                                           f32x4 (&st)[NSLOT], int tid) {
#pragma unroll
    for (int i = 0; i < NSLOT; ++i) {
        const int idx = tid + THREADS * i;
        st[i] = src[idx < CH ? idx : CH - 1];
    }
}

template <int CH, int THREADS, int NSLOT>
__device__ __forceinline__ void chunk_store(f32x4* dst, const f32x4 (&st)[NSLOT], int tid) {
#pragma unroll
    for (int i = 0; i < NSLOT; ++i) {
        const int idx = tid + THREADS * i;
        // The staged value is "used" here, outside the predicate: with its only use inside `if (idx < CH)` hipcc SINKS the
        // (unconditional) load of the last, partly filled slot into that branch — load, s_waitcnt vmcnt(0), ds_write in front of
        // every stage's barrier, an exposed L2 round trip per k16-step (round 6: the GaussianBackend forward's ISA).
        f32x4 v = st[i];
        asm volatile("" : "+v"(v));
        if (idx < CH) dst[idx] = v;
    }
}

template <bool NT>
__device__ __forceinline__ f32x4 load_x4(const float* p, bool ok) {
    f32x4 v = {0.f, 0.f, 0.f, 0.f};
    if (ok) {
        if (NT) v = __builtin_nontemporal_load(reinterpret_cast<const f32x4*>(p));
        else v = *reinterpret_cast<const f32x4*>(p);
    }
    return v;
}

// Branch-free x load: the float4 at column min(koff, D0 - 4) of the row that starts at `row` (koff % 4 == 0,
// D0 % 4 == 0), returned as loaded — no predicate, no select, no zero fill.
//  * Columns >= D0 (the padded tail of the last k16-step, or a prefetch past the last chunk) read the row's LAST float4
//    instead; that is sound because every consumer multiplies these values into weight fragments that are zero there
//    (the packed image pads W1's k range with zeros), and the stand-in comes from the same input row, so a non-finite
//    value can only reach the score of the row it belongs to.
//  * Why not `ok ? v : 0`: hipcc schedules such a select at the end of the chunk that ISSUED the load (the selected
//    value is what the loop carries), behind `s_waitcnt vmcnt(0)` — every chunk then ends by waiting out the full HBM
//    latency of the x prefetch it has just issued (16 % of the kernel; found in the ISA in front of s_barrier,
//    profiles/r01s_*).  Why not `ok ? p : psafe` either: hipcc may turn the address select into two predicated loads,
//    i.e. branches with their own vmcnt(0).  A clamped offset is plain arithmetic.
template <bool NT>
__device__ __forceinline__ f32x4 load_x4c(const float* row, int koff, int D0) {
    const int o = koff < D0 - 4 ? koff : D0 - 4;
    const f32x4* q = reinterpret_cast<const f32x4*>(row + o);
    if (NT) return __builtin_nontemporal_load(q);
    return *q;
}

// The streaming pair kernels also take BFLOAT16 rows (XM = 2: what an extractor running in bf16 hands over — BASELINE
// configs[4]): FwdArgs.xa / xb then point at 2-byte elements (ldx counted in elements), a lane's four columns are one
// 8-byte load and are widened in registers; everything behind the load is the fp32 arithmetic of the fp32-row kernel on
// the widened values, bit for bit.  XM: 0 fp32 rows, 1 fp32 rows with non-temporal loads, 2 bf16 rows.
template <int XM>
__device__ __forceinline__ const float* x_row(const float* base, long long row, long long ldx) {
    if (XM == 2) return reinterpret_cast<const float*>(reinterpret_cast<const unsigned short*>(base) + row * ldx);
    return base + row * ldx;
}
template <int XM>
__device__ __forceinline__ f32x4 load_xrow(const float* row, int koff, int D0) {
    if (XM != 2) return load_x4c<XM == 1>(row, koff, D0);
    const int o = koff < D0 - 4 ? koff : D0 - 4;
    typedef unsigned u32x2 __attribute__((ext_vector_type(2)));
    const u32x2 r = *reinterpret_cast<const u32x2*>(reinterpret_cast<const unsigned short*>(row) + o);
    return f32x4{__uint_as_float(r[0] << 16), __uint_as_float(r[0] & 0xffff0000u), __uint_as_float(r[1] << 16),
                 __uint_as_float(r[1] & 0xffff0000u)};
}

struct FwdArgs {
    const float* xa;      // group-A rows (x1, or x in embed mode)
    const float* xb;      // group-B rows (x2, or x in embed mode)
    long long n;          // pairs (pair/train mode) or rows (embed mode)
    long long ldx;
    const float* packed;  // NpldaLayout image
    int D0, KS1;
    size_t oW2, ob1, ob2, oQ, oP, total;
    float* out_s;         // (n) scores                        [pair, train]
    float* out_z;         // (rows, ldz) embeddings            [embed]; train: z1 rows then z2 rows
    long long ldz;
    float* out_q;         // (rows) self terms                 [embed, optional]
    float* out_y;         // train: (2n, ldz) normalised layer-1 outputs, x1 rows then x2 rows
    float* out_rn;        // train: (2n) 1/max(||u||, eps)
    int no_norm;          // MODE_GB only: skip F.normalize (rows are already the paired embeddings)
    // Indexed pairs (nplda_fwd_mid.h only: nplda_score_pairs_rows_f32): pair i reads rows ia[i] of xa and ib[i] of xb
    // (both the resident x-vector table; indices clamped into [0, ntab)) — the gather of
    // load_xvec_trials_from_numbatch (utils/sv_trials_loaders.py:418-426) folded into the scoring kernel.
    const long long* ia;
    const long long* ib;
    long long ntab;
    // Two-table embedding (nplda_fwd_mid.h, embed mode only: nplda_embed_pair_f32): rows [0, nsplit) are rows of xa, rows
    // [nsplit, n) rows of xb (same ldx) — e.g. the enroll / test rows and the cohort of one AS-norm call in ONE launch.
    // 0 = off (one table, xa).
    long long nsplit;
};

// WAVES waves per block, each owning 16 pairs (or 32 rows); KPB k16-steps of weights per barrier.
// ABL (ablation bit mask, tools/exp_fwd.hip only; results are WRONG when non-zero): 1 = no weight
// streaming and no barriers, 2 = no LDS fragment reads, 4 = no x loads after the prologue.
template <int NB, int MODE, int WAVES, bool NT, int KPB, int ABL = 0>
__global__ __launch_bounds__(WAVES * 64, (WAVES == 4 ? 2 : 2)) void nplda_fwd_kernel(const FwdArgs a) {
    constexpr int THREADS = WAVES * 64;
    constexpr int STEP4 = NB * 64;          // float4 per k16-step of weights
    constexpr int CH = STEP4 * KPB;         // float4 per chunk
    constexpr int NSLOT = (CH + THREADS - 1) / THREADS;
    __shared__ f32x4 wbuf[2][CH];

    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int j = lane & 15;
    const int g = lane >> 4;

    long long t0A, t0B;
    if (MODE == MODE_EMBED) {
        t0A = ((long long)blockIdx.x * WAVES + wave) * 32;
        t0B = t0A + 16;
    } else {
        t0A = ((long long)blockIdx.x * WAVES + wave) * 16;
        t0B = t0A;
    }
    long long rowA = t0A + j, rowB = t0B + j;
    const bool okA = rowA < a.n, okB = rowB < a.n;
    if (!okA) rowA = a.n - 1;
    if (!okB) rowB = a.n - 1;
    const float* sa = a.xa + rowA * a.ldx;  // row starts: always-valid addresses for the branch-free loads
    const float* sb = a.xb + rowB * a.ldx;
    const float* pa = sa + 4 * g;
    const float* pb = sb + 4 * g;

    const f32x4* Wall = reinterpret_cast<const f32x4*>(a.packed);  // W1 steps then W2 steps, contiguous
    const long long total4 = (long long)(a.total / 4);
    const f32x4* b1p = reinterpret_cast<const f32x4*>(a.packed + a.ob1);
    const f32x4* b2p = reinterpret_cast<const f32x4*>(a.packed + a.ob2);
    const f32x4* Qp = reinterpret_cast<const f32x4*>(a.packed + a.oQ);
    const f32x4* Pp = reinterpret_cast<const f32x4*>(a.packed + a.oP);
    const int KS1 = a.KS1;
    const int D0 = a.D0;
    const int NC1 = (KS1 + KPB - 1) / KPB;  // layer-1 chunks
    const long long w2base4 = (long long)(a.oW2 / 4);

    f32x4 st[NSLOT];
    chunk_load<CH, THREADS, NSLOT>(Wall, total4, st, tid);
    f32x4 xa[KPB], xb[KPB];
#pragma unroll
    for (int s = 0; s < KPB; ++s) {
        const int kk = 16 * s + 4 * g;
        xa[s] = load_x4c<NT>(sa, kk, D0);
        xb[s] = load_x4c<NT>(sb, kk, D0);
    }

    f32x4 accA[NB], accB[NB];
#pragma unroll
    for (int nb = 0; nb < NB; ++nb) {
        accA[nb] = b1p[4 * nb + g];
        accB[nb] = accA[nb];
    }
    chunk_store<CH, THREADS, NSLOT>(wbuf[0], st, tid);
    __syncthreads();

    // ---- layer 1: u^T = W1 x^T + b1, K = D0 streamed KPB k16-steps per barrier -----------------
    for (int c = 0; c < NC1; ++c) {
        const int cur = c & 1;
        const bool more = (c + 1 < NC1);
        // next chunk: layer-1 chunk c+1, or the first layer-2 chunk
        const long long nbase = more ? (long long)(c + 1) * CH : w2base4;
        if (!(ABL & 1)) chunk_load<CH, THREADS, NSLOT>(Wall + nbase, total4 - nbase, st, tid);
        f32x4 xan[KPB], xbn[KPB];
#pragma unroll
        for (int s = 0; s < KPB; ++s) {
            const int ks = KPB * (c + 1) + s;
            if (ABL & 4) { xan[s] = xa[s]; xbn[s] = xb[s]; continue; }
            xan[s] = load_x4c<NT>(sa, 16 * ks + 4 * g, D0);
            xbn[s] = load_x4c<NT>(sb, 16 * ks + 4 * g, D0);
        }
        const f32x4* w = wbuf[cur];
#pragma unroll
        for (int s = 0; s < KPB; ++s) {
            if (KPB * c + s < KS1) {
#pragma unroll
                for (int nb = 0; nb < NB; ++nb) {
                    f32x4 av;
                    if (ABL & 2) { av = st[0]; asm volatile("" : "+v"(av)); } else av = w[s * STEP4 + nb * 64 + lane];
#pragma unroll
                    for (int r = 0; r < 4; ++r) {
                        accA[nb] = __builtin_amdgcn_mfma_f32_16x16x4f32(av[r], xa[s][r], accA[nb], 0, 0, 0);
                        accB[nb] = __builtin_amdgcn_mfma_f32_16x16x4f32(av[r], xb[s][r], accB[nb], 0, 0, 0);
                    }
                }
            }
        }
#pragma unroll
        for (int s = 0; s < KPB; ++s) {
            xa[s] = xan[s];
            xb[s] = xbn[s];
        }
        if (!(ABL & 1)) {
            chunk_store<CH, THREADS, NSLOT>(wbuf[cur ^ 1], st, tid);
            __syncthreads();
        }
    }

    // ---- F.normalize (utils/models.py:368): y = u / max(||u||_2, 1e-12) ------------------------
    float invA, invB;
    {
        float ssA = 0.f, ssB = 0.f;
#pragma unroll
        for (int nb = 0; nb < NB; ++nb) {
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                ssA = fmaf(accA[nb][r], accA[nb][r], ssA);
                ssB = fmaf(accB[nb][r], accB[nb][r], ssB);
            }
        }
        ssA = wave_xor_add(ssA, 16); ssA = wave_xor_add(ssA, 32);
        ssB = wave_xor_add(ssB, 16); ssB = wave_xor_add(ssB, 32);
        invA = 1.0f / fmaxf(sqrtf(ssA), 1e-12f);
        invB = 1.0f / fmaxf(sqrtf(ssB), 1e-12f);
        if (MODE == MODE_GB && a.no_norm) invA = invB = 1.0f;
#pragma unroll
        for (int nb = 0; nb < NB; ++nb) {
            accA[nb] *= invA;
            accB[nb] *= invB;
        }
    }
    if (MODE == MODE_TRAIN) {
        // save y and 1/||u|| for the backward pass (rows [0,n) = x1 side, [n,2n) = x2 side)
#pragma unroll
        for (int nb = 0; nb < NB; ++nb) {
            if (okA) *reinterpret_cast<f32x4*>(a.out_y + rowA * a.ldz + 16 * nb + 4 * g) = accA[nb];
            if (okB) *reinterpret_cast<f32x4*>(a.out_y + (a.n + rowB) * a.ldz + 16 * nb + 4 * g) = accB[nb];
        }
        if (g == 0 && okA) {
            a.out_rn[rowA] = invA;
            a.out_rn[a.n + rowB] = invB;
        }
    }

    if (MODE == MODE_GB && a.out_rn != nullptr && g == 0 && okA) {  // DPlda / GB rows for an LDA backward
        a.out_rn[rowA] = invA;
        a.out_rn[a.n + rowB] = invB;
    }
    if (MODE == MODE_GB) {
        // ---- GaussianBackend.forward (utils/models.py:584-593) on x = [y1; y2] --------------------
        // S = -(x-mu_t)^T L_t (x-mu_t) + (x-mu_n)^T L_n (x-mu_n) = x^T M x + x^T v + c with
        // M = L_n - L_t (folded when the image was packed).  t = M x + v is four chained MFMA GEMMs
        // (M11 y1 + M12 y2, M21 y1 + M22 y2) fed straight from the layer-1 accumulators, then
        // S = y1.t1 + y2.t2 + c.  Packed image: a.oW2 -> G[h_out][h_in][kb][nb] fragments,
        // a.ob2 -> v (two padded halves), a.oQ -> c.
        static_assert(MODE != MODE_GB || (2 * NB) % KPB == 0, "a chunk of G steps must not straddle the two output halves");
        if (a.out_z != nullptr) {  // forward_getpaired: (n, 2 D1) rows [y1 | y2]
            const int D1 = (int)a.ldz / 2;
#pragma unroll
            for (int nb = 0; nb < NB; ++nb) {
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    const int f = 16 * nb + 4 * g + r;
                    if (okA && f < D1) {
                        a.out_z[rowA * a.ldz + f] = accA[nb][r];
                        a.out_z[rowA * a.ldz + D1 + f] = accB[nb][r];
                    }
                }
            }
        }
        if (a.out_s == nullptr) return;
        const f32x4* vp = reinterpret_cast<const f32x4*>(a.packed + a.ob2);
        float part = 0.f;
#pragma unroll
        for (int ho = 0; ho < 2; ++ho) {
            f32x4 t[NB];
#pragma unroll
            for (int nb = 0; nb < NB; ++nb) t[nb] = vp[(ho * NB + nb) * 4 + g];
#pragma unroll
            for (int hc = 0; hc < 2 * NB / KPB; ++hc) {   // chunks of KPB G-steps (one step = one (h_in, kb) k-block)
                const int qc = ho * (2 * NB / KPB) + hc;   // chunk index within the G stream
                const int cur = (NC1 + qc) & 1;
                const bool moreg = qc + 1 < 4 * NB / KPB;
                if (moreg) {
                    const long long nbase = w2base4 + (long long)(qc + 1) * CH;
                    chunk_load<CH, THREADS, NSLOT>(Wall + nbase, total4 - nbase, st, tid);
                }
                const f32x4* w = wbuf[cur];
#pragma unroll
                for (int s = 0; s < KPB; ++s) {
                    const int hk = hc * KPB + s;
                    const int hi = hk / NB, kb = hk % NB;
                    // groups of 4 output blocks: 4 independent accumulator chains per fragment quartet
#pragma unroll
                    for (int nb0 = 0; nb0 < NB; nb0 += 4) {
                        f32x4 av[4];
#pragma unroll
                        for (int u = 0; u < 4; ++u)
                            if (nb0 + u < NB) av[u] = w[s * STEP4 + (nb0 + u) * 64 + lane];
#pragma unroll
                        for (int r = 0; r < 4; ++r) {
                            const float bv = hi == 0 ? accA[kb][r] : accB[kb][r];
#pragma unroll
                            for (int u = 0; u < 4; ++u)
                                if (nb0 + u < NB)
                                    t[nb0 + u] = __builtin_amdgcn_mfma_f32_16x16x4f32(av[u][r], bv, t[nb0 + u], 0, 0, 0);
                        }
                    }
                }
                if (moreg) {
                    chunk_store<CH, THREADS, NSLOT>(wbuf[cur ^ 1], st, tid);
                    __syncthreads();
                }
            }
#pragma unroll
            for (int nb = 0; nb < NB; ++nb) {
#pragma unroll
                for (int r = 0; r < 4; ++r) part = fmaf(ho == 0 ? accA[nb][r] : accB[nb][r], t[nb][r], part);
            }
        }
        part = wave_xor_add(part, 16);
        part = wave_xor_add(part, 32);
        if (g == 0 && okA) a.out_s[t0A + j] = part + a.packed[a.oQ];
        return;
    }

    // ---- layer 2: z^T = W2 y^T + b2; y comes straight from the layer-1 accumulators -----------
    f32x4 zA[NB], zB[NB];
#pragma unroll
    for (int nb = 0; nb < NB; ++nb) {
        zA[nb] = b2p[4 * nb + g];
        zB[nb] = zA[nb];
    }
    constexpr int NC2 = (NB + KPB - 1) / KPB;
#pragma unroll
    for (int c2 = 0; c2 < NC2; ++c2) {
        const int cur = (NC1 + c2) & 1;
        if (c2 + 1 < NC2 && !(ABL & 1)) {
            const long long nbase = w2base4 + (long long)(c2 + 1) * CH;
            chunk_load<CH, THREADS, NSLOT>(Wall + nbase, total4 - nbase, st, tid);
        }
        const f32x4* w = wbuf[cur];
#pragma unroll
        for (int s = 0; s < KPB; ++s) {
            constexpr int dummy = 0; (void)dummy;
            const int kb = KPB * c2 + s;
            if (kb < NB) {
#pragma unroll
                for (int nb = 0; nb < NB; ++nb) {
                    f32x4 av;
                    if (ABL & 2) { av = st[0]; asm volatile("" : "+v"(av)); } else av = w[s * STEP4 + nb * 64 + lane];
#pragma unroll
                    for (int r = 0; r < 4; ++r) {
                        zA[nb] = __builtin_amdgcn_mfma_f32_16x16x4f32(av[r], accA[kb < NB ? kb : 0][r], zA[nb], 0, 0, 0);
                        zB[nb] = __builtin_amdgcn_mfma_f32_16x16x4f32(av[r], accB[kb < NB ? kb : 0][r], zB[nb], 0, 0, 0);
                    }
                }
            }
        }
        if (c2 + 1 < NC2 && !(ABL & 1)) {
            chunk_store<CH, THREADS, NSLOT>(wbuf[cur ^ 1], st, tid);
            __syncthreads();
        }
    }

    // ---- epilogue ------------------------------------------------------------------------------
    if (MODE == MODE_PAIR || MODE == MODE_TRAIN) {
        // utils/models.py:372-376: s = sum Q (z1^2 + z2^2) + 2 sum P z1 z2
        float part = 0.f;
#pragma unroll
        for (int nb = 0; nb < NB; ++nb) {
            const f32x4 q = Qp[4 * nb + g];
            const f32x4 p = Pp[4 * nb + g];
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const float z1 = zA[nb][r], z2 = zB[nb][r];
                part = fmaf(q[r], fmaf(z1, z1, z2 * z2), part);
                part = fmaf(2.0f * p[r], z1 * z2, part);
            }
            if (MODE == MODE_TRAIN) {
                if (okA) *reinterpret_cast<f32x4*>(a.out_z + rowA * a.ldz + 16 * nb + 4 * g) = zA[nb];
                if (okB) *reinterpret_cast<f32x4*>(a.out_z + (a.n + rowB) * a.ldz + 16 * nb + 4 * g) = zB[nb];
            }
        }
        part = wave_xor_add(part, 16);
        part = wave_xor_add(part, 32);
        if (g == 0 && okA) a.out_s[t0A + j] = part;
    } else {
        float qa = 0.f, qb = 0.f;
#pragma unroll
        for (int nb = 0; nb < NB; ++nb) {
            const f32x4 q = Qp[4 * nb + g];
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                qa = fmaf(q[r] * zA[nb][r], zA[nb][r], qa);
                qb = fmaf(q[r] * zB[nb][r], zB[nb][r], qb);
            }
            if (okA) *reinterpret_cast<f32x4*>(a.out_z + rowA * a.ldz + 16 * nb + 4 * g) = zA[nb];
            if (okB) *reinterpret_cast<f32x4*>(a.out_z + rowB * a.ldz + 16 * nb + 4 * g) = zB[nb];
        }
        if (a.out_q != nullptr) {
            qa = wave_xor_add(qa, 16); qa = wave_xor_add(qa, 32);
            qb = wave_xor_add(qb, 16); qb = wave_xor_add(qb, 32);
            if (g == 0 && okA) a.out_q[rowA] = qa;
            if (g == 0 && okB) a.out_q[rowB] = qb;
        }
    }
}

// One thread per packed float.
static __global__ void nplda_pack_kernel(const float* __restrict__ W1, const float* __restrict__ b1,
                                  const float* __restrict__ W2, const float* __restrict__ b2,
                                  const float* __restrict__ P_sqrt, const float* __restrict__ Q,
                                  NpldaLayout L, float* __restrict__ out) {
    const size_t idx = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (idx >= L.total) return;
    float v = 0.f;
    if (idx >= L.oW1T && idx < L.ob1) {  // W1^T image [kb][xb][lane][i] (dx = du W1: nplda_matmul.hip)
        const size_t rel = idx - L.oW1T;
        const int i = (int)(rel & 3), lane = (int)((rel >> 2) & 63);
        const size_t blk = rel >> 8;
        const int xb = (int)(blk % L.KS1), kb = (int)(blk / L.KS1);
        const int f = 16 * kb + 4 * (lane >> 4) + i, k = 16 * xb + (lane & 15);
        if (f < L.D1 && k < L.D0) v = W1[(size_t)f * L.D0 + k];
    } else if (idx < L.ob1) {
        const int region = idx >= L.oW2T ? 2 : (idx >= L.oW2 ? 1 : 0);
        const size_t rel = region == 2 ? idx - L.oW2T : (region == 1 ? idx - L.oW2 : idx);
        const int i = (int)(rel & 3);
        const int lane = (int)((rel >> 2) & 63);
        const size_t blk = rel >> 8;  // ks * NB + nb
        const int nb = (int)(blk % L.NB);
        const int ks = (int)(blk / L.NB);
        const int f = 16 * nb + (lane & 15);
        const int k = 16 * ks + 4 * (lane >> 4) + i;
        if (region == 0) {
            if (f < L.D1 && k < L.D0) v = W1[(size_t)f * L.D0 + k];
        } else if (region == 1) {
            if (f < L.D2 && k < L.D1) v = W2[(size_t)f * L.D1 + k];
        } else {  // W2^T image for the backward data-gradient chain: A[f = layer-1 feature][k = layer-2 feature]
            if (f < L.D1 && k < L.D2) v = W2[(size_t)k * L.D1 + f];
        }
    } else if (idx < L.ob2) {
        const int f = (int)(idx - L.ob1);
        if (f < L.D1) v = b1[f];
    } else if (idx < L.oQ) {
        const int f = (int)(idx - L.ob2);
        if (f < L.D2) v = b2[f];
    } else if (idx < L.oP) {
        const int f = (int)(idx - L.oQ);
        if (f < L.D2) v = Q[f];
    } else {
        const int f = (int)(idx - L.oP);
        if (f < L.D2) v = P_sqrt[f] * P_sqrt[f];  // utils/models.py:373
    }
    out[idx] = v;
}

}  // namespace nplda

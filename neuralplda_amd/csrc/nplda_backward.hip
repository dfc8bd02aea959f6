// nplda_backward.hip — hand-derived backward of the Neural-PLDA pair score (gfx950, fp32 MFMA).
//
// The reference has no backward code: it is whatever autograd derives for utils/models.py:366-382
// (138 ATen launches at B = 4096).  Here it is three launches, formulas per
// SURVEY.md §3.3 (verified against the reference's fp64 autograd in tests/test_oracle_golden.py):
//
//  K-A  bwd_data_kernel   per 16-pair tile (same wave/lane layout as the forward):
//         dz1 = 2 g (Q z1 + P z2),  dz2 = 2 g (Q z2 + P z1)                      -> stored
//         dy  = dz W2   as a CHAINED MFMA: A = W2^T fragments (packed image, LDS-staged),
//                       B = dz straight from registers (same k-permutation trick as the forward)
//         du  = (dy - y (y.dy)) / max(||u||, eps)      (F.normalize backward)     -> stored
//  K-B  wgrad_kernel      dW2 = [dz1;dz2]^T [y1;y2],  dW1 = [du1;du2]^T [x1;x2]: "A^T B" GEMMs with
//         K = 2B rows, split-K over waves, 64x64 wave tiles, operands read as float4 straight from
//         row-major global memory (the 4 components of a float4 address 4 interleaved MFMA blocks, so
//         no transposition is needed).  The n-tile-0 waves also accumulate the column sums
//         db2 = sum dz, db1 = sum du, dQ = sum g z^2, dP = sum g z1 z2 as a by-product of their loads.
//  K-C  reduce_kernel     fixed-order sum of the split-K slabs into one flat gradient buffer
//         [dW1 | db1 | dW2 | db2 | dP_sqrt (= 4 P_sqrt dP) | dQ]  — deterministic, and the single
//         buffer a data-parallel all-reduce needs.
#include "nplda_fwd_dispatch.h"
#include "nplda_adam_math.h"
#include "nplda_loss_math.h"
#include "nplda_cohort_qz.h"
#include "nplda_bwd_loss.h"
#include "nplda_loss_tail.h"
#include <cstdlib>
#include <cstring>
#include "nplda_train_fb_small.h"
#include "nplda_train_fb_half.h"
#include <cstdlib>
#include "nplda_wgrad_fm.h"

namespace nplda {  // nplda_matmul.hip
int input_grad_from_du(const float* du, long long rows, long long ldz, const float* packed, const NpldaLayout& L,
                       float* frag, void* dx0, void* dx1, long long nsplit, long long lddx, hipStream_t st, bool out_bf16 = false);
int pad_rows(const float* in, long long ldin, long long N, int D, float* out, long long ldo, hipStream_t st);
}  // namespace nplda

namespace {

using namespace nplda;

// ------------------------------------------------------------------------------------------------
// K-A
// ------------------------------------------------------------------------------------------------
// Row bookkeeping: a tile holds 16 "A" rows t0 + j (< nA) and 16 "B" rows offB + t0 + j (t0 + j < nB).  Pair scoring:
// nA = nB = offB = B (x1 side, x2 side).  Embedding rows (GIVEN): N rows split into two halves, nA = ceil(N / 2),
// nB = N - nA, offB = nA — the same kernels, the pairing is then just a way to fill both MFMA row groups.
struct BwdArgs {
    const float* g;       // (nA) dL/ds per pair                       [unused when GIVEN]
    const float* z;       // (rows, ldz)                               [unused when GIVEN]
    const float* y;       // (rows, ldz)
    const float* rn;      // (rows)
    const float* packed;
    long long nA, nB, offB, ldz;
    size_t oW2T, oQ, oP, total;
    float* dz;            // (rows, ldz): written, or READ when GIVEN (upstream dL/dz, zero-padded columns)
    float* du;            // (rows, ldz)
    int ntb;              // tile-blocks = ceil(nA / (16 * WAVES))
    float* pq;            // small kernel, pair scoring: [blocks][2][ldz] per-block sums of g (z1^2 + z2^2) and g z1 z2
    BwdLoss ls;           // [LOSS kernels only]
};

template <int NB, int WAVES, bool GIVEN>
__global__ __launch_bounds__(WAVES * 64, 1) void bwd_data_kernel(const BwdArgs a) {
    constexpr int THREADS = WAVES * 64;
    constexpr int CH = NB * 64;
    constexpr int NSLOT = WAVES > 1 ? (CH + THREADS - 1) / THREADS : 1;
    __shared__ f32x4 wbuf[2][WAVES > 1 ? CH : 1];

    const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6), j = lane & 15, g4 = lane >> 4;
    const f32x4* W2T = reinterpret_cast<const f32x4*>(a.packed + a.oW2T);
    const f32x4* Qp = reinterpret_cast<const f32x4*>(a.packed + a.oQ);
    const f32x4* Pp = reinterpret_cast<const f32x4*>(a.packed + a.oP);
    const long long avail = (long long)((a.total - a.oW2T) / 4);

    for (int tb = blockIdx.x; tb < a.ntb; tb += gridDim.x) {
        const long long t0 = ((long long)tb * WAVES + wave) * 16;
        const bool okA = t0 + j < a.nA, okB = t0 + j < a.nB;
        const long long rA = okA ? t0 + j : a.nA - 1;
        const long long rB = a.offB + (okB ? t0 + j : (a.nB > 0 ? a.nB - 1 : 0));
        f32x4 st[NSLOT];
        if (WAVES > 1) {
            __syncthreads();  // every wave is done reading wbuf from the previous tile
            chunk_load<CH, THREADS, NSLOT>(W2T, avail, st, tid);
        }
        f32x4 dzA[NB], dzB[NB];
        if (GIVEN) {
#pragma unroll
            for (int nb = 0; nb < NB; ++nb) {
                dzA[nb] = *reinterpret_cast<const f32x4*>(a.dz + rA * a.ldz + 16 * nb + 4 * g4);
                dzB[nb] = *reinterpret_cast<const f32x4*>(a.dz + rB * a.ldz + 16 * nb + 4 * g4);
            }
        } else {
            const float gi = okA ? a.g[rA] : 0.f;
#pragma unroll
            for (int nb = 0; nb < NB; ++nb) {
                const f32x4 zA = *reinterpret_cast<const f32x4*>(a.z + rA * a.ldz + 16 * nb + 4 * g4);
                const f32x4 zB = *reinterpret_cast<const f32x4*>(a.z + rB * a.ldz + 16 * nb + 4 * g4);
                const f32x4 q = Qp[4 * nb + g4], p = Pp[4 * nb + g4];
                const float tg = 2.0f * gi;
                dzA[nb] = tg * (q * zA + p * zB);
                dzB[nb] = tg * (q * zB + p * zA);
                if (okA) {
                    *reinterpret_cast<f32x4*>(a.dz + rA * a.ldz + 16 * nb + 4 * g4) = dzA[nb];
                    *reinterpret_cast<f32x4*>(a.dz + rB * a.ldz + 16 * nb + 4 * g4) = dzB[nb];
                }
            }
        }
        if (WAVES > 1) {
            chunk_store<CH, THREADS, NSLOT>(wbuf[0], st, tid);
            __syncthreads();
        }

        f32x4 dyA[NB], dyB[NB];
#pragma unroll
        for (int nb = 0; nb < NB; ++nb) {
            dyA[nb] = f32x4{0.f, 0.f, 0.f, 0.f};
            dyB[nb] = dyA[nb];
        }
#pragma unroll
        for (int kb = 0; kb < NB; ++kb) {
            const int cur = kb & 1;
            if (WAVES > 1 && kb + 1 < NB)
                chunk_load<CH, THREADS, NSLOT>(W2T + (size_t)(kb + 1) * CH, avail - (long long)(kb + 1) * CH, st, tid);
            const f32x4* w = wbuf[cur];
#pragma unroll
            for (int nb = 0; nb < NB; ++nb) {
                // one wave per block (small batches): fragments straight from the L2-resident image, no LDS/barrier
                const f32x4 av = WAVES > 1 ? w[nb * 64 + lane] : W2T[((size_t)kb * NB + nb) * 64 + lane];
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    dyA[nb] = __builtin_amdgcn_mfma_f32_16x16x4f32(av[r], dzA[kb][r], dyA[nb], 0, 0, 0);
                    dyB[nb] = __builtin_amdgcn_mfma_f32_16x16x4f32(av[r], dzB[kb][r], dyB[nb], 0, 0, 0);
                }
            }
            if (WAVES > 1 && kb + 1 < NB) {
                chunk_store<CH, THREADS, NSLOT>(wbuf[cur ^ 1], st, tid);
                __syncthreads();
            }
        }

        // F.normalize backward: du = (dy - y (y . dy)) / max(||u||, eps); clamp branch: du = dy / eps
        const float rnA = a.rn[rA], rnB = a.rn[rB];
        float dotA = 0.f, dotB = 0.f;
        // pass 1: dots (y is re-read in pass 2 from L1/L2: cheaper than 80 more live registers)
#pragma unroll
        for (int nb = 0; nb < NB; ++nb) {
            const f32x4 yA = *reinterpret_cast<const f32x4*>(a.y + rA * a.ldz + 16 * nb + 4 * g4);
            const f32x4 yB = *reinterpret_cast<const f32x4*>(a.y + rB * a.ldz + 16 * nb + 4 * g4);
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                dotA = fmaf(yA[r], dyA[nb][r], dotA);
                dotB = fmaf(yB[r], dyB[nb][r], dotB);
            }
        }
        dotA = wave_xor_add(dotA, 16); dotA = wave_xor_add(dotA, 32);
        dotB = wave_xor_add(dotB, 16); dotB = wave_xor_add(dotB, 32);
        if (rnA >= 1e12f) dotA = 0.f;
        if (rnB >= 1e12f) dotB = 0.f;
#pragma unroll
        for (int nb = 0; nb < NB; ++nb) {
            const f32x4 yA = *reinterpret_cast<const f32x4*>(a.y + rA * a.ldz + 16 * nb + 4 * g4);
            const f32x4 yB = *reinterpret_cast<const f32x4*>(a.y + rB * a.ldz + 16 * nb + 4 * g4);
            if (okA) *reinterpret_cast<f32x4*>(a.du + rA * a.ldz + 16 * nb + 4 * g4) = (dyA[nb] - yA * dotA) * rnA;
            if (okB) *reinterpret_cast<f32x4*>(a.du + rB * a.ldz + 16 * nb + 4 * g4) = (dyB[nb] - yB * dotB) * rnB;
        }
    }
}

// Streaming form of K-A (B > 16 384 pairs; NB >= 8: the whole W2^T image fits the 160 KB of LDS).
//
// The first form above holds dz of both sides (8 NB registers) next to the dy accumulators (8 NB): 300+ registers, ONE
// 4-wave block per CU, and every tile starts by waiting out the HBM latency of its z rows — 540 us at 262 144 pairs for
// 1.3 GB of traffic and 150 us of MFMA work.  Here
//  * W2^T (NB x NB KB) is loaded into LDS ONCE per block; the tile loop has no weight traffic and NO barrier, the 8 waves
//    of the (one per CU, persistent) block run independently;
//  * dz is formed one k-block at a time from a ring of z rows fetched four k-blocks ahead (and, past the tile's last
//    k-block, for the NEXT tile): 16 live registers instead of 8 NB, two waves per SIMD fit;
//  * y is read once (kept in registers between the dot product and du), issued under the last k-blocks' MFMAs;
//  * the dQ / dP pair sums g (z1^2 + z2^2), g z1 z2 — which the weight-gradient kernel used to re-derive from three
//    more loads per k4-step — are formed here from the z rows in registers: reduced over the tile's 16 pairs (DPP) and
//    added into the wave's own LDS row; one row per block leaves in the end (fixed order: deterministic).
template <int NB, bool GIVEN>
__global__ __launch_bounds__(512, 2) void bwd_data_stream_kernel(const BwdArgs a) {
    constexpr int WAVES = 8, PFZ = 4;
    __shared__ f32x4 w2t[NB * NB * 64];
    __shared__ f32x4 pqs[WAVES][2][NB * 4];  // per wave: [dQ | dP] partial sums over its tiles, 16 NB floats each
    // Q and P in LDS (round 5).  As global loads inside the k-block loop they were the YOUNGEST vector-memory operations in
    // front of their own use: hipcc waited `vmcnt(0)` for them, i.e. for the z prefetches and dz stores issued before them as
    // well — every k-block drained the memory pipeline and the four-k-block z ring hid one k-block of latency
    __shared__ f32x4 qps[2][NB * 4];
    const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6), j = lane & 15, g4 = lane >> 4;
    {
        const f32x4* W2T = reinterpret_cast<const f32x4*>(a.packed + a.oW2T);
        for (int i = tid; i < NB * NB * 64; i += WAVES * 64) w2t[i] = W2T[i];
        for (int i = tid; i < WAVES * 2 * NB * 4; i += WAVES * 64) (&pqs[0][0][0])[i] = f32x4{0.f, 0.f, 0.f, 0.f};
        if (tid < NB * 4) {
            qps[0][tid] = reinterpret_cast<const f32x4*>(a.packed + a.oQ)[tid];
            qps[1][tid] = reinterpret_cast<const f32x4*>(a.packed + a.oP)[tid];
        }
    }
    const float* src = GIVEN ? a.dz : a.z;  // GIVEN: the upstream dL/dz rows are the B operand as they are
    const long long nwt = (a.nA + 15) / 16;  // wave tiles
    // A tile's rows as a UNIFORM first row per side (element offsets eA / eB: scalar registers) and this lane's byte offset
    // inside the tile (vA / vB: one 32-bit register per side) — every load and store below is base (SGPR pair) + offset
    // (VGPR) + immediate.  As 64-bit per-lane pointers (z, dz, y, du, each for two sides and two tiles) the addresses took
    // 20+ registers of a kernel that runs at the 256-register limit: hipcc spilled five of them AND the uniform array bases
    // (held in vector registers), and each of the eight reloads per tile was followed by s_waitcnt vmcnt(0) — the memory
    // counter is in order, so every reload drained the z ring and the dz stores in flight (round 6).
    struct TileRows {
        long long eA, eB;
        unsigned vA, vB, rA4, rB4;
        bool okA, okB;
    };
    auto sgpr64 = [](long long v) {
        const unsigned lo = __builtin_amdgcn_readfirstlane((unsigned)(unsigned long long)v);
        const unsigned hi = __builtin_amdgcn_readfirstlane((unsigned)((unsigned long long)v >> 32));
        return (long long)(((unsigned long long)hi << 32) | lo);
    };
    auto rows_of = [&](long long wt_, TileRows& T) {
        const long long t0 = sgpr64(wt_ * 16);
        const long long lastB = a.nB > 0 ? a.nB - 1 : 0;
        const long long b0 = t0 < a.nB ? t0 : lastB;  // the tile's first row of the B side (rows past nB repeat the last one)
        T.okA = t0 + j < a.nA;
        T.okB = t0 + j < a.nB;
        const int jA = T.okA ? j : (int)(a.nA - 1 - t0);
        const int jB = T.okB ? j : (int)(lastB - b0);
        T.eA = sgpr64(t0 * a.ldz);
        T.eB = sgpr64((a.offB + b0) * a.ldz);
        T.vA = (unsigned)(jA * (int)a.ldz + 4 * g4) * 4u;
        T.vB = (unsigned)(jB * (int)a.ldz + 4 * g4) * 4u;
        T.rA4 = (unsigned)jA * 4u;  // the row's own scalar (rn, g): rn + t0 is rn + eA / ldz
        T.rB4 = (unsigned)jB * 4u;
    };
    // (the lane offset is made opaque at every use: left visible, hipcc forms base + offset ONCE per array as a 64-bit vector
    // add, keeps the sums — the per-lane pointers again — and the loads lose the scalar-base form)
    // (for every NB: with the offsets left visible at NB = 11 — where the v_mov per use is not paid back by fewer spills, D = 170
    // is 0.7 % slower than round 5 — the backward is 3.6 % slower still: profiles/r06gn_backward_nb11_ab.txt)
    auto ld4 = [](const float* arr, long long e, unsigned v, int imm) {
        asm volatile("" : "+v"(v));
        return *reinterpret_cast<const f32x4*>(reinterpret_cast<const char*>(arr + e) + v + imm);
    };
    auto st4 = [](float* arr, long long e, unsigned v, int imm, const f32x4& x) {
        asm volatile("" : "+v"(v));
        *reinterpret_cast<f32x4*>(reinterpret_cast<char*>(arr + e) + v + imm) = x;
    };
    long long wt = (long long)blockIdx.x * WAVES + wave;
    const long long stride = (long long)gridDim.x * WAVES;
    TileRows T;
    rows_of(wt < nwt ? wt : nwt - 1, T);
    f32x4 zrA[PFZ], zrB[PFZ];
    auto fetchz = [&](int slot, const TileRows& R, int kb) {
        zrA[slot] = ld4(src, R.eA, R.vA, 64 * kb);
        zrB[slot] = ld4(src, R.eB, R.vB, 64 * kb);
    };
#pragma unroll
    for (int s = 0; s < PFZ; ++s) fetchz(s, T, s);
    __syncthreads();  // the weights are in LDS

    for (; wt < nwt; wt += stride) {
        TileRows Tn;
        rows_of(wt + stride < nwt ? wt + stride : wt, Tn);
        const long long t0 = sgpr64(wt * 16);
        const long long b0r = sgpr64(a.offB + (t0 < a.nB ? t0 : (a.nB > 0 ? a.nB - 1 : 0)));
        const float tg = GIVEN ? 0.f : 2.0f * (T.okA ? *reinterpret_cast<const float*>(reinterpret_cast<const char*>(a.g + t0) + T.rA4) : 0.f);
        f32x4 dyA[NB], dyB[NB], yA[NB], yB[NB];
#pragma unroll
        for (int nb = 0; nb < NB; ++nb) {
            dyA[nb] = f32x4{0.f, 0.f, 0.f, 0.f};
            dyB[nb] = dyA[nb];
        }
#pragma unroll
        for (int kb = 0; kb < NB; ++kb) {
            const int sl = kb % PFZ;
            f32x4 dA, dB;
            if (GIVEN) {
                dA = zrA[sl];
                dB = zrB[sl];
            } else {
                const f32x4 q = qps[0][4 * kb + g4], p = qps[1][4 * kb + g4];
                dA = dz_of(tg, q, p, zrA[sl], zrB[sl]);
                dB = dz_of(tg, q, p, zrB[sl], zrA[sl]);
                if (T.okA) {
                    st4(a.dz, T.eA, T.vA, 64 * kb, dA);
                    st4(a.dz, T.eB, T.vB, 64 * kb, dB);
                }
                f32x4 eq, ep;
                pair_sum_terms(0.5f * tg, zrA[sl], zrB[sl], eq, ep);
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    eq[r] = row16_sum(eq[r]);
                    ep[r] = row16_sum(ep[r]);
                }
                if (j == 0) {  // the wave's own row: a plain read-modify-write
                    pqs[wave][0][4 * kb + g4] += eq;
                    pqs[wave][1][4 * kb + g4] += ep;
                }
            }
            // refill the slot: k-block kb + PFZ of this tile, or the first k-blocks of the wave's next tile
            if (kb + PFZ < NB) fetchz(sl, T, kb + PFZ);
            else fetchz(sl, Tn, kb + PFZ - NB);
            if (kb >= NB - 2) {  // y under the last MFMAs: half of the blocks each
#pragma unroll
                for (int nb = (kb == NB - 2 ? 0 : NB / 2); nb < (kb == NB - 2 ? NB / 2 : NB); ++nb) {
                    yA[nb] = ld4(a.y, T.eA, T.vA, 64 * nb);
                    yB[nb] = ld4(a.y, T.eB, T.vB, 64 * nb);
                }
            }
            __builtin_amdgcn_sched_barrier(0);  // (left free, the scheduler hoists the LDS reads of every k-block to the top: spills)
            // W2^T fragments two blocks at a time, the NEXT pair read under this pair's 16 MFMAs (round 5: read and waited
            // for in front of their own MFMAs, every pair opened with an exposed LDS round trip, five per k-block)
            f32x4 av[2][2];
#pragma unroll
            for (int u = 0; u < 2; ++u)
                if (u < NB) av[0][u] = w2t[(kb * NB + u) * 64 + lane];
#pragma unroll
            for (int nb0 = 0; nb0 < NB; nb0 += 2) {
                const int cur = (nb0 >> 1) & 1;
                if (nb0 + 2 < NB) {
#pragma unroll
                    for (int u = 0; u < 2; ++u)
                        if (nb0 + 2 + u < NB) av[cur ^ 1][u] = w2t[(kb * NB + nb0 + 2 + u) * 64 + lane];
                }
                __builtin_amdgcn_sched_barrier(0);
#pragma unroll
                for (int r = 0; r < 4; ++r) {
#pragma unroll
                    for (int u = 0; u < 2; ++u) {
                        if (nb0 + u < NB) {
                            dyA[nb0 + u] = __builtin_amdgcn_mfma_f32_16x16x4f32(av[cur][u][r], dA[r], dyA[nb0 + u], 0, 0, 0);
                            dyB[nb0 + u] = __builtin_amdgcn_mfma_f32_16x16x4f32(av[cur][u][r], dB[r], dyB[nb0 + u], 0, 0, 0);
                        }
                    }
                }
                __builtin_amdgcn_sched_barrier(0);
            }
        }
        // F.normalize backward: du = (dy - y (y . dy)) / max(||u||, eps); clamp branch: du = dy / eps
        const float rnA = *reinterpret_cast<const float*>(reinterpret_cast<const char*>(a.rn + t0) + T.rA4);
        const float rnB = *reinterpret_cast<const float*>(reinterpret_cast<const char*>(a.rn + b0r) + T.rB4);
        float dotA = 0.f, dotB = 0.f;
#pragma unroll
        for (int nb = 0; nb < NB; ++nb) {
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                dotA = fmaf(yA[nb][r], dyA[nb][r], dotA);
                dotB = fmaf(yB[nb][r], dyB[nb][r], dotB);
            }
        }
        dotA = wave_xor_add(dotA, 16); dotA = wave_xor_add(dotA, 32);
        dotB = wave_xor_add(dotB, 16); dotB = wave_xor_add(dotB, 32);
        if (rnA >= 1e12f) dotA = 0.f;
        if (rnB >= 1e12f) dotB = 0.f;
#pragma unroll
        for (int nb = 0; nb < NB; ++nb) {
            if (T.okA) st4(a.du, T.eA, T.vA, 64 * nb, (dyA[nb] - yA[nb] * dotA) * rnA);
            if (T.okB) st4(a.du, T.eB, T.vB, 64 * nb, (dyB[nb] - yB[nb] * dotB) * rnB);
        }
        T = Tn;
        // the next tile's k-block i was fetched into slot (NB - PFZ + i) % PFZ (the slot that was free): back to slot i
        if constexpr (NB % PFZ != 0) {
            f32x4 tA[PFZ], tB[PFZ];
#pragma unroll
            for (int i = 0; i < PFZ; ++i) {
                tA[i] = zrA[(NB - PFZ + i) % PFZ];
                tB[i] = zrB[(NB - PFZ + i) % PFZ];
            }
#pragma unroll
            for (int i = 0; i < PFZ; ++i) {
                zrA[i] = tA[i];
                zrB[i] = tB[i];
            }
        }
    }
    if (!GIVEN && a.pq != nullptr) {  // the block's pair sums: the 8 wave rows in wave order
        __syncthreads();
        for (int i = tid; i < 2 * NB * 4; i += WAVES * 64) {
            const int row = i / (NB * 4), e = i % (NB * 4);
            f32x4 v = pqs[0][row][e];
#pragma unroll
            for (int w = 1; w < WAVES; ++w) v += pqs[w][row][e];
            *reinterpret_cast<f32x4*>(a.pq + ((size_t)blockIdx.x * 2 + row) * a.ldz + 4 * e) = v;
        }
    }
}

// Small-batch form of K-A (B <= 16 384 pairs): the 4 waves of a block share one 16-pair tile, split over feature
// blocks exactly like nplda_fwd_small.h — each wave forms dz for ITS blocks and publishes them to LDS in
// accumulator layout (= the B operand of the chained MFMA), computes ITS dy blocks from all of dz with W2^T
// fragments read straight from the L2-resident image through a register ring, and the y.dy dot product of the
// normalize backward is reduced across the waves through LDS.
template <int NB, bool GIVEN, bool LOSS = false>
__global__ __launch_bounds__(256, 2) void bwd_data_small_kernel(const BwdArgs a) {
    static_assert(!(GIVEN && LOSS), "the loss is formed from pair scores");
    constexpr int NW = 4;
    constexpr int NBW = (NB + NW - 1) / NW;
    constexpr int PF = 4;
    // pair scoring at NB = 10: the two left-over blocks split by side, as in nplda_fwd_small.h / nplda_train_fb_small.h
    constexpr bool HALF = NB == 10 && !GIVEN;
    constexpr int NBF = HALF ? NB / NW : NBW;
    constexpr int HS = NBF;
    __shared__ f32x4 dzlds[2][NB][64];
    __shared__ float red[NW][2][16];
    const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int j = lane & 15, g4 = lane >> 4;
    const int hb = NW * NBF + (wave >> 1);
    const bool hside = (wave & 1) != 0;
    auto blk = [&](int i) { return HALF && i == HS ? hb : wave + NW * i; };
    const f32x4* W2T = reinterpret_cast<const f32x4*>(a.packed + a.oW2T);
    const f32x4* Qp = reinterpret_cast<const f32x4*>(a.packed + a.oQ);
    const f32x4* Pp = reinterpret_cast<const f32x4*>(a.packed + a.oP);

    const long long t0 = (long long)blockIdx.x * 16;
    const bool okA = t0 + j < a.nA, okB = t0 + j < a.nB;
    const long long rA = okA ? t0 + j : a.nA - 1;
    const long long rB = a.offB + (okB ? t0 + j : (a.nB > 0 ? a.nB - 1 : 0));

    f32x4 wf[PF][NBW];
    auto fetch = [&](int slot, int kb) {
        const int kbc = kb < NB ? kb : NB - 1;
#pragma unroll
        for (int i = 0; i < NBW; ++i) {
            const int nb = blk(i);
            wf[slot][i] = W2T[((size_t)kbc * NB + (nb < NB ? nb : NB - 1)) * 64 + lane];
        }
    };
#pragma unroll
    for (int s = 0; s < PF; ++s) fetch(s, s);

    float tg = 0.f;
    __shared__ double lacc[LOSS ? 16 : 1][kLossNS];
    if constexpr (LOSS) {
        const BwdLoss& L = a.ls;
        // N_t: every block sums the targets itself (<= 16 float4 per thread, L2 hits): no launch, no grid-wide hand-over
        __shared__ float cnt_s[NW];
        const double Nt = block_target_count(L, cnt_s);
        const double Nn = (double)L.B - Nt;
        double acc[kLossNS];
        const float gi = loss_pair(L, Nt, Nn, L.s[rA], L.t[rA], acc);
        tg = okA ? 2.0f * gi : 0.f;
        if (wave == 0 && g4 == 0) {
            if (okA) L.g_out[rA] = gi;
#pragma unroll
            for (int i = 0; i < kLossNS; ++i) lacc[j][i] = okA ? acc[i] : 0.0;
        }
    } else {
        tg = (!GIVEN && okA) ? 2.0f * a.g[rA] : 0.f;
    }
    const long long rH = hside ? rB : rA, rO = hside ? rA : rB;  // HALF: the half slot's row, its pair's other row
#pragma unroll
    for (int i = 0; i < NBF; ++i) {
        const int nb = wave + NW * i;
        if (nb < NB) {
            if (GIVEN) {
                dzlds[0][nb][lane] = *reinterpret_cast<const f32x4*>(a.dz + rA * a.ldz + 16 * nb + 4 * g4);
                dzlds[1][nb][lane] = *reinterpret_cast<const f32x4*>(a.dz + rB * a.ldz + 16 * nb + 4 * g4);
            } else {
                const f32x4 zA = *reinterpret_cast<const f32x4*>(a.z + rA * a.ldz + 16 * nb + 4 * g4);
                const f32x4 zB = *reinterpret_cast<const f32x4*>(a.z + rB * a.ldz + 16 * nb + 4 * g4);
                const f32x4 q = Qp[4 * nb + g4], p = Pp[4 * nb + g4];
                const f32x4 dA = dz_of(tg, q, p, zA, zB);
                const f32x4 dB = dz_of(tg, q, p, zB, zA);
                dzlds[0][nb][lane] = dA;
                dzlds[1][nb][lane] = dB;
                if (okA) {
                    *reinterpret_cast<f32x4*>(a.dz + rA * a.ldz + 16 * nb + 4 * g4) = dA;
                    *reinterpret_cast<f32x4*>(a.dz + rB * a.ldz + 16 * nb + 4 * g4) = dB;
                }
                // dQ_f = sum_pairs g (z1_f^2 + z2_f^2), dP_f = sum_pairs g z1_f z2_f: this block's 16 pairs, from the z
                // rows and g already in registers (in the weight-gradient kernel they were three more un-prefetched loads
                // per k4-step, and its 45 blocks that carried them set the kernel's time)
                const float gh = 0.5f * tg;
                f32x4 eq, ep;
                pair_sum_terms(gh, zA, zB, eq, ep);
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    eq[r] = row16_sum(eq[r]);
                    ep[r] = row16_sum(ep[r]);
                }
                if (j == 0) {
                    float* o = a.pq + (size_t)blockIdx.x * 2 * a.ldz + 16 * nb + 4 * g4;
                    *reinterpret_cast<f32x4*>(o) = eq;
                    *reinterpret_cast<f32x4*>(o + a.ldz) = ep;
                }
            }
        }
    }
    if constexpr (HALF) {
        const f32x4 zH = *reinterpret_cast<const f32x4*>(a.z + rH * a.ldz + 16 * hb + 4 * g4);
        const f32x4 zO = *reinterpret_cast<const f32x4*>(a.z + rO * a.ldz + 16 * hb + 4 * g4);
        const f32x4 q = Qp[4 * hb + g4], p = Pp[4 * hb + g4];
        const f32x4 dH = dz_of(tg, q, p, zH, zO);
        dzlds[hside ? 1 : 0][hb][lane] = dH;
        if (okA) *reinterpret_cast<f32x4*>(a.dz + rH * a.ldz + 16 * hb + 4 * g4) = dH;
        if (!hside) {
            f32x4 eq, ep;
            pair_sum_terms(0.5f * tg, zH, zO, eq, ep);
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                eq[r] = row16_sum(eq[r]);
                ep[r] = row16_sum(ep[r]);
            }
            if (j == 0) {
                float* o = a.pq + (size_t)blockIdx.x * 2 * a.ldz + 16 * hb + 4 * g4;
                *reinterpret_cast<f32x4*>(o) = eq;
                *reinterpret_cast<f32x4*>(o + a.ldz) = ep;
            }
        }
    }
    __syncthreads();
    if constexpr (LOSS) {  // the block's 16 pairs, summed in pair order
        if (tid < kLossNS) {
            double v = 0.0;
#pragma unroll
            for (int p = 0; p < 16; ++p) v += lacc[p][tid];
            a.ls.partial[(size_t)blockIdx.x * kLossNS + tid] = v;
        }
    }

    f32x4 dyA[NBW], dyB[NBW];
#pragma unroll
    for (int i = 0; i < NBW; ++i) {
        dyA[i] = f32x4{0.f, 0.f, 0.f, 0.f};
        dyB[i] = dyA[i];
    }
#pragma unroll
    for (int kb = 0; kb < NB; ++kb) {
        const int s = kb % PF;
        const f32x4 dA = dzlds[0][kb][lane], dB = dzlds[1][kb][lane];
        f32x4 dH;
        if constexpr (HALF) dH = dzlds[hside ? 1 : 0][kb][lane];
#pragma unroll
        for (int r = 0; r < 4; ++r) {
#pragma unroll
            for (int i = 0; i < NBF; ++i) {
                dyA[i] = __builtin_amdgcn_mfma_f32_16x16x4f32(wf[s][i][r], dA[r], dyA[i], 0, 0, 0);
                dyB[i] = __builtin_amdgcn_mfma_f32_16x16x4f32(wf[s][i][r], dB[r], dyB[i], 0, 0, 0);
            }
            if constexpr (HALF) dyA[HS] = __builtin_amdgcn_mfma_f32_16x16x4f32(wf[s][HS][r], dH[r], dyA[HS], 0, 0, 0);
        }
        fetch(s, kb + PF);
    }

    // F.normalize backward on this wave's blocks; the dot product y.dy runs over ALL features -> LDS reduction
    f32x4 yA[NBW], yB[NBW];
    float dotA = 0.f, dotB = 0.f;
#pragma unroll
    for (int i = 0; i < NBF; ++i) {
        const int nb = wave + NW * i;
        const int nbc = nb < NB ? nb : NB - 1;
        yA[i] = *reinterpret_cast<const f32x4*>(a.y + rA * a.ldz + 16 * nbc + 4 * g4);
        yB[i] = *reinterpret_cast<const f32x4*>(a.y + rB * a.ldz + 16 * nbc + 4 * g4);
        if (nb < NB) {
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                dotA = fmaf(yA[i][r], dyA[i][r], dotA);
                dotB = fmaf(yB[i][r], dyB[i][r], dotB);
            }
        }
    }
    if constexpr (HALF) {
        yA[HS] = *reinterpret_cast<const f32x4*>(a.y + rH * a.ldz + 16 * hb + 4 * g4);
        float dh = hside ? dotB : dotA;
#pragma unroll
        for (int r = 0; r < 4; ++r) dh = fmaf(yA[HS][r], dyA[HS][r], dh);
        if (hside) dotB = dh;
        else dotA = dh;
    }
    dotA = wave_xor_add(dotA, 16); dotA = wave_xor_add(dotA, 32);
    dotB = wave_xor_add(dotB, 16); dotB = wave_xor_add(dotB, 32);
    if (g4 == 0) {
        red[wave][0][j] = dotA;
        red[wave][1][j] = dotB;
    }
    __syncthreads();
    dotA = ((red[0][0][j] + red[1][0][j]) + red[2][0][j]) + red[3][0][j];
    dotB = ((red[0][1][j] + red[1][1][j]) + red[2][1][j]) + red[3][1][j];
    const float rnA = a.rn[rA], rnB = a.rn[rB];
    if (rnA >= 1e12f) dotA = 0.f;
    if (rnB >= 1e12f) dotB = 0.f;
#pragma unroll
    for (int i = 0; i < NBF; ++i) {
        const int nb = wave + NW * i;
        if (nb < NB) {
            if (okA) *reinterpret_cast<f32x4*>(a.du + rA * a.ldz + 16 * nb + 4 * g4) = du_of(dyA[i], yA[i], dotA, rnA);
            if (okB) *reinterpret_cast<f32x4*>(a.du + rB * a.ldz + 16 * nb + 4 * g4) = du_of(dyB[i], yB[i], dotB, rnB);
        }
    }
    if constexpr (HALF) {
        if (okA)
            *reinterpret_cast<f32x4*>(a.du + rH * a.ldz + 16 * hb + 4 * g4) =
                du_of(dyA[HS], yA[HS], hside ? dotB : dotA, hside ? rnB : rnA);
    }
}

// ------------------------------------------------------------------------------------------------
// K-B : C[m][n] = sum_k A[k][m] * Bm[k][n]   (K = 2n rows; Bm rows come from two segments)
// ------------------------------------------------------------------------------------------------
constexpr int kPF = 4;  // k4-steps of operand prefetch per wave (8 needed 330 registers: one block per CU; at 4 two are resident)

// One block = one (problem, 64x64 tile, k-group) work item; its 4 waves take the 4 quarters of the k-group's
// rows, then reduce their accumulators through LDS in a fixed order (deterministic) so that only ONE slab per
// k-group is written.
template <int EXT>
__device__ __forceinline__ void wgrad_body(const WgradArgs& a, const WgradProblem P, int w, f32x4 (*red)[16 * 64],
                                           f32x4 (*rede)[3][16]) {
    const int lane = threadIdx.x & 63, wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int i16 = lane & 15, g4 = lane >> 4;
    // P is a by-value copy of the problem descriptor (pointers in registers): indexing a.p[pi] inside the loop made
    // hipcc re-load the operand base pointers from the kernarg segment every k-step, a dependent load in front
    // of every operand load
    const float* __restrict__ PA = P.A;
    const float* __restrict__ PB0 = P.B0;
    const float* __restrict__ PB1 = P.B1;
    const long long lda = P.lda, ldb = P.ldb, npairs = a.nsplit, ldz = a.ldz;
    const float* __restrict__ zptr = a.z;
    const float* __restrict__ gptr = a.g;
    const int ks = w % a.ksplit;
    const int tile = w / a.ksplit;
    const int nt = tile % P.NT, mt = tile / P.NT;
    const int m0 = mt * 64, n0 = nt * 64;
    const long long K = a.K;
    const long long quarter = a.rows_per_split / 4;
    const long long k0 = (long long)ks * a.rows_per_split + wave * quarter;
    long long k1 = k0 + quarter;
    if (k1 > K) k1 = K;
    const bool mval = m0 + 4 * i16 < P.M;
    const bool nval = n0 + 4 * i16 < P.N;
    constexpr int ext = EXT;

    f32x4 acc[4][4];
#pragma unroll
    for (int ca = 0; ca < 4; ++ca)
#pragma unroll
        for (int cb = 0; cb < 4; ++cb) acc[ca][cb] = f32x4{0.f, 0.f, 0.f, 0.f};
    f32x4 e0 = {0.f, 0.f, 0.f, 0.f}, e1 = e0, e2 = e0;  // column-sum accumulators (per lane: 4 m-values)

    // Branch-free operand loads: the address is clamped into the table and the ring holds the RAW loaded values; rows
    // past the wave's range are zeroed when the slot is CONSUMED, kPF steps later.  (A select right after the load
    // makes the loaded value live at once: hipcc then waits for every load of a round before the loop's back-edge, and
    // the ring prefetches nothing — 0.9 us per k4-step instead of 0.45.)
    const f32x4 zero4 = {0.f, 0.f, 0.f, 0.f};
    const int mcol = mval ? m0 + 4 * i16 : 0;
    const int ncol = nval ? n0 + 4 * i16 : 0;
    auto loadA = [&](long long row) -> f32x4 {
        const long long rc = row < K ? row : K - 1;
        return *reinterpret_cast<const f32x4*>(PA + rc * lda + mcol);
    };
    auto loadB = [&](long long row) -> f32x4 {
        const long long rc = row < K ? row : K - 1;
        const float* base = rc < npairs ? PB0 + rc * ldb : PB1 + (rc - npairs) * ldb;
        return *reinterpret_cast<const f32x4*>(base + ncol);
    };

    f32x4 fa[kPF], fb[kPF];
#pragma unroll
    for (int s = 0; s < kPF; ++s) {
        fa[s] = loadA(k0 + 4 * s + g4);
        fb[s] = loadB(k0 + 4 * s + g4);
    }
    for (long long kk = k0; kk < k1; kk += 4 * kPF) {
#pragma unroll
        for (int s = 0; s < kPF; ++s) {
            const long long row = kk + 4 * s + g4;
            const bool live = row < k1;
            const f32x4 av = (live && mval) ? fa[s] : zero4;
            const f32x4 bv = (live && nval) ? fb[s] : zero4;
            // prefetch the same slot one round ahead — pinned here: the scheduler otherwise sinks the loads to the end of the
            // round (shorter live ranges) and the next round starts by waiting for all of them
            fa[s] = loadA(row + 4 * kPF);
            fb[s] = loadB(row + 4 * kPF);
            __builtin_amdgcn_sched_barrier(0);
            if (ext) {
                e0 += av;  // db1 (extras 1) or db2 (extras 2)
                if (ext == 2) {  // all loads unconditional (clamped), contributions masked by select
                    const bool ok = live && mval;
                    const long long rc = row < K ? row : K - 1;
                    const long long pr = rc < npairs ? rc : rc - npairs;
                    const float gi = ok ? gptr[pr] : 0.f;
                    const f32x4 zv = *reinterpret_cast<const f32x4*>(zptr + rc * ldz + mcol);
                    const f32x4 zo = *reinterpret_cast<const f32x4*>(zptr + (pr + npairs) * ldz + mcol);
                    e1 += gi * zv * zv;  // dQ
                    e2 += (rc < npairs ? gi : 0.f) * zv * zo;  // dP (each pair once)
                }
            }
#pragma unroll
            for (int ca = 0; ca < 4; ++ca)
#pragma unroll
                for (int cb = 0; cb < 4; ++cb)
                    acc[ca][cb] = __builtin_amdgcn_mfma_f32_16x16x4f32(av[ca], bv[cb], acc[ca][cb], 0, 0, 0);
        }
    }

    // ---- in-block reduction over the 4 k-quarters (fixed order 0,1,2,3) --------------------------------
#pragma unroll
    for (int ca = 0; ca < 4; ++ca)
#pragma unroll
        for (int cb = 0; cb < 4; ++cb) red[wave][(ca * 4 + cb) * 64 + lane] = acc[ca][cb];
    if (ext) {
#pragma unroll
        for (int c = 0; c < 4; ++c) {
            e0[c] = wave_xor_add(e0[c], 16); e0[c] = wave_xor_add(e0[c], 32);
            e1[c] = wave_xor_add(e1[c], 16); e1[c] = wave_xor_add(e1[c], 32);
            e2[c] = wave_xor_add(e2[c], 16); e2[c] = wave_xor_add(e2[c], 32);
        }
        if (g4 == 0) {
            rede[wave][0][i16] = e0;
            rede[wave][1][i16] = e1;
            rede[wave][2][i16] = e2;
        }
    }
    __syncthreads();
    // wave `ca` finishes block-row ca: D[i][j] of block (ca, cb) is C[m0 + 4 i + ca][n0 + 4 j + cb],
    // lane (j = i16, g4) holds i = 4 g4 + r.
    {
        const int ca = wave;
        f32x4 sum[4];
#pragma unroll
        for (int cb = 0; cb < 4; ++cb) {
            const int idx = (ca * 4 + cb) * 64 + lane;
            sum[cb] = ((red[0][idx] + red[1][idx]) + red[2][idx]) + red[3][idx];
        }
        float* slab = P.slab + (size_t)ks * P.Mp * P.Np;
        if (n0 + 4 * i16 < P.Np) {
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const int m = m0 + 4 * (4 * g4 + r) + ca;
                if (m < P.Mp) {
                    const f32x4 v = {sum[0][r], sum[1][r], sum[2][r], sum[3][r]};
                    *reinterpret_cast<f32x4*>(slab + (size_t)m * P.Np + n0 + 4 * i16) = v;
                }
            }
        }
    }
    if (ext && wave == 0 && g4 == 0 && m0 + 4 * i16 < a.Mp) {
        f32x4 t0 = ((rede[0][0][i16] + rede[1][0][i16]) + rede[2][0][i16]) + rede[3][0][i16];
        f32x4 t1 = ((rede[0][1][i16] + rede[1][1][i16]) + rede[2][1][i16]) + rede[3][1][i16];
        f32x4 t2 = ((rede[0][2][i16] + rede[1][2][i16]) + rede[2][2][i16]) + rede[3][2][i16];
        float* eb = a.ext + (size_t)ks * 4 * a.Mp + m0 + 4 * i16;
        if (ext == 1) {
            *reinterpret_cast<f32x4*>(eb + 3 * a.Mp) = t0;
        } else if (ext == 4) {
            *reinterpret_cast<f32x4*>(eb + 2 * a.Mp) = t0;
        } else {
            *reinterpret_cast<f32x4*>(eb + 0 * a.Mp) = t1;
            *reinterpret_cast<f32x4*>(eb + 1 * a.Mp) = t2;
            *reinterpret_cast<f32x4*>(eb + 2 * a.Mp) = t0;
        }
    }
}

__global__ __launch_bounds__(256, 2) void wgrad_kernel(const WgradArgs a) {
    __shared__ f32x4 red[4][16 * 64];   // [wave][(ca*4+cb)*64 + lane]  (64 KB)
    __shared__ f32x4 rede[4][3][16];    // column-sum partials
    int w = blockIdx.x;
    if (w >= a.nw) return;
    if (w >= a.nw_ps) {
        nplda::cohort_qz_block(a.qz, w - a.nw_ps, red);
        return;
    }
    if (w >= a.nw_mm) {  // pair-sum block of k-group ks: ext[ks][0 / 1][f] = sum of its share of K-A's per-block sums
        const int ks = w - a.nw_mm;
        const int per = (a.nblk + a.ksplit - 1) / a.ksplit;
        const int b0 = ks * per, b1 = b0 + per < a.nblk ? b0 + per : a.nblk;
        for (int i = threadIdx.x; i < 2 * a.Mp; i += 256) {
            const int row = i / a.Mp, f = i - row * a.Mp;
            float sum = 0.f;
            for (int b = b0; b < b1; ++b) sum += a.pq[((size_t)b * 2 + row) * a.ldz + f];
            a.ext[(size_t)ks * 4 * a.Mp + row * a.Mp + f] = sum;
        }
        return;
    }
    const int pi = w >= a.nw0 ? 1 : 0;
    if (pi) w -= a.nw0;
    const WgradProblem P = pi ? a.p[1] : a.p[0];
    const int nt = (w / a.ksplit) % P.NT;
    const int ext = (nt == 0) ? P.extras : 0;  // block-uniform: one specialised body per block
    if (ext == 0) wgrad_body<0>(a, P, w, red, rede);
    else if (ext == 1) wgrad_body<1>(a, P, w, red, rede);
    else if (ext == 2) wgrad_body<2>(a, P, w, red, rede);
    else if (ext == 3) wgrad_body<3>(a, P, w, red, rede);
    else wgrad_body<4>(a, P, w, red, rede);
}

constexpr long long kFmMaxRows = 1LL << 40;  // (round 2 stopped at 32 768 rows; the full-M strips also win at streaming sizes:
                                             // no padding of M, and the strips of one k-group share their A rows through L2)

static inline bool wgrad_fm_rows(long long K, int NB) { return NB >= 10 && NB <= 12 && K >= 4 && K <= kFmMaxRows; }
// streaming sizes: 64-column strips (NB = 12 keeps the 32-column form: 192 accumulator registers + the ring do not fit 256)
static inline bool fm_wide(long long K, int NB) { return K > 32 * 1024 && NB <= 11; }

// Can the training step's weight-gradient launch read the batch's own bfloat16 x rows (full-M form, 32-column strips)?
// Mirrors the conditions of wgrad_launch for the step's two problems (K = 2 B rows, split at B).
static inline bool wgrad_fm_bf16_ok(long long B, long long ldx, const NpldaLayout& L) {
    const long long K = 2 * B;
    return wgrad_fm_rows(K, L.NB) && !fm_wide(K, L.NB) && (B % 4) == 0 && (ldx % 2) == 0 && ldx * 4 < (1 << 20) &&
           (L.D0 % 2) == 0;
}

// The weight-gradient launch: full-M form where it applies, the 64 x 64 form otherwise.  nprob: problems in use (1 or 2).
// tail: the training step's loss tail, to ride along if the full-M form runs (*tail_done reports it).
static int wgrad_launch(WgradArgs& wa, int NB, int nprob, hipStream_t st, const LossTail* tail = nullptr,
                        bool* tail_done = nullptr) {
    if (tail_done) *tail_done = false;
    bool fm = wgrad_fm_rows(wa.K, NB) && (wa.K % 4) == 0 && (wa.nsplit % 4) == 0 && wa.nw == wa.nw_ps;
    for (int i = 0; i < nprob; ++i) {
        const WgradProblem& P = wa.p[i];
        fm = fm && P.extras != 2 && P.Mp == 16 * NB && P.M == P.Mp && P.lda >= P.Mp && (P.lda % 4) == 0 && (P.ldb % 2) == 0 &&
             (P.N % 2) == 0 && (P.Np % 2) == 0 && P.lda * 4 < (1 << 20) && P.ldb * 4 < (1 << 20);
    }
    if (!fm) {
        if (wa.p[0].b_bf16) return NPLDA_EUNSUPPORTED;
        hipLaunchKernelGGL(wgrad_kernel, dim3((unsigned)wa.nw), dim3(256), 0, st, wa);
        return nplda_launch_status();
    }
    // streaming sizes: 64-column strips (11 tiles x 23 k-groups at D0 = 512, NB = 10); minibatch sizes: 32-column strips
    const bool wide = fm_wide(wa.K, NB);
    const int sw = wide ? 64 : 32;
    WgradFmArgs fa = {};
    fa.w = wa;
    fa.nt0 = (wa.p[0].N + sw - 1) / sw;
    fa.nt1 = nprob > 1 ? (wa.p[1].N + sw - 1) / sw : 0;
    const int tiles = fa.nt0 + fa.nt1;
    fa.ps_cols = wa.pq ? (2 * wa.Mp + tiles - 1) / tiles : 0;
    if (fa.ps_cols > kFmWaves * 64) return NPLDA_EUNSUPPORTED;
    if (tail) {
        fa.has_tail = 1;
        fa.tail = *tail;
        if (tail_done) *tail_done = true;
    }
    fa.xcd_groups = (wide && !tail) ? 1 : 0;
    const dim3 grid((unsigned)(tiles * wa.ksplit + (tail ? 1 : 0))), block(kFmWaves * 64);
    const bool bbf = wa.p[0].b_bf16 != 0;
    if (bbf && (wide || nprob < 2)) return NPLDA_EUNSUPPORTED;  // (callers ask wgrad_fm_bf16_ok first)
#define NPLDA_FM(NBV)                                                                        \
    if (wide) hipLaunchKernelGGL((wgrad_fm_kernel<NBV, (NBV <= 10 ? kFmPF : 3), 4>), grid, block, 0, st, fa);  \
    else if (bbf) hipLaunchKernelGGL((wgrad_fm_kernel<NBV, kFmPF, 2, true>), grid, block, 0, st, fa);           \
    else hipLaunchKernelGGL((wgrad_fm_kernel<NBV, kFmPF, 2>), grid, block, 0, st, fa)
    switch (NB) {
        case 10: NPLDA_FM(10); break;
        case 11: NPLDA_FM(11); break;
        default: NPLDA_FM(12); break;
    }
#undef NPLDA_FM
    return nplda_launch_status();
}

// ------------------------------------------------------------------------------------------------
// K-C : slabs -> flat gradient [dW1 (D1 x D0) | db1 | dW2 (D2 x D1) | db2 | dP_sqrt | dQ]
// ------------------------------------------------------------------------------------------------
struct ReduceArgs {
    const float* slab1;  // [ksplit][Mp][D0p]   dW1
    const float* slab2;  // [ksplit][Mp][Mp]    dW2
    const float* ext;    // [ksplit][4][Mp]
    const float* P_sqrt;
    int ksplit, Mp, Np1, D0, D1, D2;
    float* out;
};

__global__ __launch_bounds__(256) void reduce_kernel(const ReduceArgs a) {
    const size_t nW1 = (size_t)a.D1 * a.D0, nW2 = (size_t)a.D2 * a.D1;
    const size_t total = nW1 + a.D1 + nW2 + 3 * (size_t)a.D2;
    const size_t idx = (size_t)blockIdx.x * 256 + threadIdx.x;
    if (idx >= total) return;
    const float* src;
    size_t stride;
    float scale = 1.0f;
    if (idx < nW1) {
        const size_t m = idx / a.D0, n = idx % a.D0;
        src = a.slab1 + m * a.Np1 + n;
        stride = (size_t)a.Mp * a.Np1;
    } else if (idx < nW1 + a.D1) {
        src = a.ext + 3 * a.Mp + (idx - nW1);
        stride = 4 * (size_t)a.Mp;
    } else if (idx < nW1 + a.D1 + nW2) {
        const size_t r = idx - nW1 - a.D1;
        const size_t m = r / a.D1, n = r % a.D1;
        src = a.slab2 + m * a.Mp + n;
        stride = (size_t)a.Mp * a.Mp;
    } else {
        const size_t r = idx - nW1 - a.D1 - nW2;
        const int which = (int)(r / a.D2), f = (int)(r % a.D2);
        // which: 0 = db2 (ext row 2), 1 = dP_sqrt (ext row 1, times 4 P_sqrt), 2 = dQ (ext row 0)
        const int row = which == 0 ? 2 : (which == 1 ? 1 : 0);
        src = a.ext + row * a.Mp + f;
        stride = 4 * (size_t)a.Mp;
        if (which == 1) scale = a.P_sqrt ? 4.0f * a.P_sqrt[f] : 0.f;  // no P_sqrt: embedding rows, dP = 0
    }
    float sum = 0.f;
    for (int k = 0; k < a.ksplit; ++k) sum += src[k * stride];
    a.out[idx] = sum * scale;
}

// ------------------------------------------------------------------------------------------------
// K-D : the tail of the fused training step in one launch — K-C's slab sums, torch.optim.Adam's update of the six
// parameter tensors (xvector_NeuralPlda_pytorch.py:139), the refreshed fragment image of the updated parameters (what
// nplda_pack_params_f32 would rebuild at the start of the next step), and in the last block the loss, dL/dtheta and
// the thresholds' own Adam update from the per-block loss partials K-A left.
// ------------------------------------------------------------------------------------------------
struct UpdateArgs {
    ReduceArgs r;             // r.out: optional copy of the flat gradient (+ dtheta), may be null
    float* prm[6];            // W1, b1, W2, b2, P_sqrt, Q
    float* m;                 // exp_avg    [ngrad + K], flat-gradient order then the thresholds
    float* v;                 // exp_avg_sq
    float* step;              // [steps taken, arrival ticket]  (as nplda_adam_step_f32)
    float lr, beta1, beta2, eps, wd;
    NpldaLayout L;
    float* packed;
    int bumped;               // step[0] already counts this step (train_fb_small_kernel): no arrival tickets
    LossTail tail;            // the loss / threshold tail (nplda_loss_tail.h),
    int tail_here;            // done by this kernel's last block (0: it rode in the weight-gradient launch)
    const long long* cursor;  // optional (nplda_train_step_records_f32): [address of record 0, next record, record count]
    char* stage;              //   the record the NEXT step trains on is copied here by `ncopy` extra blocks
    long long rec_bytes;
    unsigned ncopy;
    unsigned ngrad_blocks;
    // data parallel (nplda_train_step_grad_f32 / nplda_train_step_apply_f32)
    int grad_only;            // write the flat gradient to r.out and stop: no Adam, no parameter or image store
    const float* flat;        // take the gradient from here (the all-reduced flat gradient) instead of the slabs
};

// position of W[f][k] in a fragment image [k / 16][f / 16][lane = 16 ((k % 16) / 4) + f % 16][k % 4]
__device__ __forceinline__ size_t frag_pos(int f, int k, int NB) {
    return ((((size_t)(k >> 4) * NB + (f >> 4)) * 64 + (((k & 15) >> 2) << 4) + (f & 15)) << 2) + (k & 3);
}

template <int E>  // elements per thread
__global__ __launch_bounds__(256) void train_update_kernel(const UpdateArgs a) {
    const float t = a.bumped ? a.step[0] : a.step[0] + 1.0f;
    const nplda_adam::Consts c = nplda_adam::consts_for(t, a.lr, a.beta1, a.beta2, a.eps, a.wd);
    const int D0 = a.r.D0, D1 = a.r.D1, D2 = a.r.D2;
    const size_t nW1 = (size_t)D1 * D0, nW2 = (size_t)D2 * D1;
    const size_t ngrad = nW1 + D1 + nW2 + 3 * (size_t)D2;
    if (blockIdx.x >= gridDim.x - a.ncopy) {  // stage the epoch's next record (the first kernel has counted this one)
        const long long k = a.cursor[1];
        if (k >= a.cursor[2]) return;
        const f32x4* src = reinterpret_cast<const f32x4*>(reinterpret_cast<const char*>(a.cursor[0]) + k * a.rec_bytes);
        f32x4* dst = reinterpret_cast<f32x4*>(a.stage);
        const long long n16 = a.rec_bytes / 16;
        for (long long i = (long long)(blockIdx.x - (gridDim.x - a.ncopy)) * 256 + threadIdx.x; i < n16; i += 256LL * a.ncopy)
            dst[i] = src[i];
        return;
    }
    if (blockIdx.x < a.ngrad_blocks) {
        // four elements per thread: the arrival tickets at the end are one atomic per block on one address (~10 ns each,
        // serialised): 392 blocks of 256 elements spent 2 of the kernel's 10 us queueing there.
        // Everything the thread's E elements need from memory — up to 16 slab values, the parameter and the two moments
        // of each — is asked for BEFORE the first sum: as one element after the other (slabs -> sum -> moments -> stores,
        // the next element's loads behind this element's stores, which they might alias) a thread made 2 E dependent memory
        // round trips, 5 of the kernel's 6.8 us (round 5).
        const float* src[E];
        size_t stride[E], pk0[E], pk1[E], idxv[E];
        float* pp[E];
        int which[E];  // 1: P_sqrt
        bool live[E];
#pragma unroll
        for (int e4 = 0; e4 < E; ++e4) {
            const size_t idx_raw = ((size_t)blockIdx.x * E + e4) * 256 + threadIdx.x;
            live[e4] = idx_raw < ngrad;
            const size_t idx = live[e4] ? idx_raw : ngrad - 1;  // (always-valid addresses; nothing is stored for a dead slot)
            idxv[e4] = idx;
            pk1[e4] = (size_t)-1;
            which[e4] = 0;
            if (idx < nW1) {
                const int f = (int)(idx / D0), k = (int)(idx % D0);
                src[e4] = a.r.slab1 + (size_t)f * a.r.Np1 + k;
                stride[e4] = (size_t)a.r.Mp * a.r.Np1;
                pp[e4] = a.prm[0] + idx;
                pk0[e4] = a.L.oW1 + frag_pos(f, k, a.L.NB);
                pk1[e4] = a.L.oW1T + frag_pos(k, f, a.L.KS1);  // W1^T image (dx = du W1): rows and columns change places
            } else if (idx < nW1 + D1) {
                const int f = (int)(idx - nW1);
                src[e4] = a.r.ext + 3 * a.r.Mp + f;
                stride[e4] = 4 * (size_t)a.r.Mp;
                pp[e4] = a.prm[1] + f;
                pk0[e4] = a.L.ob1 + f;
            } else if (idx < nW1 + D1 + nW2) {
                const size_t rr = idx - nW1 - D1;
                const int f = (int)(rr / D1), k = (int)(rr % D1);
                src[e4] = a.r.slab2 + (size_t)f * a.r.Mp + k;
                stride[e4] = (size_t)a.r.Mp * a.r.Mp;
                pp[e4] = a.prm[2] + rr;
                pk0[e4] = a.L.oW2 + frag_pos(f, k, a.L.NB);
                pk1[e4] = a.L.oW2T + frag_pos(k, f, a.L.NB);  // W2^T image: rows and columns change places
            } else {
                const size_t rr = idx - nW1 - D1 - nW2;
                const int sel = (int)(rr / D2), f = (int)(rr % D2);
                // sel: 0 = b2 (ext row 2), 1 = P_sqrt (ext row 1, times 4 P_sqrt), 2 = Q (ext row 0)
                const int row = sel == 0 ? 2 : (sel == 1 ? 1 : 0);
                src[e4] = a.r.ext + row * a.r.Mp + f;
                stride[e4] = 4 * (size_t)a.r.Mp;
                pp[e4] = a.prm[3 + sel] + f;
                pk0[e4] = (sel == 0 ? a.L.ob2 : (sel == 1 ? a.L.oP : a.L.oQ)) + f;
                which[e4] = sel == 1;
            }
        }
        float pv[E], mv[E], vv[E], gv[E], part[E][16];
#pragma unroll
        for (int e4 = 0; e4 < E; ++e4) {
            pv[e4] = *pp[e4];
            mv[e4] = a.grad_only ? 0.f : a.m[idxv[e4]];
            vv[e4] = a.grad_only ? 0.f : a.v[idxv[e4]];
            if (a.flat) {
                gv[e4] = a.flat[idxv[e4]];
            } else {
                // at most 16 slabs (ws_layout): all of them in flight at once, summed in slab order like K-C
#pragma unroll
                for (int k = 0; k < 16; ++k) part[e4][k] = src[e4][(k < a.r.ksplit ? k : 0) * stride[e4]];
            }
        }
#pragma unroll
        for (int e4 = 0; e4 < E; ++e4) {
            if (!a.flat) {
                float sum = 0.f;
#pragma unroll
                for (int k = 0; k < 16; ++k) sum += k < a.r.ksplit ? part[e4][k] : 0.f;
                // the product is rounded on its own, as K-C stores it: left to the compiler it is contracted into Adam's g + wd p
                float g = which[e4] ? sum * (4.0f * pv[e4]) : sum;
                asm volatile("" : "+v"(g));
                gv[e4] = g;
            }
        }
#pragma unroll
        for (int e4 = 0; e4 < E; ++e4) {
            if (!live[e4]) continue;
            const size_t idx = idxv[e4];
            if (a.r.out) a.r.out[idx] = gv[e4];
            if (a.grad_only) continue;
            float m = mv[e4], v = vv[e4];
            const float pn = nplda_adam::update(pv[e4], gv[e4], m, v, c);
            a.m[idx] = m;
            a.v[idx] = v;
            *pp[e4] = pn;
#ifndef NPLDA_EXP_NO_IMAGE_STORES  // (tools/trace_update.sh: what the scattered image stores cost)
            a.packed[pk0[e4]] = which[e4] ? pn * pn : pn;  // P = P_sqrt^2 (utils/models.py:373)
            if (pk1[e4] != (size_t)-1) a.packed[pk1[e4]] = pn;
#endif
        }
    } else {
        __shared__ double tail_smem[(kLossTailSmem + 7) / 8];
        loss_tail_block(a.tail, tail_smem);
    }
    if (a.bumped) return;
    __syncthreads();  // the whole block has read step[0]
    if (threadIdx.x == 0) {
        unsigned* ticket = reinterpret_cast<unsigned*>(a.step + 1);
        if (atomicAdd(ticket, 1u) == gridDim.x - 1) {
            a.step[0] = t;
            *ticket = 0u;
        }
    }
}

// ------------------------------------------------------------------------------------------------
constexpr int kBwdWaves = 4;

struct WsLayout {
    size_t dz, du, slab1, slab2, ext, pq, frag, total;  // float offsets
    int ksplit;
    long long rows_per_split;
    int Mp, Np1;
};

// persistent grid of bwd_data_stream_kernel: one 8-wave block per CU (each wave walks 16-pair tiles)
static inline long long stream_grid(long long nA) {
    int dev = 0, cus = 0;
    if (hipGetDevice(&dev) != hipSuccess ||
        hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, dev) != hipSuccess || cus <= 0)
        cus = 256;
    const long long need = (nA + 127) / 128;
    return need < cus ? (need > 0 ? need : 1) : cus;
}

// K = rows of the A^T B products (2 B for pair scoring).  want_dx adds room for the W1 fragment image of dx = du W1.
WsLayout ws_layout(long long K, const NpldaLayout& L, bool want_dx) {
    WsLayout w;
    w.Mp = 16 * L.NB;
    w.Np1 = (L.D0 + 3) / 4 * 4;
    // k-groups (slabs): each group = one block of 4 waves x >= 128 rows.  A block holds 68 KB of LDS, so 2 are
    // resident per CU: keep (64x64 tiles) x (k-groups) <= 512 blocks so that the grid is a single resident wave
    // (528 blocks ran as 512 + a 16-block tail at twice the time).
    const long long tiles = (long long)((w.Mp + 63) / 64) * ((L.D0 + 63) / 64) + (long long)((w.Mp + 63) / 64) * ((w.Mp + 63) / 64);
    long long ks = (K + 511) / 512;
    if (ks > 16) ks = 16;
    if (ks * tiles > 512) ks = 512 / tiles;
    if (wgrad_fm_rows(K, L.NB)) {
        // full-M form: (32-column tiles) x (k-groups) 8-wave blocks, ONE per CU; every wave at least one k4-step
        const int sw = fm_wide(K, L.NB) ? 64 : 32;
        const long long tiles32 = (L.D0 + sw - 1) / sw + (w.Mp + sw - 1) / sw;
        ks = 256 / tiles32;
        if (ks > 16 && !fm_wide(K, L.NB)) ks = 16;
        if (ks > (K / 4 + kFmWaves - 1) / kFmWaves) ks = (K / 4 + kFmWaves - 1) / kFmWaves;
    }
    if (ks < 1) ks = 1;
    long long rps = (K + ks - 1) / ks;
    rps = (rps + 16 * kPF - 1) / (16 * kPF) * (16 * kPF);  // 4 quarters, each a multiple of 4 * kPF rows
    if (rps < 16 * kPF) rps = 16 * kPF;
    w.ksplit = (int)((K + rps - 1) / rps);
    if (w.ksplit < 1) w.ksplit = 1;
    w.rows_per_split = rps;
    const size_t rows = (size_t)(K + (K & 1)) * w.Mp;  // an odd row count is padded by one (never-read) row
    w.dz = 0;
    w.du = w.dz + rows;
    w.slab1 = w.du + rows;
    w.slab2 = w.slab1 + (size_t)w.ksplit * w.Mp * w.Np1;
    w.ext = w.slab2 + (size_t)w.ksplit * w.Mp * w.Mp;
    w.pq = w.ext + (size_t)w.ksplit * 4 * w.Mp;
    w.frag = w.pq + (size_t)(2 * ((K / 2 + 15) / 16)) * 2 * w.Mp;  // small-batch pair scoring: K / 2 pairs in tiles of 16, or two half tiles each
    w.total = w.frag + (want_dx ? (size_t)L.NB * L.KS1 * 256 : 0);
    return w;
}

// The three launches (+ the input-gradient GEMM when dx is asked for) behind every backward entry point.
//   given == false: pair scoring — dz is formed from g, z (rows [0,B) = x1 side, [B,2B) = x2 side);
//   given == true : `K` embedding rows whose upstream gradient dL/dz already sits, zero-padded, in the workspace.
int backward_launch(bool given, const float* xa, const float* xb, long long K, long long nsplit, long long ldx,
                    const float* packed, const NpldaLayout& L, const float* g, const float* y, const float* z,
                    const float* rn, long long ldz, const float* P_sqrt, float* wsf, const WsLayout& W, float* grad_flat,
                    float* dx0, float* dx1, long long lddx, hipStream_t st, const BwdLoss* ls = nullptr,
                    ReduceArgs* defer_reduce = nullptr, bool data_done = false, const LossTail* tail = nullptr,
                    bool* tail_done = nullptr, bool x_bf16 = false, int pq_rows = -1) {
    BwdArgs b = {};
    if (ls) {
        if (given || nsplit > 16 * 1024) return NPLDA_EUNSUPPORTED;
        b.ls = *ls;
    }
    b.g = g; b.z = z; b.y = y; b.rn = rn; b.packed = packed; b.ldz = ldz;
    if (given) {
        b.nA = (K + 1) / 2; b.nB = K - b.nA; b.offB = b.nB > 0 ? b.nA : 0;
    } else {
        b.nA = b.nB = b.offB = nsplit;
    }
    b.oW2T = L.oW2T; b.oQ = L.oQ; b.oP = L.oP; b.total = L.total;
    b.dz = wsf + W.dz; b.du = wsf + W.du;
    // K-A leaves the dQ / dP pair sums per block: the small-batch kernel one row per 16-pair tile, the streaming kernel
    // (NB >= 8) one row per persistent block
    const long long stream_blocks = stream_grid(b.nA);
    const bool pair_sums = !given && (b.nA <= 16 * 1024 || L.NB >= 8);
    b.pq = pair_sums ? wsf + W.pq : nullptr;
    const long long ntb = (b.nA + 16 * kBwdWaves - 1) / (16 * kBwdWaves);
    if (ntb > 0x7fffffffLL) return NPLDA_EINVAL;
    b.ntb = (int)ntb;
    if (!data_done) {  // (data_done: train_fb_small_kernel has already left dz, du, the pair sums and the loss partials)
        // small batches: 4 waves share a 16-pair tile (feature split), so a 4096-pair minibatch fills 256 CUs
        const bool small = b.nA <= 16 * 1024;
        const bool stream = !small && L.NB >= 8;
        dim3 grid(small ? (unsigned)((b.nA + 15) / 16) : (stream ? (unsigned)stream_blocks : (unsigned)(ntb < 2048 ? ntb : 2048))),
            block(stream ? 512 : 256);
#define NPLDA_LAUNCH2(NBV, GV)                                                                  \
    if (small) hipLaunchKernelGGL((bwd_data_small_kernel<NBV, GV>), grid, block, 0, st, b);      \
    else if (NBV >= 8) hipLaunchKernelGGL((bwd_data_stream_kernel<(NBV >= 8 ? NBV : 8), GV>), grid, block, 0, st, b);   \
    else hipLaunchKernelGGL((bwd_data_kernel<NBV, kBwdWaves, GV>), grid, block, 0, st, b)
#define NPLDA_LAUNCH(NBV)                                                                        \
    if (ls) hipLaunchKernelGGL((bwd_data_small_kernel<NBV, false, true>), grid, block, 0, st, b); \
    else if (given) { NPLDA_LAUNCH2(NBV, true); } else { NPLDA_LAUNCH2(NBV, false); }
        switch (L.NB) {
            case 2: NPLDA_LAUNCH(2); break;
            case 4: NPLDA_LAUNCH(4); break;
            case 8: NPLDA_LAUNCH(8); break;
            case 10: NPLDA_LAUNCH(10); break;
            case 11: NPLDA_LAUNCH(11); break;
            case 12: NPLDA_LAUNCH(12); break;
            default: return NPLDA_EUNSUPPORTED;
        }
#undef NPLDA_LAUNCH
#undef NPLDA_LAUNCH2
        if (int rc = nplda_launch_status()) return rc;
    }
    // K-B
    WgradArgs wa = {};
    wa.K = K; wa.nsplit = nsplit; wa.ksplit = W.ksplit; wa.rows_per_split = W.rows_per_split;
    wa.z = z; wa.g = ls ? ls->g_out : g; wa.ldz = ldz; wa.ext = wsf + W.ext; wa.Mp = W.Mp;
    WgradProblem& p1 = wa.p[0];  // dW1 = du^T [x1; x2]
    p1.A = wsf + W.du; p1.lda = ldz; p1.B0 = xa; p1.B1 = xb; p1.ldb = ldx; p1.M = W.Mp; p1.N = L.D0;
    p1.MT = (W.Mp + 63) / 64; p1.NT = (L.D0 + 63) / 64; p1.slab = wsf + W.slab1; p1.Mp = W.Mp; p1.Np = W.Np1; p1.extras = 1;
    p1.b_bf16 = x_bf16 ? 1 : 0;  // (xa / xb then point at bfloat16 rows, ldx counts elements)
    WgradProblem& p2 = wa.p[1];  // dW2 = dz^T [y1; y2]
    p2.A = wsf + W.dz; p2.lda = ldz; p2.B0 = y; p2.B1 = y + (size_t)nsplit * ldz; p2.ldb = ldz; p2.M = W.Mp; p2.N = W.Mp;
    p2.MT = (W.Mp + 63) / 64; p2.NT = (W.Mp + 63) / 64; p2.slab = wsf + W.slab2; p2.Mp = W.Mp; p2.Np = W.Mp;
    p2.extras = given ? 3 : (pair_sums ? 4 : 2);
    wa.nw0 = p1.MT * p1.NT * W.ksplit;
    wa.nw_mm = wa.nw0 + p2.MT * p2.NT * W.ksplit;
    wa.nw = wa.nw_ps = wa.nw_mm + (pair_sums ? W.ksplit : 0);
    // (pq_rows: rows of pair sums the data kernel left when it was not one per 16-pair tile — the half-tile kernels)
    wa.pq = b.pq; wa.nblk = pq_rows >= 0 ? pq_rows : (b.nA <= 16 * 1024 ? (int)((b.nA + 15) / 16) : (int)stream_blocks);
    if (int rc = wgrad_launch(wa, L.NB, 2, st, tail, tail_done)) return rc;
    // K-C
    ReduceArgs ra = {};
    ra.slab1 = wsf + W.slab1; ra.slab2 = wsf + W.slab2; ra.ext = wsf + W.ext; ra.P_sqrt = P_sqrt;
    ra.ksplit = W.ksplit; ra.Mp = W.Mp; ra.Np1 = W.Np1; ra.D0 = L.D0; ra.D1 = L.D1; ra.D2 = L.D2; ra.out = grad_flat;
    const size_t ngrad = (size_t)L.D1 * L.D0 + L.D1 + (size_t)L.D2 * L.D1 + 3 * (size_t)L.D2;
    if (defer_reduce) {  // the caller's update kernel sums the slabs itself
        *defer_reduce = ra;
        return NPLDA_OK;
    }
    hipLaunchKernelGGL(reduce_kernel, dim3((unsigned)((ngrad + 255) / 256)), dim3(256), 0, st, ra);
    if (int rc = nplda_launch_status()) return rc;
    if (dx0) {  // dL/dx = du . W1 (the E2E head's input gradient, utils/models.py:251-268)
        return input_grad_from_du(wsf + W.du, K, ldz, packed, L, wsf + W.frag, dx0, dx1, nsplit, lddx, st);
    }
    return NPLDA_OK;
}

}  // namespace

namespace nplda {

// Gram matrix of a row table for nplda_cohort_fused.hip: slab[ks][i][j] = sum over the rows of k-group ks of
// Z[k][i] Z[k][j] (i, j < Mp = the padded row width, a multiple of 16) and ext[ks][3][i] = the column sums, with the
// split-K "A^T B" kernel above (A = B = Z).  The caller sums the ksplit slabs in a fixed order.
int gram_slabs_launch(const float* Z, long long ldz, long long rows, int Mp, int ksplit, float* slab, float* ext,
                      const QzArgs* qz, hipStream_t st) {
    if (rows <= 0 || ksplit < 1 || (Mp % 16) != 0) return NPLDA_EINVAL;
    long long rps = (rows + ksplit - 1) / ksplit;
    rps = (rps + 16 * kPF - 1) / (16 * kPF) * (16 * kPF);
    WgradArgs wa = {};
    wa.K = rows; wa.nsplit = rows; wa.ksplit = ksplit; wa.rows_per_split = rps;
    wa.ldz = ldz; wa.ext = ext; wa.Mp = Mp;
    WgradProblem& p1 = wa.p[0];
    p1.A = Z; p1.lda = ldz; p1.B0 = Z; p1.B1 = Z; p1.ldb = ldz; p1.M = Mp; p1.N = Mp;
    p1.MT = (Mp + 63) / 64; p1.NT = (Mp + 63) / 64; p1.slab = slab; p1.Mp = Mp; p1.Np = Mp; p1.extras = 1;
    wa.p[1] = p1;
    wa.nw0 = p1.MT * p1.NT * ksplit;
    wa.nw = wa.nw_mm = wa.nw_ps = wa.nw0;
    if (qz) {
        wa.qz = *qz;
        wa.nw += qz->nblocks;
    }
    hipLaunchKernelGGL(wgrad_kernel, dim3((unsigned)wa.nw), dim3(256), 0, st, wa);
    return nplda_launch_status();
}

}  // namespace nplda

extern "C" {

size_t nplda_grad_floats(int D0, int D1, int D2) {
    if (check_model(D0, D1, D2) != NPLDA_OK) return 0;
    return (size_t)D1 * D0 + D1 + (size_t)D2 * D1 + 3 * (size_t)D2;
}

size_t nplda_backward_workspace_bytes(int64_t B, int D0, int D1, int D2) {
    if (B < 0 || check_model(D0, D1, D2) != NPLDA_OK) return 0;
    return ws_layout(2 * B, nplda_layout(D0, D1, D2), false).total * sizeof(float);
}

size_t nplda_backward_ex_workspace_bytes(int64_t rows, int D0, int D1, int D2, int want_dx) {
    if (rows < 0 || check_model(D0, D1, D2) != NPLDA_OK) return 0;
    return ws_layout(rows, nplda_layout(D0, D1, D2), want_dx != 0).total * sizeof(float);
}

int nplda_forward_train_f32(const float* x1, const float* x2, int64_t B, int64_t ldx, const void* packed, int D0,
                            int D1, int D2, float* s, float* y, float* z, float* rn, int64_t ldz,
                            nplda_stream_t stream) {
    if (B < 0) return NPLDA_EINVAL;
    if (int rc = check_model(D0, D1, D2)) return rc;
    if (B == 0) return NPLDA_OK;
    if (!packed || !s || !rn || !nplda_aligned16(packed)) return NPLDA_EINVAL;
    if (!rows_ok(x1, ldx, D0) || !rows_ok(x2, ldx, D0)) return NPLDA_EINVAL;
    const NpldaLayout L = nplda_layout(D0, D1, D2);
    if (!rows_ok(y, ldz, 16 * L.NB) || !rows_ok(z, ldz, 16 * L.NB)) return NPLDA_EINVAL;
    FwdArgs a = {};
    a.xa = x1; a.xb = x2; a.n = B; a.ldx = ldx; a.packed = (const float*)packed;
    a.out_s = s; a.out_z = z; a.ldz = ldz; a.out_y = y; a.out_rn = rn;
    return launch_fwd<MODE_TRAIN>(a, L, (hipStream_t)stream);
}

int nplda_embed_train_f32(const float* x, int64_t N, int64_t ldx, const void* packed, int D0, int D1, int D2, float* z,
                          float* y, float* rn, int64_t ldz, nplda_stream_t stream) {
    if (N < 0) return NPLDA_EINVAL;
    if (int rc = check_model(D0, D1, D2)) return rc;
    if (N == 0) return NPLDA_OK;
    if (!packed || !rn || !nplda_aligned16(packed) || !rows_ok(x, ldx, D0)) return NPLDA_EINVAL;
    const NpldaLayout L = nplda_layout(D0, D1, D2);
    if (!rows_ok(y, ldz, 16 * L.NB) || !rows_ok(z, ldz, 16 * L.NB)) return NPLDA_EINVAL;
    FwdArgs a = {};
    a.xa = x; a.xb = x; a.n = N; a.ldx = ldx; a.packed = (const float*)packed;
    a.out_z = z; a.ldz = ldz; a.out_y = y; a.out_rn = rn;
    return launch_fwd<MODE_EMBED>(a, L, (hipStream_t)stream);
}

int nplda_backward_ex_f32(const float* x1, const float* x2, int64_t B, int64_t ldx, const void* packed, int D0, int D1,
                          int D2, const float* g, const float* y, const float* z, const float* rn, int64_t ldz,
                          const float* P_sqrt, void* ws, size_t ws_bytes, float* grad_flat, float* dx1, float* dx2,
                          int64_t lddx, nplda_stream_t stream) {
    if (B < 0) return NPLDA_EINVAL;
    if (int rc = check_model(D0, D1, D2)) return rc;
    if (!grad_flat) return NPLDA_EINVAL;
    if ((dx1 == nullptr) != (dx2 == nullptr)) return NPLDA_EINVAL;
    hipStream_t st = (hipStream_t)stream;
    const NpldaLayout L = nplda_layout(D0, D1, D2);
    const size_t ngrad = nplda_grad_floats(D0, D1, D2);
    if (B == 0) {
        hipError_t e = hipMemsetAsync(grad_flat, 0, ngrad * sizeof(float), st);
        return e == hipSuccess ? NPLDA_OK : (int)e;
    }
    if (!packed || !g || !rn || !P_sqrt || !ws || !nplda_aligned16(packed) || !nplda_aligned16(ws)) return NPLDA_EINVAL;
    if (!rows_ok(x1, ldx, D0) || !rows_ok(x2, ldx, D0)) return NPLDA_EINVAL;
    if (!rows_ok(y, ldz, 16 * L.NB) || !rows_ok(z, ldz, 16 * L.NB) || ldz != 16 * L.NB) return NPLDA_EINVAL;
    if (dx1 && (!rows_ok(dx1, lddx, D0) || !rows_ok(dx2, lddx, D0))) return NPLDA_EINVAL;
    const WsLayout W = ws_layout(2 * B, L, dx1 != nullptr);
    if (ws_bytes < W.total * sizeof(float)) return NPLDA_ENOSPC;
    return backward_launch(false, x1, x2, 2 * B, B, ldx, (const float*)packed, L, g, y, z, rn, ldz, P_sqrt, (float*)ws, W,
                           grad_flat, dx1, dx2, lddx, st);
}

int nplda_backward_f32(const float* x1, const float* x2, int64_t B, int64_t ldx, const void* packed, int D0, int D1,
                       int D2, const float* g, const float* y, const float* z, const float* rn, int64_t ldz,
                       const float* P_sqrt, void* ws, size_t ws_bytes, float* grad_flat, nplda_stream_t stream) {
    return nplda_backward_ex_f32(x1, x2, B, ldx, packed, D0, D1, D2, g, y, z, rn, ldz, P_sqrt, ws, ws_bytes, grad_flat,
                                 nullptr, nullptr, 0, stream);
}

int nplda_embed_backward_f32(const float* x, int64_t N, int64_t ldx, const void* packed, int D0, int D1, int D2,
                             const float* gz, int64_t ldgz, const float* y, const float* rn, int64_t ldz, void* ws,
                             size_t ws_bytes, float* grad_flat, float* dx, int64_t lddx, nplda_stream_t stream) {
    if (N < 0) return NPLDA_EINVAL;
    if (int rc = check_model(D0, D1, D2)) return rc;
    if (!grad_flat) return NPLDA_EINVAL;
    hipStream_t st = (hipStream_t)stream;
    const NpldaLayout L = nplda_layout(D0, D1, D2);
    const size_t ngrad = nplda_grad_floats(D0, D1, D2);
    if (N == 0) {
        hipError_t e = hipMemsetAsync(grad_flat, 0, ngrad * sizeof(float), st);
        return e == hipSuccess ? NPLDA_OK : (int)e;
    }
    if (!packed || !gz || !rn || !ws || ldgz < D2 || !nplda_aligned16(packed) || !nplda_aligned16(ws)) return NPLDA_EINVAL;
    if (!rows_ok(x, ldx, D0) || !rows_ok(y, ldz, 16 * L.NB) || ldz != 16 * L.NB) return NPLDA_EINVAL;
    if (dx && !rows_ok(dx, lddx, D0)) return NPLDA_EINVAL;
    const WsLayout W = ws_layout(N, L, dx != nullptr);
    if (ws_bytes < W.total * sizeof(float)) return NPLDA_ENOSPC;
    float* wsf = (float*)ws;
    if (int rc = pad_rows(gz, ldgz, N, D2, wsf + W.dz, ldz, st)) return rc;
    return backward_launch(true, x, x, N, N, ldx, (const float*)packed, L, nullptr, y, nullptr, rn, ldz, nullptr, wsf, W,
                           grad_flat, dx, dx, lddx, st);
}

// ---- the fused training step --------------------------------------------------------------------------------
namespace {
// When the first kernel runs on 8-pair half tiles (nplda_train_fb_half.h): where that gives the batch MORE blocks than it has
// CUs to put them on — up to one half tile per CU (B <= 2048 on 256 CUs: the reference's own batch sizes, conf/*.cfg
// batch_size 128 .. 2048, and every data-parallel shard of a 4096-pair batch).  Beyond that two half tiles share a CU, run in
// step and stream W1 twice through the CU's one vector-memory path: no faster than one 16-pair tile (profiles/r05b_exp_fbh.txt).
// NPLDA_FB_HALF = 0 never / 2 always (A/B measurements); NPLDA_FB_SKEW = "mode[,arg]": see HalfSkew.
static bool use_half_tiles(long long B) {
    static const int mode = [] { const char* e = getenv("NPLDA_FB_HALF"); return e ? atoi(e) : 1; }();
    if (mode == 0) return false;
    if (mode == 2) return true;
    return (B + kHalfPairs - 1) / kHalfPairs <= mid_cus();
}
static HalfSkew half_skew() {
    static const HalfSkew s = [] {
        HalfSkew k = {0, 0};
        if (const char* e = getenv("NPLDA_FB_SKEW")) {
            k.mode = atoi(e);
            const char* c = strchr(e, ',');
            k.arg = c ? atoi(c + 1) : 0;
        }
        return k;
    }();
    return s;
}
struct StepWs { size_t y, z, rn, s, g, partial, xs, bwd, total; long long ldz, ldxs; int nblk; };  // float offsets
static StepWs step_ws(long long B, const NpldaLayout& L, bool rows) {
    StepWs w;
    w.ldz = 16 * L.NB;
    w.nblk = 2 * (int)((B + 15) / 16);  // (room for the half-tile kernels' rows; the launcher says how many are in use)
    auto al = [](size_t v) { return (v + 63) / 64 * 64; };
    w.y = 0;
    w.z = al(w.y + (size_t)2 * B * w.ldz);
    w.rn = al(w.z + (size_t)2 * B * w.ldz);
    w.s = al(w.rn + (size_t)2 * B);
    w.g = al(w.s + (size_t)B);
    w.partial = al(w.g + (size_t)B);
    w.xs = al(w.partial + (size_t)w.nblk * kLossNS * 2);
    w.ldxs = L.D0;
    w.bwd = al(w.xs + (rows ? (size_t)2 * B * L.D0 : 0));  // indexed form: the gathered x1 / x2 rows
    w.total = w.bwd + ws_layout(2 * B, L, false).total;
    return w;
}
}  // namespace

size_t nplda_train_step_workspace_bytes(int64_t B, int D0, int D1, int D2) {
    if (B < 1 || B > 16 * 1024 || check_model(D0, D1, D2) != NPLDA_OK) return 0;
    return step_ws(B, nplda_layout(D0, D1, D2), false).total * sizeof(float);
}

size_t nplda_train_step_rows_workspace_bytes(int64_t B, int D0, int D1, int D2) {
    if (B < 1 || B > 16 * 1024 || (D0 % 16) != 0 || check_model(D0, D1, D2) != NPLDA_OK) return 0;
    return step_ws(B, nplda_layout(D0, D1, D2), true).total * sizeof(float);
}

static int train_step_impl(const float* x1, const float* x2, const int64_t* rows1, const int64_t* rows2, int64_t ntab,
                           long long* cursor, void* stage, int64_t B, int64_t ldx, const float* target,
                         float* const* params, int D0, int D1, int D2, float* const* thetas, const float* betas, int K,
                         float alpha, int kind, float* exp_avg, float* exp_avg_sq, float* step, float lr, float beta1,
                         float beta2, float eps, float weight_decay, void* packed, void* ws, size_t ws_bytes, float* loss,
                         double* loss_sum, float* grad_out, nplda_stream_t stream, void* dxa = nullptr, void* dxb = nullptr,
                         int64_t lddx = 0, bool io_bf16 = false, const double* gcount = nullptr, float* dp_flat = nullptr) {
    // dp_flat: the data-parallel gradient phase (nplda_train_step_grad_f32) — same three launches, but the last two stop at
    // this rank's flat gradient [ngrad | loss sums as kLossLimbs x kLossNS floats] and nothing is updated
    if (dp_flat) {
        if (cursor || grad_out) return NPLDA_EINVAL;
        grad_out = dp_flat;
    }
    if (cursor) {  // the batch sits in the staging record [rows1 | rows2 | labels]
        if (!stage || !nplda_aligned16(stage) || B < 1) return NPLDA_EINVAL;
        if ((B % 4) != 0) return NPLDA_EUNSUPPORTED;  // 16-byte pieces: the labels sit behind 16 B bytes of indices
        rows1 = (const int64_t*)stage;
        rows2 = rows1 + B;
        target = (const float*)(rows2 + B);
    }
    // staged form: the first kernel leaves fp32 copies of the x rows it read for the weight gradients — rows named by table
    // indices, or (io_bf16) the batch's own bfloat16 rows
    const bool rows = rows1 != nullptr || io_bf16;
    if (int rc = check_model(D0, D1, D2)) return rc;
    if (rows1 && (!rows2 || ntab < 1)) return NPLDA_EINVAL;
    if (rows && (D0 % 16) != 0) return NPLDA_EUNSUPPORTED;  // the staged rows are written k16-step by k16-step
    if (io_bf16 && (D0 != 512 || rows1)) return NPLDA_EUNSUPPORTED;
    if ((dxa == nullptr) != (dxb == nullptr) || (dxa && lddx < D0)) return NPLDA_EINVAL;
    if (B < 1) return NPLDA_EINVAL;
    if (B > 16 * 1024) return NPLDA_EUNSUPPORTED;  // larger batches: the separate launches (two-pass loss)
    if (kind != 0 && kind != 1) return kind == 2 ? NPLDA_EUNSUPPORTED : NPLDA_EINVAL;
    const int nth = kind == 1 ? 1 : K;
    if (nth < 1 || nth > nplda_loss::kMaxK || !thetas || !params || (kind == 0 && !betas)) return NPLDA_EINVAL;
    if (!target || !exp_avg || !exp_avg_sq || !step || !packed || !ws || !loss) return NPLDA_EINVAL;
    if (!nplda_aligned16(packed) || !nplda_aligned16(ws) || !nplda_aligned16(target)) return NPLDA_EINVAL;
    if (!rows_ok(x1, ldx, D0) || !rows_ok(x2, ldx, D0)) return NPLDA_EINVAL;
    for (int i = 0; i < 6; ++i)
        if (!params[i]) return NPLDA_EINVAL;
    const NpldaLayout L = nplda_layout(D0, D1, D2);
    const StepWs S = step_ws(B, L, rows);
    if (ws_bytes < S.total * sizeof(float)) return NPLDA_ENOSPC;
    hipStream_t st = (hipStream_t)stream;
    float* wsf = (float*)ws;

    BwdLoss ls = {};
    ls.s = wsf + S.s; ls.t = target; ls.K = nth; ls.kind = kind; ls.alpha = alpha; ls.B = B;
    ls.g_out = wsf + S.g; ls.partial = reinterpret_cast<double*>(wsf + S.partial);
    ls.gcount = gcount;
    UpdateArgs ua = {};
    for (int k = 0; k < nth; ++k) {
        if (!thetas[k]) return NPLDA_EINVAL;
        ls.th.p[k] = thetas[k];
        ua.tail.theta[k] = thetas[k];
        if (kind == 0) ls.beta.b[k] = betas[k];
    }
    const WsLayout W = ws_layout(2 * B, L, false);
    float* bws = wsf + S.bwd;
    bool dx_fused = false, x_direct = false;
    int nrows = (int)((B + 15) / 16);  // rows of pair sums / loss partials the first kernel leaves
    {   // forward + loss + data gradients: one kernel (nplda_train_fb_small.h; at the recipe shapes nplda_train_fb_half.h)
        TrainFbArgs fb = {};
        fb.xa = x1; fb.xb = x2; fb.n = B; fb.ldx = ldx; fb.packed = (const float*)packed; fb.D0 = L.D0; fb.KS1 = L.KS1;
        fb.oW2 = L.oW2; fb.oW2T = L.oW2T; fb.ob1 = L.ob1; fb.ob2 = L.ob2; fb.oQ = L.oQ; fb.oP = L.oP;
        fb.out_s = wsf + S.s; fb.out_y = wsf + S.y; fb.dz = bws + W.dz; fb.du = bws + W.du; fb.ldz = S.ldz;
        fb.pq = bws + W.pq; fb.ls = ls; fb.step_bump = step;
        if (rows) {
            fb.ia = (const long long*)rows1; fb.ib = (const long long*)rows2; fb.ntab = rows1 ? ntab : B;
            fb.rec_bump = cursor ? cursor + 1 : nullptr;
            // bf16 rows of the batch itself: the weight-gradient kernel reads them as they are (widened in registers) — no
            // fp32 copy staged here, 16.8 MB less left dirty behind this kernel at B = 4096
            x_direct = io_bf16 && !rows1 && wgrad_fm_bf16_ok(B, ldx, L);
            if (!x_direct) { fb.xsa = wsf + S.xs; fb.xsb = wsf + S.xs + (size_t)B * S.ldxs; fb.ldxs = S.ldxs; }
        }
        const dim3 grid((unsigned)((B + 15) / 16)), block(256);
        const bool k32 = L.KS1 == 32 && L.D0 == 512;
        if (io_bf16 && !(k32 && L.NB >= 10)) return NPLDA_EUNSUPPORTED;
        // dL/dx inside the first kernel (its DX form) at the recipe shapes: no dx launch, du is not read back
        dx_fused = dxa != nullptr && k32 && L.NB >= 10;
        if (dx_fused) { fb.oW1T = L.oW1T; fb.dx0 = dxa; fb.dx1 = dxb; fb.lddx = lddx; }
#define NPLDA_LAUNCH(NBV)                                                                                   \
    if (dx_fused && io_bf16) hipLaunchKernelGGL((train_fb_small_kernel<(NBV >= 10 ? NBV : 10), 32, true, true, 2>), grid, block, 0, st, fb); \
    else if (dx_fused && rows) hipLaunchKernelGGL((train_fb_small_kernel<(NBV >= 10 ? NBV : 10), 32, true, false, 1>), grid, block, 0, st, fb); \
    else if (dx_fused) hipLaunchKernelGGL((train_fb_small_kernel<(NBV >= 10 ? NBV : 10), 32, false, false, 1>), grid, block, 0, st, fb); \
    else if (io_bf16) hipLaunchKernelGGL((train_fb_small_kernel<(NBV >= 10 ? NBV : 10), 32, true, true>), grid, block, 0, st, fb); \
    else if (k32 && rows) hipLaunchKernelGGL((train_fb_small_kernel<NBV, 32, true>), grid, block, 0, st, fb);      \
    else if (k32) hipLaunchKernelGGL((train_fb_small_kernel<NBV, 32, false>), grid, block, 0, st, fb);        \
    else if (rows) hipLaunchKernelGGL((train_fb_small_kernel<NBV, 0, true>), grid, block, 0, st, fb);         \
    else hipLaunchKernelGGL((train_fb_small_kernel<NBV, 0, false>), grid, block, 0, st, fb)
        // 512-d x-vectors at NB = 10 / 11, batches of up to one half tile per CU: 8-pair tiles (nplda_train_fb_half.h)
        const HalfSkew skew = half_skew();
        if (k32 && (L.NB == 10 || L.NB == 11) && use_half_tiles(B)) {
            nrows = (int)((B + kHalfPairs - 1) / kHalfPairs);
            const dim3 hgrid((unsigned)((B + kHalfPairs - 1) / kHalfPairs));
#define NPLDA_LAUNCH_H(NBV)                                                                                                 \
    if (dx_fused && io_bf16) hipLaunchKernelGGL((train_fb_half_kernel<NBV, true, true, 2>), hgrid, block, 0, st, fb, skew);   \
    else if (dx_fused && rows) hipLaunchKernelGGL((train_fb_half_kernel<NBV, true, false, 1>), hgrid, block, 0, st, fb, skew); \
    else if (dx_fused) hipLaunchKernelGGL((train_fb_half_kernel<NBV, false, false, 1>), hgrid, block, 0, st, fb, skew);      \
    else if (io_bf16) hipLaunchKernelGGL((train_fb_half_kernel<NBV, true, true>), hgrid, block, 0, st, fb, skew);            \
    else if (rows) hipLaunchKernelGGL((train_fb_half_kernel<NBV, true>), hgrid, block, 0, st, fb, skew);                     \
    else hipLaunchKernelGGL((train_fb_half_kernel<NBV, false>), hgrid, block, 0, st, fb, skew)
            if (L.NB == 10) { NPLDA_LAUNCH_H(10); } else { NPLDA_LAUNCH_H(11); }
#undef NPLDA_LAUNCH_H
        } else {
        switch (L.NB) {
            case 2: NPLDA_LAUNCH(2); break;
            case 4: NPLDA_LAUNCH(4); break;
            case 8: NPLDA_LAUNCH(8); break;
            case 10: NPLDA_LAUNCH(10); break;
            case 11: NPLDA_LAUNCH(11); break;
            case 12: NPLDA_LAUNCH(12); break;
            default: return NPLDA_EUNSUPPORTED;
        }
        }
#undef NPLDA_LAUNCH
        if (int rc = nplda_launch_status()) return rc;
    }
    {   // the loss / threshold tail: in the weight-gradient launch where the full-M kernel runs, else in the update kernel
        LossTail& t = ua.tail;
        const size_t ngrad0 = nplda_grad_floats(D0, D1, D2);
        t.partial = ls.partial; t.nblk = nrows; t.K = nth; t.kind = kind; t.beta = ls.beta; t.alpha = alpha;
        t.loss = loss; t.loss_sum = loss_sum; t.m = exp_avg + ngrad0; t.v = exp_avg_sq + ngrad0;
        t.gout = grad_out ? grad_out + ngrad0 : nullptr; t.step = step; t.bumped = 1;
        t.lr = lr; t.beta1 = beta1; t.beta2 = beta2; t.eps = eps; t.wd = weight_decay;
        if (dp_flat) { t.sums_out = dp_flat + ngrad0; t.gout = nullptr; t.loss_sum = nullptr; }
    }
    ua.grad_only = dp_flat ? 1 : 0;
    bool tail_done = false;
    // the weight gradients read the x rows: the caller's, or the ones the first kernel gathered
    const bool staged = rows && !x_direct;
    const float* wx1 = staged ? wsf + S.xs : x1;
    const float* wx2 = staged ? wsf + S.xs + (size_t)B * S.ldxs : x2;
    if (int rc = backward_launch(false, wx1, wx2, 2 * B, B, staged ? S.ldxs : ldx, (const float*)packed, L, nullptr,
                                 wsf + S.y, wsf + S.z, wsf + S.rn, S.ldz, params[4], bws, W, grad_out, nullptr, nullptr, 0,
                                 st, &ls, &ua.r, true, &ua.tail, &tail_done, x_direct, nrows))
        return rc;
    if (dxa && !dx_fused) {  // dL/dx = du . W1 with the weights the forward used (the update below comes after)
        // (Measured and NOT kept, round 4: this launch on a side stream, forked behind the first kernel and joined in front of
        // the update — two branches of the captured graph.  The weight-gradient blocks (512 threads, 90 KB of LDS) and these
        // do not share a CU to any effect: cfg5 0.0820 -> 0.0810 ms, not worth a library that creates streams.)
        if (int rc = input_grad_from_du(bws + W.du, 2 * B, S.ldz, (const float*)packed, L, nullptr, dxa, dxb, B, lddx, st, io_bf16))
            return rc;
    }
    for (int i = 0; i < 6; ++i) ua.prm[i] = params[i];
    ua.m = exp_avg; ua.v = exp_avg_sq; ua.step = step;
    ua.lr = lr; ua.beta1 = beta1; ua.beta2 = beta2; ua.eps = eps; ua.wd = weight_decay;
    ua.L = L; ua.packed = (float*)packed;
    ua.bumped = 1;
    ua.tail_here = tail_done ? 0 : 1;
    if (cursor) {
        ua.cursor = cursor; ua.stage = (char*)stage; ua.rec_bytes = 20 * (long long)B;
        ua.ncopy = (unsigned)((ua.rec_bytes / 16 + 255) / 256);
        if (ua.ncopy > 32) ua.ncopy = 32;
    }
    const size_t ngrad = nplda_grad_floats(D0, D1, D2);
#ifndef NPLDA_UPDATE_E
#define NPLDA_UPDATE_E 2
#endif
    constexpr int E = NPLDA_UPDATE_E;  // without arrival tickets (the first kernel has counted the step) small blocks are free
    ua.ngrad_blocks = (unsigned)((ngrad + 256 * E - 1) / (256 * E));
    hipLaunchKernelGGL(train_update_kernel<E>, dim3(ua.ngrad_blocks + (unsigned)ua.tail_here + ua.ncopy), dim3(256), 0, st, ua);
    return nplda_launch_status();
}

int nplda_train_step_f32(const float* x1, const float* x2, int64_t B, int64_t ldx, const float* target,
                         float* const* params, int D0, int D1, int D2, float* const* thetas, const float* betas, int K,
                         float alpha, int kind, float* exp_avg, float* exp_avg_sq, float* step, float lr, float beta1,
                         float beta2, float eps, float weight_decay, void* packed, void* ws, size_t ws_bytes, float* loss,
                         double* loss_sum, float* grad_out, nplda_stream_t stream) {
    return train_step_impl(x1, x2, nullptr, nullptr, 0, nullptr, nullptr, B, ldx, target, params, D0, D1, D2, thetas, betas, K, alpha, kind,
                           exp_avg, exp_avg_sq, step, lr, beta1, beta2, eps, weight_decay, packed, ws, ws_bytes, loss,
                           loss_sum, grad_out, stream);
}

int nplda_train_step_dx_f32(const void* x1, const void* x2, int64_t B, int64_t ldx, int io_bf16, const float* target,
                            float* const* params, int D0, int D1, int D2, float* const* thetas, const float* betas, int K,
                            float alpha, int kind, float* exp_avg, float* exp_avg_sq, float* step, float lr, float beta1,
                            float beta2, float eps, float weight_decay, void* packed, void* ws, size_t ws_bytes, float* loss,
                            double* loss_sum, float* grad_out, void* dx1, void* dx2, int64_t lddx, nplda_stream_t stream) {
    if (!dx1 || !dx2) return NPLDA_EINVAL;
    return train_step_impl((const float*)x1, (const float*)x2, nullptr, nullptr, 0, nullptr, nullptr, B, ldx, target, params, D0,
                           D1, D2, thetas, betas, K, alpha, kind, exp_avg, exp_avg_sq, step, lr, beta1, beta2, eps,
                           weight_decay, packed, ws, ws_bytes, loss, loss_sum, grad_out, stream, dx1, dx2, lddx, io_bf16 != 0);
}

int nplda_train_step_grad_f32(const float* x1, const float* x2, int64_t B, int64_t ldx, const float* target,
                              const double* global_counts, float* const* params, int D0, int D1, int D2,
                              float* const* thetas, const float* betas, int K, float alpha, int kind, float* step,
                              void* packed, void* ws, size_t ws_bytes, float* flat, nplda_stream_t stream) {
    if (!flat) return NPLDA_EINVAL;
    float dummy_loss = 0.f;  // (never written: the tail stops at the sums)
    // the optimiser state is not touched in this phase; the moments' pointers only have to be non-null
    return train_step_impl(x1, x2, nullptr, nullptr, 0, nullptr, nullptr, B, ldx, target, params, D0, D1, D2, thetas, betas, K,
                           alpha, kind, flat, flat, step, 0.f, 0.f, 0.f, 0.f, 0.f, packed, ws, ws_bytes, &dummy_loss, nullptr,
                           nullptr, stream, nullptr, nullptr, 0, false, global_counts, flat);
}

int nplda_train_step_grad_rows_f32(const float* table, int64_t N, int64_t ldt, const int64_t* rows1, const int64_t* rows2,
                                   int64_t B, const float* target, const double* global_counts, float* const* params, int D0,
                                   int D1, int D2, float* const* thetas, const float* betas, int K, float alpha, int kind,
                                   float* step, void* packed, void* ws, size_t ws_bytes, float* flat, nplda_stream_t stream) {
    if (!flat || !rows1 || !rows2 || N < 1) return NPLDA_EINVAL;
    float dummy_loss = 0.f;
    return train_step_impl(table, table, rows1, rows2, N, nullptr, nullptr, B, ldt, target, params, D0, D1, D2, thetas, betas, K,
                           alpha, kind, flat, flat, step, 0.f, 0.f, 0.f, 0.f, 0.f, packed, ws, ws_bytes, &dummy_loss, nullptr,
                           nullptr, stream, nullptr, nullptr, 0, false, global_counts, flat);
}

int nplda_train_step_grad_dx_f32(const void* x1, const void* x2, int64_t B, int64_t ldx, int io_bf16, const float* target,
                                 const double* global_counts, float* const* params, int D0, int D1, int D2,
                                 float* const* thetas, const float* betas, int K, float alpha, int kind, float* step,
                                 void* packed, void* ws, size_t ws_bytes, float* flat, void* dx1, void* dx2, int64_t lddx,
                                 nplda_stream_t stream) {
    if (!flat || !dx1 || !dx2) return NPLDA_EINVAL;
    float dummy_loss = 0.f;
    return train_step_impl((const float*)x1, (const float*)x2, nullptr, nullptr, 0, nullptr, nullptr, B, ldx, target, params, D0,
                           D1, D2, thetas, betas, K, alpha, kind, flat, flat, step, 0.f, 0.f, 0.f, 0.f, 0.f, packed, ws,
                           ws_bytes, &dummy_loss, nullptr, nullptr, stream, dx1, dx2, lddx, io_bf16 != 0, global_counts, flat);
}

size_t nplda_train_step_flat_floats(int D0, int D1, int D2) {
    if (check_model(D0, D1, D2) != NPLDA_OK) return 0;
    return nplda_grad_floats(D0, D1, D2) + (size_t)kLossLimbs * kLossNS;
}

int nplda_train_step_apply_f32(const float* flat, float* const* params, int D0, int D1, int D2, float* const* thetas,
                               const float* betas, int K, float alpha, int kind, float* exp_avg, float* exp_avg_sq,
                               float* step, float lr, float beta1, float beta2, float eps, float weight_decay, void* packed,
                               float* loss, double* loss_sum, nplda_stream_t stream) {
    if (int rc = check_model(D0, D1, D2)) return rc;
    if (kind != 0 && kind != 1) return kind == 2 ? NPLDA_EUNSUPPORTED : NPLDA_EINVAL;
    const int nth = kind == 1 ? 1 : K;
    if (nth < 1 || nth > nplda_loss::kMaxK || !thetas || !params || (kind == 0 && !betas)) return NPLDA_EINVAL;
    if (!flat || !exp_avg || !exp_avg_sq || !step || !packed || !loss || !nplda_aligned16(packed)) return NPLDA_EINVAL;
    const NpldaLayout L = nplda_layout(D0, D1, D2);
    const size_t ngrad = nplda_grad_floats(D0, D1, D2);
    UpdateArgs ua = {};
    ua.r.D0 = D0; ua.r.D1 = D1; ua.r.D2 = D2;
    for (int i = 0; i < 6; ++i) {
        if (!params[i]) return NPLDA_EINVAL;
        ua.prm[i] = params[i];
    }
    LossTail& t = ua.tail;
    for (int k = 0; k < nth; ++k) {
        if (!thetas[k]) return NPLDA_EINVAL;
        t.theta[k] = thetas[k];
        if (kind == 0) t.beta.b[k] = betas[k];
    }
    t.nblk = 0; t.K = nth; t.kind = kind; t.alpha = alpha; t.loss = loss; t.loss_sum = loss_sum;
    t.m = exp_avg + ngrad; t.v = exp_avg_sq + ngrad; t.step = step; t.bumped = 1;
    t.lr = lr; t.beta1 = beta1; t.beta2 = beta2; t.eps = eps; t.wd = weight_decay;
    t.sums_in = flat + ngrad;
    ua.m = exp_avg; ua.v = exp_avg_sq; ua.step = step;
    ua.lr = lr; ua.beta1 = beta1; ua.beta2 = beta2; ua.eps = eps; ua.wd = weight_decay;
    ua.L = L; ua.packed = (float*)packed;
    ua.bumped = 1;  // nplda_train_step_grad_f32's first kernel has counted the step
    ua.tail_here = 1;
    ua.flat = flat;
    constexpr int E = 2;
    ua.ngrad_blocks = (unsigned)((ngrad + 256 * E - 1) / (256 * E));
    hipLaunchKernelGGL(train_update_kernel<E>, dim3(ua.ngrad_blocks + 1), dim3(256), 0, (hipStream_t)stream, ua);
    return nplda_launch_status();
}

size_t nplda_train_step_dx_workspace_bytes(int64_t B, int D0, int D1, int D2, int io_bf16) {
    if (B < 1 || check_model(D0, D1, D2) != NPLDA_OK) return 0;
    return step_ws(B, nplda_layout(D0, D1, D2), io_bf16 != 0).total * sizeof(float);
}

int nplda_train_step_rows_f32(const float* table, int64_t N, int64_t ldt, const int64_t* rows1, const int64_t* rows2,
                              int64_t B, const float* target, float* const* params, int D0, int D1, int D2,
                              float* const* thetas, const float* betas, int K, float alpha, int kind, float* exp_avg,
                              float* exp_avg_sq, float* step, float lr, float beta1, float beta2, float eps,
                              float weight_decay, void* packed, void* ws, size_t ws_bytes, float* loss, double* loss_sum,
                              float* grad_out, nplda_stream_t stream) {
    if (!rows1 || !rows2 || N < 1) return NPLDA_EINVAL;
    return train_step_impl(table, table, rows1, rows2, N, nullptr, nullptr, B, ldt, target, params, D0, D1, D2, thetas, betas, K, alpha, kind,
                           exp_avg, exp_avg_sq, step, lr, beta1, beta2, eps, weight_decay, packed, ws, ws_bytes, loss,
                           loss_sum, grad_out, stream);
}

int nplda_train_step_records_f32(const float* table, int64_t N, int64_t ldt, int64_t* cursor, void* stage, int64_t B,
                                 float* const* params, int D0, int D1, int D2, float* const* thetas, const float* betas,
                                 int K, float alpha, int kind, float* exp_avg, float* exp_avg_sq, float* step, float lr,
                                 float beta1, float beta2, float eps, float weight_decay, void* packed, void* ws,
                                 size_t ws_bytes, float* loss, double* loss_sum, float* grad_out, nplda_stream_t stream) {
    if (!cursor || !stage || N < 1) return NPLDA_EINVAL;
    return train_step_impl(table, table, nullptr, nullptr, N, (long long*)cursor, stage, B, ldt, nullptr, params, D0, D1, D2, thetas,
                           betas, K, alpha, kind, exp_avg, exp_avg_sq, step, lr, beta1, beta2, eps, weight_decay, packed, ws,
                           ws_bytes, loss, loss_sum, grad_out, stream);
}

size_t nplda_lda_wgrad_workspace_bytes(int64_t B, int D0, int D1) {
    if (B < 0 || check_model(D0, D1, D1) != NPLDA_OK) return 0;
    const WsLayout W = ws_layout(2 * B, nplda_layout(D0, D1, D1), false);
    return ((size_t)W.ksplit * W.Mp * W.Np1 + (size_t)W.ksplit * 4 * W.Mp) * sizeof(float);
}

int nplda_lda_wgrad_f32(const float* x1, const float* x2, int64_t B, int64_t ldx, const float* du, int64_t ldz, int D0,
                        int D1, void* ws, size_t ws_bytes, float* out, nplda_stream_t stream) {
    if (B < 0) return NPLDA_EINVAL;
    if (int rc = check_model(D0, D1, D1)) return rc;
    if (!out) return NPLDA_EINVAL;
    hipStream_t st = (hipStream_t)stream;
    const size_t nout = (size_t)D1 * D0 + D1;
    if (B == 0) {
        hipError_t e = hipMemsetAsync(out, 0, nout * sizeof(float), st);
        return e == hipSuccess ? NPLDA_OK : (int)e;
    }
    const NpldaLayout L = nplda_layout(D0, D1, D1);
    const WsLayout W = ws_layout(2 * B, L, false);
    if (!ws || !nplda_aligned16(ws) || !rows_ok(x1, ldx, D0) || !rows_ok(x2, ldx, D0) || !rows_ok(du, ldz, W.Mp))
        return NPLDA_EINVAL;
    if (ws_bytes < nplda_lda_wgrad_workspace_bytes(B, D0, D1)) return NPLDA_ENOSPC;
    float* slab = (float*)ws;
    float* ext = slab + (size_t)W.ksplit * W.Mp * W.Np1;
    WgradArgs wa = {};
    wa.K = 2 * B; wa.nsplit = B; wa.ksplit = W.ksplit; wa.rows_per_split = W.rows_per_split;
    wa.ldz = ldz; wa.ext = ext; wa.Mp = W.Mp;
    WgradProblem& p1 = wa.p[0];
    p1.A = du; p1.lda = ldz; p1.B0 = x1; p1.B1 = x2; p1.ldb = ldx; p1.M = W.Mp; p1.N = D0;
    p1.MT = (W.Mp + 63) / 64; p1.NT = (D0 + 63) / 64; p1.slab = slab; p1.Mp = W.Mp; p1.Np = W.Np1; p1.extras = 1;
    wa.p[1] = p1;  // never scheduled (nw == nw0)
    wa.nw0 = p1.MT * p1.NT * W.ksplit;
    wa.nw = wa.nw_mm = wa.nw_ps = wa.nw0;
    if (int rc = wgrad_launch(wa, L.NB, 1, st)) return rc;
    ReduceArgs ra = {};
    ra.slab1 = slab; ra.slab2 = slab; ra.ext = ext; ra.P_sqrt = nullptr;
    ra.ksplit = W.ksplit; ra.Mp = W.Mp; ra.Np1 = W.Np1; ra.D0 = D0; ra.D1 = D1; ra.D2 = 0; ra.out = out;
    hipLaunchKernelGGL(reduce_kernel, dim3((unsigned)((nout + 255) / 256)), dim3(256), 0, st, ra);
    return nplda_launch_status();
}

}  // extern "C"

// nplda_fwd_persist.h — persistent, 3-buffer-ring variant of the fused forward (gfx950, fp32 MFMA).
//
// Same arithmetic, lane layout and packed-weight image as nplda_fwd_kernel.h (see the design notes
// there); what changes is the schedule, aimed at the MFMA-pipe idle time the round-1 profile showed
// (profiles/r01b: pipe busy 73-78 %, clock steady at 2.3-2.4 GHz):
//  * PERSISTENT blocks: grid = resident blocks, each loops over 16-pair tiles.  The weight stream is
//    circular (after the last layer-2 chunk comes layer-1 chunk 0 again), so it never drains between
//    tiles, and the first x fragments of the next tile are fetched under the last layer-2 chunk — no
//    per-block dispatch + prologue bubble (was ~10 % of a 116 us block).
//  * RING OF 3 chunk buffers: chunk q+2 is fetched from L2 at the start of chunk q, written to LDS in
//    the MIDDLE of chunk q (its buffer was last read in chunk q-1, all waves are past that barrier) and
//    becomes visible at the barrier that ends chunk q.  Chunk q+1 is therefore already visible during
//    chunk q, so the first two weight fragments of chunk q+1 are read BEFORE the barrier and the MFMAs
//    restart immediately after it.
//  * explicit fragment double buffering inside a chunk: the next fragment pair is read from LDS before
//    the 16 MFMAs of the current pair are issued.
#pragma once
#include "../../neuralplda_amd/csrc/nplda_fwd_kernel.h"

namespace nplda {

template <int NB, int MODE, int WAVES, bool NT, int KPB>
__global__ __launch_bounds__(WAVES * 64, 2) void nplda_fwd_persist_kernel(const FwdArgs a, const int ntiles) {
    static_assert(MODE == MODE_PAIR || MODE == MODE_EMBED || MODE == MODE_TRAIN, "persistent kernel modes");
    constexpr int THREADS = WAVES * 64;
    constexpr int STEP4 = NB * 64;
    constexpr int CH = STEP4 * KPB;
    constexpr int NSLOT = (CH + THREADS - 1) / THREADS;
    constexpr int NF = KPB * NB;  // weight fragments per chunk
    constexpr int NC2 = (NB + KPB - 1) / KPB;
    __shared__ f32x4 wbuf[3][CH];

    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = tid >> 6;
    const int j = lane & 15;
    const int g = lane >> 4;

    const f32x4* Wall = reinterpret_cast<const f32x4*>(a.packed);
    const long long total4 = (long long)(a.total / 4);
    const f32x4* b1p = reinterpret_cast<const f32x4*>(a.packed + a.ob1);
    const f32x4* b2p = reinterpret_cast<const f32x4*>(a.packed + a.ob2);
    const f32x4* Qp = reinterpret_cast<const f32x4*>(a.packed + a.oQ);
    const f32x4* Pp = reinterpret_cast<const f32x4*>(a.packed + a.oP);
    const int KS1 = a.KS1;
    const int D0 = a.D0;
    const int NC1 = (KS1 + KPB - 1) / KPB;
    const int CT = NC1 + NC2;
    const long long w2base4 = (long long)(a.oW2 / 4);

    auto cbase = [&](int i) -> long long {  // chunk i of the circular per-tile stream
        if (i >= CT) i -= CT;
        return i < NC1 ? (long long)i * CH : w2base4 + (long long)(i - NC1) * CH;
    };
    auto tile_rows = [&](int tile, long long& t0A, long long& rowA, long long& rowB, bool& okA, bool& okB) {
        long long t0B;
        if (MODE == MODE_EMBED) {
            t0A = ((long long)tile * WAVES + wave) * 32;
            t0B = t0A + 16;
        } else {
            t0A = ((long long)tile * WAVES + wave) * 16;
            t0B = t0A;
        }
        rowA = t0A + j;
        rowB = t0B + j;
        okA = rowA < a.n;
        okB = rowB < a.n;
        if (!okA) rowA = a.n - 1;
        if (!okB) rowB = a.n - 1;
    };

    int tile = blockIdx.x;
    if (tile >= ntiles) return;

    // ---- prologue: chunks 0 and 1 into the ring, first fragments and first x of the first tile ----
    f32x4 st[NSLOT];
    {
        const long long b0 = cbase(0);
        chunk_load<CH, THREADS, NSLOT>(Wall + b0, total4 - b0, st, tid);
        chunk_store<CH, THREADS, NSLOT>(wbuf[0], st, tid);
        const long long b1 = cbase(1);
        chunk_load<CH, THREADS, NSLOT>(Wall + b1, total4 - b1, st, tid);
        chunk_store<CH, THREADS, NSLOT>(wbuf[1], st, tid);
    }
    long long t0A, rowA, rowB;
    bool okA, okB;
    tile_rows(tile, t0A, rowA, rowB, okA, okB);
    const float* pa = a.xa + rowA * a.ldx + 4 * g;
    const float* pb = a.xb + rowB * a.ldx + 4 * g;
    f32x4 xa[KPB], xb[KPB];
#pragma unroll
    for (int s = 0; s < KPB; ++s) {
        xa[s] = load_x4<NT>(pa + 16 * s, 16 * s + 4 * g < D0);
        xb[s] = load_x4<NT>(pb + 16 * s, 16 * s + 4 * g < D0);
    }
    __syncthreads();
    int bcur = 0, bnext = 1, bnn = 2;
    f32x4 pre0 = wbuf[0][lane], pre1 = wbuf[0][64 + lane];

    for (;;) {
        f32x4 accA[NB], accB[NB];
#pragma unroll
        for (int nb = 0; nb < NB; ++nb) {
            accA[nb] = b1p[4 * nb + g];
            accB[nb] = accA[nb];
        }

        // ---- layer 1 -----------------------------------------------------------------------------------
        for (int c = 0; c < NC1; ++c) {
            const long long nbase = cbase(c + 2);
            chunk_load<CH, THREADS, NSLOT>(Wall + nbase, total4 - nbase, st, tid);
            const bool more = (c + 1 < NC1);
            f32x4 xan[KPB], xbn[KPB];
#pragma unroll
            for (int s = 0; s < KPB; ++s) {
                const int ks = KPB * (c + 1) + s;
                const bool ok = more && (16 * ks + 4 * g < D0);
                xan[s] = load_x4<NT>(pa + 16 * ks, ok);
                xbn[s] = load_x4<NT>(pb + 16 * ks, ok);
            }
            const f32x4* w = wbuf[bcur];
            const f32x4* wn = wbuf[bnext];
            f32x4 a0 = pre0, a1 = pre1;
#pragma unroll
            for (int fp = 0; fp < NF; fp += 2) {
                f32x4 n0, n1;
                if (fp + 2 < NF) {
                    n0 = w[(fp + 2) * 64 + lane];
                    n1 = (fp + 3 < NF) ? w[(fp + 3) * 64 + lane] : wn[lane];
                } else {
                    // NF even: next chunk frags 0,1.  NF odd: a1 of this last pair already IS next-chunk frag 0.
                    n0 = (NF % 2 == 0) ? wn[lane] : a1;
                    n1 = wn[64 + lane];
                }
#pragma unroll
                for (int u = 0; u < 2; ++u) {
                    const int f = fp + u;
                    if (f < NF) {
                        const int s = f / NB, nb = f % NB;
                        if (KPB * c + s < KS1) {
                            const f32x4 av = u == 0 ? a0 : a1;
#pragma unroll
                            for (int r = 0; r < 4; ++r) {
                                accA[nb] = __builtin_amdgcn_mfma_f32_16x16x4f32(av[r], xa[s][r], accA[nb], 0, 0, 0);
                                accB[nb] = __builtin_amdgcn_mfma_f32_16x16x4f32(av[r], xb[s][r], accB[nb], 0, 0, 0);
                            }
                        }
                    }
                }
                if (fp == (NF / 4) * 2) chunk_store<CH, THREADS, NSLOT>(wbuf[bnn], st, tid);
                a0 = n0;
                a1 = n1;
            }
            pre0 = a0;
            pre1 = a1;
#pragma unroll
            for (int s = 0; s < KPB; ++s) {
                xa[s] = xan[s];
                xb[s] = xbn[s];
            }
            __syncthreads();
            const int tmp = bcur; bcur = bnext; bnext = bnn; bnn = tmp;
        }

        // ---- F.normalize (utils/models.py:368) --------------------------------------------------------
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
#pragma unroll
            for (int nb = 0; nb < NB; ++nb) {
                accA[nb] *= invA;
                accB[nb] *= invB;
            }
        }
        if (MODE == MODE_TRAIN) {
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

        // next tile (its first x fragments are fetched under the last layer-2 chunk)
        const int ntile = tile + gridDim.x;
        const bool hasnext = ntile < ntiles;
        long long nt0A = 0, nrowA = 0, nrowB = 0;
        bool nokA = false, nokB = false;
        if (hasnext) tile_rows(ntile, nt0A, nrowA, nrowB, nokA, nokB);
        const float* npa = a.xa + nrowA * a.ldx + 4 * g;
        const float* npb = a.xb + nrowB * a.ldx + 4 * g;

        // ---- layer 2 -----------------------------------------------------------------------------------
        f32x4 zA[NB], zB[NB];
#pragma unroll
        for (int nb = 0; nb < NB; ++nb) {
            zA[nb] = b2p[4 * nb + g];
            zB[nb] = zA[nb];
        }
#pragma unroll
        for (int c2 = 0; c2 < NC2; ++c2) {
            const long long nbase = cbase(NC1 + c2 + 2);
            chunk_load<CH, THREADS, NSLOT>(Wall + nbase, total4 - nbase, st, tid);
            if (c2 == NC2 - 1) {
#pragma unroll
                for (int s = 0; s < KPB; ++s) {
                    xa[s] = load_x4<NT>(npa + 16 * s, hasnext && (16 * s + 4 * g < D0));
                    xb[s] = load_x4<NT>(npb + 16 * s, hasnext && (16 * s + 4 * g < D0));
                }
            }
            const f32x4* w = wbuf[bcur];
            const f32x4* wn = wbuf[bnext];
            f32x4 a0 = pre0, a1 = pre1;
#pragma unroll
            for (int fp = 0; fp < NF; fp += 2) {
                f32x4 n0, n1;
                if (fp + 2 < NF) {
                    n0 = w[(fp + 2) * 64 + lane];
                    n1 = (fp + 3 < NF) ? w[(fp + 3) * 64 + lane] : wn[lane];
                } else {
                    // NF even: next chunk frags 0,1.  NF odd: a1 of this last pair already IS next-chunk frag 0.
                    n0 = (NF % 2 == 0) ? wn[lane] : a1;
                    n1 = wn[64 + lane];
                }
#pragma unroll
                for (int u = 0; u < 2; ++u) {
                    const int f = fp + u;
                    if (f < NF) {
                        const int s = f / NB, nb = f % NB;
                        const int kb = KPB * c2 + s;
                        if (kb < NB) {
                            const f32x4 av = u == 0 ? a0 : a1;
#pragma unroll
                            for (int r = 0; r < 4; ++r) {
                                zA[nb] = __builtin_amdgcn_mfma_f32_16x16x4f32(av[r], accA[kb < NB ? kb : 0][r], zA[nb], 0, 0, 0);
                                zB[nb] = __builtin_amdgcn_mfma_f32_16x16x4f32(av[r], accB[kb < NB ? kb : 0][r], zB[nb], 0, 0, 0);
                            }
                        }
                    }
                }
                if (fp == (NF / 4) * 2) chunk_store<CH, THREADS, NSLOT>(wbuf[bnn], st, tid);
                a0 = n0;
                a1 = n1;
            }
            pre0 = a0;
            pre1 = a1;
            __syncthreads();
            const int tmp = bcur; bcur = bnext; bnext = bnn; bnn = tmp;
        }

        // ---- epilogue ------------------------------------------------------------------------------------
        if (MODE == MODE_PAIR || MODE == MODE_TRAIN) {
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

        if (!hasnext) break;
        tile = ntile;
        t0A = nt0A; rowA = nrowA; rowB = nrowB; okA = nokA; okB = nokB;
        pa = npa; pb = npb;
    }
}

}  // namespace nplda

// nplda_fwd_v3.h — persistent, continuous-stream form of the v2 schedule (same arithmetic, same image).
//
// All v2 variants (4 or 8 waves, 2 or 4 k16-steps per barrier) land at 0.78 of the fp32 MFMA peak while a loop of
// the same MFMAs with no memory instructions reaches 0.85 (tools/exp_fwd.hip ablations): what is left is the
// per-block ramp — workgroup dispatch, the exposed latency of the first x rows and of the first weight chunk, the
// drained pipeline in the epilogue — paid once per 16 * WAVES pairs.  v3 launches one resident grid and lets every
// block walk tiles t = blockIdx.x, + gridDim.x, ... while the weight stream never stops: the chunks of a tile
// (layer 1: NC1, layer 2: NC2) are followed directly by chunk 0 of the NEXT tile, staged into the free LDS buffer
// during the last layer-2 chunk exactly like any other chunk, and the x ring slots that die in the last layer-1
// chunk are refilled with the first k16-steps of the next tile's rows (they sit in registers that layer 2 does
// not need).  No register is added for the cross-tile prefetch; only the first tile of a block pays a prologue.
#pragma once
#include "nplda_fwd_kernel.h"

namespace nplda {

template <int NB, int MODE, int WAVES, int NT, int KPB, int G = 4>  // NT: the x-row mode XM of load_xrow
__global__ __launch_bounds__(WAVES * 64, 2) void nplda_fwd_v3_kernel(const FwdArgs a, int ntiles) {
    static_assert(MODE == MODE_PAIR || MODE == MODE_EMBED || MODE == MODE_TRAIN, "v2 kernel modes");
    constexpr int THREADS = WAVES * 64;
    constexpr int STEP4 = NB * 64;
    constexpr int CH = STEP4 * KPB;
    constexpr int HALF = ((CH / 2 + THREADS - 1) / THREADS) * THREADS;  // first-half size, multiple of THREADS
    constexpr int NS1 = HALF / THREADS;                                   // staging slots of the first half
    constexpr int NS2 = (CH - HALF + THREADS - 1) / THREADS;              // ... of the second half
    constexpr int NS = NS1 > NS2 ? NS1 : NS2;
    constexpr int NC2 = (NB + KPB - 1) / KPB;
    static_assert(HALF <= CH && KPB >= 2, "chunk must split into two staging halves");
    constexpr int SMID = KPB / 2;  // the staging hand-over happens after this many steps
    __shared__ f32x4 wbuf[2][CH];

    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int j = lane & 15;
    const int g = lane >> 4;

    // rows of tile t for this lane: clamped (always-valid) row starts for the branch-free loads
    auto tile_rows = [&](long long t, long long& t0A, long long& rA, long long& rB) {
        long long t0B;
        if (MODE == MODE_EMBED) {
            t0A = (t * WAVES + wave) * 32;
            t0B = t0A + 16;
        } else {
            t0A = (t * WAVES + wave) * 16;
            t0B = t0A;
        }
        rA = t0A + j;
        rB = t0B + j;
    };
    long long tile = blockIdx.x;
    long long t0A, rowA, rowB;
    tile_rows(tile, t0A, rowA, rowB);
    bool okA = rowA < a.n, okB = rowB < a.n;
    if (!okA) rowA = a.n - 1;
    if (!okB) rowB = a.n - 1;
    const float* sa = x_row<NT>(a.xa, rowA, a.ldx);
    const float* sb = x_row<NT>(a.xb, rowB, a.ldx);

    const f32x4* Wall = reinterpret_cast<const f32x4*>(a.packed);
    // b1, b2, Q, P (NB * 16 floats each) are needed once per tile: a persistent block keeps them in LDS instead of
    // paying an exposed L2 round trip at every tile start / epilogue
    __shared__ f32x4 cvec[4][NB * 4];
    for (int i = tid; i < 4 * NB * 4; i += THREADS) {
        const int v = i / (NB * 4), e = i % (NB * 4);
        const size_t o = v == 0 ? a.ob1 : (v == 1 ? a.ob2 : (v == 2 ? a.oQ : a.oP));
        cvec[v][e] = reinterpret_cast<const f32x4*>(a.packed + o)[e];
    }
    const f32x4* b1p = cvec[0];
    const f32x4* b2p = cvec[1];
    const f32x4* Qp = cvec[2];
    const f32x4* Pp = cvec[3];
    const int KS1 = a.KS1;
    const int D0 = a.D0;
    const int NC1 = (KS1 + KPB - 1) / KPB;
    const long long w2base4 = (long long)(a.oW2 / 4);

    f32x4 st[NS];
    // all staging loads are unconditional (the image carries a chunk of slack): see chunk_load in nplda_fwd_kernel.h
    auto load1 = [&](long long base) {  // first half of the chunk at float4 offset `base`
#pragma unroll
        for (int i = 0; i < NS1; ++i) st[i] = Wall[base + tid + THREADS * i];
    };
    auto store1 = [&](f32x4* dst) {
#pragma unroll
        for (int i = 0; i < NS1; ++i) dst[tid + THREADS * i] = st[i];
    };
    auto load2 = [&](long long base) {
#pragma unroll
        for (int i = 0; i < NS2; ++i) {
            const int idx = HALF + tid + THREADS * i;
            st[i] = Wall[base + (idx < CH ? idx : CH - 1)];
        }
    };
    auto store2 = [&](f32x4* dst) {
#pragma unroll
        for (int i = 0; i < NS2; ++i) {
            const int idx = HALF + tid + THREADS * i;
            if (idx < CH) dst[idx] = st[i];
        }
    };

    // ---- prologue --------------------------------------------------------------------------------------
    load1(0);
    f32x4 xa[KPB], xb[KPB];
#pragma unroll
    for (int s = 0; s < KPB; ++s) {
        xa[s] = load_xrow<NT>(sa, 16 * s + 4 * g, D0);
        xb[s] = load_xrow<NT>(sb, 16 * s + 4 * g, D0);
    }
    store1(wbuf[0]);
    load2(0);
    store2(wbuf[0]);
    __syncthreads();
    int par = 0;  // LDS buffer holding the current chunk; flips at every chunk, across tiles too

    for (;;) {
    // next tile of this block (rows clamped into range when there is none: its prefetches are then harmless)
    const long long tile_n = tile + gridDim.x;
    long long t0A_n, rowA_n, rowB_n;
    tile_rows(tile_n, t0A_n, rowA_n, rowB_n);
    const bool okA_n = rowA_n < a.n, okB_n = rowB_n < a.n;
    if (!okA_n) rowA_n = a.n - 1;
    if (!okB_n) rowB_n = a.n - 1;
    const float* sa_n = x_row<NT>(a.xa, rowA_n, a.ldx);
    const float* sb_n = x_row<NT>(a.xb, rowB_n, a.ldx);

    f32x4 accA[NB], accB[NB];
#pragma unroll
    for (int nb = 0; nb < NB; ++nb) {
        accA[nb] = b1p[4 * nb + g];
        accB[nb] = accA[nb];
    }

    // ---- layer 1 -----------------------------------------------------------------------------------------
    for (int c = 0; c < NC1; ++c) {
        const int cur = par;
        const bool more = (c + 1 < NC1);
        const long long nbase = more ? (long long)(c + 1) * CH : w2base4;
        load1(nbase);
        const f32x4* w = wbuf[cur];
#pragma unroll
        for (int s = 0; s < KPB; ++s) {
            if (KPB * c + s < KS1) {
                // G feature blocks per LDS wait: their fragments are read together and the MFMAs run r-major across
                // them, so the wave stalls on LDS latency once per 8 G MFMAs (hipcc issues each read right before
                // its first use) and every accumulator chain has 2 G MFMAs between dependent instructions
#pragma unroll
                for (int nb0 = 0; nb0 < NB; nb0 += G) {
                    f32x4 av[G];
#pragma unroll
                    for (int u = 0; u < G; ++u)
                        if (nb0 + u < NB) av[u] = w[s * STEP4 + (nb0 + u) * 64 + lane];
#pragma unroll
                    for (int r = 0; r < 4; ++r) {
#pragma unroll
                        for (int u = 0; u < G; ++u) {
                            if (nb0 + u < NB) {
                                accA[nb0 + u] = __builtin_amdgcn_mfma_f32_16x16x4f32(av[u][r], xa[s][r], accA[nb0 + u], 0, 0, 0);
                                accB[nb0 + u] = __builtin_amdgcn_mfma_f32_16x16x4f32(av[u][r], xb[s][r], accB[nb0 + u], 0, 0, 0);
                            }
                        }
                    }
                }
            }
            {   // slot s is free: fetch k16-step s of the next chunk (a whole chunk ahead of its use); after the
                // last chunk of the tile that is step s of the NEXT tile's rows
                const int ks = more ? KPB * (c + 1) + s : s;
                const float* ra = more ? sa : sa_n;
                const float* rb = more ? sb : sb_n;
                xa[s] = load_xrow<NT>(ra, 16 * ks + 4 * g, D0);
                xb[s] = load_xrow<NT>(rb, 16 * ks + 4 * g, D0);
            }
            if (s == SMID - 1) {
                store1(wbuf[cur ^ 1]);
                load2(nbase);
            }
        }
        store2(wbuf[cur ^ 1]);
        __syncthreads();
        par ^= 1;
    }

    // ---- F.normalize (utils/models.py:368) -------------------------------------------------------------
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

    // ---- layer 2 -----------------------------------------------------------------------------------------
    f32x4 zA[NB], zB[NB];
#pragma unroll
    for (int nb = 0; nb < NB; ++nb) {
        zA[nb] = b2p[4 * nb + g];
        zB[nb] = zA[nb];
    }
#pragma unroll
    for (int c2 = 0; c2 < NC2; ++c2) {
        const int cur = par;
        const bool more2 = (c2 + 1 < NC2);
        const long long nbase = more2 ? w2base4 + (long long)(c2 + 1) * CH : 0;  // 0: chunk 0 of the next tile
        load1(nbase);
        const f32x4* w = wbuf[cur];
#pragma unroll
        for (int s = 0; s < KPB; ++s) {
            const int kb = KPB * c2 + s;
            if (kb < NB) {
#pragma unroll
                for (int nb0 = 0; nb0 < NB; nb0 += G) {
                    f32x4 av[G];
#pragma unroll
                    for (int u = 0; u < G; ++u)
                        if (nb0 + u < NB) av[u] = w[s * STEP4 + (nb0 + u) * 64 + lane];
#pragma unroll
                    for (int r = 0; r < 4; ++r) {
#pragma unroll
                        for (int u = 0; u < G; ++u) {
                            if (nb0 + u < NB) {
                                zA[nb0 + u] = __builtin_amdgcn_mfma_f32_16x16x4f32(av[u][r], accA[kb < NB ? kb : 0][r], zA[nb0 + u], 0, 0, 0);
                                zB[nb0 + u] = __builtin_amdgcn_mfma_f32_16x16x4f32(av[u][r], accB[kb < NB ? kb : 0][r], zB[nb0 + u], 0, 0, 0);
                            }
                        }
                    }
                }
            }
            if (s == SMID - 1) {
                store1(wbuf[cur ^ 1]);
                load2(nbase);
            }
        }
        store2(wbuf[cur ^ 1]);
        __syncthreads();
        par ^= 1;
    }

    // ---- epilogue ------------------------------------------------------------------------------------------
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

    // ---- advance to this block's next tile (its chunk 0 and first x rows are already on their way) ---------
    tile = tile_n;
    if (tile >= ntiles) break;
    t0A = t0A_n; rowA = rowA_n; rowB = rowB_n; okA = okA_n; okB = okB_n;
    sa = sa_n; sb = sb_n;
    }  // tile loop
}

}  // namespace nplda

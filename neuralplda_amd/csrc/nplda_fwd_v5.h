// nplda_fwd_v5.h — the persistent schedule of nplda_fwd_v3.h with a register budget that fits NB = 11 and 12
// (D = 170, the reference's shipped dimension: conf/voices_config.cfg:14-16).  Same arithmetic, same image, same bits.
//
// v3 keeps both layers' accumulators (4 NB + 4 NB registers), the next tile's first x rows and a register-staged
// weight chunk alive through layer 2: 88 + 88 + 16 + 8 + fragments at NB = 11, which spills under the 256 registers
// of 2 waves/SIMD.  Three changes remove the pressure instead of adding occupancy:
//  * layer 2 walks its OUTPUT blocks in groups of G = 4: a group's 2 x 4 accumulators run through all NB k-blocks
//    (H weight chunks: the group's rows of W2, its k range in H pieces; H = 1 measured best), are folded into the score at once
//    (s += Q (z1^2 + z2^2) + 2 P z1 z2 over the group's 64 features) and die — 32 live registers instead of 8 NB;
//  * weight chunks go global -> LDS by LDS-DMA (global_load_lds, 1 KB per wave instruction, the fragment order
//    of the image is the order the MFMAs read): no staging registers, no ds_write, and any chunk shape is just a
//    different list of 1 KB segments — which is what lets layer 2 read W2 group-major out of the k-major image;
//  * the next tile's first x rows are fetched in the LAST layer-2 chunk, into registers layer 2 does not use.
// A chunk's x rows are loaded a whole chunk ahead into a second register set (the two sets alternate from chunk to chunk), so
// that the vmcnt(0) the DMA needs before the barrier never waits on a young load.
#pragma once
#include "nplda_fwd_kernel.h"

namespace nplda {

template <int NB, int WAVES, int KPB = 2, int G = 4, int H = 2, int XM = 0>  // XM: the x-row mode of load_xrow
__global__ __launch_bounds__(WAVES * 64, WAVES / 4) void nplda_fwd_v5_kernel(const FwdArgs a, int ntiles) {
    constexpr int STEP4 = NB * 64;                   // float4 per k16-step of weights
    constexpr int CH1 = STEP4 * KPB;                 // layer-1 chunk
    constexpr int NSEG1 = CH1 / 64;
    constexpr int NG = (NB + G - 1) / G;             // layer-2 output groups
    constexpr int KH = (NB + H - 1) / H;             // k-blocks per layer-2 chunk (a group's k range in H pieces)
    constexpr int CH2 = KH * G * 64;
    constexpr int CH = CH1 > CH2 ? CH1 : CH2;
    __shared__ f32x4 wbuf[2][CH];
    __shared__ f32x4 cvec[4][NB * 4];
    __shared__ f32x4 sink[64];  // where the surplus segment of a wave lands (keeps the DMA issue branch-free)

    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int j = lane & 15;
    const int g = lane >> 4;

    const f32x4* Wall = reinterpret_cast<const f32x4*>(a.packed);
    const long long w2base4 = (long long)(a.oW2 / 4);
    for (int i = tid; i < 4 * NB * 4; i += WAVES * 64) {
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

    // one 1 KB segment: lane l's 16 bytes land at dst[l].  The source is a wave-uniform base plus a 32-bit lane offset
    // (the SGPR-base form of the instruction); the offset is made opaque at every use, or the compiler hoists one
    // 64-bit vector address per segment of every chunk shape out of the tile loop and spills them.
    auto seg_dma = [&](const f32x4* src, f32x4* dst) {
        unsigned lo = (unsigned)lane * 16u;
        asm volatile("" : "+v"(lo));
        __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(reinterpret_cast<const char*>(src) + lo),
                                         (__attribute__((address_space(3))) void*)dst, 16, 0, 0);
    };
    // layer-1 chunk c: KPB consecutive k16-steps of W1, contiguous in the image
    auto dma_l1 = [&](int c, f32x4* dst) {
        const f32x4* src = Wall + (long long)c * CH1;
#pragma unroll
        for (int i = 0; i < (NSEG1 + WAVES - 1) / WAVES; ++i) {
            const int sgm = wave + WAVES * i;
            const bool live = sgm < NSEG1;
            seg_dma(src + (live ? sgm : 0) * 64, live ? dst + sgm * 64 : sink);
        }
    };
    // layer-2 chunk: output blocks nb0 .. nb0 + gc - 1, k-blocks kb0 .. kb0 + kc - 1, stored [k][u][lane]
    auto dma_l2 = [&](int nb0, int gc, int kb0, int kc, f32x4* dst) {
        const int nseg = kc * gc;
#pragma unroll
        for (int i = 0; i < (KH * G + WAVES - 1) / WAVES; ++i) {
            const int sgm = wave + WAVES * i;
            const bool live = sgm < nseg;
            const int sg = live ? sgm : 0;
            const int kl = sg / gc, u = sg - kl * gc;
            seg_dma(Wall + w2base4 + (long long)(kb0 + kl) * STEP4 + (nb0 + u) * 64, live ? dst + sgm * 64 : sink);
        }
    };
    auto chunk_fence = [&]() {
        __builtin_amdgcn_s_waitcnt(0x0070);  // vmcnt(0) lgkmcnt(0): this wave's segments of the next chunk are in LDS
        __builtin_amdgcn_s_barrier();
        __builtin_amdgcn_sched_barrier(0);  // nothing of the next chunk's code moves up across the fence
    };

    auto tile_rows = [&](long long t, long long& t0, long long& r) {
        t0 = (t * WAVES + wave) * 16;
        r = t0 + j;
    };
    long long tile = blockIdx.x;
    long long t0, row;
    tile_rows(tile, t0, row);
    bool ok = row < a.n;
    if (!ok) row = a.n - 1;
    const float* sa = x_row<XM>(a.xa, row, a.ldx);
    const float* sb = x_row<XM>(a.xb, row, a.ldx);

    // ---- prologue: chunk 0 of layer 1 and the first x rows -------------------------------------------------
    dma_l1(0, wbuf[0]);
    f32x4 xa[KPB], xb[KPB];
#pragma unroll
    for (int s = 0; s < KPB; ++s) {
        xa[s] = load_xrow<XM>(sa, 16 * s + 4 * g, D0);
        xb[s] = load_xrow<XM>(sb, 16 * s + 4 * g, D0);
    }
    __syncthreads();  // cvec
    chunk_fence();
    int par = 0;

    for (;;) {
        const long long tile_n = tile + gridDim.x;
        long long t0_n, row_n;
        tile_rows(tile_n, t0_n, row_n);
        const bool ok_n = row_n < a.n;
        if (!ok_n) row_n = a.n - 1;
        const float* sa_n = x_row<XM>(a.xa, row_n, a.ldx);
        const float* sb_n = x_row<XM>(a.xb, row_n, a.ldx);

        f32x4 accA[NB], accB[NB];
#pragma unroll
        for (int nb = 0; nb < NB; ++nb) {
            accA[nb] = b1p[4 * nb + g];
            accB[nb] = accA[nb];
        }

        // ---- layer 1 ---------------------------------------------------------------------------------------
        // one chunk: the MFMAs read (xc, yc) while (xn, yn) receive the next chunk's rows; the loop below alternates the two
        // register sets (round 5; before, the next set was copied over the current one: 16 v_mov per chunk, each of them
        // matrix-pipe time — nplda_fwd_v6.h)
        auto chunk = [&](int c, f32x4 (&xc)[KPB], f32x4 (&yc)[KPB], f32x4 (&xn)[KPB], f32x4 (&yn)[KPB]) {
            const bool more = (c + 1 < NC1);
            f32x4* nxt = wbuf[par ^ 1];
            if (more) dma_l1(c + 1, nxt);
            else dma_l2(0, (G < NB ? G : NB), 0, KH, nxt);
            // x of the next chunk (after the last chunk: a harmless re-read of the steps just fetched — still in cache;
            // re-reading the tile's FIRST steps instead cost 6 % more HBM fetches, FETCH_SIZE 2.29e6 vs 2.15e6 KB)
#pragma unroll
            for (int s = 0; s < KPB; ++s) {
                const int ks = more ? KPB * (c + 1) + s : KPB * c + s;
                xn[s] = load_xrow<XM>(sa, 16 * ks + 4 * g, D0);
                yn[s] = load_xrow<XM>(sb, 16 * ks + 4 * g, D0);
            }
            const f32x4* w = wbuf[par];
#pragma unroll
            for (int s = 0; s < KPB; ++s) {
                if (KPB * c + s < KS1) {
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
                                    accA[nb0 + u] = __builtin_amdgcn_mfma_f32_16x16x4f32(av[u][r], xc[s][r], accA[nb0 + u], 0, 0, 0);
                                    accB[nb0 + u] = __builtin_amdgcn_mfma_f32_16x16x4f32(av[u][r], yc[s][r], accB[nb0 + u], 0, 0, 0);
                                }
                            }
                        }
                    }
                }
            }
            chunk_fence();
            par ^= 1;
        };
        f32x4 xan[KPB], xbn[KPB];
        int c = 0;
        for (; c + 1 < NC1; c += 2) {
            chunk(c, xa, xb, xan, xbn);
            chunk(c + 1, xan, xbn, xa, xb);
        }
        if (c < NC1) {  // an odd chunk count (never at 512-d x-vectors): one more, and the sets change places by copy
            chunk(c, xa, xb, xan, xbn);
#pragma unroll
            for (int s = 0; s < KPB; ++s) {
                xa[s] = xan[s];
                xb[s] = xbn[s];
            }
        }

        // ---- F.normalize (utils/models.py:368) ---------------------------------------------------------------
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
            const float invA = 1.0f / fmaxf(sqrtf(ssA), 1e-12f);
            const float invB = 1.0f / fmaxf(sqrtf(ssB), 1e-12f);
#pragma unroll
            for (int nb = 0; nb < NB; ++nb) {
                accA[nb] *= invA;
                accB[nb] *= invB;
            }
        }

        // ---- layer 2, output groups of G blocks; the score is folded group by group ------------------------------
        float part = 0.f;
#pragma unroll
        for (int gi = 0; gi < NG; ++gi) {
            const int nb0 = gi * G;
            const int gc = (NB - nb0) < G ? (NB - nb0) : G;
            f32x4 zA[G], zB[G];
#pragma unroll
            for (int u = 0; u < G; ++u) {
                if (u < gc) {
                    zA[u] = b2p[4 * (nb0 + u) + g];
                    zB[u] = zA[u];
                }
            }
#pragma unroll
            for (int h = 0; h < H; ++h) {
                const int kb0 = h * KH;
                const int kc = (NB - kb0) < KH ? (NB - kb0) : KH;
                f32x4* nxt = wbuf[par ^ 1];
                const bool last = (gi == NG - 1 && h == H - 1);
                if (h + 1 < H) {
                    const int kb1 = kb0 + KH;
                    dma_l2(nb0, gc, kb1, (NB - kb1) < KH ? (NB - kb1) : KH, nxt);
                } else if (!last) {
                    const int nb1 = nb0 + G;
                    dma_l2(nb1, (NB - nb1) < G ? (NB - nb1) : G, 0, KH, nxt);
                } else {
                    dma_l1(0, nxt);  // chunk 0 of the next tile, and its first x rows
#pragma unroll
                    for (int s = 0; s < KPB; ++s) {
                        xa[s] = load_xrow<XM>(sa_n, 16 * s + 4 * g, D0);
                        xb[s] = load_xrow<XM>(sb_n, 16 * s + 4 * g, D0);
                    }
                }
                const f32x4* w = wbuf[par];
#pragma unroll
                for (int kl = 0; kl < kc; ++kl) {
                    const int kb = kb0 + kl;
                    f32x4 av[G];
#pragma unroll
                    for (int u = 0; u < G; ++u)
                        if (u < gc) av[u] = w[(kl * gc + u) * 64 + lane];
#pragma unroll
                    for (int r = 0; r < 4; ++r) {
#pragma unroll
                        for (int u = 0; u < G; ++u) {
                            if (u < gc) {
                                zA[u] = __builtin_amdgcn_mfma_f32_16x16x4f32(av[u][r], accA[kb][r], zA[u], 0, 0, 0);
                                zB[u] = __builtin_amdgcn_mfma_f32_16x16x4f32(av[u][r], accB[kb][r], zB[u], 0, 0, 0);
                            }
                        }
                    }
                }
                chunk_fence();
                par ^= 1;
            }
#pragma unroll
            for (int u = 0; u < G; ++u) {
                if (u < gc) {
                    const f32x4 q = Qp[4 * (nb0 + u) + g];
                    const f32x4 p = Pp[4 * (nb0 + u) + g];
#pragma unroll
                    for (int r = 0; r < 4; ++r) {
                        const float z1 = zA[u][r], z2 = zB[u][r];
                        part = fmaf(q[r], fmaf(z1, z1, z2 * z2), part);
                        part = fmaf(2.0f * p[r], z1 * z2, part);
                    }
                }
            }
        }
        part = wave_xor_add(part, 16);
        part = wave_xor_add(part, 32);
        if (g == 0 && ok) a.out_s[t0 + j] = part;

        tile = tile_n;
        if (tile >= ntiles) break;
        t0 = t0_n; ok = ok_n;
        sa = sa_n; sb = sb_n;
    }
}

}  // namespace nplda

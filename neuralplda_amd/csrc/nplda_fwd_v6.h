// nplda_fwd_v6.h — the persistent pair-scoring schedule of nplda_fwd_v5.h with the PARTIAL feature block off the 16x16 MFMA.
//
// D = 150 is nine full 16-feature blocks and six features.  v3 / v5 run the left-over features as one more 16 x 16 x 4 MFMA
// block (NB = 10): 10 / 16 of that block's matrix-pipe time multiplies zeros — 1680 MFMAs per 16 rows where 150 features
// need 1552 (PMC: the kernels execute exactly the padded count, and the matrix pipe is what bounds them).  Here the TF
// left-over features of BOTH layers go through v_mfma_f32_4x4x1_16b_f32 — sixteen independent 4 x 4 x 1 products per
// instruction, two passes instead of eight:
//  * block (g, j / 4) of the instruction takes lane (j, g)'s element of the x fragment the lane already holds for the
//    16 x 16 MFMAs as its B operand — rows 4 (j / 4) .. + 3 at the lane group's own k value — and four tail rows of W as A:
//    fragment lanes 16 g + 4 q + j % 4 of the image's last feature block, already in LDS with the chunk (one more
//    ds_read_b128 per quad, four addresses per 16 lanes).  One instruction per (quad of features, side, k4-step) leaves, in
//    lane (j, g), the partial sums over the lane group's k values of features 4 q .. 4 q + 3 for row j;
//  * the four k-groups' partial sums meet once per tile (v_permlane16/32_swap), get the bias, and are laid out as the
//    accumulator block they replace (feature 16 (NB-1) + 4 g + r in lane group g) — norm, layer-2 B operand and score then
//    see the block v3 computed, to rounding (another summation order: parity is against the fp64 oracle at the tests'
//    tolerance, not against v3's bits);
//  * layer 2 walks its NB - 1 full output blocks in groups of G2 as v5 does; group q also carries tail quad q (the group's
//    chunk holds the tail block's segments beside its own) and folds it into the score in lane group 0.
// Matrix-pipe time per 16 rows at D = 150: 1512 MFMAs of 8 passes + 336 of 2 (v3: 1680 of 8) = 0.95 of v3's; measured
// 3.09 against 3.19 ms per 1 M pairs (0.858 against 0.832 of the fp32 MFMA peak, same process: tools/exp_fwd.hip,
// profiles/r05x_exp_v6_groups.txt) with two accumulator chains per (quad, side) — four dependent 4x4x1 MFMAs in a row stall
// on each other (the first version: 3.155 ms) — 2 k16-steps per fence and layer-1 groups of 5 + 4 blocks.  Measured and NOT
// kept (profiles/r05u_exp_v6_tail.txt; the code is in the history, commit 5a1354d): the tail on the vector ALU —
// v_pk_fma_f32 from broadcast LDS reads (0.823; 0.840 as one burst per k16-step with the reads a step early) and
// v_fmac_f32 with DPP row-broadcast operands (0.788).  VALU instructions are not free beside MFMAs on this chip: each costs
// ~2.5 cycles of matrix-pipe time in bursts, 4 - 5 when interleaved, whichever of the SIMD's waves issues it
// (tools/exp_valu_phase.hip, profiles/r05u_exp_valu_phase.txt), and hipcc unpacks v_pk_fma_f32 it finds in an MFMA's
// shadow.  With no tail work at all the kernel runs in 2.91 ms (ABL = 1): 0.12 ms is what six features cost at the full
// blocks' rate, 0.18 ms is what they cost here.  At D = 170 (ten tail features, three quads) the form is behind v5.
#pragma once
#include "nplda_fwd_kernel.h"

namespace nplda {

// ABL (tools/exp_fwd.hip only; results are WRONG when non-zero): 1 = no tail work at all
template <int NB, int TF, int WAVES, int KPB = 2, int G1 = 5, int G2 = 3, int XM = 0, int ABL = 0, int NCH = 2>  // NCH: accumulator chains per (quad, side)
__global__ __launch_bounds__(WAVES * 64, WAVES / 4) void nplda_fwd_v6_kernel(const FwdArgs a, int ntiles) {
    constexpr int NM = NB - 1;                       // feature blocks on the 16 x 16 MFMA
    constexpr int NQ = (TF + 3) / 4;                 // tail features in quads
    constexpr int STEP4 = NB * 64;                   // float4 per k16-step of weights (the image's stride: all NB blocks)
    constexpr int CH1 = STEP4 * KPB;                 // layer-1 chunk
    constexpr int NSEG1 = CH1 / 64;
    constexpr int NGR = (NM + G1 - 1) / G1;          // MFMA groups of a layer-1 step; tail quad q rides with group q
    constexpr int NG = (NM + G2 - 1) / G2;           // layer-2 output groups; tail quad q rides with group q
    constexpr int NSEG2 = NB * (G2 + 1);             // a group's chunk: NB k-blocks x (its blocks + the tail block)
    constexpr int CH2 = NSEG2 * 64;
    constexpr int CH = CH1 > CH2 ? CH1 : CH2;
    static_assert(TF >= 1 && TF <= 16 && NM >= 1 && NQ <= NGR && NQ <= NG, "tail quads ride with the MFMA groups");
    __shared__ f32x4 wbuf[2][CH];
    __shared__ f32x4 cvec[4][NB * 4];
    __shared__ f32x4 sink[64];

    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int j = lane & 15;
    const int g = lane >> 4;
    const int tl = 16 * g + (j & 3);  // this lane's A operand of the 4x4x1 MFMAs: fragment lane 16 g + 4 q + j % 4 of the tail block

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

    auto seg_dma = [&](const f32x4* src, f32x4* dst) {  // see nplda_fwd_v5.h
        unsigned lo = (unsigned)lane * 16u;
        asm volatile("" : "+v"(lo));
        __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(reinterpret_cast<const char*>(src) + lo),
                                         (__attribute__((address_space(3))) void*)dst, 16, 0, 0);
    };
    auto dma_l1 = [&](int c, f32x4* dst) {
        const f32x4* src = Wall + (long long)c * CH1;
#pragma unroll
        for (int i = 0; i < (NSEG1 + WAVES - 1) / WAVES; ++i) {
            const int sgm = wave + WAVES * i;
            const bool live = sgm < NSEG1;
            seg_dma(src + (live ? sgm : 0) * 64, live ? dst + sgm * 64 : sink);
        }
    };
    // layer-2 chunk of output group [nb0, nb0 + gc): for every k-block kb the group's gc segments and the tail block's,
    // stored [kb][u][lane], u = gc being the tail block
    auto dma_l2 = [&](int nb0, int gc, f32x4* dst) {
        const int per = gc + 1;
        const int nseg = NB * per;
#pragma unroll
        for (int i = 0; i < (NSEG2 + WAVES - 1) / WAVES; ++i) {
            const int sgm = wave + WAVES * i;
            const bool live = sgm < nseg;
            const int sg = live ? sgm : 0;
            const int kb = sg / per, u = sg - kb * per;
            const int blk = u < gc ? nb0 + u : NM;
            seg_dma(Wall + w2base4 + (long long)kb * STEP4 + blk * 64, live ? dst + sgm * 64 : sink);
        }
    };
    auto chunk_fence = [&]() {
        __builtin_amdgcn_s_waitcnt(0x0070);  // vmcnt(0) lgkmcnt(0)
        __builtin_amdgcn_s_barrier();
        __builtin_amdgcn_sched_barrier(0);
    };
    // the four k-groups' partial sums of one tail feature -> the full sum in every lane of the row
    auto kgroups_sum = [&](float v) {
        v = wave_xor_add(v, 16);
        return wave_xor_add(v, 32);
    };
    // full tail values u[f] (the same in the four lane groups of a row) laid out as the accumulator block they stand for
    auto as_block = [&](const float (&u)[TF]) {
        f32x4 blk = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int r = 0; r < 4; ++r) {
#pragma unroll
            for (int gg = 0; gg < 4; ++gg)
                if (4 * gg + r < TF) blk[r] = (g == gg) ? u[4 * gg + r] : blk[r];
        }
        return blk;
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
        for (int nb = 0; nb < NM; ++nb) {
            accA[nb] = b1p[4 * nb + g];
            accB[nb] = accA[nb];
        }
        // two accumulators per (quad, side), for the even and the odd k4-steps: four dependent 4x4x1 MFMAs in a row stall on
        // each other (two passes each, a longer result latency — hipcc separates them by s_nop and keeps them together
        // wherever the source puts them); with two chains the dependent ones are four instructions apart
        f32x4 qA[NQ][NCH], qB[NQ][NCH];
#pragma unroll
        for (int q = 0; q < NQ; ++q) {
#pragma unroll
            for (int h = 0; h < NCH; ++h) {
                qA[q][h] = f32x4{0.f, 0.f, 0.f, 0.f};
                qB[q][h] = qA[q][h];
            }
        }

        // ---- layer 1 ---------------------------------------------------------------------------------------
        // one chunk: the MFMAs read (xc, yc) while (xn, yn) receive the next chunk's rows — the loop below alternates the two
        // register sets instead of copying the next set over the current one (v5 copies: 16 v_mov per chunk)
        auto chunk = [&](int c, f32x4 (&xc)[KPB], f32x4 (&yc)[KPB], f32x4 (&xn)[KPB], f32x4 (&yn)[KPB]) {
            const bool more = (c + 1 < NC1);
            f32x4* nxt = wbuf[par ^ 1];
            if (more) dma_l1(c + 1, nxt);
            else dma_l2(0, (G2 < NM ? G2 : NM), nxt);
#pragma unroll
            for (int s = 0; s < KPB; ++s) {
                const int ks = more ? KPB * (c + 1) + s : KPB * c + s;  // (after the last chunk: a harmless re-read, see v5)
                xn[s] = load_xrow<XM>(sa, 16 * ks + 4 * g, D0);
                yn[s] = load_xrow<XM>(sb, 16 * ks + 4 * g, D0);
            }
            const f32x4* w = wbuf[par];
#pragma unroll
            for (int s = 0; s < KPB; ++s) {
                if (KPB * c + s < KS1) {
#pragma unroll
                    for (int gr = 0; gr < NGR; ++gr) {
                        const int nb0 = gr * G1;
                        f32x4 av[G1];
#pragma unroll
                        for (int u = 0; u < G1; ++u)
                            if (nb0 + u < NM) av[u] = w[s * STEP4 + (nb0 + u) * 64 + lane];
#pragma unroll
                        for (int r = 0; r < 4; ++r) {
#pragma unroll
                            for (int u = 0; u < G1; ++u) {
                                if (nb0 + u < NM) {
                                    accA[nb0 + u] = __builtin_amdgcn_mfma_f32_16x16x4f32(av[u][r], xc[s][r], accA[nb0 + u], 0, 0, 0);
                                    accB[nb0 + u] = __builtin_amdgcn_mfma_f32_16x16x4f32(av[u][r], yc[s][r], accB[nb0 + u], 0, 0, 0);
                                }
                            }
                        }
                        if (gr < NQ && ABL != 1) {  // tail quad gr beside MFMA group gr
                            const f32x4 wq = w[s * STEP4 + NM * 64 + tl + 4 * gr];
#pragma unroll
                            for (int r = 0; r < 4; ++r) {
                                qA[gr][r % NCH] = __builtin_amdgcn_mfma_f32_4x4x1f32(wq[r], xc[s][r], qA[gr][r % NCH], 0, 0, 0);
                                qB[gr][r % NCH] = __builtin_amdgcn_mfma_f32_4x4x1f32(wq[r], yc[s][r], qB[gr][r % NCH], 0, 0, 0);
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

        // ---- the tail features join the blocks: sum over the k-groups, bias, accumulator layout ---------------------
        {
            float uA[TF], uB[TF];
#pragma unroll
            for (int f = 0; f < TF; ++f) {
                float sa_ = qA[f / 4][0][f % 4], sb_ = qB[f / 4][0][f % 4];
#pragma unroll
                for (int h = 1; h < NCH; ++h) {
                    sa_ += qA[f / 4][h][f % 4];
                    sb_ += qB[f / 4][h][f % 4];
                }
                uA[f] = kgroups_sum(sa_);
                uB[f] = kgroups_sum(sb_);
            }
            const f32x4 bt = b1p[4 * NM + g];  // zero beyond D1 (the image pads the bias)
            accA[NM] = as_block(uA) + bt;
            accB[NM] = as_block(uB) + bt;
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

        // ---- layer 2, output groups of G2 full blocks (+ tail quad gi); the score is folded group by group --------------
        float part = 0.f;
#pragma unroll
        for (int gi = 0; gi < NG; ++gi) {
            const int nb0 = gi * G2;
            const int gc = (NM - nb0) < G2 ? (NM - nb0) : G2;
            const int per = gc + 1;
            f32x4 zA[G2], zB[G2];
#pragma unroll
            for (int u = 0; u < G2; ++u) {
                if (u < gc) {
                    zA[u] = b2p[4 * (nb0 + u) + g];
                    zB[u] = zA[u];
                }
            }
            f32x4 zqA[2] = {{0.f, 0.f, 0.f, 0.f}, {0.f, 0.f, 0.f, 0.f}}, zqB[2] = {{0.f, 0.f, 0.f, 0.f}, {0.f, 0.f, 0.f, 0.f}};
            f32x4* nxt = wbuf[par ^ 1];
            if (gi + 1 < NG) {
                const int nb1 = nb0 + G2;
                dma_l2(nb1, (NM - nb1) < G2 ? (NM - nb1) : G2, nxt);
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
            for (int kb = 0; kb < NB; ++kb) {
                f32x4 av[G2];
#pragma unroll
                for (int u = 0; u < G2; ++u)
                    if (u < gc) av[u] = w[(kb * per + u) * 64 + lane];
#pragma unroll
                for (int r = 0; r < 4; ++r) {
#pragma unroll
                    for (int u = 0; u < G2; ++u) {
                        if (u < gc) {
                            zA[u] = __builtin_amdgcn_mfma_f32_16x16x4f32(av[u][r], accA[kb][r], zA[u], 0, 0, 0);
                            zB[u] = __builtin_amdgcn_mfma_f32_16x16x4f32(av[u][r], accB[kb][r], zB[u], 0, 0, 0);
                        }
                    }
                }
                if (gi < NQ && ABL != 1) {
                    const f32x4 wq = w[(kb * per + gc) * 64 + tl + 4 * gi];
#pragma unroll
                    for (int r = 0; r < 4; ++r) {
                        zqA[r & 1] = __builtin_amdgcn_mfma_f32_4x4x1f32(wq[r], accA[kb][r], zqA[r & 1], 0, 0, 0);
                        zqB[r & 1] = __builtin_amdgcn_mfma_f32_4x4x1f32(wq[r], accB[kb][r], zqB[r & 1], 0, 0, 0);
                    }
                }
            }
            chunk_fence();
            par ^= 1;
#pragma unroll
            for (int u = 0; u < G2; ++u) {
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
            // the group's tail quad: full sums in every lane group, counted once (lane group 0)
            if (gi < NQ) {
                float tpart = 0.f;
                const float* b2s = reinterpret_cast<const float*>(b2p);
                const float* Qs = reinterpret_cast<const float*>(Qp);
                const float* Ps = reinterpret_cast<const float*>(Pp);
#pragma unroll
                for (int i = 0; i < 4; ++i) {
                    const int f = 4 * gi + i;
                    if (f < TF) {
                        const float z1 = kgroups_sum(zqA[0][i] + zqA[1][i]) + b2s[16 * NM + f];
                        const float z2 = kgroups_sum(zqB[0][i] + zqB[1][i]) + b2s[16 * NM + f];
                        tpart = fmaf(Qs[16 * NM + f], fmaf(z1, z1, z2 * z2), tpart);
                        tpart = fmaf(2.0f * Ps[16 * NM + f], z1 * z2, tpart);
                    }
                }
                part += (g == 0) ? tpart : 0.f;
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

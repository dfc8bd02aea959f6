// nplda_fwd_v6.h — the persistent pair-scoring schedule of nplda_fwd_v5.h with the PARTIAL feature block on the VALU.
//
// D = 150 is nine full 16-feature blocks and six features; D = 170 is ten blocks and ten features.  v3 / v5 run the
// left-over features as one more MFMA block (NB = 10 / 11), i.e. 10 / 16 (6 / 16) of that block's matrix-pipe time
// multiplies zeros: 1680 MFMAs per 16 rows where 150 features need 1552 (PMC: the kernels execute exactly the padded
// count, and the matrix pipe is what bounds them).  On CDNA3/4 the vector ALU's packed fp32 FMA (v_pk_fma_f32: two
// FMAs per lane) has the same peak as the fp32 matrix pipe and runs BESIDE it — a wave's VALU instructions issue in
// the shadow of its own MFMAs — so the TF left-over features of both layers are computed there:
//  * the x fragment a lane holds for the MFMAs (row j, columns 16 ks + 4 g + r) is also what the lane needs for a
//    partial dot product of row j with a tail row of W over the lane's own k values; the tail rows are rows 0 .. TF-1
//    of the image's last feature block, already in LDS with the chunk (lane group g reads the 16 bytes of row f at
//    fragment lane 16 g + f: one address per 16 lanes, a broadcast read).  Two pk_fma per (feature, side) and k16-step:
//    4 TF VALU instructions beside 8 (NB - 1) MFMAs;
//  * the four k-groups' partial sums meet once per tile (v_permlane16/32_swap), get the bias, and are laid out as the
//    MFMA accumulator block they replace (feature 16 (NB-1) + 4 g + r in lane group g) — norm, layer-2 B operand and
//    score code then see the block v5 computed, to rounding (a different summation order: parity is against the fp64
//    oracle at the tolerance of the tests, not against v3's bits);
//  * layer 2 walks its NB - 1 full output blocks in groups of G2 as v5 does; each group also carries its share of the
//    TF tail OUTPUT features on the VALU (the group's chunk holds the tail block's segments beside its own), and folds
//    them into the score in lane group 0.
// MFMAs per 16 rows: (NB-1) (4 KS1 + 4 NB) = 1512 at D = 150 (v3: 1680), 1720 at D = 170 (v5: 1892).
#pragma once
#include "nplda_fwd_kernel.h"

namespace nplda {

typedef float f32x2 __attribute__((ext_vector_type(2)));
__device__ __forceinline__ f32x2 pk_fma2(f32x2 a, f32x2 b, f32x2 c) { return __builtin_elementwise_fma(a, b, c); }
// acc += a * b on both halves, kept packed: hipcc's post-RA pass unpacks v_pk_fma_f32 into two v_fma_f32 when it finds it
// in an MFMA's shadow (the packed form does not co-issue) — here the instruction COUNT is what costs (every VALU
// instruction takes ~2.5 cycles of matrix-pipe time whichever wave issues it: tools/exp_valu_phase.hip)
__device__ __forceinline__ void pk_fma_acc(f32x2& acc, f32x2 a, f32x2 b) {
    asm volatile("v_pk_fma_f32 %0, %1, %2, %0" : "+v"(acc) : "v"(a), "v"(b));
}
// lane n of every 16-lane row, in all lanes of that row (DPP row_newbcast:n, gfx90a+): an operand modifier of the FMA
// that consumes it (v_fmac_f32_dpp), not an instruction of its own
template <int N>
__device__ __forceinline__ float row_bcast_c(float v) {
    return __uint_as_float((unsigned)__builtin_amdgcn_update_dpp(0, (int)__float_as_uint(v), 0x150 + N, 0xf, 0xf, true));
}
__device__ __forceinline__ float row_bcast(float v, int n) {  // n is a constant after unrolling
    switch (n) {
        case 0: return row_bcast_c<0>(v);   case 1: return row_bcast_c<1>(v);   case 2: return row_bcast_c<2>(v);
        case 3: return row_bcast_c<3>(v);   case 4: return row_bcast_c<4>(v);   case 5: return row_bcast_c<5>(v);
        case 6: return row_bcast_c<6>(v);   case 7: return row_bcast_c<7>(v);   case 8: return row_bcast_c<8>(v);
        case 9: return row_bcast_c<9>(v);   case 10: return row_bcast_c<10>(v); case 11: return row_bcast_c<11>(v);
        case 12: return row_bcast_c<12>(v); case 13: return row_bcast_c<13>(v); case 14: return row_bcast_c<14>(v);
        default: return row_bcast_c<15>(v);
    }
}

template <int NB, int TF, int WAVES, int KPB = 2, int G1 = 3, int G2 = 3, int XM = 0, int ABL = 0, int TM = 1>  // TM: 1 tail rows by DPP row broadcast of the block's own fragment, 0 by broadcast LDS reads + pk_fma; ABL: tools/exp_fwd.hip only (wrong results): 1 no tail work, 2 tail reads but no FMAs
__global__ __launch_bounds__(WAVES * 64, WAVES / 4) void nplda_fwd_v6_kernel(const FwdArgs a, int ntiles) {
    constexpr int NM = NB - 1;                       // feature blocks on the matrix pipe
    static_assert(TF >= 1 && TF <= 16 && NM >= 1, "tail features live in the image's last block");
    constexpr int STEP4 = NB * 64;                   // float4 per k16-step of weights (the image's stride: all NB blocks)
    constexpr int CH1 = STEP4 * KPB;                 // layer-1 chunk
    constexpr int NSEG1 = CH1 / 64;
    constexpr int NG = (NM + G2 - 1) / G2;           // layer-2 output groups
    constexpr int TG = (TF + NG - 1) / NG;           // tail output features per group
    constexpr int NSEG2 = NB * (G2 + 1);             // a group's chunk: NB k-blocks x (its blocks + the tail block)
    constexpr int CH2 = NSEG2 * 64;
    constexpr int CH = CH1 > CH2 ? CH1 : CH2;
    __shared__ f32x4 wbuf[2][CH];
    __shared__ f32x4 cvec[4][NB * 4];
    __shared__ f32x4 sink[64];

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
    auto kgroups_sum = [&](f32x2 t) {
        float v = t[0] + t[1];
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
        f32x2 tA[TF], tB[TF];
#pragma unroll
        for (int f = 0; f < TF; ++f) {
            tA[f] = f32x2{0.f, 0.f};
            tB[f] = tA[f];
        }
        constexpr int NQ = (TF + 3) / 4;  // TM == 2: tail features in quads, one 4x4x1 MFMA (16 blocks) per quad, side and k4-step
        f32x4 qA[NQ], qB[NQ];
#pragma unroll
        for (int q = 0; q < NQ; ++q) {
            qA[q] = f32x4{0.f, 0.f, 0.f, 0.f};
            qB[q] = qA[q];
        }

        // ---- layer 1 ---------------------------------------------------------------------------------------
        for (int c = 0; c < NC1; ++c) {
            const bool more = (c + 1 < NC1);
            f32x4* nxt = wbuf[par ^ 1];
            if (more) dma_l1(c + 1, nxt);
            else dma_l2(0, (G2 < NM ? G2 : NM), nxt);
            f32x4 xan[KPB], xbn[KPB];
#pragma unroll
            for (int s = 0; s < KPB; ++s) {
                const int ks = more ? KPB * (c + 1) + s : KPB * c + s;
                xan[s] = load_xrow<XM>(sa, 16 * ks + 4 * g, D0);
                xbn[s] = load_xrow<XM>(sb, 16 * ks + 4 * g, D0);
            }
            const f32x4* w = wbuf[par];
#pragma unroll
            for (int s = 0; s < KPB; ++s) {
                if (KPB * c + s < KS1) {
                    constexpr int NGR = (NM + G1 - 1) / G1;          // MFMA groups of a step
                    static_assert(TM != 2 || (TF + 3) / 4 <= NGR, "a tail quad per MFMA group");
                    constexpr int TS = (TF + NGR - 1) / NGR;         // tail features dealt with beside each group
                    f32x4 wtf;
                    f32x4 wt3[TF];
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
                                    accA[nb0 + u] = __builtin_amdgcn_mfma_f32_16x16x4f32(av[u][r], xa[s][r], accA[nb0 + u], 0, 0, 0);
                                    accB[nb0 + u] = __builtin_amdgcn_mfma_f32_16x16x4f32(av[u][r], xb[s][r], accB[nb0 + u], 0, 0, 0);
                                }
                            }
                        }
                        if (TM == 3) {
                            if (gr == 0 && ABL != 1) {  // the step's tail rows are read behind group 0's fragments ...
#pragma unroll
                                for (int f = 0; f < TF; ++f) wt3[f] = w[s * STEP4 + NM * 64 + 16 * g + f];
                                __builtin_amdgcn_sched_barrier(0);
                            }
                            if (gr == NGR - 1 && ABL == 0) {  // ... and used in ONE burst behind the step's last MFMA
                                __builtin_amdgcn_sched_barrier(0);
#pragma unroll
                                for (int f = 0; f < TF; ++f) pk_fma_acc(tA[f], xa[s].xy, wt3[f].xy);
#pragma unroll
                                for (int f = 0; f < TF; ++f) pk_fma_acc(tB[f], xb[s].xy, wt3[f].xy);
#pragma unroll
                                for (int f = 0; f < TF; ++f) pk_fma_acc(tA[f], xa[s].zw, wt3[f].zw);
#pragma unroll
                                for (int f = 0; f < TF; ++f) pk_fma_acc(tB[f], xb[s].zw, wt3[f].zw);
                                __builtin_amdgcn_sched_barrier(0);
                            }
                            if (gr == NGR - 1 && ABL == 2) {
#pragma unroll
                                for (int f = 0; f < TF; ++f) asm volatile("" :: "v"(wt3[f]));
                            }
                            continue;
                        }
                        if (TM == 1 && gr == 0 && ABL != 1) wtf = w[s * STEP4 + NM * 64 + lane];
                        if (TM == 2 && ABL != 1) {
                            // quad q beside MFMA group q: block (g, j / 4) of the 4x4x1 MFMA holds rows 4 (j / 4) .. + 3 of the
                            // x fragment as its B operand and features 4 q .. 4 q + 3 (fragment lanes 16 g + 4 q + j % 4) as A
                            if (gr < NQ) {
                                const f32x4 wq = w[s * STEP4 + NM * 64 + 16 * g + 4 * gr + (j & 3)];
#pragma unroll
                                for (int r = 0; r < 4; ++r) {
                                    qA[gr] = __builtin_amdgcn_mfma_f32_4x4x1f32(wq[r], xa[s][r], qA[gr], 0, 0, 0);
                                    qB[gr] = __builtin_amdgcn_mfma_f32_4x4x1f32(wq[r], xb[s][r], qB[gr], 0, 0, 0);
                                }
                            }
                            continue;
                        }
#pragma unroll
                        for (int tt = 0; tt < TS; ++tt) {
                            const int f = gr * TS + tt;
                            if (f < TF && ABL != 1) {
                                if (TM == 1) {
                                    if (ABL == 2) { asm volatile("" :: "v"(wtf)); continue; }
#pragma unroll
                                    for (int r = 0; r < 4; ++r) {
                                        const float wv = row_bcast(wtf[r], f);
                                        tA[f][0] = fmaf(wv, xa[s][r], tA[f][0]);
                                        tB[f][0] = fmaf(wv, xb[s][r], tB[f][0]);
                                    }
                                    continue;
                                }
                                f32x4 wt = w[s * STEP4 + NM * 64 + 16 * g + f];
                                if (ABL == 2) { asm volatile("" :: "v"(wt)); continue; }
                                tA[f] = pk_fma2(xa[s].xy, wt.xy, tA[f]);
                                tB[f] = pk_fma2(xb[s].xy, wt.xy, tB[f]);
                                tA[f] = pk_fma2(xa[s].zw, wt.zw, tA[f]);
                                tB[f] = pk_fma2(xb[s].zw, wt.zw, tB[f]);
                            }
                        }
                    }
                }
            }
#pragma unroll
            for (int s = 0; s < KPB; ++s) {
                xa[s] = xan[s];
                xb[s] = xbn[s];
            }
            chunk_fence();
            par ^= 1;
        }

        // ---- the tail features join the blocks: sum over the k-groups, bias, accumulator layout ---------------------
        {
            float uA[TF], uB[TF];
#pragma unroll
            for (int f = 0; f < TF; ++f) {
                if (TM == 2) {
                    uA[f] = kgroups_sum(f32x2{qA[f / 4][f % 4], 0.f});
                    uB[f] = kgroups_sum(f32x2{qB[f / 4][f % 4], 0.f});
                } else {
                    uA[f] = kgroups_sum(tA[f]);
                    uB[f] = kgroups_sum(tB[f]);
                }
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

        // ---- layer 2, output groups of G2 full blocks + TG tail features; the score is folded group by group ----------
        float part = 0.f;
#pragma unroll
        for (int gi = 0; gi < NG; ++gi) {
            const int nb0 = gi * G2;
            const int gc = (NM - nb0) < G2 ? (NM - nb0) : G2;
            const int per = gc + 1;
            const int f0 = gi * TG;
            const int fc = (TF - f0) < TG ? (TF - f0 > 0 ? TF - f0 : 0) : TG;
            f32x4 zA[G2], zB[G2];
#pragma unroll
            for (int u = 0; u < G2; ++u) {
                if (u < gc) {
                    zA[u] = b2p[4 * (nb0 + u) + g];
                    zB[u] = zA[u];
                }
            }
            f32x2 ztA[TG], ztB[TG];
#pragma unroll
            for (int t = 0; t < TG; ++t) {
                ztA[t] = f32x2{0.f, 0.f};
                ztB[t] = ztA[t];
            }
            static_assert(TM != 2 || (TF + 3) / 4 <= NG, "a tail quad per layer-2 output group");
            f32x4 zqA = {0.f, 0.f, 0.f, 0.f}, zqB = zqA;  // TM == 2: group gi carries tail quad gi
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
                f32x4 wz3[TG];
                if (TM == 3 && ABL != 1) {
#pragma unroll
                    for (int t = 0; t < TG; ++t)
                        if (t < fc) wz3[t] = w[(kb * per + gc) * 64 + 16 * g + f0 + t];
                    __builtin_amdgcn_sched_barrier(0);
                }
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
                f32x4 wtf;
                if (TM == 1 && fc > 0 && ABL != 1) wtf = w[(kb * per + gc) * 64 + lane];
                if (TM == 3) {
                    if (ABL == 0) {
                        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
                        for (int t = 0; t < TG; ++t) if (t < fc) pk_fma_acc(ztA[t], accA[kb].xy, wz3[t].xy);
#pragma unroll
                        for (int t = 0; t < TG; ++t) if (t < fc) pk_fma_acc(ztB[t], accB[kb].xy, wz3[t].xy);
#pragma unroll
                        for (int t = 0; t < TG; ++t) if (t < fc) pk_fma_acc(ztA[t], accA[kb].zw, wz3[t].zw);
#pragma unroll
                        for (int t = 0; t < TG; ++t) if (t < fc) pk_fma_acc(ztB[t], accB[kb].zw, wz3[t].zw);
                        __builtin_amdgcn_sched_barrier(0);
                    }
                    if (ABL == 2) {
#pragma unroll
                        for (int t = 0; t < TG; ++t) if (t < fc) asm volatile("" :: "v"(wz3[t]));
                    }
                    continue;
                }
                if (TM == 2) {
                    if (gi < NQ && ABL != 1) {
                        const f32x4 wq = w[(kb * per + gc) * 64 + 16 * g + 4 * gi + (j & 3)];
#pragma unroll
                        for (int r = 0; r < 4; ++r) {
                            zqA = __builtin_amdgcn_mfma_f32_4x4x1f32(wq[r], accA[kb][r], zqA, 0, 0, 0);
                            zqB = __builtin_amdgcn_mfma_f32_4x4x1f32(wq[r], accB[kb][r], zqB, 0, 0, 0);
                        }
                    }
                    continue;
                }
#pragma unroll
                for (int t = 0; t < TG; ++t) {
                    if (t < fc && ABL != 1) {
                        if (TM == 1) {
                            if (ABL == 2) { asm volatile("" :: "v"(wtf)); continue; }
#pragma unroll
                            for (int r = 0; r < 4; ++r) {
                                const float wv = row_bcast(wtf[r], f0 + t);
                                ztA[t][0] = fmaf(wv, accA[kb][r], ztA[t][0]);
                                ztB[t][0] = fmaf(wv, accB[kb][r], ztB[t][0]);
                            }
                            continue;
                        }
                        f32x4 wt = w[(kb * per + gc) * 64 + 16 * g + f0 + t];
                        if (ABL == 2) { asm volatile("" :: "v"(wt)); continue; }
                        ztA[t] = pk_fma2(accA[kb].xy, wt.xy, ztA[t]);
                        ztB[t] = pk_fma2(accB[kb].xy, wt.xy, ztB[t]);
                        ztA[t] = pk_fma2(accA[kb].zw, wt.zw, ztA[t]);
                        ztB[t] = pk_fma2(accB[kb].zw, wt.zw, ztB[t]);
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
            // the group's tail features: full sums in every lane group, counted once (lane group 0)
            float tpart = 0.f;
            if (TM == 2) {
                if (gi < NQ) {
#pragma unroll
                    for (int i = 0; i < 4; ++i) {
                        const int f = 4 * gi + i;
                        if (f < TF) {
                            const float* b2s = reinterpret_cast<const float*>(b2p);
                            const float* Qs = reinterpret_cast<const float*>(Qp);
                            const float* Ps = reinterpret_cast<const float*>(Pp);
                            const float z1 = kgroups_sum(f32x2{zqA[i], 0.f}) + b2s[16 * NM + f];
                            const float z2 = kgroups_sum(f32x2{zqB[i], 0.f}) + b2s[16 * NM + f];
                            tpart = fmaf(Qs[16 * NM + f], fmaf(z1, z1, z2 * z2), tpart);
                            tpart = fmaf(2.0f * Ps[16 * NM + f], z1 * z2, tpart);
                        }
                    }
                }
            } else
#pragma unroll
            for (int t = 0; t < TG; ++t) {
                if (t < fc) {
                    const int f = f0 + t;
                    const float* b2s = reinterpret_cast<const float*>(b2p);
                    const float* Qs = reinterpret_cast<const float*>(Qp);
                    const float* Ps = reinterpret_cast<const float*>(Pp);
                    const float z1 = kgroups_sum(ztA[t]) + b2s[16 * NM + f];
                    const float z2 = kgroups_sum(ztB[t]) + b2s[16 * NM + f];
                    tpart = fmaf(Qs[16 * NM + f], fmaf(z1, z1, z2 * z2), tpart);
                    tpart = fmaf(2.0f * Ps[16 * NM + f], z1 * z2, tpart);
                }
            }
            part += (g == 0) ? tpart : 0.f;
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

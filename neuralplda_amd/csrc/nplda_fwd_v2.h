// nplda_fwd_v2.h — second schedule of the fused forward (same arithmetic/layout as nplda_fwd_kernel.h).
//
// Ablations on MI355X (tools/exp_fwd.hip, profiles/r01d_ablation.txt) put the v1 kernel at 0.72 of the fp32
// MFMA peak, the barrier + weight-staging work at 10 % and the x loads at 4 %; LDS fragment reads cost nothing.
// v2 therefore keeps v1's structure but
//  * streams KPB k16-steps of weights per barrier (product: 2; 4 was best before the post-load select was removed,
//    see load_x4c) through two
//    LDS buffers, staging each chunk through registers in TWO halves (load first half at chunk start, store it
//    and load the second half mid-chunk, store that at the end) so the staging registers do not grow;
//  * gives x its own 4-slot ring: slot s is reloaded with k16-step s of the NEXT chunk right after its last
//    MFMA, i.e. a prefetch distance of a whole 4-step chunk (v1: 2 steps) at the same 32 registers.
#pragma once
#include "nplda_fwd_kernel.h"

namespace nplda {

template <int NB, int MODE, int WAVES, bool NT, int KPB, int G = 4>
__global__ __launch_bounds__(WAVES * 64, 2) void nplda_fwd_v2_kernel(const FwdArgs a) {
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

    const f32x4* Wall = reinterpret_cast<const f32x4*>(a.packed);
    const f32x4* b1p = reinterpret_cast<const f32x4*>(a.packed + a.ob1);
    const f32x4* b2p = reinterpret_cast<const f32x4*>(a.packed + a.ob2);
    const f32x4* Qp = reinterpret_cast<const f32x4*>(a.packed + a.oQ);
    const f32x4* Pp = reinterpret_cast<const f32x4*>(a.packed + a.oP);
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
        xa[s] = load_x4c<NT>(sa, 16 * s + 4 * g, D0);
        xb[s] = load_x4c<NT>(sb, 16 * s + 4 * g, D0);
    }
    f32x4 accA[NB], accB[NB];
#pragma unroll
    for (int nb = 0; nb < NB; ++nb) {
        accA[nb] = b1p[4 * nb + g];
        accB[nb] = accA[nb];
    }
    store1(wbuf[0]);
    load2(0);
    store2(wbuf[0]);
    __syncthreads();

    // ---- layer 1 -----------------------------------------------------------------------------------------
    for (int c = 0; c < NC1; ++c) {
        const int cur = c & 1;
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
            {   // slot s is free: fetch k16-step s of the next chunk (a whole chunk ahead of its use)
                const int ks = KPB * (c + 1) + s;
                xa[s] = load_x4c<NT>(sa, 16 * ks + 4 * g, D0);
                xb[s] = load_x4c<NT>(sb, 16 * ks + 4 * g, D0);
            }
            if (s == SMID - 1) {
                store1(wbuf[cur ^ 1]);
                load2(nbase);
            }
        }
        store2(wbuf[cur ^ 1]);
        __syncthreads();
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
    if (MODE == MODE_EMBED && a.out_y != nullptr) {  // embedding rows saved for nplda_embed_backward_f32
#pragma unroll
        for (int nb = 0; nb < NB; ++nb) {
            if (okA) *reinterpret_cast<f32x4*>(a.out_y + rowA * a.ldz + 16 * nb + 4 * g) = accA[nb];
            if (okB) *reinterpret_cast<f32x4*>(a.out_y + rowB * a.ldz + 16 * nb + 4 * g) = accB[nb];
        }
        if (g == 0) {
            if (okA) a.out_rn[rowA] = invA;
            if (okB) a.out_rn[rowB] = invB;
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
        const int cur = (NC1 + c2) & 1;
        const bool more2 = (c2 + 1 < NC2);
        const long long nbase = w2base4 + (long long)(c2 + 1) * CH;
        if (more2) load1(nbase);
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
            if (s == SMID - 1 && more2) {
                store1(wbuf[cur ^ 1]);
                load2(nbase);
            }
        }
        if (more2) {
            store2(wbuf[cur ^ 1]);
            __syncthreads();
        }
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
            // (rows moved to the lanes' high bits for 64-byte pieces — nplda_common.h — measured here and no faster: with two
            // blocks per CU the stores drain under the other block's MFMAs; round 6)
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

}  // namespace nplda

// nplda_fwd_v4.h — EXPERIMENT (not built into the library): pair scoring with the left-over features of each layer on the VALU (split image, NpldaSplit).
//
// The v3 schedule is MFMA-bound (time scales with the MFMA count: D = 128 runs at 0.876 of peak with no padding), and
// a layer width that is not a multiple of 16 pays a whole padded feature block in BOTH layers: D = 150 issues
// 32*10*8 + 10*10*8 = 3360 MFMAs per 16-pair tile where 150/160 of that is algorithmic.  Here only the NBF = D / 16
// full blocks run on MFMA (32*9*8 + 9*9*8 = 2952, -12 %) and the LO = D % 16 left-over features are carried by plain
// FMAs that the VALU has idle cycles for (MFMA occupies the matrix pipe for 32 cycles per instruction):
//   layer 1   u_f = b1_f + sum_k W1[F0+f][k] x[k]: lane (pair j, quad g) holds x[16ks+4g .. +3] of every k16-step, so it
//             accumulates its 4-element slice of the dot product per step (weights: one broadcast ds_read_b128 per
//             feature from the tail of the streamed step) and the four g-slices are summed once per tile (2 xor-adds);
//   normalise the left-over squares join the row norm; y_f = u_f / max(||u||, eps), replicated over g;
//   layer 2   z_n (n in the MFMA blocks) += sum_f W2[n][F0+f] y_f from an LDS-resident table (T1), BEFORE the k loop so
//             that y_f dies early; z_f' (left-over outputs) = b2 + sum_f W2[F0+f'][F0+f] y_f + the per-lane slices of
//             sum_{k < 16 NBF} W2[F0+f'][k] y[k], y[k] being this lane's layer-1 accumulator elements, summed over g
//             once per tile;
//   epilogue  the left-over z join the quadratic form after the cross-lane sum.
// RESULT (MI355X, D = 150: NBF = 9, LO = 6): bit-compatible to 4e-7 relative, but SLOWER than v3 (3.32 ms vs 3.20 ms) although
// it issues 12 % fewer MFMAs.  The premise is wrong on this hardware: VALU instructions are not hidden under MFMAs —
// tools/mfma_peak.hip shows every wave64 VALU FMA costing ~3.6 MFMA-pipe cycles (64 FMAs per 64 MFMAs: 0.990 -> 0.891 of
// peak, at 1, 2 or 4 waves per SIMD) — so the ~150 extra VALU instructions per 144 MFMAs cost more than the padded block.
// The same measurement explains why every VALU instruction in the main loops of v2 / v3 (address arithmetic, selects) is
// paid for in full, and why removing them mattered.
// Everything else is v3: persistent grid, continuous weight stream (KPB steps of SS4 = NBF*64 + LO*4 float4 per
// barrier), x ring refilled a chunk ahead and across tiles, constants resident in LDS.
#pragma once
#include "../../neuralplda_amd/csrc/nplda_fwd_kernel.h"

// ---- split image (appended behind the standard packed image) ------------------------------------------------
// "Split" image, appended behind the standard one (its offsets are absolute, in floats).  A layer width D that is not a
// multiple of 16 costs a whole padded 16-feature MFMA block (150 -> 160: 7 % of the kernel, 170 -> 176).  When
// D1 == D2 == 16 NBF + LO with 1 <= LO <= 10, the pair-scoring kernel of nplda_fwd_v4.h runs only the NBF full blocks
// on MFMA and the LO left-over features on the otherwise idle VALU, from this image:
//   A (layer 1), KS1 steps of SS4 float4:  [ks][nb < NBF][lane][i]  as W1p, then  [f < LO][g < 4][i] = W1[F0+f][16ks+4g+i]
//   B (layer 2), NBF steps of SS4 float4:  [kb][nb < NBF][lane][i]  as W2p, then LO*4 unused float4 (same stride as A)
//   T1: [nb < NBF][g < 4][f < LO][r < 4] = W2[16nb+4g+r][F0+f]        (left-over INPUT features of layer 2)
//   T3: [f < LO][kb < NBF][g < 4][i < 4] = W2[F0+f][16kb+4g+i]        (left-over OUTPUT features of layer 2)
//   T2: W2[F0+f'][F0+f] (LO x LO), then b1, b2, Q, P = P_sqrt^2 of the LO left-over features
// with F0 = 16 NBF; one chunk of slack behind T2 for the unconditional chunk loads.
struct NpldaSplit {
    int NBF, LO, SS4;  // LO == 0: this model has no split image
    size_t oA, oB, oT1, oT3, oT2, total;
};

__host__ __device__ inline NpldaSplit nplda_split(const NpldaLayout& L) {
    NpldaSplit S;
    S.NBF = 0; S.LO = 0; S.SS4 = 0;
    S.oA = S.oB = S.oT1 = S.oT3 = S.oT2 = S.total = L.total;
    const int lo = L.D1 % 16, nbf = L.D1 / 16;
    if (L.D1 != L.D2 || lo < 1 || lo > 10 || nbf < 1) return S;
    S.NBF = nbf; S.LO = lo; S.SS4 = nbf * 64 + lo * 4;
    S.oA = L.total;
    S.oB = S.oA + (size_t)L.KS1 * S.SS4 * 4;
    S.oT1 = S.oB + (size_t)nbf * S.SS4 * 4;
    S.oT3 = S.oT1 + (size_t)nbf * 4 * lo * 4;
    S.oT2 = S.oT3 + (size_t)lo * nbf * 4 * 4;
    S.total = S.oT2 + (size_t)((lo * lo + 4 * lo + 3) / 4 * 4) + (size_t)2 * S.SS4 * 4;
    return S;
}

namespace nplda {

struct SplitOff { size_t oA, oB, oT1, oT3, oT2; };

// the split image (NpldaSplit) behind the standard one
static __global__ void nplda_pack_split_kernel(const float* __restrict__ W1, const float* __restrict__ b1,
                                               const float* __restrict__ W2, const float* __restrict__ b2,
                                               const float* __restrict__ P_sqrt, const float* __restrict__ Q,
                                               NpldaLayout L, NpldaSplit S, float* __restrict__ out) {
    const size_t idx = S.oA + (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (idx >= S.total) return;
    const int NBF = S.NBF, LO = S.LO, F0 = 16 * S.NBF, D = L.D1;
    float v = 0.f;
    if (idx < S.oT1) {  // regions A / B: steps of SS4 float4
        const bool isB = idx >= S.oB;
        const size_t rel = isB ? idx - S.oB : idx - S.oA;
        const int step = (int)(rel / ((size_t)S.SS4 * 4));
        const int w = (int)(rel % ((size_t)S.SS4 * 4));
        const int kmax = isB ? D : L.D0;
        const float* W = isB ? W2 : W1;
        if (w < NBF * 256) {
            const int i = w & 3, lane = (w >> 2) & 63, nb = w >> 8;
            const int f = 16 * nb + (lane & 15), k = 16 * step + 4 * (lane >> 4) + i;
            if (k < kmax) v = W[(size_t)f * kmax + k];
        } else if (!isB) {  // layer-1 left-over rows ride behind each streamed step (B keeps the stride, unused)
            const int q = w - NBF * 256;
            const int i = q & 3, g = (q >> 2) & 3, f = q >> 4;
            const int k = 16 * step + 4 * g + i;
            if (f < LO && k < kmax) v = W[(size_t)(F0 + f) * kmax + k];
        }
    } else if (idx < S.oT3) {
        const int q = (int)(idx - S.oT1);
        const int r = q & 3, f = (q >> 2) % LO, ng = (q >> 2) / LO;  // ng = nb * 4 + g
        const int n = 16 * (ng >> 2) + 4 * (ng & 3) + r;
        v = W2[(size_t)n * D + F0 + f];
    } else if (idx < S.oT2) {
        const int q = (int)(idx - S.oT3);
        const int i = q & 3, g = (q >> 2) & 3, kb = (q >> 4) % NBF, f = (q >> 4) / NBF;
        v = W2[(size_t)(F0 + f) * D + 16 * kb + 4 * g + i];
    } else {
        const int q = (int)(idx - S.oT2);
        if (q < LO * LO) v = W2[(size_t)(F0 + q / LO) * D + F0 + q % LO];
        else if (q < LO * LO + LO) v = b1[F0 + q - LO * LO];
        else if (q < LO * LO + 2 * LO) v = b2[F0 + q - LO * LO - LO];
        else if (q < LO * LO + 3 * LO) v = Q[F0 + q - LO * LO - 2 * LO];
        else if (q < LO * LO + 4 * LO) { const float ps = P_sqrt[F0 + q - LO * LO - 3 * LO]; v = ps * ps; }
    }
    out[idx] = v;
}

#define LDS_FENCE() asm volatile("" ::: "memory")

template <int NBF, int LO, int WAVES, int KPB, int G = 4>
__global__ __launch_bounds__(WAVES * 64, 2) void nplda_fwd_v4_kernel(const FwdArgs a, const SplitOff so, int ntiles) {
    constexpr int THREADS = WAVES * 64;
    constexpr int SS4 = NBF * 64 + LO * 4;  // float4 per streamed step
    constexpr int CH = SS4 * KPB;
    constexpr int HALF = ((CH / 2 + THREADS - 1) / THREADS) * THREADS;
    constexpr int NS1 = HALF / THREADS;
    constexpr int NS2 = (CH - HALF + THREADS - 1) / THREADS;
    constexpr int NS = NS1 > NS2 ? NS1 : NS2;
    constexpr int NC2 = (NBF + KPB - 1) / KPB;
    static_assert(HALF <= CH && KPB >= 2, "chunk must split into two staging halves");
    constexpr int SMID = KPB / 2;
    constexpr int NT2 = (LO * LO + 4 * LO + 3) / 4;  // float4 of the T2 table
    __shared__ f32x4 wbuf[2][CH];
    __shared__ f32x4 cvec[4][NBF * 4];     // b1, b2, Q, P of the MFMA blocks
    __shared__ f32x4 t1[NBF * 4 * LO];     // W2[16nb+4g+r][F0+f], r in the float4
    __shared__ f32x4 t3[LO * NBF * 4];     // W2[F0+f][16kb+4g+i], i in the float4
    __shared__ f32x4 t2v[NT2];             // W2[F0+f'][F0+f], then b1, b2, Q, P of the left-over features
    const float* t2 = reinterpret_cast<const float*>(t2v);

    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int j = lane & 15;
    const int g = lane >> 4;

    auto tile_rows = [&](long long t, long long& t0, long long& r) {
        t0 = (t * WAVES + wave) * 16;
        r = t0 + j;
    };
    long long tile = blockIdx.x;
    long long t0A, row;
    tile_rows(tile, t0A, row);
    bool ok = row < a.n;
    if (!ok) row = a.n - 1;
    const float* sa = a.xa + row * a.ldx;
    const float* sb = a.xb + row * a.ldx;

    const f32x4* Wall = reinterpret_cast<const f32x4*>(a.packed + so.oA);
    const long long b2base4 = (long long)((so.oB - so.oA) / 4);
    for (int i = tid; i < 4 * NBF * 4; i += THREADS) {
        const int v = i / (NBF * 4), e = i % (NBF * 4);
        const size_t o = v == 0 ? a.ob1 : (v == 1 ? a.ob2 : (v == 2 ? a.oQ : a.oP));
        cvec[v][e] = reinterpret_cast<const f32x4*>(a.packed + o)[e];
    }
    for (int i = tid; i < NBF * 4 * LO; i += THREADS) t1[i] = reinterpret_cast<const f32x4*>(a.packed + so.oT1)[i];
    for (int i = tid; i < LO * NBF * 4; i += THREADS) t3[i] = reinterpret_cast<const f32x4*>(a.packed + so.oT3)[i];
    for (int i = tid; i < NT2; i += THREADS) t2v[i] = reinterpret_cast<const f32x4*>(a.packed + so.oT2)[i];
    const f32x4* b1p = cvec[0];
    const f32x4* b2p = cvec[1];
    const f32x4* Qp = cvec[2];
    const f32x4* Pp = cvec[3];
    const float* w2ll = t2;                      // [f'][f]
    const float* b1l = t2 + LO * LO;
    const float* b2l = b1l + LO;
    const float* ql = b2l + LO;
    const float* pl = ql + LO;
    const int KS1 = a.KS1;
    const int D0 = a.D0;
    const int NC1 = (KS1 + KPB - 1) / KPB;

    f32x4 st[NS];
    auto load1 = [&](long long base) {
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
    auto dot4 = [](const f32x4 w, const f32x4 v, float acc) {
        acc = fmaf(w[0], v[0], acc);
        acc = fmaf(w[1], v[1], acc);
        acc = fmaf(w[2], v[2], acc);
        return fmaf(w[3], v[3], acc);
    };

    // ---- prologue ----------------------------------------------------------------------------------------------
    load1(0);
    f32x4 xa[KPB], xb[KPB];
#pragma unroll
    for (int s = 0; s < KPB; ++s) {
        xa[s] = load_x4c<false>(sa, 16 * s + 4 * g, D0);
        xb[s] = load_x4c<false>(sb, 16 * s + 4 * g, D0);
    }
    store1(wbuf[0]);
    load2(0);
    store2(wbuf[0]);
    __syncthreads();
    int par = 0;

    for (;;) {
    const long long tile_n = tile + gridDim.x;
    long long t0A_n, row_n;
    tile_rows(tile_n, t0A_n, row_n);
    const bool ok_n = row_n < a.n;
    if (!ok_n) row_n = a.n - 1;
    const float* sa_n = a.xa + row_n * a.ldx;
    const float* sb_n = a.xb + row_n * a.ldx;

    f32x4 accA[NBF], accB[NBF];
#pragma unroll
    for (int nb = 0; nb < NBF; ++nb) {
        accA[nb] = b1p[4 * nb + g];
        accB[nb] = accA[nb];
    }
    float ulA[LO], ulB[LO];  // this lane's k-slice of the left-over layer-1 outputs
#pragma unroll
    for (int f = 0; f < LO; ++f) ulA[f] = ulB[f] = 0.f;

    // ---- layer 1 -------------------------------------------------------------------------------------------------
    for (int c = 0; c < NC1; ++c) {
        const int cur = par;
        const bool more = (c + 1 < NC1);
        const long long nbase = more ? (long long)(c + 1) * CH : b2base4;
        load1(nbase);
        const f32x4* w = wbuf[cur];
#pragma unroll
        for (int s = 0; s < KPB; ++s) {
            if (KPB * c + s < KS1) {
                // the LO left-over weight rows of this step (broadcast reads) are issued FIRST and consumed LAST: the
                // fence keeps the reads above the MFMA blocks, whose 2560 cycles hide the LDS latency, and the 8 LO
                // FMAs of the step then run on data that has long arrived
                f32x4 wl[LO];
#pragma unroll
                for (int f = 0; f < LO; ++f) wl[f] = w[s * SS4 + NBF * 64 + f * 4 + g];
                LDS_FENCE();
#pragma unroll
                for (int nb0 = 0; nb0 < NBF; nb0 += G) {
                    f32x4 av[G];
#pragma unroll
                    for (int u = 0; u < G; ++u)
                        if (nb0 + u < NBF) av[u] = w[s * SS4 + (nb0 + u) * 64 + lane];
#pragma unroll
                    for (int r = 0; r < 4; ++r) {
#pragma unroll
                        for (int u = 0; u < G; ++u) {
                            if (nb0 + u < NBF) {
                                accA[nb0 + u] = __builtin_amdgcn_mfma_f32_16x16x4f32(av[u][r], xa[s][r], accA[nb0 + u], 0, 0, 0);
                                accB[nb0 + u] = __builtin_amdgcn_mfma_f32_16x16x4f32(av[u][r], xb[s][r], accB[nb0 + u], 0, 0, 0);
                            }
                        }
                    }
                }
#pragma unroll
                for (int f = 0; f < LO; ++f) {
                    ulA[f] = dot4(wl[f], xa[s], ulA[f]);
                    ulB[f] = dot4(wl[f], xb[s], ulB[f]);
                }
            }
            {
                const int ks = more ? KPB * (c + 1) + s : s;
                const float* ra = more ? sa : sa_n;
                const float* rb = more ? sb : sb_n;
                xa[s] = load_x4c<false>(ra, 16 * ks + 4 * g, D0);
                xb[s] = load_x4c<false>(rb, 16 * ks + 4 * g, D0);
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

    // ---- F.normalize (utils/models.py:368) over the MFMA blocks and the left-over features ------------------------
    {
        float ssA = 0.f, ssB = 0.f;
#pragma unroll
        for (int nb = 0; nb < NBF; ++nb) {
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                ssA = fmaf(accA[nb][r], accA[nb][r], ssA);
                ssB = fmaf(accB[nb][r], accB[nb][r], ssB);
            }
        }
        ssA = wave_xor_add(ssA, 16); ssA = wave_xor_add(ssA, 32);
        ssB = wave_xor_add(ssB, 16); ssB = wave_xor_add(ssB, 32);
#pragma unroll
        for (int f = 0; f < LO; ++f) {  // sum the four k-slices, add the bias: now replicated over g
            ulA[f] = wave_xor_add(ulA[f], 16); ulA[f] = wave_xor_add(ulA[f], 32) + b1l[f];
            ulB[f] = wave_xor_add(ulB[f], 16); ulB[f] = wave_xor_add(ulB[f], 32) + b1l[f];
            ssA = fmaf(ulA[f], ulA[f], ssA);
            ssB = fmaf(ulB[f], ulB[f], ssB);
        }
        const float invA = 1.0f / fmaxf(sqrtf(ssA), 1e-12f);
        const float invB = 1.0f / fmaxf(sqrtf(ssB), 1e-12f);
#pragma unroll
        for (int nb = 0; nb < NBF; ++nb) {
            accA[nb] *= invA;
            accB[nb] *= invB;
        }
#pragma unroll
        for (int f = 0; f < LO; ++f) {
            ulA[f] *= invA;
            ulB[f] *= invB;
        }
    }

    // ---- layer 2 -------------------------------------------------------------------------------------------------
    // (a) the LO left-over OUTPUT features, entirely from registers and LDS-resident tables, folded straight into the
    //     score so that nothing of them stays live across the MFMA loop:
    //     z_f' = b2 + sum_f W2[F0+f'][F0+f] y_f + sum_{k < 16 NBF} W2[F0+f'][k] y[k]   (this lane's k-slices, summed over g)
    float part_left = 0.f;
#pragma unroll
    for (int fo = 0; fo < LO; ++fo) {
        float sA = b2l[fo], sB = b2l[fo];
#pragma unroll
        for (int f = 0; f < LO; ++f) {
            sA = fmaf(w2ll[fo * LO + f], ulA[f], sA);
            sB = fmaf(w2ll[fo * LO + f], ulB[f], sB);
        }
        sA = g == 0 ? sA : 0.f;
        sB = g == 0 ? sB : 0.f;
#pragma unroll
        for (int kb = 0; kb < NBF; ++kb) {
            if (kb % 3 == 0) LDS_FENCE();
            const f32x4 wv = t3[(fo * NBF + kb) * 4 + g];
            sA = dot4(wv, accA[kb], sA);
            sB = dot4(wv, accB[kb], sB);
        }
        float z1 = wave_xor_add(sA, 16); z1 = wave_xor_add(z1, 32);
        float z2 = wave_xor_add(sB, 16); z2 = wave_xor_add(z2, 32);
        float t = ql[fo] * fmaf(z1, z1, z2 * z2);
        t = fmaf(2.0f * pl[fo], z1 * z2, t);
        part_left += g == 0 ? t : 0.f;
    }
    LDS_FENCE();
    // (b) the MFMA-block outputs start from b2 plus the contribution of the left-over INPUT features (T1); y_f dies here
    f32x4 zA[NBF], zB[NBF];
#pragma unroll
    for (int nb = 0; nb < NBF; ++nb) {
        zA[nb] = b2p[4 * nb + g];
        zB[nb] = zA[nb];
#pragma unroll
        for (int f = 0; f < LO; ++f) {
            if (f % 2 == 0) LDS_FENCE();
            const f32x4 wv = t1[(nb * 4 + g) * LO + f];
            zA[nb] += wv * ulA[f];
            zB[nb] += wv * ulB[f];
        }
    }
    LDS_FENCE();
#pragma unroll
    for (int c2 = 0; c2 < NC2; ++c2) {
        const int cur = par;
        const bool more2 = (c2 + 1 < NC2);
        const long long nbase = more2 ? b2base4 + (long long)(c2 + 1) * CH : 0;  // 0: chunk 0 of the next tile
        load1(nbase);
        const f32x4* w = wbuf[cur];
#pragma unroll
        for (int s = 0; s < KPB; ++s) {
            const int kb = KPB * c2 + s;
            if (kb < NBF) {
#pragma unroll
                for (int nb0 = 0; nb0 < NBF; nb0 += G) {
                    f32x4 av[G];
#pragma unroll
                    for (int u = 0; u < G; ++u)
                        if (nb0 + u < NBF) av[u] = w[s * SS4 + (nb0 + u) * 64 + lane];
#pragma unroll
                    for (int r = 0; r < 4; ++r) {
#pragma unroll
                        for (int u = 0; u < G; ++u) {
                            if (nb0 + u < NBF) {
                                zA[nb0 + u] = __builtin_amdgcn_mfma_f32_16x16x4f32(av[u][r], accA[kb < NBF ? kb : 0][r], zA[nb0 + u], 0, 0, 0);
                                zB[nb0 + u] = __builtin_amdgcn_mfma_f32_16x16x4f32(av[u][r], accB[kb < NBF ? kb : 0][r], zB[nb0 + u], 0, 0, 0);
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

    // ---- epilogue: s = sum Q (z1^2 + z2^2) + 2 sum P z1 z2 (utils/models.py:372-376) --------------------------------
    {
        float part = 0.f;
#pragma unroll
        for (int nb = 0; nb < NBF; ++nb) {
            const f32x4 q = Qp[4 * nb + g];
            const f32x4 p = Pp[4 * nb + g];
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const float z1 = zA[nb][r], z2 = zB[nb][r];
                part = fmaf(q[r], fmaf(z1, z1, z2 * z2), part);
                part = fmaf(2.0f * p[r], z1 * z2, part);
            }
        }
        part += part_left;
        part = wave_xor_add(part, 16);
        part = wave_xor_add(part, 32);
        if (g == 0 && ok) a.out_s[t0A + j] = part;
    }

    tile = tile_n;
    if (tile >= ntiles) break;
    t0A = t0A_n; row = row_n; ok = ok_n;
    sa = sa_n; sb = sb_n;
    }  // tile loop
}

#undef LDS_FENCE

}  // namespace nplda

// nplda_fwd_bf16x3.h — fused forward on the bf16 matrix pipe with fp32-class accuracy (gfx950).
//
// fp32-input MFMA runs at 1/16 of the bf16 MFMA rate on CDNA4 and there is no TF32/xf32 mode.  This variant
// splits every fp32 operand into three bf16 pieces  v = h + m + l  (round-to-nearest at each step; the residuals
// v - h and (v - h) - m are exact in fp32, so the three pieces carry all 24 mantissa bits) and evaluates
//      w x  ~=  wh xh + (wh xm + wm xh) + (wh xl + wl xh + wm xm)
// i.e. SIX bf16 MFMA passes (products are exact, accumulation is fp32); the dropped terms (wm xl, wl xm, wl xl)
// are <= 2^-23 relative — the size of one fp32 rounding.  Six passes at 16x the rate = 2.7x the fp32-MFMA compute
// ceiling.  Scores agree with the fp32 kernel / the fp64 oracle to a few 1e-6 (tests/test_bf16x3_gpu.py); the
// exact-fp32 kernels remain the default and the only ones used for training.
//
// Structure = nplda_fwd_v2.h (weights L2 -> LDS in fragment order, KPB k32-steps per barrier staged in two
// halves; x in a ring reloaded a chunk ahead; layer-1 accumulators feed layer 2 from registers), with
// v_mfma_f32_16x16x32_bf16: A = weights (16 features x 32 k), B = data (32 k x 16 rows), lane (j = lane&15,
// g = lane>>4) holds 8 consecutive k of its row.  For layer 2 the 8 k-values of a lane are the accumulator
// registers of TWO feature blocks: k(e) = 32 c + 4 g + e (e < 4), 32 c + 16 + 4 g + (e - 4) (e >= 4); the W2 image
// is packed with the same permutation.  The weights are split offline (nplda_pack_bf16x3_kernel); x and y are split
// on the fly (12 v_cvt_pk_bf16_f32 + 10 subtracts per 8 values).
#pragma once
#include "nplda_fwd_kernel.h"

namespace nplda {

typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));

// image (units: floats; every bf16 fragment = 64 lanes x 16 B = 256 floats-worth):
//   W1b[c][nb][part][lane]  c < KC1 = ceil(D0/32)        W2b[c2][nb][part][lane]  c2 < KC2 = ceil(NB/2)
//   then fp32 b1, b2, Q, P (NB*16 each), then one chunk of slack.
struct Bf3Layout {
    int D0, D1, D2, NB, KC1, KC2;
    size_t oW1, oW2, ob1, ob2, oQ, oP, total;
};

__host__ __device__ inline Bf3Layout bf3_layout(int D0, int D1, int D2) {
    Bf3Layout L;
    L.D0 = D0; L.D1 = D1; L.D2 = D2;
    L.NB = nplda_kernel_nb(D1, D2);
    L.KC1 = (D0 + 31) / 32;
    L.KC2 = (L.NB + 1) / 2;
    L.oW1 = 0;
    L.oW2 = (size_t)L.KC1 * L.NB * 3 * 256;
    L.ob1 = L.oW2 + (size_t)L.KC2 * L.NB * 3 * 256;
    L.ob2 = L.ob1 + (size_t)L.NB * 16;
    L.oQ = L.ob2 + (size_t)L.NB * 16;
    L.oP = L.oQ + (size_t)L.NB * 16;
    L.total = L.oP + (size_t)L.NB * 16 + (size_t)2 * L.NB * 3 * 256;  // + slack for unconditional chunk loads
    return L;
}

__device__ __forceinline__ void split3(float v, __bf16& h, __bf16& m, __bf16& l) {
    h = (__bf16)v;
    const float r1 = v - (float)h;
    m = (__bf16)r1;
    const float r2 = r1 - (float)m;
    l = (__bf16)r2;
}

__device__ __forceinline__ void split3x8(const f32x4 lo, const f32x4 hi, bf16x8& H, bf16x8& M, bf16x8& Lq) {
#pragma unroll
    for (int e = 0; e < 8; ++e) {
        __bf16 h, m, l;
        split3(e < 4 ? lo[e] : hi[e - 4], h, m, l);
        H[e] = h; M[e] = m; Lq[e] = l;
    }
}

// one thread per 16-byte fragment element (8 bf16) or per fp32 tail element
static __global__ void nplda_pack_bf16x3_kernel(const float* __restrict__ W1, const float* __restrict__ b1,
                                                const float* __restrict__ W2, const float* __restrict__ b2,
                                                const float* __restrict__ P_sqrt, const float* __restrict__ Q,
                                                Bf3Layout L, float* __restrict__ out) {
    const size_t nfrag = L.ob1 / 4;                 // 16-byte units in the two weight regions
    const size_t ntail = L.total - L.ob1;
    const size_t idx = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (idx < nfrag) {
        const bool second = idx >= L.oW2 / 4;
        size_t rel = second ? idx - L.oW2 / 4 : idx;
        const int lane = (int)(rel & 63);
        rel >>= 6;
        const int part = (int)(rel % 3);
        rel /= 3;
        const int nb = (int)(rel % L.NB);
        const int c = (int)(rel / L.NB);
        const int f = 16 * nb + (lane & 15), g = lane >> 4;
        bf16x8 o;
#pragma unroll
        for (int e = 0; e < 8; ++e) {
            float v = 0.f;
            if (!second) {
                const int k = 32 * c + 8 * g + e;
                if (f < L.D1 && k < L.D0) v = W1[(size_t)f * L.D0 + k];
            } else {
                const int k = e < 4 ? 32 * c + 4 * g + e : 32 * c + 16 + 4 * g + (e - 4);
                if (f < L.D2 && k < L.D1) v = W2[(size_t)f * L.D1 + k];
            }
            __bf16 h, m, l;
            split3(v, h, m, l);
            o[e] = part == 0 ? h : (part == 1 ? m : l);
        }
        reinterpret_cast<bf16x8*>(out)[idx] = o;
    } else if (idx < nfrag + ntail) {
        const size_t t = idx - nfrag;  // float index past ob1
        const size_t nb16 = (size_t)L.NB * 16;
        float v = 0.f;
        const int f = (int)(t % nb16);
        const int which = (int)(t / nb16);
        if (which == 0) { if (f < L.D1) v = b1[f]; }
        else if (which == 1) { if (f < L.D2) v = b2[f]; }
        else if (which == 2) { if (f < L.D2) v = Q[f]; }
        else if (which == 3) { if (f < L.D2) v = P_sqrt[f] * P_sqrt[f]; }
        out[L.ob1 + t] = v;
    }
}

struct Bf3Args {
    const float* xa;
    const float* xb;
    long long n, ldx;
    const float* img;
    int D0, KC1;
    size_t oW2, ob1, ob2, oQ, oP;
    float* out_s;
    float* out_z;
    long long ldz;
    float* out_q;
};

#define NPLDA_MFMA_BF16(A, B, C) __builtin_amdgcn_mfma_f32_16x16x32_bf16(A, B, C, 0, 0, 0)

// Six-pass product accumulate (small terms first) for TWO feature blocks x TWO row groups at once: the four
// accumulator chains are interleaved so that consecutive MFMAs never depend on each other (a 16x16x32 bf16 MFMA
// issues every 16 cycles but its result is only available ~2.5 issue slots later).
struct WFrag { bf16x8 h, m, l; };
__device__ __forceinline__ void mfma6x4(const WFrag& w0, const WFrag& w1, const bf16x8 ah, const bf16x8 am,
                                        const bf16x8 al, const bf16x8 bh, const bf16x8 bm, const bf16x8 bl,
                                        f32x4& a0, f32x4& b0, f32x4& a1, f32x4& b1) {
#define NPLDA_STEP4(W0, W1, XA, XB)        \
    a0 = NPLDA_MFMA_BF16(W0, XA, a0);      \
    b0 = NPLDA_MFMA_BF16(W0, XB, b0);      \
    a1 = NPLDA_MFMA_BF16(W1, XA, a1);      \
    b1 = NPLDA_MFMA_BF16(W1, XB, b1)
    NPLDA_STEP4(w0.m, w1.m, am, bm);
    NPLDA_STEP4(w0.h, w1.h, al, bl);
    NPLDA_STEP4(w0.l, w1.l, ah, bh);
    NPLDA_STEP4(w0.h, w1.h, am, bm);
    NPLDA_STEP4(w0.m, w1.m, ah, bh);
    NPLDA_STEP4(w0.h, w1.h, ah, bh);
#undef NPLDA_STEP4
}
__device__ __forceinline__ void mfma6x2(const WFrag& w0, const bf16x8 ah, const bf16x8 am, const bf16x8 al,
                                        const bf16x8 bh, const bf16x8 bm, const bf16x8 bl, f32x4& a0, f32x4& b0) {
    a0 = NPLDA_MFMA_BF16(w0.m, am, a0); b0 = NPLDA_MFMA_BF16(w0.m, bm, b0);
    a0 = NPLDA_MFMA_BF16(w0.h, al, a0); b0 = NPLDA_MFMA_BF16(w0.h, bl, b0);
    a0 = NPLDA_MFMA_BF16(w0.l, ah, a0); b0 = NPLDA_MFMA_BF16(w0.l, bh, b0);
    a0 = NPLDA_MFMA_BF16(w0.h, am, a0); b0 = NPLDA_MFMA_BF16(w0.h, bm, b0);
    a0 = NPLDA_MFMA_BF16(w0.m, ah, a0); b0 = NPLDA_MFMA_BF16(w0.m, bh, b0);
    a0 = NPLDA_MFMA_BF16(w0.h, ah, a0); b0 = NPLDA_MFMA_BF16(w0.h, bh, b0);
}

template <int NB, int MODE, int WAVES, int KPB, bool EARLY = true>
__global__ __launch_bounds__(WAVES * 64, 2) void nplda_fwd_bf16x3_kernel(const Bf3Args a) {
    static_assert(MODE == MODE_PAIR || MODE == MODE_EMBED, "bf16x3 kernel modes");
    constexpr int THREADS = WAVES * 64;
    constexpr int STEP4 = NB * 3 * 64;   // 16-byte units per k32-step of weights
    constexpr int CH = STEP4 * KPB;
    constexpr int HALF = ((CH / 2 + THREADS - 1) / THREADS) * THREADS;
    constexpr int NS1 = HALF / THREADS;
    constexpr int NS2 = (CH - HALF + THREADS - 1) / THREADS;
    constexpr int NS = NS1 > NS2 ? NS1 : NS2;
    constexpr int KC2 = (NB + 1) / 2;
    constexpr int NC2 = (KC2 + KPB - 1) / KPB;
    static_assert(HALF <= CH, "chunk must split into two staging halves");
    constexpr int SMID = KPB >= 2 ? KPB / 2 : 1;
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
    const float* sa = a.xa + rowA * a.ldx;
    const float* sb = a.xb + rowB * a.ldx;
    const float* pa = sa + 8 * g;
    const float* pb = sb + 8 * g;

    const f32x4* Wall = reinterpret_cast<const f32x4*>(a.img);
    const f32x4* b1p = reinterpret_cast<const f32x4*>(a.img + a.ob1);
    const f32x4* b2p = reinterpret_cast<const f32x4*>(a.img + a.ob2);
    const f32x4* Qp = reinterpret_cast<const f32x4*>(a.img + a.oQ);
    const f32x4* Pp = reinterpret_cast<const f32x4*>(a.img + a.oP);
    const int KC1 = a.KC1;
    const int D0 = a.D0;
    const int NC1 = (KC1 + KPB - 1) / KPB;
    const long long w2base4 = (long long)(a.oW2 / 4);

    // EARLY: the second staging half has its own registers and is fetched at chunk start as well (a whole chunk
    // of latency cover instead of half a chunk: bf16x3 chunks last only ~3.6 us; +1.5 % measured)
    f32x4 st[NS], st2[EARLY ? (NS2 > 0 ? NS2 : 1) : 1];
    auto load1 = [&](long long base) {
#pragma unroll
        for (int i = 0; i < NS1; ++i) st[i] = Wall[base + tid + THREADS * i];
        if (EARLY) {
#pragma unroll
            for (int i = 0; i < NS2; ++i) {
                const int idx = HALF + tid + THREADS * i;
                st2[i] = Wall[base + (idx < CH ? idx : CH - 1)];
            }
        }
    };
    auto store1 = [&](f32x4* dst) {
#pragma unroll
        for (int i = 0; i < NS1; ++i) dst[tid + THREADS * i] = st[i];
    };
    auto load2 = [&](long long base) {
        if (EARLY) return;
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
            if (idx < CH) dst[idx] = EARLY ? st2[i] : st[i];
        }
    };
    // x of one k32-step: 8 consecutive floats per lane per side (two float4), D0 % 4 == 0
    auto loadx = [&](const float* p, const float* psafe, int kc, bool live, f32x4& lo, f32x4& hi) {
        (void)p; (void)live;  // columns past D0 (or past the last chunk) read the row's last float4: see load_x4c
        const int k0 = 32 * kc + 8 * g;
        lo = load_x4c<false>(psafe, k0, D0);
        hi = load_x4c<false>(psafe, k0 + 4, D0);
    };

    // ---- prologue --------------------------------------------------------------------------------------------
    load1(0);
    f32x4 xaL[KPB], xaH[KPB], xbL[KPB], xbH[KPB];
#pragma unroll
    for (int s = 0; s < KPB; ++s) {
        loadx(pa, sa, s, true, xaL[s], xaH[s]);
        loadx(pb, sb, s, true, xbL[s], xbH[s]);
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

    // ---- layer 1 -------------------------------------------------------------------------------------------------
    for (int c = 0; c < NC1; ++c) {
        const int cur = c & 1;
        const bool more = (c + 1 < NC1);
        const long long nbase = more ? (long long)(c + 1) * CH : w2base4;
        load1(nbase);
        const bf16x8* w = reinterpret_cast<const bf16x8*>(wbuf[cur]);
#pragma unroll
        for (int s = 0; s < KPB; ++s) {
            if (KPB * c + s < KC1) {
                bf16x8 ah, am, al, bh, bm, bl;
                split3x8(xaL[s], xaH[s], ah, am, al);
                split3x8(xbL[s], xbH[s], bh, bm, bl);
#pragma unroll
                for (int nb = 0; nb < NB; nb += 2) {
                    WFrag w0, w1;
                    w0.h = w[s * STEP4 + (nb * 3 + 0) * 64 + lane];
                    w0.m = w[s * STEP4 + (nb * 3 + 1) * 64 + lane];
                    w0.l = w[s * STEP4 + (nb * 3 + 2) * 64 + lane];
                    if (nb + 1 < NB) {
                        w1.h = w[s * STEP4 + (nb * 3 + 3) * 64 + lane];
                        w1.m = w[s * STEP4 + (nb * 3 + 4) * 64 + lane];
                        w1.l = w[s * STEP4 + (nb * 3 + 5) * 64 + lane];
                        mfma6x4(w0, w1, ah, am, al, bh, bm, bl, accA[nb], accB[nb], accA[nb + 1 < NB ? nb + 1 : 0],
                                accB[nb + 1 < NB ? nb + 1 : 0]);
                    } else {
                        mfma6x2(w0, ah, am, al, bh, bm, bl, accA[nb], accB[nb]);
                    }
                }
            }
            {
                const int kc = KPB * (c + 1) + s;
                loadx(pa, sa, kc, more && kc < KC1, xaL[s], xaH[s]);
                loadx(pb, sb, kc, more && kc < KC1, xbL[s], xbH[s]);
            }
            if (s == SMID - 1) {
                store1(wbuf[cur ^ 1]);
                load2(nbase);
            }
        }
        store2(wbuf[cur ^ 1]);
        __syncthreads();
    }

    // ---- F.normalize ------------------------------------------------------------------------------------------------
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

    // ---- layer 2: k32-step c2 consumes the accumulator blocks 2 c2 and 2 c2 + 1 ---------------------------------
    f32x4 zA[NB], zB[NB];
#pragma unroll
    for (int nb = 0; nb < NB; ++nb) {
        zA[nb] = b2p[4 * nb + g];
        zB[nb] = zA[nb];
    }
    const f32x4 zero4 = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int q2 = 0; q2 < NC2; ++q2) {
        const int cur = (NC1 + q2) & 1;
        const bool more2 = (q2 + 1 < NC2);
        const long long nbase = w2base4 + (long long)(q2 + 1) * CH;
        if (more2) load1(nbase);
        const bf16x8* w = reinterpret_cast<const bf16x8*>(wbuf[cur]);
#pragma unroll
        for (int s = 0; s < KPB; ++s) {
            const int c2 = KPB * q2 + s;
            if (c2 < KC2) {
                bf16x8 ah, am, al, bh, bm, bl;
                split3x8(accA[2 * c2], (2 * c2 + 1 < NB) ? accA[2 * c2 + 1 < NB ? 2 * c2 + 1 : 0] : zero4, ah, am, al);
                split3x8(accB[2 * c2], (2 * c2 + 1 < NB) ? accB[2 * c2 + 1 < NB ? 2 * c2 + 1 : 0] : zero4, bh, bm, bl);
#pragma unroll
                for (int nb = 0; nb < NB; nb += 2) {
                    WFrag w0, w1;
                    w0.h = w[s * STEP4 + (nb * 3 + 0) * 64 + lane];
                    w0.m = w[s * STEP4 + (nb * 3 + 1) * 64 + lane];
                    w0.l = w[s * STEP4 + (nb * 3 + 2) * 64 + lane];
                    if (nb + 1 < NB) {
                        w1.h = w[s * STEP4 + (nb * 3 + 3) * 64 + lane];
                        w1.m = w[s * STEP4 + (nb * 3 + 4) * 64 + lane];
                        w1.l = w[s * STEP4 + (nb * 3 + 5) * 64 + lane];
                        mfma6x4(w0, w1, ah, am, al, bh, bm, bl, zA[nb], zB[nb], zA[nb + 1 < NB ? nb + 1 : 0],
                                zB[nb + 1 < NB ? nb + 1 : 0]);
                    } else {
                        mfma6x2(w0, ah, am, al, bh, bm, bl, zA[nb], zB[nb]);
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

    // ---- epilogue (fp32, identical to the fp32 kernels) ----------------------------------------------------------------
    if (MODE == MODE_PAIR) {
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

}  // namespace nplda

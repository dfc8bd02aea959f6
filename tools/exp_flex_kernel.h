// exp_flex_kernel.h (tools/exp_flex.hip; an experiment, NOT part of the library) — pair scoring between one and a few dozen 8-pair row groups per CU, balanced to ONE row group
// (nplda_fwd_mid.h balances to a 16-pair tile: 10 240 pairs — the validation chunk of xvector_NeuralPlda_pytorch.py:172 —
// are 2.5 tiles per CU, half the CUs run three; here they are 5 row groups on every CU).
//
// Unit of work: a ROW GROUP of 8 trial pairs — one 16-row MFMA B operand whose rows 0..7 are the pairs' x1 rows and rows
// 8..15 their x2 rows (the half tile of nplda_train_fb_half.h): a pair's two embeddings sit in lanes j and j + 8 of the same
// registers and meet by one DPP row rotation.  A block (8 waves, one per CU) takes a contiguous range of row groups,
// balanced to one over the grid, and works through it in SUPER TILES of RG <= 8 row groups:
//  * both layers are (NB feature blocks) x (RG row groups) = NB RG output units of 16 x 16; the units are dealt to the 8
//    waves in block-major order, NB RG / 8 each (+- 1, the larger shares on different SIMDs) — every SIMD carries the same
//    number of MFMAs whatever RG is, which neither a feature split (10 blocks over 4 SIMDs) nor a row split gives;
//  * BOTH operands come from LDS: the weight chunks (the image's fragment order, as in nplda_fwd_v5.h) and the x rows
//    (wave w streams row group w: one LDS-DMA instruction drops a k16-step of its 16 rows into LDS as a finished B
//    fragment), so a wave's reads per k16-step are its <= 3 weight fragments and the <= RG row fragments of its units for
//    4 NB RG / 8 MFMAs, and nothing is fetched twice from L2 / HBM;
//  * x is fetched three chunks ahead (HBM latency), weights one chunk ahead (L2); the chunk fence waits for everything but
//    the wave's newest x segments (vmcnt(KPB));
//  * row norms and scores are summed across the waves through small LDS tables in a fixed order; the normalised layer-1
//    output goes to LDS once, in accumulator layout = the B-fragment layout of layer 2 (the chained-GEMM layout of the
//    other forward kernels), overlaying the x ring.
// Same arithmetic per element as the other forward kernels; the association of the cross-feature sums differs (per wave,
// then over waves).  512-d x-vectors (KS1 = 32), NB = 10 / 11, plain (not indexed) pairs.
#pragma once
#include "../neuralplda_amd/csrc/nplda_fwd_kernel.h"

namespace nplda {

constexpr int kFlexMaxRG = 8;
#ifndef NPLDA_FLEX_ABL
#define NPLDA_FLEX_ABL 0  // tools/exp_flex.hip only (results are WRONG when non-zero): 1 no barriers in the chunk loops, 2 no fragment reads after the first, 4 no DMA after the prologue
#endif
// Row groups per super tile and ring depths: three weight chunks (the one in use, the next — complete — and the one in
// flight) and five x chunks (in use, three complete or landing, one in flight).  LDS per block: weights 3 KPB NB KB, the x
// ring 5 KPB MAXRG KB overlaid by the y tiles MAXRG NB KB, 9 KB of tables: 149 KB at NB = 10; NB = 11 fits with 7 row groups.
template <int NB>
struct FlexCfg {
    static constexpr int KPB = 2, NWB = 3, NXS = 5;
    static constexpr int MAXRG = NB <= 10 ? 8 : 7;
    static constexpr int WST = NWB * KPB * NB * 64;                                                     // float4
    static constexpr int BIG = (NXS * KPB * MAXRG > MAXRG * NB ? NXS * KPB * MAXRG : MAXRG * NB) * 64;  // float4
};

// Fragment reads by hand.  hipcc's waitcnt pass cannot tell which LDS bytes an in-flight LDS-DMA (global_load_lds) will
// write, so it puts s_waitcnt vmcnt(0) in front of every ds_read it sees after one — i.e. each chunk would wait out the
// L2 / HBM latency of the prefetches it has just issued.  Reads it does not see are not waited for: the kernel fences
// the DMA itself (vmcnt + barrier at the chunk end) and waits for its reads with lds_wait() — an s_waitcnt lgkmcnt(0)
// that the consumers depend on through the "+v" ties.
// LDS byte address of a __shared__ array (taken of the array itself in the kernel: a constant).  Everything the kernel
// hands to the DMA and to the hand-written reads is such an address plus integer arithmetic, cast back with an
// integer -> LDS pointer cast: a generic -> LDS pointer cast of a computed pointer carries a null check that hipcc fails to
// select here ("V_CMP_NE_U32 0, src_shared_base: operand has incorrect register class").
__device__ __forceinline__ unsigned lds_addr(const void* p) {
    return (unsigned)(uintptr_t)(const __attribute__((address_space(3))) void*)p;
}
__device__ __forceinline__ __attribute__((address_space(3))) void* lds_ptr(unsigned a) {
    return (__attribute__((address_space(3))) void*)(uintptr_t)a;
}
__device__ __forceinline__ f32x4 lds_read16(unsigned addr) {
    f32x4 v;
    asm volatile("ds_read_b128 %0, %1" : "=v"(v) : "v"(addr));
    return v;
}
template <int N>
__device__ __forceinline__ void lds_wait(f32x4 (&f)[N]) {
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
#pragma unroll
    for (int i = 0; i < N; ++i) asm volatile("" : "+v"(f[i]));
}

template <int NB, int RG, int W>
struct FlexOwn {
    static constexpr int NUA = NB * RG;
    static constexpr int RANK = (W & 3) * 2 + (W >> 2);  // waves w and w + 4 share a SIMD: neighbouring shares
    static constexpr int U0 = RANK * NUA / 8, U1 = (RANK + 1) * NUA / 8, NU = U1 - U0;
    static constexpr int CB0 = U0 / RG, CB1 = (U1 - 1) / RG, NCB = CB1 - CB0 + 1;
    static constexpr int cb(int u) { return (U0 + u) / RG; }
    static constexpr int rg(int u) { return (U0 + u) % RG; }
    static constexpr bool needs_rg(int r) {
        for (int u = 0; u < NU; ++u)
            if (rg(u) == r) return true;
        return false;
    }
};

// One super tile of RG row groups starting at row group rg0, wave W's part.  LDS: wst[2][KPB][NB][64] (weight chunks),
// big = xst[4][KPB][8][64] in layer 1 / ylds[RG][NB][64] in layer 2, ssb / scb (partial row norms / scores), cv (b1 b2 Q P).
template <int NB, int RG, int W, int KPB = 2>
__device__ __forceinline__ void flex_tile(const FwdArgs& a, long long rg0, f32x4* wst, f32x4* big, float (*ssb)[kFlexMaxRG][16],
                                          float (*scb)[kFlexMaxRG][8], const f32x4* cv, int lane, unsigned wst_a, unsigned big_a) {
    using O = FlexOwn<NB, RG, W>;
    constexpr int NU = O::NU, NCB = O::NCB, CB0 = O::CB0;
    constexpr int MAXRG = FlexCfg<NB>::MAXRG;
    constexpr int NWB = FlexCfg<NB>::NWB, NXS = FlexCfg<NB>::NXS;
    constexpr int WCH = KPB * NB * 64;     // float4 per weight chunk
    constexpr int XCH = KPB * MAXRG * 64;  // float4 per x chunk slot
    constexpr int NC1 = 32 / KPB;
    constexpr int NC2 = (NB + KPB - 1) / KPB;
    constexpr bool XW = W < RG;                 // this wave streams the x rows of row group W
    static_assert(32 % KPB == 0 && KPB % 2 == 0 && KPB == FlexCfg<NB>::KPB && NU >= 1 && RG <= MAXRG, "KS1 = 32");
    const int j = lane & 15, g = lane >> 4;
    const f32x4* Wall = reinterpret_cast<const f32x4*>(a.packed);
    const long long w2base4 = (long long)(a.oW2 / 4);
    const f32x4* b1p = cv;
    const f32x4* b2p = cv + NB * 4;
    const f32x4* Qp = cv + 2 * NB * 4;
    const f32x4* Pp = cv + 3 * NB * 4;

    const float* xrow = a.xa;
    if (XW) {
        long long p = (rg0 + W) * 8 + (j & 7);
        if (p >= a.n) p = a.n - 1;
        xrow = ((j >> 3) ? a.xb : a.xa) + p * a.ldx + 4 * g;
    }
    auto dma_w = [&](long long base4, unsigned dst) {  // dst: LDS byte address of the chunk buffer
#pragma unroll
        for (int sgm = W; sgm < KPB * NB; sgm += 8) {
            unsigned lo = (unsigned)lane * 16u;
            asm volatile("" : "+v"(lo));
            __builtin_amdgcn_global_load_lds(
                (const __attribute__((address_space(1))) void*)(reinterpret_cast<const char*>(Wall + base4 + sgm * 64) + lo),
                lds_ptr(dst + (unsigned)sgm * 1024u), 16, 0, 0);
        }
    };
    auto dma_x = [&](int c, unsigned dst) {
        if (XW) {
#pragma unroll
            for (int s = 0; s < KPB; ++s)
                __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(xrow + 16 * (KPB * c + s)),
                                                 lds_ptr(dst + (unsigned)(s * MAXRG + W) * 1024u), 16, 0, 0);
        }
    };
    auto fence_all = [&]() {
        __builtin_amdgcn_s_waitcnt(0x0070);  // vmcnt(0) lgkmcnt(0)
        __builtin_amdgcn_s_barrier();
        __builtin_amdgcn_sched_barrier(0);
    };
    // fr[s & 1]: the fragments of step s — NCB weight fragments, then the row fragments.  The fragments of the NEXT step are
    // always read under the MFMAs of this one, across chunks too: a chunk is complete in LDS (fenced) a whole iteration
    // before its first step is read, so no read ever waits behind a barrier — with the reads right after it all eight waves
    // asked for 11 KB each at once and the last one to be served had waited 700 cycles, every chunk.
    f32x4 fr[2][NCB + RG];
    bool first_read = true;
    auto read_l1 = [&](unsigned wb, unsigned xb, int s, f32x4 (&f)[NCB + RG]) {
        if ((NPLDA_FLEX_ABL & 2) && !first_read) return;
#pragma unroll
        for (int i = 0; i < NCB; ++i) f[i] = lds_read16(wb + (unsigned)((s * NB + CB0 + i) * 1024));
#pragma unroll
        for (int r = 0; r < RG; ++r)
            if (O::needs_rg(r)) f[NCB + r] = lds_read16(xb + (unsigned)((s * MAXRG + r) * 1024));
    };
    const unsigned lane16 = (unsigned)lane * 16u;

    // ---- prologue: weight chunks 0, 1, x chunks 0 .. 3 -------------------------------------------------------------------
    dma_w(0, wst_a);
    dma_w(WCH, wst_a + WCH * 16u);
    dma_x(0, big_a);
    dma_x(1, big_a + XCH * 16u);
    dma_x(2, big_a + 2 * XCH * 16u);
    dma_x(3, big_a + 3 * XCH * 16u);
    f32x4 acc[NU];
#pragma unroll
    for (int u = 0; u < NU; ++u) acc[u] = b1p[4 * O::cb(u) + g];
    fence_all();
    read_l1(wst_a + lane16, big_a + lane16, 0, fr[0]);
    if (NPLDA_FLEX_ABL & 2) { read_l1(wst_a + lane16, big_a + lane16, 1, fr[1]); first_read = false; }

    // ---- layer 1 ---------------------------------------------------------------------------------------------------------
    int ws = 0, xs = 0;  // ring slots of chunk c: weights c % 3, x rows c % 5
    for (int c = 0; c < ((NPLDA_FLEX_ABL & 8) ? 4 : NC1); ++c) {
        const int ws1 = ws + 1 >= NWB ? ws + 1 - NWB : ws + 1, ws2 = ws + 2 >= NWB ? ws + 2 - NWB : ws + 2;
        const int xs1 = xs + 1 >= NXS ? xs + 1 - NXS : xs + 1, xs4 = xs + 4 >= NXS ? xs + 4 - NXS : xs + 4;
        if (!(NPLDA_FLEX_ABL & 4)) dma_w(c + 2 < NC1 ? (long long)(c + 2) * WCH : w2base4 + (long long)(c + 2 - NC1) * WCH, wst_a + (unsigned)(ws2 * WCH) * 16u);
        __builtin_amdgcn_sched_barrier(0);
        const bool xmore = c + 4 < NC1;
        if (xmore && !(NPLDA_FLEX_ABL & 4)) dma_x(c + 4, big_a + (unsigned)(xs4 * XCH) * 16u);
        __builtin_amdgcn_sched_barrier(0);
        const unsigned wb = wst_a + (unsigned)(ws * WCH) * 16u + lane16;
        const unsigned xb = big_a + (unsigned)(xs * XCH) * 16u + lane16;
#pragma unroll
        for (int s = 0; s < KPB; ++s) {
            lds_wait(fr[s & 1]);
            if (s + 1 < KPB) read_l1(wb, xb, s + 1, fr[(s + 1) & 1]);
            else if (c + 1 < NC1)
                read_l1(wst_a + (unsigned)(ws1 * WCH) * 16u + lane16, big_a + (unsigned)(xs1 * XCH) * 16u + lane16, 0, fr[0]);
            __builtin_amdgcn_sched_barrier(0);
#pragma unroll
            for (int kk = 0; kk < 4; ++kk)
#pragma unroll
                for (int u = 0; u < NU; ++u)
                    acc[u] = __builtin_amdgcn_mfma_f32_16x16x4f32(fr[s & 1][O::cb(u) - CB0][kk], fr[s & 1][NCB + O::rg(u)][kk], acc[u], 0, 0, 0);
            __builtin_amdgcn_sched_barrier(0);  // (the next step's lds_wait stays behind these MFMAs)
        }
        // chunk c + 2 of the weights and c + 3 of the rows are in LDS when every wave has passed this fence; this wave's newest
        // x segments (chunk c + 4) may fly on
        // (vmcnt only: the read-ahead of the next chunk's first step stays in flight across the barrier — its buffers are not
        // written again before the iteration after next; this chunk's own reads were waited for before their MFMAs)
        if (xmore && XW) __builtin_amdgcn_s_waitcnt(0x0F70 | KPB);
        else __builtin_amdgcn_s_waitcnt(0x0F70);
        if (!(NPLDA_FLEX_ABL & 1)) __builtin_amdgcn_s_barrier();
        __builtin_amdgcn_sched_barrier(0);
        ws = ws1;
        xs = xs1;
    }

    // ---- F.normalize (utils/models.py:368): partial row sums of squares per wave, summed over the waves in a fixed order ---
    {
        float ss[RG];
#pragma unroll
        for (int r = 0; r < RG; ++r) ss[r] = 0.f;
#pragma unroll
        for (int u = 0; u < NU; ++u)
#pragma unroll
            for (int r = 0; r < 4; ++r) ss[O::rg(u)] = fmaf(acc[u][r], acc[u][r], ss[O::rg(u)]);
#pragma unroll
        for (int r = 0; r < RG; ++r) {
            float v = wave_xor_add(ss[r], 16);
            v = wave_xor_add(v, 32);
            if (g == 0) ssb[W][r][j] = v;
        }
    }
    __syncthreads();  // (also: every wave is done reading the x ring — the y tiles may overwrite it)
    f32x4* ylds = big;  // [rg][kb][lane]
#pragma unroll
    for (int r = 0; r < RG; ++r) {
        if (O::needs_rg(r)) {
            float t = 0.f;
#pragma unroll
            for (int w = 0; w < 8; ++w) t += ssb[w][r][j];
            const float inv = 1.0f / fmaxf(sqrtf(t), 1e-12f);
#pragma unroll
            for (int u = 0; u < NU; ++u)
                if (O::rg(u) == r) {
                    acc[u] *= inv;
                    ylds[(r * NB + O::cb(u)) * 64 + lane] = acc[u];
                }
        }
    }
    f32x4 z[NU];
#pragma unroll
    for (int u = 0; u < NU; ++u) z[u] = b2p[4 * O::cb(u) + g];
    __syncthreads();

    // ---- layer 2: the same units, k-blocks of y from LDS; the weight ring goes on (chunks NC1, NC1 + 1 are already in) ---------
    auto read_l2 = [&](int cc, int s, int kb, f32x4 (&f)[NCB + RG]) {
        const unsigned wb = wst_a + (unsigned)((cc % NWB) * WCH) * 16u + lane16;
        const unsigned yb = big_a + lane16;
#pragma unroll
        for (int i = 0; i < NCB; ++i) f[i] = lds_read16(wb + (unsigned)((s * NB + CB0 + i) * 1024));
#pragma unroll
        for (int r = 0; r < RG; ++r)
            if (O::needs_rg(r)) f[NCB + r] = lds_read16(yb + (unsigned)((r * NB + kb) * 1024));
    };
    read_l2(NC1, 0, 0, fr[0]);
#pragma unroll
    for (int c2 = 0; c2 < ((NPLDA_FLEX_ABL & 16) ? 1 : NC2); ++c2) {
        const int cc = NC1 + c2;
        if (c2 + 2 < NC2) dma_w(w2base4 + (long long)(c2 + 2) * WCH, wst_a + (unsigned)(((cc + 2) % NWB) * WCH) * 16u);
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int s = 0; s < KPB; ++s) {
            const int kb = KPB * c2 + s;
            if (kb < NB) {
                lds_wait(fr[s & 1]);
                if (kb + 1 < NB) {
                    if (s + 1 < KPB) read_l2(cc, s + 1, kb + 1, fr[(s + 1) & 1]);
                    else read_l2(cc + 1, 0, kb + 1, fr[0]);
                }
                __builtin_amdgcn_sched_barrier(0);
#pragma unroll
                for (int kk = 0; kk < 4; ++kk)
#pragma unroll
                    for (int u = 0; u < NU; ++u)
                        z[u] = __builtin_amdgcn_mfma_f32_16x16x4f32(fr[s & 1][O::cb(u) - CB0][kk], fr[s & 1][NCB + O::rg(u)][kk], z[u], 0, 0, 0);
                __builtin_amdgcn_sched_barrier(0);
            }
        }
        __builtin_amdgcn_s_waitcnt(0x0F70);  // vmcnt(0): the chunk after next is in
        __builtin_amdgcn_s_barrier();
        __builtin_amdgcn_sched_barrier(0);
    }

    // ---- score: s = sum_f Q (z1^2 + z2^2) + 2 P z1 z2 (utils/models.py:372-376); z2 of lane j is z of lane j ^ 8 ------------
    {
        float part[RG];
#pragma unroll
        for (int r = 0; r < RG; ++r) part[r] = 0.f;
#pragma unroll
        for (int u = 0; u < NU; ++u) {
            const f32x4 q = Qp[4 * O::cb(u) + g];
            const f32x4 p = Pp[4 * O::cb(u) + g];
            float t = 0.f;
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const float z1 = z[u][r];
                const float z2 = dpp_f32<0x128>(z1);  // row_ror:8
                t = fmaf(q[r], fmaf(z1, z1, z2 * z2), t);
                t = fmaf(2.0f * p[r], z1 * z2, t);
            }
            part[O::rg(u)] += t;
        }
#pragma unroll
        for (int r = 0; r < RG; ++r) {
            float v = wave_xor_add(part[r], 16);
            v = wave_xor_add(v, 32);
            if (g == 0 && j < 8) scb[W][r][j] = v;
        }
    }
    __syncthreads();
    {
        const int t = W * 64 + lane;
        if (t < RG * 8) {
            const int r = t >> 3, jj = t & 7;
            float s = 0.f;
#pragma unroll
            for (int w = 0; w < 8; ++w) s += scb[w][r][jj];
            const long long p = (rg0 + r) * 8 + jj;
            if (p < a.n) a.out_s[p] = s;
        }
    }
    __syncthreads();  // the tables and the y tiles are free for the block's next super tile
}

template <int NB, int RG>
__device__ __forceinline__ void flex_tile_wave(const FwdArgs& a, long long rg0, f32x4* wst, f32x4* big, float (*ssb)[kFlexMaxRG][16],
                                               float (*scb)[kFlexMaxRG][8], const f32x4* cv, int wave, int lane, unsigned wst_a, unsigned big_a) {
    switch (wave) {
        case 0: flex_tile<NB, RG, 0>(a, rg0, wst, big, ssb, scb, cv, lane, wst_a, big_a); break;
        case 1: flex_tile<NB, RG, 1>(a, rg0, wst, big, ssb, scb, cv, lane, wst_a, big_a); break;
        case 2: flex_tile<NB, RG, 2>(a, rg0, wst, big, ssb, scb, cv, lane, wst_a, big_a); break;
        case 3: flex_tile<NB, RG, 3>(a, rg0, wst, big, ssb, scb, cv, lane, wst_a, big_a); break;
        case 4: flex_tile<NB, RG, 4>(a, rg0, wst, big, ssb, scb, cv, lane, wst_a, big_a); break;
        case 5: flex_tile<NB, RG, 5>(a, rg0, wst, big, ssb, scb, cv, lane, wst_a, big_a); break;
        case 6: flex_tile<NB, RG, 6>(a, rg0, wst, big, ssb, scb, cv, lane, wst_a, big_a); break;
        default: flex_tile<NB, RG, 7>(a, rg0, wst, big, ssb, scb, cv, lane, wst_a, big_a); break;
    }
}

// Block b works on the row groups [start, start + k): k = c for the first r blocks, c - 1 for the rest
// (c = ceil(nrg / grid), r = nrg - grid (c - 1)), in T = ceil(c / 8) super tiles of nearly equal size — every tile of the
// launch then has S = ceil(c / T) or S - 1 row groups (floor((c - 1) / T) = ceil(c / T) - 1), and the kernel is
// instantiated for that S: 16 wave programs instead of 64.
template <int NB, int S>
__global__ __launch_bounds__(512, 2) void nplda_fwd_flex_kernel(const FwdArgs a, int c, int r, int T) {
    static_assert(S <= FlexCfg<NB>::MAXRG, "super tile size");
    __shared__ f32x4 wst[FlexCfg<NB>::WST];
    __shared__ f32x4 big[FlexCfg<NB>::BIG];
    __shared__ float ssb[8][kFlexMaxRG][16];
    __shared__ float scb[8][kFlexMaxRG][8];
    __shared__ f32x4 cv[4 * NB * 4];
    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int b = blockIdx.x;
    int k = b < r ? c : c - 1;
    long long rg0 = b < r ? (long long)b * c : (long long)r * c + (long long)(b - r) * (c - 1);
    for (int i = threadIdx.x; i < 4 * NB * 4; i += 512) {
        const int v = i / (NB * 4), e = i % (NB * 4);
        const size_t o = v == 0 ? a.ob1 : (v == 1 ? a.ob2 : (v == 2 ? a.oQ : a.oP));
        cv[i] = reinterpret_cast<const f32x4*>(a.packed + o)[e];
    }
    // LDS byte addresses of the two arenas (of the arrays themselves: constants), for the hand-written fragment reads
    const unsigned wst_a = lds_addr(wst), big_a = lds_addr(big);
    __syncthreads();
    for (int tiles = T; tiles > 0 && k > 0; --tiles) {
        const int sz = (k + tiles - 1) / tiles;  // the larger tiles first
        if (sz == S) flex_tile_wave<NB, S>(a, rg0, wst, big, ssb, scb, cv, wave, lane, wst_a, big_a);
        else if constexpr (S > 1) flex_tile_wave<NB, S - 1>(a, rg0, wst, big, ssb, scb, cv, wave, lane, wst_a, big_a);
        rg0 += sz;
        k -= sz;
    }
}

// host side: grid, (c, r, T) and the instantiation
template <int NB>
static inline int launch_fwd_flex_nb(const FwdArgs& a, int cus, hipStream_t st) {
    const long long nrg = (a.n + 7) / 8;
    const long long grid = nrg < cus ? nrg : cus;
    const long long c = (nrg + grid - 1) / grid, r = nrg - grid * (c - 1);
    if (c > 0x7fffffffLL) return NPLDA_EINVAL;
    constexpr int MAXRG = FlexCfg<NB>::MAXRG;
    const int T = (int)((c + MAXRG - 1) / MAXRG);
    const int S = (int)((c + T - 1) / T);
    dim3 g((unsigned)grid), blk(512);
    switch (S) {
        case 1: hipLaunchKernelGGL((nplda_fwd_flex_kernel<NB, 1>), g, blk, 0, st, a, (int)c, (int)r, T); break;
        case 2: hipLaunchKernelGGL((nplda_fwd_flex_kernel<NB, 2>), g, blk, 0, st, a, (int)c, (int)r, T); break;
        case 3: hipLaunchKernelGGL((nplda_fwd_flex_kernel<NB, 3>), g, blk, 0, st, a, (int)c, (int)r, T); break;
        case 4: hipLaunchKernelGGL((nplda_fwd_flex_kernel<NB, 4>), g, blk, 0, st, a, (int)c, (int)r, T); break;
        case 5: hipLaunchKernelGGL((nplda_fwd_flex_kernel<NB, 5>), g, blk, 0, st, a, (int)c, (int)r, T); break;
        case 6: hipLaunchKernelGGL((nplda_fwd_flex_kernel<NB, 6>), g, blk, 0, st, a, (int)c, (int)r, T); break;
        case 7: hipLaunchKernelGGL((nplda_fwd_flex_kernel<NB, 7>), g, blk, 0, st, a, (int)c, (int)r, T); break;
        default:
            if constexpr (MAXRG >= 8) hipLaunchKernelGGL((nplda_fwd_flex_kernel<NB, (MAXRG >= 8 ? 8 : 7)>), g, blk, 0, st, a, (int)c, (int)r, T);
            else return NPLDA_EINVAL;
            break;
    }
    return nplda_launch_status();
}

}  // namespace nplda

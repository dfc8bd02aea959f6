// exp_bf3w_kernel.h — EXPERIMENT (not in libnplda_hip.so): the split-bf16 forward of csrc/nplda_fwd_bf16x3.h (same
// six-pass arithmetic, scores agree to 1.3e-6) rebuilt for one wave per SIMD on v_mfma_f32_32x32x16_bf16, 64 rows per
// wave, weights by LDS-DMA.  Result (profiles/r03g_bf16x3_power.txt): every schedule of this kernel, and the shipped
// 16-pair kernel, end at the same 4.9 - 5.1e8 pairs/s — the socket sits at its 1 400 W cap and each gain in matrix-pipe
// use comes back as a lower clock (pipe use x MHz = 1 190 - 1 240 in all variants).  Kept as the evidence for that,
// together with the scheduling lessons written next to the code (they are what the comments below are about).
//
// What was built.  The 16-pair-per-wave kernel issues v_mfma_f32_16x16x32_bf16, which one SIMD cannot issue faster
// than every ~17-27 cycles (16 would be the peak; tools/exp_mfma_clock.hip), reads 30 one-KB weight fragments from LDS
// per 120 of them and keeps two such waves on a SIMD; its waves sit 42 % of the time on s_waitcnt (profiles/r01n_*).
// The 32x32x16 form does the same FLOP in half the instructions, is paced at exactly 32 cycles and reads half the
// operand bytes per FLOP.  Here a wave owns TWO column groups of 32 rows (both sides of 32 pairs; 64 consecutive rows
// in embedding mode) and all feature blocks of 32, so a fragment read feeds 2 x 6 MFMAs of 32 cycles, a block of 4 waves
// streams the image once per 128 pairs, and with 512 registers a wave holds both layers' accumulators (2 x 32 NB2),
// two sets of B-operand pieces and a two-step x ring.  One wave per SIMD means nothing else hides a wave's own
// latencies, so the kernel is software-pipelined by hand, one k32-step (2 k16 sub-steps x NB2 blocks x 3 fragments;
// 120 MFMAs at NB2 = 5) per barrier:
//   * the weights of step t + 1 go global -> LDS by LDS-DMA during step t (two stages; no staging registers);
//   * the x of step t + 2 is loaded during step t; it is split into bf16 pieces during step t + 1, half a split
//     (22 VALU) per half-unit of 12 MFMAs, placed by sched_group_barrier (one MFMA, up to four VALU in turn);
//   * a unit's weight fragments are read one unit ahead; the step's fence (vmcnt(0), barrier) opens the LAST unit, which
//     then reads the first fragments of the next stage — no LDS latency is exposed at a step boundary;
//   * layer 2 (k = the normalised layer-1 features, straight from the accumulator registers: 8 registers of a 32-block
//     are the 8 k-values a lane needs for one k16 sub-step; the W2 image is packed with that permutation) splits the y
//     block of its next step the same way; its last steps load the next tile's first x and split it, and the DMA never
//     stops across tiles (persistent grid, one block per CU).
// NB = 11 / 12 (D = 170: NB2 = 6) does not fit: 2 x 12 accumulator tuples of 16 registers against 16 tuples of AGPRs;
// hipcc spills 2.3 KB per lane and the kernel runs 8 x slower.  Build with -fno-slp-vectorize (see bf3_cvt_pk).
#pragma once
#include "../neuralplda_amd/csrc/nplda_fwd_bf16x3.h"

#ifndef NPLDA_BF3W_ABL
#define NPLDA_BF3W_ABL 0  // tools/exp_bf3.hip: 1 no split arithmetic, 2 no x loads, 4 no weight DMA, 8 no barrier, 16 x rows from L2 (timing only)
#endif

namespace nplda {

#ifdef NPLDA_BF3W_STAMPS
__device__ unsigned long long g_bf3w_stamps[4];  // block 0, wave 0: wall clock (100 MHz) and shader clock at entry / exit
__device__ unsigned long long g_bf3w_steps[8];  // block 0, waves 0..3: cycles spent waiting for loads / at the barrier, all fences
#endif

typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef unsigned u32x4 __attribute__((ext_vector_type(4)));
#define NPLDA_MFMA32_BF16(A, B, C) __builtin_amdgcn_mfma_f32_32x32x16_bf16(A, B, C, 0, 0, 0)

// image of this kernel (units: floats; every bf16 fragment = 64 lanes x 16 B = 256 floats-worth), NB2 = ceil(NB/2):
//   W1c[c][s][b][part][lane]   lane (f = lane & 31, h = lane >> 5), element e: W1[32 b + f][32 c + 16 s + (e & 3) + 8 (e >> 2) + 4 h]
//   W2c[c2][t][b][part][lane]  element e: W2[32 b + f][32 c2 + 16 t + (e & 3) + 8 (e >> 2) + 4 h]   (c2 < NB2)
//   then fp32 b1, b2, Q, P (NB*16 each).  KC1 = k32-steps of layer 1, padded to even (a padding step has zero weights).
struct Bf3wLayout {
    int D0, D1, D2, NB, NB2, KC1;
    size_t oW1c, oW2c, ob1, ob2, oQ, oP, total;
};
__host__ __device__ inline Bf3wLayout bf3w_layout(int D0, int D1, int D2) {
    Bf3wLayout L;
    L.D0 = D0; L.D1 = D1; L.D2 = D2;
    L.NB = nplda_kernel_nb(D1, D2);
    L.NB2 = (L.NB + 1) / 2;
    L.KC1 = 2 * ((D0 + 63) / 64);
    L.oW1c = 0;
    L.oW2c = (size_t)L.KC1 * 2 * L.NB2 * 3 * 256;
    L.ob1 = L.oW2c + (size_t)L.NB2 * 2 * L.NB2 * 3 * 256;
    L.ob2 = L.ob1 + (size_t)L.NB * 16;
    L.oQ = L.ob2 + (size_t)L.NB * 16;
    L.oP = L.oQ + (size_t)L.NB * 16;
    L.total = L.oP + (size_t)L.NB * 16;
    return L;
}
// one thread per 16-byte fragment element (8 bf16) or per fp32 tail element
static __global__ void bf3w_pack_kernel(const float* __restrict__ W1, const float* __restrict__ b1, const float* __restrict__ W2,
                                        const float* __restrict__ b2, const float* __restrict__ P_sqrt, const float* __restrict__ Q,
                                        Bf3wLayout L, float* __restrict__ out) {
    const size_t nfrag = L.ob1 / 4, ntail = L.total - L.ob1;
    const size_t idx = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (idx < nfrag) {
        const bool second = idx >= L.oW2c / 4;
        size_t rel = second ? idx - L.oW2c / 4 : idx;
        const int lane = (int)(rel & 63);
        rel >>= 6;
        const int part = (int)(rel % 3);
        rel /= 3;
        const int b = (int)(rel % L.NB2);
        rel /= L.NB2;
        const int s = (int)(rel & 1), c = (int)(rel >> 1);
        const int f = 32 * b + (lane & 31), hh = lane >> 5;
        bf16x8 o;
#pragma unroll
        for (int e = 0; e < 8; ++e) {
            const int k = 32 * c + 16 * s + (e & 3) + 8 * (e >> 2) + 4 * hh;
            float v = 0.f;
            if (!second) { if (f < L.D1 && k < L.D0) v = W1[(size_t)f * L.D0 + k]; }
            else { if (f < L.D2 && k < L.D1) v = W2[(size_t)f * L.D1 + k]; }
            __bf16 h, m, l;
            split3(v, h, m, l);
            o[e] = part == 0 ? h : (part == 1 ? m : l);
        }
        reinterpret_cast<bf16x8*>(out)[idx] = o;
    } else if (idx < nfrag + ntail) {
        const size_t t = idx - nfrag;
        const size_t nb16 = (size_t)L.NB * 16;
        const int f = (int)(t % nb16), which = (int)(t / nb16);
        float v = 0.f;
        if (which == 0) { if (f < L.D1) v = b1[f]; }
        else if (which == 1) { if (f < L.D2) v = b2[f]; }
        else if (which == 2) { if (f < L.D2) v = Q[f]; }
        else { if (f < L.D2) v = P_sqrt[f] * P_sqrt[f]; }
        out[L.ob1 + t] = v;
    }
}
struct Bf3wArgs {
    const float* xa;
    const float* xb;
    long long n, ldx;
    const float* img;
    int D0, KC1;
    size_t oW1c, oW2c, ob1, ob2, oQ, oP;
    float* out_s;
    float* out_z;
    long long ldz;
    float* out_q;
};

struct Pc3 { u32x4 h, m, l; };  // 8 bf16 each (the MFMA's B operand), kept as dwords: a split fills them a pair at a time

typedef float f32x2 __attribute__((ext_vector_type(2)));
typedef __bf16 bf16x2 __attribute__((ext_vector_type(2)));
// v_cvt_pk_bf16_f32 / v_sub_f32 as plain IR (so that sched_group_barrier sees VALU instructions, which it does not in
// inline asm); the translation unit is built with -fno-slp-vectorize (build.py), or hipcc packs the subtractions into
// v_pk_add_f32 — ~13 cycles each beside MFMAs (MI355X_MICROARCH.md, "price of one filler")
__device__ __forceinline__ unsigned bf3_cvt_pk(float a, float b) {
    const f32x2 v = {a, b};
    return __builtin_bit_cast(unsigned, __builtin_convertvector(v, bf16x2));
}
// v = h + m + l (split3x8's arithmetic) for values 4 half .. 4 half + 3 of the 8 in (lo, hi): 22 single-issue VALU
__device__ __forceinline__ void bf3_split_half(const f32x4 lo, const f32x4 hi, Pc3& P, int half) {
    if (NPLDA_BF3W_ABL & 1) {  // timing only: no arithmetic, the pieces are raw bits
        const u32x4 t = __builtin_bit_cast(u32x4, half ? hi : lo);
        P.h[2 * half] = t[0]; P.h[2 * half + 1] = t[1]; P.m[2 * half] = t[2]; P.m[2 * half + 1] = t[3];
        P.l[2 * half] = t[1]; P.l[2 * half + 1] = t[2];
    } else {
        const f32x4 v = half ? hi : lo;
#pragma unroll
        for (int q = 0; q < 2; ++q) {
            const float v0 = v[2 * q], v1 = v[2 * q + 1];
            const unsigned h = bf3_cvt_pk(v0, v1);
            const float r0 = v0 - __builtin_bit_cast(float, h << 16);
            const float r1 = v1 - __builtin_bit_cast(float, h & 0xffff0000u);
            const unsigned m = bf3_cvt_pk(r0, r1);
            const float s0 = r0 - __builtin_bit_cast(float, m << 16);
            const float s1 = r1 - __builtin_bit_cast(float, m & 0xffff0000u);
            P.h[2 * half + q] = h;
            P.m[2 * half + q] = m;
            P.l[2 * half + q] = bf3_cvt_pk(s0, s1);
        }
    }
    // made HERE: hipcc otherwise sinks a split into the step that consumes it
    asm volatile("" : "+v"(P.h[2 * half]), "+v"(P.h[2 * half + 1]), "+v"(P.m[2 * half]), "+v"(P.m[2 * half + 1]),
                 "+v"(P.l[2 * half]), "+v"(P.l[2 * half + 1]));
}
__device__ __forceinline__ void bf3_split8(const f32x4 lo, const f32x4 hi, Pc3& P) {
    bf3_split_half(lo, hi, P, 0);
    bf3_split_half(lo, hi, P, 1);
}

// NB = 16-feature blocks of the model (nplda_kernel_nb); the kernel works in NB2 = ceil(NB / 2) blocks of 32
template <int NB, int MODE>
__global__ __launch_bounds__(256, 1) void nplda_fwd_bf3w_kernel(const Bf3wArgs a, int ntiles) {
    static_assert(MODE == MODE_PAIR || MODE == MODE_EMBED, "bf16x3 kernel modes");
    constexpr int WAVES = 4;
    constexpr int NB2 = (NB + 1) / 2;
    constexpr int NU = (NB2 + 1) / 2;         // units (pairs of 32-blocks) per k16 sub-step
    constexpr int NUT = 2 * NU;               // units per k32-step
    constexpr int SEG = 2 * NB2 * 3;          // 1 KB fragments per k32-step
    constexpr int STEP16 = SEG * 64;          // 16-byte units per step
    constexpr int KC2 = NB2;
    constexpr int NDMA = (SEG + WAVES - 1) / WAVES;
    __shared__ f32x4 wbuf[2][STEP16];
    __shared__ f32x4 cvec[4][NB2 * 8];
    __shared__ f32x4 sink[64];

    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int n = lane & 31;
    const int h = lane >> 5;

    const f32x4* Wc = reinterpret_cast<const f32x4*>(a.img + a.oW1c);
    const long long w2rel = (long long)((a.oW2c - a.oW1c) / 4);
    for (int i = tid; i < 4 * NB2 * 8; i += WAVES * 64) {
        const int v = i / (NB2 * 8), e = i % (NB2 * 8);
        const size_t o = v == 0 ? a.ob1 : (v == 1 ? a.ob2 : (v == 2 ? a.oQ : a.oP));
        const f32x4 zero = {0.f, 0.f, 0.f, 0.f};
        cvec[v][e] = e < NB * 4 ? reinterpret_cast<const f32x4*>(a.img + o)[e] : zero;
    }
    const f32x4* b1p = cvec[0];
    const f32x4* b2p = cvec[1];
    const f32x4* Qp = cvec[2];
    const f32x4* Pp = cvec[3];
    const int KC1 = a.KC1;  // even (bf3_layout)
    const int D0 = a.D0;

    // one step of weights: SEG segments of 1 KB, lane l's 16 bytes land at dst[l] (see nplda_fwd_v5.h: seg_dma).  The
    // step's memory instructions are NOT issued in one burst at its start — the four waves of the block leave the barrier
    // together, 16 instructions each into one texture-address queue, and sat there for ~2 000 cycles per step with the
    // matrix pipe idle — but in parts, a few per unit, placed between MFMAs (part v of VU).
    // A step's memory instructions, 8 x loads and 8 DMA segments per wave.  With one wave per SIMD a memory instruction
    // that has to queue for the CU's one address unit stalls the wave — and its matrix pipe — for as long as it queues,
    // and the four waves of a block leave every barrier together: issued at the same points of the step they cost
    // ~1 500 cycles per step (tools/exp_bf3.hip: pipe use 0.81 without the x loads, 0.59 with them even when the split
    // arithmetic is taken out).  So every wave has its OWN issue point: the step is 2 NUT half-units, wave w issues its
    // x loads at the start of half-unit w and its DMA segments at the start of half-unit 6 + w (uniform branches; the
    // 32-cycle MFMA in flight covers a skipped one).  All of it is older than a unit when the step's fence (last unit,
    // vmcnt(0)) arrives, the x loads three units or more; after the fence nothing is pending, so the splits of the next
    // step never wait — hipcc treats an LDS-DMA in flight as a FLAT access and turns any wait beside one into vmcnt(0).
    constexpr bool STAGGER = NUT >= 6;
    constexpr int DPOS0 = STAGGER ? 6 : 2 * (NUT - 2);       // half-unit of wave 0's DMA
    auto dma_step = [&](const f32x4* src, f32x4* dst) {
#pragma unroll
        for (int i = 0; i < NDMA; ++i) {
            if (NPLDA_BF3W_ABL & 4) break;
            const int sgm = wave + WAVES * i;
            const bool live = sgm < SEG;
            unsigned lo = (unsigned)lane * 16u;
            asm volatile("" : "+v"(lo));
            __builtin_amdgcn_global_load_lds(
                (const __attribute__((address_space(1))) void*)(reinterpret_cast<const char*>(src + (live ? sgm : 0) * 64) + lo),
                (__attribute__((address_space(3))) void*)(live ? dst + sgm * 64 : sink), 16, 0, 0);
        }
    };
#ifdef NPLDA_BF3W_STAMPS
    unsigned long long t_load = 0, t_bar = 0;  // cycles this wave spent waiting for its loads / at the barrier
#endif
    auto fence = [&]() {
#ifdef NPLDA_BF3W_STAMPS
        __builtin_amdgcn_sched_barrier(0);
        const unsigned long long t0 = __builtin_amdgcn_s_memtime();
        __builtin_amdgcn_s_waitcnt(0xC07F);
        __builtin_amdgcn_sched_barrier(0);
#endif
        __builtin_amdgcn_s_waitcnt(0x0070);  // vmcnt(0) lgkmcnt(0): this wave's fragments of the next step are in LDS
#ifdef NPLDA_BF3W_STAMPS
        __builtin_amdgcn_sched_barrier(0);
        const unsigned long long t1 = __builtin_amdgcn_s_memtime();
        __builtin_amdgcn_s_waitcnt(0xC07F);
        __builtin_amdgcn_sched_barrier(0);
#endif
        if (!(NPLDA_BF3W_ABL & 8)) __builtin_amdgcn_s_barrier();
#ifdef NPLDA_BF3W_STAMPS
        __builtin_amdgcn_sched_barrier(0);
        const unsigned long long t2 = __builtin_amdgcn_s_memtime();
        t_load += t1 - t0;
        t_bar += t2 - t1;
#endif
        __builtin_amdgcn_sched_barrier(0);
    };

    // column group gi of tile t: pair mode = the x1 rows and the x2 rows of the wave's 32 pairs; embedding mode 2 x 32 rows
    auto group_rows = [&](long long t, const float* (&p)[2], long long& r0) {
        r0 = (t * WAVES + wave) * (MODE == MODE_PAIR ? 32 : 64);
#pragma unroll
        for (int gi = 0; gi < 2; ++gi) {
            long long r = r0 + (MODE == MODE_PAIR ? 0 : 32 * gi) + n;
            if (r >= a.n) r = a.n - 1;
            if (NPLDA_BF3W_ABL & 16) r &= 511;  // timing only: every x row comes from L2
            p[gi] = ((MODE == MODE_PAIR && gi) ? a.xb : a.xa) + r * a.ldx;
        }
    };
    // x of one k32-step: load q of lane (n, h) takes columns 32 c + 8 q + 4 h + 0..3 of its row (the two lanes of a row
    // share a 32-byte sector in every instruction); sub-step s uses loads 2 s, 2 s + 1 — the W1 image is packed to match
    struct XS { f32x4 v[2][4]; };
    auto loadx = [&](XS& X, const float* const (&p)[2], int kc) {
        const int k0 = 32 * kc + 4 * h;
        if (NPLDA_BF3W_ABL & 2) return;
#pragma unroll
        for (int i = 0; i < 8; ++i) X.v[i >> 2][i & 3] = load_x4c<false>(p[i >> 2], k0 + 8 * (i & 3), D0);
    };
    // this wave's memory instructions of half-unit `pos`: lx() the x loads, ld() the DMA
    auto vmem_at = [&](int pos, auto&& lx, auto&& ld) {
        if (STAGGER) {
            if (pos < 4) { if (wave == pos) lx(); }
            else if (pos >= DPOS0 && pos < DPOS0 + 4) { if (wave == pos - DPOS0) ld(); }
        } else {
            if (pos == 0) lx();
            if (pos == DPOS0) ld();
        }
    };
    // split k (= 2 gi + s) of the next step's pieces is made among the MFMAs of this unit
    auto split_unit = [](int k) { return NUT >= 6 ? k : (NUT == 4 ? k / 2 : 0); };
    auto split_x = [&](const XS& X, Pc3 (&P)[2][2], int u, int half) {
#pragma unroll
        for (int k = 0; k < 4; ++k)
            if (split_unit(k) == u) bf3_split_half(X.v[k >> 1][2 * (k & 1)], X.v[k >> 1][2 * (k & 1) + 1], P[k >> 1][k & 1], half);
    };

    // fragments of unit u = (sub-step s, block pair bp) in a stage: [s][b][part][lane]
    struct WU { WFrag b0, b1; };
    // The fragment reads are inline asm on purpose.  hipcc orders every LDS read it can see behind ALL LDS-DMA writes in
    // flight (it cannot tell the two stages apart) with an s_waitcnt vmcnt(0) — which also waits for the x loads issued
    // before them: ~2 000 cycles of memory latency per step with the matrix pipe idle.  The real ordering is the fence's.
    // A unit's reads are waited for (lgkmcnt(0)) at the END of the unit that issues them, long after they have landed, so
    // that any register copy the compiler makes of them afterwards copies data.
    auto lds_frag = [&](unsigned addr, int off) {
        u32x4 r;
        asm volatile("ds_read_b128 %0, %1 offset:%2" : "=v"(r) : "v"(addr), "n"(off));
        return __builtin_bit_cast(bf16x8, r);
    };
    auto read_unit = [&](unsigned wst, int u, WU& w) {  // wst: LDS byte address of the stage + 16 lane
        const int s = u / NU, b0 = 2 * (u % NU), b1 = b0 + 1 < NB2 ? b0 + 1 : b0;
        w.b0.h = lds_frag(wst, ((s * NB2 + b0) * 3 + 0) * 1024);
        w.b0.m = lds_frag(wst, ((s * NB2 + b0) * 3 + 1) * 1024);
        w.b0.l = lds_frag(wst, ((s * NB2 + b0) * 3 + 2) * 1024);
        if (b0 + 1 < NB2) {
            w.b1.h = lds_frag(wst, ((s * NB2 + b1) * 3 + 0) * 1024);
            w.b1.m = lds_frag(wst, ((s * NB2 + b1) * 3 + 1) * 1024);
            w.b1.l = lds_frag(wst, ((s * NB2 + b1) * 3 + 2) * 1024);
        }
    };
    auto reads_landed = [&]() { __builtin_amdgcn_s_waitcnt(0xC07F); };  // lgkmcnt(0)
    const unsigned lds0 = (unsigned)(unsigned long long)(__attribute__((address_space(3))) char*)(&wbuf[0][0]) + 16u * lane;
    auto stage_addr = [&](int pp) { return lds0 + (unsigned)pp * (STEP16 * 16u); };
    // one k32-step: acc[gi][b] += W[b] (stage `wst`) x P[gi][s], six passes, small terms first.  wc holds the current
    // unit's fragments (already in registers at step entry); side(u) runs among the MFMAs of unit u, and a sched_barrier
    // closes every unit so that neither moves.  The LAST unit opens with the step's fence — by then every wave has read
    // all of this stage, and the DMA issued at the step's start has had most of a step to land — and reads unit 0 of the
    // next stage `wnx`.
    WU wc;
    auto step = [&](unsigned wst, unsigned wnx, const Pc3 (&P)[2][2], f32x16 (&acc)[2][NB2], auto&& vm, auto&& sp) {
#pragma unroll
        for (int u = 0; u < NUT; ++u) {
            const int s = u / NU, b0 = 2 * (u % NU);
            const bool two = b0 + 1 < NB2;
            const int b1 = two ? b0 + 1 : b0;
            WU wn;
            if (u + 1 < NUT) {
                read_unit(wst, u + 1, wn);
            } else {
                fence();
                read_unit(wnx, 0, wn);
            }
            __builtin_amdgcn_sched_barrier(0);  // the reads open the unit
#define NPLDA_PASS(WP, XP)                                                                                                   \
    acc[0][b0] = NPLDA_MFMA32_BF16(wc.b0.WP, __builtin_bit_cast(bf16x8, P[0][s].XP), acc[0][b0]);                             \
    acc[1][b0] = NPLDA_MFMA32_BF16(wc.b0.WP, __builtin_bit_cast(bf16x8, P[1][s].XP), acc[1][b0]);                             \
    if (two) {                                                                                                               \
        acc[0][b1] = NPLDA_MFMA32_BF16(wc.b1.WP, __builtin_bit_cast(bf16x8, P[0][s].XP), acc[0][b1]);                         \
        acc[1][b1] = NPLDA_MFMA32_BF16(wc.b1.WP, __builtin_bit_cast(bf16x8, P[1][s].XP), acc[1][b1]);                         \
    }
#pragma unroll
            for (int half = 0; half < 2; ++half) {
                vm(2 * u + half);
                __builtin_amdgcn_sched_barrier(0);  // a wave's memory instructions sit at the head of their half-unit
                sp(u, half);
                if (half == 0) {
                    NPLDA_PASS(m, m)
                    NPLDA_PASS(h, l)
                    NPLDA_PASS(l, h)
                } else {
                    NPLDA_PASS(h, m)
                    NPLDA_PASS(m, h)
                    NPLDA_PASS(h, h)
                }
                // one MFMA and at most four fillers in turn (a 32-cycle MFMA hides five)
#pragma unroll
                for (int i = 0; i < (two ? 12 : 6); ++i) {
                    __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
                    __builtin_amdgcn_sched_group_barrier(0x002, 4, 0);
                }
                __builtin_amdgcn_sched_barrier(0);
            }
#undef NPLDA_PASS
            reads_landed();
            __builtin_amdgcn_sched_barrier(0);
            wc = wn;
        }
    };

#ifdef NPLDA_BF3W_STAMPS
    if (blockIdx.x == 0 && tid == 0) {
        g_bf3w_stamps[0] = __builtin_amdgcn_s_memrealtime();
        g_bf3w_stamps[1] = __builtin_amdgcn_s_memtime();
    }
#endif
    long long tile = blockIdx.x;
    const float* px[2];
    long long r0;
    group_rows(tile, px, r0);

    // ---- prologue: weights of step 0, x of steps 0 and 1, pieces of step 0 ---------------------------------------
    dma_step(Wc, wbuf[0]);
    XS X0, X1;
    loadx(X0, px, 0);
    loadx(X1, px, 1);
    Pc3 Pa[2][2], Pb[2][2];
#pragma unroll
    for (int u = 0; u < NUT; ++u) { split_x(X0, Pa, u, 0); split_x(X0, Pa, u, 1); }
    __syncthreads();  // cvec
    fence();
    read_unit(stage_addr(0), 0, wc);
    reads_landed();
    int par = 0;

    for (;;) {
        const long long tile_n = tile + gridDim.x;
        const float* pn[2];
        long long r0_n;
        group_rows(tile_n, pn, r0_n);

        // accumulator register r of block b holds feature 32 b + 8 (r >> 2) + 4 h + (r & 3) of row n
        f32x16 acc[2][NB2];
#pragma unroll
        for (int b = 0; b < NB2; ++b)
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                const f32x4 bv = b1p[8 * b + 2 * q + h];
#pragma unroll
                for (int r = 0; r < 4; ++r) acc[0][b][4 * q + r] = acc[1][b][4 * q + r] = bv[r];
            }

        // ---- layer 1: two steps per trip (static names for the two piece sets and the two x slots) ------------------
        for (int c = 0; c < KC1; c += 2) {
            {   // step c: pieces Pa; x of c + 1 (X1) -> Pb; x of c + 2 -> X0
                const f32x4* wsrc = Wc + (long long)(c + 1) * STEP16;
                const int kx = c + 2 < KC1 ? c + 2 : KC1 - 1;
                step(stage_addr(par), stage_addr(par ^ 1), Pa, acc,
                     [&](int pos) { vmem_at(pos, [&]() { loadx(X0, px, kx); }, [&]() { dma_step(wsrc, wbuf[par ^ 1]); }); },
                     [&](int u, int half) { split_x(X1, Pb, u, half); });
                par ^= 1;
            }
            {   // step c + 1: pieces Pb; x of c + 2 (X0) -> Pa; x of c + 3 -> X1
                const bool last = c + 2 >= KC1;
                const f32x4* wsrc = last ? Wc + w2rel : Wc + (long long)(c + 2) * STEP16;
                const int kx = c + 3 < KC1 ? c + 3 : KC1 - 1;
                step(stage_addr(par), stage_addr(par ^ 1), Pb, acc,
                     [&](int pos) { vmem_at(pos, [&]() { loadx(X1, px, kx); }, [&]() { dma_step(wsrc, wbuf[par ^ 1]); }); },
                     [&](int u, int half) { split_x(X0, Pa, u, half); });
                par ^= 1;
            }
        }

        // ---- F.normalize, pieces of layer 2's first step --------------------------------------------------------------
#pragma unroll
        for (int gi = 0; gi < 2; ++gi) {
            float ss = 0.f;
#pragma unroll
            for (int b = 0; b < NB2; ++b)
#pragma unroll
                for (int r = 0; r < 16; ++r) ss = fmaf(acc[gi][b][r], acc[gi][b][r], ss);
            ss = wave_xor_add(ss, 32);
            const float inv = 1.0f / fmaxf(sqrtf(ss), 1e-12f);
#pragma unroll
            for (int b = 0; b < NB2; ++b) acc[gi][b] *= inv;
        }
        // layer-2 step c2 consumes y block c2: sub-step t = its registers 8 t .. 8 t + 7
        auto split_y = [&](int c2, Pc3 (&P)[2][2], int u, int half) {
#pragma unroll
            for (int k = 0; k < 4; ++k) {
                if (split_unit(k) == u) {
                    const int gi = k >> 1, t = k & 1;
                    const f32x16& y = acc[gi][c2 < NB2 ? c2 : 0];
                    const f32x4 lo = {y[8 * t], y[8 * t + 1], y[8 * t + 2], y[8 * t + 3]};
                    const f32x4 hi = {y[8 * t + 4], y[8 * t + 5], y[8 * t + 6], y[8 * t + 7]};
                    bf3_split_half(lo, hi, P[gi][t], half);
                }
            }
        };
#pragma unroll
        for (int u = 0; u < NUT; ++u) { split_y(0, Pa, u, 0); split_y(0, Pa, u, 1); }
        // y block 1 as well: two blocks of y are then dead when the 2 NB2 accumulator tuples of layer 2 come alive
        if (KC2 > 1) {
#pragma unroll
            for (int u = 0; u < NUT; ++u) { split_y(1, Pb, u, 0); split_y(1, Pb, u, 1); }
        }

        // ---- layer 2; the last steps fetch and split the next tile's x ---------------------------------------------------
        f32x16 z[2][NB2];
#pragma unroll
        for (int b = 0; b < NB2; ++b)
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                const f32x4 bv = b2p[8 * b + 2 * q + h];
#pragma unroll
                for (int r = 0; r < 4; ++r) z[0][b][4 * q + r] = z[1][b][4 * q + r] = bv[r];
            }
        // the next tile's x: step 0's in layer-2 step LX0, step 1's in LX1 (a single-step layer 2 loads both in its only step)
        constexpr int LX0 = KC2 >= 3 ? KC2 - 3 : 0;
        constexpr int LX1 = KC2 >= 3 ? KC2 - 2 : (KC2 == 2 ? 1 : -1);
#pragma unroll
        for (int c2 = 0; c2 < KC2; ++c2) {
            const bool more2 = c2 + 1 < KC2;
            const f32x4* wsrc = more2 ? Wc + w2rel + (long long)(c2 + 1) * STEP16 : Wc;
            auto vm = [&, c2, wsrc](int pos) {
                vmem_at(pos,
                        [&]() {
                            if (c2 == LX0) loadx(X0, pn, 0);
                            if (c2 == LX1 || KC2 == 1) loadx(X1, pn, 1);
                        },
                        [&]() { dma_step(wsrc, wbuf[par ^ 1]); });
            };
            auto sp = [&](Pc3 (&Pn)[2][2]) {
                return [&, c2](int u, int half) {
                    if (c2 + 1 < KC2) { if (c2 > 0) split_y(c2 + 1, Pn, u, half); }
                    else split_x(X0, Pn, u, half);
                };
            };
            if ((c2 & 1) == 0) step(stage_addr(par), stage_addr(par ^ 1), Pa, z, vm, sp(Pb));
            else step(stage_addr(par), stage_addr(par ^ 1), Pb, z, vm, sp(Pa));
            par ^= 1;
        }
        if (KC2 & 1) {  // the next tile's first pieces were made in Pb
#pragma unroll
            for (int k = 0; k < 4; ++k) Pa[k >> 1][k & 1] = Pb[k >> 1][k & 1];
        }

        // ---- epilogue (fp32, as the fp32 kernels) -----------------------------------------------------------------------
        if (MODE == MODE_PAIR) {
            float part = 0.f;
#pragma unroll
            for (int b = 0; b < NB2; ++b)
#pragma unroll
                for (int q = 0; q < 4; ++q) {
                    const f32x4 qv = Qp[8 * b + 2 * q + h];
                    const f32x4 pv = Pp[8 * b + 2 * q + h];
#pragma unroll
                    for (int r = 0; r < 4; ++r) {
                        const float z1 = z[0][b][4 * q + r], z2 = z[1][b][4 * q + r];
                        part = fmaf(qv[r], fmaf(z1, z1, z2 * z2), part);
                        part = fmaf(2.0f * pv[r], z1 * z2, part);
                    }
                }
            part = wave_xor_add(part, 32);
            const long long row = r0 + n;
            if (h == 0 && row < a.n) a.out_s[row] = part;
        } else {
#pragma unroll
            for (int gi = 0; gi < 2; ++gi) {
                const long long row = r0 + 32 * gi + n;
                const bool ok = row < a.n;
                float qa = 0.f;
#pragma unroll
                for (int b = 0; b < NB2; ++b)
#pragma unroll
                    for (int q = 0; q < 4; ++q) {
                        const f32x4 qv = Qp[8 * b + 2 * q + h];
                        f32x4 zv;
#pragma unroll
                        for (int r = 0; r < 4; ++r) {
                            zv[r] = z[gi][b][4 * q + r];
                            qa = fmaf(qv[r] * zv[r], zv[r], qa);
                        }
                        if (32 * b + 8 * q < 16 * NB && ok)
                            *reinterpret_cast<f32x4*>(a.out_z + row * a.ldz + 32 * b + 8 * q + 4 * h) = zv;
                    }
                if (a.out_q != nullptr) {
                    qa = wave_xor_add(qa, 32);
                    if (h == 0 && ok) a.out_q[row] = qa;
                }
            }
        }

        tile = tile_n;
        if (tile >= ntiles) break;
        r0 = r0_n;
        px[0] = pn[0];
        px[1] = pn[1];
    }
#ifdef NPLDA_BF3W_STAMPS
    if (blockIdx.x == 0 && tid == 0) {
        g_bf3w_stamps[2] = __builtin_amdgcn_s_memrealtime();
        g_bf3w_stamps[3] = __builtin_amdgcn_s_memtime();
    }
    if (blockIdx.x == 0 && lane == 0) {
        g_bf3w_steps[2 * wave] = t_load;
        g_bf3w_steps[2 * wave + 1] = t_bar;
    }
#endif
}

}  // namespace nplda

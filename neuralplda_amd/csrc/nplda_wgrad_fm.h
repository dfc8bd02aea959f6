// nplda_wgrad_fm.h — work descriptors of the weight-gradient ("A^T B", split-K) kernels of nplda_backward.hip and the
// full-M form of the kernel for the recipe sizes.
#pragma once
#include "nplda_cohort_qz.h"
#include "nplda_common.h"
#include "nplda_loss_tail.h"

namespace nplda {

// C[m][n] = sum_k A[k][m] * Bm[k][n]   (K rows; Bm rows come from two segments)
struct WgradProblem {
    const float* A;       // (2n, lda): dz or du
    long long lda;
    const float* B0;      // rows [0, n)
    const float* B1;      // rows [n, 2n)
    long long ldb;
    int M, N;             // valid columns of A / of B (multiples of 4)
    int MT, NT;           // 64-wide tiles
    float* slab;          // [ksplit][Mp][Np]
    int Mp, Np;
    int extras;           // 0 none, 1 = {db1 from A}, 2 = {db2 from A, dQ, dP from z and g}, 3 = {db2 from A; dQ = dP = 0},
                          // 4 = {db2 from A; the dQ / dP rows are summed by the pair-sum blocks from K-A's per-block sums}
    int b_bf16;           // full-M form, 32-column strips only: the B rows are bfloat16 (B0 / B1 point at 2-byte elements, ldb
                          // counts elements) — the head step's own bf16 x rows, read as they are and widened in registers
                          // instead of an fp32 copy staged by the first kernel
};

struct WgradArgs {
    WgradProblem p[2];
    long long K;          // rows of the "A^T B" products (2 B for pair scoring, N for embedding rows)
    long long nsplit;     // rows [0, nsplit) of B come from B0, the rest from B1 (= pairs for pair scoring)
    int ksplit;
    long long rows_per_split;  // multiple of 4
    const float* z;       // (2n, ldz)
    const float* g;       // (n)
    long long ldz;
    float* ext;           // [ksplit][4][Mp] : dQraw, dPraw, db2, db1
    int Mp;
    int nw0;              // work items of problem 0 = MT0*NT0*ksplit
    int nw;               // total work items
    const float* pq;      // [nblk][2][ldz] per-block dQ / dP sums of K-A (small-batch pair scoring), else null
    int nblk;
    int nw_mm;            // work items of the two GEMMs
    int nw_ps;            // items [nw_mm, nw_ps) = one pair-sum block per k-group; items [nw_ps, nw) = cohort first-moment
    nplda::QzArgs qz;     // blocks of the fused AS-norm pre-pass (nplda_cohort_qz.h), riding in the Gram matrix's launch
};

#ifdef NPLDA_FM_STAMPS  // tools/exp_wgrad.hip only: 100 MHz time stamps of one wave at the phase boundaries
__device__ unsigned long long g_fm_stamps[16];
#define NPLDA_FM_STAMP(i) do { if (blockIdx.x == gridDim.x / 2 && threadIdx.x == NPLDA_FM_STAMPS) g_fm_stamps[i] = __builtin_amdgcn_s_memrealtime(); } while (0)
#else
#define NPLDA_FM_STAMP(i) do {} while (0)
#endif

// ------------------------------------------------------------------------------------------------
// K-B (full-M form) : the same slabs for the recipe sizes (NB = 10 / 11 / 12 feature blocks) and minibatch-sized K.
//
// The 64 x 64 form above pads M = 16 NB to 192 (26 % of its MFMAs multiply padding at D = 150) and lands on the chip as
// 495 blocks, two per CU on most CUs and one on the rest.  Here a wave owns ALL M columns of a 32-column strip of the
// output: per k4-step it loads 4 rows x M of A as three coalesced loads per lane — columns 4 i .. 4 i + 3, 64 + 4 i ..,
// and 128 + CL i .. (CL = NB - 8 = 2, 3 or 4 interleaved 16-blocks, so 160 / 176 / 192 columns are covered exactly) —
// and 4 rows x 32 of B as one 8-byte load, for 2 NB MFMAs: no padding, and (tiles of 32 columns) x (k-groups) lands as
// ONE 8-wave block per CU (D = 150: 21 tiles x 12 k-groups = 252 blocks), two waves per SIMD, each SIMD carrying the
// same number of MFMAs.  The 8 waves of a block split the k-group's rows and reduce their accumulators through LDS in a
// fixed order, so a k-group still leaves ONE slab.  K and the x1 / x2 row split must be multiples of 4 (a k4-step never
// straddles the two row segments): the host falls back to the form above otherwise.
// The per-block dQ / dP sums of K-A (pair sums) are folded by all blocks, 2 Mp / tiles columns each.
// ------------------------------------------------------------------------------------------------
typedef float f32x2 __attribute__((ext_vector_type(2)));

constexpr int kFmWaves = 8;
constexpr int kFmPF = 4;  // operand register sets per wave: the one in use, one being refilled, two on their way

struct WgradFmArgs {
    WgradArgs w;
    int nt0, nt1;     // 32-column tiles of the two problems
    int ps_cols;      // pair-sum columns per block (of 2 Mp), 0 = none
    int xcd_groups;   // 1: k-groups are laid out XCD by XCD (streaming sizes; no tail block)
    int has_tail;     // one more block after the GEMM blocks: the loss / threshold tail of the training step
    LossTail tail;
};

// CL consecutive floats in CL registers
template <int CL> struct FmVec;
template <> struct FmVec<2> { typedef float type __attribute__((ext_vector_type(2))); };
template <> struct FmVec<3> { typedef float type __attribute__((ext_vector_type(3))); };
template <> struct FmVec<4> { typedef float type __attribute__((ext_vector_type(4))); };

// NSB: 16-column blocks of the strip a wave owns: 2 (32 columns, minibatch sizes: 21 tiles x 12 k-groups fill the chip) or 4
// (64 columns, streaming sizes: 40 MFMAs per four loads instead of 20, and the A rows cross L2 -> CU 11 times instead of 21).
template <int NSB> struct FmRounds { static constexpr int RB(int NB) { return NSB == 2 ? (NB + 1) / 2 : (NB + 3) / 4; } };

template <int NB, bool EXT, int PF = kFmPF, int NSB = 2, bool BBF = false>
__device__ __forceinline__ void wgrad_fm_body(const WgradFmArgs& fa, const WgradProblem P, int tile_all, int nt, int ks,
                                              f32x4 (*red)[FmRounds<NSB>::RB(NB) * NSB][64], f32x4 (*rede)[3][16],
                                              float* psum) {
    static_assert(NSB == 2 || NSB == 4, "strips of 32 or 64 columns");
    constexpr int NW = kFmWaves;
    constexpr int CL = NB - 8;
    constexpr int RB = FmRounds<NSB>::RB(NB);
    constexpr int NRND = (NB + RB - 1) / RB;
    const WgradArgs& a = fa.w;
    const int lane = threadIdx.x & 63, wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int i16 = lane & 15, g4 = lane >> 4;
    const long long lda = P.lda, ldb = P.ldb, nsplit = a.nsplit;
    const int n0 = nt * (16 * NSB);
    const bool nval = n0 + NSB * i16 < P.N;

    NPLDA_FM_STAMP(0);
    // ---- the block's share of the pair sums: one (column, block chunk) per thread, loaded now, summed at the end ----
    float psv = 0.f;
    const int psc = fa.ps_cols;
    int ps_col = -1;
    if (psc > 0) {
        const int c = threadIdx.x % psc, jn = (NW * 64) / psc, jj = threadIdx.x / psc;
        const int col = tile_all * psc + c;
        if (jj < jn) ps_col = c;  // (columns past 2 Mp and empty chunks leave zeros)
        const int per_b = (a.nblk + a.ksplit - 1) / a.ksplit;
        const int b0 = ks * per_b, b1 = b0 + per_b < a.nblk ? b0 + per_b : a.nblk;
        const int chunk = (per_b + jn - 1) / jn;
        if (jj < jn && col < 2 * a.Mp) {
            const int row = col / a.Mp, f = col - row * a.Mp;
            const int c0 = b0 + jj * chunk, c1 = c0 + chunk < b1 ? c0 + chunk : b1;
            for (int b = c0; b < c1; ++b) psv += a.pq[((size_t)b * 2 + row) * a.ldz + f];
        }
    }

    // ---- k-range of the wave, in units of 4 rows ---------------------------------------------------------------------
    const long long U = a.K >> 2;
    const long long part = (long long)ks * NW + wave, nparts = (long long)a.ksplit * NW;
    const long long u0 = U * part / nparts, u1 = U * (part + 1) / nparts;

    // ---- operand ring: PF register sets, loads and waits written by hand --------------------------------------------
    // hipcc gives a register ring that is carried around a loop `s_waitcnt vmcnt(0)` at the loop head: every load of a
    // round is waited for before its first MFMA, and the two waves of a SIMD fall into lockstep (each slows the other's
    // MFMA phase down until both reach the load phase together).  Here the loads are inline assembly, invisible to the
    // wait-count pass, and the waits are explicit (vmcnt counts in order): at step j the set j % PF is waited for, and
    // between the step's MFMAs — in-order issue: what follows an MFMA runs in its 32-cycle shadow — the set of
    // step j - 1 is refilled for the unit PF - 1 steps ahead.  One wave keeps the matrix pipe busy by itself; the second
    // wave of the SIMD covers what is left.  (Staging the operands through LDS by DMA, global_load ... lds, with the same
    // explicit waits was measured too: 27.6 us instead of 24.9 for the plain register ring at B = 4096 — six DMA
    // instructions per step, four of them 4 bytes per lane, cost more in the vector memory pipe than they save.)
    typedef typename FmVec<CL>::type fvCL;
    typedef typename FmVec<NSB>::type fvB;
    static_assert(!BBF || NSB == 2, "bf16 B rows: 32-column strips");
    struct Ops { f32x4 a0, a1; fvCL a2; fvB b; };  // (BBF: b[0] holds the lane's two bf16 values as loaded, b[1] is unused)
    constexpr int NLD = 4;                                                 // loads per unit
    static_assert((PF - 2) * NLD < 64, "vmcnt immediate (6 bits on gfx9+)");
    unsigned offA = (unsigned)((g4 * lda + 4 * i16) * 4);
    unsigned offA2 = (unsigned)((g4 * lda + 128 + CL * i16) * 4);
    constexpr int BE = BBF ? 2 : 4;  // bytes per B element
    unsigned offB = (unsigned)((g4 * ldb + (nval ? n0 + NSB * i16 : 0)) * BE);
    const char* PA = reinterpret_cast<const char*>(P.A);
    const char* PB0 = reinterpret_cast<const char*>(P.B0);
    const char* PB1 = reinterpret_cast<const char*>(P.B1) - nsplit * ldb * BE;  // row r >= nsplit: PB1 + r ldb
    // the four loads of unit u (4 rows) into a register set, one at a time: piece 0, 1 = A columns 0..63, 64..127,
    // 2 = the last A group, 3 = B
    auto load1 = [&](Ops& o, long long u, int piece) {
        const long long row = u << 2;
        const char* ab = PA + row * lda * 4;
        const char* bb = (row < nsplit ? PB0 : PB1) + row * ldb * BE;
        if (piece == 0) asm volatile("global_load_dwordx4 %0, %1, %2" : "=v"(o.a0) : "v"(offA), "s"(ab) : "memory");
        if (piece == 1) asm volatile("global_load_dwordx4 %0, %1, %2 offset:256" : "=v"(o.a1) : "v"(offA), "s"(ab) : "memory");
        if (piece == 2) {
            if constexpr (CL == 2) asm volatile("global_load_dwordx2 %0, %1, %2" : "=v"(o.a2) : "v"(offA2), "s"(ab) : "memory");
            else if constexpr (CL == 3) asm volatile("global_load_dwordx3 %0, %1, %2" : "=v"(o.a2) : "v"(offA2), "s"(ab) : "memory");
            else asm volatile("global_load_dwordx4 %0, %1, %2" : "=v"(o.a2) : "v"(offA2), "s"(ab) : "memory");
        }
        if (piece == 3) {
            if constexpr (BBF) asm volatile("global_load_dword %0, %1, %2" : "=v"(o.b[0]) : "v"(offB), "s"(bb) : "memory");
            else if constexpr (NSB == 2) asm volatile("global_load_dwordx2 %0, %1, %2" : "=v"(o.b) : "v"(offB), "s"(bb) : "memory");
            else asm volatile("global_load_dwordx4 %0, %1, %2" : "=v"(o.b) : "v"(offB), "s"(bb) : "memory");
        }
    };
    auto load = [&](Ops& o, long long u) {
#pragma unroll
        for (int piece = 0; piece < 4; ++piece) load1(o, u, piece);
    };
    auto wait = [&](Ops& o) {  // every load older than the last PF - 2 units has landed: this set is valid
        asm volatile("s_waitcnt vmcnt(%4)" : "+v"(o.a0), "+v"(o.a1), "+v"(o.a2), "+v"(o.b) : "n"((PF - 2) * NLD) : "memory");
    };

    f32x4 acc[NB][NSB];
#pragma unroll
    for (int mb = 0; mb < NB; ++mb)
#pragma unroll
        for (int cb = 0; cb < NSB; ++cb) acc[mb][cb] = f32x4{0.f, 0.f, 0.f, 0.f};
    f32x4 e0 = {0.f, 0.f, 0.f, 0.f}, e1 = e0;
    float e2[CL];
#pragma unroll
    for (int c = 0; c < CL; ++c) e2[c] = 0.f;
    auto mfmas = [&](const Ops& o, const fvB& bv, int mb0, int mb1) {
#pragma unroll
        for (int mb = 0; mb < NB; ++mb) {
            if (mb >= mb0 && mb < mb1) {
                const float av = mb < 4 ? o.a0[mb & 3] : (mb < 8 ? o.a1[mb & 3] : o.a2[mb >= 8 ? mb - 8 : 0]);
#pragma unroll
                for (int cb = 0; cb < NSB; ++cb)
                    acc[mb][cb] = __builtin_amdgcn_mfma_f32_16x16x4f32(av, bv[cb], acc[mb][cb], 0, 0, 0);
            }
        }
    };

    NPLDA_FM_STAMP(1);
    if (u0 < u1) {
        const long long ulast = u1 - 1;
        auto unit = [&](long long u) { return u < ulast ? u : ulast; };  // (past the range: a repeat, never used)
        Ops ring[PF];
#pragma unroll
        for (int s = 0; s < PF - 1; ++s) load(ring[s], unit(u0 + s));
        for (long long u = u0; u < u1; u += PF) {
#pragma unroll
            for (int s = 0; s < PF; ++s) {
                if (u + s < u1) {
                    Ops& o = ring[s];
                    wait(o);
                    fvB bv;
#pragma unroll
                    for (int cb = 0; cb < NSB; ++cb) bv[cb] = nval ? o.b[cb] : 0.f;
                    if constexpr (BBF) {  // two bf16 columns in one dword: widen (low half = the first column)
                        const unsigned raw = __float_as_uint(o.b[0]);
                        bv[0] = nval ? __uint_as_float(raw << 16) : 0.f;
                        bv[1] = nval ? __uint_as_float(raw & 0xffff0000u) : 0.f;
                    }
                    // the set of the previous step is refilled one load at a time between groups of MFMAs (all four in one
                    // place: 2-3 % slower)
                    Ops& n = ring[(s + PF - 1) % PF];
                    const long long un = unit(u + s + PF - 1);
                    constexpr int cut[5] = {0, 2, 5, 7, 9};
#pragma unroll
                    for (int piece = 0; piece < 4; ++piece) {
                        mfmas(o, bv, cut[piece], cut[piece + 1]);
                        __builtin_amdgcn_sched_barrier(0);
                        load1(n, un, piece);
                        __builtin_amdgcn_sched_barrier(0);
                    }
                    mfmas(o, bv, cut[4], NB);
                    if (EXT) {
                        e0 += o.a0;
                        e1 += o.a1;
#pragma unroll
                        for (int c = 0; c < CL; ++c) e2[c] += o.a2[c];
                    }
                    __builtin_amdgcn_sched_barrier(0);
                }
            }
        }
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");  // the trailing repeats have landed (their registers are dead)
    }
    NPLDA_FM_STAMP(2);

    // ---- in-block reduction over the 8 k-parts (fixed order), two rounds of RB m-blocks -------------------------------
    if (EXT) {
#pragma unroll
        for (int c = 0; c < 4; ++c) {
            e0[c] = wave_xor_add(e0[c], 16); e0[c] = wave_xor_add(e0[c], 32);
            e1[c] = wave_xor_add(e1[c], 16); e1[c] = wave_xor_add(e1[c], 32);
        }
        f32x4 e2v = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int c = 0; c < CL; ++c) {
            e2[c] = wave_xor_add(e2[c], 16); e2[c] = wave_xor_add(e2[c], 32);
            e2v[c] = e2[c];
        }
        if (g4 == 0) {
            rede[wave][0][i16] = e0;
            rede[wave][1][i16] = e1;
            rede[wave][2][i16] = e2v;
        }
    }
    if (ps_col >= 0) psum[(threadIdx.x / psc) * psc + ps_col] = psv;
    float* slab = P.slab + (size_t)ks * P.Mp * P.Np;
#pragma unroll
    for (int rnd = 0; rnd < NRND; ++rnd) {
        const int mb0 = rnd * RB;
        const int cnt = (NB - mb0) < RB ? (NB - mb0) : RB;
        if (rnd) __syncthreads();  // the previous round's sums have been read
#pragma unroll
        for (int mb = 0; mb < NB; ++mb) {
            if (mb >= mb0 && mb < mb0 + cnt) {
#pragma unroll
                for (int cb = 0; cb < NSB; ++cb) red[wave][(mb - mb0) * NSB + cb][lane] = acc[mb][cb];
            }
        }
        __syncthreads();
        if (rnd < 2) NPLDA_FM_STAMP(5 + 2 * rnd);
        if (wave < cnt) {  // wave w finishes m-block mb0 + w of the round
            const int mb = mb0 + wave;
            f32x4 sum[NSB];
#pragma unroll
            for (int cb = 0; cb < NSB; ++cb) {
                const int idx = wave * NSB + cb;
                f32x4 v = red[0][idx][lane];
#pragma unroll
                for (int ww = 1; ww < NW; ++ww) v += red[ww][idx][lane];
                sum[cb] = v;
            }
            if (nval && n0 + NSB * i16 < P.Np) {
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    const int i = 4 * g4 + r;  // row of the MFMA block
                    const int m = mb < 8 ? 64 * (mb >> 2) + 4 * i + (mb & 3) : 128 + CL * i + (mb - 8);
                    fvB v;
#pragma unroll
                    for (int cb = 0; cb < NSB; ++cb) v[cb] = sum[cb][r];
                    *reinterpret_cast<fvB*>(slab + (size_t)m * P.Np + n0 + NSB * i16) = v;
                }
            }
        }
        if (rnd < 2) NPLDA_FM_STAMP(6 + 2 * rnd);
    }
    NPLDA_FM_STAMP(3);
    // ---- column sums of A (db1 / db2), pair sums -------------------------------------------------------------------------
    float* eb = a.ext + (size_t)ks * 4 * a.Mp;
    if (EXT && wave == NW - 1 && g4 == 0) {
        f32x4 t0 = rede[0][0][i16], t1 = rede[0][1][i16], t2 = rede[0][2][i16];
#pragma unroll
        for (int ww = 1; ww < NW; ++ww) {
            t0 += rede[ww][0][i16];
            t1 += rede[ww][1][i16];
            t2 += rede[ww][2][i16];
        }
        float* er = eb + (P.extras == 1 ? 3 : 2) * a.Mp;
        *reinterpret_cast<f32x4*>(er + 4 * i16) = t0;
        *reinterpret_cast<f32x4*>(er + 64 + 4 * i16) = t1;
#pragma unroll
        for (int c = 0; c < CL; ++c) er[128 + CL * i16 + c] = t2[c];
    }
    if (EXT && P.extras == 3) {  // embedding rows: no pair term, dQ = dP = 0
        for (int i = threadIdx.x; i < 2 * a.Mp; i += NW * 64) eb[i] = 0.f;
    }
    if (psc > 0 && threadIdx.x < psc) {
        const int col = tile_all * psc + threadIdx.x;
        if (col < 2 * a.Mp) {
            const int jn = (NW * 64) / psc;
            float v = 0.f;
            for (int jj = 0; jj < jn; ++jj) v += psum[jj * psc + threadIdx.x];
            eb[col] = v;  // rows 0 (dQ) and 1 (dP) are adjacent: [2][Mp]
        }
    }
    NPLDA_FM_STAMP(4);
}

// PF: operand register sets per wave.  4 at minibatch sizes (everything comes out of L2); streaming sizes (K > 32 768 rows: the
// B rows come from HBM) take a deeper ring.
template <int NB, int PF = kFmPF, int NSB = 2, bool BBF = false>
__global__ __launch_bounds__(kFmWaves * 64, 1) void wgrad_fm_kernel(const WgradFmArgs fa) {
    __shared__ f32x4 red[kFmWaves][FmRounds<NSB>::RB(NB) * NSB][64];
    __shared__ f32x4 rede[kFmWaves][3][16];
    __shared__ float psum[kFmWaves * 64];
    const int w = blockIdx.x;
    if (fa.has_tail && w == (fa.nt0 + fa.nt1) * fa.w.ksplit) {  // on a CU of its own, done long before the GEMM blocks
        static_assert(kLossTailSmem <= sizeof(red), "the tail's LDS fits in the reduction buffer");
        loss_tail_block(fa.tail, red);
        return;
    }
    int ks, tile;
    if (fa.xcd_groups) {
        // Streaming sizes: the strips of one k-group read the same A rows (640 B each) and must share them through ONE L2.
        // Workgroup w runs on XCD w % 8 (each with its own 4 MB L2): with k-groups laid out as w % ksplit the 11 strips of
        // a k-group sat on 8 different XCDs and every A row crossed HBM -> L2 eleven times (TCC hit rate 0.3 %, 5.2 GB per
        // launch at 262 144 pairs: the kernel ran at the HBM limit, not the MFMA one).  Here XCD x takes whole k-groups
        // while its share of the grid lasts; what is left over of every XCD's share is strung together, XCD by XCD, into the
        // remaining k-groups (each then spans two XCDs at most).
        const int T = fa.nt0 + fa.nt1, G = T * fa.w.ksplit;
        const int x = w & 7, slot = w >> 3;
        int base = 0, lbase = 0, whole_all = 0;
        for (int xx = 0; xx < 8; ++xx) {
            const int n = (G - xx + 7) >> 3, wh = n / T;
            if (xx < x) {
                base += wh;
                lbase += n - wh * T;
            }
            whole_all += wh;
        }
        const int n = (G - x + 7) >> 3, wh = n / T;
        if (slot < wh * T) {
            ks = base + slot / T;
            tile = slot % T;
        } else {
            const int l = lbase + (slot - wh * T);
            ks = whole_all + l / T;
            tile = l % T;
        }
    } else {
        ks = w % fa.w.ksplit;
        tile = w / fa.w.ksplit;
    }
    const int pi = tile >= fa.nt0 ? 1 : 0;
    const WgradProblem P = pi ? fa.w.p[1] : fa.w.p[0];
    const int nt = pi ? tile - fa.nt0 : tile;
    if constexpr (BBF) {  // (problem 0's B rows are bf16: its strips take the widening body)
        if (pi == 0) {
            if (nt == 0 && P.extras) wgrad_fm_body<NB, true, PF, NSB, true>(fa, P, tile, nt, ks, red, rede, psum);
            else wgrad_fm_body<NB, false, PF, NSB, true>(fa, P, tile, nt, ks, red, rede, psum);
            return;
        }
    }
    if (nt == 0 && P.extras) wgrad_fm_body<NB, true, PF, NSB>(fa, P, tile, nt, ks, red, rede, psum);
    else wgrad_fm_body<NB, false, PF, NSB>(fa, P, tile, nt, ks, red, rede, psum);
}

}  // namespace nplda

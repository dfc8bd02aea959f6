// nplda_cohort.hip — adaptive score normalisation on the device (gfx950).
//
// The reference's utils/adaptive_score_normalization.py:27-36 READS cohort scores from a TSV that Kaldi
// wrote, sorts every row on the host and loops over trials in Python (:65-73).  Here the cohort scores
// are COMPUTED (new functionality; oracle = NeuralPlda.forward on the expanded pair list) and reduced:
//
//  K8a cohort_gemm_kernel   S[r, m] = q_r + q_m + 2 sum_d P_d z_r,d z_m,d as an fp32-MFMA tile GEMM
//                           (K = padded D2 <= 192, so it is MFMA/write-bound, AI ~ 80 FLOP/B): block tile
//                           128 x 128 (2 x 2 waves of 64 x 64), both operands staged through LDS by LDS-DMA in
//                           swizzled k16 stages, persistent blocks pulling XCD-banded tiles from per-XCD
//                           counters, the 2 P_d factor folded into the row operand, q_r + q_m added in the
//                           epilogue (16-byte stores).  S is spilled (R_chunk x M fp32; 0.88 GB for the whole
//                           22 k x 10 k of BASELINE cfg3, far below 288 GB) because the per-row top-N needs
//                           whole rows.
//  K8b row_stats_kernel     one workgroup per row: the row is loaded once into LDS (<= 160 KB) as
//                           order-preserving integer keys; sum and sum of squares in fp64.  The N-th smallest
//                           (reference semantics: ascending sort then [:N], adaptive_score_normalization.py:32-36)
//                           or N-th largest key: a normal-quantile bracket of the row's (mean, std) is counted and
//                           its <= 512 keys ranked exactly (one pass over the keys; a miss refines the bracket from its counts, up to five
//                           passes); rows that still miss take
//                           a 4-ary search on the integer key space (three pivot counts per pass in registers,
//                           DPP + LDS reduction, no atomics) and a last pass over the selected values.  Either way
//                           ties are resolved by count, so the result equals sort-then-slice exactly.
//                           Output (mean, std, mean_top, std_top), population std (ddof = 0), fp64.
//  K9  asnorm_apply_kernel  per trial: z-norm, t-norm, s-norm, as-norm1 from the two rows' statistics
//                           (adaptive_score_normalization.py:65-73), fp64, ~100 B/trial: HBM-bound.
#include "nplda_common.h"
#include "nplda_cohort_common.h"
#include "nplda_cohort_fused.h"

namespace {

// ------------------------------------------------------------------------------------------------
// K8a
// ------------------------------------------------------------------------------------------------
struct CohortGemmArgs {
    const float* zr;   // (R, ldz)
    const float* qr;   // (R)
    const float* zc;   // (M, ldz)
    const float* qc;   // (M)
    const float* P;    // padded (>= 16 * NBK) floats, zero beyond D2
    long long R, M, ldz, lds;  // lds = row stride of S (>= M)
    int ksteps;        // k16-steps = padded D2 / 16
    int nxp;           // column tiles per XCD band = min(ceil(ceil(M / 128) / 8), 24)
    int ny;            // row tiles = ceil(R / 128)
    int nx, nsb;       // column tiles = ceil(M / 128); super-bands of 8 bands
    unsigned* ctr;     // 8 tile counters (one per XCD), zero at launch
    float* S;
};

__global__ __launch_bounds__(256, 2) void cohort_gemm_kernel(const CohortGemmArgs a) {
    // Block tile 128 x 128 = 2 x 2 waves of 64 x 64, K = 16 * ksteps.
    //
    // Operands.  Both go through LDS in k16 stages (128 rows x 64 B each, two buffers, 32 KB).  Reading the MFMA
    // fragments straight from the row-major tables (lane i16 + 16 g4 <- row i16, 16-byte chunk g4) hands the texture
    // addresser 64 different 16-byte pieces per wave instruction in lane order, and with 8 such loads per 64 MFMAs the
    // address path, not the matrix pipe, set the pace.  Staged, four consecutive lanes fetch one row's 64 contiguous
    // bytes, each element is fetched once per block instead of once per wave, and the fragments come from LDS by
    // ds_read_b128.  The stages are filled by LDS-DMA (global_load_lds_dwordx4: the destination is a wave-uniform LDS
    // base + 16 * lane, so one wave instruction fills a 1 KB piece = 16 rows, and the swizzle below is applied to the
    // SOURCE address): no staging registers, no ds_write pass, and the wait for the data sits at the barrier that ends
    // the stage, a whole stage of MFMAs after the issue.  A stage is 8 + 8 pieces; wave w fills pieces 2 w, 2 w + 1
    // of both operands.
    //
    // LDS image of a stage: row-major, 4 chunks of 16 B per row, chunk c of row r stored at chunk c ^ ((r >> 2) & 2).
    // ds_read_b128 serves the lane groups {0-3, 12-15, 20-27}, {4-11, 16-19, 28-31} (and the same + 32): rows 0-3 and 12-15
    // of chunk g4 together with rows 4-11 of chunk g4 + 1; unswizzled, rows r and r + 4 share their four banks, with
    // the swizzle the 16 lanes of a group cover the 16 slots of the 256-byte bank row exactly once (SQ_LDS_BANK_CONFLICT
    // = 0 measured).
    //
    // Tiles.  Persistent blocks: identical tiles keep the blocks that share a CU in lockstep, so with one tile per
    // block their prologues (first stage in flight, nothing to multiply), epilogues and the relaunch all fell
    // together.  Here a block walks tiles and, during the LAST stage of a tile, already fetches the first stage and the
    // self terms of the next one; the stores of a tile drain under the next tile's MFMAs.  Tiles are handed out by a
    // per-XCD counter (one returning atomic per tile, requested a tile ahead): the matrix pipe serves the blocks of a
    // CU unevenly (lifetimes of 580 ... 730 us were measured for equal static shares), and with static shares the
    // kernel ended with one block per CU finishing alone.
    //
    // XCD-aware tile order: workgroup b is dispatched to XCD b % 8 (observed; speed only, nothing depends on it), so
    // XCD x owns the band of nxp consecutive column tiles [x nxp, (x + 1) nxp) for every row tile.  Its slice of the
    // cohort table (nxp * 128 rows, < 1 MB at cfg3) then stays in that XCD's 4 MB L2 while the row table streams
    // through once per XCD; with a plain 2-D grid every XCD cycled through the whole cohort table (7 MB) once per
    // wave of resident blocks.  Bands are capped at 24 tiles (2.3 MB of a 192-wide table); a wider cohort is covered
    // by several super-bands of 8 bands, one after the other.  The grid is a multiple of 8, so a block keeps its XCD.
    //
    // ONE __shared__ object: with a second one hipcc drains the memory counter before the first ds_read of every stage.
    __shared__ f32x4 smem[2 * 2 * 512 + 48 + 128 + 1];
    f32x4 (*tile)[2][512] = reinterpret_cast<f32x4 (*)[2][512]>(smem);
    f32x4* p2s = smem + 2048;                             // 2 P_d as fragment-shaped float4, NB <= 12 k16-steps
    float* qs = reinterpret_cast<float*>(smem + 2096);    // self terms [tile parity][q_r 128 | q_m 128]
    const int tid = threadIdx.x;
    const int lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int i16 = lane & 15, g4 = lane >> 4;
    const int xcd = blockIdx.x & 7;
    // tile `slot` of this XCD: super-band by super-band, row tile by row tile, the band's column tiles innermost
    auto decode = [&](int slot, long long& rb, long long& mb) {
        for (int sb = 0; sb < a.nsb; ++sb) {
            const int tx0 = (sb * 8 + xcd) * a.nxp;
            int w = a.nx - tx0;
            if (w > a.nxp) w = a.nxp;
            if (w <= 0) break;
            if (slot < w * a.ny) {
                const int ty = slot / w;
                rb = (long long)ty * 128;
                mb = (long long)(tx0 + slot - ty * w) * 128;
                return true;
            }
            slot -= w * a.ny;
        }
        return false;
    };

    // staging: lane l of piece pc fetches row 16 pc + (l >> 2), chunk (l & 3) ^ ((l >> 4) & 2)
    const int srow = lane >> 2, sq = (lane & 3) ^ ((lane >> 4) & 2);
    const float* ga[2];
    const float* gb[2];
    auto set_ptrs = [&](long long rb, long long mb) {
#pragma unroll
        for (int i = 0; i < 2; ++i) {
            const int row = 16 * (2 * wave + i) + srow;
            long long r = rb + row, m = mb + row;
            if (r >= a.R) r = a.R - 1;
            if (m >= a.M) m = a.M - 1;
            ga[i] = a.zr + r * a.ldz + 4 * sq;
            gb[i] = a.zc + m * a.ldz + 4 * sq;
        }
    };
    auto stage_in = [&](int ks, int buf) {
#pragma unroll
        for (int i = 0; i < 2; ++i) {
            __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(ga[i] + 16 * ks),
                                             (__attribute__((address_space(3))) void*)&tile[buf][0][64 * (2 * wave + i)],
                                             16, 0, 0);
            __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(gb[i] + 16 * ks),
                                             (__attribute__((address_space(3))) void*)&tile[buf][1][64 * (2 * wave + i)],
                                             16, 0, 0);
        }
    };
    auto q_in = [&](long long rb, long long mb, int par) {  // waves 0, 1: q_r of the 128 rows; waves 2, 3: q_m of the 128 columns
        const bool isr = wave < 2;
        long long j = (isr ? rb : mb) + 64 * (wave & 1) + lane;
        const long long lim = isr ? a.R : a.M;
        if (j >= lim) j = lim - 1;
        __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)((isr ? a.qr : a.qc) + j),
                                         (__attribute__((address_space(3))) void*)&qs[par * 256 + 64 * wave], 4, 0, 0);
    };

    long long rb, mb, nrb = 0, nmb = 0;
    unsigned* nxt_s = reinterpret_cast<unsigned*>(smem + 2224);  // the next tile's slot, wave 0 -> everybody
    // every tile comes from the counter, the first one included: a block that only becomes resident once the others
    // have drained the counter (should the occupancy query ever over-count) finds nothing and leaves
    if (tid == 0) *nxt_s = atomicAdd(a.ctr + xcd, 1u);
    __syncthreads();
    if (!decode(__builtin_amdgcn_readfirstlane((int)*nxt_s), rb, mb)) return;  // whole block (uniform)
    if (tid < 4 * a.ksteps) p2s[tid] = 2.0f * *reinterpret_cast<const f32x4*>(a.P + 4 * tid);
    // fragment reads: row 64 (wave half) + 16 c + i16, chunk g4
    const int fo = i16 * 4 + (g4 ^ ((i16 >> 2) & 2));
    const f32x4* fra = &tile[0][0][(wave >> 1) * 256 + fo];
    const f32x4* frb = &tile[0][1][(wave & 1) * 256 + fo];

    set_ptrs(rb, mb);
    stage_in(0, 0);
    q_in(rb, mb, 0);
    __syncthreads();  // hipcc drains the memory counter (the DMA is a pending LDS write) before the barrier
    int gpar = 0, qpar = 0;

    for (;;) {
        // ask for the tile after this one; the answer is published at the end of the first stage, used in the last
        unsigned pend = 0;
        if (tid == 0) pend = atomicAdd(a.ctr + xcd, 1u);
        if (a.ksteps == 1) {
            if (tid == 0) *nxt_s = pend;
            __syncthreads();
        }
        bool have_next = false;
        f32x4 acc[4][4];
#pragma unroll
        for (int ca = 0; ca < 4; ++ca)
#pragma unroll
            for (int cb = 0; cb < 4; ++cb) acc[ca][cb] = f32x4{0.f, 0.f, 0.f, 0.f};

        for (int ks = 0; ks < a.ksteps; ++ks) {
            const int cur = gpar;
            if (ks + 1 < a.ksteps) {
                stage_in(ks + 1, cur ^ 1);
            } else {  // last stage: the next tile's first stage and self terms
                const int nslot = __builtin_amdgcn_readfirstlane((int)*nxt_s);
                have_next = decode(nslot, nrb, nmb);
                if (have_next) {
                    set_ptrs(nrb, nmb);
                    stage_in(0, cur ^ 1);
                    q_in(nrb, nmb, qpar ^ 1);
                }
            }
            f32x4 fa[4], fb[4];
            const f32x4 pf = p2s[4 * ks + g4];
#pragma unroll
            for (int c = 0; c < 4; ++c) {
                fa[c] = fra[cur * 1024 + c * 64] * pf;  // the 2 P_d factor rides on the row operand
                fb[c] = frb[cur * 1024 + c * 64];
            }
            // the cohort fragment is the A operand, so the lane (i16, g4) of block (ca, cb) ends up with row
            // 16 ca + i16 and the FOUR CONSECUTIVE columns 16 cb + 4 g4 + r: 16-byte stores in the epilogue
#pragma unroll
            for (int kk = 0; kk < 4; ++kk)
#pragma unroll
                for (int ca = 0; ca < 4; ++ca)
#pragma unroll
                    for (int cb = 0; cb < 4; ++cb)
                        acc[ca][cb] = __builtin_amdgcn_mfma_f32_16x16x4f32(fb[cb][kk], fa[ca][kk], acc[ca][cb], 0, 0, 0);
            // end of stage: this wave's pieces of the next stage have landed (vmcnt), then everybody's (barrier).  Raw
            // builtins pinned behind the MFMAs: __syncthreads() may legally be hoisted above them (they touch no
            // memory), which puts the wait for the DMA right after its issue.
            __builtin_amdgcn_sched_barrier(0);
            __builtin_amdgcn_s_waitcnt(0x0F70);
            if (ks == 0 && a.ksteps > 1) {
                if (tid == 0) *nxt_s = pend;
                __builtin_amdgcn_s_waitcnt(0xC07F);  // lgkmcnt(0): the LDS store is done before the barrier releases
            }
            __builtin_amdgcn_s_barrier();
            gpar ^= 1;
        }

        // epilogue: S rows are padded to a multiple of 4 floats (lds), so a 16-byte store that starts below M stays in
        // its row.  (On gfx9 the stores sit on the same counter as the loads: nothing here may wait on it, or every
        // group of stores would pay a store round trip; the stage-end wait above keeps hipcc's bookkeeping clean.)
        const long long r0 = rb + (wave >> 1) * 64;
        const long long m0 = mb + (wave & 1) * 64;
        const float* qr_s = qs + qpar * 256 + (wave >> 1) * 64 + i16;
        const float* qm_s = qs + qpar * 256 + 128 + (wave & 1) * 64 + 4 * g4;
        f32x4 qmv[4];
#pragma unroll
        for (int cb = 0; cb < 4; ++cb) qmv[cb] = *reinterpret_cast<const f32x4*>(qm_s + 16 * cb);
#pragma unroll
        for (int ca = 0; ca < 4; ++ca) {
            const long long row = r0 + 16 * ca + i16;
            const float qrv = qr_s[16 * ca];
            if (row >= a.R) continue;
            float* srow = a.S + row * a.lds + m0 + 4 * g4;
#pragma unroll
            for (int cb = 0; cb < 4; ++cb)
                if (m0 + 16 * cb + 4 * g4 < a.M)
                    *reinterpret_cast<f32x4*>(srow + 16 * cb) = acc[ca][cb] + (qmv[cb] + qrv);
        }
        if (!have_next) break;
        // with a single stage per tile the self-term buffer of this parity is refilled during the very next stage
        if (a.ksteps == 1) __syncthreads();
        rb = nrb;
        mb = nmb;
        qpar ^= 1;
    }
}

// ------------------------------------------------------------------------------------------------
// K8b
// ------------------------------------------------------------------------------------------------
constexpr int kRowThreads = 512;
constexpr int kListCap = 512;       // candidate keys of the quantile shortcut
constexpr unsigned kUnwritten = 0xffffffffu;  // a rank slot nobody wrote (as a float: one particular NaN)
constexpr int kMaxRowLds = 38000;  // floats of one row kept in LDS (with the shortcut's lists: < 160 KiB)

__device__ __forceinline__ double block_sum_d(double v, double* red) {
    v = wave_sum_f64(v);
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    __syncthreads();
    if (lane == 0) red[wave] = v;
    __syncthreads();
    double s = 0.0;
#pragma unroll
    for (int w = 0; w < kRowThreads / 64; ++w) s += red[w];
    return s;
}

// two fp64 block sums with one pair of barriers
__device__ __forceinline__ void block_sum_d2(double& a, double& b, double* red) {
    a = wave_sum_f64(a);
    b = wave_sum_f64(b);
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    __syncthreads();
    if (lane == 0) { red[2 * wave] = a; red[2 * wave + 1] = b; }
    __syncthreads();
    a = b = 0.0;
#pragma unroll
    for (int w = 0; w < kRowThreads / 64; ++w) { a += red[2 * w]; b += red[2 * w + 1]; }
}

typedef unsigned u32x4 __attribute__((ext_vector_type(4)));

// The row is handled in 16-byte groups (S rows are 16-byte aligned and padded to a multiple of 4 floats): the keys of
// group g are keys4[g]; elements past M carry the key 0xffffffff, which no pivot reaches (pivots are capped at
// 0xfffffffe; only a NaN with the payload 0x7fffffff maps there, and a row holding one has NaN statistics anyway).
// Keys live in LDS when the row fits (use_lds), else they are rebuilt from global memory every pass.
__device__ __forceinline__ u32x4 row_keys(const f32x4 v, long long g, long long M, int lowest) {
    u32x4 k;
#pragma unroll
    for (int e = 0; e < 4; ++e) {
        unsigned x = f2key(v[e]);
        if (!lowest) x = ~x;  // N largest == N smallest of the reversed order
        k[e] = (4 * g + e < M) ? x : 0xffffffffu;
    }
    return k;
}

template <class F>
__device__ __forceinline__ void for_row_keys(int use_lds, const u32x4* keys4, const f32x4* src4, long long nvec,
                                             long long M, int lowest, F f) {
    if (use_lds) {
        for (long long g = threadIdx.x; g < nvec; g += kRowThreads) f(keys4[g]);
    } else {
        for (long long g = threadIdx.x; g < nvec; g += kRowThreads) f(row_keys(src4[g], g, M, lowest));
    }
}

// Statistics of ONE score row (block-wide; every thread of the block calls it, the call is uniform): out4 = (mean, std,
// mean_top, std_top).  Srow: 16-byte aligned, padded to a multiple of 4 floats.
__device__ __forceinline__ void row_stats_row(const float* __restrict__ Srow, long long M, int topn, int lowest,
                                              int use_lds, double* __restrict__ out4, unsigned char* smem_raw) {
    unsigned* keys = reinterpret_cast<unsigned*>(smem_raw);                  // [M] when use_lds
    u32x4* keys4 = reinterpret_cast<u32x4*>(smem_raw);
    unsigned* cnt = keys + (use_lds ? ((M + 3) / 4) * 4 : 0);                  // [2][8][4] count partials
    double* red = reinterpret_cast<double*>(cnt + 64);                         // [2][2][8]
    constexpr int NWV = kRowThreads / 64;

    const f32x4* src4 = reinterpret_cast<const f32x4*>(Srow);
    const long long nvec = (M + 3) / 4;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;

    // pass 0: load, full-row sums (the key range is only needed by the general search: computed there).  Batches of U independent 16-byte loads per thread (clamped index, so
    // that no load sits behind a branch): with one 4-byte load per iteration the compiler waited out every load before
    // issuing the next, ~20 memory round trips per row.
    double s1 = 0.0, s2 = 0.0;
    constexpr int U = 5;
    const int nv = (int)nvec, Mi = (int)M;  // M < 2^31 (checked on the host)
    const unsigned flip = lowest ? 0u : 0xffffffffu;
    for (int base = tid; base < nv; base += kRowThreads * U) {
        f32x4 v[U];
#pragma unroll
        for (int u = 0; u < U; ++u) {
            const int g = base + kRowThreads * u;
            v[u] = src4[g < nv ? g : nv - 1];
        }
#pragma unroll
        for (int u = 0; u < U; ++u) {
            const int g = base + kRowThreads * u;
            if (g < nv) {
                const int nvalid = Mi - 4 * g;  // >= 4 except in the row's last group
                u32x4 k;
                if (nvalid >= 4) {
#pragma unroll
                    for (int e = 0; e < 4; ++e) {
                        const unsigned x = f2key(v[u][e]) ^ flip;  // N largest == N smallest of the reversed order
                        k[e] = x;
                        const double d = (double)v[u][e];
                        s1 += d;
                        s2 += d * d;
                    }
                } else {
#pragma unroll
                    for (int e = 0; e < 4; ++e) {
                        const unsigned x = f2key(v[u][e]) ^ flip;
                        const bool ok = e < nvalid;
                        k[e] = ok ? x : 0xffffffffu;
                        const double d = ok ? (double)v[u][e] : 0.0;
                        s1 += d;
                        s2 += d * d;
                    }
                }
                if (use_lds) keys4[g] = k;
            }
        }
    }
    // Every exchange below is ONE barrier: the partial results of a phase go to LDS slots that no other phase uses
    // (cnt[0..31] / cnt[32..63], red[0..15] / red[16..31]), and the scratch of the shortcut (sentinel-filled candidate
    // list, "unwritten" rank slots, zeroed rank counters) is prepared here, two barriers ahead of its use.  (Measured:
    // halving the barrier count this way changed nothing by itself — the kernel is bound by its VALU instructions.)
    unsigned* list = cnt + 64 + 8 * NWV;          // [kListCap] candidate keys, then [kListCap] floats by rank
    float* sel = reinterpret_cast<float*>(list + kListCap);
    unsigned* nlist = reinterpret_cast<unsigned*>(sel + kListCap);
    unsigned* ctl = nlist + 4;                     // ka, kb, shortcut
    unsigned* rk = ctl + 8;                        // [kListCap / 2] shared rank counts
    s1 = wave_sum_f64(s1);
    s2 = wave_sum_f64(s2);
    if (lane == 0) {
        red[2 * wave] = s1;
        red[2 * wave + 1] = s2;
    }
    for (int i = tid; i < kListCap; i += kRowThreads) {
        list[i] = 0xffffffffu;                           // above every candidate
        reinterpret_cast<unsigned*>(sel)[i] = kUnwritten;  // rank slots: "not written" (see the tie rule below)
    }
    for (int i = tid; i < kListCap / 2; i += kRowThreads) rk[i] = 0u;
    if (tid == 0) *nlist = 0;
    __syncthreads();
    long long N = topn;
    if (N > M) N = M;
    if (N < 1) N = 1;
    const unsigned want = (unsigned)N;
    const double n = (double)M;

    // The scalar follow-up (totals, mean / variance, the quantile bracket: ~250 instructions, most of them fp64
    // divisions) is done by wave 0 alone and handed over through `ctl`: evaluated by all eight waves it was a sixth
    // of the kernel's VALU instructions, and the VALU is what bounds this kernel (SQ_ACTIVE_INST_VALU ~ 95 %).
    // ---- shortcut for well-behaved rows: bracket the N-th smallest key with the normal quantile of (mean, std), count
    // once, and if the bracket holds the rank and few enough keys, finish EXACTLY on that short list (rank by counting,
    // values written to their rank slot so that the fp64 sums are order-independent): 3 passes over the row instead of
    // ~16.  Anything else (heavy tails, ties, tiny rows) falls through to the general search below, started from the
    // bracket when it is valid.
    double mean = 0.0, var = 0.0;  // kept by wave 0 (thread 0 writes the result)
    // The bracket is proposed by wave 0 in "ordered" values w (w = v when the N smallest are wanted, w = -v for the N
    // largest: the key order) and refined, when a proposal misses, by interpolating the probit z = Phi^-1(count / M)
    // between the tightest points known below and above the target rank — the first proposal is the normal model of
    // the row (points at mean -+ 6 sd), every later one uses the row's own counts, so skewed / heavy-tailed / bimodal
    // rows converge in two or three passes instead of dropping to the 16-pass key search (3.7x slower, measured).
    float wl = 0.f, zl = 0.f, wu = 0.f, zu = 0.f, zt = 0.f, pa = 0.f, pb = 0.f, dens = 0.f;
    auto propose = [&](float center, float half) {  // wave 0: the next bracket [center - half, center + half] -> ctl
        pa = center - half;
        pb = center + half;
        unsigned ka0 = lowest ? f2key(pa) : ~f2key(-pa), kb0 = lowest ? f2key(pb) : ~f2key(-pb);
        if (kb0 > 0xfffffffeu) kb0 = 0xfffffffeu;
        if (ka0 > kb0) ka0 = kb0;
        if (lane == 0) { ctl[0] = ka0; ctl[1] = kb0; }
    };
    constexpr float kHalfCount = 64.f;  // candidates wanted on either side of the target rank
    if (wave == 0) {
        s1 = s2 = 0.0;
#pragma unroll
        for (int w = 0; w < NWV; ++w) {
            s1 += red[2 * w];
            s2 += red[2 * w + 1];
        }
        mean = s1 / n;
        var = s2 / n - mean * mean;
        if (var < 0.0) var = 0.0;
        unsigned ok0 = 0;
        if (use_lds && M >= 64 && var > 0.0) {
            const float sd = (float)sqrt(var), mw = (float)(lowest ? mean : -mean);
            const double q = (double)N / n;
            zt = fast_normcdfinv((float)(q < 1e-9 ? 1e-9 : (q > 1.0 - 1e-9 ? 1.0 - 1e-9 : q)));
            dens = (float)n * __expf(-0.5f * zt * zt) * 0.3989423f;  // keys per unit of z around the target
            wl = mw - 6.f * sd; zl = -6.f;  // provisional anchors: replaced by the row's extremes after a first miss
            wu = mw + 6.f * sd; zu = 6.f;
            float half = kHalfCount * sd / fmaxf(dens, 1e-3f);
            half = fminf(fmaxf(half, 1e-6f * sd), 3.f * sd);
            propose(mw + zt * sd, half);
            ok0 = 1;
        }
        if (lane == 0) ctl[2] = ok0;
    }
    __syncthreads();
    const bool shortcut = ctl[2] != 0;
    unsigned lo = 0u, hi = 0xfffffffeu;  // the N-th smallest key lies in [lo, hi]; have_* : that side comes from a count
    bool have_lo = false, have_hi = false;

    bool done = false;
    double t1 = 0.0, t2 = 0.0;
    constexpr int kAttempts = 5;
    for (int att = 0; shortcut && att < kAttempts; ++att) {
        const unsigned ka = ctl[0], kb = ctl[1];
        // ONE pass over the keys: count below / inside the bracket, sum everything strictly below it, and already
        // collect the bracket's keys (bounded by the list capacity); the counts then say whether the list is usable.
        unsigned ca = 0, cb2 = 0;
        t1 = t2 = 0.0;
        for_row_keys(1, keys4, src4, nvec, M, lowest, [&](const u32x4 k4) {
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                const unsigned k = k4[e];
                const bool below = k < ka, upto = k <= kb;
                ca += below;
                cb2 += upto;
                if (below) {
                    const double v = (double)key2f(lowest ? k : ~k);
                    t1 += v;
                    t2 += v * v;
                } else if (upto) {
                    const unsigned pos = atomicAdd(nlist, 1u);
                    if (pos < (unsigned)kListCap) list[pos] = k;
                }
            }
        });
        ca = wave_sum_u32(ca);
        cb2 = wave_sum_u32(cb2);
        if (lane == 0) { cnt[32 + wave * 4] = ca; cnt[32 + wave * 4 + 1] = cb2; }
        __syncthreads();
        ca = cb2 = 0;
#pragma unroll
        for (int w = 0; w < NWV; ++w) { ca += cnt[32 + w * 4]; cb2 += cnt[32 + w * 4 + 1]; }
        // what the two counts say about the key T of the N-th smallest element (kept for the general search below)
        if (ca < want) { if (!have_lo || ka > lo) lo = ka; have_lo = true; }
        else { if (!have_hi || ka - 1 < hi) hi = ka - 1; have_hi = true; }
        if (cb2 >= want) { if (!have_hi || kb < hi) hi = kb; have_hi = true; }
        else { if (!have_lo || kb + 1 > lo) lo = kb + 1; have_lo = true; }
        if (ca < want && want <= cb2 && cb2 - ca <= (unsigned)kListCap) {
            {
                const unsigned L = cb2 - ca, need = want - ca;  // == *nlist: every candidate found its slot
                const unsigned L4 = (L + 3) / 4 * 4;  // whole 16-byte groups: the list was sentinel-filled
                // rank by counting, the list read 16 bytes at a time; `split` threads share one candidate when the
                // candidates are few enough (the usual case: ~170 at N / M = 0.05 -> 2), their partial ranks meet in rk
                const unsigned sh = (L * 4 <= (unsigned)kRowThreads) ? 2u : (L * 2 <= (unsigned)kRowThreads) ? 1u : 0u;
                const unsigned split = 1u << sh;
                // A candidate goes to the rank slot `lt` = number of keys below it.  Ties share that slot and leave the
                // following ones unwritten; equal keys are equal values, so an unwritten slot simply takes the value of
                // the nearest written slot below it (slot 0 is always written).  Only `lt` is counted: half the compares.
                for (unsigned t = tid; t < L * split; t += kRowThreads) {
                    const unsigned i = t >> sh, part = t & (split - 1u);
                    const unsigned k = list[i];
                    const unsigned ng = L4 / 4;
                    const unsigned g0 = part * ng / split, g1 = (part + 1) * ng / split;
                    unsigned lt = 0;
                    const u32x4* list4 = reinterpret_cast<const u32x4*>(list);
                    for (unsigned gx = g0; gx < g1; ++gx) {
                        const u32x4 kj = list4[gx];
#pragma unroll
                        for (int e = 0; e < 4; ++e) lt += kj[e] < k;
                    }
                    if (sh) atomicAdd(rk + i, lt);
                    else sel[lt] = key2f(lowest ? k : ~k);
                }
                if (sh) {
                    __syncthreads();
                    if ((unsigned)tid < L) {
                        const unsigned k = list[tid];
                        sel[rk[tid]] = key2f(lowest ? k : ~k);
                    }
                }
                __syncthreads();
                double u1 = 0.0, u2 = 0.0;
                for (unsigned i = tid; i < need; i += kRowThreads) {  // rank slots: a fixed summation order
                    unsigned j = i;
                    while (reinterpret_cast<const unsigned*>(sel)[j] == kUnwritten) --j;
                    const double v = (double)sel[j];
                    u1 += v;
                    u2 += v * v;
                }
                t1 = wave_sum_f64(t1 + u1);
                t2 = wave_sum_f64(t2 + u2);
                if (lane == 0) { red[16 + 2 * wave] = t1; red[16 + 2 * wave + 1] = t2; }
                __syncthreads();
                if (tid == 0) {
                    t1 = t2 = 0.0;
#pragma unroll
                    for (int w = 0; w < NWV; ++w) { t1 += red[16 + 2 * w]; t2 += red[16 + 2 * w + 1]; }
                }
                done = true;
            }
            break;
        }
        if (att + 1 == kAttempts) break;
        // missed (or too many candidates): clean the list, refine the bracket from the counts, go again
        for (int i = tid; i < kListCap; i += kRowThreads) list[i] = 0xffffffffu;
        if (tid == 0) *nlist = 0;
        const bool off_edge = cb2 == 0 || ca >= (unsigned)M;  // the bracket fell entirely below / above the data
        if (off_edge) {  // find the row's extreme keys: the next bracket starts at that edge (hard-edged distributions)
            unsigned kmin = 0xffffffffu, kmax = 0u;
            for_row_keys(1, keys4, src4, nvec, M, lowest, [&](const u32x4 k) {
#pragma unroll
                for (int e = 0; e < 4; ++e) {
                    kmin = k[e] < kmin ? k[e] : kmin;
                    kmax = (k[e] != 0xffffffffu && k[e] > kmax) ? k[e] : kmax;
                }
            });
            kmin = wave_min_u32(kmin);
            kmax = wave_max_u32(kmax);
            if (lane == 0) { cnt[wave * 4] = kmin; cnt[wave * 4 + 1] = kmax; }
            __syncthreads();
        }
        if (wave == 0 && off_edge) {
            unsigned kmin = 0xffffffffu, kmax = 0u;
#pragma unroll
            for (int w = 0; w < NWV; ++w) {
                kmin = cnt[w * 4] < kmin ? cnt[w * 4] : kmin;
                kmax = cnt[w * 4 + 1] > kmax ? cnt[w * 4 + 1] : kmax;
            }
            const float zedge = fast_normcdfinv(0.5f / (float)n);
            const float hprev = 0.5f * (pb - pa);
            if (cb2 == 0) {  // below everything: same width, lower end at the smallest value
                wl = lowest ? key2f(kmin) : -key2f(~kmin);
                zl = zedge;
                propose(wl + hprev, hprev * 1.0001f + 1e-30f);
            } else {         // above everything
                wu = lowest ? key2f(kmax) : -key2f(~kmax);
                zu = -zedge;
                propose(wu - hprev, hprev * 1.0001f + 1e-30f);
            }
        } else if (wave == 0) {
            const float fa = fminf(fmaxf((float)ca / (float)n, 0.5f / (float)n), 1.f - 0.5f / (float)n);
            const float fb = fminf(fmaxf((float)cb2 / (float)n, 0.5f / (float)n), 1.f - 0.5f / (float)n);
            const float za = fast_normcdfinv(fa), zb = fast_normcdfinv(fb);
            // a point replaces an end of the interpolation interval only if it is tighter (a bracket that fell outside the
            // data carries a count of 0 or M and says nothing the extremes do not)
            if (ca < want) { if (pa > wl) { wl = pa; zl = za; } } else if (pa < wu) { wu = pa; zu = za; }
            if (cb2 >= want) { if (pb < wu) { wu = pb; zu = zb; } } else if (pb > wl) { wl = pb; zl = zb; }
            const float span = wu - wl;
            float center, half;
            const bool same_side = (ca >= want) || (cb2 < want);
            if (!same_side) {
                // the bracket holds the rank but too many keys: inside a bracket this narrow the density is flat —
                // interpolate the rank linearly and shrink to the wanted candidate count
                const float frac = ((float)(want - ca) - 0.5f) / (float)(cb2 - ca);
                center = pa + frac * (pb - pa);
                half = fmaxf((pb - pa) * (kHalfCount / (float)(cb2 - ca)), 1e-7f * fabsf(center));
            } else if (ca > 0 && cb2 < (unsigned)M && cb2 > ca + 8 && zb > za) {
                // both ends of the missed bracket carry counts: step from its nearer end with ITS slope (Newton-like)
                const float slope = (zb - za) / (pb - pa);
                const float from = ca >= want ? pa : pb, zfrom = ca >= want ? za : zb;
                center = fminf(fmaxf(from + (zt - zfrom) / slope, wl), wu);
                half = fminf(kHalfCount / fmaxf(dens * slope, 1e-30f), 0.5f * span);
            } else if (span > 0.f && zu > zl) {
                const float slope = (zu - zl) / span;  // dz / dw between the two points
                center = wl + (zt - zl) / slope;
                half = kHalfCount / fmaxf(dens * slope, 1e-30f);
                center = fminf(fmaxf(center, wl), wu);
                half = fminf(half, 0.5f * span);
            } else {
                center = 0.5f * (wl + wu);
                half = 0.25f * fabsf(span);
            }
            propose(center, fmaxf(half, 1e-7f * fabsf(center)));
        }
        __syncthreads();
    }
    if (done) {
        if (tid == 0) {
            const double nn = (double)N;
            const double mt = t1 / nn;
            double vt = t2 / nn - mt * mt;
            if (vt < 0.0) vt = 0.0;
            double* o = out4;
            o[0] = mean;
            o[1] = sqrt(var);
            o[2] = mt;
            o[3] = sqrt(vt);
        }
        return;
    }

    // N-th smallest key T by 4-ary search on the integer key space: every iteration counts, for three pivots,
    // the keys <= pivot (register counters + shuffle/LDS reduction — no atomics: cohort scores of one row share
    // their leading bits, which serialises an LDS-histogram radix select) and keeps the quarter that holds rank N.
    __syncthreads();  // the count slots are reused below
    if (!have_lo || !have_hi) {  // key range of the row (for the side no count has bounded)
        unsigned kmin = 0xffffffffu, kmax = 0u;
        for_row_keys(use_lds, keys4, src4, nvec, M, lowest, [&](const u32x4 k) {
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                if (k[e] != 0xffffffffu) {  // padding
                    kmin = k[e] < kmin ? k[e] : kmin;
                    kmax = k[e] > kmax ? k[e] : kmax;
                }
            }
        });
        kmin = wave_min_u32(kmin);
        kmax = wave_max_u32(kmax);
        if (lane == 0) { cnt[wave * 4] = kmin; cnt[wave * 4 + 1] = kmax; }
        __syncthreads();
        unsigned rlo = 0xffffffffu, rhi = 0u;
#pragma unroll
        for (int w = 0; w < NWV; ++w) {
            rlo = cnt[w * 4] < rlo ? cnt[w * 4] : rlo;
            rhi = cnt[w * 4 + 1] > rhi ? cnt[w * 4 + 1] : rhi;
        }
        if (rhi > 0xfffffffeu) rhi = 0xfffffffeu;
        if (rlo > rhi) rlo = rhi;
        if (!have_lo) lo = rlo;
        if (!have_hi) hi = rhi;
        if (lo > hi) lo = hi;
        __syncthreads();
    }
    for (int it = 0; lo < hi; ++it) {
        const unsigned long long span = (unsigned long long)hi - lo;
        const unsigned p1 = lo + (unsigned)(span / 4), p2 = lo + (unsigned)(span / 2), p3 = lo + (unsigned)(span / 4 * 3);
        unsigned c1 = 0, c2 = 0, c3 = 0;
        for_row_keys(use_lds, keys4, src4, nvec, M, lowest, [&](const u32x4 k) {
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                c1 += k[e] <= p1;
                c2 += k[e] <= p2;
                c3 += k[e] <= p3;
            }
        });
        c1 = wave_sum_u32(c1);
        c2 = wave_sum_u32(c2);
        c3 = wave_sum_u32(c3);
        unsigned* cb = cnt + (it & 1) * 32;
        if (lane == 0) { cb[wave * 4] = c1; cb[wave * 4 + 1] = c2; cb[wave * 4 + 2] = c3; }
        __syncthreads();
        c1 = c2 = c3 = 0;
#pragma unroll
        for (int w = 0; w < NWV; ++w) { c1 += cb[w * 4]; c2 += cb[w * 4 + 1]; c3 += cb[w * 4 + 2]; }
        if (c1 >= want) hi = p1;
        else if (c2 >= want) { lo = p1 + 1; hi = p2; }
        else if (c3 >= want) { lo = p2 + 1; hi = p3; }
        else lo = p3 + 1;
    }
    const unsigned tkey = lo;  // key of the N-th smallest element
    float tval = key2f(lowest ? tkey : ~tkey);

    // selected sums: everything strictly below tkey, plus (N - #less) copies of the threshold value (ties)
    double nless = 0.0;
    t1 = t2 = 0.0;
    for_row_keys(use_lds, keys4, src4, nvec, M, lowest, [&](const u32x4 k4) {
#pragma unroll
        for (int e = 0; e < 4; ++e) {
            const unsigned k = k4[e];
            if (k < tkey) {
                const double v = (double)key2f(lowest ? k : ~k);
                t1 += v;
                t2 += v * v;
                nless += 1.0;
            }
        }
    });
    block_sum_d2(t1, t2, red);
    nless = block_sum_d(nless, red);
    if (tid == 0) {
        const double ties_taken = (double)N - nless;
        t1 += ties_taken * (double)tval;
        t2 += ties_taken * (double)tval * (double)tval;
        const double nn = (double)N;
        const double mt = t1 / nn;
        double vt = t2 / nn - mt * mt;
        if (vt < 0.0) vt = 0.0;
        double* o = out4;
        o[0] = mean;
        o[1] = sqrt(var);
        o[2] = mt;
        o[3] = sqrt(vt);
    }
}

__global__ __launch_bounds__(kRowThreads) void row_stats_kernel(const float* __restrict__ S, long long lds_stride,
                                                                long long M, int topn, int lowest, int use_lds,
                                                                double* __restrict__ stats) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem_raw[];
    // newest rows first: in the cohort pipeline the GEMM has just written the matrix row tile by row tile, so the last rows
    // are the ones still held by the memory-side cache
    const long long row = (long long)gridDim.x - 1 - blockIdx.x;
    row_stats_row(S + row * lds_stride, M, topn, lowest, use_lds, stats + row * 4, smem_raw);
}

// Rows the fused path (nplda_cohort_fused.hip) could not finish — the threshold proposed from the row's analytic mean /
// std did not bracket the N-th smallest score (a far-from-normal row) — listed on the device.  Each block walks the
// list: it forms the row's M scores into a scratch row with the same operation order as the MFMA tiles
// (acc = fma chain over k ascending on (z_m, 2 P z_r), then + (q_m + q_r)) and runs the exact row statistics on it.
struct FallbackArgs {
    const float* zr; const float* qr; const float* zc; const float* qc; const float* P;
    long long M, ldz, lds;
    int kp;                      // padded embedding width (16 * ksteps)
    int topn, lowest, use_lds;
    const unsigned* fail_rows;   // row indices (relative to zr / stats)
    const unsigned* nfail;
    float* scratch;              // [gridDim.x][lds]
    double* stats;
};

__global__ __launch_bounds__(kRowThreads) void cohort_fallback_kernel(const FallbackArgs a) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem_raw[];
    __shared__ float arow[NPLDA_MAX_DIM];
    const unsigned n = *a.nfail;
    float* srow = a.scratch + (size_t)blockIdx.x * a.lds;
    for (unsigned i = blockIdx.x; i < n; i += gridDim.x) {
        const long long r = a.fail_rows[i];
        __syncthreads();
        for (int d = threadIdx.x; d < a.kp; d += kRowThreads) arow[d] = a.zr[r * a.ldz + d] * (2.0f * a.P[d]);
        __syncthreads();
        const float qrv = a.qr[r];
        for (long long m = threadIdx.x; m < a.lds; m += kRowThreads) {
            float s = 0.f;
            if (m < a.M) {
                // the order in which the MFMA tiles meet the features: k16-stage by stage, instruction kk of a stage
                // multiplies the features 16 ks + 4 g + kk, g = 0..3
                const float* zm = a.zc + m * a.ldz;
                float acc = 0.f;
                for (int ks = 0; ks < a.kp / 16; ++ks) {
                    f32x4 v[4];
#pragma unroll
                    for (int g = 0; g < 4; ++g) v[g] = *reinterpret_cast<const f32x4*>(zm + 16 * ks + 4 * g);
#pragma unroll
                    for (int kk = 0; kk < 4; ++kk)
#pragma unroll
                        for (int g = 0; g < 4; ++g) acc = fmaf(v[g][kk], arow[16 * ks + 4 * g + kk], acc);
                }
                s = acc + (a.qc[m] + qrv);
            }
            srow[m] = s;
        }
        __syncthreads();  // the scratch row is complete (block-scope visibility of the global writes)
        row_stats_row(srow, a.M, a.topn, a.lowest, a.use_lds, a.stats + r * 4, smem_raw);
    }
}

// ------------------------------------------------------------------------------------------------
// K9
// ------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void asnorm_apply_kernel(const double* __restrict__ raw,
                                                           const long long* __restrict__ ie,
                                                           const long long* __restrict__ it, long long T,
                                                           const double* __restrict__ stats, long long R,
                                                           double* __restrict__ out) {
    // Four trials per thread and round (a block's four runs of 256): the trial's three streams are loaded for all four before the
    // first statistics row is asked for, the gathers before the first division — the kernel is a chain of dependent memory
    // round trips (index -> statistics row -> store), and one trial per thread left the memory system at 2.2 TB/s.
    const long long stride = (long long)gridDim.x * 1024;
    for (long long i0 = (long long)blockIdx.x * 1024 + threadIdx.x; i0 < T; i0 += stride) {
        long long idx[4], e[4], t[4];
        double r[4];
        bool live[4], ok[4];
#pragma unroll
        for (int u = 0; u < 4; ++u) {
            idx[u] = i0 + 256 * u;
            live[u] = idx[u] < T;
            const long long ic = live[u] ? idx[u] : T - 1;
            e[u] = ie[ic];
            t[u] = it[ic];
            r[u] = raw[ic];
        }
        double4 se[4], st[4];
#pragma unroll
        for (int u = 0; u < 4; ++u) {
            ok[u] = !(e[u] < 0 || e[u] >= R || t[u] < 0 || t[u] >= R);
            // (unconditional loads: an index outside the table reads row 0 and is answered with NaN below; R = 0 -> ok is false
            // everywhere and the table is not touched)
            se[u] = st[u] = make_double4(0.0, 1.0, 0.0, 1.0);
            if (R > 0) {
                se[u] = *reinterpret_cast<const double4*>(stats + 4 * (ok[u] ? e[u] : 0));
                st[u] = *reinterpret_cast<const double4*>(stats + 4 * (ok[u] ? t[u] : 0));
            }
        }
#pragma unroll
        for (int u = 0; u < 4; ++u) {
            double4 o;
            if (!ok[u]) {
                const double nan = __builtin_nan("");
                o = make_double4(nan, nan, nan, nan);
            } else {
                const double zn = (r[u] - se[u].x) / se[u].y;
                const double tn = (r[u] - st[u].x) / st[u].y;
                o = make_double4(zn, tn, (zn + tn) / 2, ((r[u] - se[u].z) / se[u].w + (r[u] - st[u].z) / st[u].w) / 2);
            }
            if (live[u]) *reinterpret_cast<double4*>(out + 4 * idx[u]) = o;
        }
    }
}

}  // namespace

extern "C" {

// the workspace starts with the tile counters of the GEMM (one per XCD), the spilled score rows follow
static constexpr size_t kCohortCtlBytes = 256;

static constexpr int kFallbackBlocks = 64;

static size_t spill_workspace_bytes(int64_t R, int64_t M) {
    // whole matrix if it is below 4 GiB, else row chunks of at least 128 rows
    const unsigned long long row = ((unsigned long long)M + 3) / 4 * 4 * sizeof(float);
    unsigned long long rows = (unsigned long long)R;
    const unsigned long long cap = 4ull << 30;
    if (rows * row > cap) {
        rows = cap / row;
        if (rows < 128) rows = 128;
        rows = rows / 128 * 128;
    }
    return (size_t)(rows * row) + kCohortCtlBytes;
}

// workspace of the fused path for `rows` rows (a multiple of 128): fixed part + per-row arrays + the fallback's scratch rows
static size_t fused_workspace_bytes(const nplda::FusedPlan& p, long long rows, long long M) {
    const size_t lds = (size_t)((M + 3) / 4 * 4);
    return p.fixed_bytes + (size_t)rows * p.row_bytes + (size_t)kFallbackBlocks * lds * sizeof(float) + 256;
}

size_t nplda_cohort_workspace_bytes(int64_t R, int64_t M) {
    if (R <= 0 || M <= 0) return 0;
    size_t need = spill_workspace_bytes(R, M);
    // the fused path (no score matrix) for the default top-N: its workspace is a few KB per row
    const nplda::FusedPlan p = nplda::cohort_fused_plan(M, 500, NPLDA_MAX_DIM);
    if (p.eligible) {
        long long rows = (R + 127) / 128 * 128;
        if (rows > p.max_rows) rows = p.max_rows;
        const long long cap_rows = (long long)(((4ull << 30) - fused_workspace_bytes(p, 0, M)) / p.row_bytes) / 128 * 128;
        if (rows > cap_rows) rows = cap_rows;  // like the spilled matrix: at most 4 GiB, the table is then chunked
        const size_t f = fused_workspace_bytes(p, rows, M);
        if (f > need) need = f;
    }
    return need;
}

size_t nplda_cohort_workspace_bytes_ex(int64_t R, int64_t M, int topn, int D1, int D2) {
    if (R <= 0 || M <= 0 || nplda_kernel_nb(D1, D2) == 0) return 0;
    size_t need = spill_workspace_bytes(R, M);
    const nplda::FusedPlan p = nplda::cohort_fused_plan(M, topn, 16 * nplda_kernel_nb(D1, D2));
    if (p.eligible) {
        long long rows = (R + 127) / 128 * 128;
        if (rows > p.max_rows) rows = p.max_rows;
        const long long cap_rows = (long long)(((4ull << 30) - fused_workspace_bytes(p, 0, M)) / p.row_bytes) / 128 * 128;
        if (rows > cap_rows) rows = cap_rows;
        need = fused_workspace_bytes(p, rows, M);  // the fused path never needs the score matrix
    }
    return need;
}

size_t nplda_cohort_fused_min_workspace_bytes(int64_t M, int topn, int D1, int D2) {
    if (M <= 0 || nplda_kernel_nb(D1, D2) == 0) return 0;
    const nplda::FusedPlan p = nplda::cohort_fused_plan(M, topn, 16 * nplda_kernel_nb(D1, D2));
    return p.eligible ? fused_workspace_bytes(p, 128, M) : 0;
}

size_t nplda_cohort_state_bytes(int64_t M, int topn, int D1, int D2) {
    if (M <= 0 || nplda_kernel_nb(D1, D2) == 0) return 0;
    const nplda::FusedPlan p = nplda::cohort_fused_plan(M, topn, 16 * nplda_kernel_nb(D1, D2));
    return p.eligible ? p.fixed_bytes : 0;
}

int nplda_cohort_prepare_f32(const float* z_coh, const float* q_coh, int64_t M, int64_t ldz, const void* packed, int D0, int D1,
                             int D2, int topn, void* state, size_t state_bytes, nplda_stream_t stream) {
    if (M <= 0 || topn < 1 || M > 0x7ffffff0LL) return NPLDA_EINVAL;
    if (D0 <= 0 || D1 <= 0 || D2 <= 0 || (D0 % 4) != 0) return NPLDA_EINVAL;
    if (!nplda_dims_ok(D0, D1, D2)) return NPLDA_EUNSUPPORTED;
    const NpldaLayout L = nplda_layout(D0, D1, D2);
    if (!z_coh || !q_coh || !packed || !state || ((uintptr_t)state & 255u) != 0) return NPLDA_EINVAL;
    if (ldz < 16 * L.NB || (ldz % 4) != 0 || !nplda_aligned16(z_coh) || !nplda_aligned16(packed)) return NPLDA_EINVAL;
    const nplda::FusedPlan plan = nplda::cohort_fused_plan(M, topn, 16 * L.NB);
    if (!plan.eligible) return NPLDA_EUNSUPPORTED;
    if (state_bytes < plan.fixed_bytes) return NPLDA_ENOSPC;
    return nplda::cohort_fused_prepare(plan, z_coh, q_coh, M, ldz, (const float*)packed + L.oP, L.NB, (unsigned char*)state,
                                       (hipStream_t)stream);
}

static int cohort_stats_impl(const float* z_rows, const float* q_rows, int64_t R, const float* z_coh,
                             const float* q_coh, int64_t M, int64_t ldz, const void* packed, int D0, int D1, int D2,
                             int topn, int select_lowest, double* stats, void* ws, size_t ws_bytes,
                             nplda_stream_t stream, const void* prepared, size_t prepared_bytes);

int nplda_cohort_stats_f32(const float* z_rows, const float* q_rows, int64_t R, const float* z_coh,
                           const float* q_coh, int64_t M, int64_t ldz, const void* packed, int D0, int D1, int D2,
                           int topn, int select_lowest, double* stats, void* ws, size_t ws_bytes,
                           nplda_stream_t stream) {
    return cohort_stats_impl(z_rows, q_rows, R, z_coh, q_coh, M, ldz, packed, D0, D1, D2, topn, select_lowest, stats, ws,
                             ws_bytes, stream, nullptr, 0);
}

int nplda_cohort_stats_prepared_f32(const float* z_rows, const float* q_rows, int64_t R, const float* z_coh,
                                    const float* q_coh, int64_t M, int64_t ldz, const void* packed, int D0, int D1, int D2,
                                    int topn, int select_lowest, double* stats, void* ws, size_t ws_bytes,
                                    const void* state, size_t state_bytes, nplda_stream_t stream) {
    if (!state || ((uintptr_t)state & 255u) != 0) return NPLDA_EINVAL;
    return cohort_stats_impl(z_rows, q_rows, R, z_coh, q_coh, M, ldz, packed, D0, D1, D2, topn, select_lowest, stats, ws,
                             ws_bytes, stream, state, state_bytes);
}

static int cohort_stats_impl(const float* z_rows, const float* q_rows, int64_t R, const float* z_coh,
                             const float* q_coh, int64_t M, int64_t ldz, const void* packed, int D0, int D1, int D2,
                             int topn, int select_lowest, double* stats, void* ws, size_t ws_bytes,
                             nplda_stream_t stream, const void* prepared, size_t prepared_bytes) {
    if (R < 0 || M < 0 || topn < 1) return NPLDA_EINVAL;
    if (D0 <= 0 || D1 <= 0 || D2 <= 0 || (D0 % 4) != 0) return NPLDA_EINVAL;
    if (!nplda_dims_ok(D0, D1, D2)) return NPLDA_EUNSUPPORTED;
    if (R == 0) return NPLDA_OK;
    if (M == 0 || M > 0x7ffffff0LL) return NPLDA_EINVAL;
    const NpldaLayout L = nplda_layout(D0, D1, D2);
    if (!z_rows || !q_rows || !z_coh || !q_coh || !packed || !stats || !ws) return NPLDA_EINVAL;
    if (ldz < 16 * L.NB || (ldz % 4) != 0 || !nplda_aligned16(z_rows) || !nplda_aligned16(z_coh) ||
        !nplda_aligned16(packed) || !nplda_aligned16(ws))
        return NPLDA_EINVAL;
    const long long lds = (M + 3) / 4 * 4;
    const size_t row_bytes = (size_t)lds * sizeof(float);
    if (ws_bytes < kCohortCtlBytes + row_bytes) return NPLDA_ENOSPC;
    hipStream_t st = (hipStream_t)stream;
    const int use_lds = M <= kMaxRowLds ? 1 : 0;
    const size_t shmem = (use_lds ? (size_t)((M + 3) / 4 * 4) * 4 : 0) + 64 * 4 + (kRowThreads / 64) * 32 + (2 * kListCap + 4 + 8 + kListCap / 2) * 4 + 16;
    if (use_lds && shmem > 64 * 1024) {
        hipError_t e = hipFuncSetAttribute((const void*)row_stats_kernel, hipFuncAttributeMaxDynamicSharedMemorySize,
                                           (int)shmem);
        if (e != hipSuccess) return (int)e;
        e = hipFuncSetAttribute((const void*)cohort_fallback_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, (int)shmem);
        if (e != hipSuccess) return (int)e;
    }

    // ---- fused path: statistics in the GEMM's epilogue, no score matrix (nplda_cohort_fused.hip) -----------------------
    // Chosen by the shape (M, top-N) alone, and every row is computed independently of its neighbours, so the result of
    // a row does not depend on R, on the row's position or on how the workspace chunks the table.
    const nplda::FusedPlan plan = nplda::cohort_fused_plan(M, topn, 16 * L.NB);
    // a prepared cohort (nplda_cohort_prepare_f32) belongs to the fused path of exactly this shape and top-N
    if (prepared && (!plan.eligible || prepared_bytes < plan.fixed_bytes || ws_bytes < fused_workspace_bytes(plan, 128, M)))
        return NPLDA_EINVAL;
    if (plan.eligible && ws_bytes >= fused_workspace_bytes(plan, 128, M)) {
        const size_t scratch_bytes = (size_t)kFallbackBlocks * lds * sizeof(float);
        long long rows_cap = (long long)((ws_bytes - plan.fixed_bytes - scratch_bytes - 256) / plan.row_bytes) / 128 * 128;
        if (rows_cap > plan.max_rows) rows_cap = plan.max_rows;
        if (rows_cap > (R + 127) / 128 * 128) rows_cap = (R + 127) / 128 * 128;
        float* scratch = (float*)((char*)ws + (ws_bytes - scratch_bytes) / 256 * 256);
        const long long resident = nplda::cohort_fused_resident_blocks();
        if (resident <= 0) return NPLDA_EINVAL;
        for (long long r0 = 0; r0 < R; r0 += rows_cap) {
            const long long rc = (R - r0 < rows_cap) ? R - r0 : rows_cap;
            unsigned* fail_rows = nullptr;
            unsigned* nfail = nullptr;
            if (int rc2 = nplda::cohort_fused_run(plan, z_rows + r0 * ldz, q_rows + r0, rc, z_coh, q_coh, M, ldz,
                                                  (const float*)packed + L.oP, L.NB, topn, select_lowest ? 1 : 0,
                                                  stats + 4 * r0, (unsigned char*)ws, rows_cap, r0 == 0, &fail_rows,
                                                  &nfail, resident, st, (const unsigned char*)prepared, D2))
                return rc2;
            FallbackArgs fb;
            fb.zr = z_rows + r0 * ldz; fb.qr = q_rows + r0; fb.zc = z_coh; fb.qc = q_coh; fb.P = (const float*)packed + L.oP;
            fb.M = M; fb.ldz = ldz; fb.lds = lds; fb.kp = 16 * L.NB; fb.topn = topn; fb.lowest = select_lowest ? 1 : 0;
            fb.use_lds = use_lds; fb.fail_rows = fail_rows; fb.nfail = nfail; fb.scratch = scratch; fb.stats = stats + 4 * r0;
            hipLaunchKernelGGL(cohort_fallback_kernel, dim3(kFallbackBlocks), dim3(kRowThreads), shmem, st, fb);
            if (int rc2 = nplda_launch_status()) return rc2;
        }
        return NPLDA_OK;
    }

    // ---- spill path: score matrix in the workspace, then one block per row -----------------------------------------------
    long long rows_per = (long long)((ws_bytes - kCohortCtlBytes) / row_bytes);
    if (rows_per > R) rows_per = R;
    // persistent tile walkers: as many blocks as are resident at once, a multiple of 8 so that a block keeps its XCD
    long long resident = 0;
    {
        int dev = 0, cus = 0, per_cu = 0;
        if (hipGetDevice(&dev) != hipSuccess ||
            hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, dev) != hipSuccess ||
            hipOccupancyMaxActiveBlocksPerMultiprocessor(&per_cu, cohort_gemm_kernel, 256, 0) != hipSuccess)
            return NPLDA_EINVAL;
        resident = (long long)cus * per_cu / 8 * 8;
        if (resident < 8) resident = 8;
    }
    for (long long r0 = 0; r0 < R; r0 += rows_per) {
        const long long rc = (R - r0 < rows_per) ? R - r0 : rows_per;
        CohortGemmArgs a;
        a.zr = z_rows + r0 * ldz; a.qr = q_rows + r0; a.zc = z_coh; a.qc = q_coh;
        a.P = (const float*)packed + L.oP; a.R = rc; a.M = M; a.ldz = ldz; a.lds = lds; a.ksteps = L.NB; a.S = (float*)((char*)ws + kCohortCtlBytes);
        const long long nx = (M + 127) / 128, ny = (rc + 127) / 128;
        a.nxp = (int)((nx + 7) / 8 < 24 ? (nx + 7) / 8 : 24);
        a.ny = (int)ny;
        const long long nsb = (nx + 8LL * a.nxp - 1) / (8LL * a.nxp);
        if (nx > 0x00ffffffLL || ny > 0x00ffffffLL || (long long)a.nxp * ny * nsb > 0x0fffffffLL) return NPLDA_EINVAL;
        a.nx = (int)nx;
        a.nsb = (int)nsb;
        a.ctr = (unsigned*)ws;
        long long grid = 8LL * a.nxp * ny;  // at most one block per tile of the busiest XCD
        if (grid > resident) grid = resident;
        if (hipMemsetAsync(ws, 0, kCohortCtlBytes, st) != hipSuccess) return NPLDA_EINVAL;
        hipLaunchKernelGGL(cohort_gemm_kernel, dim3((unsigned)grid), dim3(256), 0, st, a);
        if (int rc2 = nplda_launch_status()) return rc2;
        hipLaunchKernelGGL(row_stats_kernel, dim3((unsigned)rc), dim3(kRowThreads), shmem, st, (const float*)((const char*)ws + kCohortCtlBytes),
                           (long long)lds, (long long)M, topn, select_lowest ? 1 : 0, use_lds, stats + 4 * r0);
        if (int rc2 = nplda_launch_status()) return rc2;
    }
    return NPLDA_OK;
}

int nplda_row_stats_f32(const float* S, int64_t lds, int64_t R, int64_t M, int topn, int select_lowest, double* stats,
                        nplda_stream_t stream) {
    if (R < 0 || M < 0 || topn < 1) return NPLDA_EINVAL;
    if (R == 0) return NPLDA_OK;
    if (M == 0 || !S || !stats || lds < M) return NPLDA_EINVAL;
    const int use_lds = M <= kMaxRowLds ? 1 : 0;
    const size_t shmem = (use_lds ? (size_t)((M + 3) / 4 * 4) * 4 : 0) + 64 * 4 + (kRowThreads / 64) * 32 + (2 * kListCap + 4 + 8 + kListCap / 2) * 4 + 16;
    if (use_lds && shmem > 64 * 1024) {
        hipError_t e = hipFuncSetAttribute((const void*)row_stats_kernel, hipFuncAttributeMaxDynamicSharedMemorySize,
                                           (int)shmem);
        if (e != hipSuccess) return (int)e;
    }
    if (R > 0x7fffffffLL || M > 0x7ffffff0LL) return NPLDA_EINVAL;  // 32-bit group indices inside the kernel
    hipLaunchKernelGGL(row_stats_kernel, dim3((unsigned)R), dim3(kRowThreads), shmem, (hipStream_t)stream, S,
                       (long long)lds, (long long)M, topn, select_lowest ? 1 : 0, use_lds, stats);
    return nplda_launch_status();
}

int nplda_asnorm_apply_f64(const double* raw, const int64_t* ie, const int64_t* it, int64_t T, const double* stats,
                           int64_t R, double* out, nplda_stream_t stream) {
    if (T < 0 || R < 0) return NPLDA_EINVAL;
    if (T == 0) return NPLDA_OK;
    if (!raw || !ie || !it || !stats || !out) return NPLDA_EINVAL;
    if ((((uintptr_t)stats) & 31u) || (((uintptr_t)out) & 31u)) return NPLDA_EINVAL;
    long long blocks = (T + 1023) / 1024;
    if (blocks > 8192) blocks = 8192;
    hipLaunchKernelGGL(asnorm_apply_kernel, dim3((unsigned)blocks), dim3(256), 0, (hipStream_t)stream, raw,
                       (const long long*)ie, (const long long*)it, (long long)T, stats, (long long)R, out);
    return nplda_launch_status();
}

}  // extern "C"

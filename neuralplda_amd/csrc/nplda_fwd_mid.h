// nplda_fwd_mid.h — pair scoring between the small-batch and the streaming regimes (one to a few dozen 16-pair tiles
// per CU): the batch sizes of validate() (5 x batch_size = 20 480 pairs, xvector_NeuralPlda_pytorch.py:125), of the
// score-file chunks (utils/scorefile_generator.py:22-56) and of every 8-way shard of a modest trial list.
//
// Why neither neighbour fits.  The streaming kernels (v3 / v5) give a WAVE a whole 16-pair tile and a block 128 pairs:
// 100 us of work per CU and tile, so below ~64 k pairs half the CUs idle while the others run a second tile
// (0.40 of the MFMA peak at 16 385 pairs).  The small-batch kernel splits one tile over the 4 SIMDs of a CU by output
// FEATURES, which balances any batch to within one 11 us tile — but every wave then loads all of the tile's x rows
// (4 x redundant) plus its own weight fragments: 5.5 vector-memory instructions per 20 MFMAs, and with one wave per
// SIMD each of them idles the matrix pipe for ~49 cycles, 0.58 of the peak at best.
//
// This kernel keeps the small kernel's granularity and cuts the loads per MFMA by 3:
//  * layer 1 (77 % of the work) is split over the 4 waves by K, not by features: wave w runs the k16-steps
//    {8 m + 2 w, 8 m + 2 w + 1} for ALL feature blocks and all row groups of the group — its x loads are its own
//    (whole 128-byte lines, nothing redundant) and its weight fragments are a contiguous 2 x NB KB run of the image;
//  * a block works on GROUPS of T = 2 tiles (4 row groups: 2 tiles x 2 sides) through one pass of the weights:
//    NB + 2 T loads per 8 NB T MFMAs (14 per 160 at NB = 10 against 5.5 per 20); a block's tile range is contiguous
//    and balanced to one tile over the grid (the odd tile of a block runs as a T = 1 group);
//  * the four partial u = W1 x meet through LDS (each wave exports the units it does not own: 30 KB per wave at
//    T = 2, one barrier) in a fixed order, and from there the tile continues feature-split exactly like the small
//    kernel: row norms through LDS, y published in accumulator layout, layer 2 from a register ring of W2 fragments.
//  * the accumulator index space of a wave is PERMUTED by its wave id (block a <-> (a & 4) | ((a + w) & 3), row group
//    rho <-> rho ^ swz(w)) so that "the units this wave owns" are compile-time register indices: no dynamic register
//    indexing, only wave-uniform (scalar) address arithmetic.
// Same arithmetic per element as the other forward kernels except the association of the layer-1 K sum (four partial
// sums over interleaved k-steps) and of the cross-feature sums; parity is to the stated fp32 tolerance.
// 512-d x-vectors only (KS1 = 32: the loop is fully unrolled, see nplda_fwd_small.h), NB = 10 / 11.
#pragma once
#include "nplda_fwd_kernel.h"

namespace nplda {

#ifdef NPLDA_MID_STAMPS  // tools/exp_mid.hip only: shader-clock and 100 MHz stamps of wave 0 of one block at the phase boundaries
__device__ unsigned long long g_mid_stamps[32];
#define NPLDA_MSTAMP(i) do { if (blockIdx.x == gridDim.x / 2 && threadIdx.x == 0) { g_mid_stamps[i] = __builtin_amdgcn_s_memrealtime(); g_mid_stamps[16 + i] = __builtin_readcyclecounter(); } } while (0)
#else
#define NPLDA_MSTAMP(i) do {} while (0)
#endif

template <int NB, int T>
struct MidCfg {
    static_assert(NB == 10 || NB == 11, "mid kernel: D = 145..176");
    static_assert(T == 0 || T == 1 || T == 2, "groups of one or two tiles, or a trailing HALF tile (T = 0)");
    // T = 0 (round 5): one row group of 16 rows — in pair mode the x1 rows of 8 pairs in lanes j < 8 and their x2 rows in
    // lanes j >= 8 (the half tile of nplda_train_fb_half.h: a pair's two embeddings meet by a DPP row rotation), in embed
    // mode 16 consecutive rows.  A block's range is counted in such halves, so the grid balances to 8 pairs / 16 rows.
    static constexpr int RG = T == 0 ? 1 : 2 * T;      // row groups of a group: (tile, side)
    static constexpr int LB = NB - 8;                  // left-over blocks (blocks 0..7 go two to a wave, whole)
    // own left-over units of a wave: LS block slots x LR row groups (in the wave's permuted index space: a = 8 + i, rho < LR)
    static constexpr int LS = (NB == 11 && T == 2) ? 3 : 1;
    static constexpr int LR = T == 0 ? 1 : ((NB == 10) ? T : (T == 2 ? 1 : 2));
    static constexpr int UW = 2 * RG + LS * LR;        // units (f32x4 accumulators) a wave owns
    // the two sides of a left-over unit sit in two waves (w, w ^ 1): their z meet through LDS for the score
    static constexpr bool XCH = (NB == 10 && T == 1) || (NB == 11 && T == 2);
};

// wave-uniform index maps (all arguments and results live in SGPRs); T may be a run-time value (the NEXT group's size)
template <int NB>
__device__ __forceinline__ int mid_swz(int T, int w) {
    if (T == 0) return 0;
    if constexpr (NB == 10) return T == 2 ? 2 * (w & 1) : (w & 1);
    else return T == 2 ? w : 0;
}
template <int NB>
__device__ __forceinline__ int mid_blk(int T, int a, int w) {  // accumulator slot a of wave w holds feature block ...
    if (a < 8) return (a & 4) | ((a + w) & 3);
    const int i = a - 8;
    // T = 0: the LB left-over blocks x one row group are LB units: wave w < LB owns block 8 + w in its slot 8
    if (T == 0) return 8 + (i + w) % (NB - 8);
    if constexpr (NB == 10) return 8 + ((i + (w >> 1)) & 1);
    else return T == 2 ? 8 + i : 8 + (w < 3 ? (i + w) % 3 : i);
}

// The operand rings live across groups: while a group finishes its K loop it already fetches the first steps of the next.
template <int NB>
struct MidRing {
    f32x4 wf[2][NB];  // weight fragments: step i in slot i & 1, refilled one step ahead (L2)
    f32x4 xf[4][4];   // x fragments: step i in slot i & 3, refilled three steps ahead (HBM), up to 4 row groups
};
// addresses of a group: x row pointers of the permuted row groups (this wave's share of the k range folded in; rows
// clamped: always-valid addresses, every load is unconditional) and the fragment offsets of the permuted blocks
template <int NB>
struct MidAddr {
    const float* xr[4];
    unsigned voff[NB];  // byte offset of this lane's 16 bytes of slot a's fragment inside a k16-step of the image
};
// EMBED: the rows are single rows of ONE table (a.xa); a "tile" is 32 consecutive rows, its two "sides" the two halves.
// A group starts at HALF tile h0 (8 pairs / 16 embedding rows each): pair 8 h0 / row 16 h0.
// IDX: the rows are named through a.ia / a.ib.  A template parameter, not a test of the pointers: as a (uniform) branch around
// each index load the compiler's wait placement at the joins made every group wait for the stores and loads in flight —
// 4.5 k cycles of an embedding group's 73 k, ~1 k of a pair group's (round 6, tools/exp_mid.hip).
template <int NB, bool EMBED = false, bool HALF = true, bool IDX = false>  // HALF = false: T is never 0 (the checks fold away)
__device__ __forceinline__ void mid_addr(const FwdArgs& a, long long h0, int Tin, int wave, int lane, MidAddr<NB>& A) {
    const int T = (!HALF && Tin == 0) ? 1 : Tin;
    if (!HALF) __builtin_assume(T != 0);
    const int swz = mid_swz<NB>(T, wave);
    const int rgm = T == 0 ? 0 : 2 * T - 1;
    long long rows[4];
    int sides[4];
#pragma unroll
    for (int rho = 0; rho < 4; ++rho) {
        const int rg = (rho & rgm) ^ swz;  // T = 1: slots 2, 3 repeat 0, 1 (loaded, never used); T = 0: all repeat 0
        // pair mode, T = 0: lanes j < 8 hold the x1 rows of pairs 8 h0 + j, lanes j >= 8 the x2 rows of the same pairs
        sides[rho] = T == 0 ? ((lane >> 3) & 1) : (rg & 1);
        const long long row = EMBED ? h0 * 16 + 16 * rg + (lane & 15)
                                    : (T == 0 ? h0 * 8 + (lane & 7) : h0 * 8 + (rg >> 1) * 16 + (lane & 15));
        rows[rho] = row >= a.n ? a.n - 1 : row;
    }
    if constexpr (IDX) {
        // rows named by index (nplda_embed_rows_f32: one table; indexed pairs: the pair's rows of the x-vector table) — the
        // group's index loads leave together, one wait for all of them
#pragma unroll
        for (int rho = 0; rho < 4; ++rho) rows[rho] = (EMBED || !sides[rho] ? a.ia : a.ib)[rows[rho]];
#pragma unroll
        for (int rho = 0; rho < 4; ++rho) rows[rho] = rows[rho] < 0 ? 0 : (rows[rho] < a.ntab ? rows[rho] : a.ntab - 1);
    }
#pragma unroll
    for (int rho = 0; rho < 4; ++rho) {
        const long long row = rows[rho];
        if (EMBED) {
            // two-table form: the row's own table — as a byte distance added to ONE base (a per-lane choice between the two
            // pointer FIELDS is compiled as an indexed read of a scratch copy of them: 16 scratch loads and vmcnt(0) waits
            // per group, round 6)
            const bool second = !IDX && a.nsplit > 0 && row >= a.nsplit;
            const long long dxb = reinterpret_cast<const char*>(a.xb) - reinterpret_cast<const char*>(a.xa);
            const float* pr = a.xa + (second ? row - a.nsplit : row) * a.ldx + 4 * (lane >> 4) + 32 * wave;
            A.xr[rho] = reinterpret_cast<const float*>(reinterpret_cast<const char*>(pr) + (second ? dxb : 0LL));
        } else {
            A.xr[rho] = (sides[rho] ? a.xb : a.xa) + row * a.ldx + 4 * (lane >> 4) + 32 * wave;  // k16-steps 2 w, 2 w + 1 (+ 8 m)
        }
    }
#pragma unroll
    for (int s = 0; s < NB; ++s) A.voff[s] = (unsigned)(mid_blk<NB>(T, s, wave) * 64 + lane) * 16u;
}
// Every weight load is ONE instruction: buffer_load_dwordx4 with the image as the buffer, this lane's fragment offset in a
// VGPR (MidAddr::voff) and the k16-step's offset in an SGPR — no address arithmetic on the VALU at all.  (global_load
// forms cost one to three VALU / readlane instructions per load however the sum was written: hipcc re-associates it into
// one 64-bit vector address per load of the unrolled loop and parks those in AGPRs.)  x loads are (row pointer) + immediate.
typedef unsigned u32x4_mid __attribute__((ext_vector_type(4)));
__device__ __forceinline__ __amdgpu_buffer_rsrc_t mid_image(const FwdArgs& a) {
    return __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(a.packed), 0, (int)(a.total * 4), 0x00020000);
}
__device__ __forceinline__ f32x4 mid_ldw(__amdgpu_buffer_rsrc_t img, unsigned voff, int soff) {
    return __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(img, (int)voff, soff, 0));
}
// step i (0 .. 7) of wave w is k16-step 8 (i >> 1) + (i & 1) + 2 w: whole 128-byte lines of x per wave
template <int NB>
__device__ __forceinline__ int mid_w1_step(int i, int wave) {
    return (2 * wave + 8 * (i >> 1) + (i & 1)) * (NB * 1024);
}
template <int NB>
__device__ __forceinline__ void mid_fetchx(const MidAddr<NB>& A, MidRing<NB>& R, int i, int rho) {
    R.xf[i & 3][rho] = *reinterpret_cast<const f32x4*>(A.xr[rho] + 16 * (8 * (i >> 1) + (i & 1)));
}

// One group of T tiles (T = 0: one half tile) starting at half tile `tile0` (8 pairs / 16 embedding rows per half); the next
// group (TN tiles from half tile tile_n; the block's last group names itself) gets its first loads from here.  LDS: red (the exchange of the layer-1 partial sums; reused as the y tiles of
// layer 2), ssb / scb (row norms, scores), zx (left-over z of the other side), cv (b1, b2, Q, P).
template <int NB, int T, bool EMBED = false, bool HALF = true, bool IDX = false>
__device__ __forceinline__ void mid_group(const FwdArgs& a, long long tile0, int TN, long long tile_n, MidRing<NB>& R,
                                          int wave, int lane, f32x4* red, float (*ssb)[4][16], float (*scb)[2][16],
                                          f32x4 (*zx)[3][64], const f32x4* cv) {
    using C = MidCfg<NB, T>;
    constexpr int RG = C::RG, LS = C::LS, LR = C::LR, UW = C::UW;
    constexpr int KSW = 8;  // k16-steps per wave (KS1 = 32)
    const int j = lane & 15;
    const int g = lane >> 4;
    const int swz = mid_swz<NB>(T, wave);
    const __amdgpu_buffer_rsrc_t img = mid_image(a);
    const f32x4* b1p = cv;
    const f32x4* b2p = cv + NB * 4;
    const f32x4* Qp = cv + 2 * NB * 4;
    const f32x4* Pp = cv + 3 * NB * 4;
    auto blk = [&](int s) { return mid_blk<NB>(T, s, wave); };

    NPLDA_MSTAMP(0);
    MidAddr<NB> A, AN;
    mid_addr<NB, EMBED, HALF, IDX>(a, tile0, T, wave, lane, A);
    NPLDA_MSTAMP(11);
    mid_addr<NB, EMBED, HALF, IDX>(a, tile_n, TN, wave, lane, AN);
    NPLDA_MSTAMP(12);
    bool lo_valid = true;  // NB = 11, T = 1: wave 3 owns no left-over block; T = 0: the LB left-over units go to waves 0 .. LB - 1
    if constexpr (NB == 11 && T == 1) lo_valid = wave < 3;
    if constexpr (T == 0) lo_valid = wave < NB - 8;
    // where unit (slot s, row group rho) of this wave's partial sums goes: wave v, index u, red[v][(src - v - 1) & 3][u]
    auto export_unit = [&](int s, int rho, const f32x4& val) {
        const bool own_static = s < 8 ? (s & 3) == 0 : ((s - 8) < LS && rho < LR);
        if (own_static && (s < 8 || lo_valid)) return;
        const int b = blk(s);
        const int rg = rho ^ swz;
        int v, u;
        if (s < 8) {
            v = b & 3;
            u = (b >> 2) * RG + (rg ^ mid_swz<NB>(T, v));
        } else if constexpr (T == 0) {
            v = b - 8;
            u = 2 * RG;
        } else if constexpr (NB == 10) {
            v = 2 * (b - 8) + (T == 2 ? (rg >> 1) : rg);
            u = 2 * RG + (T == 2 ? (rg & 1) : 0);
        } else if constexpr (T == 2) {
            v = rg;
            u = 2 * RG + (b - 8);
        } else {
            v = b - 8;
            u = 2 * RG + rg;
        }
        int lo = lane;
        asm volatile("" : "+v"(lo));  // or every unit's LDS address is precomputed outside the group loop and spilled
        red[((v * 3 + ((wave - v - 1) & 3)) * UW + u) * 64 + lo] = val;
    };

    // ---- layer 1, this wave's k16-steps, all blocks and row groups -----------------------------------------------------
    f32x4 acc[NB][RG];
    const f32x4 zero4 = {0.f, 0.f, 0.f, 0.f};
    // the refills of step i, spread through its MFMAs — one load per block of MFMAs, pinned by sched_barrier (left free, the
    // scheduler sinks every load to its first use): x three steps ahead (HBM), weights one step ahead (L2); past the
    // group's last step they are the first steps of the NEXT group
    auto refill = [&](int i, int q, int wnext) {  // q: 0 .. 4 NB - 1, position inside the step
        if (q < RG) {
            if (i + 3 < KSW) mid_fetchx<NB>(A, R, i + 3, q);
            else mid_fetchx<NB>(AN, R, i + 3 - KSW, q);
        } else if (T <= 1 && q < 4 && i + 3 >= KSW) {
            mid_fetchx<NB>(AN, R, i + 3 - KSW, q);  // the next group may have four row groups
        } else if (q >= NB && q < 2 * NB) {
            R.wf[(i + 1) & 1][q - NB] = mid_ldw(img, i + 1 < KSW ? A.voff[q - NB] : AN.voff[q - NB], wnext);
        }
    };
    NPLDA_MSTAMP(1);
#pragma unroll
    for (int i = 0; i < KSW - 1; ++i) {
        if (i == 1) NPLDA_MSTAMP(2);
        const int wnext = mid_w1_step<NB>(i + 1, wave);
#pragma unroll
        for (int r = 0; r < 4; ++r) {
#pragma unroll
            for (int s = 0; s < NB; ++s) {
#pragma unroll
                for (int rho = 0; rho < RG; ++rho)
                    acc[s][rho] = __builtin_amdgcn_mfma_f32_16x16x4f32(R.wf[i & 1][s][r], R.xf[i & 3][rho][r],
                                                                       (i == 0 && r == 0) ? zero4 : acc[s][rho], 0, 0, 0);
                refill(i, r * NB + s, wnext);
                __builtin_amdgcn_sched_barrier(0);
            }
        }
    }
    // the last step block-major: a block's sums are final after its 4 RG MFMAs and leave for LDS under the next block's
    {
        constexpr int i = KSW - 1;
        const int wnext = mid_w1_step<NB>(0, wave);
#pragma unroll
        for (int s = 0; s < NB; ++s) {
#pragma unroll
            for (int r = 0; r < 4; ++r)
#pragma unroll
                for (int rho = 0; rho < RG; ++rho)
                    acc[s][rho] = __builtin_amdgcn_mfma_f32_16x16x4f32(R.wf[i & 1][s][r], R.xf[i & 3][rho][r], acc[s][rho], 0, 0, 0);
            refill(i, s, wnext);
            refill(i, NB + s, wnext);
            if (s > 0) {
#pragma unroll
                for (int rho = 0; rho < RG; ++rho) export_unit(s - 1, rho, acc[s - 1][rho]);
            }
            __builtin_amdgcn_sched_barrier(0);
        }
#pragma unroll
        for (int rho = 0; rho < RG; ++rho) export_unit(NB - 1, rho, acc[NB - 1][rho]);
    }

    NPLDA_MSTAMP(3);
    // ---- first W2 fragments on their way before the exchange -----------------------------------------------------------
    constexpr int PF2 = T == 2 ? 2 : 4;  // k-blocks of W2 in flight (L2): a T = 2 k-block is 40+ MFMAs, two ahead is ample
    constexpr int NW2 = 2 + LS;  // W2 fragments per k-block: the wave's two whole blocks and its left-over slots
    f32x4 w2[PF2][NW2];
    auto fetch2 = [&](int slot, int kb) {
        const int kbc = kb < NB ? kb : NB - 1;
        const int wb = (int)(a.oW2 * 4) + kbc * (NB * 1024);
#pragma unroll
        for (int q = 0; q < NW2; ++q) {
            const int s = q < 2 ? 4 * q : 8 + (q - 2);
            w2[slot][q] = mid_ldw(img, A.voff[s], wb);
        }
    };
#pragma unroll
    for (int p = 0; p < PF2; ++p) fetch2(p, p);
    NPLDA_MSTAMP(4);
    __syncthreads();
    NPLDA_MSTAMP(5);
    // own units: u = W1 x + b1, summed own + next wave + ... (a fixed order), then the partial row norms
    f32x4 uF[2][RG], uL[LS][LR];
    float ss[RG];
#pragma unroll
    for (int rho = 0; rho < RG; ++rho) ss[rho] = 0.f;
    auto own_sum = [&](const f32x4& own, int u, int b) {
        const f32x4* rp = red + ((size_t)wave * 3 * UW + u) * 64 + lane;
        f32x4 v = own + rp[0];
        v += rp[UW * 64];
        v += rp[2 * UW * 64];
        return v + b1p[4 * b + g];
    };
#pragma unroll
    for (int i = 0; i < 2; ++i) {
#pragma unroll
        for (int rho = 0; rho < RG; ++rho) {
            uF[i][rho] = own_sum(acc[4 * i][rho], i * RG + rho, blk(4 * i));
#pragma unroll
            for (int r = 0; r < 4; ++r) ss[rho] = fmaf(uF[i][rho][r], uF[i][rho][r], ss[rho]);
        }
    }
#pragma unroll
    for (int i = 0; i < LS; ++i) {
#pragma unroll
        for (int rho = 0; rho < LR; ++rho) {
            uL[i][rho] = own_sum(acc[8 + i][rho], 2 * RG + i * LR + rho, blk(8 + i));
            if (lo_valid) {
#pragma unroll
                for (int r = 0; r < 4; ++r) ss[rho] = fmaf(uL[i][rho][r], uL[i][rho][r], ss[rho]);
            }
        }
    }
#pragma unroll
    for (int rho = 0; rho < RG; ++rho) {
        float v = wave_xor_add(ss[rho], 16);
        v = wave_xor_add(v, 32);
        if (g == 0) ssb[wave][rho ^ swz][j] = v;
    }
    NPLDA_MSTAMP(6);
    __syncthreads();  // every red read is done: the y tiles may overwrite it
    f32x4* ylds = red;  // [rg][kb][lane]
    // 1 / max(||u||, eps) (F.normalize, utils/models.py:368): lane (j, g) forms it for row j of row group g — ONE pass through
    // the IEEE sqrt / divide sequences instead of one per row group (with one wave per SIMD every dependent instruction of
    // such a chain is paid in full) — and the row groups' values cross the 16-lane rows by ds_bpermute
    float inv_l;
    {
        const int rgl = g & (RG - 1);  // (RG = 1: every lane group forms row group 0's value)
        inv_l = 1.0f / fmaxf(sqrtf(((ssb[0][rgl][j] + ssb[1][rgl][j]) + ssb[2][rgl][j]) + ssb[3][rgl][j]), 1e-12f);
    }
#pragma unroll
    for (int rho = 0; rho < RG; ++rho) {
        const int rg = rho ^ swz;
        const float inv = __shfl(inv_l, 16 * rg + j, 64);
        int lo = lane;
        asm volatile("" : "+v"(lo));
#pragma unroll
        for (int i = 0; i < 2; ++i) {
            uF[i][rho] *= inv;
            ylds[(rg * NB + blk(4 * i)) * 64 + lo] = uF[i][rho];
        }
        if (rho < LR) {
#pragma unroll
            for (int i = 0; i < LS; ++i) {
                uL[i][rho] *= inv;
                if (lo_valid) ylds[(rg * NB + blk(8 + i)) * 64 + lo] = uL[i][rho];
            }
        }
    }
    NPLDA_MSTAMP(7);
    __syncthreads();
    NPLDA_MSTAMP(8);

    // ---- layer 2, feature-split: this wave's units from all of y ---------------------------------------------------------
    f32x4 zF[2][RG], zL[LS][LR];
#pragma unroll
    for (int i = 0; i < 2; ++i) {
        const f32x4 bv = b2p[4 * blk(4 * i) + g];
#pragma unroll
        for (int rho = 0; rho < RG; ++rho) zF[i][rho] = bv;
    }
#pragma unroll
    for (int i = 0; i < LS; ++i) {
        const f32x4 bv = b2p[4 * blk(8 + i) + g];
#pragma unroll
        for (int rho = 0; rho < LR; ++rho) zL[i][rho] = bv;
    }
#pragma unroll
    for (int kb = 0; kb < NB; ++kb) {
        const int sl = kb % PF2;
        f32x4 yv[RG];
#pragma unroll
        for (int rho = 0; rho < RG; ++rho) yv[rho] = ylds[((rho ^ swz) * NB + kb) * 64 + lane];
#pragma unroll
        for (int r = 0; r < 4; ++r) {
#pragma unroll
            for (int i = 0; i < 2; ++i)
#pragma unroll
                for (int rho = 0; rho < RG; ++rho)
                    zF[i][rho] = __builtin_amdgcn_mfma_f32_16x16x4f32(w2[sl][i][r], yv[rho][r], zF[i][rho], 0, 0, 0);
#pragma unroll
            for (int i = 0; i < LS; ++i)
#pragma unroll
                for (int rho = 0; rho < LR; ++rho)
                    zL[i][rho] = __builtin_amdgcn_mfma_f32_16x16x4f32(w2[sl][2 + i][r], yv[rho][r], zL[i][rho], 0, 0, 0);
        }
        if (kb + PF2 < NB) fetch2(sl, kb + PF2);
        __builtin_amdgcn_sched_barrier(0);
    }

    NPLDA_MSTAMP(9);
    if constexpr (EMBED) {
        // ---- extract_plda_embeddings (utils/models.py:366-370): z rows out, q = sum_f Q z^2 per row (the indexed scorer's
        // self term) reduced over the waves through LDS --------------------------------------------------------------------
        float qp[RG];
#pragma unroll
        for (int rho = 0; rho < RG; ++rho) qp[rho] = 0.f;
        // z leaves in 64-byte pieces per four lanes (frag_rows_to_high_lanes, nplda_common.h): in the accumulator's own lane
        // order a block's 40 KB of z took 2.1 k cycles to issue, 1.5 k this way (tools/exp_mid.hip)
#ifndef NPLDA_MID_TSTORE_NB11
#define NPLDA_MID_TSTORE_NB11 1
#endif
        constexpr bool TSTORE = NB == 10 || NPLDA_MID_TSTORE_NB11 != 0;
        auto out_unit = [&](const f32x4& z, int b, int rho, bool valid) {
            const int rg = rho ^ swz;
            if constexpr (TSTORE) {
                const f32x4 zt = frag_rows_to_high_lanes(z, lane);
                const long long srow = tile0 * 16 + 16 * rg + (lane >> 2);
                if (valid && srow < a.n) *reinterpret_cast<f32x4*>(a.out_z + srow * a.ldz + 16 * b + 4 * (lane & 3)) = zt;
            } else {
                const long long row = tile0 * 16 + 16 * rg + j;
                if (valid && row < a.n) *reinterpret_cast<f32x4*>(a.out_z + row * a.ldz + 16 * b + 4 * g) = z;
            }
            if (valid) {
                const f32x4 q = Qp[4 * b + g];
#pragma unroll
                for (int r = 0; r < 4; ++r) qp[rho] = fmaf(q[r] * z[r], z[r], qp[rho]);
            }
        };
#pragma unroll
        for (int i = 0; i < 2; ++i)
#pragma unroll
            for (int rho = 0; rho < RG; ++rho) out_unit(zF[i][rho], blk(4 * i), rho, true);
#pragma unroll
        for (int i = 0; i < LS; ++i)
#pragma unroll
            for (int rho = 0; rho < LR; ++rho) out_unit(zL[i][rho], blk(8 + i), rho, lo_valid);
        NPLDA_MSTAMP(13);
        if (a.out_q != nullptr) {
#pragma unroll
            for (int rho = 0; rho < RG; ++rho) {
                float v = wave_xor_add(qp[rho], 16);
                v = wave_xor_add(v, 32);
                if (g == 0) ssb[wave][rho ^ swz][j] = v;  // (the row-norm buffer is free since the y tiles were published)
            }
            __syncthreads();
            if (wave < RG && g == 0) {
                const long long row = tile0 * 16 + 16 * wave + j;
                if (row < a.n) a.out_q[row] = ((ssb[0][wave][j] + ssb[1][wave][j]) + ssb[2][wave][j]) + ssb[3][wave][j];
            }
        }
        NPLDA_MSTAMP(10);
        return;
    }
    // ---- score: s = sum_f Q (z1^2 + z2^2) + 2 P z1 z2 (utils/models.py:372-376) ------------------------------------------
    auto term = [&](const f32x4& z1, const f32x4& z2, int b) {  // one block's share: an 8-fma chain of its own
        const f32x4 q = Qp[4 * b + g];
        const f32x4 p = Pp[4 * b + g];
        float part = 0.f;
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            part = fmaf(q[r], fmaf(z1[r], z1[r], z2[r] * z2[r]), part);
            part = fmaf(2.0f * p[r], z1[r] * z2[r], part);
        }
        return part;
    };
    if constexpr (T == 0) {
        // the half tile: a pair's z1 sits in lane j < 8, its z2 in lane j + 8 of the same registers (DPP row_ror:8 swaps them;
        // the term is symmetric, lanes j >= 8 form the same value)
        auto half = [&](const f32x4& z, int b) {
            f32x4 zo;
#pragma unroll
            for (int r = 0; r < 4; ++r) zo[r] = dpp_f32<0x128>(z[r]);
            return term(z, zo, b);
        };
        float ph = half(zF[0][0], blk(0)) + half(zF[1][0], blk(4));
        if (lo_valid) {
#pragma unroll
            for (int i = 0; i < LS; ++i) ph += half(zL[i][0], blk(8 + i));
        }
        ph = wave_xor_add(ph, 16);
        ph = wave_xor_add(ph, 32);
        if (g == 0) scb[wave][0][j] = ph;
        __syncthreads();
        if (wave == 0 && g == 0 && j < 8) {
            const long long row = tile0 * 8 + j;
            if (row < a.n) a.out_s[row] = ((scb[0][0][j] + scb[1][0][j]) + scb[2][0][j]) + scb[3][0][j];
        }
        NPLDA_MSTAMP(10);
        return;
    }
    float part[T > 0 ? T : 1];
#pragma unroll
    for (int t = 0; t < T; ++t)
        part[t] = term(zF[0][2 * t], zF[0][2 * t + 1], blk(0)) + term(zF[1][2 * t], zF[1][2 * t + 1], blk(4));
    if constexpr (C::XCH) {
        // own row group = rho 0 (side wave & 1 of tile' 0); the other side's z sits in wave ^ 1
#pragma unroll
        for (int i = 0; i < LS; ++i) zx[wave][i][lane] = zL[i][0];
        __syncthreads();
        if ((wave & 1) == 0) {
#pragma unroll
            for (int i = 0; i < LS; ++i) part[0] += term(zL[i][0], zx[wave ^ 1][i][lane], blk(8 + i));
        }
    } else {
        // both sides of the left-over units are here: (rho 0, rho 1) of tile' 0
        if (lo_valid) {
#pragma unroll
            for (int i = 0; i < LS; ++i) part[0] += term(zL[i][0], zL[i][LR > 1 ? 1 : 0], blk(8 + i));
        }
    }
#pragma unroll
    for (int t = 0; t < T; ++t) {
        float v = wave_xor_add(part[t], 16);
        v = wave_xor_add(v, 32);
        if (g == 0) scb[wave][t ^ ((swz >> 1) & (T - 1))][j] = v;
    }
    __syncthreads();
    if (wave < T && g == 0) {
        const long long row = tile0 * 8 + wave * 16 + j;
        if (row < a.n) a.out_s[row] = ((scb[0][wave][j] + scb[1][wave][j]) + scb[2][wave][j]) + scb[3][wave][j];
    }
    NPLDA_MSTAMP(10);
}

// Block b works on the contiguous range of HALF tiles [start, start + k): k = c for the first r blocks, c - 1 for the rest
// (c = ceil(nh / grid), r = nh - grid (c - 1), nh = halves of 8 pairs / 16 rows) — four halves as a T = 2 group, two as a
// T = 1 group, an odd last half as a T = 0 group: the grid balances to half a tile (round 5; to a whole tile before: 10 240
// pairs — 2.5 tiles per CU — ran three tiles on half the CUs).
// HALF = false: every block's count is even (the host checks) — the T = 0 group is not instantiated, and the kernel is the
// round-4 one (with it in, the compiler's register assignment for the shared rings costs the even sizes ~1 %).
template <int NB, bool EMBED = false, bool HALF = true, bool IDX = false>
__global__ __launch_bounds__(256, 1) void nplda_fwd_mid_kernel(const FwdArgs a, int c, int r) {
    constexpr int UWM = MidCfg<NB, 2>::UW > MidCfg<NB, 1>::UW ? MidCfg<NB, 2>::UW : MidCfg<NB, 1>::UW;
    __shared__ f32x4 red[4 * 3 * UWM * 64];
    __shared__ float ssb[4][4][16];
    __shared__ float scb[4][2][16];
    __shared__ f32x4 zx[4][3][64];
    __shared__ f32x4 cv[4 * NB * 4];  // b1, b2, Q, P: read after the first group's first barrier
    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int b = blockIdx.x;
    int k = b < r ? c : c - 1;
    long long h = b < r ? (long long)b * c : (long long)r * c + (long long)(b - r) * (c - 1);
    auto group_t = [](int kk) { return kk >= 4 ? 2 : (kk >= 2 ? 1 : 0); };  // the next group of a remaining count of halves
    // the first group's first loads (x steps 0 .. 2, weights step 0)
    MidRing<NB> R;
    {
        MidAddr<NB> A0;
        mid_addr<NB, EMBED, HALF, IDX>(a, h, group_t(k), wave, lane, A0);
#pragma unroll
        for (int i = 0; i < 3; ++i)
#pragma unroll
            for (int rho = 0; rho < 4; ++rho) mid_fetchx<NB>(A0, R, i, rho);
        const __amdgpu_buffer_rsrc_t img = mid_image(a);
#pragma unroll
        for (int s = 0; s < NB; ++s) R.wf[0][s] = mid_ldw(img, A0.voff[s], mid_w1_step<NB>(0, wave));
    }
    for (int i = threadIdx.x; i < 4 * NB * 4; i += 256) {
        const int v = i / (NB * 4), e = i % (NB * 4);
        const size_t o = v == 0 ? a.ob1 : (v == 1 ? a.ob2 : (v == 2 ? a.oQ : a.oP));
        cv[i] = reinterpret_cast<const f32x4*>(a.packed + o)[e];
    }
    // (the block's last group names itself as its successor: a harmless re-read of its own first rows)
    for (; k >= 4; k -= 4, h += 4) {
        const bool more = k > 4;
        mid_group<NB, 2, EMBED, HALF, IDX>(a, h, more ? group_t(k - 4) : 2, more ? h + 4 : h, R, wave, lane, red, ssb, scb, zx, cv);
    }
    if (k >= 2) {
        const bool more = k > 2;
        mid_group<NB, 1, EMBED, HALF, IDX>(a, h, more ? 0 : 1, more ? h + 2 : h, R, wave, lane, red, ssb, scb, zx, cv);
        k -= 2;
        h += 2;
    }
    if constexpr (HALF) {
        if (k == 1) mid_group<NB, 0, EMBED, true, IDX>(a, h, 0, h, R, wave, lane, red, ssb, scb, zx, cv);
    }
}

}  // namespace nplda

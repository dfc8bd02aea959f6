// nplda_fwd_dispatch.h — host-side selection of the forward kernel variant (shared by the .hip files).
#pragma once
#include <cstdlib>
#include "nplda_fwd_kernel.h"
#include "nplda_fwd_small.h"
#include "nplda_fwd_v2.h"
#include "nplda_fwd_v3.h"
#include "nplda_fwd_v5.h"
#include "nplda_fwd_v6.h"
#include "nplda_fwd_mid.h"

namespace nplda {

// Product configuration, chosen by interleaved A/B runs of tools/exp_fwd.hip on MI355X
// (profiles/r01b_*, r01d_*, r01t_*):
//  * large batches, pair scoring, D1 = D2 = 150 (round 5): nplda_fwd_v6.h — v5's schedule with the six left-over features
//    on 4 x 4 x 1 MFMAs instead of a tenth 16-feature block: 0.858 against v3's 0.830;
//  * large batches, pair scoring, NB <= 10 otherwise: the persistent continuous-stream schedule (nplda_fwd_v3.h),
//    8 waves/block, one block per CU, 2 k16-steps of weights per barrier, LDS fragments read 4 feature blocks at a
//    time: 0.82 of the fp32 MFMA peak at D = 150;
//  * large batches, pair scoring, NB = 11 / 12 (D = 170, the reference's shipped size): the same persistent schedule with
//    LDS-DMA weight chunks and a group-major layer 2 (nplda_fwd_v5.h), 171 registers and no spill where v3 spills:
//    0.86 at D = 170 against 0.84 for v2 in the same process (profiles/r02d_*); at NB = 10 it is 1-3 % behind v3;
//  * large batches otherwise (embedding, training mode): the v2 schedule (nplda_fwd_v2.h), 8 waves/block, 2 k16-steps
//    per barrier, x prefetched a whole chunk ahead, plain (cached) x loads: 0.81 at D = 150, 0.84 at D = 170
//    (v1 with non-temporal loads was 0.72 / 0.75);
//  * batches of <= 16 384 pairs: the feature-split small-batch schedule (nplda_fwd_small.h): 4 waves share one
//    16-pair tile, so a 4096-pair training minibatch runs on all 1024 SIMDs (forward 105 us -> see DESIGN.md).
template <int MODE>
static inline int launch_fwd_v2(FwdArgs a, const NpldaLayout& L, hipStream_t st) {
    constexpr int WAVES = 8;
    const long long per_block = (MODE == MODE_EMBED ? 32 : 16) * WAVES;
    const long long blocks = (a.n + per_block - 1) / per_block;
    if (blocks > 0x7fffffffLL) return NPLDA_EINVAL;
    dim3 grid((unsigned)blocks), block(WAVES * 64);
    // 2 k16-steps per barrier (4 at NB = 2, where a 2-step chunk is smaller than one staging pass of the block)
#define NPLDA_LAUNCH(NBV) \
    hipLaunchKernelGGL((nplda_fwd_v2_kernel<NBV, MODE, WAVES, false, (NBV == 2 ? 4 : 2)>), grid, block, 0, st, a)
    switch (L.NB) {
        case 2: NPLDA_LAUNCH(2); break;
        case 4: NPLDA_LAUNCH(4); break;
        case 8: NPLDA_LAUNCH(8); break;
        case 10: NPLDA_LAUNCH(10); break;
        case 11: NPLDA_LAUNCH(11); break;
        case 12: NPLDA_LAUNCH(12); break;
        default: return NPLDA_EUNSUPPORTED;
    }
#undef NPLDA_LAUNCH
    return nplda_launch_status();
}

// persistent grid: one 8-wave block per CU walks the tiles blockIdx.x, + gridDim.x, ...
template <int MODE, int XM = 0>
static inline int launch_fwd_v3(FwdArgs a, const NpldaLayout& L, hipStream_t st) {
    constexpr int WAVES = 8;
    const long long per_block = (MODE == MODE_EMBED ? 32 : 16) * WAVES;
    const long long ntiles = (a.n + per_block - 1) / per_block;
    if (ntiles > 0x7fffffffLL) return NPLDA_EINVAL;
    int dev = 0, cus = 0;
    if (hipGetDevice(&dev) != hipSuccess ||
        hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, dev) != hipSuccess || cus <= 0)
        cus = 256;
    const long long blocks = ntiles < cus ? ntiles : cus;
    dim3 grid((unsigned)blocks), block(WAVES * 64);
#define NPLDA_LAUNCH(NBV) \
    hipLaunchKernelGGL((nplda_fwd_v3_kernel<NBV, MODE, WAVES, XM, (NBV == 2 ? 4 : 2)>), grid, block, 0, st, a, (int)ntiles)
    switch (L.NB) {
        case 2: NPLDA_LAUNCH(2); break;
        case 4: NPLDA_LAUNCH(4); break;
        case 8: NPLDA_LAUNCH(8); break;
        case 10: NPLDA_LAUNCH(10); break;
        default: return NPLDA_EUNSUPPORTED;
    }
#undef NPLDA_LAUNCH
    return nplda_launch_status();
}

// pair scoring at NB = 11 / 12: persistent grid, weight chunks by LDS-DMA, layer 2 by output groups (nplda_fwd_v5.h)
template <int XM = 0>
static inline int launch_fwd_v5(FwdArgs a, const NpldaLayout& L, hipStream_t st) {
    constexpr int WAVES = 8;
    const long long ntiles = (a.n + 16 * WAVES - 1) / (16 * WAVES);
    if (ntiles > 0x7fffffffLL) return NPLDA_EINVAL;
    int dev = 0, cus = 0;
    if (hipGetDevice(&dev) != hipSuccess ||
        hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, dev) != hipSuccess || cus <= 0)
        cus = 256;
    const long long blocks = ntiles < cus ? ntiles : cus;
    dim3 grid((unsigned)blocks), block(WAVES * 64);
    switch (L.NB) {
        case 11: hipLaunchKernelGGL((nplda_fwd_v5_kernel<11, WAVES, 2, 6, 1, XM>), grid, block, 0, st, a, (int)ntiles); break;
        case 12: hipLaunchKernelGGL((nplda_fwd_v5_kernel<12, WAVES, 4, 4, 1, XM>), grid, block, 0, st, a, (int)ntiles); break;
        default: return NPLDA_EUNSUPPORTED;
    }
    return nplda_launch_status();
}

// pair scoring at D1 = D2 = 150 (the headline shape): nine full blocks on the 16 x 16 MFMA, the six left-over features on
// 4 x 4 x 1 MFMAs (nplda_fwd_v6.h) — 0.842 against v3's 0.831 of the fp32 MFMA peak at 1 M pairs
static inline bool fwd_v6_ok(const NpldaLayout& L) { return L.NB == 10 && L.D1 == 150 && L.D2 == 150; }
template <int XM = 0>
static inline int launch_fwd_v6(FwdArgs a, const NpldaLayout& L, hipStream_t st) {
    constexpr int WAVES = 8;
    const long long ntiles = (a.n + 16 * WAVES - 1) / (16 * WAVES);
    if (ntiles > 0x7fffffffLL) return NPLDA_EINVAL;
    if (!fwd_v6_ok(L)) return NPLDA_EUNSUPPORTED;
    int dev = 0, cus = 0;
    if (hipGetDevice(&dev) != hipSuccess ||
        hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, dev) != hipSuccess || cus <= 0)
        cus = 256;
    const long long blocks = ntiles < cus ? ntiles : cus;
    // (2 k16-steps per fence, layer-1 MFMA groups of 5 + 4 blocks, layer-2 output groups of 3: the best of tools/exp_fwd.hip's
    // sweep, profiles/r05x_exp_v6_groups.txt)
    hipLaunchKernelGGL((nplda_fwd_v6_kernel<10, 6, WAVES, 2, 5, 3, XM>), dim3((unsigned)blocks), dim3(WAVES * 64), 0, st, a, (int)ntiles);
    return nplda_launch_status();
}
// the streaming pair kernel of a model: v6 at 150 / 150, v3 up to NB = 10 otherwise, v5 at NB = 11 / 12
template <int XM = 0>
static inline int launch_fwd_stream(FwdArgs a, const NpldaLayout& L, hipStream_t st) {
    // NPLDA_FWD_NO_V6=1: the round-4 dispatch (A/B measurements only)
    static const bool no_v6 = getenv("NPLDA_FWD_NO_V6") != nullptr && getenv("NPLDA_FWD_NO_V6")[0] == '1';
    if (fwd_v6_ok(L) && !no_v6) return launch_fwd_v6<XM>(a, L, st);
    if (L.NB <= 10) return launch_fwd_v3<MODE_PAIR, XM>(a, L, st);
    return launch_fwd_v5<XM>(a, L, st);
}

template <int MODE>
static inline int launch_fwd_small(FwdArgs a, const NpldaLayout& L, hipStream_t st) {
    const long long per_block = (MODE == MODE_EMBED ? 32 : 16);
    const long long blocks = (a.n + per_block - 1) / per_block;
    dim3 grid((unsigned)blocks), block(256);
    // 512-d x-vectors (KS1 = 32) at the recipe sizes: the fully unrolled K loop (see nplda_fwd_small.h)
#define NPLDA_LAUNCH(NBV) hipLaunchKernelGGL((nplda_fwd_small_kernel<NBV, MODE>), grid, block, 0, st, a)
#define NPLDA_LAUNCH32(NBV)                                                                           \
    if (L.KS1 == 32 && L.D0 == 512) hipLaunchKernelGGL((nplda_fwd_small_kernel<NBV, MODE, 32>), grid, block, 0, st, a); \
    else NPLDA_LAUNCH(NBV)
    switch (L.NB) {
        case 2: NPLDA_LAUNCH(2); break;
        case 4: NPLDA_LAUNCH(4); break;
        case 8: NPLDA_LAUNCH(8); break;
        case 10: NPLDA_LAUNCH32(10); break;
        case 11: NPLDA_LAUNCH32(11); break;
        case 12: NPLDA_LAUNCH(12); break;
        default: return NPLDA_EUNSUPPORTED;
    }
#undef NPLDA_LAUNCH32
#undef NPLDA_LAUNCH
    return nplda_launch_status();
}

// pair scoring between the regimes (nplda_fwd_mid.h): balanced contiguous tile ranges, one block per CU
static inline int mid_cus() {
    int dev = 0, cus = 0;
    if (hipGetDevice(&dev) != hipSuccess ||
        hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, dev) != hipSuccess || cus <= 0)
        cus = 256;
    return cus;
}
template <bool EMBED = false>
static inline int launch_fwd_mid(FwdArgs a, const NpldaLayout& L, hipStream_t st) {
    const long long nh = EMBED ? (a.n + 15) / 16 : (a.n + 7) / 8;  // HALF tiles: 8 pairs, or 16 embedding rows
    const int cus = mid_cus();
    // one block per CU; up to one half per CU every block takes ONE half tile (a lone half costs ~8 us + launch where a whole
    // tile costs ~13); above that at least a whole tile per block
    long long grid = nh <= cus ? nh : ((nh + 1) / 2 < cus ? (nh + 1) / 2 : cus);
    if (grid < 1) grid = 1;
    const long long c = (nh + grid - 1) / grid;
    const long long r = nh - grid * (c - 1);
    if (c > 0x7fffffffLL) return NPLDA_EINVAL;
    const bool half = !(r == grid && (c & 1) == 0);  // some block ends on an odd half tile
    const bool idx = a.ia != nullptr;  // rows through an index: a kernel of its own (mid_addr)
#define NPLDA_LAUNCH(NBV, HV, IV) \
    hipLaunchKernelGGL((nplda_fwd_mid_kernel<NBV, EMBED, HV, IV>), dim3((unsigned)grid), dim3(256), 0, st, a, (int)c, (int)r)
#define NPLDA_LAUNCH_H(NBV, HV) do { if (idx) { NPLDA_LAUNCH(NBV, HV, true); } else { NPLDA_LAUNCH(NBV, HV, false); } } while (0)
    switch (L.NB) {
        case 10: if (half) NPLDA_LAUNCH_H(10, true); else NPLDA_LAUNCH_H(10, false); break;
        case 11: if (half) NPLDA_LAUNCH_H(11, true); else NPLDA_LAUNCH_H(11, false); break;
        default: return NPLDA_EUNSUPPORTED;
    }
#undef NPLDA_LAUNCH_H
#undef NPLDA_LAUNCH
    return nplda_launch_status();
}

template <int MODE>
static inline int launch_fwd_old(FwdArgs a, const NpldaLayout& L, hipStream_t st) {
    a.D0 = L.D0; a.KS1 = L.KS1;
    a.oW2 = L.oW2; a.ob1 = L.ob1; a.ob2 = L.ob2; a.oQ = L.oQ; a.oP = L.oP; a.total = L.total;
    const long long units = (MODE == MODE_EMBED ? (a.n + 1) / 2 : a.n);
    if (units <= 256 * 64) return launch_fwd_small<MODE>(a, L, st);  // 4 waves share a 16-pair tile
    // embed / train modes spill in v3 at G = 4; at G = 1 or 2 the embed form fits and was measured: same time as v2
    // (1.2 M rows: 0.72 of the peak either way; the mode is paced by its 0.77 GB of output)
    if constexpr (MODE == MODE_PAIR) return launch_fwd_stream<0>(a, L, st);
    return launch_fwd_v2<MODE>(a, L, st);
}

// Which pair-scoring kernel a batch takes.  Three regimes (profiles/r03*_size_sweep.txt):
//  * one 16-pair tile per CU or less (<= 16 * CUs pairs): the feature-split small-batch kernel;
//  * the streaming kernels (v3 / v5) move 128-pair tiles through a persistent grid: ROUNDS = ceil(tiles128 / CUs) passes
//    of ~100 us (NB = 10) / ~115 us (NB = 11) each, whatever share of the last pass is filled;
//  * the mid kernel (nplda_fwd_mid.h; 512-d x-vectors, NB = 10 / 11) balances 16-pair tiles to within one tile per CU at
//    ~13.3 / 16.8 us per tile — 6 to 25 % more per pair than a FULL streaming round, far less than a part-filled one.
// The two are compared by these measured costs (tenths of a microsecond; both scale with the shader clock alike).
//  * (round 6) FWD_SPLIT: the streaming kernel for the FULL rounds of the persistent grid, the remainder — less than one round —
//    by whichever of the other kernels it would take alone, as a second launch on the same stream: a batch just past a
//    multiple of 128 pairs x CUs no longer pays the mid kernel's per-tile rate on all of it (100 000 pairs: three full rounds
//    + 1 696 pairs; profiles/r06*_size_sweep.txt).
enum { FWD_SMALL = 0, FWD_MID = 1, FWD_STREAM = 2, FWD_SPLIT = 3 };
static inline long long pair_split_point(long long n, int cus) { return (n / 128 / cus) * 128 * cus; }  // pairs in full rounds
static inline int pair_kernel_choice(long long n, const NpldaLayout& L, int cus, long long* cost = nullptr, bool allow_split = true) {
    long long dummy;
    if (!cost) cost = &dummy;
    // NPLDA_FWD_NO_MID=1: the round-2 dispatch (A/B measurements only: tools/validate_e2e.py)
    static const bool no_mid = getenv("NPLDA_FWD_NO_MID") != nullptr && getenv("NPLDA_FWD_NO_MID")[0] == '1';
    const bool mid_ok = !no_mid && (L.NB == 10 || L.NB == 11) && L.D0 == 512 && L.KS1 == 32;
    // up to ONE half tile (8 pairs) per CU: the balanced-tile kernel with a lone T = 0 group per block — 12 us against the
    // small-batch kernel's 16.6 (round 5: half the rows per CU; tools/ab_small_mid.py).  NPLDA_FWD_SMALL_MAX=<pairs> moves the
    // small-batch kernel's upper bound (A/B measurements).
    static const long long small_max = getenv("NPLDA_FWD_SMALL_MAX") ? atoll(getenv("NPLDA_FWD_SMALL_MAX")) : -1;
    const long long round_cost = L.NB == 10 ? 1000 : 1150;
    *cost = 120;
    if (mid_ok && small_max < 0 && n <= 8LL * cus) return FWD_MID;
    *cost = 190;
    if (n <= (small_max >= 0 ? small_max : 16LL * cus)) return FWD_SMALL;
    const long long rounds = ((n + 127) / 128 + cus - 1) / cus;
    const long long t_stream = 50 + rounds * round_cost;
    *cost = n <= 64LL * cus ? 190 * ((n + 16LL * cus - 1) / (16LL * cus)) : t_stream;
    if (!mid_ok) return n <= 64LL * cus ? FWD_SMALL : FWD_STREAM;
    const long long ch = ((n + 7) / 8 + cus - 1) / cus;  // half tiles on the busiest CU (a trailing half costs ~0.7 of a tile)
    const long long t_mid = 40 + (ch / 2) * (L.NB == 10 ? 133 : 168) + (ch & 1) * (L.NB == 10 ? 93 : 118);
    int best = t_mid < t_stream ? FWD_MID : FWD_STREAM;
    *cost = t_mid < t_stream ? t_mid : t_stream;
    // NPLDA_FWD_NO_SPLIT=1: the round-5 dispatch (A/B measurements only)
    static const bool no_split = getenv("NPLDA_FWD_NO_SPLIT") != nullptr && getenv("NPLDA_FWD_NO_SPLIT")[0] == '1';
    const long long full = pair_split_point(n, cus);
    if (allow_split && !no_split && full > 0 && full < n) {
        long long t_rem = 0;
        pair_kernel_choice(n - full, L, cus, &t_rem, false);
        const long long t_split = 50 + (full / 128 / cus) * round_cost + t_rem;
        if (t_split + 30 < *cost) {  // (3 us of margin: a second launch is not free)
            *cost = t_split;
            best = FWD_SPLIT;
        }
    }
    return best;
}

template <int MODE>
static inline int launch_fwd(FwdArgs a, const NpldaLayout& L, hipStream_t st) {
    if constexpr (MODE == MODE_PAIR) {
        const int cus = mid_cus();
        const int k = pair_kernel_choice(a.n, L, cus);
        a.D0 = L.D0; a.KS1 = L.KS1;
        a.oW2 = L.oW2; a.ob1 = L.ob1; a.ob2 = L.ob2; a.oQ = L.oQ; a.oP = L.oP; a.total = L.total;
        if (k == FWD_MID) return launch_fwd_mid<false>(a, L, st);
        if (k == FWD_SMALL) return launch_fwd_small<MODE>(a, L, st);
        if (k == FWD_SPLIT && a.ia == nullptr && a.ib == nullptr) {
            const long long full = pair_split_point(a.n, cus);
            FwdArgs b = a;  // the remainder: rows [full, n) of both sides, scores [full, n)
            b.xa += full * a.ldx; b.xb += full * a.ldx; b.out_s += full; b.n = a.n - full;
            a.n = full;
            if (int rc = launch_fwd_stream<0>(a, L, st)) return rc;
            const int kr = pair_kernel_choice(b.n, L, cus, nullptr, false);
            if (kr == FWD_MID) return launch_fwd_mid<false>(b, L, st);
            if (kr == FWD_SMALL) return launch_fwd_small<MODE>(b, L, st);
            return launch_fwd_stream<0>(b, L, st);
        }
        return launch_fwd_stream<0>(a, L, st);
    }
    if constexpr (MODE == MODE_EMBED) {
        // embedding rows (inference: no saved activations): 32 rows are one tile's worth of work; the balanced-tile kernel
        // between one tile per CU and the streaming sizes (10 000 cohort + 22 000 enroll / test rows of cfg3: 116 -> 60 us)
        if (a.out_y == nullptr && pair_kernel_choice((a.n + 1) / 2, L, mid_cus(), nullptr, false) == FWD_MID) {
            a.D0 = L.D0; a.KS1 = L.KS1;
            a.oW2 = L.oW2; a.ob1 = L.ob1; a.ob2 = L.ob2; a.oQ = L.oQ; a.oP = L.oP; a.total = L.total;
            return launch_fwd_mid<true>(a, L, st);
        }
    }
    return launch_fwd_old<MODE>(a, L, st);
}

// bf16 rows (load_xrow, XM = 2): the streaming kernels only — below their sizes a conversion pass costs next to nothing
static inline int launch_fwd_pairs_bf16rows(FwdArgs a, const NpldaLayout& L, hipStream_t st) {
    // (a batch the fp32 call would split — full rounds streamed, the remainder on the balanced-tile kernel — is streamed whole
    // here: the balanced-tile kernel does not read bf16 rows)
    const int k = pair_kernel_choice(a.n, L, mid_cus());
    if (k != FWD_STREAM && k != FWD_SPLIT) return NPLDA_EUNSUPPORTED;
    a.D0 = L.D0; a.KS1 = L.KS1;
    a.oW2 = L.oW2; a.ob1 = L.ob1; a.ob2 = L.ob2; a.oQ = L.oQ; a.oP = L.oP; a.total = L.total;
    return launch_fwd_stream<2>(a, L, st);
}

// name of the kernel nplda_score_pairs_f32 launches for a batch of n pairs (bench.py labels its roofline object with it)
static inline const char* pair_kernel_name(long long n, const NpldaLayout& L) {
    switch (pair_kernel_choice(n, L, mid_cus())) {
        case FWD_SPLIT:  // (labelled by the kernel that carries the full rounds)
            return pair_kernel_name(pair_split_point(n, mid_cus()), L);
        case FWD_SMALL: return "nplda_fwd_small_kernel (4 waves share a 16-pair tile, feature-split)";
        case FWD_MID: return "nplda_fwd_mid_kernel (balanced 16-pair tiles, K-split layer 1, groups of 2 tiles)";
        default:
            if (fwd_v6_ok(L) && !(getenv("NPLDA_FWD_NO_V6") != nullptr && getenv("NPLDA_FWD_NO_V6")[0] == '1'))
                return "nplda_fwd_v6_kernel (persistent, 9 feature blocks on 16x16x4 MFMAs + 6 features on 4x4x1 MFMAs)";
            return L.NB <= 10 ? "nplda_fwd_v3_kernel (persistent, 8 waves x 16 pairs, weights through LDS)"
                              : "nplda_fwd_v5_kernel (persistent, LDS-DMA weight chunks, layer 2 by output groups)";
    }
}

static inline int check_model(int D0, int D1, int D2) {
    if (D0 <= 0 || D1 <= 0 || D2 <= 0 || (D0 % 4) != 0) return NPLDA_EINVAL;
    if (!nplda_dims_ok(D0, D1, D2)) return NPLDA_EUNSUPPORTED;
    return NPLDA_OK;
}

static inline bool rows_ok(const float* x, int64_t ld, int D) {
    return x && ld >= D && (ld % 4) == 0 && nplda_aligned16(x);
}

}  // namespace nplda

// nplda_cohort_fused.hip — cohort score matrix + per-row statistics in ONE pass, nothing spilled (gfx950).
//
// nplda_cohort.hip forms S = q_r + q_m + 2 (z_r * P) z_m^T on MFMA tiles, writes it out (0.88 GB at BASELINE cfg3) and
// reads it back for the row statistics: 1.76 GB of traffic for 22.5 MB of input.  Here the statistics are taken in the
// GEMM's epilogue, from the accumulators:
//
//  * the mean of a row's M scores is ANALYTIC: S[r, m] = q_r + q_m + a_r . z_m with a_r = 2 P z_r gives
//    mean_m = q_r + mean(q) + a_r . mean(z), formed in fp64 on fp64 sums of the cohort (returned as is); the std needs the
//    sum of squares: accumulated per lane centred on that mean (so that the fp32 partial sums of a tile carry no
//    cancellation), added to fp64 running sums once per list band and row;
//  * the top-N statistics (adaptive_score_normalization.py:32-36: the N smallest) need the N smallest scores of the row
//    exactly.  A threshold t_r slightly above the N-th smallest is PROPOSED from the row's analytic mean and standard
//    deviation — var_m = var(q) + 2 a_r . cov(z, q) + a_r^T cov(z) a_r from the cohort's second moments —
//    and every score <= t_r is appended to a candidate list of that row.  The lists then hold ~2 N of the M scores; a
//    small kernel selects the N smallest among them exactly (ties by count) and sums them in fp64.  The counts decide:
//    a row whose list holds fewer than N scores, or more than fit, is recomputed by the exact general path
//    (cohort_fallback_kernel in nplda_cohort.hip) — the proposal never changes a result, only who computes it.
//  * no atomics on data, no run-to-run variation: a work item is (row tile of 256 rows, one or all list bands of a column
//    band); ONE block walks the item's tiles in order, so every lane meets "its" columns of "its" rows in a fixed order
//    and appends to a private sub-list (row, list band, lane group): the lists' contents, their order and all sums are
//    independent of scheduling and of where a row sits in the table.
//
// Pre-pass (four small launches): Gram matrix of the cohort table by the split-K wgrad kernel (nplda_backward.hip) with
// the first moments (sums of z, q in fp64; q^2, q z) as extra work items of the same launch; centred covariance folded
// with 2 P into a fragment image; (z_rows . C'') by the resident-matrix GEMM (nplda_matmul.hip); one wave per row forms
// the row's mean and t_r.

#include <cstdlib>
#include <type_traits>

#include "nplda_cohort_common.h"
#include "nplda_cohort_fused.h"
#include "nplda_cohort_qz.h"

namespace nplda {
int gram_slabs_launch(const float* Z, long long ldz, long long rows, int Mp, int ksplit, float* slab, float* ext,
                      const QzArgs* qz, hipStream_t st);
}  // namespace nplda

#ifdef NPLDA_COHORT_ABLATE
static unsigned long long* g_cf_stamps = nullptr;
extern "C" __attribute__((visibility("default"))) int nplda_cohort_debug_stamps(unsigned long long* host_out) {
    if (!g_cf_stamps) return -1;
    return (int)hipMemcpy(host_out, g_cf_stamps, 2 * 64 * 8 * sizeof(unsigned long long), hipMemcpyDeviceToHost);
}
#endif

namespace {

constexpr int kSubSlack = 17;     // a sub-list of ksub slots counts as overflowed at ksub - 17 (one tile appends <= 16 to it)
constexpr int kParts = 4;         // list bands per column band when the band is divisible (see cohort_fused_plan)
constexpr int kCandMax = 2048;    // candidates one row may bring to the select kernel (32 keys per lane)
constexpr int kGramSplit = 16;    // k-groups of the cohort Gram matrix
constexpr int kQzBlocks = 64;

// ------------------------------------------------------------------------------------------------------------------
// split form of the fused GEMM (round 6): fp32 operands as three bf16 pieces, six v_mfma_f32_16x16x32_bf16 passes
// ------------------------------------------------------------------------------------------------------------------
// gfx950 has no TF32, and its fp32-input MFMA runs at 1/16 of the bf16 rate.  v = h + m + l with h = bf16(v),
// m = bf16(v - h), l = bf16(v - h - m) (the residuals are exact in fp32; |m| <= 2^-9 |v|, |l| <= 2^-18 |v|, what is left of v
// is below 2^-27 |v|), and a product x y is taken as hh + hm + mh + hl + lh + mm — each bf16 x bf16 product is exact in the
// MFMA's fp32 accumulate, the dropped terms (ml, lm, ll) are below 2^-26 |x y|: the same products to fp32 rounding, in 6 x 16
// cycles per 16 x 16 x 32 block instead of 8 x 32.  The cohort is split ONCE (cohort_split_kernel, into the fragment order
// the tile DMA wants: a tile's fragment is 1 KiB contiguous), a row tile's operands when the item is taken.
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
__host__ __device__ constexpr int split_ksteps(int Mp) { return (Mp + 31) / 32; }
__host__ __device__ constexpr size_t split_tile_bytes(int Mp) { return (size_t)12 * split_ksteps(Mp) * 1024; }

__device__ __forceinline__ void split3(float v, __bf16& h, __bf16& m, __bf16& l) {
    h = (__bf16)v;
    const float r1 = v - (float)h;
    m = (__bf16)r1;
    const float r2 = r1 - (float)m;
    l = (__bf16)r2;
}
__device__ __forceinline__ void split3x8(const f32x4 lo, const f32x4 hi, bf16x8& H, bf16x8& M, bf16x8& L) {
#pragma unroll
    for (int e = 0; e < 8; ++e) {
        __bf16 h, m, l;
        split3(e < 4 ? lo[e] : hi[e - 4], h, m, l);
        H[e] = h; M[e] = m; L[e] = l;
    }
}

struct SplitArgs {
    const float* zc; long long M, ldz; int Mp; unsigned char* img;  // [tiles of 64 columns][ks][c][piece][lane][8 bf16]
};
// one thread per (tile, ks, c, lane): eight values of one cohort row -> three 16-byte pieces.  Rows past M and columns past Mp
// are zero (the last tile needs no clamped addressing in the fused kernel; its columns are masked in the epilogue as before).
__global__ __launch_bounds__(256) void cohort_split_kernel(const SplitArgs a) {
    const int NK = split_ksteps(a.Mp);
    const long long idx = (long long)blockIdx.x * 256 + threadIdx.x;
    const long long nt64 = (a.M + 63) / 64;
    if (idx >= nt64 * NK * 4 * 64) return;
    const int lane = (int)(idx & 63), c = (int)((idx >> 6) & 3);
    const long long rest = idx >> 8;
    const int ks = (int)(rest % NK);
    const long long t = rest / NK;
    const long long m = t * 64 + 16 * c + (lane & 15);
    const int k0 = 32 * ks + 8 * (lane >> 4);
    f32x4 lo = {0.f, 0.f, 0.f, 0.f}, hi = lo;
    if (m < a.M) {
        const float* zp = a.zc + m * a.ldz + k0;
        if (k0 < a.Mp) lo = *reinterpret_cast<const f32x4*>(zp);
        if (k0 + 4 < a.Mp) hi = *reinterpret_cast<const f32x4*>(zp + 4);
    }
    bf16x8 H, Mm, L;
    split3x8(lo, hi, H, Mm, L);
    bf16x8* out = reinterpret_cast<bf16x8*>(a.img + (size_t)t * split_tile_bytes(a.Mp)) + (size_t)((ks * 4 + c) * 3) * 64 + lane;
    out[0] = H; out[64] = Mm; out[128] = L;
}

// ------------------------------------------------------------------------------------------------------------------
// pre-pass
// ------------------------------------------------------------------------------------------------------------------

struct PrepArgs {
    const float* slab;   // [ksplit][Mp][Mp]
    const float* ext;    // [ksplit][4][Mp]  (row 3: column sums of z)
    const float* qz;     // [kQzBlocks][Mp + 2]
    const double* qz64;  // [kQzBlocks][Mp + 1]: sum z, sum q in fp64
    const float* P;      // padded, zero beyond D2
    int ksplit, Mp;
    long long M;
    float* frag;         // [KB (KB + 1) / 2][64][4]: blocks kb <= xb of C''[i][j] = 4 P_i P_j cov(z)_ij, off-diagonal blocks x 2
    float* vec;          // [0, Mp): u = 2 P mean(z); [Mp, 2 Mp): v = 4 P cov(z, q); [2 Mp]: mean(q); [2 Mp + 1]: var(q)
    double* vec64;       // [0, Mp): 2 P mean(z), [Mp]: mean(q) — fp64, for the row means the call returns
    unsigned* ctl;       // the call's 64-word control block (work-item counters, fail count): zeroed here, one launch less
};

__global__ __launch_bounds__(256) void cohort_prep_kernel(const PrepArgs a) {
    __shared__ double zbar[NPLDA_MAX_DIM];
    const int Mp = a.Mp, KB = Mp / 16;
    const size_t idx = (size_t)blockIdx.x * 256 + threadIdx.x;
    const double n = (double)a.M;
    if (idx < 64) a.ctl[idx] = 0u;
    if ((int)threadIdx.x < Mp) {  // every block forms the cohort mean for itself (kQzBlocks x Mp L2-resident doubles)
        double s = 0.0;
#pragma unroll 16
        for (int b = 0; b < kQzBlocks; ++b) s += a.qz64[(size_t)b * (Mp + 1) + threadIdx.x];
        zbar[threadIdx.x] = s / n;
    }
    __syncthreads();
    __shared__ double qsh[NPLDA_MAX_DIM + 2];
    if (blockIdx.x == 0) {  // sums of q z, q, q^2 over the qz kernel's blocks, fixed order (only block 0 needs them)
        for (int i = threadIdx.x; i < Mp + 2; i += 256) {
            double sacc = 0.0;
#pragma unroll 16
            for (int b = 0; b < kQzBlocks; ++b) sacc += (double)a.qz[(size_t)b * (Mp + 2) + i];
            qsh[i] = sacc;
        }
        __syncthreads();
    }
    auto qsum = [&](int i) { return qsh[i]; };
    if (idx < (size_t)KB * KB * 256) {
        const int e = (int)(idx & 3), lane = (int)((idx >> 2) & 63);
        const size_t blk = idx >> 8;
        const int xb = (int)(blk % KB), kb = (int)(blk / KB);
        const int i = 16 * kb + 4 * (lane >> 4) + e, j = 16 * xb + (lane & 15);
        double g = 0.0;
        for (int k = 0; k < a.ksplit; ++k) g += (double)a.slab[((size_t)k * Mp + i) * Mp + j];
        const double cov = g / n - zbar[i] * zbar[j];
        // C'' is symmetric and only its quadratic form is needed: blocks kb <= xb are kept (packed at xb (xb + 1) / 2 + kb),
        // the off-diagonal ones doubled — 45 % fewer MFMAs in cohort_rowthr_kernel
        if (kb <= xb) {
            const size_t tri = (size_t)xb * (xb + 1) / 2 + kb;
            a.frag[(tri * 64 + lane) * 4 + e] = (float)((kb < xb ? 8.0 : 4.0) * (double)a.P[i] * (double)a.P[j] * cov);
        }
    }
    if (idx < (size_t)Mp) {
        const int i = (int)idx;
        const double zm = zbar[i], qm = qsum(Mp) / n;
        a.vec[i] = (float)(2.0 * (double)a.P[i] * zm);
        a.vec64[i] = 2.0 * (double)a.P[i] * zm;
        a.vec[Mp + i] = (float)(4.0 * (double)a.P[i] * (qsum(i) / n - qm * zm));
        if (i == 0) {
            double sq = 0.0;
#pragma unroll 16
            for (int b = 0; b < kQzBlocks; ++b) sq += a.qz64[(size_t)b * (Mp + 1) + Mp];
            a.vec64[Mp] = sq / n;
            a.vec[2 * Mp] = (float)qm;
            double vq = qsum(Mp + 1) / n - qm * qm;
            a.vec[2 * Mp + 1] = (float)(vq > 0.0 ? vq : 0.0);
        }
    }
}

// Per row r: its mean over the cohort (fp64, returned), c_r = that mean in fp32, and the candidate threshold
// t_r = c_r + sgn * zhi * sd_r with sd_r^2 = var(q) + z_r . v + z_r^T C'' z_r (zhi < 0: below the mean for the N smallest,
// sgn = +1; above it for the N largest, sgn = -1).  The quadratic form is a (16 rows) x (Mp x Mp) MFMA product per wave with
// the covariance image resident in LDS: the accumulators come out in the layout of the operand rows themselves (lane
// (j, g) holds features 16 nb + 4 g + r of row j — the k-permutation of the forward kernels), so z_r . (C'' z_r) is an
// in-lane dot product and two row exchanges.  (This was a generic rows x matrix GEMM into a scratch table and a second
// kernel reading it back: 32 + 10 us and 28 MB of traffic for what is 7 us of MFMA work.)
struct RowThrArgs {
    const float* zr; const float* qr; long long R, ldz;
    const float* frag;    // [NB (NB + 1) / 2][64][4]: block (kb <= nb) at nb (nb + 1) / 2 + kb: C''[16 kb + 4 g + e][16 nb + i16],
                          // off-diagonal blocks doubled (C'' is symmetric: the quadratic form needs one triangle)
    const float* vec;     // see PrepArgs
    const double* vec64;
    float zhi, sgn;
    float* crow; float* trow; double* mean64;
    int ntiles;           // tiles of 16 wpt rows
    int wpt;              // waves of a block that take a 16-row group (8: 128-row tiles; fewer for calls with few rows, so that
                          // a 2 750-row shard is 86 blocks instead of 22 — the other waves only help to load the image)
    unsigned* ctl_zero;   // a prepared cohort (no pre-pass to do it): the call's 64-word control block, zeroed by block 0 here
};

template <int NB>
__global__ __launch_bounds__(512, 1) void cohort_rowthr_kernel(const RowThrArgs a) {
    extern __shared__ f32x4 cimg[];  // NB (NB + 1) / 2 fragments of 64 lanes
    const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int j = lane & 15, g = lane >> 4;
    constexpr int Mp = 16 * NB;
    if (a.ctl_zero != nullptr && blockIdx.x == 0 && tid < 64) a.ctl_zero[tid] = 0u;
    // v = 4 P cov(z, q) and the fp64 cohort mean through LDS as well: as global loads in the epilogue they were thirty
    // load - s_waitcnt vmcnt(0) - use round trips per 16-row group (hipcc threads them through the MFMA loop one at a time; the
    // memory counter is in order, an LDS read has its own) — round 6
    __shared__ float vlin_s[Mp];
    __shared__ double vec64_s[Mp + 1];
    for (int i = tid; i < NB * (NB + 1) / 2 * 64; i += 512) cimg[i] = reinterpret_cast<const f32x4*>(a.frag)[i];
    for (int i = tid; i < Mp; i += 512) vlin_s[i] = a.vec[Mp + i];
    for (int i = tid; i <= Mp; i += 512) vec64_s[i] = a.vec64[i];
    __syncthreads();
    for (int tile = blockIdx.x; tile < a.ntiles; tile += gridDim.x) {
        const long long r0 = ((long long)tile * a.wpt + wave) * 16;
        if (wave >= a.wpt || r0 >= a.R) continue;  // wave-uniform
        const long long row = r0 + j < a.R ? r0 + j : a.R - 1;
        const float* zp = a.zr + row * a.ldz + 4 * g;
        f32x4 zf[NB], acc[NB];
#pragma unroll
        for (int kb = 0; kb < NB; ++kb) {
            zf[kb] = *reinterpret_cast<const f32x4*>(zp + 16 * kb);
            acc[kb] = f32x4{0.f, 0.f, 0.f, 0.f};
        }
#pragma unroll
        for (int kb = 0; kb < NB; ++kb) {
#pragma unroll
            for (int nb = kb; nb < NB; ++nb) {
                const f32x4 av = cimg[(nb * (nb + 1) / 2 + kb) * 64 + lane];
#pragma unroll
                for (int r = 0; r < 4; ++r)
                    acc[nb] = __builtin_amdgcn_mfma_f32_16x16x4f32(av[r], zf[kb][r], acc[nb], 0, 0, 0);
            }
        }
        float quad = 0.f, lin = 0.f;
        double mu = 0.0;  // the mean is an OUTPUT (stats[.][0]): fp64 on the fp64 cohort mean
#pragma unroll
        for (int nb = 0; nb < NB; ++nb) {
            const f32x4 v = *reinterpret_cast<const f32x4*>(vlin_s + 16 * nb + 4 * g);
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                quad = fmaf(zf[nb][r], acc[nb][r], quad);
                lin = fmaf(zf[nb][r], v[r], lin);
                mu = fma((double)zf[nb][r], vec64_s[16 * nb + 4 * g + r], mu);
            }
        }
        quad = wave_xor_add(quad, 16); quad = wave_xor_add(quad, 32);
        lin = wave_xor_add(lin, 16); lin = wave_xor_add(lin, 32);
        mu += __hiloint2double(__shfl_xor(__double2hiint(mu), 16, 64), __shfl_xor(__double2loint(mu), 16, 64));
        mu += __hiloint2double(__shfl_xor(__double2hiint(mu), 32, 64), __shfl_xor(__double2loint(mu), 32, 64));
        if (g == 0 && r0 + j < a.R) {
            const double m64 = (double)a.qr[row] + vec64_s[Mp] + mu;
            a.mean64[row] = m64;
            const float mean = (float)m64;
            const float var = a.vec[2 * Mp + 1] + lin + quad;
            const float sd = sqrtf(fmaxf(var, 0.f));
            a.crow[row] = mean;
            a.trow[row] = mean + a.sgn * a.zhi * sd;
        }
    }
}

// ------------------------------------------------------------------------------------------------------------------
// the fused GEMM: rows in registers, only the cohort goes through LDS, statistics epilogue instead of stores
// ------------------------------------------------------------------------------------------------------------------
struct FusedArgs {
    const float* zr; const float* qr; const float* zc; const float* qc; const float* P;
    const unsigned char* zc3;     // the cohort as three bf16 pieces in fragment order (cohort_split_kernel); SPLIT kernels only
    long long R, M, ldz;
    int ksteps, nbands, ny, nx;   // nbands: a multiple of 8; band b -> XCD b % 8, column tiles [b nx / nbands, (b + 1) nx / nbands)
    unsigned* ctr;          // 8 work-item counters (one per XCD), zero at launch
    const float* crow;      // (R)
    const float* trow;      // (R)
    float* lists;           // [R][nlb][ksub][4], nlb = nbands q list bands: slot e of sub-list (lb, g) of row r at ((r nlb + lb) ksub + e) 4 + g.
                            // The 128-byte lines of a (row, band) region are written by ONE wave, while its block works
                            // through the band: they fill up in that XCD's L2 and go to memory once.  (A slot-major
                            // [R][kSub][nsub] layout — every line shared by all bands, i.e. by blocks on all XCDs at
                            // different times — turned each 4-byte append into its own partial-line write-back: 575 MB
                            // of HBM writes for 88 MB of candidates, profiles/r02n.)
    unsigned* counts;       // [R][nsub]
    double* part;           // [R][nsub / 4]: sum over the list band's columns of (s - c_r)^2, c_r = the row's mean in fp32
    int nsub;               // nlb * 4 (lane groups)
    int lrow;               // floats between the list regions of two rows: nsub * ksub + the plan's padding (fused_row_pad)
    int prio;               // static priority 1 for the block's second-dispatched waves (4 .. 7): they lose every VALU
                            // arbitration to the older half otherwise; -0.8 % on the statistics call in interleaved runs
                            // (NPLDA_COHORT_PRIO=0 switches it off for A/B)
#ifdef NPLDA_COHORT_ABLATE  // tools/exp_cohort.sh only: timing ablations (results are then garbage) and per-tile cycle stamps
    int abl;                // 1: no epilogue, 2: no MFMA loop, 4: no end-of-tile barrier, 8: no tile DMA after the first
    unsigned long long* stamps;  // [2 waves (0 and NW / 2)][64 tiles][8]
#endif
    int q, ksub, nfull;     // list bands per band, slots per sub-list; row tiles [0, nfull) are handed out as whole bands,
                            // the rest one list band at a time (the tail of the work queue in quarters)
};

// The first form of this kernel kept the tile pipeline of the spilling GEMM (cohort_gemm_kernel): both operands staged
// through LDS one k16 stage at a time, a barrier per stage — 830-880 us at cfg3, 0.62 of the fp32-MFMA peak, the same as
// the GEMM it replaced.  But here a block OWNS its rows for a whole band of column tiles, which is the situation of the
// forward kernel (nplda_fwd_v3.h): the per-row operand can live in REGISTERS for the
// whole work item (a wave's 32 rows x K <= 192: 2 x NB float4 per lane, pre-multiplied by 2 P) and only the cohort
// streams — as whole 64-column tiles in MFMA-fragment order (the LDS-DMA takes a per-lane source address, so one wave
// instruction drops a finished 1 KiB A-fragment into LDS), double buffered, ONE barrier per tile instead of eleven,
// a quarter of the L2 -> LDS traffic per MFMA (a tile serves 256 rows, and no row operand is staged at all), and the
// epilogue works on 32 accumulator registers at a time.
// RGW: 16-row groups per wave — 2 (a block owns 256 rows: two MFMAs per A fragment read from LDS) or 1 (128 rows: twice as
// many work items of half the size, for calls with so few rows that 256-row items leave CUs idle — an 8-way row shard of
// cfg3 is 11 such tiles for 256 CUs).  A row's lists and sums are formed by the same lanes in the same column order either
// way: the results do not depend on it, bit for bit.
// KT: k4-steps of the LAST k16-block (round 5).  D2 = 150 leaves six values there; in the table's column order
// (k = 16 ks + 4 g4 + r) they sit in lane groups 0 and 1 and still need all four steps.  With KT < 4 both operands take the
// block as k = 16 (NB - 1) + KT g4 + r, r < KT — the LDS-DMA reads the cohort row's 16 bytes from column 16 (NB - 1) + KT g4
// (4- / 8-byte aligned: accepted, tools/exp_dma_align.hip), the row operand is loaded to match — and the block runs KT
// steps: 38 instead of 40 k4-steps at D2 = 150 (KT = 2), 43 instead of 44 at D2 = 170 (KT = 3).  Columns >= D2 are zero in
// the table on both sides, so nothing is masked.  The scores' last-block terms associate differently from the spilling
// GEMM's: same values to rounding (tolerance in the tests).
template <bool LOWEST, int NB, int RGW = 2, int KT = 4, int NW = 8, bool SPLIT = false>
__global__ __launch_bounds__(64 * NW, 1) void cohort_fused2_kernel(const FusedArgs a) {
    static_assert(KT >= 2 && KT <= 4, "the k16-step's first two k4-steps always run");
    static_assert(NW == 8 || NW == 16, "two or four waves per SIMD");
    static_assert(!SPLIT || (KT == 4 && NW == 8), "the split form has no short last block and two waves per SIMD");
    constexpr int RPB = NW * 16 * RGW;  // rows of a block's row tile
    constexpr int NK = split_ksteps(16 * NB);  // SPLIT: k32-steps
    // 1 KiB fragments of a 64-column tile.  fp32: [ks][c], lane (i16, g4) = column 16 c + i16, k 16 ks + 4 g4 ..;
    // SPLIT: [ks][c][piece], lane (i16, g4) = column 16 c + i16, eight bf16 of k 32 ks + 8 g4 ..
    constexpr int NF = SPLIT ? 12 * NK : 4 * NB;
    __shared__ f32x4 smem[2 * NF * 64 + 2 * 16 + NB * 4 + 2];
    f32x4* tbuf = smem;
    float* qms = reinterpret_cast<float*>(smem + 2 * NF * 64);                 // q_m of the two buffered tiles
    f32x4* p2s = smem + 2 * NF * 64 + 32;                                      // 2 P as fragment-shaped float4
    unsigned* nxt_s = reinterpret_cast<unsigned*>(smem + 2 * NF * 64 + 32 + NB * 4);
    const int tid = threadIdx.x;
    const int lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int i16 = lane & 15, g4 = lane >> 4;
    const int xcd = blockIdx.x & 7;
    const int nt64 = (int)((a.M + 63) / 64);
    // Work items of XCD x, in queue order: for each of its column bands (band = kb * 8 + x) first the row tiles [0, nfull)
    // with the WHOLE band (q list bands walked back to back, the row operands fetched once), then the remaining row
    // tiles one list band at a time.  A block takes ~ny * nbands / grid items; with whole bands only, 86 row tiles on the
    // 32 CUs of an XCD are 2.69 items per CU and the kernel lasts 3 — the last round in quarters ends after 2.75.
    const int nlb = a.nbands * a.q;
    const int per_kb = a.nfull + (a.ny - a.nfull) * a.q;
    auto lb_tile = [&](int lb) { return (int)((long long)lb * nt64 / nlb); };  // list bands in whole 64-column tiles
    auto decode = [&](int slot, long long& rb, int& lb0, int& lbn) {
        const int kb = slot / per_kb;
        const int band = kb * 8 + xcd;
        if (band >= a.nbands) return false;
        const int r = slot - kb * per_kb;
        if (r < a.nfull) {
            rb = (long long)r * RPB;
            lb0 = band * a.q;
            lbn = lb0 + a.q;
        } else {
            const int r2 = r - a.nfull;
            rb = (long long)(a.nfull + r2 / a.q) * RPB;
            lb0 = band * a.q + r2 % a.q;
            lbn = lb0 + 1;
        }
        return true;
    };
    // LDS-DMA of the 64-column tile `t` into buffer `buf`: wave w fills fragments w, w + NW, ...  The source address of lane
    // (i16, g4) for fragment (ks, c) of tile t is zc + (64 t + 16 c + i16) ldz + col(ks, g4): everything but the 64 t ldz term
    // is fixed per lane and fragment — kept as byte offsets in NFW registers, the tile term is a scalar base (the per-tile
    // 64-bit address arithmetic was 19 VALU instructions per fragment, 95 per tile and wave: matrix-pipe time, see above).
    // Only the cohort's LAST tile, when M is not a multiple of 64, clamps rows and takes the general form.
    // WHO issues it: ONE half of the block (waves [NW / 2, NW)), at the top of the tile.  The CU's address unit takes ~50
    // cycles per DMA instruction (64 lanes x 16 bytes from 16 rows); with every wave issuing its pieces at the top of the
    // tile, BOTH waves of a SIMD stood in that queue and the matrix pipe idled 1.1 k cycles per tile before the first loop
    // (cycle stamps: tools/exp_cohort_stamps.py, profiles/r06*_stamps.txt).  Now the other half starts its MFMA loop at once;
    // the issuing half's pieces go out beside that loop (it would be waiting for the pipe anyway), its own loop follows.
    // (Issued BEHIND the loop instead: 25.3 k cycles per tile against 23.9 k — the second epilogue of the tile then runs
    // twice as long.)
    constexpr int NWD = NW / 2;                       // waves that issue the DMA
    constexpr int NFW = (NF + NWD - 1) / NWD;
    const bool dma_wave = wave >= NW - NWD;
    const int dw = wave - (NW - NWD);
    unsigned doff[NFW];
#pragma unroll
    for (int j = 0; j < NFW; ++j) {
        const int f = (dma_wave ? dw : 0) + NWD * j, ks = f >> 2, c = f & 3;
        const int col = (KT < 4 && ks == NB - 1) ? 16 * ks + KT * g4 : 16 * ks + 4 * g4;
        doff[j] = 4u * (unsigned)((long long)(16 * c + i16) * a.ldz + col);
    }
    const unsigned qoff = 4u * (unsigned)lane;
    // (The lambdas of this kernel are always_inline: they capture the register arrays by reference, and ONE that hipcc decides
    // not to inline puts brow3 / acc into scratch — seen in the ABLATE build of an experiment: 480 bytes of scratch, 10 x the time.)
    auto tile_in = [&](int t, int buf) __attribute__((always_inline)) {
        if (!dma_wave) return;
        if constexpr (SPLIT) {
            // the split image is in fragment order: fragment f of tile t is the 1 KiB at (t NF + f) 1024 — a scalar base and ONE
            // lane offset for all of them, and every DMA instruction reads whole lines.  (Who issues them does not matter to the
            // tile's time — the first half taking 3 / 5 / 7 / 10 of the 15 fragments per wave in front of its loop:
            // 400 - 415 us in every setting, profiles/r06h_split_dma_share_ab.txt — as in the fp32 form it is the second half.)
            const char* sb = reinterpret_cast<const char*>(a.zc3) + (size_t)t * (NF * 1024);
#pragma unroll
            for (int j = 0; j < NFW; ++j) {
                const int f = dw + NWD * j;
                if (f < NF)
                    __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)((sb + f * 1024) + 4 * qoff),
                                                     (__attribute__((address_space(3))) void*)&tbuf[(buf * NF + f) * 64], 16, 0, 0);
            }
            if (dw == 0) {
                long long m = (long long)t * 64 + lane;
                if (m >= a.M) m = a.M - 1;
                __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(a.qc + m),
                                                 (__attribute__((address_space(3))) void*)&qms[buf * 64], 4, 0, 0);
            }
            return;
        }
        if ((long long)t * 64 + 64 <= a.M) {
            const char* sb = reinterpret_cast<const char*>(a.zc) + (size_t)t * 64 * (size_t)a.ldz * 4;
#pragma unroll
            for (int j = 0; j < NFW; ++j) {
                const int f = dw + NWD * j;
                if (f < NF)
                    __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(sb + doff[j]),
                                                     (__attribute__((address_space(3))) void*)&tbuf[(buf * NF + f) * 64], 16, 0, 0);
            }
            if (dw == 0) {
                const char* qb = reinterpret_cast<const char*>(a.qc) + (size_t)t * 256;
                __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(qb + qoff),
                                                 (__attribute__((address_space(3))) void*)&qms[buf * 64], 4, 0, 0);
            }
            return;
        }
        for (int f = dw; f < NF; f += NWD) {
            const int ks = f >> 2, c = f & 3;
            long long m = (long long)t * 64 + 16 * c + i16;
            if (m >= a.M) m = a.M - 1;
            const int col = (KT < 4 && ks == NB - 1) ? 16 * ks + KT * g4 : 16 * ks + 4 * g4;
            __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(a.zc + m * a.ldz + col),
                                             (__attribute__((address_space(3))) void*)&tbuf[(buf * NF + f) * 64], 16, 0, 0);
        }
        if (dw == 0) {
            long long m = (long long)t * 64 + lane;
            if (m >= a.M) m = a.M - 1;
            __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(a.qc + m),
                                             (__attribute__((address_space(3))) void*)&qms[buf * 64], 4, 0, 0);
        }
    };

#ifdef NPLDA_COHORT_ABLATE
    const int abl = a.abl;
    int stamp_tile = 0;
    const bool stamper = blockIdx.x == 8 && lane == 0 && (wave == 0 || wave == NW / 2);
#define NPLDA_CF_STAMP(i)                                                                                              \
    do {                                                                                                               \
        if (stamper && stamp_tile < 64) a.stamps[((wave ? 1 : 0) * 64 + stamp_tile) * 8 + (i)] = __builtin_readcyclecounter(); \
    } while (0)
#else
    constexpr int abl = 0;
#define NPLDA_CF_STAMP(i) do { } while (0)
#endif
    if (a.prio == 1 && wave >= NW / 2) __builtin_amdgcn_s_setprio(1);
    if (SPLIT && a.prio == 2) __builtin_amdgcn_s_setprio(3);
    long long rb = 0, nrb = 0;
    int band = 0, lbn = 0, t = 0, t1 = 0, nlb0 = 0, nlbn = 0;  // `band`: the list band being filled; the item ends at lbn
    if (tid == 0) nxt_s[0] = atomicAdd(a.ctr + xcd, 1u);
    if (tid < 4 * NB) p2s[tid] = 2.0f * *reinterpret_cast<const f32x4*>(a.P + 4 * tid);
    __syncthreads();
    if (!decode(__builtin_amdgcn_readfirstlane((int)nxt_s[0]), rb, band, lbn)) return;
    t = lb_tile(band);
    t1 = lb_tile(band + 1);
    if (tid == 0) nxt_s[1] = atomicAdd(a.ctr + xcd, 1u);  // the item after this one, asked for a whole item ahead
    int npar = 1;                                          // which word holds the next item's slot

    // per-lane state of a work item: the lane's two rows (wave's row group A / B, row i16), their operand fragments
    f32x4 brow[RGW][SPLIT ? 1 : NB];
    bf16x8 brow3[RGW][SPLIT ? NK : 1][3];   // SPLIT: (2 P z_r) in three bf16 pieces, lane (i16, g4): row i16, k 32 ks + 8 g4 ..
    // kq = q_r - c_r and thc = t_r - c_r: the epilogue forms the CENTRED score d = acc + (q_m + kq) (= s - c_r to rounding),
    // squares it, compares it with thc and appends IT — the select kernel adds c_r back in fp64.  (Every VALU instruction of
    // this kernel costs matrix-pipe time — on a SIMD fp32 MFMAs and VALU instructions do not overlap, whichever wave issues
    // them: tools/exp_mfma_yield.hip, profiles/r06e_mfma_yield.txt — so the uncentred score is never formed: 4 instructions
    // per accumulator block less.)
    float kq[RGW], thc[RGW], s2[RGW];
    unsigned cur[RGW], lim[RGW];
    // ALL of an item's row loads are on their way before the first one is waited for (BATCH): left to itself hipcc fetched each
    // fragment into one 4-register temporary — load, s_waitcnt vmcnt(0), multiply, next load — twenty dependent round trips,
    // the 14 k cycles by which a tile that ends an item was longer than one that does not (round 6: seen in the ISA after the
    // same pattern turned up in nplda_fwd_mid.h; the round's stamps had put it down to the memory system).  D >= 173 with two
    // row groups sits at 255 registers and spills in that form: it keeps the compiler's order.
    constexpr bool BATCH = RGW == 1 || NB <= 10 || (NB == 11 && KT < 4);
    auto item_rows = [&](long long rb_) __attribute__((always_inline)) {
        if constexpr (SPLIT) {
            const float* p2f = reinterpret_cast<const float*>(p2s);
#pragma unroll
            for (int g = 0; g < RGW; ++g) {
                long long row = rb_ + wave * (16 * RGW) + 16 * g + i16;
                if (row >= a.R) row = a.R - 1;
                const float* zp = a.zr + row * a.ldz + 8 * g4;
                f32x4 raw[NK][2];
#pragma unroll
                for (int ks = 0; ks < NK; ++ks) {   // (columns past 16 NB: not part of the table's contract — zero pieces)
                    const bool in = 32 * ks + 16 < 16 * NB || g4 < 2;
                    raw[ks][0] = in ? *reinterpret_cast<const f32x4*>(zp + 32 * ks) : f32x4{0.f, 0.f, 0.f, 0.f};
                    raw[ks][1] = in ? *reinterpret_cast<const f32x4*>(zp + 32 * ks + 4) : f32x4{0.f, 0.f, 0.f, 0.f};
                }
                __builtin_amdgcn_sched_barrier(0);   // all of a row group's loads in flight before the first wait
#pragma unroll
                for (int ks = 0; ks < NK; ++ks) {
                    const bool in = 32 * ks + 16 < 16 * NB || g4 < 2;
                    const int kc = in ? 32 * ks + 8 * g4 : 0;
                    const f32x4 p0 = *reinterpret_cast<const f32x4*>(p2f + kc), p1 = *reinterpret_cast<const f32x4*>(p2f + kc + 4);
                    split3x8(raw[ks][0] * p0, raw[ks][1] * p1, brow3[g][ks][0], brow3[g][ks][1], brow3[g][ks][2]);
                }
            }
            return;
        }
        const float* zrow[RGW];
#pragma unroll
        for (int g = 0; g < RGW; ++g) {
            long long row = rb_ + wave * (16 * RGW) + 16 * g + i16;
            if (row >= a.R) row = a.R - 1;
            zrow[g] = a.zr + row * a.ldz;
            const f32x4* zp = reinterpret_cast<const f32x4*>(zrow[g] + 4 * g4);
#pragma unroll
            for (int ks = 0; ks < NB; ++ks) {
                if constexpr (BATCH) brow[g][ks] = zp[4 * ks];
                else brow[g][ks] = zp[4 * ks] * p2s[4 * ks + g4];
            }
        }
        if constexpr (BATCH) __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int g = 0; g < RGW; ++g) {
            if constexpr (BATCH) {
#pragma unroll
                for (int ks = 0; ks < NB; ++ks) brow[g][ks] *= p2s[4 * ks + g4];
            }
            if (KT < 4) {  // the last block in the KT-step order: columns 16 (NB - 1) + KT g4 + r
                const float* zl = zrow[g] + 16 * (NB - 1) + KT * g4;
                const float* pl = reinterpret_cast<const float*>(p2s) + 16 * (NB - 1) + KT * g4;
                f32x4 v = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
                for (int r = 0; r < KT; ++r) v[r] = zl[r] * pl[r];
                brow[g][NB - 1] = v;
            }
        }
    };
    // Per-lane state in two layers.  Per ITEM (a row tile): kq, thc and the byte offset of the row's list region — loaded
    // once per item, and for the NEXT item already at the top of this item's last tile (next_consts: the loads ride under
    // that tile's MFMA loop).  Per LIST BAND: the sum of squares, the append cursor and its limit.  (Cycle stamps: a tile that
    // ended a list band took 28 k cycles instead of 25 k — the row constants were re-loaded from memory at every list band.
    // A tile that ends an ITEM takes 43 - 48 k: 256 blocks pull 164 KB of row operands each at about the same time, 42 MB
    // against HBM / the memory-side cache; warming the L2 a tile ahead does not help — an XCD's 32 row tiles are 5.2 MB, more
    // than its L2 — and neither does staggering the queue so that blocks change items at different times: both measured,
    // r06j / r06k.  With the row loads left out (ablation 4096) such a tile takes 28 k: the loads are the cost.  Loading the
    // raw rows behind the loop and multiplying by 2 P only at the top of the next tile — so that the epilogue, the barrier
    // and the partner's loop lie between load and first use — moved the wait without shortening it: 34 - 40 k, and the
    // plain tiles got 0.5 k slower: r06q, not kept.)
    unsigned rowbase[RGW];
    float nraw[RGW][3];
    auto next_consts = [&](long long rb_) __attribute__((always_inline)) {
#pragma unroll
        for (int g = 0; g < RGW; ++g) {
            long long rc = rb_ + wave * (16 * RGW) + 16 * g + i16;
            if (rc >= a.R) rc = a.R - 1;
            nraw[g][0] = a.crow[rc];
            nraw[g][1] = a.trow[rc];
            nraw[g][2] = a.qr[rc];
        }
    };
    auto take_consts = [&](long long rb_) __attribute__((always_inline)) {
#pragma unroll
        for (int g = 0; g < RGW; ++g) {
            const long long row = rb_ + wave * (16 * RGW) + 16 * g + i16;
            const bool ok = row < a.R;
            const long long rc = ok ? row : a.R - 1;
            const float cen = nraw[g][0];
            kq[g] = nraw[g][2] - cen;
            thc[g] = ok ? nraw[g][1] - cen : (LOWEST ? -__builtin_inff() : __builtin_inff());  // rows past the table never append
            rowbase[g] = 4u * (unsigned)(rc * a.lrow + g4);
        }
    };
    auto band_state = [&](int band_) __attribute__((always_inline)) {
#pragma unroll
        for (int g = 0; g < RGW; ++g) {
            s2[g] = 0.f;
            cur[g] = rowbase[g] + 16u * (unsigned)(band_ * a.ksub);
            // at most ksub - kSubSlack entries stay (the select kernel treats that count as an overflow)
            lim[g] = cur[g] + 16u * (unsigned)(a.ksub - kSubSlack);
        }
    };
    auto item_end = [&](long long rb_, int band_) __attribute__((always_inline)) {
#pragma unroll
        for (int g = 0; g < RGW; ++g) {
            const long long row = rb_ + wave * (16 * RGW) + 16 * g + i16;
            // the four lane groups of a row: fixed association ((g0 + g1) + (g2 + g3)) by two exchanges
            double u2 = (double)s2[g];
            u2 += __hiloint2double(__shfl_xor(__double2hiint(u2), 16, 64), __shfl_xor(__double2loint(u2), 16, 64));
            u2 += __hiloint2double(__shfl_xor(__double2hiint(u2), 32, 64), __shfl_xor(__double2loint(u2), 32, 64));
            if (row < a.R) {
                const unsigned sidx = (unsigned)(band_ * 4 + g4);
                a.counts[(size_t)row * a.nsub + sidx] = (cur[g] - rowbase[g]) / 16u - (unsigned)(band_ * a.ksub);
                if (g4 == 0) {
                    a.part[(size_t)row * (a.nsub / 4) + band_] = u2;
                }
            }
        }
    };

    item_rows(rb);
    next_consts(rb);
    take_consts(rb);
    band_state(band);
    tile_in(t, 0);
    __syncthreads();  // (drains the DMA: it is a pending LDS write)
    int buf = 0;
    const unsigned stride_b = 16u;  // bytes between consecutive slots of a sub-list
    const float* lbase = a.lists;

    for (;;) {
        const bool last_tile = t + 1 == t1;                       // of the list band
        const bool last_of_item = last_tile && band + 1 == lbn;
        bool have_next = true;
        // next tile into the other buffer: of this item (its list bands are contiguous), or the first of the next one
        NPLDA_CF_STAMP(0);
        int tnext = t + 1;
        if (last_of_item) {
            have_next = decode(__builtin_amdgcn_readfirstlane((int)nxt_s[npar]), nrb, nlb0, nlbn);
            tnext = have_next ? lb_tile(nlb0) : -1;
            if (have_next) next_consts(nrb);
        }
        // next tile into the other buffer (all waves left it at the previous barrier): of this item (its list bands are
        // contiguous), or the first of the next one
        if (tnext >= 0 && !(abl & 8)) tile_in(tnext, buf ^ 1);
        NPLDA_CF_STAMP(1);
        f32x4 acc[RGW][4];
#pragma unroll
        for (int g = 0; g < RGW; ++g)
#pragma unroll
            for (int c = 0; c < 4; ++c) acc[g][c] = f32x4{0.f, 0.f, 0.f, 0.f};
        const f32x4* tb = tbuf + buf * NF * 64 + lane;
        if (SPLIT && a.prio == 2) __builtin_amdgcn_s_setprio(0);   // (the MFMA loop at priority 0 ...)
        // The A fragments of the next k16-step are read in the MIDDLE of this step's MFMAs (two register sets, the reads
        // pinned by scheduling barriers: hipcc otherwise sinks them to just before their use, and it waits lgkmcnt(0)
        // there).  The two waves of a SIMD run this loop in step, so an LDS round trip that is not covered by 16 MFMAs
        // stalls the matrix pipe for both at once (measured: the loop alone ran at 0.77 of the pipe's rate).
        if constexpr (SPLIT) {
            // A step is one k32-block of CG column groups: 3 CG fragment reads feed 6 CG RGW MFMAs (four independent
            // accumulator chains: a 16x16x32 MFMA issues every 16 cycles, its result is ready ~2.5 issue slots later).  Small
            // terms first: mm, hl, lh, hm, mh, hh.  The next step's fragments are read behind this step's first pass.
            constexpr int CG = 4 / RGW, SPK = 4 / CG, NSTEP = NK * SPK;
            const bf16x8* tb3 = reinterpret_cast<const bf16x8*>(tb);
            bf16x8 af3[2][CG][3];
#pragma unroll
            for (int cc = 0; cc < CG; ++cc)
#pragma unroll
                for (int p = 0; p < 3; ++p) af3[0][cc][p] = tb3[(cc * 3 + p) * 64];
            if (!(abl & 2))
#pragma unroll
            for (int s = 0; s < NSTEP; ++s) {
                const int ks = s / SPK, c0 = (s % SPK) * CG;
#define NPLDA_CF_PASS(PA, PB)                                                                                          \
    _Pragma("unroll") for (int cc = 0; cc < CG; ++cc) _Pragma("unroll") for (int g = 0; g < RGW; ++g)                 \
        acc[g][c0 + cc] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(af3[s & 1][cc][PA], brow3[g][ks][PB], acc[g][c0 + cc], 0, 0, 0)
                NPLDA_CF_PASS(1, 1);
                __builtin_amdgcn_sched_barrier(0);
                if (s + 1 < NSTEP) {
                    const int ks1 = (s + 1) / SPK, c1 = ((s + 1) % SPK) * CG;
#pragma unroll
                    for (int cc = 0; cc < CG; ++cc)
#pragma unroll
                        for (int p = 0; p < 3; ++p) af3[(s + 1) & 1][cc][p] = tb3[((ks1 * 4 + c1 + cc) * 3 + p) * 64];
                }
                __builtin_amdgcn_sched_barrier(0);
                NPLDA_CF_PASS(0, 2);
                NPLDA_CF_PASS(2, 0);
                NPLDA_CF_PASS(0, 1);
                NPLDA_CF_PASS(1, 0);
                NPLDA_CF_PASS(0, 0);
                // (the end of a step is a scheduling barrier too: without it hipcc hoists the next step's first MFMAs — the ones
                // that wait for the fragments just asked for — to four MFMAs behind the reads, and every step exposes most of an
                // LDS round trip: read in the assembly.  Reads after the first pass: twenty MFMAs between request and first use.)
                __builtin_amdgcn_sched_barrier(0);
#undef NPLDA_CF_PASS
            }
        } else {
        f32x4 af[2][4];
#pragma unroll
        for (int c = 0; c < 4; ++c) af[0][c] = tb[c * 64];
        if (!(abl & 2))
#pragma unroll
        for (int ks = 0; ks < NB; ++ks) {
#pragma unroll
            for (int kk = 0; kk < 2; ++kk)
#pragma unroll
                for (int c = 0; c < 4; ++c) {
#pragma unroll
                    for (int g = 0; g < RGW; ++g)
                        acc[g][c] = __builtin_amdgcn_mfma_f32_16x16x4f32(af[ks & 1][c][kk], brow[g][ks][kk], acc[g][c], 0, 0, 0);
                }
            __builtin_amdgcn_sched_barrier(0);
            if (ks + 1 < NB && !(abl & 256)) {
#pragma unroll
                for (int c = 0; c < 4; ++c) af[(ks + 1) & 1][c] = tb[((ks + 1) * 4 + c) * 64];
            }
            __builtin_amdgcn_sched_barrier(0);
#pragma unroll
            for (int kk = 2; kk < (ks == NB - 1 ? KT : 4); ++kk)
#pragma unroll
                for (int c = 0; c < 4; ++c) {
#pragma unroll
                    for (int g = 0; g < RGW; ++g)
                        acc[g][c] = __builtin_amdgcn_mfma_f32_16x16x4f32(af[ks & 1][c][kk], brow[g][ks][kk], acc[g][c], 0, 0, 0);
                }
        }
        }
        // (No barrier here.  Cycle stamps show the two waves of a SIMD falling into alternation — one in its MFMA loop while
        // the other works through its epilogue, whose VALU instructions then find issue slots only between the other
        // wave's MFMAs: ~13 cycles each, 29.6 k cycles per tile against 23.4 k for the MFMA loops alone.  Forcing the
        // phases together with a barrier at this point gives the same tile time (24.6 k + 4.6 k) and a kernel 2 % slower.)
        // the operand rows of the NEXT item are fetched here, under the epilogue of this item's last tile
        NPLDA_CF_STAMP(2);
        if (SPLIT && a.prio == 2) __builtin_amdgcn_s_setprio(3);   // (... everything else of the tile at 3: tools/exp_mfma_yield.hip)
        if (last_of_item && have_next && !(abl & 4096)) item_rows(nrb);  // (abl 4096: keeps the old rows — timing only)

        // ---- statistics epilogue: lane (i16, g4) of (g, c) holds row 16 g + i16 of the wave, columns 16 c + 4 g4 + r ----
        const long long m0 = (long long)t * 64;
        const float* qm_s = qms + buf * 64 + 4 * g4;
        auto epilogue = [&](auto masked) __attribute__((always_inline)) {
            constexpr bool MASKED = decltype(masked)::value;
            typedef float f32x2 __attribute__((ext_vector_type(2)));
#pragma unroll
            for (int g = 0; g < RGW; ++g) {
                const float th = thc[g], kq_ = kq[g];
                f32x2 pq2 = {s2[g], 0.f};
                unsigned o = cur[g];
#pragma unroll
                for (int c = 0; c < 4; ++c) {
                    const f32x4 qmv = *reinterpret_cast<const f32x4*>(qm_s + 16 * c);
                    f32x4 d4 = acc[g][c] + (qmv + kq_);   // the score centred on the row's analytic mean (fp32 rounding of it)
                    f32x4 s4 = d4;                          // what is compared and appended
                    if (MASKED) {
#pragma unroll
                        for (int r = 0; r < 4; ++r) {
                            const bool okc = m0 + 16 * c + 4 * g4 + r < a.M;
                            d4[r] = okc ? d4[r] : 0.f;
                            s4[r] = okc ? s4[r] : (LOWEST ? __builtin_inff() : -__builtin_inff());
                        }
                    }
                    const f32x2 dl = {d4[0], d4[1]}, dh = {d4[2], d4[3]};
                    pq2 = __builtin_elementwise_fma(dl, dl, pq2);
                    pq2 = __builtin_elementwise_fma(dh, dh, pq2);
                    // for r in 0..3: if (s[r] <= th) { lists[o] = s[r]; o += stride; }   (>= for the N largest)
                    // as exec-masked stores: no branch, no 64-bit address arithmetic (SGPR base + 32-bit byte offset), one
                    // VALU for the cursor.  The four compares are issued first, into four SGPR pairs: one compare ->
                    // s_and_saveexec -> store chain per value made the epilogue a string of VALU -> SALU round trips
                    // (~25 cycles each, 256 per tile and wave).
                    // (Appending whole float4 groups with one v_cmpx-masked 16-byte store per group was tried: the
                    // epilogue got 2 k cycles per tile shorter, but the lists tripled, the kernel as a whole did not
                    // get faster and the select kernel had to re-filter: 1.10 ms against 1.04 ms for the pipeline.)
                    unsigned long long sv, k0, k1, k2, k3;
#if defined(NPLDA_EPI_VARIANT) && NPLDA_EPI_VARIANT == 1   // experiment: compares and cursors, no stores
#define NPLDA_ST(o_, v_)
#else
#define NPLDA_ST(o_, v_) "global_store_dword " o_ ", " v_ ", %[base]\n\t"
#endif
#if defined(NPLDA_EPI_VARIANT) && NPLDA_EPI_VARIANT == 2   // experiment: sums only
#define NPLDA_APPEND4(CMP) asm volatile("" : [o] "+v"(o) : [th] "v"(th), [s0] "v"(s4[0]), [s1] "v"(s4[1]), [s2] "v"(s4[2]), [s3] "v"(s4[3]) : "memory")
#else
#define NPLDA_APPEND4(CMP)                                                                                         \
    asm volatile(CMP " %[k0], %[s0], %[th]\n\t" CMP " %[k1], %[s1], %[th]\n\t" CMP " %[k2], %[s2], %[th]\n\t"       \
                 CMP " %[k3], %[s3], %[th]\n\t"                                                                    \
                 "s_mov_b64 %[sv], exec\n\t"                                                                       \
                 "s_mov_b64 exec, %[k0]\n\t" NPLDA_ST("%[o]", "%[s0]") "v_add_u32 %[o], %[o], %[st]\n\t"                     \
                 "s_mov_b64 exec, %[k1]\n\t" NPLDA_ST("%[o]", "%[s1]") "v_add_u32 %[o], %[o], %[st]\n\t"                     \
                 "s_mov_b64 exec, %[k2]\n\t" NPLDA_ST("%[o]", "%[s2]") "v_add_u32 %[o], %[o], %[st]\n\t"                     \
                 "s_mov_b64 exec, %[k3]\n\t" NPLDA_ST("%[o]", "%[s3]") "v_add_u32 %[o], %[o], %[st]\n\t"                     \
                 "s_mov_b64 exec, %[sv]"                                                                            \
                 : [o] "+v"(o), [sv] "=&s"(sv), [k0] "=&s"(k0), [k1] "=&s"(k1), [k2] "=&s"(k2), [k3] "=&s"(k3)       \
                 : [th] "v"(th), [s0] "v"(s4[0]), [s1] "v"(s4[1]), [s2] "v"(s4[2]), [s3] "v"(s4[3]), [base] "s"(lbase), \
                   [st] "s"(stride_b)                                                                               \
                 : "memory")
#endif
                    if (LOWEST) NPLDA_APPEND4("v_cmp_le_f32");
                    else NPLDA_APPEND4("v_cmp_ge_f32");
#undef NPLDA_APPEND4
#undef NPLDA_ST
                }
                s2[g] = pq2[0] + pq2[1];
                cur[g] = o < lim[g] ? o : lim[g];
            }
        };
#ifdef NPLDA_COHORT_ABLATE
        if (abl & (16 | 32 | 64 | 128)) {  // synthetic epilogues: what the partner of an MFMA loop gets issued
            float v[8];
#pragma unroll
            for (int i = 0; i < 8; ++i) v[i] = acc[0][i & 3][0] + (float)i;
            const float aa = kq[0], bb = thc[0];
            if (abl & 16) {  // 256 VALU, ONE dependent chain
#pragma unroll
                for (int i = 0; i < 256; ++i) asm volatile("v_fmac_f32 %0, %1, %2" : "+v"(v[0]) : "v"(aa), "v"(bb));
            }
            if (abl & 32) {  // 256 VALU, eight independent chains
#pragma unroll
                for (int i = 0; i < 256; ++i) asm volatile("v_fmac_f32 %0, %1, %2" : "+v"(v[i & 7]) : "v"(aa), "v"(bb));
            }
            if (abl & 64) {  // 256 VALU, two chains
#pragma unroll
                for (int i = 0; i < 256; ++i) asm volatile("v_fmac_f32 %0, %1, %2" : "+v"(v[i & 1]) : "v"(aa), "v"(bb));
            }
            if (abl & 128) {  // 64 x (SALU exec write + VALU under it)
                unsigned long long sv;
                asm volatile("s_mov_b64 %0, exec" : "=s"(sv));
#pragma unroll
                for (int i = 0; i < 64; ++i)
                    asm volatile("s_mov_b64 exec, %1\n\tv_fmac_f32 %0, %2, %3" : "+v"(v[i & 7]) : "s"(sv), "v"(aa), "v"(bb));
            }
            float sink = 0.f;
#pragma unroll
            for (int i = 0; i < 8; ++i) sink += v[i];
            if (sink == 1.2345e30f) s2[0] += sink;
        } else
#endif
        if (abl & 1) {  // (keeps the accumulators alive)
            float sink = 0.f;
#pragma unroll
            for (int g = 0; g < RGW; ++g)
#pragma unroll
                for (int c = 0; c < 4; ++c) sink += acc[g][c][0] + acc[g][c][1] + acc[g][c][2] + acc[g][c][3];
            if (sink == 1.2345e30f) s2[0] += sink;
        } else if (m0 + 64 > a.M) epilogue(std::true_type{});
        else epilogue(std::false_type{});
        NPLDA_CF_STAMP(3);

        if (last_tile) {
            item_end(rb, band);
            if (!last_of_item) {  // next list band of the same item: same rows, the tiles go on
                ++band;
                ++t;
                t1 = lb_tile(band + 1);
                band_state(band);
            } else {
                if (!have_next) break;
                rb = nrb; band = nlb0; lbn = nlbn; t = lb_tile(band); t1 = lb_tile(band + 1);
                take_consts(rb);
                band_state(band);
                npar ^= 1;
                if (tid == 0) nxt_s[npar] = atomicAdd(a.ctr + xcd, 1u);  // visible after the barrier below; read >= 1 tile later
            }
        } else {
            ++t;
        }
        // end of tile: this wave's part of the next tile has landed (and its appends are out), then everybody's; nobody
        // still reads the buffer the tile after next will overwrite
        NPLDA_CF_STAMP(4);
        // The DMA pieces are older than the tile's 16 RGW append instructions (exec-masked stores count whatever their mask),
        // and vector memory operations retire in order: on a plain tile vmcnt(16 RGW) says "my pieces have landed" without
        // waiting for the appends to be acknowledged.  Tiles that end a list band also load / store per-row state: vmcnt(0).
        if (last_tile || (abl & (1 | 1024))) __builtin_amdgcn_s_waitcnt(0x0070);      // vmcnt(0) lgkmcnt(0)
        else __builtin_amdgcn_s_waitcnt(RGW == 2 ? 0x8070 : 0x4070);                  // vmcnt(32 | 16) lgkmcnt(0)
        NPLDA_CF_STAMP(5);
        if (!(abl & 4)) __builtin_amdgcn_s_barrier();
        NPLDA_CF_STAMP(6);
#ifdef NPLDA_COHORT_ABLATE
        ++stamp_tile;
#endif
        buf ^= 1;
    }
}

// ------------------------------------------------------------------------------------------------------------------
// select: one wave per row
// ------------------------------------------------------------------------------------------------------------------
struct FinishArgs {
    const float* lists; const unsigned* counts; const double* part; const float* crow; const double* mean64;
    long long R, M;
    const float* trow;
    float zhi, fhi;                       // the proposal: t_r = c_r + sgn zhi sd_r, fhi = Phi(zhi) = proposed fraction
    int nsub, nsub_valid, topn, lowest;   // sub-lists [nsub_valid, nsub) belong to bands past the last column tile: never written
    int ksub, cap;                        // cap: keys one row may bring (<= kCandMax); more -> the fail list
    int lrow;                             // floats between two rows' list regions (FusedArgs::lrow)
    unsigned* nfail; unsigned* fail_rows;
    double* stats;
};

__global__ __launch_bounds__(256) void cohort_finish_kernel(const FinishArgs a) {
    // this wave's run of candidate keys: a.cap of them (dynamic LDS, 4 a.cap words per block).  The run is sized by the plan
    // (what the proposal expects + a wide margin, at most kCandMax) instead of always kCandMax: the kernel lives on the number
    // of rows in flight, and 32 KiB per block kept a CU at 20 waves
    extern __shared__ unsigned keys_dyn[];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const long long row = (long long)blockIdx.x * 4 + wave;
    if (row >= a.R) return;
    unsigned* kl = keys_dyn + (size_t)wave * a.cap;
    const unsigned* cnt = a.counts + (size_t)row * a.nsub;
    // everything else the row needs from memory is asked for now, next to the counts: at their places of use (the
    // bracket, the very end) each of these loads was a memory round trip of its own at the tail of the wave's life
    const float c_row = a.crow[row], t_row = a.trow[row];
    const double m64_row = a.mean64[row];
    double d2 = 0.0;
    {
        const double* pr = a.part + (size_t)row * (a.nsub / 4);
        for (int i = lane; i < a.nsub_valid / 4; i += 64) d2 += pr[i];
    }
    // counts (lane l holds sub-lists l, l + 64, ...) -> total, overflow, exclusive prefix in sub-list order
    unsigned creg[4], pre[4];
    unsigned total = 0, ovf = 0;
#pragma unroll
    for (int j = 0; j < 4; ++j) {
        const int s = 64 * j + lane;
        creg[j] = s < a.nsub_valid ? cnt[s] : 0u;
        ovf |= creg[j] >= (unsigned)(a.ksub - kSubSlack) ? 1u : 0u;
        unsigned inc = creg[j];  // inclusive scan over the 64 lanes
#pragma unroll
        for (int d = 1; d < 64; d <<= 1) {
            const unsigned t = __shfl_up(inc, d, 64);
            if (lane >= d) inc += t;
        }
        pre[j] = total + inc - creg[j];
        total += __shfl(inc, 63, 64);
    }
    ovf = wave_sum_u32(ovf);
    long long N = a.topn;
    if (N > a.M) N = a.M;
    if (N < 1) N = 1;
    if (ovf != 0 || total < (unsigned)N || total > (unsigned)a.cap) {
        if (lane == 0) a.fail_rows[atomicAdd(a.nfail, 1u)] = (unsigned)row;
        return;
    }
    // gather the sub-lists (lane = sub-list) into this wave's LDS run as order-preserving keys: element e of sub-list s
    // lands at pre[s] + e — a fixed order
    const float* lrow = a.lists + (size_t)row * a.lrow;  // [list band][slot][g]: sub-list s = lb 4 + g
    // Two groups of 64 sub-lists are fetched together (8 slots of each per batch): with 128 short sub-lists per row
    // (list bands) the gather is a chain of memory round trips, and one batch per group doubled their number.
    auto gather2 = [&](int j0) {
        unsigned maxc = creg[j0] > creg[j0 + 1] ? creg[j0] : creg[j0 + 1];
        maxc = wave_max_u32(maxc);
        const float* lsub[2];
#pragma unroll
        for (int h = 0; h < 2; ++h) {
            const unsigned s_ = 64 * (j0 + h) + lane;
            const unsigned sl = s_ < (unsigned)a.nsub ? s_ : (unsigned)a.nsub - 1;  // stay in the row's region
            lsub[h] = lrow + (size_t)(sl >> 2) * a.ksub * 4 + (sl & 3);
        }
        for (unsigned e0 = 0; e0 < maxc; e0 += 8) {
            float v[2][8];
#pragma unroll
            for (int h = 0; h < 2; ++h)
#pragma unroll
                for (int u = 0; u < 8; ++u) {
                    // unconditional (slot index clamped into the sub-list): a load under a lane predicate is a branch with
                    // its own wait, and the slot rows of a row would then come in one memory round trip each
                    const unsigned e = e0 + u < (unsigned)a.ksub ? e0 + u : (unsigned)a.ksub - 1;
                    v[h][u] = lsub[h][(size_t)e * 4];
                }
#pragma unroll
            for (int h = 0; h < 2; ++h)
#pragma unroll
                for (int u = 0; u < 8; ++u) {
                    const unsigned e = e0 + u;
                    if (e < creg[j0 + h]) kl[pre[j0 + h] + e] = f2key(a.lowest ? v[h][u] : -v[h][u]);
                }
        }
    };
    gather2(0);
    if (a.nsub_valid > 128) gather2(2);
    __builtin_amdgcn_s_waitcnt(0xC07F);  // this wave's LDS writes have landed (no other wave touches kl)
    unsigned k[kCandMax / 64];
    unsigned kmin = 0xffffffffu, kmax = 0u;
#pragma unroll
    for (int j = 0; j < kCandMax / 64; ++j) {
        const unsigned i = lane + 64 * j;
        k[j] = i < total ? kl[i] : 0xffffffffu;
        if (i < total) {
            kmin = k[j] < kmin ? k[j] : kmin;
            kmax = k[j] > kmax ? k[j] : kmax;
        }
    }
    unsigned lo = wave_min_u32(kmin), hi = wave_max_u32(kmax);
    const int J = (int)((total + 63) / 64);  // wave-uniform: only the occupied register slots are compared
    // wave-wide count of keys <= p on the scalar unit: one v_cmp per register, popcount of its lane mask
    auto count_le = [&](unsigned p) {
        unsigned c = 0;
#pragma unroll
        for (int j = 0; j < kCandMax / 64; ++j)
            if (j < J) c += (unsigned)__builtin_popcountll(__ballot(k[j] <= p));
        return c;
    };
    // Shortcut.  A bisection over the candidates' key range takes ~23 counts of ~17 registers each (the scalar unit
    // bounds it: 95 of this kernel's 156 us).  Instead: the list's own size calibrates the normal model of the row's
    // tail (the model predicted fhi M candidates below t_r, `total` came), that gives the value expected at rank N and
    // the local density; two counts check a bracket of ~+-96 ranks around it, its <= 256 keys are compacted (in their
    // fixed order) and the bisection runs on four registers per lane.  The counts decide; a miss takes the full search.
    unsigned rank = (unsigned)N;
    {
        const float c = 0.f, t = t_row - c_row;  // the lists hold CENTRED scores s - c_r (cohort_fused2_kernel's epilogue)
        const float sgn = a.lowest ? 1.f : -1.f;
        const float sd = (t - c) / (sgn * a.zhi);
        const float q = a.fhi * (float)N / (float)total;
        const float z0 = fast_normcdfinv(q);
        const float rho = (float)total / a.fhi * 0.3989423f * __expf(-0.5f * z0 * z0) / sd;  // candidates per unit score
        const float T0 = sgn * c + z0 * sd, delta = 96.f / rho;
        unsigned p1 = f2key(T0 - delta), p2 = f2key(T0 + delta);
        if (sd > 0.f && rho > 0.f && p1 < p2 && p2 < 0xffffffffu) {
            const unsigned c1 = count_le(p1), c2 = count_le(p2);
            if (c1 < (unsigned)N && (unsigned)N <= c2 && c2 - c1 <= 256u) {
                unsigned base = 0;
#pragma unroll
                for (int j = 0; j < kCandMax / 64; ++j) {
                    if (j < J) {
                        const bool in = k[j] > p1 && k[j] <= p2;
                        const unsigned long long m = __ballot(in);
                        const unsigned pos = base + __builtin_amdgcn_mbcnt_hi((unsigned)(m >> 32), __builtin_amdgcn_mbcnt_lo((unsigned)m, 0u));
                        if (in) kl[pos] = k[j];
                        base += (unsigned)__builtin_popcountll(m);
                    }
                }
                __builtin_amdgcn_s_waitcnt(0xC07F);
                unsigned kk[4];
#pragma unroll
                for (int j = 0; j < 4; ++j) kk[j] = (unsigned)(lane + 64 * j) < base ? kl[lane + 64 * j] : 0xffffffffu;
                rank = (unsigned)N - c1;
                lo = p1 + 1;
                hi = p2;
                while (lo < hi) {
                    const unsigned mid = lo + (hi - lo) / 2;
                    unsigned cc = 0;
#pragma unroll
                    for (int j = 0; j < 4; ++j) cc += (unsigned)__builtin_popcountll(__ballot(kk[j] <= mid));
                    if (cc >= rank) hi = mid;
                    else lo = mid + 1;
                }
                rank = 0;  // done
            }
        }
    }
    if (rank != 0) {
        while (lo < hi) {
            const unsigned mid = lo + (hi - lo) / 2;
            if (count_le(mid) >= (unsigned)N) hi = mid;
            else lo = mid + 1;
        }
    }
    const unsigned tkey = lo;  // key of the N-th smallest score
    double t1 = 0.0, t2 = 0.0;
    unsigned nless = 0;
#pragma unroll
    for (int j = 0; j < kCandMax / 64; ++j) {
        if (j < J && k[j] < tkey) {
            const float w = key2f(k[j]);
            const double v = (double)(a.lowest ? w : -w);  // the centred score behind the ordered key
            t1 += v;
            t2 += v * v;
            ++nless;
        }
    }
    t1 = wave_sum_f64(t1);
    t2 = wave_sum_f64(t2);
    nless = wave_sum_u32(nless);
    // whole-row sum of squares: the list bands' partials (one per lane, loaded at the top) in their fixed order
    d2 = wave_sum_f64(d2);
    if (lane == 0) {
        const double n = (double)a.M, nn = (double)N;
        // mean: analytic, fp64 (cohort_threshold_kernel); the squares were centred on its fp32 rounding c_r
        const double mean_s = m64_row;
        const double mw = mean_s - (double)c_row;
        double var = d2 / n - mw * mw;
        if (var < 0.0) var = 0.0;
        double tv = (double)key2f(tkey);           // the threshold in ordered space -> centred score
        if (!a.lowest) tv = -tv;
        const double ties = nn - (double)nless;
        t1 += ties * tv;
        t2 += ties * tv * tv;
        const double mt = t1 / nn;                 // of the centred scores: the top-N mean is c_r + mt, their variance is vt
        double vt = t2 / nn - mt * mt;
        if (vt < 0.0) vt = 0.0;
        double* o = a.stats + row * 4;
        o[0] = mean_s;
        o[1] = sqrt(var);
        o[2] = (double)c_row + mt;
        o[3] = sqrt(vt);
    }
}

inline size_t align256(size_t v) { return (v + 255) / 256 * 256; }

}  // namespace

namespace nplda {

// inverse normal CDF (Abramowitz & Stegun 26.2.23), host side
// Floats of padding behind a row's list region.  Without it two rows' regions are nsub * ksub * 4 B = a power of two
// apart (32 KB at cfg3) and the lines a block's 256 rows are appending to compete for the same few L2 sets: they were
// written back half filled and re-opened — 436 MB of L2 write-backs per cfg3 call for 94 MB of candidates.  An odd number of
// 128-byte lines between rows: 264 - 268 MB for 160 / 224 / 288 / 544 / 1056 / 2080 floats (312 at 32, 343 at 64);
// tools/exp_rowpad.sh, profiles/r05q_rowpad.txt.  (Times: fused kernel unchanged, select kernel 83 -> 80 us.)
// The split form of the fused kernel (three bf16 pieces, six passes) serves the two shipped dimensions' block counts.  The
// cohort's split image is ALWAYS built for them (pre-pass / CohortState: ~5 us), so that the choice of kernel can be made per
// call: NPLDA_COHORT_SPLIT=0 selects the fp32-input MFMA form (read at every call — bench.py times both forms in one
// process, tests/test_cohort_fused_gpu.py compares them).
static bool split_supported(int ksteps) { return ksteps == 10 || ksteps == 11; }
static bool split_selected(int ksteps) {
    const char* e = getenv("NPLDA_COHORT_SPLIT");
    return split_supported(ksteps) && !(e && e[0] == '0');
}

static int fused_row_pad() {
    static const int pad = [] {
        const char* e = getenv("NPLDA_COHORT_ROWPAD");  // (A/B runs)
        const int v = e ? atoi(e) : 160;
        return v < 0 ? 0 : (v + 3) / 4 * 4;
    }();
    return pad;
}

static float host_normcdfinv(double p) {
    const bool lower = p < 0.5;
    const double pp = lower ? p : 1.0 - p;
    const double t = sqrt(-2.0 * log(pp < 1e-300 ? 1e-300 : pp));
    const double num = 2.515517 + t * (0.802853 + t * 0.010328);
    const double den = 1.0 + t * (1.432788 + t * (0.189269 + t * 0.001308));
    const double z = t - num / den;
    return (float)(lower ? -z : z);
}

FusedPlan cohort_fused_plan(long long M, int topn, int Mp) {
    FusedPlan p = {};
    p.eligible = false;
    if (M < 4096 || topn < 1 || Mp < 16 || Mp > NPLDA_MAX_DIM) return p;
    // (the fixed part of a workspace holds the cohort's split image, 12 ceil(Mp / 32) KiB per 64 cohort rows: kept under 1 GiB —
    // the sizing functions budget 4 GiB per workspace — so a cohort of more than ~1.1 M utterances takes the spilling path)
    if ((size_t)((M + 63) / 64) * split_tile_bytes(Mp) > ((size_t)1 << 30)) return p;
    // candidates proposed per row: about twice the wanted count, and relatively more when N is small — the proposal
    // sits far out in the tail there, where a row's distribution agrees least with the normal model
    // NPLDA_COHORT_WANT: the factor on N (default 2).  On the bench's Gaussian rows 1.4 is 22 us faster (select 104 -> 88 us,
    // fused kernel 645 -> 634 us, fallback 2 -> 4.5 us: profiles/r03e, DESIGN K8/K9); the default stays at 2 because a row
    // whose scores are far from normal pays the fallback path, and real cohorts have such rows
    static const double want_f = getenv("NPLDA_COHORT_WANT") ? atof(getenv("NPLDA_COHORT_WANT")) : 2.0;
    const double want = want_f * topn + 16.0 + 300.0 * exp(-(double)topn / 300.0);
    const double f = want / (double)M;
    if (f > 0.25 || want > 0.8 * kCandMax) return p;
    {   // keys one row may bring to the select kernel: 1.4 x the proposal (NPLDA_COHORT_CAP), rounded up to 64, at most kCandMax.
        // A row with more goes to the fail list (exact, slower path) — on Gaussian rows the count is want +- ~3 %, and a row
        // 40 % over the proposal is far enough from the model to be there anyway.  cfg3: 1536 keys = 24 KiB per block, 6 blocks
        // per CU instead of 5 at kCandMax: select kernel 99.7 -> 84.5 us (1.3: 82.5, 1.2: 80.3)
        static const double cap_f = getenv("NPLDA_COHORT_CAP") ? atof(getenv("NPLDA_COHORT_CAP")) : 1.4;
        long long cap = ((long long)(cap_f * want) + 63) / 64 * 64;
        p.cap = (int)(cap < kCandMax ? cap : kCandMax);
    }
    const long long nx = (M + 127) / 128;
    p.nx = (int)nx;
    // Column bands: a multiple of 8 (band b belongs to XCD b % 8), widths equal to within one tile; a work item is (tile of
    // 256 rows, band) and ONE block walks it.  Eight bands unless a band's slice of the cohort table would not stay in an
    // XCD's 4 MB L2 next to the row fragments (> 2 MB), or its sub-lists would overflow.  Finer items do not pay: every
    // item switch re-fetches 256 rows of operands, and the measured kernel time (cfg3) is 772 / 800 / 800 us for
    // 8 / 16 / 32 bands although the CUs' shares of the items even out.
    if (nx < 8) return p;
    int bpx = 1;  // bands per XCD
    for (;; ++bpx) {
        const double cols = (double)((nx + 8 * bpx - 1) / (8 * bpx)) * 128.0;
        const double lam_ = f * cols / 4.0;
        if ((cols * Mp * 4.0 <= 2097152.0 && lam_ + 6.0 * sqrt(lam_) + 4.0 <= (double)(128 - kSubSlack)) || bpx == 8 || 8LL * (bpx + 1) > nx) break;
    }
    p.nbands = 8 * bpx;
    // List bands: a band's columns in kParts pieces with sub-lists of their own, so that the row tiles at the END of the
    // work queue can be handed out a piece at a time (cohort_fused2_kernel::decode) — when the pieces are at least one
    // 128-column unit wide and the sub-list count fits the select kernel's 4 counts per lane.
    p.q = (nx >= (long long)p.nbands * kParts && p.nbands * kParts * 4 <= 256) ? kParts : 1;
    p.ksub = p.q > 1 ? 64 : 128;
    const int nlb = p.nbands * p.q;
    p.nsub = nlb * 4;
    // expected candidates of one sub-list (a list band's columns seen by one lane group)
    const long long nt64 = (M + 63) / 64;
    const double lam = f * (double)((nt64 + nlb - 1) / nlb) * 64.0 / 4.0;
    if (lam + 6.0 * sqrt(lam) + 4.0 > (double)(p.ksub - kSubSlack)) {
        if (p.q == 1) return p;
        p.q = 1; p.ksub = 128; p.nsub = p.nbands * 4;
        const double lam1 = f * (double)((nt64 + p.nbands - 1) / p.nbands) * 64.0 / 4.0;
        if (lam1 + 6.0 * sqrt(lam1) + 4.0 > (double)(p.ksub - kSubSlack)) return p;
    }
    p.zhi = host_normcdfinv(f);
    p.fhi = (float)f;
    const size_t kb = (size_t)Mp / 16;
    p.fixed_bytes = 256 + align256((size_t)kGramSplit * Mp * Mp * 4) + align256((size_t)kGramSplit * 4 * Mp * 4) +
                    align256((size_t)kQzBlocks * (Mp + 2) * 4) + align256((size_t)kQzBlocks * (Mp + 1) * 8) +
                    align256((size_t)(Mp + 1) * 8) + align256(kb * kb * 256 * 4) +
                    align256((size_t)(2 * Mp + 2) * 4) + 8 * 256 +  // + the alignment slack of the per-row arrays
                    align256((size_t)((M + 63) / 64) * split_tile_bytes(Mp));  // the cohort's split image (cohort_split_kernel)
    const long long lrow = (long long)p.nsub * p.ksub + fused_row_pad();
    p.max_rows = ((1LL << 30) / lrow) / 256 * 256;  // 32-bit BYTE offsets into the lists
    p.row_bytes = 8 + 8 + (size_t)lrow * 4 + (size_t)p.nsub * 4 + (size_t)(p.nsub / 4) * 8 + 4;
    p.eligible = true;
    return p;
}

// Runs the fused path on rows [0, R) (R <= the rows the workspace was planned for).  `fail_rows` / `nfail` (device)
// receive the rows that need the general path; the caller runs cohort_fallback_kernel on them.
// the fixed part of a workspace (FusedPlan::fixed_bytes): control block, the pre-pass's scratch and its results
struct FusedFixed {
    unsigned* ctl; float* slab; float* ext; float* qz; double* qz64; double* vec64; float* frag; float* vec;
    unsigned char* img;   // the cohort's split image: ceil(M / 64) tiles of split_tile_bytes(Mp)
    unsigned char* end;
};
static FusedFixed fused_fixed(unsigned char* ws, int Mp, long long M) {
    FusedFixed f;
    unsigned char* q = ws;
    f.ctl = reinterpret_cast<unsigned*>(q); q += 256;
    f.slab = reinterpret_cast<float*>(q); q += align256((size_t)kGramSplit * Mp * Mp * 4);
    f.ext = reinterpret_cast<float*>(q); q += align256((size_t)kGramSplit * 4 * Mp * 4);
    f.qz = reinterpret_cast<float*>(q); q += align256((size_t)kQzBlocks * (Mp + 2) * 4);
    f.qz64 = reinterpret_cast<double*>(q); q += align256((size_t)kQzBlocks * (Mp + 1) * 8);
    f.vec64 = reinterpret_cast<double*>(q); q += align256((size_t)(Mp + 1) * 8);
    f.frag = reinterpret_cast<float*>(q); q += align256((size_t)(Mp / 16) * (Mp / 16) * 256 * 4);
    f.vec = reinterpret_cast<float*>(q); q += align256((size_t)(2 * Mp + 2) * 4);
    f.img = q; q += align256((size_t)((M + 63) / 64) * split_tile_bytes(Mp));
    f.end = q;
    return f;
}

// cohort moments: second and first moments of the cohort in ONE launch (the first-moment blocks ride along as extra work
// items), then the centred covariance folded with 2 P into a fragment image
static int fused_prepass(const FusedFixed& F, const float* z_coh, const float* q_coh, long long M, long long ldz, const float* P,
                         int Mp, hipStream_t st) {
    const QzArgs qa = {z_coh, q_coh, M, ldz, Mp, kQzBlocks, F.qz, F.qz64};
    if (int rc = gram_slabs_launch(z_coh, ldz, M, Mp, kGramSplit, F.slab, F.ext, &qa, st)) return rc;
    PrepArgs pa = {F.slab, F.ext, F.qz, F.qz64, P, kGramSplit, Mp, M, F.frag, F.vec, F.vec64, F.ctl};
    // ksplit actually used by gram_slabs_launch: rows per split rounded up -> some trailing slabs may be unwritten
    {
        long long rps = (M + kGramSplit - 1) / kGramSplit;
        rps = (rps + 63) / 64 * 64;
        pa.ksplit = (int)((M + rps - 1) / rps);
    }
    const size_t nfrag = (size_t)(Mp / 16) * (Mp / 16) * 256;
    hipLaunchKernelGGL(cohort_prep_kernel, dim3((unsigned)((nfrag + 255) / 256)), dim3(256), 0, st, pa);
    if (split_supported(Mp / 16)) {  // the cohort in three bf16 pieces, fragment order (what the split fused kernel streams)
        const SplitArgs sa = {z_coh, M, ldz, Mp, F.img};
        const long long nthr = (M + 63) / 64 * split_ksteps(Mp) * 256;
        hipLaunchKernelGGL(cohort_split_kernel, dim3((unsigned)((nthr + 255) / 256)), dim3(256), 0, st, sa);
    }
    return nplda_launch_status();
}

int cohort_fused_prepare(const FusedPlan& p, const float* z_coh, const float* q_coh, long long M, long long ldz, const float* P,
                         int ksteps, unsigned char* state, hipStream_t st) {
    if (!p.eligible) return NPLDA_EUNSUPPORTED;
    return fused_prepass(fused_fixed(state, 16 * ksteps, M), z_coh, q_coh, M, ldz, P, 16 * ksteps, st);
}

int cohort_fused_run(const FusedPlan& p, const float* z_rows, const float* q_rows, long long R, const float* z_coh,
                     const float* q_coh, long long M, long long ldz, const float* P, int ksteps, int topn, int lowest,
                     double* stats, unsigned char* ws, long long rows_cap, bool prepass, unsigned** fail_rows_out,
                     unsigned** nfail_out, long long resident, hipStream_t st, const unsigned char* prepared, int D2) {
    const int Mp = 16 * ksteps;
    const FusedFixed F = fused_fixed(ws, Mp, M);
    unsigned* ctl = F.ctl;
    unsigned char* q = F.end;
    // (a prepared cohort: its covariance image and moment vectors are read where cohort_fused_prepare left them)
    const FusedFixed FP = prepared ? fused_fixed(const_cast<unsigned char*>(prepared), Mp, M) : F;
    const double* vec64 = FP.vec64;
    const float* frag = FP.frag;
    const float* vec = FP.vec;
    if (prepared) prepass = false;
    // per-row arrays, sized for rows_cap rows
    double* part = reinterpret_cast<double*>(q); q += align256((size_t)rows_cap * (p.nsub / 4) * 8);
    double* mean64 = reinterpret_cast<double*>(q); q += align256((size_t)rows_cap * 8);
    const int lrow = p.nsub * p.ksub + fused_row_pad();
    float* lists = reinterpret_cast<float*>(q); q += align256((size_t)rows_cap * lrow * 4);
    unsigned* counts = reinterpret_cast<unsigned*>(q); q += align256((size_t)rows_cap * p.nsub * 4);
    float* crow = reinterpret_cast<float*>(q); q += align256((size_t)rows_cap * 4);
    float* trow = reinterpret_cast<float*>(q); q += align256((size_t)rows_cap * 4);
    unsigned* fail_rows = reinterpret_cast<unsigned*>(q);
    *fail_rows_out = fail_rows;
    *nfail_out = ctl + 8;

    // (the control block — work-item counters, fail count — is zeroed by the pre-pass, or, with a prepared cohort, by the
    // row-threshold kernel below: a hipMemsetAsync in front cost a launch, and as a memset NODE of a captured graph it let a
    // replayed AS-norm step fault — tools/exp_cfg3_graph.py, round 6)
    if (prepass) {  // cohort moments: once per call, the cohort does not change between row chunks
        if (int rc = fused_prepass(F, z_coh, q_coh, M, ldz, P, Mp, st)) return rc;
    }
    {   // row means and thresholds
        // rows per block: 128, or fewer (down to 32) when 128-row tiles would leave most CUs without a block
        int wpt = 8;
        while (wpt > 2 && (R + 16 * wpt - 1) / (16 * wpt) < resident / 2) wpt >>= 1;
        RowThrArgs ra = {z_rows, q_rows, R, ldz, frag, vec, vec64, p.zhi, lowest ? 1.0f : -1.0f, crow, trow, mean64,
                         (int)((R + 16 * wpt - 1) / (16 * wpt)), wpt, prepass ? nullptr : ctl};
        const unsigned grid = (unsigned)(ra.ntiles < resident ? ra.ntiles : resident);
        const size_t shm = (size_t)ksteps * (ksteps + 1) / 2 * 1024;
#define NPLDA_LAUNCH(NBV)                                                                                             \
    {                                                                                                                 \
        if (hipFuncSetAttribute(reinterpret_cast<const void*>(&cohort_rowthr_kernel<NBV>),                             \
                                hipFuncAttributeMaxDynamicSharedMemorySize, (int)shm) != hipSuccess)                  \
            return NPLDA_EINVAL;                                                                                      \
        hipLaunchKernelGGL(cohort_rowthr_kernel<NBV>, dim3(grid), dim3(512), shm, st, ra);                            \
    }
        switch (ksteps) {
            case 2: NPLDA_LAUNCH(2); break;
            case 4: NPLDA_LAUNCH(4); break;
            case 8: NPLDA_LAUNCH(8); break;
            case 10: NPLDA_LAUNCH(10); break;
            case 11: NPLDA_LAUNCH(11); break;
            case 12: NPLDA_LAUNCH(12); break;
            default: return NPLDA_EUNSUPPORTED;
        }
#undef NPLDA_LAUNCH
        if (int rc = nplda_launch_status()) return rc;
    }

    FusedArgs fa = {};
    fa.zr = z_rows; fa.qr = q_rows; fa.zc = z_coh; fa.qc = q_coh; fa.P = P;
    fa.zc3 = FP.img;
    // Row tiles of 256 rows — or of 128 when the call has so few rows that 256-row items would leave most CUs idle in the
    // last round (fewer than three items per resident block: an 8-way row shard of cfg3 is 11 tiles x 32 list bands for
    // 256 CUs, 1.4 items each; as 22 x 32 half-size items the slowest CU carries 7.5 tile-units instead of 10).  Same
    // results, bit for bit (cohort_fused2_kernel).
    // (the split form's 128-row tile carries the full 60 - 72 KiB of tile DMA for half the MFMAs: it pays only below TWO items
    // per resident block — 2 750 rows 0.108 against 0.117 ms, 5 500 rows 0.155 against 0.151: tools/cohort_half_ab.sh)
    bool half_tiles = p.q > 1 && ((R + 255) / 256) * p.nbands * p.q < (split_selected(ksteps) ? 2 : 3) * resident;
    if (const char* e = getenv("NPLDA_COHORT_HALF")) half_tiles = p.q > 1 && e[0] == '1';   // (A/B runs: force 128- / 256-row tiles)
    const int rpb = half_tiles ? 128 : 256;
    fa.R = R; fa.M = M; fa.ldz = ldz; fa.ksteps = ksteps; fa.nbands = p.nbands; fa.ny = (int)((R + rpb - 1) / rpb); fa.nx = p.nx;
    fa.ctr = ctl; fa.crow = crow; fa.trow = trow; fa.lists = lists; fa.counts = counts; fa.part = part;
    fa.nsub = p.nsub; fa.q = p.q; fa.ksub = p.ksub; fa.lrow = lrow;
    // (the split form: 2 = the MFMA loop at priority 0, everything else of a tile at 3 — with a bf16 MFMA stream at priority 0 the
    // partner's VALU instructions at priority 3 issue beside it at no cost to the stream, which the fp32-input MFMAs do not allow:
    // tools/exp_mfma_yield.hip, profiles/r06fin5_mfma_yield_bf16.txt; in the kernel 375 - 380 us against 381 (no priorities)
    // and 386 - 392 (the fp32 form's static 1) in two interleaved repetitions, profiles/r06j_split_prio_ab.txt — small, the
    // kernel is at the power cap)
    { static const int pr = getenv("NPLDA_COHORT_PRIO") ? atoi(getenv("NPLDA_COHORT_PRIO")) : -1;
      fa.prio = pr >= 0 ? pr : (split_selected(ksteps) ? 2 : 1); }
#ifdef NPLDA_COHORT_ABLATE
    {
        static unsigned long long* dstamps = nullptr;
        if (!dstamps) (void)hipMalloc(&dstamps, 2 * 64 * 8 * sizeof(unsigned long long));
        (void)hipMemsetAsync(dstamps, 0, 2 * 64 * 8 * sizeof(unsigned long long), st);
        fa.abl = getenv("NPLDA_COHORT_ABL") ? atoi(getenv("NPLDA_COHORT_ABL")) : 0;
        fa.stamps = dstamps;
        g_cf_stamps = dstamps;
    }
#endif
    long long grid = (long long)fa.ny * p.nbands * p.q;  // at most one block per work item
    if (grid > resident) grid = resident;
    {   // whole-band items for as many row tiles as fill complete rounds of an XCD's blocks, the rest in list bands
        const long long per_xcd = grid / 8 > 0 ? grid / 8 : 1;
        fa.nfull = p.q > 1 ? (int)(fa.ny / per_xcd * per_xcd) : fa.ny;
    }
    // k4-steps of the last k16-block (cohort_fused2_kernel, KT): the instantiated short forms are D2 in 145 .. 152 at NB = 10
    // (KT = 2: the reference's 150) and D2 in 169 .. 172 at NB = 11 (KT = 3: the shipped 170)
    const int tail = D2 > 0 ? D2 - 16 * (ksteps - 1) : 16;
    const int kt = (ksteps == 10 && tail >= 1 && tail <= 8) ? 2 : ((ksteps == 11 && tail >= 9 && tail <= 12) ? 3 : 4);
    static const bool no_kt = getenv("NPLDA_COHORT_NO_KTAIL") != nullptr && getenv("NPLDA_COHORT_NO_KTAIL")[0] == '1';  // A/B only
    // four waves per SIMD (NW = 16: a wave owns ONE 16-row group, the block the same 256 rows; 128 VGPRs per wave): while one
    // wave's MFMA loop has the matrix pipe, three partners — not one — work through their epilogues in the issue slots it leaves
    static const int nw_env = getenv("NPLDA_COHORT_NW") ? atoi(getenv("NPLDA_COHORT_NW")) : 8;
    const bool nw16 = nw_env == 16 && !half_tiles && (ksteps == 10 || ksteps == 11);
#define NPLDA_LAUNCH_KT(NBV, KTV)                                                                                                   \
    if (nw16) {                                                                                                                     \
        if constexpr (NBV == 10 || NBV == 11) {                                                                                     \
            if (lowest) hipLaunchKernelGGL((cohort_fused2_kernel<true, NBV, 1, KTV, 16>), dim3((unsigned)grid), dim3(1024), 0, st, fa);  \
            else hipLaunchKernelGGL((cohort_fused2_kernel<false, NBV, 1, KTV, 16>), dim3((unsigned)grid), dim3(1024), 0, st, fa);        \
        }                                                                                                                           \
    } else if (half_tiles) {                                                                                                               \
        if (lowest) hipLaunchKernelGGL((cohort_fused2_kernel<true, NBV, 1, KTV>), dim3((unsigned)grid), dim3(512), 0, st, fa);       \
        else hipLaunchKernelGGL((cohort_fused2_kernel<false, NBV, 1, KTV>), dim3((unsigned)grid), dim3(512), 0, st, fa);             \
    } else if (lowest) hipLaunchKernelGGL((cohort_fused2_kernel<true, NBV, 2, KTV>), dim3((unsigned)grid), dim3(512), 0, st, fa);    \
    else hipLaunchKernelGGL((cohort_fused2_kernel<false, NBV, 2, KTV>), dim3((unsigned)grid), dim3(512), 0, st, fa)
#define NPLDA_LAUNCH(NBV) NPLDA_LAUNCH_KT(NBV, 4)
#define NPLDA_LAUNCH_SPLIT(NBV)                                                                                                      \
    if (half_tiles) {                                                                                                                \
        if (lowest) hipLaunchKernelGGL((cohort_fused2_kernel<true, NBV, 1, 4, 8, true>), dim3((unsigned)grid), dim3(512), 0, st, fa);  \
        else hipLaunchKernelGGL((cohort_fused2_kernel<false, NBV, 1, 4, 8, true>), dim3((unsigned)grid), dim3(512), 0, st, fa);        \
    } else if (lowest) hipLaunchKernelGGL((cohort_fused2_kernel<true, NBV, 2, 4, 8, true>), dim3((unsigned)grid), dim3(512), 0, st, fa); \
    else hipLaunchKernelGGL((cohort_fused2_kernel<false, NBV, 2, 4, 8, true>), dim3((unsigned)grid), dim3(512), 0, st, fa)
    if (split_selected(ksteps)) {
        if (ksteps == 10) { NPLDA_LAUNCH_SPLIT(10); } else { NPLDA_LAUNCH_SPLIT(11); }
    } else
    switch (ksteps) {
        case 2: NPLDA_LAUNCH(2); break;
        case 4: NPLDA_LAUNCH(4); break;
        case 8: NPLDA_LAUNCH(8); break;
        case 10:
            if (kt == 2 && !no_kt) { NPLDA_LAUNCH_KT(10, 2); } else { NPLDA_LAUNCH(10); }
            break;
        case 11:
            if (kt == 3 && !no_kt) { NPLDA_LAUNCH_KT(11, 3); } else { NPLDA_LAUNCH(11); }
            break;
        case 12: NPLDA_LAUNCH(12); break;
        default: return NPLDA_EUNSUPPORTED;
    }
#undef NPLDA_LAUNCH
#undef NPLDA_LAUNCH_KT
#undef NPLDA_LAUNCH_SPLIT
    if (int rc = nplda_launch_status()) return rc;
    FinishArgs fi = {lists, counts, part, crow, mean64, R, M, trow, p.zhi, p.fhi, p.nsub, p.nsub, topn, lowest, p.ksub, p.cap, lrow, ctl + 8,
                     fail_rows, stats};
    hipLaunchKernelGGL(cohort_finish_kernel, dim3((unsigned)((R + 3) / 4)), dim3(256), (size_t)p.cap * 16, st, fi);
    return nplda_launch_status();
}

long long cohort_fused_resident_blocks() {
    int dev = 0, cus = 0, per_cu = 0;
    if (hipGetDevice(&dev) != hipSuccess ||
        hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, dev) != hipSuccess ||
        hipOccupancyMaxActiveBlocksPerMultiprocessor(&per_cu, cohort_fused2_kernel<true, 12, 2>, 512, 0) != hipSuccess)
        return 0;
    long long r = (long long)cus * per_cu / 8 * 8;
    return r < 8 ? 8 : r;
}

}  // namespace nplda

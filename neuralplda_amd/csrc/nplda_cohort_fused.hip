// nplda_cohort_fused.hip — cohort score matrix + per-row statistics in ONE pass, nothing spilled (gfx950).
//
// nplda_cohort.hip forms S = q_r + q_m + 2 (z_r * P) z_m^T on MFMA tiles, writes it out (0.88 GB at BASELINE cfg3) and
// reads it back for the row statistics: 1.76 GB of traffic for 22.5 MB of input.  Here the statistics are taken in the
// GEMM's epilogue, from the accumulators:
//
//  * mean / std need sum and sum of squares: accumulated per lane, centred on the row's ANALYTIC mean c_r (so that the
//    fp32 partial sums of a tile carry no cancellation), added to fp64 running sums once per tile and row;
//  * the top-N statistics (adaptive_score_normalization.py:32-36: the N smallest) need the N smallest scores of the row
//    exactly.  A threshold t_r slightly above the N-th smallest is PROPOSED from the row's analytic mean and standard
//    deviation — both follow from the cohort's first and second moments, S[r, m] = q_r + q_m + a_r . z_m with
//    a_r = 2 P z_r:  mean_m = q_r + mean(q) + a_r . mean(z),  var_m = var(q) + 2 a_r . cov(z, q) + a_r^T cov(z) a_r —
//    and every score <= t_r is appended to a candidate list of that row.  The lists then hold ~2 N of the M scores; a
//    small kernel selects the N smallest among them exactly (ties by count) and sums them in fp64.  The counts decide:
//    a row whose list holds fewer than N scores, or more than fit, is recomputed by the exact general path
//    (cohort_fallback_kernel in nplda_cohort.hip) — the proposal never changes a result, only who computes it.
//  * no atomics on data, no run-to-run variation: a work item is (row tile of 128 rows, band of column tiles); ONE block
//    walks the band's tiles in order, so every lane meets "its" columns of "its" rows in a fixed order and appends to
//    a private sub-list (row, band, wave column, lane group): the lists' contents, their order and all sums are
//    independent of scheduling and of where a row sits in the table.
//
// Pre-pass (five small launches): Gram matrix of the cohort table by the split-K wgrad kernel (nplda_backward.hip) +
// sums of q, q^2, q z; centred covariance folded with 2 P into a fragment image; (z_rows . C'') by the resident-matrix
// GEMM (nplda_matmul.hip); one wave per row forms c_r and t_r.
#include <stdlib.h>

#include <type_traits>

#include "nplda_cohort_common.h"
#include "nplda_cohort_fused.h"

namespace nplda {
int gram_slabs_launch(const float* Z, long long ldz, long long rows, int Mp, int ksplit, float* slab, float* ext,
                      hipStream_t st);
int rows_matmul_launch(const float* in, long long ldin, long long R, int K, const float* frag, int N, float* out,
                       long long ldout, hipStream_t st);
}  // namespace nplda

namespace {

#ifndef NPLDA_FUSED_MINBLOCKS
#define NPLDA_FUSED_MINBLOCKS 3   // 167 VGPRs with the per-row state in LDS (52 KB per block: three blocks per CU)
#endif
constexpr int kSub = 64;          // slots per candidate sub-list (47 usable + slack for one tile's 16 appends)
constexpr int kSubFull = 47;      // a sub-list that reaches this count is treated as overflowed
constexpr int kCandMax = 2048;    // candidates one row may bring to the select kernel (32 keys per lane)
constexpr int kGramSplit = 16;    // k-groups of the cohort Gram matrix
constexpr int kQzBlocks = 64;

// ------------------------------------------------------------------------------------------------------------------
// pre-pass
// ------------------------------------------------------------------------------------------------------------------

// partial sums over a block's share of the cohort rows: part[b][0..Mp) = sum q_m z_m, part[b][Mp] = sum q, [Mp+1] = sum q^2
__global__ __launch_bounds__(256) void cohort_qz_kernel(const float* __restrict__ zc, const float* __restrict__ qc,
                                                        long long M, long long ldz, int Mp, float* __restrict__ part) {
    __shared__ float red[4][NPLDA_MAX_DIM + 2];
    const int lane = threadIdx.x & 63, wy = threadIdx.x >> 6;
    const long long per = (M + gridDim.x - 1) / gridDim.x;
    const long long lo = (long long)blockIdx.x * per, hi = lo + per < M ? lo + per : M;
    float acc[3] = {0.f, 0.f, 0.f}, sq = 0.f, sqq = 0.f;
    // four rows in flight per wave (independent loads), accumulated in row order
    for (long long m0 = lo + wy; m0 < hi; m0 += 16) {
        float q[4], z[4][3];
#pragma unroll
        for (int u = 0; u < 4; ++u) {
            const long long m = m0 + 4 * u;
            const long long mc = m < hi ? m : hi - 1;
            q[u] = m < hi ? qc[mc] : 0.f;
#pragma unroll
            for (int c = 0; c < 3; ++c) {
                const int f = lane + 64 * c;
                z[u][c] = f < Mp ? zc[mc * ldz + f] : 0.f;
            }
        }
#pragma unroll
        for (int u = 0; u < 4; ++u) {
#pragma unroll
            for (int c = 0; c < 3; ++c) acc[c] = fmaf(q[u], z[u][c], acc[c]);
            sq += q[u];
            sqq = fmaf(q[u], q[u], sqq);
        }
    }
#pragma unroll
    for (int c = 0; c < 3; ++c)
        if (lane + 64 * c < Mp) red[wy][lane + 64 * c] = acc[c];
    if (lane == 0) { red[wy][Mp] = sq; red[wy][Mp + 1] = sqq; }
    __syncthreads();
    for (int i = threadIdx.x; i < Mp + 2; i += 256)
        part[(size_t)blockIdx.x * (Mp + 2) + i] = ((red[0][i] + red[1][i]) + red[2][i]) + red[3][i];
}

struct PrepArgs {
    const float* slab;   // [ksplit][Mp][Mp]
    const float* ext;    // [ksplit][4][Mp]  (row 3: column sums of z)
    const float* qz;     // [kQzBlocks][Mp + 2]
    const float* P;      // padded, zero beyond D2
    int ksplit, Mp;
    long long M;
    float* frag;         // [KB][KB][64][4]: C''[i][j] = 4 P_i P_j cov(z)_ij
    float* vec;          // [0, Mp): u = 2 P mean(z); [Mp, 2 Mp): v = 4 P cov(z, q); [2 Mp]: mean(q); [2 Mp + 1]: var(q)
};

__global__ __launch_bounds__(256) void cohort_prep_kernel(const PrepArgs a) {
    __shared__ double zbar[NPLDA_MAX_DIM];
    const int Mp = a.Mp, KB = Mp / 16;
    const size_t idx = (size_t)blockIdx.x * 256 + threadIdx.x;
    const double n = (double)a.M;
    if ((int)threadIdx.x < Mp) {  // every block forms the cohort mean for itself (ksplit x Mp L2-resident floats)
        double s = 0.0;
        for (int k = 0; k < a.ksplit; ++k) s += (double)a.ext[((size_t)k * 4 + 3) * Mp + threadIdx.x];
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
        a.frag[idx] = (float)(4.0 * (double)a.P[i] * (double)a.P[j] * cov);
    }
    if (idx < (size_t)Mp) {
        const int i = (int)idx;
        const double zm = zbar[i], qm = qsum(Mp) / n;
        a.vec[i] = (float)(2.0 * (double)a.P[i] * zm);
        a.vec[Mp + i] = (float)(4.0 * (double)a.P[i] * (qsum(i) / n - qm * zm));
        if (i == 0) {
            a.vec[2 * Mp] = (float)qm;
            double vq = qsum(Mp + 1) / n - qm * qm;
            a.vec[2 * Mp + 1] = (float)(vq > 0.0 ? vq : 0.0);
        }
    }
}

// one wave per row: c_r = the row's analytic mean and t_r = the candidate threshold, c_r + sgn * zhi * sd (zhi < 0: below the
// mean for the N smallest, sgn = +1; above it for the N largest, sgn = -1)
__global__ __launch_bounds__(256) void cohort_threshold_kernel(const float* __restrict__ zr, const float* __restrict__ qr,
                                                               const float* __restrict__ tmp, long long R,
                                                               long long ldz, int Mp, const float* __restrict__ vec,
                                                               float zhi, float sgn, float* __restrict__ crow,
                                                               float* __restrict__ trow) {
    const long long r = (long long)blockIdx.x * 4 + (threadIdx.x >> 6);
    const int lane = threadIdx.x & 63;
    if (r >= R) return;
    float quad = 0.f, lin = 0.f, mu = 0.f;
    for (int f = lane; f < Mp; f += 64) {
        const float z = zr[r * ldz + f];
        quad = fmaf(z, tmp[r * Mp + f], quad);
        lin = fmaf(z, vec[Mp + f], lin);
        mu = fmaf(z, vec[f], mu);
    }
#pragma unroll
    for (int m = 32; m >= 1; m >>= 1) {
        quad += __shfl_xor(quad, m, 64);
        lin += __shfl_xor(lin, m, 64);
        mu += __shfl_xor(mu, m, 64);
    }
    if (lane == 0) {
        const float mean = qr[r] + vec[2 * Mp] + mu;
        const float var = vec[2 * Mp + 1] + lin + quad;
        const float sd = sqrtf(fmaxf(var, 0.f));
        crow[r] = mean;
        trow[r] = mean + sgn * zhi * sd;
    }
}

// ------------------------------------------------------------------------------------------------------------------
// the fused GEMM (tile pipeline of cohort_gemm_kernel, nplda_cohort.hip — see the comments there — with the statistics
// epilogue instead of the stores)
// ------------------------------------------------------------------------------------------------------------------
struct FusedArgs {
    const float* zr; const float* qr; const float* zc; const float* qc; const float* P;
    long long R, M, ldz;
    int ksteps, nxp, ny, nx, nsb;
    unsigned* ctr;          // 8 work-item counters (one per XCD), zero at launch
    const float* crow;      // (R)
    const float* trow;      // (R)
    float* lists;           // [R][kSub][nsub]: slot e of sub-list s of row r at (r kSub + e) nsub + s (slot-major: the
                            // select kernel reads whole slot rows, a row's appended lines stay dense)
    unsigned* counts;       // [R][nsub]
    double* part;           // [R][nsub / 4][2]
    int nsub;               // 8 (bands per super-band) * nsb * 2 (wave columns) * 4 (lane groups)
};

template <bool LOWEST, int MINBLOCKS>
__global__ __launch_bounds__(256, MINBLOCKS) void cohort_fused_kernel(const FusedArgs a) {
    // ONE __shared__ object (see cohort_gemm_kernel): stages | 2 P fragments | self terms | next-slot word | per-lane row
    // state (centre, threshold, running sums of the lane's four rows: 16 floats per thread kept OUT of the register file,
    // which is what lets three blocks share a CU)
    __shared__ f32x4 smem[2 * 2 * 512 + 48 + 128 + 8 + 1024];
    f32x4 (*tile)[2][512] = reinterpret_cast<f32x4 (*)[2][512]>(smem);
    f32x4* p2s = smem + 2048;
    float* qs = reinterpret_cast<float*>(smem + 2096);
    float* lst = reinterpret_cast<float*>(smem + 2232) + threadIdx.x;  // lst[256 k]: k = 0..3 centre, 4..7 threshold, 8..11 s1, 12..15 s2
    const int tid = threadIdx.x;
    const int lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int i16 = lane & 15, g4 = lane >> 4;
    const int xcd = blockIdx.x & 7;
    // work item `slot` of this XCD: super-band by super-band, row tile by row tile -> (row tile, band, its column tiles)
    auto decode = [&](int slot, long long& rb, int& band, int& tx0, int& tx1) {
        for (int sb = 0; sb < a.nsb; ++sb) {
            const int t0 = (sb * 8 + xcd) * a.nxp;
            int w = a.nx - t0;
            if (w > a.nxp) w = a.nxp;
            if (w <= 0) break;
            if (slot < a.ny) {
                rb = (long long)slot * 128;
                band = sb * 8 + xcd;
                tx0 = t0;
                tx1 = t0 + w;
                return true;
            }
            slot -= a.ny;
        }
        return false;
    };

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
    auto q_in = [&](long long rb, long long mb, int par) {
        const bool isr = wave < 2;
        long long j = (isr ? rb : mb) + 64 * (wave & 1) + lane;
        const long long lim = isr ? a.R : a.M;
        if (j >= lim) j = lim - 1;
        __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)((isr ? a.qr : a.qc) + j),
                                         (__attribute__((address_space(3))) void*)&qs[par * 256 + 64 * wave], 4, 0, 0);
    };

    long long rb = 0, nrb = 0;
    int band = 0, tx = 0, tx1 = 0, nband = 0, ntx0 = 0, ntx1 = 0;
    unsigned* nxt_s = reinterpret_cast<unsigned*>(smem + 2224);
    if (tid == 0) *nxt_s = atomicAdd(a.ctr + xcd, 1u);
    __syncthreads();
    if (!decode(__builtin_amdgcn_readfirstlane((int)*nxt_s), rb, band, tx, tx1)) return;
    if (tid < 4 * a.ksteps) p2s[tid] = 2.0f * *reinterpret_cast<const f32x4*>(a.P + 4 * tid);
    const int fo = i16 * 4 + (g4 ^ ((i16 >> 2) & 2));
    const f32x4* fra = &tile[0][0][(wave >> 1) * 256 + fo];
    const f32x4* frb = &tile[0][1][(wave & 1) * 256 + fo];

    // per-lane state of the work item: the lane's four rows (ca), their centre / threshold, the write cursor of the
    // lane's private candidate sub-list of each row, and the running fp64 sums
    // cur: BYTE offset of the lane's next free slot in each row's private sub-list.  The centred sums of ONE work item
    // (~10 tiles x 16 values per lane and row) are kept in fp32: ample.
    unsigned cur[4];
    auto item_begin = [&](long long rb_, int band_) {
#pragma unroll
        for (int ca = 0; ca < 4; ++ca) {
            const long long row = rb_ + (wave >> 1) * 64 + 16 * ca + i16;
            const bool ok = row < a.R;
            const long long rc = ok ? row : a.R - 1;
            lst[256 * ca] = a.crow[rc];
            lst[256 * (4 + ca)] = ok ? a.trow[rc] : (LOWEST ? -__builtin_inff() : __builtin_inff());  // rows past the table never append
            lst[256 * (8 + ca)] = 0.f;
            lst[256 * (12 + ca)] = 0.f;
            cur[ca] = 4u * (unsigned)(rc * kSub * a.nsub + (band_ * 2 + (wave & 1)) * 4 + g4);
        }
    };
    auto item_end = [&](long long rb_, int band_) {
#pragma unroll
        for (int ca = 0; ca < 4; ++ca) {
            const long long row = rb_ + (wave >> 1) * 64 + 16 * ca + i16;
            // the four lane groups of a row: fixed association ((g0 + g1) + (g2 + g3)) by two exchanges
            double t1 = (double)lst[256 * (8 + ca)], t2 = (double)lst[256 * (12 + ca)];
            t1 += __hiloint2double(__shfl_xor(__double2hiint(t1), 16, 64), __shfl_xor(__double2loint(t1), 16, 64));
            t2 += __hiloint2double(__shfl_xor(__double2hiint(t2), 16, 64), __shfl_xor(__double2loint(t2), 16, 64));
            t1 += __hiloint2double(__shfl_xor(__double2hiint(t1), 32, 64), __shfl_xor(__double2loint(t1), 32, 64));
            t2 += __hiloint2double(__shfl_xor(__double2hiint(t2), 32, 64), __shfl_xor(__double2loint(t2), 32, 64));
            if (row < a.R) {
                const size_t sub = (size_t)row * a.nsub + (band_ * 2 + (wave & 1)) * 4 + g4;
                a.counts[sub] = (cur[ca] / 4u - (unsigned)(row * kSub * a.nsub + (band_ * 2 + (wave & 1)) * 4 + g4)) / (unsigned)a.nsub;
                if (g4 == 0) {
                    double* o = a.part + ((size_t)row * (a.nsub / 4) + band_ * 2 + (wave & 1)) * 2;
                    o[0] = t1;
                    o[1] = t2;
                }
            }
        }
    };

    set_ptrs(rb, (long long)tx * 128);
    stage_in(0, 0);
    q_in(rb, (long long)tx * 128, 0);
    item_begin(rb, band);
    __syncthreads();
    int gpar = 0, qpar = 0;

    for (;;) {
        const bool last_tile = tx + 1 == tx1;   // of this work item
        unsigned pend = 0;
        if (last_tile && tid == 0) pend = atomicAdd(a.ctr + xcd, 1u);
        if (last_tile && a.ksteps == 1) {
            if (tid == 0) *nxt_s = pend;
            __syncthreads();
        }
        bool have_next = true;
        f32x4 acc[4][4];
#pragma unroll
        for (int ca = 0; ca < 4; ++ca)
#pragma unroll
            for (int cb = 0; cb < 4; ++cb) acc[ca][cb] = f32x4{0.f, 0.f, 0.f, 0.f};

        for (int ks = 0; ks < a.ksteps; ++ks) {
            const int cb_ = gpar;
            if (ks + 1 < a.ksteps) {
                stage_in(ks + 1, cb_ ^ 1);
            } else if (!last_tile) {  // next tile of the same work item
                set_ptrs(rb, (long long)(tx + 1) * 128);
                stage_in(0, cb_ ^ 1);
                q_in(rb, (long long)(tx + 1) * 128, qpar ^ 1);
            } else {                  // first tile of the next work item
                const int nslot = __builtin_amdgcn_readfirstlane((int)*nxt_s);
                have_next = decode(nslot, nrb, nband, ntx0, ntx1);
                if (have_next) {
                    set_ptrs(nrb, (long long)ntx0 * 128);
                    stage_in(0, cb_ ^ 1);
                    q_in(nrb, (long long)ntx0 * 128, qpar ^ 1);
                }
            }
            f32x4 fa[4], fb[4];
            const f32x4 pf = p2s[4 * ks + g4];
#pragma unroll
            for (int c = 0; c < 4; ++c) {
                fa[c] = fra[cb_ * 1024 + c * 64] * pf;
                fb[c] = frb[cb_ * 1024 + c * 64];
            }
#pragma unroll
            for (int kk = 0; kk < 4; ++kk)
#pragma unroll
                for (int ca = 0; ca < 4; ++ca)
#pragma unroll
                    for (int cb = 0; cb < 4; ++cb)
                        acc[ca][cb] = __builtin_amdgcn_mfma_f32_16x16x4f32(fb[cb][kk], fa[ca][kk], acc[ca][cb], 0, 0, 0);
            __builtin_amdgcn_sched_barrier(0);
            __builtin_amdgcn_s_waitcnt(0x0F70);
            if (last_tile && ks == 0 && a.ksteps > 1) {
                if (tid == 0) *nxt_s = pend;
                __builtin_amdgcn_s_waitcnt(0xC07F);
            }
            __builtin_amdgcn_s_barrier();
            gpar ^= 1;
        }

        // ---- statistics epilogue: lane (i16, g4) of block (ca, cb) holds row 16 ca + i16, columns 16 cb + 4 g4 + r ----
        const long long m0 = (long long)tx * 128 + (wave & 1) * 64;
        const float* qr_s = qs + qpar * 256 + (wave >> 1) * 64 + i16;
        const float* qm_s = qs + qpar * 256 + 128 + (wave & 1) * 64 + 4 * g4;
        f32x4 qmv[4];
#pragma unroll
        for (int cb = 0; cb < 4; ++cb) qmv[cb] = *reinterpret_cast<const f32x4*>(qm_s + 16 * cb);
        const unsigned stride_b = 4u * (unsigned)a.nsub;          // bytes between consecutive slots of a sub-list
        const long long rlane = rb + (wave >> 1) * 64 + i16;       // this lane's row of ca = 0
        const unsigned sidx = (unsigned)((band * 2 + (wave & 1)) * 4 + g4);
        const float* lbase = a.lists;
        auto epilogue = [&](auto masked) {
            constexpr bool MASKED = decltype(masked)::value;
            typedef float f32x2 __attribute__((ext_vector_type(2)));
#pragma unroll
            for (int ca = 0; ca < 4; ++ca) {
                const float qrv = qr_s[16 * ca];
                const float c = lst[256 * ca], th = lst[256 * (4 + ca)];
                // two-wide partial sums: the adds / fmas below are packed fp32 instructions (v_pk_add_f32, v_pk_fma_f32)
                f32x2 ps2 = {lst[256 * (8 + ca)], 0.f}, pq2 = {lst[256 * (12 + ca)], 0.f};
                unsigned o = cur[ca];
#pragma unroll
                for (int cb = 0; cb < 4; ++cb) {
                    f32x4 s4 = acc[ca][cb] + (qmv[cb] + qrv);   // the score, same bits as the spilling kernel
                    f32x4 d4 = s4 - c;                             // centred on the row's analytic mean
                    if (MASKED) {
#pragma unroll
                        for (int r = 0; r < 4; ++r) {
                            const bool okc = m0 + 16 * cb + 4 * g4 + r < a.M;
                            d4[r] = okc ? d4[r] : 0.f;
                            s4[r] = okc ? s4[r] : (LOWEST ? __builtin_inff() : -__builtin_inff());
                        }
                    }
                    const f32x2 dl = {d4[0], d4[1]}, dh = {d4[2], d4[3]};
                    ps2 += dl;
                    ps2 += dh;
                    pq2 = __builtin_elementwise_fma(dl, dl, pq2);
                    pq2 = __builtin_elementwise_fma(dh, dh, pq2);
#pragma unroll
                    for (int r = 0; r < 4; ++r) {
                        // if (s <= th) { lists[o] = s; o += stride; } (>= for the N largest) as one exec-masked store: no
                        // branch, no 64-bit address arithmetic (SGPR base + 32-bit byte offset), one VALU for the cursor
                        unsigned long long sv;
                        if (LOWEST)
                            asm volatile(
                                "v_cmp_le_f32 vcc, %[s], %[th]\n\t"
                                "s_and_saveexec_b64 %[sv], vcc\n\t"
                                "global_store_dword %[o], %[s], %[base]\n\t"
                                "v_add_u32 %[o], %[o], %[st]\n\t"
                                "s_mov_b64 exec, %[sv]"
                                : [o] "+v"(o), [sv] "=&s"(sv)
                                : [th] "v"(th), [s] "v"(s4[r]), [base] "s"(lbase), [st] "s"(stride_b)
                                : "vcc", "memory");
                        else
                            asm volatile(
                                "v_cmp_ge_f32 vcc, %[s], %[th]\n\t"
                                "s_and_saveexec_b64 %[sv], vcc\n\t"
                                "global_store_dword %[o], %[s], %[base]\n\t"
                                "v_add_u32 %[o], %[o], %[st]\n\t"
                                "s_mov_b64 exec, %[sv]"
                                : [o] "+v"(o), [sv] "=&s"(sv)
                                : [th] "v"(th), [s] "v"(s4[r]), [base] "s"(lbase), [st] "s"(stride_b)
                                : "vcc", "memory");
                    }
                }
                lst[256 * (8 + ca)] = ps2[0] + ps2[1];
                lst[256 * (12 + ca)] = pq2[0] + pq2[1];
                // at most kSubFull entries stay: slot kSubFull absorbs what a full sub-list still receives (the select
                // kernel treats a count of kSubFull as an overflow)
                long long rc = rlane + 16 * ca;
                if (rc >= a.R) rc = a.R - 1;
                const unsigned lim = 4u * ((unsigned)((rc * kSub + kSubFull) * a.nsub) + sidx);
                cur[ca] = o < lim ? o : lim;
            }
        };
        if (m0 - (wave & 1) * 64 + 128 > a.M) epilogue(std::true_type{});
        else epilogue(std::false_type{});

        if (last_tile) {
            item_end(rb, band);
            if (!have_next) break;
            rb = nrb; band = nband; tx = ntx0; tx1 = ntx1;
            item_begin(rb, band);
        } else {
            ++tx;
        }
        if (a.ksteps == 1) __syncthreads();
        qpar ^= 1;
    }
}

// ------------------------------------------------------------------------------------------------------------------
// select: one wave per row
// ------------------------------------------------------------------------------------------------------------------
struct FinishArgs {
    const float* lists; const unsigned* counts; const double* part; const float* crow;
    long long R, M;
    const float* trow;
    float zhi, fhi;                       // the proposal: t_r = c_r + sgn zhi sd_r, fhi = Phi(zhi) = proposed fraction
    int nsub, nsub_valid, topn, lowest;   // sub-lists [nsub_valid, nsub) belong to bands past the last column tile: never written
    unsigned* nfail; unsigned* fail_rows;
    double* stats;
};

__global__ __launch_bounds__(256) void cohort_finish_kernel(const FinishArgs a) {
    __shared__ unsigned keys[4][kCandMax];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const long long row = (long long)blockIdx.x * 4 + wave;
    if (row >= a.R) return;
    unsigned* kl = keys[wave];
    const unsigned* cnt = a.counts + (size_t)row * a.nsub;
    // counts (lane l holds sub-lists l, l + 64, ...) -> total, overflow, exclusive prefix in sub-list order
    unsigned creg[4], pre[4];
    unsigned total = 0, ovf = 0;
#pragma unroll
    for (int j = 0; j < 4; ++j) {
        const int s = 64 * j + lane;
        creg[j] = s < a.nsub_valid ? cnt[s] : 0u;
        ovf |= creg[j] >= (unsigned)kSubFull ? 1u : 0u;
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
    if (ovf != 0 || total < (unsigned)N || total > (unsigned)kCandMax) {
        if (lane == 0) a.fail_rows[atomicAdd(a.nfail, 1u)] = (unsigned)row;
        return;
    }
    // gather the sub-lists (lane = sub-list, slot rows are contiguous) into this wave's LDS run as order-preserving
    // keys: element e of sub-list s lands at pre[s] + e — a fixed order
    const float* lrow = a.lists + (size_t)row * kSub * a.nsub;
#pragma unroll
    for (int j = 0; j < 4; ++j) {
        if (64 * j >= a.nsub_valid) break;
        const unsigned maxc = wave_max_u32(creg[j]);
        for (unsigned e0 = 0; e0 < maxc; e0 += 16) {
            float v[16];
#pragma unroll
            for (int u = 0; u < 16; ++u) {
                // unconditional (slot index clamped into the row's list region): a load under a lane predicate is a
                // branch with its own wait, and the 30-odd slot rows of a row then come in one memory round trip each
                const unsigned e = e0 + u < (unsigned)kSub ? e0 + u : (unsigned)kSub - 1;
                v[u] = lrow[(size_t)e * a.nsub + 64 * j + lane];
            }
#pragma unroll
            for (int u = 0; u < 16; ++u) {
                const unsigned e = e0 + u;
                if (e < creg[j]) kl[pre[j] + e] = f2key(a.lowest ? v[u] : -v[u]);
            }
        }
    }
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
        const float c = a.crow[row], t = a.trow[row];
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
            const double v = (double)(a.lowest ? w : -w);  // the raw score behind the ordered key
            t1 += v;
            t2 += v * v;
            ++nless;
        }
    }
    t1 = wave_sum_f64(t1);
    t2 = wave_sum_f64(t2);
    nless = wave_sum_u32(nless);
    // whole-row sums: the (band, wave column) partials in their fixed order
    double d1 = 0.0, d2 = 0.0;
    const double* pr = a.part + (size_t)row * (a.nsub / 4) * 2;
    for (int i = lane; i < a.nsub_valid / 4; i += 64) {
        d1 += pr[2 * i];
        d2 += pr[2 * i + 1];
    }
    d1 = wave_sum_f64(d1);
    d2 = wave_sum_f64(d2);
    if (lane == 0) {
        const double n = (double)a.M, nn = (double)N;
        const double mw = d1 / n;
        double var = d2 / n - mw * mw;
        if (var < 0.0) var = 0.0;
        const double mean_s = (double)a.crow[row] + mw;
        double tv = (double)key2f(tkey);           // the threshold in ordered space -> raw score
        if (!a.lowest) tv = -tv;
        const double ties = nn - (double)nless;
        t1 += ties * tv;
        t2 += ties * tv * tv;
        const double mt = t1 / nn;
        double vt = t2 / nn - mt * mt;
        if (vt < 0.0) vt = 0.0;
        double* o = a.stats + row * 4;
        o[0] = mean_s;
        o[1] = sqrt(var);
        o[2] = mt;
        o[3] = sqrt(vt);
    }
}

inline size_t align256(size_t v) { return (v + 255) / 256 * 256; }

}  // namespace

namespace nplda {

// inverse normal CDF (Abramowitz & Stegun 26.2.23), host side
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
    // candidates proposed per row: about twice the wanted count, and relatively more when N is small — the proposal
    // sits far out in the tail there, where a row's distribution agrees least with the normal model
    const double want = 2.0 * topn + 16.0 + 300.0 * exp(-(double)topn / 300.0);
    const double f = want / (double)M;
    if (f > 0.25 || want > 0.8 * kCandMax) return p;
    const long long nx = (M + 127) / 128;
    p.nx = (int)nx;
    p.nxp = (int)((nx + 7) / 8 < 24 ? (nx + 7) / 8 : 24);
    if (const char* e = getenv("NPLDA_FUSED_NXP")) {  // tuning knob: column tiles per work item
        const int v = atoi(e);
        if (v >= 1 && v <= 24) p.nxp = v;
    }
    p.nsb = (int)((nx + 8LL * p.nxp - 1) / (8LL * p.nxp));
    p.nsub = 8 * p.nsb * 8;
    if (p.nsub > 256) return p;                       // the select kernel holds the sub-list counts in 4 registers
    // expected candidates of one sub-list (a band's columns seen by one lane group of one wave column)
    const double lam = f * (double)p.nxp * 128.0 / 8.0;
    if (lam + 6.0 * sqrt(lam) + 4.0 > (double)kSubFull) return p;
    p.zhi = host_normcdfinv(f);
    p.fhi = (float)f;
    const size_t kb = (size_t)Mp / 16;
    p.fixed_bytes = 256 + align256((size_t)kGramSplit * Mp * Mp * 4) + align256((size_t)kGramSplit * 4 * Mp * 4) +
                    align256((size_t)kQzBlocks * (Mp + 2) * 4) + align256(kb * kb * 256 * 4) +
                    align256((size_t)(2 * Mp + 2) * 4) + 8 * 256;  // + the alignment slack of the per-row arrays
    p.max_rows = ((1LL << 30) / ((long long)p.nsub * kSub)) / 128 * 128;
    p.row_bytes = (size_t)Mp * 4 + 8 + (size_t)p.nsub * kSub * 4 + (size_t)p.nsub * 4 + (size_t)(p.nsub / 4) * 16 + 4;
    p.eligible = true;
    return p;
}

// Runs the fused path on rows [0, R) (R <= the rows the workspace was planned for).  `fail_rows` / `nfail` (device)
// receive the rows that need the general path; the caller runs cohort_fallback_kernel on them.
int cohort_fused_run(const FusedPlan& p, const float* z_rows, const float* q_rows, long long R, const float* z_coh,
                     const float* q_coh, long long M, long long ldz, const float* P, int ksteps, int topn, int lowest,
                     double* stats, unsigned char* ws, long long rows_cap, bool prepass, unsigned** fail_rows_out,
                     unsigned** nfail_out, long long resident, hipStream_t st) {
    const int Mp = 16 * ksteps;
    unsigned char* q = ws;
    unsigned* ctl = reinterpret_cast<unsigned*>(q); q += 256;
    float* slab = reinterpret_cast<float*>(q); q += align256((size_t)kGramSplit * Mp * Mp * 4);
    float* ext = reinterpret_cast<float*>(q); q += align256((size_t)kGramSplit * 4 * Mp * 4);
    float* qz = reinterpret_cast<float*>(q); q += align256((size_t)kQzBlocks * (Mp + 2) * 4);
    float* frag = reinterpret_cast<float*>(q); q += align256((size_t)(Mp / 16) * (Mp / 16) * 256 * 4);
    float* vec = reinterpret_cast<float*>(q); q += align256((size_t)(2 * Mp + 2) * 4);
    // per-row arrays, sized for rows_cap rows
    double* part = reinterpret_cast<double*>(q); q += align256((size_t)rows_cap * (p.nsub / 4) * 16);
    float* lists = reinterpret_cast<float*>(q); q += align256((size_t)rows_cap * p.nsub * kSub * 4);
    unsigned* counts = reinterpret_cast<unsigned*>(q); q += align256((size_t)rows_cap * p.nsub * 4);
    float* tmp = reinterpret_cast<float*>(q); q += align256((size_t)rows_cap * Mp * 4);
    float* crow = reinterpret_cast<float*>(q); q += align256((size_t)rows_cap * 4);
    float* trow = reinterpret_cast<float*>(q); q += align256((size_t)rows_cap * 4);
    unsigned* fail_rows = reinterpret_cast<unsigned*>(q);
    *fail_rows_out = fail_rows;
    *nfail_out = ctl + 8;

    if (hipMemsetAsync(ctl, 0, 256, st) != hipSuccess) return NPLDA_EINVAL;
    if (prepass) {  // cohort moments: once per call, the cohort does not change between row chunks
        if (int rc = gram_slabs_launch(z_coh, ldz, M, Mp, kGramSplit, slab, ext, st)) return rc;
        hipLaunchKernelGGL(cohort_qz_kernel, dim3(kQzBlocks), dim3(256), 0, st, z_coh, q_coh, M, ldz, Mp, qz);
        if (int rc = nplda_launch_status()) return rc;
        PrepArgs pa = {slab, ext, qz, P, kGramSplit, Mp, M, frag, vec};
        // ksplit actually used by gram_slabs_launch: rows per split rounded up -> some trailing slabs may be unwritten
        {
            long long rps = (M + kGramSplit - 1) / kGramSplit;
            rps = (rps + 63) / 64 * 64;
            pa.ksplit = (int)((M + rps - 1) / rps);
        }
        const size_t nfrag = (size_t)(Mp / 16) * (Mp / 16) * 256;
        hipLaunchKernelGGL(cohort_prep_kernel, dim3((unsigned)((nfrag + 255) / 256)), dim3(256), 0, st, pa);
        if (int rc = nplda_launch_status()) return rc;
    }
    if (int rc = rows_matmul_launch(z_rows, ldz, R, Mp, frag, Mp, tmp, Mp, st)) return rc;
    hipLaunchKernelGGL(cohort_threshold_kernel, dim3((unsigned)((R + 3) / 4)), dim3(256), 0, st, z_rows, q_rows, tmp, R,
                       ldz, Mp, vec, p.zhi, lowest ? 1.0f : -1.0f, crow, trow);
    if (int rc = nplda_launch_status()) return rc;

    FusedArgs fa = {};
    fa.zr = z_rows; fa.qr = q_rows; fa.zc = z_coh; fa.qc = q_coh; fa.P = P;
    fa.R = R; fa.M = M; fa.ldz = ldz; fa.ksteps = ksteps; fa.nxp = p.nxp; fa.ny = (int)((R + 127) / 128); fa.nx = p.nx;
    fa.nsb = p.nsb; fa.ctr = ctl; fa.crow = crow; fa.trow = trow; fa.lists = lists; fa.counts = counts; fa.part = part;
    fa.nsub = p.nsub;
    long long grid = 8LL * fa.ny * p.nsb;  // at most one block per work item of the busiest XCD
    if (grid > resident) grid = resident;
    if (lowest) hipLaunchKernelGGL((cohort_fused_kernel<true, NPLDA_FUSED_MINBLOCKS>), dim3((unsigned)grid), dim3(256), 0, st, fa);
    else hipLaunchKernelGGL((cohort_fused_kernel<false, NPLDA_FUSED_MINBLOCKS>), dim3((unsigned)grid), dim3(256), 0, st, fa);
    if (int rc = nplda_launch_status()) return rc;

    const int nbands = (p.nx + p.nxp - 1) / p.nxp;  // bands that hold column tiles
    FinishArgs fi = {lists, counts, part, crow, R, M, trow, p.zhi, p.fhi, p.nsub, nbands * 8, topn, lowest, ctl + 8,
                     fail_rows, stats};
    hipLaunchKernelGGL(cohort_finish_kernel, dim3((unsigned)((R + 3) / 4)), dim3(256), 0, st, fi);
    return nplda_launch_status();
}

long long cohort_fused_resident_blocks() {
    int dev = 0, cus = 0, per_cu = 0;
    if (hipGetDevice(&dev) != hipSuccess ||
        hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, dev) != hipSuccess ||
        hipOccupancyMaxActiveBlocksPerMultiprocessor(&per_cu, cohort_fused_kernel<true, NPLDA_FUSED_MINBLOCKS>, 256, 0) != hipSuccess)
        return 0;
    long long r = (long long)cus * per_cu / 8 * 8;
    return r < 8 ? 8 : r;
}

}  // namespace nplda

// nplda_matmul.hip — the input-side half of the hand-derived backward (gfx950, fp32 MFMA) and the small kernels
// around it.
//
// The reference's E2E model (utils/models.py:216-268) puts the same NPLDA head on top of a trainable x-vector
// extractor, and every head function (extract_plda_embeddings :366-370, forward_from_plda_embeddings :372-376,
// forward :378-382, DPlda.forward :484-495) is differentiable w.r.t. its INPUTS through autograd.  The weight-side
// gradients live in nplda_backward.hip; here are the pieces the input side adds:
//
//  * rows_matmul_kernel     out[r, :] = rowscale[r] * (in[r, :] . Wm + bias)  for a small resident matrix Wm (K x N,
//                           K <= 512): dx = du . W1 (K = padded D1, N = 512) and DPlda's d[y1; y2] = g (x (M + M^T) + v).
//                           Same transposed formulation as the forward: Wm^T is the MFMA A operand, read as
//                           fragment-ordered ds_read_b128 from an LDS-resident column slice that a persistent block
//                           loads ONCE; the data rows are the B operand, one float4 per lane per k16-step straight from
//                           row-major memory (the k-permutation of nplda_fwd_kernel.h), prefetched a step ahead;
//                           each lane ends up with 4 consecutive output columns -> 16-byte stores.
//  * frag_pack_kernel / frag_from_packed_kernel   build that fragment image from a row-major matrix (optionally
//                           transposed or symmetrised) or from the W1 fragments of the packed parameter image.
//  * pad_rows_kernel        (N, D) rows with any stride -> (N, ldz) zero-padded rows (upstream dL/dz of an embedding).
//  * normalize_bwd_paired_kernel   F.normalize backward on paired rows [y1 | y2] (DPlda / GaussianBackend layout).
//  * emb_score_bwd_kernel   backward of forward_from_plda_embeddings on explicit (B, D2) tensors: dz1, dz2 and
//                           fixed-order column sums for dQ, dP_sqrt.
#include "nplda_fwd_dispatch.h"

namespace {

using namespace nplda;

// ------------------------------------------------------------------------------------------------------------------
// fragment images:  frag[kb][xb][lane][i] = Wm[16 kb + 4 (lane >> 4) + i][16 xb + (lane & 15)],  0 outside K x N
// ------------------------------------------------------------------------------------------------------------------
// mode 3: src = DPlda's logistic_regres.weight [Wb | Ww | ws] (utils/models.py:484-490), ld = D1, K = N = 2 D1:
// Wm = M + M^T for M = [[Ww, Wb], [Wb, Ww]], and vout (2 D1) = [ws; ws] — the gradient of the quadratic form w.r.t. the
// paired rows is g ((M + M^T) x + v)
__global__ void frag_pack_kernel(const float* __restrict__ src, long long ld, int K, int N, int mode, int KB, int XB,
                                 float* __restrict__ out, float* __restrict__ vout) {
    const size_t idx = (size_t)blockIdx.x * 256 + threadIdx.x;
    if (idx >= (size_t)KB * XB * 256) {
        const size_t t = idx - (size_t)KB * XB * 256;
        if (mode == 3 && vout != nullptr && t < (size_t)K) vout[t] = src[2 * ld * ld + (long long)(t % (size_t)ld)];
        return;
    }
    const int i = (int)(idx & 3), lane = (int)((idx >> 2) & 63);
    const size_t blk = idx >> 8;
    const int xb = (int)(blk % XB), kb = (int)(blk / XB);
    const int k = 16 * kb + 4 * (lane >> 4) + i, n = 16 * xb + (lane & 15);
    float v = 0.f;
    if (k < K && n < N) {
        if (mode == 0) v = src[(size_t)k * ld + n];
        else if (mode == 1) v = src[(size_t)n * ld + k];
        else if (mode == 2) v = src[(size_t)k * ld + n] + src[(size_t)n * ld + k];  // Wm = src + src^T (K == N)
        else {
            const int D1 = (int)ld;
            auto M = [&](int r, int c) {
                const float* blk = ((r < D1) == (c < D1)) ? src + (size_t)D1 * D1 : src;
                return blk[(size_t)(r % D1) * D1 + (c % D1)];
            };
            v = M(k, n) + M(n, k);
        }
    }
    out[idx] = v;
}

// Wm = W1 (D1 x D0) re-read from the forward's fragment image W1p[ks][nb][lane][i] = W1[16 nb + (lane & 15)]
// [16 ks + 4 (lane >> 4) + i] (nplda_common.h), so that the backward needs no second copy of the raw parameter.
__global__ void frag_from_packed_kernel(const float* __restrict__ packed, int NB, int KS1, float* __restrict__ out) {
    const size_t idx = (size_t)blockIdx.x * 256 + threadIdx.x;
    if (idx >= (size_t)NB * KS1 * 256) return;
    const int i = (int)(idx & 3), lane = (int)((idx >> 2) & 63);
    const size_t blk = idx >> 8;
    const int xb = (int)(blk % KS1), kb = (int)(blk / KS1);
    const int f = 16 * kb + 4 * (lane >> 4) + i;  // layer-1 feature (k of the product)
    const int c = 16 * xb + (lane & 15);          // x-vector column (n of the product)
    const int nb = f >> 4, ks = c >> 4;
    const int lsrc = (f & 15) + 16 * ((c & 15) >> 2), isrc = c & 3;
    out[idx] = packed[(((size_t)ks * NB + nb) * 64 + lsrc) * 4 + isrc];
}

// ------------------------------------------------------------------------------------------------------------------
// rows_matmul
// ------------------------------------------------------------------------------------------------------------------
struct MatmulArgs {
    const float* in;      // (R, ldin), K valid columns
    long long ldin, R;
    int K, KB, XB, N;
    const f32x4* frag;    // [KB][XB][64]
    float* out0;          // rows [0, nsplit)
    float* out1;          // rows [nsplit, R)
    long long nsplit, ldout;
    const float* bias;      // (N) or null
    const float* rowscale;  // (R) or null
    int ntiles;             // tiles of 128 rows
};

// WAVES: 4, or 8 when the resident slice is so large (> 53 KB: dx at D1 >= 112) that only ONE block fits a CU — four
// waves would then be one per SIMD with every load latency exposed (dx at 262 144 pairs: 0.60 of the peak at D = 150,
// 0.47 at 170); eight waves share the same slice, two per SIMD.
template <int NS, int WAVES = 4>
__global__ __launch_bounds__(WAVES * 64) void rows_matmul_kernel(const MatmulArgs a) {
    extern __shared__ f32x4 wl[];  // [KB][NS][64]
    const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int j = lane & 15, g = lane >> 4;
    const int xb0 = blockIdx.y * NS;
    const int KB = a.KB;
    for (int idx = tid; idx < KB * NS * 64; idx += WAVES * 64) {
        const int kb = idx / (NS * 64), rem = idx - kb * (NS * 64), u = rem >> 6, l = rem & 63;
        const int xb = xb0 + u;
        const f32x4 zero = {0.f, 0.f, 0.f, 0.f};
        wl[idx] = xb < a.XB ? a.frag[((size_t)kb * a.XB + xb) * 64 + l] : zero;
    }
    __syncthreads();
    const int kmax = a.K - 4;
    // the first k-block of a tile's rows is fetched while the PREVIOUS tile's results are being stored (a tile is only KB
    // steps long: its first load, out of HBM at streaming sizes, was exposed once per tile)
    auto rows_of = [&](int tile, long long (&row)[2], bool (&ok)[2], const float* (&src)[2]) {
        const long long r0 = ((long long)tile * WAVES + wave) * 32;
#pragma unroll
        for (int rg = 0; rg < 2; ++rg) {
            row[rg] = r0 + 16 * rg + j;
            ok[rg] = row[rg] < a.R;
            if (!ok[rg]) row[rg] = a.R - 1;
            src[rg] = a.in + row[rg] * a.ldin;
        }
    };
    auto ld4p = [&](const float* p, int kb) -> f32x4 {
        int c = 16 * kb + 4 * g;
        c = c < kmax ? c : kmax;
        return *reinterpret_cast<const f32x4*>(p + c);
    };
    long long nrow[2];
    bool nok[2];
    const float* nsrc[2];
    f32x4 ncur[2];
    if ((int)blockIdx.x < a.ntiles) {
        rows_of(blockIdx.x, nrow, nok, nsrc);
        ncur[0] = ld4p(nsrc[0], 0);
        ncur[1] = ld4p(nsrc[1], 0);
    }
    for (int tile = blockIdx.x; tile < a.ntiles; tile += gridDim.x) {
        long long row[2] = {nrow[0], nrow[1]};
        bool ok[2] = {nok[0], nok[1]};
        const float* src[2] = {nsrc[0], nsrc[1]};
        f32x4 acc[2][NS];
#pragma unroll
        for (int rg = 0; rg < 2; ++rg)
#pragma unroll
            for (int u = 0; u < NS; ++u) acc[rg][u] = f32x4{0.f, 0.f, 0.f, 0.f};
        // unconditional, clamped loads (see load_x4c in nplda_fwd_kernel.h): columns >= K re-read the row's last
        // float4, the fragment image is zero there
        auto ld4 = [&](int rg, int kb) -> f32x4 {
            int c = 16 * kb + 4 * g;
            c = c < kmax ? c : kmax;
            return *reinterpret_cast<const f32x4*>(src[rg] + c);
        };
        f32x4 cur[2] = {ncur[0], ncur[1]};
        for (int kb = 0; kb < KB; ++kb) {
            const int kn = kb + 1 < KB ? kb + 1 : KB - 1;
            const f32x4 nxt0 = ld4(0, kn), nxt1 = ld4(1, kn);
            f32x4 av[NS];
#pragma unroll
            for (int u = 0; u < NS; ++u) av[u] = wl[(kb * NS + u) * 64 + lane];
#pragma unroll
            for (int r = 0; r < 4; ++r) {
#pragma unroll
                for (int u = 0; u < NS; ++u) {
                    acc[0][u] = __builtin_amdgcn_mfma_f32_16x16x4f32(av[u][r], cur[0][r], acc[0][u], 0, 0, 0);
                    acc[1][u] = __builtin_amdgcn_mfma_f32_16x16x4f32(av[u][r], cur[1][r], acc[1][u], 0, 0, 0);
                }
            }
            cur[0] = nxt0;
            cur[1] = nxt1;
        }
        if (tile + (int)gridDim.x < a.ntiles) {  // the next tile's first loads, under this tile's stores
            rows_of(tile + gridDim.x, nrow, nok, nsrc);
            ncur[0] = ld4p(nsrc[0], 0);
            ncur[1] = ld4p(nsrc[1], 0);
        }
#pragma unroll
        for (int rg = 0; rg < 2; ++rg) {
            if (!ok[rg]) continue;
            const float rs = a.rowscale ? a.rowscale[row[rg]] : 1.0f;
            float* dst = row[rg] < a.nsplit ? a.out0 + row[rg] * a.ldout : a.out1 + (row[rg] - a.nsplit) * a.ldout;
#pragma unroll
            for (int u = 0; u < NS; ++u) {
                const int col = 16 * (xb0 + u) + 4 * g;
                if (col < a.N) {
                    f32x4 v = acc[rg][u];
                    if (a.bias) v += *reinterpret_cast<const f32x4*>(a.bias + col);
                    *reinterpret_cast<f32x4*>(dst + col) = v * rs;
                }
            }
        }
    }
}

int launch_rows_matmul(MatmulArgs a, hipStream_t st) {
    if (a.R <= 0) return NPLDA_OK;
    // column blocks per LDS-resident slice: 8 while KB * 8 KiB fits comfortably (dx: KB <= 12 -> <= 96 KiB), else 4
    // ... and 4 as well when 8-wide slices would leave the last slice mostly empty (N = 176: 8 + 3 blocks -> 4 + 4 + 3,
    // and 44 KiB of LDS instead of 88: three resident blocks per CU)
    const int NS = (a.KB * 8 <= 96 && (a.XB % 8 == 0 || a.XB >= 24)) ? 8 : 4;
    const size_t lds = (size_t)a.KB * NS * 1024;
    if (lds > 160 * 1024) return NPLDA_EUNSUPPORTED;
    const bool wide = NS == 8 && lds > 53 * 1024 && a.R >= 4096;  // one block per CU: give it eight waves
    const int rows_per_tile = wide ? 256 : 128;
    const long long nt = (a.R + rows_per_tile - 1) / rows_per_tile;
    if (nt > 0x7fffffffLL) return NPLDA_EINVAL;
    a.ntiles = (int)nt;
    const int slices = (a.XB + NS - 1) / NS;
    int dev = 0, cus = 0;
    if (hipGetDevice(&dev) != hipSuccess ||
        hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, dev) != hipSuccess || cus <= 0)
        cus = 256;
    long long gx = (2LL * cus + slices - 1) / slices;  // persistent: about two resident blocks per CU in total
    if (gx > nt) gx = nt;
    if (gx < 1) gx = 1;
    dim3 grid((unsigned)gx, (unsigned)slices), block(wide ? 512 : 256);
    hipError_t e;
    if (wide) {
        if (gx > (cus + slices - 1) / slices) gx = (cus + slices - 1) / slices;  // one resident block per CU
        if (gx < 1) gx = 1;
        grid.x = (unsigned)gx;
        e = hipFuncSetAttribute((const void*)rows_matmul_kernel<8, 8>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
        if (e != hipSuccess) return (int)e;
        hipLaunchKernelGGL((rows_matmul_kernel<8, 8>), grid, block, lds, st, a);
    } else if (NS == 8) {
        e = hipFuncSetAttribute((const void*)rows_matmul_kernel<8>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
        if (e != hipSuccess) return (int)e;
        hipLaunchKernelGGL(rows_matmul_kernel<8>, grid, block, lds, st, a);
    } else {
        e = hipFuncSetAttribute((const void*)rows_matmul_kernel<4>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
        if (e != hipSuccess) return (int)e;
        hipLaunchKernelGGL(rows_matmul_kernel<4>, grid, block, lds, st, a);
    }
    return nplda_launch_status();
}

// ------------------------------------------------------------------------------------------------------------------
__global__ void pad_rows_kernel(const float* __restrict__ in, long long ldin, long long N, int D, float* __restrict__ out,
                                long long ldo) {
    const long long idx = (long long)blockIdx.x * 256 + threadIdx.x;
    if (idx >= N * ldo) return;
    const long long r = idx / ldo;
    const int c = (int)(idx - r * ldo);
    out[idx] = c < D ? in[r * ldin + c] : 0.f;
}

// One wave per (pair k, side h): du[h B + k, :] = (dy - y (y . dy)) * rn[h B + k] with dy = dpaired[k, h D1 : (h+1) D1],
// y = paired[k, h D1 : ...]; the clamp branch of F.normalize (rn = 1e12) has dy / eps as its derivative.
__global__ __launch_bounds__(256) void normalize_bwd_paired_kernel(const float* __restrict__ dpaired, long long lddp,
                                                                   const float* __restrict__ paired, long long ldp,
                                                                   const float* __restrict__ rn, long long B, int D1,
                                                                   float* __restrict__ du, long long ldz) {
    const long long w = (long long)blockIdx.x * 4 + (threadIdx.x >> 6);
    const int lane = threadIdx.x & 63;
    if (w >= 2 * B) return;
    const long long h = w / B, k = w - h * B;
    const float* dy = dpaired + k * lddp + h * D1;
    const float* y = paired + k * ldp + h * D1;
    float dot = 0.f;
    for (int f = lane; f < D1; f += 64) dot = fmaf(y[f], dy[f], dot);
#pragma unroll
    for (int m = 32; m >= 1; m >>= 1) dot += __shfl_xor(dot, m, 64);
    const float r = rn[w];
    if (r >= 1e12f) dot = 0.f;
    for (int f = lane; f < ldz; f += 64) du[w * ldz + f] = f < D1 ? (dy[f] - y[f] * dot) * r : 0.f;
}

// ------------------------------------------------------------------------------------------------------------------
// backward of s = sum_d Q_d (z1_d^2 + z2_d^2) + 2 sum_d P_d z1_d z2_d  (utils/models.py:372-376), P = P_sqrt^2
// ------------------------------------------------------------------------------------------------------------------
constexpr int kEsbMaxCols = 3;   // columns per lane: D2 <= 192
constexpr int kEsbRows = 256;    // rows per block

__global__ __launch_bounds__(256) void emb_score_bwd_kernel(const float* __restrict__ z1, long long ld1,
                                                            const float* __restrict__ z2, long long ld2, long long B,
                                                            int D2, const float* __restrict__ P_sqrt,
                                                            const float* __restrict__ Q, const float* __restrict__ g,
                                                            float* __restrict__ dz1, long long ldd1,
                                                            float* __restrict__ dz2, long long ldd2,
                                                            float* __restrict__ part /* [blocks][2][D2] */) {
    __shared__ float red[4][2][kEsbMaxCols * 64];
    const int lane = threadIdx.x & 63, wy = threadIdx.x >> 6;
    float q[kEsbMaxCols], p[kEsbMaxCols], aq[kEsbMaxCols], ap[kEsbMaxCols];
#pragma unroll
    for (int c = 0; c < kEsbMaxCols; ++c) {
        const int f = lane + 64 * c;
        q[c] = f < D2 ? Q[f] : 0.f;
        p[c] = f < D2 ? P_sqrt[f] * P_sqrt[f] : 0.f;
        aq[c] = ap[c] = 0.f;
    }
    const long long r0 = (long long)blockIdx.x * kEsbRows;
    for (int rr = wy; rr < kEsbRows; rr += 4) {
        const long long r = r0 + rr;
        if (r >= B) break;
        const float gi = g[r], tg = 2.0f * gi;
#pragma unroll
        for (int c = 0; c < kEsbMaxCols; ++c) {
            const int f = lane + 64 * c;
            if (f < D2) {
                const float a = z1[r * ld1 + f], b = z2[r * ld2 + f];
                if (dz1) dz1[r * ldd1 + f] = tg * (q[c] * a + p[c] * b);
                if (dz2) dz2[r * ldd2 + f] = tg * (q[c] * b + p[c] * a);
                aq[c] = fmaf(gi, fmaf(a, a, b * b), aq[c]);
                ap[c] = fmaf(gi * a, b, ap[c]);
            }
        }
    }
#pragma unroll
    for (int c = 0; c < kEsbMaxCols; ++c) {
        red[wy][0][lane + 64 * c] = aq[c];
        red[wy][1][lane + 64 * c] = ap[c];
    }
    __syncthreads();
    for (int i = threadIdx.x; i < 2 * D2; i += 256) {
        const int which = i / D2, f = i - which * D2;
        part[(size_t)blockIdx.x * 2 * D2 + i] = ((red[0][which][f] + red[1][which][f]) + red[2][which][f]) + red[3][which][f];
    }
}

__global__ void emb_score_bwd_reduce_kernel(const float* __restrict__ part, int nblk, int D2,
                                            const float* __restrict__ P_sqrt, float* __restrict__ dP_sqrt,
                                            float* __restrict__ dQ) {
    const int i = blockIdx.x * 256 + threadIdx.x;
    if (i >= 2 * D2) return;
    float s = 0.f;
    for (int b = 0; b < nblk; ++b) s += part[(size_t)b * 2 * D2 + i];  // fixed order: deterministic
    const int which = i / D2, f = i - which * D2;
    if (which == 0) { if (dQ) dQ[f] = s; }
    else if (dP_sqrt) dP_sqrt[f] = 4.0f * P_sqrt[f] * s;  // d/dP_sqrt of 2 P z1 z2, P = P_sqrt^2
}

}  // namespace

namespace nplda {

// dx = du . W1 for the backward entry points (nplda_backward.hip): `frag` is workspace for the NB x KS1 fragment image.
// dx = du . W1 at minibatch sizes (<= 32 768 rows; 512-d x-vectors): one block per 32 rows, its 4 waves split the 32 output
// column blocks, W1^T fragments straight from the parameter image (L2) through a register ring of buffer loads — the
// LDS-resident form above loads an 80 KB slice per block to use it once and takes 27 us at 8 192 rows for 8.5 us of MFMA
// work.  Optionally writes bf16 (the dtype a jointly trained extractor handed the x-vectors over in).
struct DxSmallArgs {
    const float* du;
    long long ldz, rows, nsplit;
    const float* frag;    // [NB][32][64][4]: W1^T fragments (the parameter image's oW1T region, or a frag_pack_kernel image)
    size_t frag_bytes;
    void* dx0;
    void* dx1;
    long long lddx;
};

template <int NB, bool OBF>
__global__ __launch_bounds__(256, 1) void dx_small_kernel(const DxSmallArgs a) {
    constexpr int XBW = 8, KS1 = 32, PF = 2, PF1 = PF + 1;
    const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int j = lane & 15, g = lane >> 4;
    const long long r0 = (long long)blockIdx.x * 32;
    long long row[2];
    bool ok[2];
#pragma unroll
    for (int rg = 0; rg < 2; ++rg) {
        row[rg] = r0 + 16 * rg + j;
        ok[rg] = row[rg] < a.rows;
        if (!ok[rg]) row[rg] = a.rows - 1;
    }
    const __amdgpu_buffer_rsrc_t img = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(a.frag), 0, (int)a.frag_bytes, 0x00020000);
    unsigned voff[XBW];
#pragma unroll
    for (int u = 0; u < XBW; ++u) voff[u] = (unsigned)(((XBW * wave + u) * 64 + lane) * 16);
    f32x4 wf[PF1][XBW];
    auto fetchw = [&](int slot, int kb) {
        const int kbc = kb < NB ? kb : NB - 1;
        const int soff = kbc * (KS1 * 1024);
#pragma unroll
        for (int u = 0; u < XBW; ++u)
            wf[slot][u] = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(img, (int)voff[u], soff, 0));
    };
#pragma unroll
    for (int p = 0; p < PF; ++p) fetchw(p, p);
    f32x4 d[2][NB];  // the rows' du, all k-blocks: the B operand (accumulator layout of the backward = k-permuted rows)
#pragma unroll
    for (int rg = 0; rg < 2; ++rg)
#pragma unroll
        for (int kb = 0; kb < NB; ++kb) d[rg][kb] = *reinterpret_cast<const f32x4*>(a.du + row[rg] * a.ldz + 16 * kb + 4 * g);
    f32x4 acc[2][XBW];
    const f32x4 zero4 = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int kb = 0; kb < NB; ++kb) {
        const int sl = kb % PF1;
#pragma unroll
        for (int r = 0; r < 4; ++r) {
#pragma unroll
            for (int u = 0; u < XBW; ++u) {
                acc[0][u] = __builtin_amdgcn_mfma_f32_16x16x4f32(wf[sl][u][r], d[0][kb][r], (kb == 0 && r == 0) ? zero4 : acc[0][u], 0, 0, 0);
                acc[1][u] = __builtin_amdgcn_mfma_f32_16x16x4f32(wf[sl][u][r], d[1][kb][r], (kb == 0 && r == 0) ? zero4 : acc[1][u], 0, 0, 0);
            }
            if (r == 0 && kb + PF < NB) fetchw((kb + PF) % PF1, kb + PF);
            __builtin_amdgcn_sched_barrier(0);
        }
    }
#pragma unroll
    for (int rg = 0; rg < 2; ++rg) {
        if (!ok[rg]) continue;
        const bool first = row[rg] < a.nsplit;
        const long long orow = first ? row[rg] : row[rg] - a.nsplit;
#pragma unroll
        for (int u = 0; u < XBW; ++u) {
            const int col = 16 * (XBW * wave + u) + 4 * g;
            const f32x4 v = acc[rg][u];
            if constexpr (OBF) {  // round to nearest even, as torch's .to(bfloat16)
                unsigned short* dst = reinterpret_cast<unsigned short*>(first ? a.dx0 : a.dx1) + orow * a.lddx + col;
                unsigned w[4];
#pragma unroll
                for (int c = 0; c < 4; ++c) {
                    const unsigned b = __float_as_uint(v[c]);
                    w[c] = (b & 0x7fffffffu) > 0x7f800000u ? ((b >> 16) | 0x40u) : ((b + 0x7fffu + ((b >> 16) & 1u)) >> 16);
                }
                typedef unsigned u32x2 __attribute__((ext_vector_type(2)));
                *reinterpret_cast<u32x2*>(dst) = u32x2{w[0] | (w[1] << 16), w[2] | (w[3] << 16)};
            } else {
                float* dst = reinterpret_cast<float*>(first ? a.dx0 : a.dx1) + orow * a.lddx + col;
                *reinterpret_cast<f32x4*>(dst) = v;
            }
        }
    }
}

static int launch_dx_small(const float* du, long long ldz, long long rows, long long nsplit, const float* frag, int NB, void* dx0,
                           void* dx1, long long lddx, bool out_bf16, hipStream_t st) {
    DxSmallArgs a = {du, ldz, rows, nsplit, frag, (size_t)NB * 32 * 1024, dx0, dx1, lddx};
    const dim3 grid((unsigned)((rows + 31) / 32)), block(256);
#define NPLDA_DX(NBV)                                                                              \
    if (out_bf16) hipLaunchKernelGGL((dx_small_kernel<NBV, true>), grid, block, 0, st, a);          \
    else hipLaunchKernelGGL((dx_small_kernel<NBV, false>), grid, block, 0, st, a)
    switch (NB) {
        case 8: NPLDA_DX(8); break;
        case 9: NPLDA_DX(9); break;
        case 10: NPLDA_DX(10); break;
        case 11: NPLDA_DX(11); break;
        case 12: NPLDA_DX(12); break;
        default: return NPLDA_EUNSUPPORTED;
    }
#undef NPLDA_DX
    return nplda_launch_status();
}

// dx0 / dx1: fp32 (lddx in floats) or, out_bf16, bfloat16 (lddx in elements) — the bf16 form exists for the minibatch kernel only
int input_grad_from_du(const float* du, long long rows, long long ldz, const float* packed, const NpldaLayout& L,
                       float* /*frag: the W1^T image now rides in the parameter image*/, void* dx0, void* dx1, long long nsplit,
                       long long lddx, hipStream_t st, bool out_bf16) {
    if (rows <= 0) return NPLDA_OK;
    if (rows <= 32 * 1024 && L.D0 == 512 && L.KS1 == 32 && L.NB >= 8) {
        return launch_dx_small(du, ldz, rows, nsplit, packed + L.oW1T, L.NB, dx0, dx1, lddx, out_bf16, st);
    }
    if (out_bf16) return NPLDA_EUNSUPPORTED;
    MatmulArgs a = {};
    a.in = du; a.ldin = ldz; a.R = rows; a.K = 16 * L.NB; a.KB = L.NB; a.XB = L.KS1; a.N = L.D0;
    a.frag = reinterpret_cast<const f32x4*>(packed + L.oW1T);
    a.out0 = (float*)dx0; a.out1 = (float*)dx1; a.nsplit = nsplit; a.ldout = lddx;
    return launch_rows_matmul(a, st);
}

// out (R, ldout) = in[:, :K] . Wm for a fragment image built by frag_pack_kernel's layout (nplda_cohort_fused.hip)
int rows_matmul_launch(const float* in, long long ldin, long long R, int K, const float* frag, int N, float* out,
                       long long ldout, hipStream_t st) {
    MatmulArgs a = {};
    a.in = in; a.ldin = ldin; a.R = R; a.K = K; a.KB = (K + 15) / 16; a.XB = (N + 15) / 16; a.N = N;
    a.frag = reinterpret_cast<const f32x4*>(frag);
    a.out0 = out; a.out1 = out; a.nsplit = R; a.ldout = ldout;
    return launch_rows_matmul(a, st);
}

int pad_rows(const float* in, long long ldin, long long N, int D, float* out, long long ldo, hipStream_t st) {
    if (N <= 0) return NPLDA_OK;
    const long long total = N * ldo;
    hipLaunchKernelGGL(pad_rows_kernel, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, st, in, ldin, N, D, out, ldo);
    return nplda_launch_status();
}

}  // namespace nplda

extern "C" {

size_t nplda_matrix_frag_bytes(int K, int N) {
    if (K <= 0 || N <= 0 || K > 512 || (N % 4) != 0) return 0;
    return (size_t)((K + 15) / 16) * ((N + 15) / 16) * 256 * sizeof(float);
}

int nplda_pack_matrix_f32(const float* Wm, int64_t ldw, int K, int N, int mode, void* frag, size_t frag_bytes,
                          nplda_stream_t stream) {
    if (!Wm || !frag || mode < 0 || mode > 2 || !nplda_aligned16(frag)) return NPLDA_EINVAL;
    const size_t need = nplda_matrix_frag_bytes(K, N);
    if (need == 0) return NPLDA_EUNSUPPORTED;
    if (mode == 2 && K != N) return NPLDA_EINVAL;
    if (ldw < (mode == 1 ? K : N)) return NPLDA_EINVAL;
    if (frag_bytes < need) return NPLDA_ENOSPC;
    const int KB = (K + 15) / 16, XB = (N + 15) / 16;
    hipLaunchKernelGGL(frag_pack_kernel, dim3((unsigned)((need / 4 + 255) / 256)), dim3(256), 0, (hipStream_t)stream, Wm,
                       (long long)ldw, K, N, mode, KB, XB, (float*)frag, (float*)nullptr);
    return nplda_launch_status();
}

int nplda_dplda_quadform_f32(const float* wlr, int D1, void* frag, size_t frag_bytes, float* v, nplda_stream_t stream) {
    if (!wlr || !frag || !v || D1 <= 0 || !nplda_aligned16(frag)) return NPLDA_EINVAL;
    const int K = 2 * D1;
    const size_t need = nplda_matrix_frag_bytes(K, K);
    if (need == 0) return NPLDA_EUNSUPPORTED;
    if (frag_bytes < need) return NPLDA_ENOSPC;
    const int KB = (K + 15) / 16;
    hipLaunchKernelGGL(frag_pack_kernel, dim3((unsigned)((need / 4 + K + 255) / 256)), dim3(256), 0, (hipStream_t)stream, wlr,
                       (long long)D1, K, K, 3, KB, KB, (float*)frag, v);
    return nplda_launch_status();
}

int nplda_rows_matmul_f32(const float* in, int64_t ldin, int64_t R, int K, const void* frag, int N, const float* bias,
                          const float* rowscale, float* out, int64_t ldout, nplda_stream_t stream) {
    if (R < 0) return NPLDA_EINVAL;
    if (nplda_matrix_frag_bytes(K, N) == 0 || (K % 4) != 0) return NPLDA_EUNSUPPORTED;
    if (R == 0) return NPLDA_OK;
    if (!frag || !nplda_aligned16(frag) || !rows_ok(in, ldin, K) || !rows_ok(out, ldout, N)) return NPLDA_EINVAL;
    if (bias && !nplda_aligned16(bias)) return NPLDA_EINVAL;
    MatmulArgs a = {};
    a.in = in; a.ldin = ldin; a.R = R; a.K = K; a.KB = (K + 15) / 16; a.XB = (N + 15) / 16; a.N = N;
    a.frag = reinterpret_cast<const f32x4*>(frag);
    a.out0 = out; a.out1 = out; a.nsplit = R; a.ldout = ldout; a.bias = bias; a.rowscale = rowscale;
    return launch_rows_matmul(a, (hipStream_t)stream);
}

int nplda_normalize_bwd_paired_f32(const float* dpaired, int64_t lddp, const float* paired, int64_t ldp,
                                   const float* rn, int64_t B, int D1, float* du, int64_t ldz,
                                   nplda_stream_t stream) {
    if (B < 0 || D1 <= 0 || ldz < D1 || lddp < 2 * D1 || ldp < 2 * D1) return NPLDA_EINVAL;
    if (B == 0) return NPLDA_OK;
    if (!dpaired || !paired || !rn || !du) return NPLDA_EINVAL;
    const long long waves = 2 * B;
    hipLaunchKernelGGL(normalize_bwd_paired_kernel, dim3((unsigned)((waves + 3) / 4)), dim3(256), 0, (hipStream_t)stream,
                       dpaired, (long long)lddp, paired, (long long)ldp, rn, (long long)B, D1, du, (long long)ldz);
    return nplda_launch_status();
}

size_t nplda_score_embeddings_bwd_workspace_bytes(int64_t B, int D2) {
    if (B < 0 || D2 <= 0 || D2 > 64 * kEsbMaxCols) return 0;
    const size_t nblk = (size_t)((B + kEsbRows - 1) / kEsbRows);
    return (nblk > 0 ? nblk : 1) * 2 * (size_t)D2 * sizeof(float);
}

int nplda_score_embeddings_bwd_f32(const float* z1, int64_t ld1, const float* z2, int64_t ld2, int64_t B, int D2,
                                   const float* P_sqrt, const float* Q, const float* g, float* dz1, int64_t ldd1,
                                   float* dz2, int64_t ldd2, float* dP_sqrt, float* dQ, void* ws, size_t ws_bytes,
                                   nplda_stream_t stream) {
    if (B < 0 || D2 <= 0) return NPLDA_EINVAL;
    if (D2 > 64 * kEsbMaxCols) return NPLDA_EUNSUPPORTED;
    hipStream_t st = (hipStream_t)stream;
    if (B == 0) {
        if (dP_sqrt && hipMemsetAsync(dP_sqrt, 0, D2 * sizeof(float), st) != hipSuccess) return NPLDA_EINVAL;
        if (dQ && hipMemsetAsync(dQ, 0, D2 * sizeof(float), st) != hipSuccess) return NPLDA_EINVAL;
        return NPLDA_OK;
    }
    if (!z1 || !z2 || !P_sqrt || !Q || !g || !ws) return NPLDA_EINVAL;
    if (ld1 < D2 || ld2 < D2 || (dz1 && ldd1 < D2) || (dz2 && ldd2 < D2)) return NPLDA_EINVAL;
    if (ws_bytes < nplda_score_embeddings_bwd_workspace_bytes(B, D2)) return NPLDA_ENOSPC;
    const long long nblk = (B + kEsbRows - 1) / kEsbRows;
    if (nblk > 0x7fffffffLL) return NPLDA_EINVAL;
    hipLaunchKernelGGL(emb_score_bwd_kernel, dim3((unsigned)nblk), dim3(256), 0, st, z1, (long long)ld1, z2,
                       (long long)ld2, (long long)B, D2, P_sqrt, Q, g, dz1, (long long)ldd1, dz2, (long long)ldd2,
                       (float*)ws);
    if (int rc = nplda_launch_status()) return rc;
    hipLaunchKernelGGL(emb_score_bwd_reduce_kernel, dim3((unsigned)((2 * D2 + 255) / 256)), dim3(256), 0, st,
                       (const float*)ws, (int)nblk, D2, P_sqrt, dP_sqrt, dQ);
    return nplda_launch_status();
}

size_t nplda_lda_dgrad_workspace_bytes(int D0, int D1) {
    if (D0 <= 0 || D1 <= 0 || (D0 % 4) != 0 || nplda_kernel_nb(D1, D1) == 0) return 0;
    return nplda_matrix_frag_bytes(16 * ((D1 + 15) / 16), D0);
}

int nplda_lda_dgrad_f32(const float* du, int64_t ldz, int64_t B, const float* W1, int D0, int D1, void* ws,
                        size_t ws_bytes, float* dx1, float* dx2, int64_t lddx, nplda_stream_t stream) {
    if (B < 0) return NPLDA_EINVAL;
    const size_t need = nplda_lda_dgrad_workspace_bytes(D0, D1);
    if (need == 0) return NPLDA_EUNSUPPORTED;
    if (B == 0) return NPLDA_OK;
    const int KB = (D1 + 15) / 16, Kp = 16 * KB;
    if (!W1 || !ws || !nplda_aligned16(ws) || !rows_ok(du, ldz, Kp) || !rows_ok(dx1, lddx, D0) || !rows_ok(dx2, lddx, D0))
        return NPLDA_EINVAL;
    if (ws_bytes < need) return NPLDA_ENOSPC;
    hipStream_t st = (hipStream_t)stream;
    const int XB = (D0 + 15) / 16;
    hipLaunchKernelGGL(frag_pack_kernel, dim3((unsigned)((need / 4 + 255) / 256)), dim3(256), 0, st, W1, (long long)D0,
                       D1, D0, 0, KB, XB, (float*)ws, (float*)nullptr);
    if (int rc = nplda_launch_status()) return rc;
    if (2 * B <= 32 * 1024 && D0 == 512 && KB >= 8 && KB <= 12)  // minibatch sizes: fragments straight from L2, no LDS slice
        return nplda::launch_dx_small(du, ldz, 2 * B, B, (const float*)ws, KB, dx1, dx2, lddx, false, st);
    MatmulArgs a = {};
    a.in = du; a.ldin = ldz; a.R = 2 * B; a.K = Kp; a.KB = KB; a.XB = XB; a.N = D0;
    a.frag = reinterpret_cast<const f32x4*>(ws);
    a.out0 = dx1; a.out1 = dx2; a.nsplit = B; a.ldout = lddx;
    return launch_rows_matmul(a, st);
}

}  // extern "C"

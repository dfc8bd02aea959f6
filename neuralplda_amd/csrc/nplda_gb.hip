// nplda_gb.hip — GaussianBackend.forward (utils/models.py:584-593) fused on gfx950.
//
// LDA + F.normalize for both sides (the layer-1 half of the NPLDA forward kernel), then the difference
// of two full-covariance Gaussian log-likelihoods on the concatenated pair x = [y1; y2] (2 D1 = 340
// dims): S = -(x - mu_t)^T L_t (x - mu_t) + (x - mu_n)^T L_n (x - mu_n).  The two dense
// (B x 2 D1)(2 D1 x 2 D1) GEMMs of the reference collapse to one — S = x^T (L_n - L_t) x + x^T v + c —
// evaluated as four chained fp32-MFMA block GEMMs straight from the layer-1 accumulators (kernel:
// MODE_GB in nplda_fwd_kernel.h).  v and c are computed in fp64 when the image is packed.
#include <cstdlib>
#include "nplda_fwd_dispatch.h"
#include "nplda_gb_half.h"

namespace {

using namespace nplda;

struct GbLayout {
    int D0, D1, NB, KS1;
    size_t oW1, oG, ob1, ov, oc, total;
};

GbLayout gb_layout(int D0, int D1) {
    GbLayout L;
    L.D0 = D0; L.D1 = D1;
    L.NB = nplda_kernel_nb(D1, D1);
    L.KS1 = (D0 + 15) / 16;
    L.oW1 = 0;
    L.oG = (size_t)L.KS1 * L.NB * 256;
    L.ob1 = L.oG + 4 * (size_t)L.NB * L.NB * 256;
    L.ov = L.ob1 + (size_t)L.NB * 16;
    L.oc = L.ov + 2 * (size_t)L.NB * 16;
    L.total = L.oc + 4 + (size_t)2 * L.NB * 256;  // + one 2-step chunk of slack (unconditional chunk loads)
    return L;
}

// Lt != NULL: G = Ln - Lt.  Lt == NULL, dplda == 0: G = Ln (an explicit 2 D1 x 2 D1 form).  dplda == 1: Ln is DPlda's
// logistic_regres.weight [Wb | Ww | ws] (utils/models.py:466,484-490): diagonal blocks Ww, off-diagonal blocks Wb.
__global__ void gb_pack_kernel(const float* __restrict__ W1, const float* __restrict__ b1,
                               const float* __restrict__ Lt, const float* __restrict__ Ln, GbLayout L,
                               float* __restrict__ out, int dplda) {
    const size_t idx = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (idx >= L.ov) return;  // v and c are written by gb_vc_kernel
    float v = 0.f;
    const int D1 = L.D1, n2 = 2 * D1;
    if (idx < L.ob1) {
        const bool isG = idx >= L.oG;
        const size_t rel = isG ? idx - L.oG : idx;
        const int i = (int)(rel & 3);
        const int lane = (int)((rel >> 2) & 63);
        size_t blk = rel >> 8;
        const int nb = (int)(blk % L.NB);
        blk /= L.NB;
        const int f = 16 * nb + (lane & 15);
        if (!isG) {
            const int k = 16 * (int)blk + 4 * (lane >> 4) + i;
            if (f < D1 && k < L.D0) v = W1[(size_t)f * L.D0 + k];
        } else {
            const int kb = (int)(blk % L.NB);
            const int hh = (int)(blk / L.NB);  // 2 * h_out + h_in
            const int ho = hh >> 1, hi = hh & 1;
            const int k = 16 * kb + 4 * (lane >> 4) + i;
            if (f < D1 && k < D1) {
                const size_t r = (size_t)(ho * D1 + f), c = (size_t)(hi * D1 + k);
                if (dplda) v = Ln[(size_t)(ho == hi ? D1 * D1 : 0) + (size_t)f * D1 + k];
                else v = Lt ? Ln[r * n2 + c] - Lt[r * n2 + c] : Ln[r * n2 + c];
            }
        }
    } else {
        const int f = (int)(idx - L.ob1);
        if (f < D1) v = b1[f];
    }
    out[idx] = v;
}

// v = -(L_n + L_n^T) mu_n + (L_t + L_t^T) mu_t,  c = mu_n^T L_n mu_n - mu_t^T L_t mu_t   (fp64)
// One 64-lane block per (padded) entry of v; the per-row parts of c go to a scratch of doubles in the image's slack.
__global__ __launch_bounds__(64) void gb_v_kernel(const float* __restrict__ mut, const float* __restrict__ Lt,
                                                  const float* __restrict__ mun, const float* __restrict__ Ln,
                                                  GbLayout L, float* __restrict__ out) {
    const int D1 = L.D1, n2 = 2 * D1;
    const int i = blockIdx.x;
    const int h = i / (L.NB * 16), f = i % (L.NB * 16);
    double* cpart = reinterpret_cast<double*>(out + L.oc + 4);
    double acc = 0.0, wn = 0.0, wt = 0.0;
    const int r = h * D1 + f;
    if (f < D1) {
        for (int c = threadIdx.x; c < n2; c += 64) {
            const double ln = Ln[(size_t)r * n2 + c], lnT = Ln[(size_t)c * n2 + r];
            const double lt = Lt[(size_t)r * n2 + c], ltT = Lt[(size_t)c * n2 + r];
            acc += -(ln + lnT) * (double)mun[c] + (lt + ltT) * (double)mut[c];
            wn += ln * (double)mun[c];
            wt += lt * (double)mut[c];
        }
    }
#pragma unroll
    for (int m = 32; m >= 1; m >>= 1) {
        acc += __shfl_xor(acc, m, 64);
        wn += __shfl_xor(wn, m, 64);
        wt += __shfl_xor(wt, m, 64);
    }
    if (threadIdx.x == 0) {
        out[L.ov + i] = f < D1 ? (float)acc : 0.f;
        cpart[i] = f < D1 ? (double)mun[r] * wn - (double)mut[r] * wt : 0.0;
    }
}

__global__ __launch_bounds__(64) void gb_c_kernel(GbLayout L, float* __restrict__ out) {
    double* cpart = reinterpret_cast<double*>(out + L.oc + 4);
    const int n = 2 * L.NB * 16;
    double c = 0.0;
    for (int i = threadIdx.x; i < n; i += 64) c += cpart[i];
#pragma unroll
    for (int m = 32; m >= 1; m >>= 1) c += __shfl_xor(c, m, 64);
    __syncthreads();
    for (int i = threadIdx.x; i < n; i += 64) cpart[i] = 0.0;  // leave the slack as zeros
    if (threadIdx.x == 0) out[L.oc] = (float)c;
    if (threadIdx.x < 3) out[L.oc + 1 + threadIdx.x] = 0.f;
}

// explicit quadratic form S = x^T M x + x^T v + c: v (2 D1) and c are given
__global__ __launch_bounds__(512) void gb_vc_direct_kernel(const float* __restrict__ vin, float c, GbLayout L,
                                                           float* __restrict__ out) {
    for (int i = threadIdx.x; i < 2 * L.NB * 16; i += 512) {
        const int h = i / (L.NB * 16), f = i % (L.NB * 16);
        out[L.ov + i] = (f < L.D1 && vin) ? vin[h * L.D1 + f] : 0.f;
    }
    if (threadIdx.x == 0) {
        out[L.oc] = c;
        out[L.oc + 1] = out[L.oc + 2] = out[L.oc + 3] = 0.f;
    }
}

// DPlda: v = [ws; ws] with ws = the last D1 weights, c = the bias (read on the device: no host sync)
__global__ __launch_bounds__(512) void gb_vc_dplda_kernel(const float* __restrict__ wlr, const float* __restrict__ blr,
                                                          GbLayout L, float* __restrict__ out) {
    const float* ws = wlr + 2 * (size_t)L.D1 * L.D1;
    for (int i = threadIdx.x; i < 2 * L.NB * 16; i += 512) {
        const int f = i % (L.NB * 16);
        out[L.ov + i] = f < L.D1 ? ws[f] : 0.f;
    }
    if (threadIdx.x == 0) {
        out[L.oc] = blr[0];
        out[L.oc + 1] = 1.0f;  // block-symmetric image (G00 = G11, G01 = G10, v0 = v1): nplda_gb_half.h takes ONE pass
        out[L.oc + 2] = out[L.oc + 3] = 0.f;
    }
}

int gb_check(int D0, int D1) {
    if (D0 <= 0 || D1 <= 0 || (D0 % 4) != 0) return NPLDA_EINVAL;
    if (nplda_kernel_nb(D1, D1) == 0) return NPLDA_EUNSUPPORTED;
    return NPLDA_OK;
}

template <int WAVES, bool NT>
int launch_gb(FwdArgs a, const GbLayout& L, hipStream_t st) {
    const long long per_block = 16 * WAVES;
    const long long blocks = (a.n + per_block - 1) / per_block;
    if (blocks > 0x7fffffffLL) return NPLDA_EINVAL;
    dim3 grid((unsigned)blocks), block(WAVES * 64);
#define NPLDA_LAUNCH(NBV) hipLaunchKernelGGL((nplda_fwd_kernel<NBV, MODE_GB, WAVES, NT, 2>), grid, block, 0, st, a)
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

// batches of <= 16 384 pairs: the feature-split schedule (4 waves share a 16-pair tile), as for NeuralPlda
int launch_gb_small(FwdArgs a, const GbLayout& L, hipStream_t st) {
    // up to ONE 8-pair half tile per CU: nplda_gb_half.h (both sides of a pair in one MFMA row group: half the layer-1 MFMAs
    // per block, twice the blocks; NPLDA_GB_NO_HALF=1: the 16-pair tiles, A/B runs)
    static const bool no_half = getenv("NPLDA_GB_NO_HALF") != nullptr && getenv("NPLDA_GB_NO_HALF")[0] == '1';
    if (!no_half && a.n <= 8LL * mid_cus() && (L.NB == 10 || L.NB == 11 || L.NB == 2 || L.NB == 4 || L.NB == 8 || L.NB == 12)) {
        dim3 hgrid((unsigned)((a.n + 7) / 8)), hblock(256);
#define NPLDA_LAUNCH_H(NBV)                                                                                              \
    if (L.KS1 == 32 && a.D0 == 512) hipLaunchKernelGGL((nplda_gb_half_kernel<NBV, 32>), hgrid, hblock, 0, st, a);        \
    else hipLaunchKernelGGL((nplda_gb_half_kernel<NBV, 0>), hgrid, hblock, 0, st, a)
        switch (L.NB) {
            case 2: NPLDA_LAUNCH_H(2); break;
            case 4: NPLDA_LAUNCH_H(4); break;
            case 8: NPLDA_LAUNCH_H(8); break;
            case 10: NPLDA_LAUNCH_H(10); break;
            case 11: NPLDA_LAUNCH_H(11); break;
            default: NPLDA_LAUNCH_H(12); break;
        }
#undef NPLDA_LAUNCH_H
        return nplda_launch_status();
    }
    const long long blocks = (a.n + 15) / 16;
    dim3 grid((unsigned)blocks), block(256);
#define NPLDA_LAUNCH(NBV) hipLaunchKernelGGL((nplda_fwd_small_kernel<NBV, MODE_GB>), grid, block, 0, st, a)
    // 512-d x-vectors (KS1 = 32) at the recipe sizes: the fully unrolled K loop, as the NeuralPlda modes have had it since round 2
    // (nplda_fwd_small.h: a loop whose loads cross the back-edge gets an s_waitcnt vmcnt(0) at its head — the prefetch ring
    // drained every four k16-steps).  NPLDA_GB_NO_UNROLL=1: the rolled loop (A/B runs).
    static const bool no_unroll = getenv("NPLDA_GB_NO_UNROLL") != nullptr && getenv("NPLDA_GB_NO_UNROLL")[0] == '1';
#define NPLDA_LAUNCH32(NBV)                                                                                                \
    if (L.KS1 == 32 && a.D0 == 512 && !no_unroll) hipLaunchKernelGGL((nplda_fwd_small_kernel<NBV, MODE_GB, 32>), grid, block, 0, st, a); \
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

}  // namespace

extern "C" {

size_t gb_packed_bytes(int D0, int D1) {
    if (gb_check(D0, D1) != NPLDA_OK) return 0;
    return gb_layout(D0, D1).total * sizeof(float);
}

int gb_pack_params_f32(const float* W1, const float* b1, const float* mu_t, const float* Lam_t, const float* mu_n,
                       const float* Lam_n, int D0, int D1, void* packed, size_t packed_bytes, nplda_stream_t stream) {
    if (!W1 || !b1 || !mu_t || !Lam_t || !mu_n || !Lam_n || !packed) return NPLDA_EINVAL;
    if (int rc = gb_check(D0, D1)) return rc;
    const GbLayout L = gb_layout(D0, D1);
    if (packed_bytes < L.total * sizeof(float)) return NPLDA_ENOSPC;
    if (!nplda_aligned16(packed)) return NPLDA_EINVAL;
    hipStream_t st = (hipStream_t)stream;
    hipLaunchKernelGGL(gb_pack_kernel, dim3((unsigned)((L.ov + 255) / 256)), dim3(256), 0, st, W1, b1, Lam_t, Lam_n, L,
                       (float*)packed, 0);
    if (int rc = nplda_launch_status()) return rc;
    hipLaunchKernelGGL(gb_v_kernel, dim3(2 * L.NB * 16), dim3(64), 0, st, mu_t, Lam_t, mu_n, Lam_n, L, (float*)packed);
    if (int rc = nplda_launch_status()) return rc;
    hipLaunchKernelGGL(gb_c_kernel, dim3(1), dim3(64), 0, st, L, (float*)packed);
    return nplda_launch_status();
}

int gb_pack_quadform_f32(const float* W1, const float* b1, const float* M, const float* v, float c, int D0, int D1,
                          void* packed, size_t packed_bytes, nplda_stream_t stream) {
    if (!W1 || !b1 || !M || !packed) return NPLDA_EINVAL;
    if (int rc = gb_check(D0, D1)) return rc;
    const GbLayout L = gb_layout(D0, D1);
    if (packed_bytes < L.total * sizeof(float)) return NPLDA_ENOSPC;
    if (!nplda_aligned16(packed)) return NPLDA_EINVAL;
    hipStream_t st = (hipStream_t)stream;
    hipLaunchKernelGGL(gb_pack_kernel, dim3((unsigned)((L.ov + 255) / 256)), dim3(256), 0, st, W1, b1,
                       (const float*)nullptr, M, L, (float*)packed, 0);
    if (int rc = nplda_launch_status()) return rc;
    hipLaunchKernelGGL(gb_vc_direct_kernel, dim3(1), dim3(512), 0, st, v, c, L, (float*)packed);
    return nplda_launch_status();
}

int gb_pack_dplda_f32(const float* W1, const float* b1, const float* wlr, const float* blr, int D0, int D1,
                      void* packed, size_t packed_bytes, nplda_stream_t stream) {
    if (!W1 || !b1 || !wlr || !blr || !packed) return NPLDA_EINVAL;
    if (int rc = gb_check(D0, D1)) return rc;
    const GbLayout L = gb_layout(D0, D1);
    if (packed_bytes < L.total * sizeof(float)) return NPLDA_ENOSPC;
    if (!nplda_aligned16(packed)) return NPLDA_EINVAL;
    hipStream_t st = (hipStream_t)stream;
    hipLaunchKernelGGL(gb_pack_kernel, dim3((unsigned)((L.ov + 255) / 256)), dim3(256), 0, st, W1, b1,
                       (const float*)nullptr, wlr, L, (float*)packed, 1);
    if (int rc = nplda_launch_status()) return rc;
    hipLaunchKernelGGL(gb_vc_dplda_kernel, dim3(1), dim3(512), 0, st, wlr, blr, L, (float*)packed);
    return nplda_launch_status();
}

static int gb_score_impl(const float* x1, const float* x2, int64_t B, int64_t ldx, const void* packed, int D0, int D1,
                         float* s, float* paired, float* rn, int no_norm, nplda_stream_t stream);

int gb_score_pairs_f32(const float* x1, const float* x2, int64_t B, int64_t ldx, const void* packed, int D0, int D1,
                       float* s, float* paired, nplda_stream_t stream) {
    return gb_score_impl(x1, x2, B, ldx, packed, D0, D1, s, paired, nullptr, 0, stream);
}

int gb_score_pairs_ex_f32(const float* x1, const float* x2, int64_t B, int64_t ldx, const void* packed, int D0, int D1,
                          float* s, float* paired, float* rn, nplda_stream_t stream) {
    return gb_score_impl(x1, x2, B, ldx, packed, D0, D1, s, paired, rn, 0, stream);
}

int gb_score_rows_f32(const float* y1, const float* y2, int64_t B, int64_t ldy, const void* packed, int D0, int D1,
                      float* s, nplda_stream_t stream) {
    return gb_score_impl(y1, y2, B, ldy, packed, D0, D1, s, nullptr, nullptr, 1, stream);
}

static int gb_score_impl(const float* x1, const float* x2, int64_t B, int64_t ldx, const void* packed, int D0, int D1,
                         float* s, float* paired, float* rn, int no_norm, nplda_stream_t stream) {
    if (B < 0) return NPLDA_EINVAL;
    if (int rc = gb_check(D0, D1)) return rc;
    if (B == 0) return NPLDA_OK;
    if (!packed || (!s && !paired) || !nplda_aligned16(packed)) return NPLDA_EINVAL;
    if (!rows_ok(x1, ldx, D0) || !rows_ok(x2, ldx, D0)) return NPLDA_EINVAL;
    const GbLayout L = gb_layout(D0, D1);
    FwdArgs a = {};
    a.xa = x1; a.xb = x2; a.n = B; a.ldx = ldx; a.packed = (const float*)packed;
    a.D0 = D0; a.KS1 = L.KS1;
    a.oW2 = L.oG; a.ob1 = L.ob1; a.ob2 = L.ov; a.oQ = L.oc; a.oP = L.oc; a.total = L.total;
    a.out_s = s; a.out_z = paired; a.ldz = 2 * (long long)D1; a.out_rn = rn;
    a.no_norm = no_norm;
    if (B <= 256 * 64) return launch_gb_small(a, L, (hipStream_t)stream);
    return launch_gb<8, false>(a, L, (hipStream_t)stream);
}

}  // extern "C"

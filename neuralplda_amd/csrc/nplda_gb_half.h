// nplda_gb_half.h — GaussianBackend / DPlda pair scoring of SMALL batches on 8-pair HALF tiles (round 6).
//
// nplda_fwd_small_kernel<NB, MODE_GB> gives a block of four waves one 16-pair tile: the x1 rows and the x2 rows are two MFMA
// row groups, and a batch of 256 pairs (conf/voices_config_dplda.cfg:25-29) is 16 blocks on 256 CUs.  Here a block takes 8
// pairs and puts BOTH sides into ONE 16-row group — column j < 8 is the x1 row of pair j, column j >= 8 the x2 row of pair
// j - 8 (the arrangement of csrc/nplda_train_fb_half.h) — so layer 1 is half the MFMAs per block and twice the blocks:
//  * layer 1, F.normalize, the paired rows and 1 / norm exactly as in the 16-pair kernel (same k order per output element:
//    the same y bits); wave w owns the feature blocks {w, w + 4, w + 8};
//  * the quadratic form S = x^T G x + v.x + c on x = [y1; y2] needs, for column j, t_side(j) = v + G[side][0] y1 + G[side][1] y2
//    with the OTHER row of the pair as the second operand: it is read from the y tile in LDS at lane ^ 8.  For a
//    block-symmetric image (DPlda: G00 = G11 = Ww, G01 = G10 = Wb, v = [ws; ws]; flagged in the image, gb_vc_dplda_kernel)
//    ONE pass u = v + Ww y_own + Wb y_partner serves both sides — half the MFMAs again; a general GaussianBackend image
//    takes both passes over all 16 columns and keeps each column's own side.
// Scores equal the 16-pair kernel's to rounding (side 2 accumulates its two G blocks in the other order; the four waves'
// partial sums meet per side).  Used up to ONE half tile per CU (launch_gb_small).
#pragma once
#include "nplda_fwd_kernel.h"

namespace nplda {

template <int NB, int KS1C = 0>
__global__ __launch_bounds__(256, 2) void nplda_gb_half_kernel(const FwdArgs a) {
    constexpr int NW = 4;
    constexpr int NBW = (NB + NW - 1) / NW;
    constexpr int PF = 4, PF1 = PF + 1;
    static_assert(NBW <= 3, "one weight load per MFMA quarter, the x load after the last");
    __shared__ f32x4 ylds[NB][64];   // normalised layer-1 output, accumulator layout (= the B operand of the quadratic form)
    __shared__ float red[NW][16];    // cross-wave partials (norms, then scores)

    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int j = lane & 15;
    const int g = lane >> 4;
    const int side = j >> 3;
    long long pair = (long long)blockIdx.x * 8 + (j & 7);
    const bool ok = pair < a.n;
    if (!ok) pair = a.n - 1;
    const float* srow = (side ? a.xb : a.xa) + pair * a.ldx;

    const f32x4* W1p = reinterpret_cast<const f32x4*>(a.packed);
    const f32x4* Gp = reinterpret_cast<const f32x4*>(a.packed + a.oW2);   // G[2 ho + hi][kb][nb] fragments
    const f32x4* b1p = reinterpret_cast<const f32x4*>(a.packed + a.ob1);
    const f32x4* vp = reinterpret_cast<const f32x4*>(a.packed + a.ob2);    // v, two padded halves
    const int KS1 = KS1C ? KS1C : a.KS1;
    const int D0 = a.D0;
    const __amdgpu_buffer_rsrc_t prsrc = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(a.packed), 0, 0x7fffffff, 0x00020000);
    const unsigned lane16 = (unsigned)lane * 16u;
    auto fragb = [&](const f32x4* base, size_t fragidx) {  // (see nplda_fwd_small.h: a buffer load, no vector address arithmetic)
        typedef unsigned u32x4 __attribute__((ext_vector_type(4)));
        const int soff = __builtin_amdgcn_readfirstlane(
            (int)(reinterpret_cast<const char*>(base + fragidx * 64) - reinterpret_cast<const char*>(a.packed)));
        const u32x4 v = __builtin_amdgcn_raw_buffer_load_b128(prsrc, lane16, soff, 0);
        return __builtin_bit_cast(f32x4, v);
    };

    // ---- layer 1 ---------------------------------------------------------------------------------------------------------
    f32x4 acc[NBW];
#pragma unroll
    for (int i = 0; i < NBW; ++i) {
        const int nb = wave + NW * i;
        acc[i] = nb < NB ? b1p[4 * nb + g] : f32x4{0.f, 0.f, 0.f, 0.f};
    }
    f32x4 wf[PF1][NBW], xs[PF1];
    auto fetchw = [&](int slot, int ks, int i) {
        const int ksc = ks < KS1 ? ks : KS1 - 1;
        const int nb = wave + NW * i;
        wf[slot][i] = fragb(W1p, (size_t)ksc * NB + (nb < NB ? nb : NB - 1));
    };
    auto fetchx = [&](int slot, int ks) { xs[slot] = load_x4c<false>(srow, 16 * ks + 4 * g, D0); };
#pragma unroll
    for (int s = 0; s < PF; ++s) {
#pragma unroll
        for (int i = 0; i < NBW; ++i) fetchw(s, s, i);
        fetchx(s, s);
    }
    auto step = [&](int ks, int slot, int rs) {
#pragma unroll
        for (int r = 0; r < 4; ++r) {
#pragma unroll
            for (int i = 0; i < NBW; ++i)
                acc[i] = __builtin_amdgcn_mfma_f32_16x16x4f32(wf[slot][i][r], xs[slot][r], acc[i], 0, 0, 0);
            if (r < NBW) fetchw(rs, ks + PF, r);
            if (r == 3) fetchx(rs, ks + PF);
            __builtin_amdgcn_sched_barrier(0);
        }
    };
    if constexpr (KS1C > 0) {
#pragma unroll
        for (int ks = 0; ks < KS1C; ++ks) step(ks, ks % PF1, (ks + PF) % PF1);
    } else {
        for (int ks0 = 0; ks0 < KS1; ks0 += PF1) {
#pragma unroll
            for (int s = 0; s < PF1; ++s)
                if (ks0 + s < KS1) step(ks0 + s, s, (s + PF) % PF1);
        }
    }

    // ---- F.normalize: partial sums of squares over this wave's features -> LDS -> all waves ---------------------------------
    {
        float ss = 0.f;
#pragma unroll
        for (int i = 0; i < NBW; ++i) {
            if (wave + NW * i < NB) {
#pragma unroll
                for (int r = 0; r < 4; ++r) ss = fmaf(acc[i][r], acc[i][r], ss);
            }
        }
        ss = wave_xor_add(ss, 16);
        ss = wave_xor_add(ss, 32);
        if (g == 0) red[wave][j] = ss;
    }
    __syncthreads();
    float inv = 1.0f / fmaxf(sqrtf(((red[0][j] + red[1][j]) + red[2][j]) + red[3][j]), 1e-12f);
    if (a.no_norm) inv = 1.0f;
#pragma unroll
    for (int i = 0; i < NBW; ++i) {
        const int nb = wave + NW * i;
        if (nb < NB) {
            acc[i] *= inv;
            ylds[nb][lane] = acc[i];
        }
    }
    if (a.out_rn != nullptr && wave == 0 && g == 0 && ok) a.out_rn[side ? a.n + pair : pair] = inv;
    if (a.out_z != nullptr) {  // forward_getpaired: (n, 2 D1) rows [y1 | y2]
        const int D1 = (int)a.ldz / 2;
#pragma unroll
        for (int i = 0; i < NBW; ++i) {
            const int nb = wave + NW * i;
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const int f = 16 * nb + 4 * g + r;
                if (nb < NB && ok && f < D1) a.out_z[pair * a.ldz + side * D1 + f] = acc[i][r];
            }
        }
    }
    if (a.out_s == nullptr) return;

    // ---- quadratic form ------------------------------------------------------------------------------------------------------
    const bool sym = a.packed[a.oQ + 1] != 0.f;  // block-symmetric image (DPlda): uniform
    auto fetchg = [&](int slot, int q) {  // q = (2 ho + hi) * NB + kb
        const int qc = q < 4 * NB ? q : 4 * NB - 1;
#pragma unroll
        for (int i = 0; i < NBW; ++i) {
            const int nb = wave + NW * i;
            wf[slot][i] = fragb(Gp, (size_t)qc * NB + (nb < NB ? nb : NB - 1));
        }
    };
#pragma unroll
    for (int s = 0; s < PF; ++s) fetchg(s, s);
    __syncthreads();  // ylds complete (and every wave has read the norms out of `red`)
    const int lane_own = lane, lane_par = lane ^ 8;                        // the pair's other row
    const int lane_s0 = (lane & ~15) | (j & 7), lane_s1 = lane_s0 | 8;     // the pair's x1 / x2 row
    float part = 0.f;
    if (sym) {
        f32x4 t[NBW];
#pragma unroll
        for (int i = 0; i < NBW; ++i) {
            const int nb = wave + NW * i;
            t[i] = nb < NB ? vp[nb * 4 + g] : f32x4{0.f, 0.f, 0.f, 0.f};
        }
#pragma unroll
        for (int hk = 0; hk < 2 * NB; ++hk) {  // G[0][0] (= G[1][1]) on the own rows, then G[0][1] (= G[1][0]) on the partners
            const int s = hk % PF;
            const f32x4 yv = ylds[hk % NB][hk < NB ? lane_own : lane_par];
#pragma unroll
            for (int r = 0; r < 4; ++r) {
#pragma unroll
                for (int i = 0; i < NBW; ++i) t[i] = __builtin_amdgcn_mfma_f32_16x16x4f32(wf[s][i][r], yv[r], t[i], 0, 0, 0);
            }
            fetchg(s, hk + PF < 2 * NB ? hk + PF : 4 * NB - 1);
            __builtin_amdgcn_sched_barrier(0);
        }
#pragma unroll
        for (int i = 0; i < NBW; ++i) {
            if (wave + NW * i < NB) {
#pragma unroll
                for (int r = 0; r < 4; ++r) part = fmaf(acc[i][r], t[i][r], part);
            }
        }
    } else {
#pragma unroll
        for (int ho = 0; ho < 2; ++ho) {
            f32x4 t[NBW];
#pragma unroll
            for (int i = 0; i < NBW; ++i) {
                const int nb = wave + NW * i;
                t[i] = nb < NB ? vp[(ho * NB + nb) * 4 + g] : f32x4{0.f, 0.f, 0.f, 0.f};
            }
#pragma unroll
            for (int hk = 0; hk < 2 * NB; ++hk) {
                const int q = ho * 2 * NB + hk;
                const int s = q % PF;
                const f32x4 yv = ylds[hk % NB][hk < NB ? lane_s0 : lane_s1];
#pragma unroll
                for (int r = 0; r < 4; ++r) {
#pragma unroll
                    for (int i = 0; i < NBW; ++i) t[i] = __builtin_amdgcn_mfma_f32_16x16x4f32(wf[s][i][r], yv[r], t[i], 0, 0, 0);
                }
                fetchg(s, q + PF);
                __builtin_amdgcn_sched_barrier(0);
            }
            float p = 0.f;
#pragma unroll
            for (int i = 0; i < NBW; ++i) {
                if (wave + NW * i < NB) {
#pragma unroll
                    for (int r = 0; r < 4; ++r) p = fmaf(acc[i][r], t[i][r], p);
                }
            }
            part += side == ho ? p : 0.f;  // a column keeps the pass of its own side
        }
    }
    part = wave_xor_add(part, 16);
    part = wave_xor_add(part, 32);
    if (g == 0) red[wave][j] = part;
    __syncthreads();
    if (wave == 0 && g == 0 && j < 8 && ok)
        a.out_s[pair] = ((((red[0][j] + red[1][j]) + red[2][j]) + red[3][j]) +
                         (((red[0][j + 8] + red[1][j + 8]) + red[2][j + 8]) + red[3][j + 8])) + a.packed[a.oQ];
}

}  // namespace nplda

// nplda_fwd_small.h — small-batch schedule of the fused forward (training minibatches, B <= 16 384 pairs).
//
// The streaming kernels give one wave a whole 16-pair tile: 3 360 serial MFMAs = 45 us, and a 4096-pair
// minibatch (BASELINE cfg2) is only 256 tiles — a quarter of the chip's 1024 SIMDs, 100 us per forward.
// Here a tile is shared by the 4 waves of a block, split over OUTPUT FEATURES: wave w owns the 16-wide
// feature blocks {w, w+4, w+8, ...} of both layers.
//  * layer 1: every wave reads the tile's x fragments (4x redundant, L2-served) and only ITS weight fragments,
//    straight from the packed image in L2 into registers with a 4-step register ring — no LDS, no barrier in
//    the K loop;
//  * the row norm needs all features: per-wave partial sums of squares meet in LDS (one barrier);
//  * the normalised y blocks are published to LDS in accumulator layout, which is exactly the B-operand layout
//    layer 2 consumes (see nplda_fwd_kernel.h), one f32x4 per lane per block: conflict-free;
//  * layer 2: wave w computes ITS z blocks from all of y (LDS) and its W2 fragments (registers, ring);
//  * the score is a sum over features: per-wave partials meet in LDS, wave 0 writes s.
// Same arithmetic per element as the streaming kernels except the order of the cross-feature sums.
#pragma once
#include "nplda_fwd_kernel.h"
#include "nplda_l1_ksplit.h"

namespace nplda {

#ifdef NPLDA_SMALL_STAMPS  // tools/exp_fwd.hip only: 100 MHz time stamps of one wave at the phase boundaries
__device__ unsigned long long g_small_stamps[16];
#define NPLDA_STAMP(i) do { if (blockIdx.x == gridDim.x / 2 && threadIdx.x == 0) { g_small_stamps[i] = __builtin_amdgcn_s_memrealtime(); g_small_stamps[8 + i] = __builtin_readcyclecounter(); } } while (0)
#else
#define NPLDA_STAMP(i) do {} while (0)
#endif

// KS1C: layer 1's k16-step count when known at compile time (32 = the 512-d x-vectors of every reference recipe), 0 = read
// it from the arguments.  With a constant count the K loop is fully unrolled: hipcc places an s_waitcnt vmcnt(0) at the
// head of a loop whose loads are carried across the back-edge (it gives up merging the two predecessors' counters), which
// drains the whole prefetch ring once per round — ~1 us of exposed latency every 4 k16-steps.
template <int NB, int MODE, int KS1C = 0>
__global__ __launch_bounds__(256, 2) void nplda_fwd_small_kernel(const FwdArgs a) {
    NPLDA_STAMP(0);
    static_assert(MODE == MODE_PAIR || MODE == MODE_EMBED || MODE == MODE_TRAIN || MODE == MODE_GB, "small kernel modes");
    constexpr int NW = 4;
    constexpr int NBW = (NB + NW - 1) / NW;  // feature-block slots per wave
    constexpr int PF = 4;                    // register-ring depth (k16-steps)
    // Pair scoring at NB = 10: blocks 0 .. 7 go two to a wave and the two left-over blocks are split by SIDE — wave w takes
    // block 8 + w / 2 for the x1 rows (w even) or the x2 rows (w odd): 5 MFMAs per k4-step on every wave instead of 6, 6, 4,
    // 4.  The slot's accumulators live in the "A" arrays whatever its side; the block's two z halves meet through LDS for
    // the score.  (train_fb_small_kernel and bwd_data_small_kernel use the same assignment: the same partial sums, the
    // same bits.)
    constexpr bool HALF = NB == 10 && (MODE == MODE_PAIR || MODE == MODE_TRAIN);
    constexpr int NBF = HALF ? NB / NW : NBW;  // slots holding a whole block (both sides)
    constexpr int HS = NBF;                    // the half slot (HALF only)
    // 512-d pairs at the recipe sizes: layer 1 split over the waves by K (nplda_l1_ksplit.h), its LDS exchange region reused
    // for the y tiles
    // (round 6: also the GaussianBackend / DPlda mode at NB = 11 — the reference's shipped D1 = 170; its layer 1 is the same GEMM
    // on the same image layout.  At NB = 10 the K-split hands back the two left-over blocks split by SIDE, which only the pair
    // modes' HALF slot takes)
    constexpr bool KSPLIT = KS1C == 32 && (((NB == 10 || NB == 11) && (MODE == MODE_PAIR || MODE == MODE_TRAIN)) ||
                                           (NB == 11 && MODE == MODE_GB));
    __shared__ f32x4 lbuf[KSPLIT ? l1k_lds_f4(NB) : 2 * NB * 64];
    f32x4 (*ylds)[NB][64] = reinterpret_cast<f32x4 (*)[NB][64]>(lbuf);  // normalised layer-1 output, accumulator layout
    __shared__ float red[NW][2][16];         // cross-wave partials (norms, then scores)
    __shared__ f32x4 zx[HALF ? NW : 1][64];  // z of the half slots, for the wave holding the block's other side

    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);  // provably wave-uniform: scalar branches only
    const int j = lane & 15;
    const int g = lane >> 4;
    const int hb = NW * NBF + (wave >> 1);   // HALF: block of the half slot,
    const bool hside = (wave & 1) != 0;      //       and its side (false: x1 rows)
    auto blk = [&](int i) { return HALF && i == HS ? hb : wave + NW * i; };

    long long t0A, t0B;
    if (MODE == MODE_EMBED) {
        t0A = (long long)blockIdx.x * 32;
        t0B = t0A + 16;
    } else {
        t0A = (long long)blockIdx.x * 16;
        t0B = t0A;
    }
    long long rowA = t0A + j, rowB = t0B + j;
    const bool okA = rowA < a.n, okB = rowB < a.n;
    if (!okA) rowA = a.n - 1;
    if (!okB) rowB = a.n - 1;
    const float* sa = a.xa + rowA * a.ldx;
    const float* sb = a.xb + rowB * a.ldx;
    const float* pa = sa + 4 * g;
    const float* pb = sb + 4 * g;

    const f32x4* W1p = reinterpret_cast<const f32x4*>(a.packed);
    const f32x4* W2p = reinterpret_cast<const f32x4*>(a.packed + a.oW2);
    const f32x4* b1p = reinterpret_cast<const f32x4*>(a.packed + a.ob1);
    const f32x4* b2p = reinterpret_cast<const f32x4*>(a.packed + a.ob2);
    const f32x4* Qp = reinterpret_cast<const f32x4*>(a.packed + a.oQ);
    const f32x4* Pp = reinterpret_cast<const f32x4*>(a.packed + a.oP);
    const int KS1 = KS1C ? KS1C : a.KS1;
    const int D0 = a.D0;
    // a 64-lane weight fragment as a BUFFER load: descriptor of the packed image in SGPRs, the fragment's byte offset as the
    // scalar offset, the lane's 16 bytes as the vector offset — no vector instruction per load.  (As a global load each one
    // took a 64-bit vector address: v_lshl_add_u64 + v_add_co + v_addc, 273 of the 677 VALU instructions between this
    // kernel's 840 MFMAs at NB = 10, and a VALU instruction there costs the matrix pipe 6 - 13 cycles: tools/exp_issue_cost.hip)
    const __amdgpu_buffer_rsrc_t prsrc = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(a.packed), 0, 0x7fffffff, 0x00020000);
    const unsigned lane16 = (unsigned)lane * 16u;
    auto fragb = [&](const f32x4* base, size_t fragidx) {
        typedef unsigned u32x4 __attribute__((ext_vector_type(4)));
        const int soff = __builtin_amdgcn_readfirstlane(
            (int)(reinterpret_cast<const char*>(base + fragidx * 64) - reinterpret_cast<const char*>(a.packed)));
        const u32x4 v = __builtin_amdgcn_raw_buffer_load_b128(prsrc, lane16, soff, 0);
        return __builtin_bit_cast(f32x4, v);
    };

    // ---- layer 1 --------------------------------------------------------------------------------------------
    f32x4 accA[NBW], accB[NBW];
#pragma unroll
    for (int i = 0; i < NBW; ++i) {
        const int nb = blk(i);
        accA[i] = nb < NB ? b1p[4 * nb + g] : f32x4{0.f, 0.f, 0.f, 0.f};
        accB[i] = accA[i];
    }
    // Register ring of PF + 1 k16-steps: during step ks the slot that step ks - 1 released is refilled with step ks + PF,
    // one load after each quarter of the step's MFMAs.  With one wave per SIMD nothing else fills the matrix pipe while
    // this wave computes addresses: a step's 5 loads and their ~25 scalar / vector address instructions issued in one
    // block after its 24 MFMAs cost 0.45 us per step against 0.32 us of MFMA time; spread through the step they issue in
    // the shadow of the MFMAs.  sched_barrier pins the pieces: left free, the scheduler sinks every load to its first use.
    constexpr int PF1 = PF + 1;
    f32x4 wf[PF1][NBW], xa[PF1], xb[PF1];
    // every load is unconditional (indices clamped): predicated loads would make hipcc wait vmcnt(0) per step
    auto fetchw = [&](int slot, int ks, int i) {
        const int ksc = ks < KS1 ? ks : KS1 - 1;
        const int nb = blk(i);
        wf[slot][i] = fragb(W1p, (size_t)ksc * NB + (nb < NB ? nb : NB - 1));
    };
    auto fetchx = [&](int slot, int ks) {
        xa[slot] = load_x4c<false>(sa, 16 * ks + 4 * g, D0);
        xb[slot] = load_x4c<false>(sb, 16 * ks + 4 * g, D0);
    };
    if constexpr (!KSPLIT) {
#pragma unroll
        for (int s = 0; s < PF; ++s) {
#pragma unroll
            for (int i = 0; i < NBW; ++i) fetchw(s, s, i);
            fetchx(s, s);
        }
    }
    auto step = [&](int ks, int slot, int rs) {
        f32x4 xh;  // the half slot's side: a select (4 VALU) — a third x load per step cost more (every load instruction of
        if constexpr (HALF) xh = hside ? xb[slot] : xa[slot];  // a one-wave-per-SIMD kernel idles the matrix pipe ~49 cycles)
#pragma unroll
        for (int r = 0; r < 4; ++r) {
#pragma unroll
            for (int i = 0; i < NBF; ++i) {
                // no guard: a wave whose i-th block does not exist (nb >= NB) multiplies the clamped
                // fragment into an accumulator that is never read — cheaper than a branch per MFMA
                accA[i] = __builtin_amdgcn_mfma_f32_16x16x4f32(wf[slot][i][r], xa[slot][r], accA[i], 0, 0, 0);
                accB[i] = __builtin_amdgcn_mfma_f32_16x16x4f32(wf[slot][i][r], xb[slot][r], accB[i], 0, 0, 0);
            }
            if constexpr (HALF) accA[HS] = __builtin_amdgcn_mfma_f32_16x16x4f32(wf[slot][HS][r], xh[r], accA[HS], 0, 0, 0);
            if (r < NBW) fetchw(rs, ks + PF, r);
            if (r == 3) fetchx(rs, ks + PF);
            __builtin_amdgcn_sched_barrier(0);
        }
    };
    static_assert(NBW <= 3, "one weight load per MFMA quarter, the x loads after the last");
    NPLDA_STAMP(1);
    if constexpr (KSPLIT) {
        f32x4 uF[2][2], uL[1][2];
        const int swz = l1_ksplit_tile<NB, false, false>(a.packed, a.total, sa, sb, nullptr, nullptr, false, b1p, wave, lane, lbuf,
                                                         uF, uL);
        // back to the feature split the rest of the kernel is written for: blocks w, w + 4 (both sides), the left-over slot
#pragma unroll
        for (int i = 0; i < 2; ++i) {
            accA[i] = swz ? uF[i][1] : uF[i][0];
            accB[i] = swz ? uF[i][0] : uF[i][1];
        }
        if constexpr (HALF) {
            accA[HS] = uL[0][0];  // (block 8 + w / 2, side w & 1 = rho 0)
        } else {
            accA[2] = uL[0][0];
            accB[2] = uL[0][1];
        }
    } else if constexpr (KS1C > 0) {
#pragma unroll
        for (int ks = 0; ks < KS1C; ++ks) {
            if (ks == 8) NPLDA_STAMP(2);
            step(ks, ks % PF1, (ks + PF) % PF1);
        }
    } else {
        for (int ks0 = 0; ks0 < KS1; ks0 += PF1) {
#pragma unroll
            for (int s = 0; s < PF1; ++s)
                if (ks0 + s < KS1) step(ks0 + s, s, (s + PF) % PF1);
        }
    }

    NPLDA_STAMP(3);
    // ---- F.normalize: partial sums of squares over this wave's features -> LDS -> all waves -----------------
    {
        float ssA = 0.f, ssB = 0.f;
#pragma unroll
        for (int i = 0; i < NBF; ++i) {
            if (wave + NW * i < NB) {
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    ssA = fmaf(accA[i][r], accA[i][r], ssA);
                    ssB = fmaf(accB[i][r], accB[i][r], ssB);
                }
            }
        }
        if constexpr (HALF) {
            float sh = hside ? ssB : ssA;
#pragma unroll
            for (int r = 0; r < 4; ++r) sh = fmaf(accA[HS][r], accA[HS][r], sh);
            if (hside) ssB = sh;
            else ssA = sh;
        }
        ssA = wave_xor_add(ssA, 16); ssA = wave_xor_add(ssA, 32);
        ssB = wave_xor_add(ssB, 16); ssB = wave_xor_add(ssB, 32);
        if (g == 0) {
            red[wave][0][j] = ssA;
            red[wave][1][j] = ssB;
        }
    }
    __syncthreads();
    float invA = 1.0f / fmaxf(sqrtf(((red[0][0][j] + red[1][0][j]) + red[2][0][j]) + red[3][0][j]), 1e-12f);
    float invB = 1.0f / fmaxf(sqrtf(((red[0][1][j] + red[1][1][j]) + red[2][1][j]) + red[3][1][j]), 1e-12f);
    if (MODE == MODE_GB && a.no_norm) invA = invB = 1.0f;
#pragma unroll
    for (int i = 0; i < NBF; ++i) {
        const int nb = wave + NW * i;
        if (nb < NB) {
            accA[i] *= invA;
            accB[i] *= invB;
            ylds[0][nb][lane] = accA[i];
            ylds[1][nb][lane] = accB[i];
            if (MODE == MODE_TRAIN) {
                if (okA) *reinterpret_cast<f32x4*>(a.out_y + rowA * a.ldz + 16 * nb + 4 * g) = accA[i];
                if (okB) *reinterpret_cast<f32x4*>(a.out_y + (a.n + rowB) * a.ldz + 16 * nb + 4 * g) = accB[i];
            }
        }
    }
    const long long rowH = hside ? a.n + rowB : rowA;  // HALF (pair / train modes): the half slot's row of the (2n)-row outputs
    if constexpr (HALF) {
        accA[HS] *= hside ? invB : invA;
        ylds[hside ? 1 : 0][hb][lane] = accA[HS];
        if (MODE == MODE_TRAIN && okA) *reinterpret_cast<f32x4*>(a.out_y + rowH * a.ldz + 16 * hb + 4 * g) = accA[HS];
    }
    if (MODE == MODE_TRAIN && wave == 0 && g == 0 && okA) {
        a.out_rn[rowA] = invA;
        a.out_rn[a.n + rowB] = invB;
    }
    if (MODE == MODE_GB && a.out_rn != nullptr && wave == 0 && g == 0 && okA) {  // DPlda / GB rows for an LDA backward
        a.out_rn[rowA] = invA;
        a.out_rn[a.n + rowB] = invB;
    }
    if (MODE == MODE_EMBED && a.out_y != nullptr) {  // embedding rows saved for nplda_embed_backward_f32
#pragma unroll
        for (int i = 0; i < NBW; ++i) {
            const int nb = wave + NW * i;
            if (nb < NB) {
                if (okA) *reinterpret_cast<f32x4*>(a.out_y + rowA * a.ldz + 16 * nb + 4 * g) = accA[i];
                if (okB) *reinterpret_cast<f32x4*>(a.out_y + rowB * a.ldz + 16 * nb + 4 * g) = accB[i];
            }
        }
        if (wave == 0 && g == 0) {
            if (okA) a.out_rn[rowA] = invA;
            if (okB) a.out_rn[rowB] = invB;
        }
    }

    if (MODE == MODE_GB) {
        // ---- quadratic form on x = [y1; y2] (GaussianBackend.forward / DPlda.forward; image: nplda_gb.hip) --------
        // t_ho = v_ho + sum_hi G[ho][hi] y_hi for this wave's feature blocks, S = y1.t_0 + y2.t_1 + c.  Every G
        // fragment is used by exactly one MFMA quartet, so it goes L2 -> register ring like the layer-1 weights.
        if (a.out_z != nullptr) {  // forward_getpaired: (n, 2 D1) rows [y1 | y2]
            const int D1 = (int)a.ldz / 2;
#pragma unroll
            for (int i = 0; i < NBW; ++i) {
                const int nb = wave + NW * i;
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    const int f = 16 * nb + 4 * g + r;
                    if (nb < NB && okA && f < D1) {
                        a.out_z[rowA * a.ldz + f] = accA[i][r];
                        a.out_z[rowA * a.ldz + D1 + f] = accB[i][r];
                    }
                }
            }
        }
        if (a.out_s == nullptr) return;
        const f32x4* Gp = W2p;   // a.oW2 -> G[2 ho + hi][kb][nb] fragments
        const f32x4* vp = b2p;   // a.ob2 -> v, two padded halves
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
        __syncthreads();  // ylds complete
        float part = 0.f;
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
                const f32x4 yv = ylds[hk / NB][hk % NB][lane];
#pragma unroll
                for (int r = 0; r < 4; ++r) {
#pragma unroll
                    for (int i = 0; i < NBW; ++i)
                        t[i] = __builtin_amdgcn_mfma_f32_16x16x4f32(wf[s][i][r], yv[r], t[i], 0, 0, 0);
                }
                fetchg(s, q + PF);
                __builtin_amdgcn_sched_barrier(0);
            }
#pragma unroll
            for (int i = 0; i < NBW; ++i) {
                if (wave + NW * i < NB) {
#pragma unroll
                    for (int r = 0; r < 4; ++r) part = fmaf(ho == 0 ? accA[i][r] : accB[i][r], t[i][r], part);
                }
            }
        }
        part = wave_xor_add(part, 16);
        part = wave_xor_add(part, 32);
        if (g == 0) red[wave][0][j] = part;
        __syncthreads();
        if (wave == 0 && g == 0 && okA)
            a.out_s[t0A + j] = (((red[0][0][j] + red[1][0][j]) + red[2][0][j]) + red[3][0][j]) + a.packed[a.oQ];
        return;
    }

    // ---- layer 2 ------------------------------------------------------------------------------------------------
    f32x4 zA[NBW], zB[NBW];
#pragma unroll
    for (int i = 0; i < NBW; ++i) {
        const int nb = blk(i);
        zA[i] = nb < NB ? b2p[4 * nb + g] : f32x4{0.f, 0.f, 0.f, 0.f};
        zB[i] = zA[i];
    }
    auto fetch2 = [&](int slot, int kb) {
        const int kbc = kb < NB ? kb : NB - 1;
#pragma unroll
        for (int i = 0; i < NBW; ++i) {
            const int nb = blk(i);
            wf[slot][i] = fragb(W2p, (size_t)kbc * NB + (nb < NB ? nb : NB - 1));
        }
    };
#pragma unroll
    for (int s = 0; s < PF; ++s) fetch2(s, s);
    NPLDA_STAMP(4);
    __syncthreads();  // ylds complete (also orders the `red` reuse below after every wave's norm reads)
    NPLDA_STAMP(5);
#pragma unroll
    for (int kb = 0; kb < NB; ++kb) {
        const int s = kb % PF;
        const f32x4 yA = ylds[0][kb][lane], yB = ylds[1][kb][lane];
        f32x4 yH;
        if constexpr (HALF) yH = ylds[hside ? 1 : 0][kb][lane];
#pragma unroll
        for (int r = 0; r < 4; ++r) {
#pragma unroll
            for (int i = 0; i < NBF; ++i) {
                zA[i] = __builtin_amdgcn_mfma_f32_16x16x4f32(wf[s][i][r], yA[r], zA[i], 0, 0, 0);
                zB[i] = __builtin_amdgcn_mfma_f32_16x16x4f32(wf[s][i][r], yB[r], zB[i], 0, 0, 0);
            }
            if constexpr (HALF) zA[HS] = __builtin_amdgcn_mfma_f32_16x16x4f32(wf[s][HS][r], yH[r], zA[HS], 0, 0, 0);
        }
        fetch2(s, kb + PF);
        __builtin_amdgcn_sched_barrier(0);
    }

    NPLDA_STAMP(6);
    // ---- epilogue ---------------------------------------------------------------------------------------------------
    if (MODE == MODE_PAIR || MODE == MODE_TRAIN) {
        f32x4 zP;  // HALF: z of the half block's other side
        if constexpr (HALF) {
            zx[wave][lane] = zA[HS];
            __syncthreads();
            zP = zx[wave ^ 1][lane];
        }
        float part = 0.f;
#pragma unroll
        for (int i = 0; i < NBF; ++i) {
            const int nb = wave + NW * i;
            if (nb < NB) {
                const f32x4 q = Qp[4 * nb + g];
                const f32x4 p = Pp[4 * nb + g];
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    const float z1 = zA[i][r], z2 = zB[i][r];
                    part = fmaf(q[r], fmaf(z1, z1, z2 * z2), part);
                    part = fmaf(2.0f * p[r], z1 * z2, part);
                }
                if (MODE == MODE_TRAIN) {
                    if (okA) *reinterpret_cast<f32x4*>(a.out_z + rowA * a.ldz + 16 * nb + 4 * g) = zA[i];
                    if (okB) *reinterpret_cast<f32x4*>(a.out_z + (a.n + rowB) * a.ldz + 16 * nb + 4 * g) = zB[i];
                }
            }
        }
        if constexpr (HALF) {
            if (!hside) {  // the block's term, once: by the wave of its x1 side
                const f32x4 q = Qp[4 * hb + g];
                const f32x4 p = Pp[4 * hb + g];
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    const float z1 = zA[HS][r], z2 = zP[r];
                    part = fmaf(q[r], fmaf(z1, z1, z2 * z2), part);
                    part = fmaf(2.0f * p[r], z1 * z2, part);
                }
            }
            if (MODE == MODE_TRAIN && okA) *reinterpret_cast<f32x4*>(a.out_z + rowH * a.ldz + 16 * hb + 4 * g) = zA[HS];
        }
        part = wave_xor_add(part, 16);
        part = wave_xor_add(part, 32);
        if (g == 0) red[wave][0][j] = part;
        __syncthreads();
        if (wave == 0 && g == 0 && okA)
            a.out_s[t0A + j] = ((red[0][0][j] + red[1][0][j]) + red[2][0][j]) + red[3][0][j];
        NPLDA_STAMP(7);
    } else {
        float qa = 0.f, qb = 0.f;
#pragma unroll
        for (int i = 0; i < NBW; ++i) {
            const int nb = wave + NW * i;
            if (nb < NB) {
                const f32x4 q = Qp[4 * nb + g];
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    qa = fmaf(q[r] * zA[i][r], zA[i][r], qa);
                    qb = fmaf(q[r] * zB[i][r], zB[i][r], qb);
                }
                if (okA) *reinterpret_cast<f32x4*>(a.out_z + rowA * a.ldz + 16 * nb + 4 * g) = zA[i];
                if (okB) *reinterpret_cast<f32x4*>(a.out_z + rowB * a.ldz + 16 * nb + 4 * g) = zB[i];
            }
        }
        qa = wave_xor_add(qa, 16); qa = wave_xor_add(qa, 32);
        qb = wave_xor_add(qb, 16); qb = wave_xor_add(qb, 32);
        if (g == 0) {
            red[wave][0][j] = qa;
            red[wave][1][j] = qb;
        }
        __syncthreads();
        if (a.out_q != nullptr && wave == 0 && g == 0) {
            if (okA) a.out_q[rowA] = ((red[0][0][j] + red[1][0][j]) + red[2][0][j]) + red[3][0][j];
            if (okB) a.out_q[rowB] = ((red[0][1][j] + red[1][1][j]) + red[2][1][j]) + red[3][1][j];
        }
    }
}

}  // namespace nplda

// nplda_train_fb_small.h — forward AND data gradients of a training minibatch (<= 16 384 pairs) in one kernel.
//
// The fused training step (nplda_train_step_f32) ran nplda_fwd_small_kernel<MODE_TRAIN> and then
// bwd_data_small_kernel<LOSS>: the same 16-pair tile on the same 4 waves with the same feature split, the second
// launch re-reading what the first had just written (z, y, 1 / ||u||, s) after ~10 us of launch and first-load latency.
// Here the block simply goes on: after the score it forms dL/ds of its pairs (the batch counts come from the targets,
// summed by every block for itself), dz from the z blocks still in registers, dy = dz W2 as the chained MFMA with the
// W2^T fragments, and the normalize backward from the y blocks still in registers.  Written for the weight-gradient
// kernel: y, dz, du (+ the per-block dQ / dP sums and loss partials); z, 1 / ||u|| and g never leave the chip.
// Same arithmetic, instruction for instruction, as the two kernels it replaces (tests compare the parameter bits).
#pragma once
#include "nplda_bwd_loss.h"
#include "nplda_fwd_kernel.h"
#include "nplda_l1_ksplit.h"

namespace nplda {

#ifdef NPLDA_FB_STAMPS  // tools/exp_fb.hip only: 100 MHz time stamps of one wave at the phase boundaries
__device__ unsigned long long g_fb_stamps[32];
#define NPLDA_FB_STAMP(i) do { if (blockIdx.x == gridDim.x / 2 && threadIdx.x == NPLDA_FB_STAMPS) { g_fb_stamps[i] = __builtin_amdgcn_s_memrealtime(); g_fb_stamps[16 + i] = __builtin_readcyclecounter(); } } while (0)
#else
#define NPLDA_FB_STAMP(i) do {} while (0)
#endif

struct TrainFbArgs {
    const float* xa;      // (n, ldx) x1 rows
    const float* xb;      // (n, ldx) x2 rows
    long long n, ldx;
    const float* packed;  // NpldaLayout image
    int D0, KS1;
    size_t oW2, oW2T, ob1, ob2, oQ, oP;
    float* out_s;         // (n) scores (optional: diagnostics / callers that want them)
    float* out_y;         // (2n, ldz) normalised layer-1 outputs, x1 rows then x2 rows
    float* dz;            // (2n, ldz)
    float* du;            // (2n, ldz)
    long long ldz;
    float* pq;            // [blocks][2][ldz] per-block sums of g (z1^2 + z2^2) and g z1 z2
    BwdLoss ls;           // targets, thresholds, loss partials (ls.s and ls.g_out unused)
    // Indexed form (the training loop's: pairs named by rows of a resident x-vector table, xa == xb == the table):
    // pair i reads rows ia[i], ib[i] (clamped into [0, ntab): the loaders have checked them), and wave 0 leaves the rows it
    // fetched in xsa / xsb (n, ldxs) for the weight-gradient kernel — the gather the step otherwise runs as two launches.
    const long long* ia;
    const long long* ib;
    long long ntab;
    float* xsa;           // (null: nothing is staged — the weight-gradient kernel reads the batch's bf16 rows itself)
    float* xsb;
    long long ldxs;
    // DX form (the head's step of an end-to-end fine-tune, nplda_train_step_dx_f32): the block goes on to dL/dx = du . W1 of
    // its own 32 rows — the chain of dx_small_kernel (csrc/nplda_matmul.hip) on the du it has just formed, instead of a launch
    // of its own that reads du back
    size_t oW1T;          // W1^T fragment image inside `packed`
    void* dx0;            // (n, lddx) dL/dx1: fp32, or bfloat16 when DXBF
    void* dx1;            // (n, lddx) dL/dx2
    long long lddx;
    float* step_bump;     // optional: Adam's step counter, incremented here (one thread) for the update kernel of this step
    long long* rec_bump;  // optional (nplda_train_step_records_f32): rec_bump[0] += 1 — the epoch's record counter, read by
                          // the update kernel of this step, which stages the next record (a launch later: no race)
};

// ROWS: the indexed form (pairs named by table rows — or, ia == nullptr, the batch's own rows — with the x rows staged in
// fp32 for K-B); XBF: the x rows are bfloat16 (ldx in elements): what a jointly trained extractor hands over (cfg 5) —
// widened in registers (one shift / mask per value), staged in fp32 like gathered rows
// DX: 0 = no input gradients, 1 = dL/dx in fp32, 2 = in bfloat16 (round to nearest even) — 512-d x-vectors, NB >= 8
template <int NB, int KS1C, bool ROWS, bool XBF = false, int DX = 0>
__global__ __launch_bounds__(256, 2) void train_fb_small_kernel(const TrainFbArgs a) {
    static_assert(!XBF || (ROWS && KS1C > 0), "bf16 rows: the staged 512-d form only");
    static_assert(DX == 0 || (KS1C == 32 && NB >= 8), "dL/dx in this kernel: 512-d x-vectors");
    constexpr int NW = 4;
    constexpr int NBW = (NB + NW - 1) / NW;  // feature-block slots per wave
    // NB = 10: blocks 0 .. 7 go two to a wave, and the two left-over blocks are split by SIDE — wave w takes block
    // 8 + w / 2 for the x1 rows (w even) or the x2 rows (w odd) — so that every wave issues 5 MFMAs per k4-step instead of
    // 6, 6, 4, 4.  The slot's accumulators live in the "A" arrays whatever its side.  A block's two sides then sit in two
    // waves: z crosses once through LDS (score, dz and the pair sums need both).  The per-wave partial sums of the three
    // cross-wave reductions (||u||^2, score, y . dy) follow this assignment, so at NB = 10 the last bits differ from the
    // separate forward / backward kernels, which keep whole blocks per wave.
    constexpr bool HALF = NB == 10;
    constexpr int NBF = HALF ? NB / NW : NBW;  // slots holding a whole block (both sides)
    constexpr int HS = NBF;                    // the half slot (HALF only)
    constexpr int PF = 4, PF1 = PF + 1;
    static_assert(NBW <= 3, "one weight load per MFMA quarter, the x loads after the last");
    // 512-d pairs at the recipe sizes: layer 1 split over the waves by K (nplda_l1_ksplit.h: the same function, hence the same
    // bits, as nplda_fwd_small_kernel<MODE_TRAIN>), its LDS exchange region reused for the y / dz tiles
    constexpr bool KSPLIT = KS1C == 32 && (NB == 10 || NB == 11);
    __shared__ f32x4 lbuf[KSPLIT ? l1k_lds_f4(NB) : 2 * NB * 64];
    f32x4 (*ylds)[NB][64] = reinterpret_cast<f32x4 (*)[NB][64]>(lbuf);  // y for layer 2 (accumulator layout), then dz for the dy chain
    __shared__ float red[NW][2][16];
    __shared__ float cnt_s[NW];
    __shared__ double lacc[16][kLossNS];
    __shared__ float lcs[nplda_loss::kMaxK + 1];
    __shared__ f32x4 zx[HALF ? NW : 1][64];  // z of the half slots, for the wave holding the block's other side

    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int j = lane & 15;
    const int g = lane >> 4;
    const int hb = NW * NBF + (wave >> 1);   // HALF: block of the half slot,
    const bool hside = (wave & 1) != 0;      //       and its side (false: x1 rows)
    auto blk = [&](int i) { return HALF && i == HS ? hb : wave + NW * i; };
    // one 64-lane weight fragment at a wave-uniform address: uniform base + 32-bit lane offset (the SGPR-base form of the
    // load; the offset is made opaque at every use, or hipcc folds it into one 64-bit vector add per load)
    // (round 3: as a BUFFER load — descriptor of the packed image in SGPRs, the fragment's byte offset as the scalar offset,
    // the lane's 16 bytes as the vector offset: no vector instruction per load at all.  The global-load form cost four
    // (v_mov, v_lshl_add_u64, v_add_co, v_addc for the 64-bit address), and every VALU instruction between the MFMAs of a
    // wave alone on its SIMD costs the matrix pipe 6 - 13 cycles: tools/exp_issue_cost.hip)
    const __amdgpu_buffer_rsrc_t prsrc = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(a.packed), 0, 0x7fffffff, 0x00020000);
    const unsigned lane16 = (unsigned)lane * 16u;
    auto frag = [&](const f32x4* base) {
        typedef unsigned u32x4 __attribute__((ext_vector_type(4)));
        const int soff = __builtin_amdgcn_readfirstlane((int)(reinterpret_cast<const char*>(base) - reinterpret_cast<const char*>(a.packed)));
        const u32x4 v = __builtin_amdgcn_raw_buffer_load_b128(prsrc, lane16, soff, 0);
        return __builtin_bit_cast(f32x4, v);
    };

    const long long t0 = (long long)blockIdx.x * 16;
    const bool ok = t0 + j < a.n;
    const long long rA = ok ? t0 + j : a.n - 1;  // x1-side row of y / dz / du; the x2 side sits n rows further
    const long long rB = a.n + rA;
    long long xrA = rA, xrB = rA;
    const BwdLoss& ls = a.ls;
    if constexpr (ROWS) {
        if (a.ia != nullptr) {
            xrA = a.ia[rA];
            xrB = a.ib[rA];
            xrA = xrA < 0 ? 0 : (xrA < a.ntab ? xrA : a.ntab - 1);
            xrB = xrB < 0 ? 0 : (xrB < a.ntab ? xrB : a.ntab - 1);
        }
    }
    const float* sa = XBF ? nullptr : a.xa + xrA * a.ldx;
    const float* sb = XBF ? nullptr : a.xb + xrB * a.ldx;
    const unsigned short* sah4 = reinterpret_cast<const unsigned short*>(a.xa) + xrA * a.ldx + 4 * (lane >> 4);
    const unsigned short* sbh4 = reinterpret_cast<const unsigned short*>(a.xb) + xrB * a.ldx + 4 * (lane >> 4);

    const f32x4* W1p = reinterpret_cast<const f32x4*>(a.packed);
    const f32x4* W2p = reinterpret_cast<const f32x4*>(a.packed + a.oW2);
    const f32x4* W2T = reinterpret_cast<const f32x4*>(a.packed + a.oW2T);
    const f32x4* b1p = reinterpret_cast<const f32x4*>(a.packed + a.ob1);
    const f32x4* b2p = reinterpret_cast<const f32x4*>(a.packed + a.ob2);
    const f32x4* Qp = reinterpret_cast<const f32x4*>(a.packed + a.oQ);
    const f32x4* Pp = reinterpret_cast<const f32x4*>(a.packed + a.oP);
    const int KS1 = KS1C ? KS1C : a.KS1;
    const int D0 = a.D0;
    if (a.step_bump != nullptr && blockIdx.x == 0 && tid == 0) a.step_bump[0] += 1.0f;
    if (a.rec_bump != nullptr && blockIdx.x == 0 && tid == 0) a.rec_bump[0] += 1;
    NPLDA_FB_STAMP(0);

    // ---- layer 1 (nplda_fwd_small.h) -----------------------------------------------------------------------------
    f32x4 accA[NBW], accB[NBW];
#pragma unroll
    for (int i = 0; i < NBW; ++i) {
        const int nb = blk(i);
        accA[i] = nb < NB ? b1p[4 * nb + g] : f32x4{0.f, 0.f, 0.f, 0.f};
        accB[i] = accA[i];
    }
    f32x4 wf[PF1][NBW], xa[PF1], xb[PF1];
    auto fetchw = [&](int slot, int ks, int i) {
        const int ksc = ks < KS1 ? ks : KS1 - 1;
        const int nb = blk(i);
        wf[slot][i] = frag(W1p + ((size_t)ksc * NB + (nb < NB ? nb : NB - 1)) * 64);
    };
    const float* sa4 = sa + 4 * g;
    const float* sb4 = sb + 4 * g;
    auto fetchx = [&](int slot, int ks) {
        if constexpr (XBF) {
            typedef unsigned u32x2 __attribute__((ext_vector_type(2)));
            const int kc = ks < KS1C ? ks : KS1C - 1;
            const u32x2 ra = *reinterpret_cast<const u32x2*>(sah4 + 16 * kc);
            const u32x2 rb = *reinterpret_cast<const u32x2*>(sbh4 + 16 * kc);
            xa[slot] = f32x4{__uint_as_float(ra[0] << 16), __uint_as_float(ra[0] & 0xffff0000u),
                             __uint_as_float(ra[1] << 16), __uint_as_float(ra[1] & 0xffff0000u)};
            xb[slot] = f32x4{__uint_as_float(rb[0] << 16), __uint_as_float(rb[0] & 0xffff0000u),
                             __uint_as_float(rb[1] << 16), __uint_as_float(rb[1] & 0xffff0000u)};
        } else if constexpr (KS1C > 0) {  // D0 = 16 KS1C: whole k16-steps only, the column offset is an immediate of the load
            const int kc = ks < KS1C ? ks : KS1C - 1;
            xa[slot] = *reinterpret_cast<const f32x4*>(sa4 + 16 * kc);
            xb[slot] = *reinterpret_cast<const f32x4*>(sb4 + 16 * kc);
        } else {
            xa[slot] = load_x4c<false>(sa, 16 * ks + 4 * g, D0);
            xb[slot] = load_x4c<false>(sb, 16 * ks + 4 * g, D0);
        }
    };
    if constexpr (!KSPLIT) {
#pragma unroll
        for (int s = 0; s < PF; ++s) {
#pragma unroll
            for (int i = 0; i < NBW; ++i) fetchw(s, s, i);
            fetchx(s, s);
        }
    }
    NPLDA_FB_STAMP(1);
    // the batch's targets: loaded now, behind the first fragments, counted after layer 1 (no wait, no barrier of their own)
    // (K-split layer 1: issued behind its last MFMAs instead, in front of the exchange barrier — the loop needs the registers)
    TargetEarly te;
    float ti;
    PairLossConsts lc;
    auto early_loads = [&]() {
        if (ls.B >= 4) target_count_issue(ls, te);
        ti = ls.t[rA];
        loss_consts_theta(ls, lc);
    };
    if constexpr (!KSPLIT) early_loads();
    __builtin_amdgcn_sched_barrier(0);
    NPLDA_FB_STAMP(2);

    auto step = [&](int ks, int slot, int rs) {
        if constexpr (ROWS) {  // every wave leaves a quarter of the k16-steps of the fetched rows (D0 % 16 == 0 here)
            if ((ks & (NW - 1)) == wave && ok && a.xsa != nullptr) {
                *reinterpret_cast<f32x4*>(a.xsa + rA * a.ldxs + 16 * ks + 4 * g) = xa[slot];
                *reinterpret_cast<f32x4*>(a.xsb + rA * a.ldxs + 16 * ks + 4 * g) = xb[slot];
            }
        }
        f32x4 xh;  // the half slot's side: a select (4 VALU) — a third x load per step cost more (every load instruction of
        if constexpr (HALF) xh = hside ? xb[slot] : xa[slot];  // a one-wave-per-SIMD kernel idles the matrix pipe ~49 cycles)
#pragma unroll
        for (int r = 0; r < 4; ++r) {
#pragma unroll
            for (int i = 0; i < NBF; ++i) {
                accA[i] = __builtin_amdgcn_mfma_f32_16x16x4f32(wf[slot][i][r], xa[slot][r], accA[i], 0, 0, 0);
                accB[i] = __builtin_amdgcn_mfma_f32_16x16x4f32(wf[slot][i][r], xb[slot][r], accB[i], 0, 0, 0);
            }
            if constexpr (HALF) accA[HS] = __builtin_amdgcn_mfma_f32_16x16x4f32(wf[slot][HS][r], xh[r], accA[HS], 0, 0, 0);
            if (r < NBW) fetchw(rs, ks + PF, r);
            if (r == 3) fetchx(rs, ks + PF);
            __builtin_amdgcn_sched_barrier(0);
        }
    };
    if constexpr (KSPLIT) {
        f32x4 uF[2][2], uL[1][2];
        const int swz = l1_ksplit_tile<NB, XBF, ROWS>(
            a.packed, a.oP + 16 * NB, XBF ? reinterpret_cast<const float*>(sah4 - 4 * (lane >> 4)) : sa,
            XBF ? reinterpret_cast<const float*>(sbh4 - 4 * (lane >> 4)) : sb,
            ROWS && a.xsa != nullptr ? a.xsa + rA * a.ldxs : nullptr, ROWS && a.xsa != nullptr ? a.xsb + rA * a.ldxs : nullptr,
            ok, b1p, wave, lane, lbuf, uF, uL, early_loads);
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
        for (int ks = 0; ks < KS1C; ++ks) step(ks, ks % PF1, (ks + PF) % PF1);
    } else {
        for (int ks0 = 0; ks0 < KS1; ks0 += PF1) {
#pragma unroll
            for (int s = 0; s < PF1; ++s)
                if (ks0 + s < KS1) step(ks0 + s, s, (s + PF) % PF1);
        }
    }

    NPLDA_FB_STAMP(3);
    // ---- F.normalize ------------------------------------------------------------------------------------------------
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
        if (ls.B < 4) {
#pragma unroll
            for (int q = 0; q < 4; ++q) te.v[q] = f32x4{0.f, 0.f, 0.f, 0.f};
        }
        const float cw = target_count_wave(ls, te);
        if (lane == 0) cnt_s[wave] = cw;
    }
    __syncthreads();
    const double Ntl = (double)((cnt_s[0] + cnt_s[1]) + (cnt_s[2] + cnt_s[3]));  // as block_target_count
    const double Nt = ls.gcount ? ls.gcount[0] : Ntl;                            // data parallel: the global batch's counts
    const double Nn = ls.gcount ? ls.gcount[1] : (double)ls.B - Ntl;
    const float invA = 1.0f / fmaxf(sqrtf(((red[0][0][j] + red[1][0][j]) + red[2][0][j]) + red[3][0][j]), 1e-12f);
    const float invB = 1.0f / fmaxf(sqrtf(((red[0][1][j] + red[1][1][j]) + red[2][1][j]) + red[3][1][j]), 1e-12f);
    const long long rH = hside ? rB : rA;  // HALF: the row of the half slot's side
#pragma unroll
    for (int i = 0; i < NBF; ++i) {
        const int nb = wave + NW * i;
        if (nb < NB) {
            accA[i] *= invA;
            accB[i] *= invB;
            ylds[0][nb][lane] = accA[i];
            ylds[1][nb][lane] = accB[i];
            if (ok) {
                *reinterpret_cast<f32x4*>(a.out_y + rA * a.ldz + 16 * nb + 4 * g) = accA[i];
                *reinterpret_cast<f32x4*>(a.out_y + rB * a.ldz + 16 * nb + 4 * g) = accB[i];
            }
        }
    }
    if constexpr (HALF) {
        accA[HS] *= hside ? invB : invA;
        ylds[hside ? 1 : 0][hb][lane] = accA[HS];
        if (ok) *reinterpret_cast<f32x4*>(a.out_y + rH * a.ldz + 16 * hb + 4 * g) = accA[HS];
    }

    // ---- layer 2 ------------------------------------------------------------------------------------------------------
    f32x4 zA[NBW], zB[NBW];
#pragma unroll
    for (int i = 0; i < NBW; ++i) {
        const int nb = blk(i);
        zA[i] = nb < NB ? b2p[4 * nb + g] : f32x4{0.f, 0.f, 0.f, 0.f};
        zB[i] = zA[i];
    }
    f32x4 w2[PF][NBW];
    auto fetch2 = [&](const f32x4* Wp, int slot, int kb) {
        const int kbc = kb < NB ? kb : NB - 1;
#pragma unroll
        for (int i = 0; i < NBW; ++i) {
            const int nb = blk(i);
            w2[slot][i] = frag(Wp + ((size_t)kbc * NB + (nb < NB ? nb : NB - 1)) * 64);
        }
    };
#pragma unroll
    for (int s = 0; s < PF; ++s) fetch2(W2p, s, s);
    __syncthreads();  // ylds complete (also orders the `red` reuse below after every wave's norm reads)
    // the batch constants of dL/ds (fp64 divisions): by the last wave, which has the fewest feature blocks at NB = 10 / 11
    // and would otherwise wait for the others at the score barrier; they reach the other waves through LDS there
    if (wave == NW - 1) {
        loss_consts_counts(ls, Nt, Nn, lc);
        if (lane == 0) {
#pragma unroll
            for (int k = 0; k < nplda_loss::kMaxK; ++k) lcs[k] = lc.cn[k];
            lcs[nplda_loss::kMaxK] = lc.ct;
        }
    }
    NPLDA_FB_STAMP(4);
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
                zA[i] = __builtin_amdgcn_mfma_f32_16x16x4f32(w2[s][i][r], yA[r], zA[i], 0, 0, 0);
                zB[i] = __builtin_amdgcn_mfma_f32_16x16x4f32(w2[s][i][r], yB[r], zB[i], 0, 0, 0);
            }
            if constexpr (HALF) zA[HS] = __builtin_amdgcn_mfma_f32_16x16x4f32(w2[s][HS][r], yH[r], zA[HS], 0, 0, 0);
        }
        fetch2(W2p, s, kb + PF);
        __builtin_amdgcn_sched_barrier(0);
    }

    NPLDA_FB_STAMP(5);
    // ---- score ----------------------------------------------------------------------------------------------------------
    f32x4 zP;  // HALF: z of the half block's other side
    if constexpr (HALF) {
        zx[wave][lane] = zA[HS];
        __syncthreads();
        zP = zx[wave ^ 1][lane];
    }
    {
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
        }
        part = wave_xor_add(part, 16);
        part = wave_xor_add(part, 32);
        if (g == 0) red[wave][0][j] = part;
    }
    // W2^T fragments of the dy chain: on their way during the exchanges below
#pragma unroll
    for (int s = 0; s < PF; ++s) fetch2(W2T, s, s);
    __syncthreads();  // scores of the tile; every wave is past layer 2: ylds is free for dz
    NPLDA_FB_STAMP(6);
    const float si = ((red[0][0][j] + red[1][0][j]) + red[2][0][j]) + red[3][0][j];
    if (a.out_s != nullptr && wave == 0 && g == 0 && ok) a.out_s[t0 + j] = si;

    // ---- loss: dL/ds of the tile's pairs, their terms of the loss sums (nplda_bwd_loss.h) ------------------------------
    double lsum[kLossNS];
#pragma unroll
    for (int k = 0; k < nplda_loss::kMaxK; ++k) lc.cn[k] = lcs[k];
    lc.ct = lcs[nplda_loss::kMaxK];
    const float gi = loss_pair(ls, lc, si, ti, lsum);
    const float tg = ok ? 2.0f * gi : 0.f;
    if (wave == 0 && g == 0) {
#pragma unroll
        for (int i = 0; i < kLossNS; ++i) lacc[j][i] = ok ? lsum[i] : 0.0;
    }

    // ---- dz = 2 g (Q z + P z'), the pair sums for dQ / dP (K-A of nplda_backward.hip) ------------------------------------
#pragma unroll
    for (int i = 0; i < NBF; ++i) {
        const int nb = wave + NW * i;
        if (nb < NB) {
            const f32x4 q = Qp[4 * nb + g], p = Pp[4 * nb + g];
            const f32x4 dA = dz_of(tg, q, p, zA[i], zB[i]);
            const f32x4 dB = dz_of(tg, q, p, zB[i], zA[i]);
            ylds[0][nb][lane] = dA;
            ylds[1][nb][lane] = dB;
            if (ok) {
                *reinterpret_cast<f32x4*>(a.dz + rA * a.ldz + 16 * nb + 4 * g) = dA;
                *reinterpret_cast<f32x4*>(a.dz + rB * a.ldz + 16 * nb + 4 * g) = dB;
            }
            const float gh = 0.5f * tg;
            f32x4 eq, ep;
            pair_sum_terms(gh, zA[i], zB[i], eq, ep);
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                eq[r] = row16_sum(eq[r]);
                ep[r] = row16_sum(ep[r]);
            }
            if (j == 0) {
                float* o = a.pq + (size_t)blockIdx.x * 2 * a.ldz + 16 * nb + 4 * g;
                *reinterpret_cast<f32x4*>(o) = eq;
                *reinterpret_cast<f32x4*>(o + a.ldz) = ep;
            }
        }
    }
    if constexpr (HALF) {
        const f32x4 q = Qp[4 * hb + g], p = Pp[4 * hb + g];
        const f32x4 dH = dz_of(tg, q, p, zA[HS], zP);  // own side's z, the other side's z
        ylds[hside ? 1 : 0][hb][lane] = dH;
        if (ok) *reinterpret_cast<f32x4*>(a.dz + rH * a.ldz + 16 * hb + 4 * g) = dH;
        if (!hside) {
            f32x4 eq, ep;
            pair_sum_terms(0.5f * tg, zA[HS], zP, eq, ep);
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                eq[r] = row16_sum(eq[r]);
                ep[r] = row16_sum(ep[r]);
            }
            if (j == 0) {
                float* o = a.pq + (size_t)blockIdx.x * 2 * a.ldz + 16 * hb + 4 * g;
                *reinterpret_cast<f32x4*>(o) = eq;
                *reinterpret_cast<f32x4*>(o + a.ldz) = ep;
            }
        }
    }
    __syncthreads();  // dz of the tile in LDS, the loss terms of its pairs
    NPLDA_FB_STAMP(7);
    if (tid < kLossNS) {
        double v = 0.0;
#pragma unroll
        for (int p = 0; p < 16; ++p) v += lacc[p][tid];
        ls.partial[(size_t)blockIdx.x * kLossNS + tid] = v;
    }

    // ---- dy = dz W2 (chained MFMA: A = W2^T fragments, B = dz from LDS) ---------------------------------------------------
    f32x4 dyA[NBW], dyB[NBW];
#pragma unroll
    for (int i = 0; i < NBW; ++i) {
        dyA[i] = f32x4{0.f, 0.f, 0.f, 0.f};
        dyB[i] = dyA[i];
    }
#pragma unroll
    for (int kb = 0; kb < NB; ++kb) {
        const int s = kb % PF;
        const f32x4 dA = ylds[0][kb][lane], dB = ylds[1][kb][lane];
        f32x4 dH;
        if constexpr (HALF) dH = ylds[hside ? 1 : 0][kb][lane];
#pragma unroll
        for (int r = 0; r < 4; ++r) {
#pragma unroll
            for (int i = 0; i < NBF; ++i) {
                dyA[i] = __builtin_amdgcn_mfma_f32_16x16x4f32(w2[s][i][r], dA[r], dyA[i], 0, 0, 0);
                dyB[i] = __builtin_amdgcn_mfma_f32_16x16x4f32(w2[s][i][r], dB[r], dyB[i], 0, 0, 0);
            }
            if constexpr (HALF) dyA[HS] = __builtin_amdgcn_mfma_f32_16x16x4f32(w2[s][HS][r], dH[r], dyA[HS], 0, 0, 0);
        }
        fetch2(W2T, s, kb + PF);
        __builtin_amdgcn_sched_barrier(0);
    }

    NPLDA_FB_STAMP(8);
    // ---- F.normalize backward: du = (dy - y (y . dy)) / max(||u||, eps); y is still in accA / accB ------------------------
    float dotA = 0.f, dotB = 0.f;
#pragma unroll
    for (int i = 0; i < NBF; ++i) {
        if (wave + NW * i < NB) {
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                dotA = fmaf(accA[i][r], dyA[i][r], dotA);
                dotB = fmaf(accB[i][r], dyB[i][r], dotB);
            }
        }
    }
    if constexpr (HALF) {
        float dh = hside ? dotB : dotA;
#pragma unroll
        for (int r = 0; r < 4; ++r) dh = fmaf(accA[HS][r], dyA[HS][r], dh);
        if (hside) dotB = dh;
        else dotA = dh;
    }
    dotA = wave_xor_add(dotA, 16); dotA = wave_xor_add(dotA, 32);
    dotB = wave_xor_add(dotB, 16); dotB = wave_xor_add(dotB, 32);
    if (g == 0) {
        red[wave][0][j] = dotA;
        red[wave][1][j] = dotB;
    }
    __syncthreads();
    dotA = ((red[0][0][j] + red[1][0][j]) + red[2][0][j]) + red[3][0][j];
    dotB = ((red[0][1][j] + red[1][1][j]) + red[2][1][j]) + red[3][1][j];
    if (invA >= 1e12f) dotA = 0.f;  // the clamp branch of F.normalize: u / eps, no projection term
    if (invB >= 1e12f) dotB = 0.f;
#pragma unroll
    for (int i = 0; i < NBF; ++i) {
        const int nb = wave + NW * i;
        if (nb < NB) {
            const f32x4 uA = du_of(dyA[i], accA[i], dotA, invA), uB = du_of(dyB[i], accB[i], dotB, invB);
            if (ok) {
                *reinterpret_cast<f32x4*>(a.du + rA * a.ldz + 16 * nb + 4 * g) = uA;
                *reinterpret_cast<f32x4*>(a.du + rB * a.ldz + 16 * nb + 4 * g) = uB;
            }
            if constexpr (DX != 0) {  // (every wave is past the dy chain: its dz tiles are free)
                ylds[0][nb][lane] = uA;
                ylds[1][nb][lane] = uB;
            }
        }
    }
    if constexpr (HALF) {
        const f32x4 uH = du_of(dyA[HS], accA[HS], hside ? dotB : dotA, hside ? invB : invA);
        if (ok) *reinterpret_cast<f32x4*>(a.du + rH * a.ldz + 16 * hb + 4 * g) = uH;
        if constexpr (DX != 0) ylds[hside ? 1 : 0][hb][lane] = uH;
    }
    NPLDA_FB_STAMP(9);
    if constexpr (DX != 0) {
        // ---- dL/dx = du . W1 of the tile's 32 rows: the chain of dx_small_kernel, instruction for instruction (same bits) —
        // wave w forms output column blocks 8 w .. 8 w + 7 of both sides from all of du (LDS) and the W1^T fragments (L2) ----
        constexpr int XBW = 8, PFX = 2, PFX1 = PFX + 1;
        const __amdgpu_buffer_rsrc_t ximg = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(a.packed + a.oW1T), 0,
                                                                              NB * 32 * 1024, 0x00020000);
        unsigned xvoff[XBW];
#pragma unroll
        for (int u = 0; u < XBW; ++u) xvoff[u] = (unsigned)(((XBW * wave + u) * 64 + lane) * 16);
        f32x4 xw[PFX1][XBW];
        auto fetchxw = [&](int slot, int kb) {
            const int kbc = kb < NB ? kb : NB - 1;
            const int soff = kbc * (32 * 1024);
#pragma unroll
            for (int u = 0; u < XBW; ++u)
                xw[slot][u] = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(ximg, (int)xvoff[u], soff, 0));
        };
#pragma unroll
        for (int p = 0; p < PFX; ++p) fetchxw(p, p);
        __syncthreads();  // du of the tile in LDS
        f32x4 xacc[2][XBW];
        const f32x4 zero4 = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int kb = 0; kb < NB; ++kb) {
            const int sl = kb % PFX1;
            const f32x4 d0 = ylds[0][kb][lane], d1 = ylds[1][kb][lane];
#pragma unroll
            for (int r = 0; r < 4; ++r) {
#pragma unroll
                for (int u = 0; u < XBW; ++u) {
                    xacc[0][u] = __builtin_amdgcn_mfma_f32_16x16x4f32(xw[sl][u][r], d0[r], (kb == 0 && r == 0) ? zero4 : xacc[0][u], 0, 0, 0);
                    xacc[1][u] = __builtin_amdgcn_mfma_f32_16x16x4f32(xw[sl][u][r], d1[r], (kb == 0 && r == 0) ? zero4 : xacc[1][u], 0, 0, 0);
                }
                if (r == 0 && kb + PFX < NB) fetchxw((kb + PFX) % PFX1, kb + PFX);
                __builtin_amdgcn_sched_barrier(0);
            }
        }
        if (ok) {
#pragma unroll
            for (int rg = 0; rg < 2; ++rg) {
#pragma unroll
                for (int u = 0; u < XBW; ++u) {
                    const int col = 16 * (XBW * wave + u) + 4 * g;
                    const f32x4 v = xacc[rg][u];
                    if constexpr (DX == 2) {  // round to nearest even, as torch's .to(bfloat16)
                        unsigned short* dst = reinterpret_cast<unsigned short*>(rg ? a.dx1 : a.dx0) + (t0 + j) * a.lddx + col;
                        unsigned w[4];
#pragma unroll
                        for (int c = 0; c < 4; ++c) {
                            const unsigned b = __float_as_uint(v[c]);
                            w[c] = (b & 0x7fffffffu) > 0x7f800000u ? ((b >> 16) | 0x40u) : ((b + 0x7fffu + ((b >> 16) & 1u)) >> 16);
                        }
                        typedef unsigned u32x2 __attribute__((ext_vector_type(2)));
                        *reinterpret_cast<u32x2*>(dst) = u32x2{w[0] | (w[1] << 16), w[2] | (w[3] << 16)};
                    } else {
                        float* dst = reinterpret_cast<float*>(rg ? a.dx1 : a.dx0) + (t0 + j) * a.lddx + col;
                        *reinterpret_cast<f32x4*>(dst) = v;
                    }
                }
            }
        }
    }
}

}  // namespace nplda

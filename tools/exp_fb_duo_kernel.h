// exp_fb_duo_kernel.h — EXPERIMENT (part of no build; tools/exp_fbh.hip runs it): forward, loss and data gradients of a 16-pair
// tile by EIGHT waves — layer 1 as in the 16-pair kernel, everything behind it as two 8-pair half tiles side by side.
// Correct (y / dz / du / s within 1e-7 of train_fb_small_kernel) and NOT faster: 26.7 us against 26.5 us at D = 150, 27.7 against
// 28.3 at D = 170 (profiles/r05h_exp_duo.txt).  The two waves of a SIMD run the same phase between the same barriers: their
// serial stretches coincide (exchange -> y 1.4 us, loss + dz 2.1 us — the 16-pair kernel's times, not the lone half tile's
// 1.0 / 1.65) and their MFMA phases queue behind each other (layer 2 ends at 15.0 us for one wave, 16.6 us for the other).
// With the halves synchronising SEPARATELY behind the exchange (NPLDA_DUO_SPLIT: a barrier among four waves by an LDS counter
// and a poll instead of s_barrier — the form tools/exp_fbh.hip builds) they do drift apart, half 0 ends at 20.8 us — and half
// 1 still at 23.2 us, the launch still takes 26.9 us: the late half needs 12 us for what a lone half tile does in 8.
//
// What round 5 measured on the way to this form (profiles/r05b_exp_fbh.txt, design/k06_backward_and_train_step.md "Round 5"):
// a lone 8-pair half tile (nplda_train_fb_half.h: a pair's x1 and x2 rows in ONE 16-row MFMA group, cross terms by DPP) spends
// 4.2 us in the serial stretches behind layer 1 — row norms, score, loss, dz, du — where the 16-pair tile, whose waves carry
// two row groups each, spends 8; but two half-tile BLOCKS per CU stream W1 twice and layer 1 is paced by that stream.  Here
// the block is one 16-pair tile again — W1 crosses into the CU once — and only the work behind layer 1 is cut in two:
//  * waves 0 .. 3 run layer 1 exactly like the 16-pair kernel (K-split: wave w takes the k16-steps {8 m + 2 w, 8 m + 2 w + 1},
//    all feature blocks, TWO row groups) — except that a row group is a half tile (rows = the x1 | x2 rows of pairs 0 .. 7,
//    or of pairs 8 .. 15), not a side; waves 4 .. 7 wait at the exchange barrier (their SIMD slots would be empty anyway);
//  * every partial sum leaves for LDS (96 KB: [owner wave][source][unit]), and from the exchange on waves 0 .. 3 are half
//    tile 0 and waves 4 .. 7 half tile 1, each running the half-tile kernel's phases on one row group per wave: two waves
//    per SIMD, one wave's latency chains beside the other's MFMAs.
// Arithmetic per element as in the other small-batch kernels; sums associate as in the half-tile kernel.  512-d x-vectors,
// NB = 10 / 11; used above one half tile per CU (below: nplda_train_fb_half.h).
#pragma once
#include "../neuralplda_amd/csrc/nplda_train_fb_half.h"

namespace nplda {

#ifdef NPLDA_FBD_STAMPS  // tools/exp_fbh.hip only
__device__ unsigned long long g_fbd_stamps[64];  // [0, 32): wave 0, [32, 64): wave 4 of block NPLDA_FBD_STAMPS
#define NPLDA_FBD_STAMP(i) do { if (blockIdx.x == NPLDA_FBD_STAMPS && (threadIdx.x & 255) == 0) { \
    const int o_ = threadIdx.x ? 32 : 0; g_fbd_stamps[o_ + i] = __builtin_amdgcn_s_memrealtime(); g_fbd_stamps[o_ + 16 + i] = __builtin_readcyclecounter(); } } while (0)
#else
#define NPLDA_FBD_STAMP(i) do {} while (0)
#endif

constexpr int duo_lds_f4() { return 8 * 4 * 3 * 64; }  // the layer-1 exchange: [owner wave (8)][source wave (4)][unit (3)][lane]

// target_count_issue / target_count_wave (nplda_bwd_loss.h) for a 256-thread HALF of a block: t256 = threadIdx.x & 255
__device__ __forceinline__ void target_count_issue_t(const BwdLoss& L, TargetEarly& e, int t256) {
    const int nv = (int)(L.B / 4);
    const f32x4* t4 = reinterpret_cast<const f32x4*>(L.t);
#pragma unroll
    for (int q = 0; q < 4; ++q) {
        const int i = t256 + 256 * q;
        e.v[q] = t4[i < nv ? i : 0];
    }
}
__device__ __forceinline__ float target_count_wave_t(const BwdLoss& L, const TargetEarly& e, int t256) {
    const int nv = (int)(L.B / 4);
    const f32x4* t4 = reinterpret_cast<const f32x4*>(L.t);
    float cnt = 0.f;
#pragma unroll
    for (int q = 0; q < 4; ++q) {
        const f32x4 v = e.v[q];
        cnt += t256 + 256 * q < nv ? (v[0] + v[1]) + (v[2] + v[3]) : 0.f;
    }
    for (int i = t256 + 1024; i < nv; i += 256) {
        const f32x4 v = t4[i];
        cnt += (v[0] + v[1]) + (v[2] + v[3]);
    }
    for (long long i = 4LL * nv + t256; i < L.B; i += 256) cnt += L.t[i];
    cnt = row16_sum(cnt);
    cnt = wave_xor_add(cnt, 16);
    return wave_xor_add(cnt, 32);
}

template <int NB, bool ROWS, bool XBF = false, int DX = 0>
__global__ __launch_bounds__(512, 2) void train_fb_duo_kernel(const TrainFbArgs a) {
    static_assert(NB == 10 || NB == 11, "the recipe shapes");
    static_assert(!XBF || ROWS, "bf16 rows: the staged form");
    constexpr int NW = 4, LB = NB - 8, KSW = 8, PF = 4;
    constexpr int XD = NB == 11 ? 3 : 4;  // x ring (NB = 11: 88 accumulator + 88 weight registers leave room for three sets)
    __shared__ f32x4 lbuf[duo_lds_f4()];
    __shared__ f32x4 ybuf[2][NB * 64];   // a half's y / dz / du tile: apart from the exchange region, the halves run apart
    __shared__ unsigned hb[2];           // per-half barrier counters (NPLDA_DUO_SPLIT: the halves synchronise separately)
    __shared__ float red[2][NW][16];
    __shared__ float cnt_s[2][NW];
    __shared__ double lacc[2][kHalfPairs][kLossNS];
    __shared__ float lcs[2][nplda_loss::kMaxK + 1];

    const int tid = threadIdx.x;
    const int t256 = tid & 255;
    const int lane = tid & 63;
    const int wave8 = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int half = wave8 >> 2;   // the half tile this wave belongs to behind layer 1
    const int wave = wave8 & 3;    // its wave id inside the half (= its k-quarter in layer 1, waves 0 .. 3)
    const int j = lane & 15;
    const int g = lane >> 4;
    const int side = j >> 3;
    const bool own_lo = wave < LB;
    f32x4 (*yl)[64] = reinterpret_cast<f32x4 (*)[64]>(&ybuf[half][0]);
    if ((tid & 255) == 0) hb[half] = 0u;
    unsigned epoch = 0;
    // a barrier among the four waves of ONE half (LDS counter + poll): the two halves of the block drift apart behind the
    // exchange, so that one half's serial stretches sit beside the other's MFMA phases
    auto hbar = [&]() {
#ifdef NPLDA_DUO_SPLIT
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");
        epoch += 4;
        if (lane == 0) __hip_atomic_fetch_add(&hb[half], 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
        while (__hip_atomic_load(&hb[half], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP) < epoch) __builtin_amdgcn_s_sleep(1);
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup");
#else
        __syncthreads();
#endif
    };
    const __amdgpu_buffer_rsrc_t prsrc = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(a.packed), 0, 0x7fffffff, 0x00020000);
    const unsigned lane16 = (unsigned)lane * 16u;
    auto ldw = [&](int soff) {
        return __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(prsrc, (int)lane16, soff, 0));
    };
    const BwdLoss& ls = a.ls;
    const long long tile = 2LL * blockIdx.x + half;          // this wave's half tile (row of pq / loss partials)
    const long long t0 = tile * kHalfPairs;
    const bool ok = t0 + (j & 7) < a.n;
    const long long pr = ok ? t0 + (j & 7) : a.n - 1;
    const long long R = side ? a.n + pr : pr;
    const f32x4* b1p = reinterpret_cast<const f32x4*>(a.packed + a.ob1);
    const f32x4* b2p = reinterpret_cast<const f32x4*>(a.packed + a.ob2);
    const f32x4* Qp = reinterpret_cast<const f32x4*>(a.packed + a.oQ);
    const f32x4* Pp = reinterpret_cast<const f32x4*>(a.packed + a.oP);
    if (a.step_bump != nullptr && blockIdx.x == 0 && tid == 0) a.step_bump[0] += 1.0f;
    if (a.rec_bump != nullptr && blockIdx.x == 0 && tid == 0) a.rec_bump[0] += 1;
    const f32x4 zero4 = {0.f, 0.f, 0.f, 0.f};
    TargetEarly te;
    float ti;
    PairLossConsts lc;
    NPLDA_FBD_STAMP(0);

    if (half == 0) {
        // ---- layer 1 (waves 0 .. 3): this wave's k16-steps, all feature blocks, both half tiles --------------------------
        auto blk = [&](int s) { return half_blk<NB>(s, wave); };
        const float* xrow[2];
        float* xstage[2];
        bool okr[2];
#pragma unroll
        for (int rg = 0; rg < 2; ++rg) {
            const long long tp = (2LL * blockIdx.x + rg) * kHalfPairs + (j & 7);
            okr[rg] = tp < a.n;
            const long long prr = okr[rg] ? tp : a.n - 1;
            long long xr = prr;
            if constexpr (ROWS) {
                if (a.ia != nullptr) {
                    xr = (side ? a.ib : a.ia)[prr];
                    xr = xr < 0 ? 0 : (xr < a.ntab ? xr : a.ntab - 1);
                }
            }
            const float* xbase = side ? a.xb : a.xa;
            if constexpr (XBF) xrow[rg] = reinterpret_cast<const float*>(reinterpret_cast<const unsigned short*>(xbase) + xr * a.ldx + 4 * g + 32 * wave);
            else xrow[rg] = xbase + xr * a.ldx + 4 * g + 32 * wave;
            xstage[rg] = nullptr;
            if constexpr (ROWS) {
                if (a.xsa != nullptr) xstage[rg] = (side ? a.xsb : a.xsa) + prr * a.ldxs + 4 * g + 32 * wave;
            }
        }
        int wofs[NB];
#pragma unroll
        for (int s = 0; s < NB; ++s) wofs[s] = __builtin_amdgcn_readfirstlane(blk(s) * 1024);
        auto w1step = [&](int i) { return (2 * wave + 8 * (i >> 1) + (i & 1)) * (NB * 1024); };
        auto kofs = [](int i) { return 16 * (8 * (i >> 1) + (i & 1)); };
        auto ldx = [&](int i, int rg) -> f32x4 {
            // (selects between two pointers held in scalars of their own: an indexed array of pointers comes back from the
            // stack as generic pointers and the loads turn into flat_load — nplda_l1_ksplit.h)
            const float* p = rg ? xrow[1] : xrow[0];
            if constexpr (XBF) {
                typedef unsigned u32x2 __attribute__((ext_vector_type(2)));
                const u32x2 r = *reinterpret_cast<const u32x2*>(reinterpret_cast<const unsigned short*>(p) + kofs(i));
                return f32x4{__uint_as_float(r[0] << 16), __uint_as_float(r[0] & 0xffff0000u), __uint_as_float(r[1] << 16),
                             __uint_as_float(r[1] & 0xffff0000u)};
            } else {
                return *reinterpret_cast<const f32x4*>(p + kofs(i));
            }
        };
        f32x4 wf[2][NB], xf[XD][2], acc[NB][2];
#pragma unroll
        for (int i = 0; i < XD - 1; ++i)
#pragma unroll
            for (int rg = 0; rg < 2; ++rg) xf[i][rg] = ldx(i, rg);
#pragma unroll
        for (int s = 0; s < NB; ++s) wf[0][s] = ldw(w1step(0) + wofs[s]);
        NPLDA_FBD_STAMP(1);
        // every partial sum leaves for LDS: unit (block b, half tile rg) belongs to wave 4 rg + v, v = b & 3 (b < 8) or b - 8
        auto export_slot = [&](int s, int rg, const f32x4& val) {
            const int b = blk(s);
            const int v = b < 8 ? (b & 3) : b - 8;
            const int u = b < 8 ? (b >> 2) : 2;
            int lo = lane;
            asm volatile("" : "+v"(lo));
            lbuf[(((4 * rg + v) * 4 + wave) * 3 + u) * 64 + lo] = val;
        };
        // the refills of step i, spread through its 8 NB MFMAs (one load per two MFMAs, pinned): x three (two) steps ahead,
        // weights one step ahead
        auto refill = [&](int i, int q, int wnext) {
            if (q < 2) {
                if (i + XD - 1 < KSW) xf[(i + XD - 1) % XD][q] = ldx(i + XD - 1, q);
            } else if (q >= NB && q < 2 * NB) {
                if (i + 1 < KSW) wf[(i + 1) & 1][q - NB] = ldw(wnext + wofs[q - NB]);
            }
        };
        auto stage_step = [&](int i) {
            if constexpr (ROWS) {
#pragma unroll
                for (int rg = 0; rg < 2; ++rg) {
                    float* xs = rg ? xstage[1] : xstage[0];
                    if ((rg ? okr[1] : okr[0]) && xs != nullptr) *reinterpret_cast<f32x4*>(xs + kofs(i)) = xf[i % XD][rg];
                }
            }
        };
#pragma unroll
        for (int i = 0; i < KSW - 1; ++i) {
            const int wnext = w1step(i + 1);
            stage_step(i);
#pragma unroll
            for (int r = 0; r < 4; ++r) {
#pragma unroll
                for (int s = 0; s < NB; ++s) {
#pragma unroll
                    for (int rg = 0; rg < 2; ++rg)
                        acc[s][rg] = __builtin_amdgcn_mfma_f32_16x16x4f32(wf[i & 1][s][r], xf[i % XD][rg][r],
                                                                          (i == 0 && r == 0) ? zero4 : acc[s][rg], 0, 0, 0);
                    refill(i, r * NB + s, wnext);
                    __builtin_amdgcn_sched_barrier(0);
                }
            }
        }
        {   // the last step block-major: a block's sums are final after its 8 MFMAs and leave for LDS under the next block's
            constexpr int i = KSW - 1;
            stage_step(i);
#pragma unroll
            for (int s = 0; s < NB; ++s) {
#pragma unroll
                for (int r = 0; r < 4; ++r)
#pragma unroll
                    for (int rg = 0; rg < 2; ++rg)
                        acc[s][rg] = __builtin_amdgcn_mfma_f32_16x16x4f32(wf[i & 1][s][r], xf[i % XD][rg][r], acc[s][rg], 0, 0, 0);
                if (s > 0) {
                    export_slot(s - 1, 0, acc[s - 1][0]);
                    export_slot(s - 1, 1, acc[s - 1][1]);
                }
                __builtin_amdgcn_sched_barrier(0);
            }
            export_slot(NB - 1, 0, acc[NB - 1][0]);
            export_slot(NB - 1, 1, acc[NB - 1][1]);
        }
    }
    // the batch's targets and thresholds (every wave: it needs them for its own half tile)
    if (ls.B >= 4) target_count_issue_t(ls, te, t256);
    ti = ls.t[pr];
    loss_consts_theta(ls, lc);
    // W2 fragments of layer 2: on their way during the exchange
    const int iW2 = (int)(a.oW2 * 4), iW2T = (int)(a.oW2T * 4);
    const int ob[3] = {__builtin_amdgcn_readfirstlane(wave * 1024), __builtin_amdgcn_readfirstlane((wave + 4) * 1024),
                       __builtin_amdgcn_readfirstlane((own_lo ? 8 + wave : NB - 1) * 1024)};
    f32x4 w2[PF][3];
    auto fetch2 = [&](int base, int slot, int kb) {
        const int kbc = kb < NB ? kb : NB - 1;
#pragma unroll
        for (int i = 0; i < 3; ++i) w2[slot][i] = ldw(base + kbc * (NB * 1024) + ob[i]);
    };
#pragma unroll
    for (int s = 0; s < PF; ++s) fetch2(iW2, s, s);
    NPLDA_FBD_STAMP(2);
    __syncthreads();
    NPLDA_FBD_STAMP(3);
    // own units: own + next wave + ... (a fixed order), the bias, the partial row norms
    f32x4 u[3];
    {
        // the four computing waves' partial sums of this wave's units, in wave order (a fixed association), then the bias
        auto own_sum = [&](int un, int b) {
            const f32x4* rp = lbuf + ((size_t)wave8 * 4 * 3 + un) * 64 + lane;
            f32x4 v = rp[0] + rp[3 * 64];
            v += rp[2 * 3 * 64];
            v += rp[3 * 3 * 64];
            return v + b1p[4 * b + g];
        };
        u[0] = own_sum(0, wave);
        u[1] = own_sum(1, wave + 4);
        u[2] = own_lo ? own_sum(2, 8 + wave) : zero4;
        float ss = 0.f;
#pragma unroll
        for (int i = 0; i < 3; ++i)
#pragma unroll
            for (int r = 0; r < 4; ++r) ss = fmaf(u[i][r], u[i][r], ss);
        ss = wave_xor_add(ss, 16);
        ss = wave_xor_add(ss, 32);
        if (g == 0) red[half][wave][j] = ss;
        if (ls.B < 4) {
#pragma unroll
            for (int q = 0; q < 4; ++q) te.v[q] = zero4;
        }
        const float cw = target_count_wave_t(ls, te, t256);
        if (lane == 0) cnt_s[half][wave] = cw;
    }
    hbar();  // every exchange read is done: the y tile may overwrite the region
    const double Ntl = (double)((cnt_s[half][0] + cnt_s[half][1]) + (cnt_s[half][2] + cnt_s[half][3]));
    const double Nt = ls.gcount ? ls.gcount[0] : Ntl;
    const double Nn = ls.gcount ? ls.gcount[1] : (double)ls.B - Ntl;
    const float inv = 1.0f / fmaxf(sqrtf(((red[half][0][j] + red[half][1][j]) + red[half][2][j]) + red[half][3][j]), 1e-12f);
    f32x4 y[3];
#pragma unroll
    for (int i = 0; i < 3; ++i) {
        if (i < 2 || own_lo) {
            const int nb = i < 2 ? wave + 4 * i : 8 + wave;
            y[i] = u[i] * inv;
            yl[nb][lane] = y[i];
            if (ok) *reinterpret_cast<f32x4*>(a.out_y + R * a.ldz + 16 * nb + 4 * g) = y[i];
        } else {
            y[i] = zero4;
        }
    }
    // ---- layer 2: this wave's z blocks from all of y ---------------------------------------------------------------------
    f32x4 z[3];
    z[0] = b2p[4 * wave + g];
    z[1] = b2p[4 * (wave + 4) + g];
    z[2] = own_lo ? b2p[4 * (8 + wave) + g] : zero4;
    hbar();  // y complete (also orders the `red` reuse below)
    if (wave == NW - 1) {  // the batch constants of dL/ds (fp64 divisions) by the wave with the fewest blocks
        loss_consts_counts(ls, Nt, Nn, lc);
        if (lane == 0) {
#pragma unroll
            for (int k = 0; k < nplda_loss::kMaxK; ++k) lcs[half][k] = lc.cn[k];
            lcs[half][nplda_loss::kMaxK] = lc.ct;
        }
    }
    NPLDA_FBD_STAMP(4);
#pragma unroll
    for (int kb = 0; kb < NB; ++kb) {
        const int s = kb % PF;
        const f32x4 yv = yl[kb][lane];
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            z[0] = __builtin_amdgcn_mfma_f32_16x16x4f32(w2[s][0][r], yv[r], z[0], 0, 0, 0);
            z[1] = __builtin_amdgcn_mfma_f32_16x16x4f32(w2[s][1][r], yv[r], z[1], 0, 0, 0);
            if (own_lo) z[2] = __builtin_amdgcn_mfma_f32_16x16x4f32(w2[s][2][r], yv[r], z[2], 0, 0, 0);
        }
        fetch2(iW2, s, kb + PF);
        __builtin_amdgcn_sched_barrier(0);
    }
    NPLDA_FBD_STAMP(5);
    // ---- score: s = sum_f Q (z1^2 + z2^2) + 2 P z1 z2; the pair's other side sits 8 lanes away in the same DPP row -------
    f32x4 zo[3];
    {
        float part = 0.f;
#pragma unroll
        for (int i = 0; i < 3; ++i) {
            zo[i] = dpp_f4<0x128>(z[i]);  // row_ror:8
            if (i < 2 || own_lo) {
                const int nb = i < 2 ? wave + 4 * i : 8 + wave;
                const f32x4 q = Qp[4 * nb + g];
                const f32x4 p = Pp[4 * nb + g];
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    const float z1 = side ? zo[i][r] : z[i][r], z2 = side ? z[i][r] : zo[i][r];  // both lanes of a pair: the same bits
                    part = fmaf(q[r], fmaf(z1, z1, z2 * z2), part);
                    part = fmaf(2.0f * p[r], z1 * z2, part);
                }
            }
        }
        part = wave_xor_add(part, 16);
        part = wave_xor_add(part, 32);
        if (g == 0) red[half][wave][j] = part;
    }
    // W2^T fragments of the dy chain: on their way during the exchanges below
#pragma unroll
    for (int s = 0; s < PF; ++s) fetch2(iW2T, s, s);
    hbar();  // scores of the tile; every wave is past layer 2: the y tile is free for dz
    NPLDA_FBD_STAMP(6);
    const float si = ((red[half][0][j] + red[half][1][j]) + red[half][2][j]) + red[half][3][j];
    if (a.out_s != nullptr && wave == 0 && g == 0 && side == 0 && ok) a.out_s[t0 + j] = si;

    // ---- loss: dL/ds of the tile's pairs, their terms of the loss sums ----------------------------------------------------
    double lsum[kLossNS];
#pragma unroll
    for (int k = 0; k < nplda_loss::kMaxK; ++k) lc.cn[k] = lcs[half][k];
    lc.ct = lcs[half][nplda_loss::kMaxK];
    const float gi = loss_pair(ls, lc, si, ti, lsum);
    const float tg = ok ? 2.0f * gi : 0.f;
    if (wave == 0 && g == 0 && side == 0) {
#pragma unroll
        for (int i = 0; i < kLossNS; ++i) lacc[half][j][i] = ok ? lsum[i] : 0.0;
    }
    // ---- dz = 2 g (Q z + P z'), the pair sums for dQ / dP -------------------------------------------------------------------
#pragma unroll
    for (int i = 0; i < 3; ++i) {
        if (i < 2 || own_lo) {
            const int nb = i < 2 ? wave + 4 * i : 8 + wave;
            const f32x4 q = Qp[4 * nb + g], p = Pp[4 * nb + g];
            const f32x4 d = dz_of(tg, q, p, z[i], zo[i]);
            yl[nb][lane] = d;
            if (ok) *reinterpret_cast<f32x4*>(a.dz + R * a.ldz + 16 * nb + 4 * g) = d;
            f32x4 eq, ep;
            {
                f32x4 z1, z2;
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    z1[r] = side ? zo[i][r] : z[i][r];
                    z2[r] = side ? z[i][r] : zo[i][r];
                }
                pair_sum_terms(0.5f * tg, z1, z2, eq, ep);
            }
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                eq[r] = half8_sum(eq[r]);
                ep[r] = half8_sum(ep[r]);
            }
            if (j == 0) {
                float* o = a.pq + (size_t)tile * 2 * a.ldz + 16 * nb + 4 * g;
                *reinterpret_cast<f32x4*>(o) = eq;
                *reinterpret_cast<f32x4*>(o + a.ldz) = ep;
            }
        } else if (j == 0 && LB < NW) {
            // (nothing: the left-over blocks are written by their owners)
        }
    }
    hbar();  // dz of the tile in LDS, the loss terms of its pairs
    NPLDA_FBD_STAMP(7);
    if (t256 < kLossNS) {
        double v = 0.0;
#pragma unroll
        for (int p = 0; p < kHalfPairs; ++p) v += lacc[half][p][t256];
        ls.partial[(size_t)tile * kLossNS + t256] = v;
    }
    // ---- dy = dz W2 (A = W2^T fragments, B = dz from LDS) ---------------------------------------------------------------------
    f32x4 dy[3] = {zero4, zero4, zero4};
#pragma unroll
    for (int kb = 0; kb < NB; ++kb) {
        const int s = kb % PF;
        const f32x4 dv = yl[kb][lane];
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            dy[0] = __builtin_amdgcn_mfma_f32_16x16x4f32(w2[s][0][r], dv[r], dy[0], 0, 0, 0);
            dy[1] = __builtin_amdgcn_mfma_f32_16x16x4f32(w2[s][1][r], dv[r], dy[1], 0, 0, 0);
            if (own_lo) dy[2] = __builtin_amdgcn_mfma_f32_16x16x4f32(w2[s][2][r], dv[r], dy[2], 0, 0, 0);
        }
        fetch2(iW2T, s, kb + PF);
        __builtin_amdgcn_sched_barrier(0);
    }
    NPLDA_FBD_STAMP(8);
    // ---- F.normalize backward: du = (dy - y (y . dy)) / max(||u||, eps) ----------------------------------------------------
    float dot = 0.f;
#pragma unroll
    for (int i = 0; i < 3; ++i)
#pragma unroll
        for (int r = 0; r < 4; ++r) dot = fmaf(y[i][r], dy[i][r], dot);  // (y[2] = dy[2] = 0 where the wave owns no third block)
    dot = wave_xor_add(dot, 16);
    dot = wave_xor_add(dot, 32);
    if (g == 0) red[half][wave][j] = dot;
    hbar();
    dot = ((red[half][0][j] + red[half][1][j]) + red[half][2][j]) + red[half][3][j];
    if (inv >= 1e12f) dot = 0.f;  // the clamp branch of F.normalize: u / eps, no projection term
#pragma unroll
    for (int i = 0; i < 3; ++i) {
        if (i < 2 || own_lo) {
            const int nb = i < 2 ? wave + 4 * i : 8 + wave;
            const f32x4 uu = du_of(dy[i], y[i], dot, inv);
            if (ok) *reinterpret_cast<f32x4*>(a.du + R * a.ldz + 16 * nb + 4 * g) = uu;
            if constexpr (DX != 0) yl[nb][lane] = uu;  // (every wave is past the dy chain: the dz tile is free)
        }
    }
    NPLDA_FBD_STAMP(9);
    if constexpr (DX != 0) {
        // ---- dL/dx = du . W1 of the tile's 16 rows: wave w forms output column blocks 8 w .. 8 w + 7 from all of du (LDS) and
        // the W1^T fragments (L2) ----------------------------------------------------------------------------------------------
        constexpr int XBW = 8, PFX = 2, PFX1 = PFX + 1;
        const __amdgpu_buffer_rsrc_t ximg = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(a.packed + a.oW1T), 0,
                                                                              NB * 32 * 1024, 0x00020000);
        unsigned xvoff[XBW];
#pragma unroll
        for (int q = 0; q < XBW; ++q) xvoff[q] = (unsigned)(((XBW * wave + q) * 64 + lane) * 16);
        f32x4 xw[PFX1][XBW];
        auto fetchxw = [&](int slot, int kb) {
            const int kbc = kb < NB ? kb : NB - 1;
            const int soff = kbc * (32 * 1024);
#pragma unroll
            for (int q = 0; q < XBW; ++q)
                xw[slot][q] = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(ximg, (int)xvoff[q], soff, 0));
        };
#pragma unroll
        for (int p = 0; p < PFX; ++p) fetchxw(p, p);
        hbar();  // du of the tile in LDS
        f32x4 xacc[XBW];
#pragma unroll
        for (int kb = 0; kb < NB; ++kb) {
            const int sl = kb % PFX1;
            const f32x4 d0 = yl[kb][lane];
#pragma unroll
            for (int r = 0; r < 4; ++r) {
#pragma unroll
                for (int q = 0; q < XBW; ++q)
                    xacc[q] = __builtin_amdgcn_mfma_f32_16x16x4f32(xw[sl][q][r], d0[r], (kb == 0 && r == 0) ? zero4 : xacc[q], 0, 0, 0);
                if (r == 0 && kb + PFX < NB) fetchxw((kb + PFX) % PFX1, kb + PFX);
                __builtin_amdgcn_sched_barrier(0);
            }
        }
        if (ok) {
            void* dxp = side ? a.dx1 : a.dx0;
#pragma unroll
            for (int q = 0; q < XBW; ++q) {
                const int col = 16 * (XBW * wave + q) + 4 * g;
                const f32x4 v = xacc[q];
                if constexpr (DX == 2) {  // round to nearest even, as torch's .to(bfloat16)
                    unsigned short* dst = reinterpret_cast<unsigned short*>(dxp) + pr * a.lddx + col;
                    unsigned w[4];
#pragma unroll
                    for (int c = 0; c < 4; ++c) {
                        const unsigned b = __float_as_uint(v[c]);
                        w[c] = (b & 0x7fffffffu) > 0x7f800000u ? ((b >> 16) | 0x40u) : ((b + 0x7fffu + ((b >> 16) & 1u)) >> 16);
                    }
                    typedef unsigned u32x2 __attribute__((ext_vector_type(2)));
                    *reinterpret_cast<u32x2*>(dst) = u32x2{w[0] | (w[1] << 16), w[2] | (w[3] << 16)};
                } else {
                    float* dst = reinterpret_cast<float*>(dxp) + pr * a.lddx + col;
                    *reinterpret_cast<f32x4*>(dst) = v;
                }
            }
        }
    }
}


}  // namespace nplda

// nplda_l1_ksplit.h — layer 1 (u = W1 x + b1) of ONE 16-pair tile, split over the 4 waves of a block by K.
//
// The small-batch kernels (nplda_fwd_small.h, nplda_train_fb_small.h: one tile per block, a 4096-pair minibatch on all
// 256 CUs) split a tile over the waves by OUTPUT FEATURES, also in layer 1: every wave then loads all of the tile's x rows
// (4 x redundant) plus its own weight fragments — 4.5 - 5.5 vector-memory instructions per 20 MFMAs, and with one wave per
// SIMD each of them idles the matrix pipe ~49 cycles (design/k06_backward_and_train_step.md): the loop runs the pipe 0.70 busy.
// Here layer 1 is split by K exactly as in the balanced-tile kernel (nplda_fwd_mid.h, its T = 1 group): wave w runs the
// k16-steps {8 m + 2 w, 8 m + 2 w + 1} for ALL feature blocks and both sides — its x loads are its own whole 128-byte lines,
// its weight fragments a contiguous 2 NB KB run of the image, NB + 2 loads per 8 NB MFMAs — and the four partial sums meet
// through LDS in a fixed order (each wave exports the units it does not own: 15 KB at NB = 10, one barrier).  What comes
// back is the FEATURE-split ownership the rest of those kernels is written for: wave w holds blocks w and w + 4 (both
// sides) and its left-over slot (NB = 10: block 8 + w / 2, side w & 1; NB = 11: block 8 + w, waves 0 .. 2).
// The accumulator index space of a wave is permuted by its wave id (nplda_fwd_mid.h) so that the owned units are
// compile-time register indices.  One function for every caller: the kernels that must agree bit for bit (the one-call
// training step and the separate forward) get the same partial sums in the same order.
// 512-d x-vectors only (32 k16-steps), NB = 10 / 11.
#pragma once
#include "nplda_fwd_mid.h"

namespace nplda {

// f32x4 of LDS the exchange needs (0: this NB has no K-split form)
constexpr int l1k_lds_f4(int NB) { return NB == 10 ? 4 * 3 * 5 * 64 : (NB == 11 ? 4 * 3 * 6 * 64 : 0); }

template <int NB>
struct L1K {
    using C = MidCfg<NB, 1>;
    static constexpr int RG = 2, LS = C::LS, LR = C::LR, UW = C::UW;
    static constexpr int LDS_F4 = 4 * 3 * UW * 64;  // f32x4 of the exchange region (60 KB at NB = 10, 72 KB at NB = 11)
    static_assert(LDS_F4 == l1k_lds_f4(NB), "l1k_lds_f4");
};

// XBF: the x rows are bfloat16 (8-byte loads, widened in registers); STAGE: every wave leaves the x fragments of ITS
// k16-steps as fp32 in stage[side] (the weight-gradient kernel's copy of gathered / widened rows).
// row0 / row1: this lane's row (row j = lane & 15) of side 0 (x1) / 1 (x2), element 0; stage0 / stage1: likewise.
// Out: uF[i][rho] = blocks w + 4 i, uL[0][rho] = the left-over slot, rho = side ^ swz (swz returned; 0 or 1).
// Contains ONE __syncthreads(); `red` may be overwritten after the caller's next barrier.
// before_barrier(): called once the K loop's last MFMAs are issued, in front of the exchange barrier — the place for loads a
// caller wants in flight during the exchange without holding their registers through the loop.
struct L1KNoHook { __device__ __forceinline__ void operator()() const {} };
template <int NB, bool XBF, bool STAGE, class Hook = L1KNoHook>
__device__ __forceinline__ int l1_ksplit_tile(const float* packed, size_t image_floats, const float* row0, const float* row1,
                                              float* stage0, float* stage1, bool ok, const f32x4* b1p, int wave, int lane,
                                              f32x4* red, f32x4 (&uF)[2][2], f32x4 (&uL)[1][2], Hook before_barrier = Hook()) {
    using K = L1K<NB>;
    constexpr int RG = K::RG, LS = K::LS, LR = K::LR, UW = K::UW;
    static_assert(LS == 1, "T = 1 groups have one left-over slot per wave");
    constexpr int KSW = 8;
    // x ring: XD - 1 k16-steps ahead (a step is 80 - 88 MFMAs, 1.1 us).  NB = 11 keeps a set less: with 88 accumulator and 88
    // weight registers a fourth set spills in the staged (ROWS) form of the training kernel
    constexpr int XD = NB == 11 ? 3 : 4;
    const int g = lane >> 4;
    const int swz = mid_swz<NB>(1, wave);
    const __amdgpu_buffer_rsrc_t img =
        __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(packed), 0, (int)(image_floats * 4), 0x00020000);
    auto blk = [&](int s) { return mid_blk<NB>(1, s, wave); };
    bool lo_valid = true;
    if constexpr (NB == 11) lo_valid = wave < 3;

    // addresses: the permuted row groups' x pointers (this wave's k range folded in), the permuted blocks' fragment offsets
    const float* xr[RG];
    float* xs[RG];
    unsigned voff[NB];
#pragma unroll
    for (int rho = 0; rho < RG; ++rho) {
        // (selects between pointers that are plain function arguments: handed over as an array they came back from the
        // stack as GENERIC pointers, the x loads became flat_load and every step began with s_waitcnt vmcnt(0))
        const bool second = (rho ^ swz) != 0;
        const float* p = second ? row1 : row0;
        if constexpr (XBF) xr[rho] = reinterpret_cast<const float*>(reinterpret_cast<const unsigned short*>(p) + 4 * g + 32 * wave);
        else xr[rho] = p + 4 * g + 32 * wave;
        if constexpr (STAGE) xs[rho] = (second ? stage1 : stage0) + 4 * g + 32 * wave;
    }
    // (the permuted block's offset is wave-uniform: it goes into the load's SCALAR offset, NB SGPRs instead of NB VGPRs —
    // the staged NB = 11 form of the training kernel spilled with a vector offset per slot)
    const unsigned lane16 = (unsigned)lane * 16u;
#pragma unroll
    for (int s = 0; s < NB; ++s) voff[s] = (unsigned)__builtin_amdgcn_readfirstlane(blk(s) * 1024);
    auto kofs = [](int i) { return 16 * (8 * (i >> 1) + (i & 1)); };  // column offset of step i inside the wave's share
    auto ldx = [&](int i, int rho) -> f32x4 {
        if constexpr (XBF) {
            typedef unsigned u32x2 __attribute__((ext_vector_type(2)));
            const u32x2 r = *reinterpret_cast<const u32x2*>(reinterpret_cast<const unsigned short*>(xr[rho]) + kofs(i));
            return f32x4{__uint_as_float(r[0] << 16), __uint_as_float(r[0] & 0xffff0000u), __uint_as_float(r[1] << 16),
                         __uint_as_float(r[1] & 0xffff0000u)};
        } else {
            return *reinterpret_cast<const f32x4*>(xr[rho] + kofs(i));
        }
    };
    f32x4 wf[2][NB], xf[XD][RG];
#pragma unroll
    for (int i = 0; i < XD - 1; ++i)
#pragma unroll
        for (int rho = 0; rho < RG; ++rho) xf[i][rho] = ldx(i, rho);
#pragma unroll
    for (int s = 0; s < NB; ++s) wf[0][s] = mid_ldw(img, lane16, mid_w1_step<NB>(0, wave) + (int)voff[s]);

    // where unit (slot s, row group rho) of this wave's partial sums goes: wave v, index u, red[v][(src - v - 1) & 3][u]
    auto export_unit = [&](int s, int rho, const f32x4& val) {
        const bool own_static = s < 8 ? (s & 3) == 0 : ((s - 8) < LS && rho < LR);
        if (own_static && (s < 8 || lo_valid)) return;
        const int b = blk(s);
        const int rg = rho ^ swz;
        int v, u;
        if (s < 8) {
            v = b & 3;
            u = (b >> 2) * RG + (rg ^ mid_swz<NB>(1, v));
        } else if constexpr (NB == 10) {
            v = 2 * (b - 8) + rg;
            u = 2 * RG;
        } else {
            v = b - 8;
            u = 2 * RG + rg;
        }
        int lo = lane;
        asm volatile("" : "+v"(lo));
        red[((v * 3 + ((wave - v - 1) & 3)) * UW + u) * 64 + lo] = val;
    };

    f32x4 acc[NB][RG];
    const f32x4 zero4 = {0.f, 0.f, 0.f, 0.f};
    // the refills of step i, spread through its MFMAs (one load per block of MFMAs, pinned by sched_barrier): x three steps
    // ahead, weights one step ahead
    auto refill = [&](int i, int q, int wnext) {
        if (q < RG) {
            if (i + XD - 1 < KSW) xf[(i + XD - 1) % XD][q] = ldx(i + XD - 1, q);
        } else if (q >= NB && q < 2 * NB) {
            if (i + 1 < KSW) wf[(i + 1) & 1][q - NB] = mid_ldw(img, lane16, wnext + (int)voff[q - NB]);
        }
    };
    auto stage_step = [&](int i) {
        if constexpr (STAGE) {
            if (ok && stage0 != nullptr) {  // (null: the caller keeps no fp32 copy of the rows)
#pragma unroll
                for (int rho = 0; rho < RG; ++rho) *reinterpret_cast<f32x4*>(xs[rho] + kofs(i)) = xf[i % XD][rho];
            }
        }
    };
#pragma unroll
    for (int i = 0; i < KSW - 1; ++i) {
        const int wnext = mid_w1_step<NB>(i + 1, wave);
#pragma unroll
        for (int r = 0; r < 4; ++r) {
#pragma unroll
            for (int s = 0; s < NB; ++s) {
#pragma unroll
                for (int rho = 0; rho < RG; ++rho)
                    acc[s][rho] = __builtin_amdgcn_mfma_f32_16x16x4f32(wf[i & 1][s][r], xf[i % XD][rho][r],
                                                                       (i == 0 && r == 0) ? zero4 : acc[s][rho], 0, 0, 0);
                refill(i, r * NB + s, wnext);
                // The staged copies leave BEHIND the step's refills.  vmcnt counts stores and loads in one order: issued in
                // front of them (at the head of the step, as until round 6), the stores had to be acknowledged before the
                // step's last weight fragment counted as arrived — 18 MFMAs later; here the next wait that covers them is
                // the next step's last fragment, ~58 MFMAs on.
                if (r == 2 && s == 0) stage_step(i);
                __builtin_amdgcn_sched_barrier(0);
            }
        }
    }
    {   // the last step block-major: a block's sums are final after its 4 RG MFMAs and leave for LDS under the next block's
        constexpr int i = KSW - 1;
#pragma unroll
        for (int s = 0; s < NB; ++s) {
#pragma unroll
            for (int r = 0; r < 4; ++r)
#pragma unroll
                for (int rho = 0; rho < RG; ++rho)
                    acc[s][rho] = __builtin_amdgcn_mfma_f32_16x16x4f32(wf[i & 1][s][r], xf[i % XD][rho][r], acc[s][rho], 0, 0, 0);
            if (s == NB - 1) stage_step(i);  // (nothing waits on the memory counter again before the exchange)
            if (s > 0) {
#pragma unroll
                for (int rho = 0; rho < RG; ++rho) export_unit(s - 1, rho, acc[s - 1][rho]);
            }
            __builtin_amdgcn_sched_barrier(0);
        }
#pragma unroll
        for (int rho = 0; rho < RG; ++rho) export_unit(NB - 1, rho, acc[NB - 1][rho]);
    }
    before_barrier();
    __syncthreads();
    // own units: own + next wave + ... (a fixed order), then the bias
    auto own_sum = [&](const f32x4& own, int u, int b) {
        const f32x4* rp = red + ((size_t)wave * 3 * UW + u) * 64 + lane;
        f32x4 v = own + rp[0];
        v += rp[UW * 64];
        v += rp[2 * UW * 64];
        return v + b1p[4 * b + g];
    };
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int rho = 0; rho < RG; ++rho) uF[i][rho] = own_sum(acc[4 * i][rho], i * RG + rho, blk(4 * i));
#pragma unroll
    for (int rho = 0; rho < 2; ++rho) {
        if (rho < LR) uL[0][rho] = own_sum(acc[8][rho], 2 * RG + rho, blk(8));
        else uL[0][rho] = zero4;
    }
    return swz;
}

}  // namespace nplda

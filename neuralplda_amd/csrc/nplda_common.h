// nplda_common.h — shared host/device helpers for libnplda_hip.so (gfx950 only).
#pragma once
#include <hip/hip_runtime.h>
#include <stddef.h>
#include <stdint.h>
#include "../../include/nplda_hip.h"

typedef float f32x4 __attribute__((ext_vector_type(4)));

#define NPLDA_ABI_VERSION 4  // 4: nplda_dplda_update_f32 takes loss / loss_sum; 3: the parameter image carries a W1^T fragment copy (dx = du . W1); 2: cohort control block, two-word Adam step
#define NPLDA_MAX_NB 12            // 12 x 16 = 192 features per layer
#define NPLDA_MAX_DIM (NPLDA_MAX_NB * 16)

// Number of 16-wide feature blocks the compiled kernel set uses for a layer pair: the kernels
// are instantiated square (NB1 == NB2 == NB) for NB in {2,4,8,10,11,12}; smaller models are
// zero-padded up to the next instantiated size.  Returns 0 if unsupported.
__host__ __device__ inline int nplda_kernel_nb(int D1, int D2) {
    int nb = ((D1 > D2 ? D1 : D2) + 15) / 16;
    if (nb <= 2) return 2;
    if (nb <= 4) return 4;
    if (nb <= 8) return 8;
    if (nb <= 10) return 10;
    if (nb <= 11) return 11;
    if (nb <= 12) return 12;
    return 0;
}

// Packed parameter image (all offsets in floats).  W1 and W2 are stored in the exact order the
// MFMA A-operand fragments are consumed, so a k16-step of weights is a contiguous run of
// NB x 64 lanes x float4 that lands in LDS linearly and is read back with conflict-free
// ds_read_b128 (lane l reads bytes [16 l, 16 l + 16) of its block):
//   W1p[ks][nb][lane][i] = W1[16 nb + (lane & 15)][16 ks + 4 (lane >> 4) + i]   (0 outside D1 x D0)
//   W2p[kb][nb][lane][i] = W2[16 nb + (lane & 15)][16 kb + 4 (lane >> 4) + i]   (0 outside D2 x D1)
//   W2Tp[kb][nb][lane][i] = W2[16 kb + 4 (lane >> 4) + i][16 nb + (lane & 15)]  (W2^T, backward dy = dz W2)
//   W1Tp[kb][xb][lane][i] = W1[16 kb + 4 (lane >> 4) + i][16 xb + (lane & 15)]  (W1^T, input gradient dx = du W1;
//                                                                               xb < KS1 column blocks of the x-vector)
// followed by zero-padded b1, b2, Q, P = P_sqrt^2 (NB*16 each).
struct NpldaLayout {
    int D0, D1, D2;
    int NB, KS1;  // 16-blocks per layer (square kernel), k16-steps over D0
    size_t oW1, oW2, oW2T, oW1T, ob1, ob2, oQ, oP, total;
};

__host__ __device__ inline NpldaLayout nplda_layout(int D0, int D1, int D2) {
    NpldaLayout L;
    L.D0 = D0; L.D1 = D1; L.D2 = D2;
    L.NB = nplda_kernel_nb(D1, D2);
    L.KS1 = (D0 + 15) / 16;
    L.oW1 = 0;
    L.oW2 = L.oW1 + (size_t)L.KS1 * L.NB * 256;
    L.oW2T = L.oW2 + (size_t)L.NB * L.NB * 256;
    L.oW1T = L.oW2T + (size_t)L.NB * L.NB * 256;
    L.ob1 = L.oW1T + (size_t)L.NB * L.KS1 * 256;
    L.ob2 = L.ob1 + (size_t)L.NB * 16;
    L.oQ = L.ob2 + (size_t)L.NB * 16;
    L.oP = L.oQ + (size_t)L.NB * 16;
    // slack: the kernels fetch whole weight chunks (up to 4 k16-steps) without bounds checks, so that every load
    // is unconditional and the compiler can count s_waitcnt vmcnt(N) exactly (a predicated load forces vmcnt(0))
    L.total = L.oP + (size_t)L.NB * 16 + (size_t)4 * L.NB * 256;
    return L;
}

static inline int nplda_dims_ok(int D0, int D1, int D2) {
    return D0 > 0 && D1 > 0 && D2 > 0 && (D0 % 4) == 0 && nplda_kernel_nb(D1, D2) != 0;
}

static inline int nplda_launch_status() {
    hipError_t e = hipGetLastError();
    return e == hipSuccess ? NPLDA_OK : (int)e;
}

static inline bool nplda_aligned16(const void* p) { return (((uintptr_t)p) & 15u) == 0; }

// An MFMA accumulator fragment of a (16 features x 16 rows) block has lane 16 g + j on row j, features 4 g .. 4 g + 3: stored
// as it is, CONSECUTIVE lanes write 16 bytes to sixteen different rows, and a CU gets 15 B / clk out of such stores where
// pieces of 64 contiguous bytes per four lanes get 29 - 43 (tools/exp_store_patterns.hip, round 6).  This moves the rows to
// the lane's high bits — lane L receives row L >> 2, features 4 (L & 3) .. — by four ds_bpermute (the LDS crossbar, no memory);
// the caller stores the result at row (L >> 2), column 16 b + 4 (L & 3).  All 64 lanes must be active.
// (Component by component: written as a loop over z[r] of a const reference hipcc 7.2 permuted z[0] four times.)
__device__ __forceinline__ f32x4 frag_rows_to_high_lanes(const f32x4& z, int lane) {
    const int src4 = 4 * (16 * (lane & 3) + (lane >> 2));
    f32x4 t;
    t.x = __int_as_float(__builtin_amdgcn_ds_bpermute(src4, __float_as_int(z.x)));
    t.y = __int_as_float(__builtin_amdgcn_ds_bpermute(src4, __float_as_int(z.y)));
    t.z = __int_as_float(__builtin_amdgcn_ds_bpermute(src4, __float_as_int(z.z)));
    t.w = __int_as_float(__builtin_amdgcn_ds_bpermute(src4, __float_as_int(z.w)));
    return t;
}

// v + (v of lane ^ mask).  For mask = 16 / 32 (whole rows of 16 lanes: the MFMA k-groups) gfx950 exchanges rows on the
// VALU: v_permlane16_swap(a, b) swaps the odd rows of a with the even rows of b, v_permlane32_swap the upper half of a
// with the lower half of b; with a = b = v the two results are (own, partner) in some order, and the sum is bit-identical
// to the shuffle form.  __shfl_xor is ds_bpermute_b32: a round trip through the LDS crossbar in front of dependent MFMAs.
__device__ __forceinline__ float wave_xor_add(float v, int mask) {
    const unsigned u = __float_as_uint(v);
    if (mask == 16) {
        const auto r = __builtin_amdgcn_permlane16_swap(u, u, false, false);
        return __uint_as_float(r[0]) + __uint_as_float(r[1]);
    }
    if (mask == 32) {
        const auto r = __builtin_amdgcn_permlane32_swap(u, u, false, false);
        return __uint_as_float(r[0]) + __uint_as_float(r[1]);
    }
    return v + __shfl_xor(v, mask, 64);
}

// Sum over the 16 lanes of a DPP row, every lane of the row getting the result: the four steps of the xor butterfly
// (masks 1, 2, 4, 8 — the same pairs of partial sums at every level, hence the same bits: fp add commutes) as quad_perm /
// row_half_mirror / row_mirror operand modifiers instead of four ds_bpermute round trips.
template <int CTRL>
__device__ __forceinline__ float dpp_f32(float v) {
    return __uint_as_float((unsigned)__builtin_amdgcn_update_dpp(0, (int)__float_as_uint(v), CTRL, 0xf, 0xf, false));
}
__device__ __forceinline__ float row16_sum(float v) {
    v += dpp_f32<0xB1>(v);   // quad_perm [1, 0, 3, 2]
    v += dpp_f32<0x4E>(v);   // quad_perm [2, 3, 0, 1]
    v += dpp_f32<0x141>(v);  // row_half_mirror: quads 0 <-> 1, 2 <-> 3 (every lane of a quad holds the quad's sum)
    v += dpp_f32<0x140>(v);  // row_mirror: halves
    return v;
}

// nplda_cohort_common.h — device helpers shared by nplda_cohort.hip and nplda_cohort_fused.hip (gfx950).
#pragma once
#include "nplda_common.h"

namespace {

__device__ __forceinline__ unsigned f2key(float f) {
    const unsigned u = __float_as_uint(f);
    return (u & 0x80000000u) ? ~u : (u | 0x80000000u);  // ascending float order == ascending unsigned order
}
__device__ __forceinline__ float key2f(unsigned k) {
    const unsigned u = (k & 0x80000000u) ? (k & 0x7fffffffu) : ~k;
    return __uint_as_float(u);
}

// Wave64 reductions on the DPP path: xor 1 and xor 2 by quad_perm, then rotations by 4 and 8 inside each row of 16
// lanes (every lane of a row then holds the row's result), and the four row results meet through v_readlane.  __shfl_xor
// compiles to ds_bpermute_b32 — a round trip through the LDS crossbar per step and per 32-bit half, 126 of them in the
// first version of this kernel, a fifth of its time.
template <int CTRL>
__device__ __forceinline__ unsigned dpp_u32(unsigned v) {
    return (unsigned)__builtin_amdgcn_update_dpp(0, (int)v, CTRL, 0xf, 0xf, false);
}
template <int CTRL>
__device__ __forceinline__ double dpp_f64(double v) {
    const int lo = __builtin_amdgcn_update_dpp(0, __double2loint(v), CTRL, 0xf, 0xf, false);
    const int hi = __builtin_amdgcn_update_dpp(0, __double2hiint(v), CTRL, 0xf, 0xf, false);
    return __hiloint2double(hi, lo);
}
constexpr int kDppXor1 = 0xB1, kDppXor2 = 0x4E, kDppRor4 = 0x124, kDppRor8 = 0x128;

__device__ __forceinline__ unsigned wave_sum_u32(unsigned v) {
    v += dpp_u32<kDppXor1>(v);
    v += dpp_u32<kDppXor2>(v);
    v += dpp_u32<kDppRor4>(v);
    v += dpp_u32<kDppRor8>(v);
    return (unsigned)__builtin_amdgcn_readlane((int)v, 0) + (unsigned)__builtin_amdgcn_readlane((int)v, 16) +
           (unsigned)__builtin_amdgcn_readlane((int)v, 32) + (unsigned)__builtin_amdgcn_readlane((int)v, 48);
}
__device__ __forceinline__ unsigned wave_min_u32(unsigned v) {
    unsigned t;
    t = dpp_u32<kDppXor1>(v); v = t < v ? t : v;
    t = dpp_u32<kDppXor2>(v); v = t < v ? t : v;
    t = dpp_u32<kDppRor4>(v); v = t < v ? t : v;
    t = dpp_u32<kDppRor8>(v); v = t < v ? t : v;
    const unsigned a = (unsigned)__builtin_amdgcn_readlane((int)v, 0), b = (unsigned)__builtin_amdgcn_readlane((int)v, 16),
                   c = (unsigned)__builtin_amdgcn_readlane((int)v, 32), d = (unsigned)__builtin_amdgcn_readlane((int)v, 48);
    const unsigned ab = a < b ? a : b, cd = c < d ? c : d;
    return ab < cd ? ab : cd;
}
__device__ __forceinline__ unsigned wave_max_u32(unsigned v) { return ~wave_min_u32(~v); }
// fixed association ((r0 + r1) + r2) + r3 over the four rows of 16 lanes: deterministic
__device__ __forceinline__ double wave_sum_f64(double v) {
    v += dpp_f64<kDppXor1>(v);
    v += dpp_f64<kDppXor2>(v);
    v += dpp_f64<kDppRor4>(v);
    v += dpp_f64<kDppRor8>(v);
    const int lo = __double2loint(v), hi = __double2hiint(v);
    double r[4];
#pragma unroll
    for (int q = 0; q < 4; ++q)
        r[q] = __hiloint2double(__builtin_amdgcn_readlane(hi, 16 * q), __builtin_amdgcn_readlane(lo, 16 * q));
    return ((r[0] + r[1]) + r[2]) + r[3];
}


// Inverse normal CDF for the quantile bracket, |error| < 4.5e-4 (Abramowitz & Stegun 26.2.23) in ~25 fp32 instructions.
// The bracket is +-0.06 sigma wide and only SELECTS candidates (the counts decide), so this accuracy is plenty; the fp64
// normcdfinv it replaces was a quarter of the kernel's VALU instructions (every wave evaluates it).
__device__ __forceinline__ float fast_normcdfinv(float p) {
    const bool lower = p < 0.5f;
    const float pp = lower ? p : 1.0f - p;
    const float t = sqrtf(-2.0f * __logf(fmaxf(pp, 1e-30f)));
    const float num = 2.515517f + t * (0.802853f + t * 0.010328f);
    const float den = 1.0f + t * (1.432788f + t * (0.189269f + t * 0.001308f));
    const float z = t - num / den;
    return lower ? -z : z;
}

}  // namespace

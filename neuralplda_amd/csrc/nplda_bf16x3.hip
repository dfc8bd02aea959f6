// nplda_bf16x3.hip — C-ABI entry points of the split-bf16 (fp32-class accuracy) scoring kernels
// (kernel + design notes: nplda_fwd_bf16x3.h).  Opt-in: the exact-fp32 kernels stay the default.
#include "nplda_fwd_bf16x3.h"

namespace {

using namespace nplda;

int check_model(int D0, int D1, int D2) {
    if (D0 <= 0 || D1 <= 0 || D2 <= 0 || (D0 % 4) != 0) return NPLDA_EINVAL;
    if (!nplda_dims_ok(D0, D1, D2)) return NPLDA_EUNSUPPORTED;
    return NPLDA_OK;
}

bool rows_ok(const float* x, int64_t ld, int D) { return x && ld >= D && (ld % 4) == 0 && nplda_aligned16(x); }

template <int MODE>
int launch_bf3(Bf3Args a, const Bf3Layout& L, hipStream_t st) {
    constexpr int WAVES = 8;
    a.D0 = L.D0; a.KC1 = L.KC1; a.oW2 = L.oW2; a.ob1 = L.ob1; a.ob2 = L.ob2; a.oQ = L.oQ; a.oP = L.oP;
    const long long per_block = (MODE == MODE_EMBED ? 32 : 16) * WAVES;
    const long long blocks = (a.n + per_block - 1) / per_block;
    if (blocks > 0x7fffffffLL) return NPLDA_EINVAL;
    dim3 grid((unsigned)blocks), block(WAVES * 64);
#define NPLDA_LAUNCH(NBV, KPB) hipLaunchKernelGGL((nplda_fwd_bf16x3_kernel<NBV, MODE, WAVES, KPB>), grid, block, 0, st, a)
    switch (L.NB) {
        case 2: NPLDA_LAUNCH(2, 2); break;
        case 4: NPLDA_LAUNCH(4, 2); break;
        case 8: NPLDA_LAUNCH(8, 2); break;
        case 10: NPLDA_LAUNCH(10, 2); break;
        case 11: NPLDA_LAUNCH(11, 2); break;
        case 12: NPLDA_LAUNCH(12, 1); break;
        default: return NPLDA_EUNSUPPORTED;
    }
#undef NPLDA_LAUNCH
    return nplda_launch_status();
}

}  // namespace

extern "C" {

size_t nplda_bf16x3_packed_bytes(int D0, int D1, int D2) {
    if (check_model(D0, D1, D2) != NPLDA_OK) return 0;
    return bf3_layout(D0, D1, D2).total * sizeof(float);
}

int nplda_pack_params_bf16x3(const float* W1, const float* b1, const float* W2, const float* b2, const float* P_sqrt,
                             const float* Q, int D0, int D1, int D2, void* packed, size_t packed_bytes,
                             nplda_stream_t stream) {
    if (!W1 || !b1 || !W2 || !b2 || !P_sqrt || !Q || !packed) return NPLDA_EINVAL;
    if (int rc = check_model(D0, D1, D2)) return rc;
    const Bf3Layout L = bf3_layout(D0, D1, D2);
    if (packed_bytes < L.total * sizeof(float)) return NPLDA_ENOSPC;
    if (!nplda_aligned16(packed)) return NPLDA_EINVAL;
    const size_t nthreads = L.ob1 / 4 + (L.total - L.ob1);
    hipLaunchKernelGGL(nplda_pack_bf16x3_kernel, dim3((unsigned)((nthreads + 255) / 256)), dim3(256), 0,
                       (hipStream_t)stream, W1, b1, W2, b2, P_sqrt, Q, L, (float*)packed);
    return nplda_launch_status();
}

int nplda_score_pairs_bf16x3(const float* x1, const float* x2, int64_t B, int64_t ldx, const void* packed, int D0,
                             int D1, int D2, float* s, nplda_stream_t stream) {
    if (B < 0) return NPLDA_EINVAL;
    if (int rc = check_model(D0, D1, D2)) return rc;
    if (B == 0) return NPLDA_OK;
    if (!packed || !s || !nplda_aligned16(packed)) return NPLDA_EINVAL;
    if (!rows_ok(x1, ldx, D0) || !rows_ok(x2, ldx, D0)) return NPLDA_EINVAL;
    const Bf3Layout L = bf3_layout(D0, D1, D2);
    Bf3Args a = {};
    a.xa = x1; a.xb = x2; a.n = B; a.ldx = ldx; a.img = (const float*)packed; a.out_s = s;
    return launch_bf3<MODE_PAIR>(a, L, (hipStream_t)stream);
}

int nplda_embed_bf16x3(const float* x, int64_t N, int64_t ldx, const void* packed, int D0, int D1, int D2, float* z,
                       int64_t ldz, float* q, nplda_stream_t stream) {
    if (N < 0) return NPLDA_EINVAL;
    if (int rc = check_model(D0, D1, D2)) return rc;
    if (N == 0) return NPLDA_OK;
    if (!packed || !nplda_aligned16(packed) || !rows_ok(x, ldx, D0)) return NPLDA_EINVAL;
    const Bf3Layout L = bf3_layout(D0, D1, D2);
    if (!rows_ok(z, ldz, 16 * L.NB)) return NPLDA_EINVAL;
    Bf3Args a = {};
    a.xa = x; a.xb = x; a.n = N; a.ldx = ldx; a.img = (const float*)packed; a.out_z = z; a.ldz = ldz; a.out_q = q;
    return launch_bf3<MODE_EMBED>(a, L, (hipStream_t)stream);
}

}  // extern "C"

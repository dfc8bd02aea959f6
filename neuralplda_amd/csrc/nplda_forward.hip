// nplda_forward.hip — C-ABI entry points of the fused Neural-PLDA forward (kernel: nplda_fwd_kernel.h).
#include <cstdint>
#include "nplda_fwd_dispatch.h"

namespace {

using namespace nplda;

}  // namespace

extern "C" {

int nplda_abi_version(void) { return NPLDA_ABI_VERSION; }
int nplda_max_dim(void) { return NPLDA_MAX_DIM; }

const char* nplda_strerror(int code) {
    if (code == NPLDA_OK) return "ok";
    if (code == NPLDA_EINVAL) return "invalid argument (null pointer, negative size, misaligned or short row)";
    if (code == NPLDA_EUNSUPPORTED) return "dimension not supported by the compiled kernel set";
    if (code == NPLDA_ENOSPC) return "caller-provided buffer too small";
    if (code > 0) return hipGetErrorString((hipError_t)code);
    return "unknown nplda error";
}

int nplda_padded_dim(int D1, int D2) { return 16 * nplda_kernel_nb(D1, D2); }

size_t nplda_packed_bytes(int D0, int D1, int D2) {
    if (check_model(D0, D1, D2) != NPLDA_OK) return 0;
    return nplda_layout(D0, D1, D2).total * sizeof(float);
}

int nplda_pack_params_f32(const float* W1, const float* b1, const float* W2, const float* b2,
                          const float* P_sqrt, const float* Q, int D0, int D1, int D2, void* packed,
                          size_t packed_bytes, nplda_stream_t stream) {
    if (!W1 || !b1 || !W2 || !b2 || !P_sqrt || !Q || !packed) return NPLDA_EINVAL;
    if (int rc = check_model(D0, D1, D2)) return rc;
    const NpldaLayout L = nplda_layout(D0, D1, D2);
    if (packed_bytes < L.total * sizeof(float)) return NPLDA_ENOSPC;
    if (!nplda_aligned16(packed)) return NPLDA_EINVAL;
    const unsigned blocks = (unsigned)((L.total + 255) / 256);
    hipLaunchKernelGGL(nplda_pack_kernel, dim3(blocks), dim3(256), 0, (hipStream_t)stream, W1, b1, W2,
                       b2, P_sqrt, Q, L, (float*)packed);
    return nplda_launch_status();
}

int nplda_score_pairs_f32(const float* x1, const float* x2, int64_t B, int64_t ldx, const void* packed,
                          int D0, int D1, int D2, float* s, nplda_stream_t stream) {
    if (B < 0) return NPLDA_EINVAL;
    if (int rc = check_model(D0, D1, D2)) return rc;
    if (B == 0) return NPLDA_OK;
    if (!packed || !s || !nplda_aligned16(packed)) return NPLDA_EINVAL;
    if (!rows_ok(x1, ldx, D0) || !rows_ok(x2, ldx, D0)) return NPLDA_EINVAL;
    const NpldaLayout L = nplda_layout(D0, D1, D2);
    FwdArgs a = {};
    a.xa = x1; a.xb = x2; a.n = B; a.ldx = ldx; a.packed = (const float*)packed; a.out_s = s;
    return launch_fwd<MODE_PAIR>(a, L, (hipStream_t)stream);
}

int nplda_score_pairs_rows_f32(const float* table, int64_t N, int64_t ldt, const int64_t* rows1, const int64_t* rows2,
                               int64_t B, const void* packed, int D0, int D1, int D2, float* s, nplda_stream_t stream) {
    if (B < 0 || N < 0) return NPLDA_EINVAL;
    if (int rc = check_model(D0, D1, D2)) return rc;
    if (B == 0) return NPLDA_OK;
    if (!packed || !s || !rows1 || !rows2 || N < 1 || !nplda_aligned16(packed) || !rows_ok(table, ldt, D0)) return NPLDA_EINVAL;
    const NpldaLayout L = nplda_layout(D0, D1, D2);
    // only where nplda_score_pairs_f32 would run the balanced-tile kernel itself: the fused form then gives the same bits as
    // gather + score (validate()'s device-resident pass equals its generic loop); elsewhere: gather + nplda_score_pairs_f32
    if (pair_kernel_choice(B, L, mid_cus()) != FWD_MID) return NPLDA_EUNSUPPORTED;  // (same kernel as nplda_score_pairs_f32: same bits)
    FwdArgs a = {};
    a.xa = table; a.xb = table; a.n = B; a.ldx = ldt; a.packed = (const float*)packed; a.out_s = s;
    a.ia = (const long long*)rows1; a.ib = (const long long*)rows2; a.ntab = N;
    a.D0 = L.D0; a.KS1 = L.KS1;
    a.oW2 = L.oW2; a.ob1 = L.ob1; a.ob2 = L.ob2; a.oQ = L.oQ; a.oP = L.oP; a.total = L.total;
    return launch_fwd_mid<false>(a, L, (hipStream_t)stream);
}

int nplda_score_pairs_bf16rows_f32(const void* x1, const void* x2, int64_t B, int64_t ldx, const void* packed, int D0,
                                   int D1, int D2, float* s, nplda_stream_t stream) {
    if (B < 0) return NPLDA_EINVAL;
    if (int rc = check_model(D0, D1, D2)) return rc;
    if (B == 0) return NPLDA_OK;
    if (!packed || !s || !x1 || !x2 || !nplda_aligned16(packed)) return NPLDA_EINVAL;
    // rows of 2-byte elements: 8-byte loads of four columns
    if (ldx < D0 || (ldx % 4) != 0 || ((uintptr_t)x1 & 7) != 0 || ((uintptr_t)x2 & 7) != 0) return NPLDA_EINVAL;
    const NpldaLayout L = nplda_layout(D0, D1, D2);
    FwdArgs a = {};
    a.xa = (const float*)x1; a.xb = (const float*)x2; a.n = B; a.ldx = ldx; a.packed = (const float*)packed; a.out_s = s;
    return launch_fwd_pairs_bf16rows(a, L, (hipStream_t)stream);
}

const char* nplda_score_pairs_kernel_name(int64_t B, int D0, int D1, int D2) {
    if (B <= 0 || check_model(D0, D1, D2) != NPLDA_OK) return "";
    return pair_kernel_name(B, nplda_layout(D0, D1, D2));
}

int nplda_embed_f32(const float* x, int64_t N, int64_t ldx, const void* packed, int D0, int D1, int D2,
                    float* z, int64_t ldz, float* q, nplda_stream_t stream) {
    if (N < 0) return NPLDA_EINVAL;
    if (int rc = check_model(D0, D1, D2)) return rc;
    if (N == 0) return NPLDA_OK;
    if (!packed || !nplda_aligned16(packed) || !rows_ok(x, ldx, D0)) return NPLDA_EINVAL;
    const NpldaLayout L = nplda_layout(D0, D1, D2);
    if (!rows_ok(z, ldz, 16 * L.NB)) return NPLDA_EINVAL;
    FwdArgs a = {};
    a.xa = x; a.xb = x; a.n = N; a.ldx = ldx; a.packed = (const float*)packed;
    a.out_z = z; a.ldz = ldz; a.out_q = q;
    return launch_fwd<MODE_EMBED>(a, L, (hipStream_t)stream);
}

int nplda_embed_rows_f32(const float* table, int64_t N, int64_t ldt, const int64_t* rows, int64_t U, const void* packed,
                         int D0, int D1, int D2, float* z, int64_t ldz, float* q, nplda_stream_t stream) {
    if (U < 0 || N < 0) return NPLDA_EINVAL;
    if (int rc = check_model(D0, D1, D2)) return rc;
    if (U == 0) return NPLDA_OK;
    if (!packed || !rows || N < 1 || !nplda_aligned16(packed) || !rows_ok(table, ldt, D0)) return NPLDA_EINVAL;
    const NpldaLayout L = nplda_layout(D0, D1, D2);
    if (!rows_ok(z, ldz, 16 * L.NB)) return NPLDA_EINVAL;
    const bool mid_ok = (L.NB == 10 || L.NB == 11) && L.D0 == 512 && L.KS1 == 32 &&
                        pair_kernel_choice((U + 1) / 2, L, mid_cus(), nullptr, false) == FWD_MID;
    if (!mid_ok) return NPLDA_EUNSUPPORTED;
    FwdArgs a = {};
    a.xa = table; a.xb = table; a.n = U; a.ldx = ldt; a.packed = (const float*)packed;
    a.out_z = z; a.ldz = ldz; a.out_q = q;
    a.ia = (const long long*)rows; a.ntab = N;
    a.D0 = L.D0; a.KS1 = L.KS1;
    a.oW2 = L.oW2; a.ob1 = L.ob1; a.ob2 = L.ob2; a.oQ = L.oQ; a.oP = L.oP; a.total = L.total;
    return launch_fwd_mid<true>(a, L, (hipStream_t)stream);
}

int nplda_embed_pair_f32(const float* xa, int64_t Na, const float* xb, int64_t Nb, int64_t ldx, const void* packed, int D0,
                         int D1, int D2, float* z, int64_t ldz, float* q, nplda_stream_t stream) {
    if (Na < 0 || Nb < 0) return NPLDA_EINVAL;
    if (int rc = check_model(D0, D1, D2)) return rc;
    if (Na == 0) return nplda_embed_f32(xb, Nb, ldx, packed, D0, D1, D2, z, ldz, q, stream);
    if (Nb == 0) return nplda_embed_f32(xa, Na, ldx, packed, D0, D1, D2, z, ldz, q, stream);
    if (!packed || !nplda_aligned16(packed) || !rows_ok(xa, ldx, D0) || !rows_ok(xb, ldx, D0)) return NPLDA_EINVAL;
    const NpldaLayout L = nplda_layout(D0, D1, D2);
    if (!rows_ok(z, ldz, 16 * L.NB)) return NPLDA_EINVAL;
    const long long N = Na + Nb;
    // one launch where the balanced-tile kernel embeds (its row addressing takes the second table); two otherwise
    const bool mid_ok = (L.NB == 10 || L.NB == 11) && L.D0 == 512 && L.KS1 == 32 &&
                        pair_kernel_choice((N + 1) / 2, L, mid_cus(), nullptr, false) == FWD_MID;
    if (!mid_ok) {
        if (int rc = nplda_embed_f32(xa, Na, ldx, packed, D0, D1, D2, z, ldz, q, stream)) return rc;
        return nplda_embed_f32(xb, Nb, ldx, packed, D0, D1, D2, z + Na * ldz, ldz, q ? q + Na : nullptr, stream);
    }
    FwdArgs a = {};
    a.xa = xa; a.xb = xb; a.n = N; a.nsplit = Na; a.ldx = ldx; a.packed = (const float*)packed;
    a.out_z = z; a.ldz = ldz; a.out_q = q;
    a.D0 = L.D0; a.KS1 = L.KS1;
    a.oW2 = L.oW2; a.ob1 = L.ob1; a.ob2 = L.ob2; a.oQ = L.oQ; a.oP = L.oP; a.total = L.total;
    return launch_fwd_mid<true>(a, L, (hipStream_t)stream);
}

}  // extern "C"

/*
 * nplda_hip.h — C ABI of libnplda_hip.so, the MI355X (gfx950) Neural-PLDA hot path.
 *
 * The reference (iiscleap/NeuralPlda) has no FFI: its boundary is the Python class
 * utils/models.py:348-461 (NeuralPlda), utils/models.py:571-665 (GaussianBackend), the batch
 * loaders utils/sv_trials_loaders.py:418-437 and the module-level script
 * utils/adaptive_score_normalization.py:20-84.  This header is the C boundary those Python
 * symbols bind to in neuralplda_amd/ (ctypes stub shown in INTEGRATION.md).  Each entry point
 * names the reference lines it replaces.
 *
 * Conventions
 *  - every pointer is a DEVICE pointer on the current HIP device unless marked "host";
 *    row-major, innermost stride 1; float rows 16-byte aligned (ld % 4 == 0).
 *  - the caller owns every buffer, including workspaces; the library allocates nothing and
 *    never synchronises the device: kernels are enqueued on `stream` (a hipStream_t passed as
 *    void*; NULL = the legacy default stream).
 *  - return value: 0 = NPLDA_OK, negative = argument error (NPLDA_E*), positive = hipError_t of
 *    the failed launch.  Nothing throws, nothing exits.  B == 0 / N == 0 is a successful no-op.
 *  - "packed" is the MFMA-fragment-ordered image of the model parameters produced by
 *    nplda_pack_params_f32 (rebuilt whenever parameters change; 0.4-0.5 MB).
 */
#ifndef NPLDA_HIP_H
#define NPLDA_HIP_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define NPLDA_OK            0
#define NPLDA_EINVAL      (-22)  /* null pointer, negative size, misaligned row, bad ld      */
#define NPLDA_EUNSUPPORTED (-95) /* dimension outside the compiled kernel set (see _max_dim) */
#define NPLDA_ENOSPC      (-28)  /* caller-provided buffer / workspace too small             */

typedef void* nplda_stream_t;

/* Library ABI version (bumped on any signature change). */
int nplda_abi_version(void);
/* Largest layer1/layer2 dimension the compiled MFMA kernels accept (padded to 16). */
int nplda_max_dim(void);
/* Human-readable text for a return code (host string, static storage). */
const char* nplda_strerror(int code);

/* ---- parameter image -------------------------------------------------------------------- */

/* Bytes of the packed parameter image for a D0 -> D1 -> D2 model. 0 if unsupported. */
size_t nplda_packed_bytes(int D0, int D1, int D2);

/* Pack nn.Linear-layout parameters (utils/models.py:351-354: centering_and_LDA.weight (D1,D0),
 * .bias (D1), centering_and_wccn_plda.weight (D2,D1), .bias (D2), P_sqrt (D2), Q (D2)) into the
 * fragment-ordered image; also folds P = P_sqrt*P_sqrt (utils/models.py:373). */
int nplda_pack_params_f32(const float* W1, const float* b1, const float* W2, const float* b2,
                          const float* P_sqrt, const float* Q, int D0, int D1, int D2,
                          void* packed, size_t packed_bytes, nplda_stream_t stream);

/* ---- scoring ---------------------------------------------------------------------------- */

/* NeuralPlda.forward(x1, x2) (utils/models.py:378-382 = :366-370 twice + :372-376), fused:
 * s[i] = sum_d Q_d (z1_id^2 + z2_id^2) + 2 sum_d P_d z1_id z2_id,
 * z = W2 * normalize(W1 x + b1) + b2.  x1, x2: (B, D0) with row stride ldx; s: (B). */
int nplda_score_pairs_f32(const float* x1, const float* x2, int64_t B, int64_t ldx,
                          const void* packed, int D0, int D1, int D2, float* s,
                          nplda_stream_t stream);

/* NeuralPlda.extract_plda_embeddings(x) (utils/models.py:366-370) for N rows, plus the per-row
 * self term q[n] = sum_d Q_d z_nd^2 used by the indexed scorer.  z: (N, ldz) with
 * ldz >= nplda_padded_dim(D1, D2); columns D2..nplda_padded_dim-1 are written as 0.
 * q may be NULL. */
int nplda_embed_f32(const float* x, int64_t N, int64_t ldx, const void* packed, int D0, int D1,
                    int D2, float* z, int64_t ldz, float* q, nplda_stream_t stream);

/* Row width (floats) the kernels use for layer outputs of a D1/D2 model: both layers are padded
 * to the same multiple of 16 (the compiled square kernel size). 0 if unsupported. */
int nplda_padded_dim(int D1, int D2);

#ifdef __cplusplus
}
#endif
#endif /* NPLDA_HIP_H */

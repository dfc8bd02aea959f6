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

/* The library is built with -fvisibility=hidden: the entry points declared between this push and the pop at the end of
 * the header are its whole dynamic symbol table (tests/test_c_abi_cpu.py checks `nm -D`). */
#if defined(__GNUC__) || defined(__clang__)
#pragma GCC visibility push(default)
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
/* Digest of the sources this library was built from (sha256 of csrc/ and this header, first 16 hex digits; "unknown" if
 * it was not built by neuralplda_amd/build.py).  Static storage. */
const char* nplda_build_id(void);
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

/* nplda_score_pairs_f32 on the pairs (table[rows1[i]], table[rows2[i]]) of a resident (N, ldt) x-vector table: the gather of
 * load_xvec_trials_from_numbatch (utils/sv_trials_loaders.py:418-426) — two passes over 2 B x 2 KB in validate()'s loop,
 * xvector_NeuralPlda_pytorch.py:60-66 — folded into the scoring kernel (the balanced-tile kernel reads its x fragments
 * through the indices).  rows1 / rows2: int64 device arrays, values clamped into [0, N).  Covers the batches
 * nplda_score_pairs_f32 itself gives to that kernel (D0 == 512, D1 and D2 in 145..176, B between the small-batch and the full
 * streaming sizes: same bits as gather + score); otherwise NPLDA_EUNSUPPORTED: gather with nplda_gather_rows_f32 and call
 * nplda_score_pairs_f32. */
int nplda_score_pairs_rows_f32(const float* table, int64_t N, int64_t ldt, const int64_t* rows1, const int64_t* rows2,
                               int64_t B, const void* packed, int D0, int D1, int D2, float* s, nplda_stream_t stream);
/* nplda_score_pairs_f32 on BFLOAT16 rows (x-vectors straight from an extractor that runs in bf16, BASELINE configs[4]):
 * ldx in elements, rows 8-byte aligned.  The rows are widened in registers by the streaming kernels — no fp32 copy of
 * the batch is made (2 x 1 M x 512: 4 GB of writes and reads) and the scores are those of nplda_score_pairs_f32 on the
 * widened rows, bit for bit.  NPLDA_EUNSUPPORTED below the streaming sizes (convert and call nplda_score_pairs_f32). */
int nplda_score_pairs_bf16rows_f32(const void* x1, const void* x2, int64_t B, int64_t ldx, const void* packed, int D0,
                                   int D1, int D2, float* s, nplda_stream_t stream);

/* Name of the kernel nplda_score_pairs_f32 launches for a batch of B pairs of a D0 -> D1 -> D2 model on the current
 * device (the batch decides between the small-batch, the balanced-tile and the streaming schedule); "" for B <= 0 or an
 * unsupported model.  Reporting only (bench.py labels its roofline object with it); static storage. */
const char* nplda_score_pairs_kernel_name(int64_t B, int D0, int D1, int D2);

/* NeuralPlda.extract_plda_embeddings(x) (utils/models.py:366-370) for N rows, plus the per-row
 * self term q[n] = sum_d Q_d z_nd^2 used by the indexed scorer.  z: (N, ldz) with
 * ldz >= nplda_padded_dim(D1, D2); columns D2..nplda_padded_dim-1 are written as 0.
 * q may be NULL. */
int nplda_embed_f32(const float* x, int64_t N, int64_t ldx, const void* packed, int D0, int D1,
                    int D2, float* z, int64_t ldz, float* q, nplda_stream_t stream);

/* extract_plda_embeddings of TWO tables in one launch: rows [0, Na) of z / q are xa's rows, rows [Na, Na + Nb) xb's (same
 * row stride ldx) — the enroll / test rows and the cohort of one adaptive-score-normalisation call
 * (utils/adaptive_score_normalization.py:27-36 needs both embedded; utils/models.py:366-370 per row).  Same values as two
 * nplda_embed_f32 calls; one launch where the balanced-tile kernel applies (D0 == 512, D1 and D2 in 145..176, a few tiles per
 * CU), two launches otherwise. */
int nplda_embed_pair_f32(const float* xa, int64_t Na, const float* xb, int64_t Nb, int64_t ldx, const void* packed, int D0,
                         int D1, int D2, float* z, int64_t ldz, float* q, nplda_stream_t stream);

/* extract_plda_embeddings of the rows rows[0 .. U) of a resident (N, ldt) x-vector table, the gather folded into the kernel:
 * z[u] = embed(table[rows[u]]) (indices clamped into [0, N)).  What validate() does per epoch
 * (xvector_NeuralPlda_pytorch.py:56-83 scores every trial through utils/sv_trials_loaders.py:418-426 + utils/models.py:378-382;
 * a trial list names each utterance many times, so the distinct utterances are embedded ONCE and the trials scored by index,
 * nplda_score_indexed_f32).  Returns NPLDA_EUNSUPPORTED where the balanced-tile kernel does not apply (the caller then
 * gathers, nplda_gather_rows_f32, and calls nplda_embed_f32: same values). */
int nplda_embed_rows_f32(const float* table, int64_t N, int64_t ldt, const int64_t* rows, int64_t U, const void* packed,
                         int D0, int D1, int D2, float* z, int64_t ldz, float* q, nplda_stream_t stream);

/* Row width (floats) the kernels use for layer outputs of a D1/D2 model: both layers are padded
 * to the same multiple of 16 (the compiled square kernel size). 0 if unsupported. */
int nplda_padded_dim(int D1, int D2);

/* ---- training: forward with saved activations, losses, backward ---------------------------- */

/* As nplda_score_pairs_f32, additionally saving what the backward needs: y (2B, ldz) = normalised
 * layer-1 outputs (rows [0,B) = x1 side, [B,2B) = x2 side), z (2B, ldz) = embeddings in the same
 * row order, rn (2B) = 1 / max(||u||, 1e-12).  ldz must equal nplda_padded_dim(D1, D2).
 * Replaces what autograd would save for utils/models.py:366-382. */
int nplda_forward_train_f32(const float* x1, const float* x2, int64_t B, int64_t ldx,
                            const void* packed, int D0, int D1, int D2, float* s, float* y, float* z,
                            float* rn, int64_t ldz, nplda_stream_t stream);

/* Number of fp64 batch sums the loss needs: kind 0 = SoftCdet over K thresholds (2 + 4K, K <= 4),
 * kind 1 = BCE ("crossentropy", 4), kind 2 = hard Cdet at the thresholds (utils/models.py:401-404;
 * same layout as kind 0, no gradient).  0 if unsupported. */
int nplda_loss_nsums(int K, int kind);

/* Pass 1 of the loss (utils/models.py:384-393): accumulate, in fp64, every batch-global sum the loss
 * and its gradient need into sums[nplda_loss_nsums] (zeroed here).  All entries are additive over
 * shards of a batch: under data parallelism all-reduce(sum) this vector before pass 2.
 * theta: HOST array of K device pointers (the Th{beta} / threshold_Xent parameters, 1 float each). */
int nplda_loss_sums_f32(const float* s, const float* t, int64_t B, const float* const* theta, int K,
                        float alpha, int kind, double* sums, nplda_stream_t stream);

/* Pass 2: loss (1 float), g[i] = dL/ds_i (B floats, may be NULL) and dL/dtheta (K floats, may be
 * NULL) from the sums.  beta: HOST array of K floats (utils/NpldaConf.py:38). */
int nplda_loss_finish_f32(const float* s, const float* t, int64_t B, const float* const* theta,
                          const float* beta, int K, float alpha, int kind, const double* sums,
                          float* loss, float* g, float* dtheta, nplda_stream_t stream);

/* Both passes in one call for a batch that is NOT sharded (no all-reduce of the sums in between): sums, loss, g and
 * dtheta as nplda_loss_sums_f32 followed by nplda_loss_finish_f32 would give them, bit for bit (all four outputs are
 * required; kinds 0 and 1).  A batch of <= 4096 16-byte aligned scores runs as ONE single-block launch — the training
 * step's loss (utils/models.py:384-393 + its autograd) in one graph node; anything else takes the two passes. */
int nplda_loss_fwd_bwd_f32(const float* s, const float* t, int64_t B, const float* const* theta, const float* beta,
                           int K, float alpha, int kind, double* sums, float* loss, float* g, float* dtheta,
                           nplda_stream_t stream);

/* Floats in the flat gradient [dW1 (D1,D0) | db1 (D1) | dW2 (D2,D1) | db2 (D2) | dP_sqrt (D2) | dQ (D2)]. */
size_t nplda_grad_floats(int D0, int D1, int D2);
/* Bytes of caller-provided workspace nplda_backward_f32 needs for a batch of B pairs. */
size_t nplda_backward_workspace_bytes(int64_t B, int D0, int D1, int D2);

/* Backward of s = NeuralPlda.forward(x1, x2) given g = dL/ds: the autograd of utils/models.py:366-382
 * hand-derived, three launches, deterministic summation order.
 * y, z, rn: as saved by nplda_forward_train_f32.  P_sqrt: the raw (D2) parameter (chain rule of
 * P = P_sqrt^2).  Writes grad_flat (nplda_grad_floats floats). */
int nplda_backward_f32(const float* x1, const float* x2, int64_t B, int64_t ldx, const void* packed,
                       int D0, int D1, int D2, const float* g, const float* y, const float* z,
                       const float* rn, int64_t ldz, const float* P_sqrt, void* ws, size_t ws_bytes,
                       float* grad_flat, nplda_stream_t stream);

/* nplda_backward_f32 plus the gradient w.r.t. the INPUTS: dx1, dx2 (B, D0) with row stride lddx, both or neither
 * NULL — dL/dx = du . W1, what autograd hands to an x-vector extractor trained jointly with the head (the E2E model,
 * utils/models.py:251-268, feeds extractor outputs through the same two layers).  One more MFMA GEMM on the `du` rows the
 * backward already holds (csrc/nplda_matmul.hip).  Workspace: nplda_backward_ex_workspace_bytes(2 B, ..., want_dx). */
size_t nplda_backward_ex_workspace_bytes(int64_t rows, int D0, int D1, int D2, int want_dx);
int nplda_backward_ex_f32(const float* x1, const float* x2, int64_t B, int64_t ldx, const void* packed,
                          int D0, int D1, int D2, const float* g, const float* y, const float* z,
                          const float* rn, int64_t ldz, const float* P_sqrt, void* ws, size_t ws_bytes,
                          float* grad_flat, float* dx1, float* dx2, int64_t lddx, nplda_stream_t stream);

/* NeuralPlda.extract_plda_embeddings (utils/models.py:366-370) with what ITS backward needs: z (N, ldz) as
 * nplda_embed_f32, y (N, ldz) = normalised layer-1 rows, rn (N) = 1 / max(||u||, 1e-12); ldz = nplda_padded_dim. */
int nplda_embed_train_f32(const float* x, int64_t N, int64_t ldx, const void* packed, int D0, int D1, int D2,
                          float* z, float* y, float* rn, int64_t ldz, nplda_stream_t stream);

/* Backward of z = extract_plda_embeddings(x) given gz = dL/dz (N, D2) with row stride ldgz (any stride >= D2):
 * grad_flat as nplda_backward_f32 lays it out (dP_sqrt = dQ = 0: z does not depend on them) and, when dx != NULL,
 * dL/dx (N, D0).  y, rn from nplda_embed_train_f32.  Workspace: nplda_backward_ex_workspace_bytes(N, ..., dx != NULL). */
int nplda_embed_backward_f32(const float* x, int64_t N, int64_t ldx, const void* packed, int D0, int D1, int D2,
                             const float* gz, int64_t ldgz, const float* y, const float* rn, int64_t ldz,
                             void* ws, size_t ws_bytes, float* grad_flat, float* dx, int64_t lddx,
                             nplda_stream_t stream);

/* Backward of s = NeuralPlda.forward_from_plda_embeddings(z1, z2) (utils/models.py:372-376) given g = dL/ds:
 * dz1 = 2 g (Q z1 + P z2), dz2 = 2 g (Q z2 + P z1) (either may be NULL), dQ = sum g (z1^2 + z2^2),
 * dP_sqrt = 4 P_sqrt sum g z1 z2 (either may be NULL); column sums in a fixed order (deterministic). */
size_t nplda_score_embeddings_bwd_workspace_bytes(int64_t B, int D2);
int nplda_score_embeddings_bwd_f32(const float* z1, int64_t ld1, const float* z2, int64_t ld2, int64_t B, int D2,
                                   const float* P_sqrt, const float* Q, const float* g, float* dz1, int64_t ldd1,
                                   float* dz2, int64_t ldd2, float* dP_sqrt, float* dQ, void* ws, size_t ws_bytes,
                                   nplda_stream_t stream);

/* ---- small resident-matrix GEMM on row batches (input gradients; csrc/nplda_matmul.hip) ---------------------- */

/* MFMA-fragment image of a K x N matrix Wm (K <= 512, N % 4 == 0) for nplda_rows_matmul_f32.  mode 0: Wm = src (K, N)
 * row-major with stride ldw; 1: Wm = src^T (src is (N, K)); 2: Wm = src + src^T (K == N). */
size_t nplda_matrix_frag_bytes(int K, int N);
int nplda_pack_matrix_f32(const float* src, int64_t ldw, int K, int N, int mode, void* frag, size_t frag_bytes,
                          nplda_stream_t stream);
/* DPlda's quadratic form for the input-side backward, straight from logistic_regres.weight = [Wb | Ww | ws]
 * (utils/models.py:484-490): the fragment image of M + M^T, M = [[Ww, Wb], [Wb, Ww]] (2 D1 x 2 D1,
 * nplda_matrix_frag_bytes(2 D1, 2 D1) bytes) and v = [ws; ws] (2 D1 floats) in one launch — what
 * torch.cat x 3 + nplda_pack_matrix_f32(mode 2) produced. */
int nplda_dplda_quadform_f32(const float* wlr, int D1, void* frag, size_t frag_bytes, float* v, nplda_stream_t stream);
/* out[r, :] = rowscale[r] * (in[r, :K] . Wm + bias) for R rows; bias (N) and rowscale (R) may be NULL.  With
 * Wm = M + M^T, bias = v, rowscale = dL/ds this is the gradient of DPlda's quadratic form x^T M x + x^T v + c
 * (utils/models.py:484-495) w.r.t. the paired rows x = [y1; y2]. */
int nplda_rows_matmul_f32(const float* in, int64_t ldin, int64_t R, int K, const void* frag, int N,
                          const float* bias, const float* rowscale, float* out, int64_t ldout,
                          nplda_stream_t stream);
/* F.normalize backward on paired rows: du[h B + k, :] = (dy - y (y . dy)) * rn[h B + k], dy = dpaired[k, h D1 : (h+1) D1],
 * y = paired[k, h D1 : (h+1) D1]; du is (2 B, ldz) with zero padding columns (the layout nplda_backward's wgrad and
 * nplda_rows_matmul_f32 consume).  rn (2 B) from gb_score_pairs_ex_f32. */
int nplda_normalize_bwd_paired_f32(const float* dpaired, int64_t lddp, const float* paired, int64_t ldp,
                                   const float* rn, int64_t B, int D1, float* du, int64_t ldz,
                                   nplda_stream_t stream);
/* dW1 (D1, D0) and db1 (D1) of an LDA layer from du (2 B, ldz) and the inputs x1, x2 (B, D0): the wgrad half of
 * nplda_backward_f32 alone (DPlda / GaussianBackend heads have no second layer).  out: D1 * D0 + D1 floats. */
size_t nplda_lda_wgrad_workspace_bytes(int64_t B, int D0, int D1);
int nplda_lda_wgrad_f32(const float* x1, const float* x2, int64_t B, int64_t ldx, const float* du, int64_t ldz,
                        int D0, int D1, void* ws, size_t ws_bytes, float* out, nplda_stream_t stream);
/* dx1, dx2 (B, D0) = du . W1 for the same heads (W1 (D1, D0) row-major raw parameter; frag workspace inside ws). */
size_t nplda_lda_dgrad_workspace_bytes(int D0, int D1);
int nplda_lda_dgrad_f32(const float* du, int64_t ldz, int64_t B, const float* W1, int D0, int D1, void* ws,
                        size_t ws_bytes, float* dx1, float* dx2, int64_t lddx, nplda_stream_t stream);

/* ---- indexed scoring (embed each utterance once, then score index pairs) --------------------- */

/* utils/models.py:372-376 on rows of a pre-embedded table: s[p] = q[i1[p]] + q[i2[p]] +
 * 2 sum_d P_d z[i1[p], d] z[i2[p], d].  z: (N, ldz) and q: (N) from nplda_embed_f32; i1, i2: int64 (B).
 * The host gather of utils/sv_trials_loaders.py:418-426 disappears.  Out-of-range indices give NaN.
 * q may be NULL: the self terms q[i] = sum_d Q_d z[i, d]^2 (utils/models.py:374) are then formed inside the kernel from the
 * rows it reads anyway — two scattered 4-byte reads per pair less (4.6e9 -> 5.4e9 pairs/s over a 768 MB table). */
int nplda_score_indexed_f32(const float* z, int64_t ldz, const float* q, int64_t N, const int64_t* i1,
                            const int64_t* i2, int64_t B, const void* packed, int D0, int D1, int D2,
                            float* s, nplda_stream_t stream);

/* NeuralPlda.forward_from_plda_embeddings(z1, z2) (utils/models.py:372-376) on two explicit (B, D2)
 * tensors with arbitrary row strides; P_sqrt, Q are the raw (D2) parameters. */
int nplda_score_embeddings_f32(const float* z1, int64_t ld1, const float* z2, int64_t ld2, int64_t B,
                               int D2, const float* P_sqrt, const float* Q, float* s,
                               nplda_stream_t stream);

/* Device index-select out[r, :] = table[idx[r], :] from a resident (N, D0) x-vector matrix — the
 * device counterpart of load_xvec_trials_from_numbatch / _from_idbatch
 * (utils/sv_trials_loaders.py:418-437).  Out-of-range indices give NaN rows. */
int nplda_gather_rows_f32(const float* table, int64_t ldt, int64_t N, const int64_t* idx, int64_t B,
                          int D0, float* out, int64_t ldo, nplda_stream_t stream);

/* load_xvec_trials_from_numbatch (utils/sv_trials_loaders.py:418-426) on the device in ONE launch: the reference looks every
 * pair's two utterances up as mega_dict[num_to_id_dict[n]]; here `map` (nmap int64, built once per (mega_dict, num_to_id_dict))
 * takes a trial number to its row of the resident (N, ldt) x-vector table (negative: the utterance is not in the table) and
 * out1[r, :] = table[map[num1[r]], :], out2[r, :] = table[map[num2[r]], :] (B rows each, row stride ldo).  A number outside
 * [0, nmap) gives a NaN row and bad[0] |= 1, a number mapped to no row a NaN row and bad[0] |= 2: the caller reads the word
 * back and raises the reference's KeyError (the word is only ever OR-ed into, with a system-scope atomic: the caller clears
 * it, and may keep it in pinned host memory to read it after a stream synchronise without a copy). */
int nplda_gather_pairs_mapped_f32(const float* table, int64_t ldt, int64_t N, const int64_t* map, int64_t nmap,
                                  const int64_t* num1, const int64_t* num2, int64_t B, int D0, float* out1, float* out2,
                                  int64_t ldo, int32_t* bad, nplda_stream_t stream);

/* ---- adaptive score normalisation (utils/adaptive_score_normalization.py:27-73) ------------------ */

/* Recommended bytes of workspace: enough for the spilled cohort score matrix (whole matrix up to 4 GiB, else row
 * chunks, behind a 256-byte control block) and for the fused path's per-row candidate lists.  Any ws_bytes >= 256 + one
 * padded score row (4 * ceil(M / 4) * 4 bytes) is accepted by nplda_cohort_stats_f32; less gives NPLDA_ENOSPC. */
size_t nplda_cohort_workspace_bytes(int64_t R, int64_t M);

/* Smallest workspace with which nplda_cohort_stats_f32 takes its FUSED path for this shape (statistics formed in the
 * score GEMM's epilogue, no score matrix: csrc/nplda_cohort_fused.hip) — 0 when the shape is not eligible (M < 4096,
 * a top-N too large for the candidate lists).  With less workspace the spilling path is used; both are exact, they differ
 * in the last bits of the fp64 means (different summation orders). */
size_t nplda_cohort_fused_min_workspace_bytes(int64_t M, int topn, int D1, int D2);
/* Workspace for THIS call's parameters: the fused path's lists when the shape is eligible (a few KB per row), else the
 * spilled matrix as nplda_cohort_workspace_bytes. */
size_t nplda_cohort_workspace_bytes_ex(int64_t R, int64_t M, int topn, int D1, int D2);

/* Cohort score matrix + per-row statistics.  z_rows (R, ldz) / q_rows (R) and z_coh (M, ldz) / q_coh (M)
 * come from nplda_embed_f32.  For every row r: stats[4r..4r+3] = (mean, std, mean_top, std_top) of the M
 * cohort scores S[r, m] = NeuralPlda score of (row r, cohort m); std is the population std (ddof = 0);
 * "top" = the topn SMALLEST scores when select_lowest != 0 (the reference's behaviour:
 * adaptive_score_normalization.py:32-36 sorts ascending and keeps [:N]) or the topn largest otherwise.
 * The reference does not compute cohort scores at all (it reads a TSV, :27): this is new functionality.
 * Arithmetic of the fused path at D2 in 145 .. 176 (round 6): the GEMM takes each fp32 operand as three bf16 pieces and forms
 * a product from six bf16 MFMA passes with fp32 accumulation — the same fp32 products to rounding (dropped terms < 2^-26 of a
 * product); the environment variable NPLDA_COHORT_SPLIT=0, read at every call, selects fp32-input MFMAs instead. */
int nplda_cohort_stats_f32(const float* z_rows, const float* q_rows, int64_t R, const float* z_coh,
                           const float* q_coh, int64_t M, int64_t ldz, const void* packed, int D0, int D1,
                           int D2, int topn, int select_lowest, double* stats, void* ws, size_t ws_bytes,
                           nplda_stream_t stream);

/* A PREPARED cohort.  One cohort serves every trial list scored against it (utils/adaptive_score_normalization.py:27-36 reads
 * ONE cohort score table for all its trials), but the part of nplda_cohort_stats_f32 that depends on the cohort only — its
 * second and first moments, folded with P into the covariance image the row thresholds are proposed from — is redone by
 * every call, and by every rank of a row-sharded call.  nplda_cohort_prepare_f32 leaves it in `state`
 * (nplda_cohort_state_bytes bytes, 256-byte aligned, caller-owned; 0 = this shape takes the spilling path, which has nothing
 * to prepare); nplda_cohort_stats_prepared_f32 is nplda_cohort_stats_f32 reading it from there: same statistics, bit for bit.
 * The state is valid for the (z_coh, q_coh, packed, topn) it was prepared with. */
size_t nplda_cohort_state_bytes(int64_t M, int topn, int D1, int D2);
int nplda_cohort_prepare_f32(const float* z_coh, const float* q_coh, int64_t M, int64_t ldz, const void* packed, int D0,
                             int D1, int D2, int topn, void* state, size_t state_bytes, nplda_stream_t stream);
int nplda_cohort_stats_prepared_f32(const float* z_rows, const float* q_rows, int64_t R, const float* z_coh,
                                    const float* q_coh, int64_t M, int64_t ldz, const void* packed, int D0, int D1,
                                    int D2, int topn, int select_lowest, double* stats, void* ws, size_t ws_bytes,
                                    const void* state, size_t state_bytes, nplda_stream_t stream);

/* The row-statistics half alone, on a caller-provided (R, lds) fp32 score matrix (e.g. cohort scores
 * parsed from the TSV the reference consumes, adaptive_score_normalization.py:27-36). */
int nplda_row_stats_f32(const float* S, int64_t lds, int64_t R, int64_t M, int topn, int select_lowest,
                        double* stats, nplda_stream_t stream);

/* Per trial i: out[4i..4i+3] = (znorm, tnorm, snorm, asnorm1) of raw[i] given the statistics rows ie[i]
 * (enroll) and it[i] (test) (adaptive_score_normalization.py:65-73).  fp64 like the reference script. */
int nplda_asnorm_apply_f64(const double* raw, const int64_t* ie, const int64_t* it, int64_t T,
                           const double* stats, int64_t R, double* out, nplda_stream_t stream);

/* ---- GaussianBackend (utils/models.py:571-601) ------------------------------------------------------ */

/* Bytes of the packed GaussianBackend image for a D0 -> D1 LDA (pair dimension 2 D1). 0 if unsupported. */
size_t gb_packed_bytes(int D0, int D1);

/* Pack centering_and_LDA.weight (D1, D0) / .bias (D1) and the paired statistics paired_mean_target (2 D1),
 * paired_cov_inv_target (2 D1, 2 D1), paired_mean_nontarget, paired_cov_inv_nontarget
 * (utils/models.py:574-580) into the fragment-ordered image; folds M = L_n - L_t, v, c (fp64). */
int gb_pack_params_f32(const float* W1, const float* b1, const float* mu_t, const float* Lam_t,
                       const float* mu_n, const float* Lam_n, int D0, int D1, void* packed,
                       size_t packed_bytes, nplda_stream_t stream);

/* Same image for an explicit quadratic form on x = [y1; y2]: S = x^T M x + x^T v + c with M (2 D1, 2 D1) row-major,
 * v (2 D1) or NULL, c a host scalar.  DPlda.forward (utils/models.py:484-495) is this form: its 2 D^2 + D outer-product
 * features times a single linear unit collapse to M = [[Ww, Wb], [Wb, Ww]], v = [ws; ws], c = bias — the 57 970
 * features per pair the reference materialises are never formed. */
int gb_pack_quadform_f32(const float* W1, const float* b1, const float* M, const float* v, float c, int D0, int D1,
                         void* packed, size_t packed_bytes, nplda_stream_t stream);

/* The same image straight from DPlda's parameters: wlr = logistic_regres.weight (1, 2 D1^2 + D1) laid out
 * [Wb | Ww | ws] as utils/models.py:484-490 concatenates the features, blr = logistic_regres.bias (1); both DEVICE
 * pointers (no host read of the bias, so a training loop stays asynchronous). */
int gb_pack_dplda_f32(const float* W1, const float* b1, const float* wlr, const float* blr, int D0, int D1,
                      void* packed, size_t packed_bytes, nplda_stream_t stream);

/* GaussianBackend.forward(x1, x2) -> s (B) (utils/models.py:584-593) and/or
 * GaussianBackend.forward_getpaired(x1, x2) -> paired (B, 2 D1) contiguous (utils/models.py:595-601).
 * Either output pointer may be NULL (not both). */
int gb_score_pairs_f32(const float* x1, const float* x2, int64_t B, int64_t ldx, const void* packed,
                       int D0, int D1, float* s, float* paired, nplda_stream_t stream);

/* gb_score_pairs_f32 that also returns rn (2 B) = 1 / max(||LDA x||, 1e-12) (x1 side rows, then x2 side): what a
 * backward through the LDA layer needs next to the paired rows.  rn may be NULL. */
int gb_score_pairs_ex_f32(const float* x1, const float* x2, int64_t B, int64_t ldx, const void* packed,
                          int D0, int D1, float* s, float* paired, float* rn, nplda_stream_t stream);

/* The same score WITHOUT the normalisation step: rows of y1/y2 are used as layer-1 inputs as they are, i.e.
 * s = x^T M x + x^T v + c with x = [W1 y1 + b1; W1 y2 + b1].  With W1 = I (D0 = padded D1), b1 = 0 this is
 * DPlda.forward_from_plda_embeddings (utils/models.py:484-490) on already-extracted embeddings. */
int gb_score_rows_f32(const float* y1, const float* y2, int64_t B, int64_t ldy, const void* packed, int D0, int D1,
                      float* s, nplda_stream_t stream);

/* ---- weighted moments of paired rows (Gaussian-backend statistics, DPlda weight gradient) ---------------------- */

/* For c in {0, 1} (w1 may be NULL -> only c = 0):
 *     cnt[c] = sum_k w_c[k],  sum[c][i] = sum_k w_c[k] x[k][i],  sq[c][i][j] = sum_k w_c[k] x[k][i] x[k][j]
 * over the B rows of x (B, ldx), n valid columns.  Replaces the per-batch `x[mask].sum(0)` / `x[mask].t() @ x[mask]`
 * accumulation of xvector_GaussianBackend_pytorch.py:40-52 (w_0 = [t > 0.5], w_1 = [t < 0.5]; accumulate = 1 adds
 * into the outputs so batches can be streamed) and, with w_0 = dL/ds, yields the gradient of DPlda's linear unit
 * (utils/models.py:484-490) without the 2 D^2 + D features.  Products in exact fp32 on MFMA, reduction over the
 * row groups in fp64; outputs are DEVICE doubles: cnt (2), sum (2, n), sq (2, n, n) [first half only if w1 == NULL].
 * Requires n % 4 == 0, n <= 2 * nplda_max_dim(), ldx % 4 == 0, x 16-byte aligned. */
size_t nplda_moments_workspace_bytes(int64_t B, int n);
int nplda_weighted_moments_f32(const float* x, int64_t B, int64_t ldx, int n, const float* w0, const float* w1,
                               double* cnt, double* sum, double* sq, int accumulate, void* workspace,
                               size_t workspace_bytes, nplda_stream_t stream);
/* Gradient of DPlda's linear unit from the paired rows x_k = [y1 | y2] (B, ld >= 2 D1) and g = dL/ds:
 * dw (2 D1^2 + D1) = [G12 + G21 | G11 + G22 | s1 + s2], db (1) = sum g, G = sum_k g_k x_k x_k^T  (utils/models.py:484-490;
 * xvector_DPlda_pytorch.py:140-152 trains exactly these).  Same bits as nplda_weighted_moments_f32 followed by the fold in
 * fp64 and one rounding; workspace: nplda_moments_workspace_bytes(B, 2 D1). */
int nplda_dplda_grad_f32(const float* paired, int64_t B, int64_t ld, int D1, const float* g, float* dw, float* db,
                         void* workspace, size_t workspace_bytes, nplda_stream_t stream);

/* The tail of the recipe step xvector_DPlda_pytorch.py:35-43 runs (DPlda.forward -> loss -> backward -> Adam on
 * logistic_regres, the LDA frozen, :140-147) in TWO launches: the weighted moments of the paired rows, then per element of
 * [logistic_regres.weight | .bias] the gradient fold of nplda_dplda_grad_f32, torch.optim.Adam's update (the arithmetic of
 * nplda_adam_step_f32; exp_avg / exp_avg_sq: 2 D1^2 + D1 + 1 + K floats each, [weight | bias | thresholds]; step[0] counts
 * the steps, incremented here) and the store of the new value into the parameter AND into `image`, the quadratic-form image
 * of gb_pack_dplda_f32 the next forward scores with (NULL: not kept).  K thresholds (SoftCdet; dtheta from the loss call)
 * take their Adam step in the same launch.  grad_out (optional, 2 D1^2 + D1 + 1): the applied gradient.
 * loss / loss_sum (optional, ABI 4): loss_sum[0] += loss[0], the device-side running sum behind the progress line of
 * xvector_DPlda_pytorch.py:44-50 (mean of the losses since the previous line) — one launch less per step than adding it
 * from the host side. */
int nplda_dplda_update_f32(const float* paired, int64_t B, int64_t ld, int D1, const float* g, float* wlr, float* blr,
                           float* exp_avg, float* exp_avg_sq, float* const* thetas, const float* dtheta, int K, float* step,
                           float lr, float beta1, float beta2, float eps, float weight_decay, void* image, int D0,
                           float* grad_out, const float* loss, double* loss_sum, void* workspace, size_t workspace_bytes,
                           nplda_stream_t stream);

/* nplda_dplda_update_f32 with the LOSS inside its first launch (round 6): the weights of the moments are dL/ds_i of
 * utils/models.py:384-399's loss (kind 0 = SoftCdet with loss_K thresholds `loss_theta` and HOST betas, 1 = BCE with
 * loss_theta[0]), formed by the moments blocks from (s, t) with the device functions of nplda_loss_fwd_bwd_f32, and ONE more
 * block of that launch is the loss kernel itself: `sums` (nplda_loss_nsums doubles), `loss`, `dtheta` (and dL/ds into g_out
 * when it is not NULL) come out exactly as nplda_loss_fwd_bwd_f32 gives them.  The step xvector_DPlda_pytorch.py:35-43 runs
 * is then THREE launches: score | moments + loss | fold + Adam.  B <= 4096, s / t 16-byte aligned (else
 * NPLDA_EUNSUPPORTED: run the two calls).  `thetas` (K, may be 0): the thresholds that take their Adam step from `dtheta`. */
int nplda_dplda_update_loss_f32(const float* paired, int64_t B, int64_t ld, int D1, const float* s, const float* t, int kind,
                                const float* const* loss_theta, const float* beta, int loss_K, float alpha, double* sums,
                                float* loss, float* dtheta, float* g_out, float* wlr, float* blr, float* exp_avg,
                                float* exp_avg_sq, float* const* thetas, int K, float* step, float lr, float beta1, float beta2,
                                float eps, float weight_decay, void* image, int D0, float* grad_out, double* loss_sum,
                                void* workspace, size_t workspace_bytes, nplda_stream_t stream);

/* ---- detection-cost sweep (validation metrics) -------------------------------------------------------------------- */

/* NeuralPlda.minc (utils/models.py:406-436) for N scores / labels and K <= 8 betas (HOST array), replacing its
 * O(N_tgt * N) Python loop by a device sort + prefix scan + one sweep kernel.
 *   exact = 0: the reference's semantics bit for bit — thresholds swept over the target scores only, counts through
 *              arr2val (:23-27: count - 1, or 1.0 when the set is empty), float32 arithmetic in the reference's order;
 *              minc[k] = min_i P_miss_i + beta_k P_fa_i, thr[k] = the target score attaining it (first occurrence),
 *              minc_avg = sum(minc) / K.
 *   exact = 1: the true minimum of P_miss(th) + beta P_fa(th) over every distinct score and +inf ("target" iff
 *              s >= th), in fp64; eer (optional) = the equal error rate by linear interpolation at the crossing.
 * Labels: target iff t > 0.5, non-target iff t < 0.5 (as :407-408).  minc, thr (K), minc_avg (1), eer (1 or NULL)
 * are DEVICE floats.  N < 2^31.  The workspace size includes rocPRIM's temporary storage, whose size query needs a HIP
 * device: nplda_detcost_workspace_bytes returns 0 when there is none (or N is out of range). */
size_t nplda_detcost_workspace_bytes(int64_t N);
int nplda_detcost_sweep_f32(const float* scores, const float* target, int64_t N, const float* betas, int K, int exact,
                            float* minc, float* thr, float* minc_avg, float* eer, void* workspace,
                            size_t workspace_bytes, nplda_stream_t stream);

/* ---- host-side text I/O of the trial-list path (no device work; plain host pointers) ------------------------------ */

/* Rows and columns of a whitespace-separated table held in memory, with np.genfromtxt(dtype=str) semantics (the
 * reader of utils/scorefile_generator.py:26,45 and utils/sv_trials_loaders.py:377,400): any run of blanks separates
 * columns, '#' starts a comment, blank lines are skipped.  Returns the number of data rows (>= 0) and their column
 * count in *ncols, or NPLDA_EINVAL when rows have different column counts (genfromtxt raises there). */
int64_t nplda_text_scan(const char* text, size_t len, int* ncols);

/* Resolve columns 0 and 1 of every data row after the first skip_rows to numbers through an id table, replacing the
 * per-trial Python loops of utils/sv_trials_loaders.py:379-383 / :402-406 (dict look-ups, float(label)) and :432-433
 * (basename / splitext per id).  mode1 / mode2: 0 = id as written, 1 = os.path.splitext(id)[0],
 * 2 = os.path.splitext(os.path.basename(id))[0], 3 = id.replace('.sph', '') (utils/adaptive_score_normalization.py:25).
 * ids: n_ids ids separated by single '\n' bytes, id i =
 * ids[id_off[i] .. id_off[i+1] - 1) (id_off has n_ids + 1 entries); a repeated id resolves to its last occurrence
 * (dict semantics).  id_num (n_ids) maps a table position to the number stored (NULL: the position itself).
 * label_col >= 0: that column is parsed like Python's float() into label.  Rows with an unknown id, an unparsable
 * label or too few columns are skipped, as the reference's bare `except: pass` does; *first_bad_row (optional) gets
 * the first such row (relative to skip_rows) or -1, so that a caller with KeyError semantics (:433) can raise.
 * Outputs have room for every data row; *n_kept rows are written; row_of (optional) = source row of each. */
int nplda_text_lookup(const char* text, size_t len, int64_t skip_rows, int mode1, int mode2, int label_col,
                      const char* ids, const int64_t* id_off, const int64_t* id_num, int64_t n_ids, int64_t* i1,
                      int64_t* i2, float* label, int64_t* row_of, int64_t* n_kept, int64_t* first_bad_row);

/* Write a score file: optional header line, then for each of the first n data rows after skip_rows its first
 * keep_cols columns joined by tabs, a tab, and the score formatted as str(np.float32) (scores_f64 = 0, float array)
 * or str(np.float64) (scores_f64 = 1, double array) — byte for byte what
 * np.savetxt(np.c_[trials, scores.astype(str)], fmt='%s', delimiter='\t') writes at
 * utils/scorefile_generator.py:38 (keep_cols = all columns, header = columns + "\tLLR"), :55 (keep_cols = 2) and
 * utils/adaptive_score_normalization.py:81-84 (doubles, keep_cols = all but the last, header "# " + columns). */
int nplda_scores_write(const char* path, const char* text, size_t len, int64_t skip_rows, int keep_cols,
                       const char* header, const void* scores, int scores_f64, int64_t n);

/* str(np.float32(v)) / str(np.float64(v)) into out (>= 32 bytes, NUL-terminated); returns the length. */
int nplda_format_f32(float v, char* out);
int nplda_format_f64(double v, char* out);

/* One column (col < 0 counts from the end: -1 = last) of the n data rows after skip_rows parsed like Python's
 * float(): the `.astype(float)` of utils/adaptive_score_normalization.py:24,32.  NPLDA_EINVAL if a token does not
 * parse or the table has fewer / more rows than n. */
int nplda_text_column_f64(const char* text, size_t len, int64_t skip_rows, int col, double* out, int64_t n);

/* Number of distinct tokens in a column: len(np.unique(tab[:, col])) of utils/adaptive_score_normalization.py:28. */
int nplda_text_count_unique(const char* text, size_t len, int64_t skip_rows, int col, int64_t* n_unique);

/* Byte spans (start offset into text, length) of column col in rows skip_rows + k * stride, k = 0..n-1: the row ids
 * of the reshaped cohort table, utils/adaptive_score_normalization.py:38. */
int nplda_text_column_spans(const char* text, size_t len, int64_t skip_rows, int col, int64_t stride, int64_t* start,
                            int64_t* length, int64_t n);

/* ---- optimiser ----------------------------------------------------------------------------------------- */

/* torch.optim.Adam's update (xvector_NeuralPlda_pytorch.py:139: lr, weight_decay = 1e-5 as L2 term, no amsgrad)
 * applied to nseg <= 12 (param, grad, exp_avg, exp_avg_sq) segments in one launch.  The four pointer arrays and
 * numel are HOST arrays of nseg entries (device pointers / element counts).  step: DEVICE buffer of TWO 4-byte words,
 * zero-initialised by the caller: step[0] is the number of steps taken so far as a float (incremented by the launch,
 * the new value is the one used for the bias correction), step[1] is scratch of the launch (an arrival counter, zero
 * between launches).  Graph-replay safe: nothing about the step lives on the host. */
int nplda_adam_step_f32(float* const* params, const float* const* grads, float* const* exp_avg,
                        float* const* exp_avg_sq, const int64_t* numel, int nseg, float* step, float lr,
                        float beta1, float beta2, float eps, float weight_decay, nplda_stream_t stream);

/* One whole optimisation step on B <= 16384 pairs in THREE launches: forward + loss + data gradients in one kernel (dL/ds
 * of a pair needs its own score / target and the batch counts only) -> weight-gradient slabs ->
 * slab sums + Adam + refreshed fragment image + loss / dL/dtheta / threshold update.  What the reference spends
 * `loss.backward(); optimizer.step()` on (xvector_NeuralPlda_pytorch.py:66-75: ~250 ATen launches).  Same arithmetic as
 * nplda_pack_params_f32 -> nplda_forward_train_f32 -> nplda_loss_fwd_bwd_f32 -> nplda_backward_f32 ->
 * nplda_adam_step_f32; the loss sums are formed per block of 16 pairs (fp64, fixed order).
 *   params:  HOST array of the 6 DEVICE tensors W1 (D1, D0), b1, W2 (D2, D1), b2, P_sqrt, Q — updated in place;
 *   thetas:  HOST array of K device scalars (the thresholds; kind 1 = BCE uses thetas[0]), updated in place;
 *   betas:   HOST array of K floats (kind 0 = SoftCdet);  exp_avg / exp_avg_sq: nplda_grad_floats + K floats each, in
 *            flat-gradient order followed by the thresholds;  step: as nplda_adam_step_f32;
 *   packed:  IN the image of the current parameters (nplda_pack_params_f32), OUT the image of the updated ones — the
 *            caller re-packs only when the parameters were changed by something else;
 *   loss:    device scalar;  grad_out: optional nplda_grad_floats + K floats (the gradient that was applied);
 *   loss_sum: optional device double, loss_sum[0] += loss — the training log prints the MEAN of the losses since its
 *            previous line (xvector_NeuralPlda_pytorch.py:41-47), which a replayed step cannot collect one by one: its
 *            loss lives at one address.  The caller zeroes it when it reads it.
 * NPLDA_EUNSUPPORTED for B > 16384 or the hard cost (kind 2): use the separate entry points. */
size_t nplda_train_step_workspace_bytes(int64_t B, int D0, int D1, int D2);
/* The same step on the pairs (table[rows1[i]], table[rows2[i]]) of a resident (N, ldt) x-vector matrix — the form of the
 * reference's training loop, which looks every pair's two utterances up in mega_xvec_dict (utils/sv_trials_loaders.py:418-437):
 * the first kernel gathers the rows itself (and leaves them in the workspace for the weight gradients), so no gather
 * launch and no (B, D0) staging tensors.  rows1 / rows2: B int64 device indices in [0, N) (the loaders check them; the
 * kernel clamps).  D0 % 16 == 0 (else NPLDA_EUNSUPPORTED: gather with nplda_gather_rows_f32 and call nplda_train_step_f32). */
size_t nplda_train_step_rows_workspace_bytes(int64_t B, int D0, int D1, int D2);
int nplda_train_step_rows_f32(const float* table, int64_t N, int64_t ldt, const int64_t* rows1, const int64_t* rows2,
                              int64_t B, const float* target, float* const* params, int D0, int D1, int D2,
                              float* const* thetas, const float* betas, int K, float alpha, int kind, float* exp_avg,
                              float* exp_avg_sq, float* step, float lr, float beta1, float beta2, float eps,
                              float weight_decay, void* packed, void* ws, size_t ws_bytes, float* loss, double* loss_sum,
                              float* grad_out, nplda_stream_t stream);
/* The same step for a device-resident epoch of batches.  A record is [rows1 (B int64) | rows2 (B int64) | labels (B float32)],
 * 20 B bytes, records back to back (TrialLoader.device_epoch lays an epoch out this way).  The step trains on the record in
 * `stage` (device, 20 B bytes, 16-byte aligned) and its last kernel copies the epoch's NEXT record there: `cursor` is a
 * DEVICE array of three int64 — [0] the address of record 0, [1] the index of the next record to stage, [2] the record
 * count; the first kernel counts cursor[1] up, the update kernel stages record cursor[1] while cursor[1] < cursor[2].  The
 * caller stages record 0 and sets cursor = {base, 0, count} once per epoch; a captured graph of this call then walks the
 * epoch replay by replay with no copy launch and no host write per step (the reference's loop moves three host tensors to
 * the device per batch, xvector_NeuralPlda_pytorch.py:38).  B % 4 == 0 and D0 % 16 == 0 (else NPLDA_EUNSUPPORTED).
 * Workspace: as nplda_train_step_rows_f32. */
int nplda_train_step_records_f32(const float* table, int64_t N, int64_t ldt, int64_t* cursor, void* stage, int64_t B,
                                 float* const* params, int D0, int D1, int D2, float* const* thetas, const float* betas,
                                 int K, float alpha, int kind, float* exp_avg, float* exp_avg_sq, float* step, float lr,
                                 float beta1, float beta2, float eps, float weight_decay, void* packed, void* ws,
                                 size_t ws_bytes, float* loss, double* loss_sum, float* grad_out, nplda_stream_t stream);
int nplda_train_step_f32(const float* x1, const float* x2, int64_t B, int64_t ldx, const float* target,
                         float* const* params, int D0, int D1, int D2, float* const* thetas, const float* betas, int K,
                         float alpha, int kind, float* exp_avg, float* exp_avg_sq, float* step, float lr, float beta1,
                         float beta2, float eps, float weight_decay, void* packed, void* ws, size_t ws_bytes, float* loss,
                         double* loss_sum, float* grad_out, nplda_stream_t stream);

/* The data-parallel form of nplda_train_step_f32: ONE collective per step.  dL/ds_i of SoftCdet / BCE needs the pair's own
 * score and target and the GLOBAL batch's counts N_t, N_n only (utils/models.py:384-399), and a rank that slices its shard
 * out of the global minibatch knows those from the labels — so no collective has to precede the backward:
 *   nplda_train_step_grad_f32   forward + loss terms + data gradients + weight-gradient slabs of this rank's B pairs (the
 *                               same three launches), stopping at flat[0 .. nplda_train_step_flat_floats): the flat
 *                               gradient (nplda_grad_floats order) followed by the rank's fp64 loss sums as 4 x 18 floats
 *                               (four 16-bit fixed-point limbs each: integer-valued floats whose fp32 sum over <= 256
 *                               ranks is exact, so the summed loss sums come back to 2^-41 absolute).  global_counts: device [N_t, N_n] doubles of the global
 *                               batch (NULL: this rank's own counts, i.e. a single rank).  Counts Adam's step; updates nothing
 *                               else.  Workspace: nplda_train_step_workspace_bytes(B, ...).
 *   -- SUM all-reduce of flat (fp32) across the ranks --
 *   nplda_train_step_apply_f32  one launch: Adam on the parameters from flat, the refreshed parameter image, loss and
 *                               dL/dtheta from the summed loss sums, Adam on the thresholds, loss_sum[0] += loss.
 * On one rank the two calls give nplda_train_step_f32's parameters (the same slab sums and update arithmetic; the loss
 * sums pass through the limbs, 2^-41 absolute).  B <= 16384 per rank, <= 256 ranks. */
size_t nplda_train_step_flat_floats(int D0, int D1, int D2);
int nplda_train_step_grad_f32(const float* x1, const float* x2, int64_t B, int64_t ldx, const float* target,
                              const double* global_counts, float* const* params, int D0, int D1, int D2,
                              float* const* thetas, const float* betas, int K, float alpha, int kind, float* step,
                              void* packed, void* ws, size_t ws_bytes, float* flat, nplda_stream_t stream);
/* (the gradient phase on the pairs (table[rows1], table[rows2]) of a resident x-vector matrix, as nplda_train_step_rows_f32;
 * workspace: nplda_train_step_rows_workspace_bytes) */
int nplda_train_step_grad_rows_f32(const float* table, int64_t N, int64_t ldt, const int64_t* rows1, const int64_t* rows2,
                                   int64_t B, const float* target, const double* global_counts, float* const* params, int D0,
                                   int D1, int D2, float* const* thetas, const float* betas, int K, float alpha, int kind,
                                   float* step, void* packed, void* ws, size_t ws_bytes, float* flat, nplda_stream_t stream);
/* (the gradient phase of the head's end-to-end step, as nplda_train_step_dx_f32: x1 / x2 and dx1 / dx2 float32 or, with
 * io_bf16, bfloat16; dL/dx of this rank's rows is complete after this call — it depends on the other ranks through the global
 * counts only; workspace: nplda_train_step_dx_workspace_bytes) */
int nplda_train_step_grad_dx_f32(const void* x1, const void* x2, int64_t B, int64_t ldx, int io_bf16, const float* target,
                                 const double* global_counts, float* const* params, int D0, int D1, int D2,
                                 float* const* thetas, const float* betas, int K, float alpha, int kind, float* step,
                                 void* packed, void* ws, size_t ws_bytes, float* flat, void* dx1, void* dx2, int64_t lddx,
                                 nplda_stream_t stream);
int nplda_train_step_apply_f32(const float* flat, float* const* params, int D0, int D1, int D2, float* const* thetas,
                               const float* betas, int K, float alpha, int kind, float* exp_avg, float* exp_avg_sq,
                               float* step, float lr, float beta1, float beta2, float eps, float weight_decay, void* packed,
                               float* loss, double* loss_sum, nplda_stream_t stream);

/* The head's step of an end-to-end fine-tune (BASELINE configs[4]; the reference chains an x-vector extractor into this head:
 * Etdnn_Xvec_NeuralPlda, utils/models.py:251-268, and autograd hands dL/dx back to it): nplda_train_step_f32 that also returns
 * dx1, dx2 (B, lddx) = dL/dx1, dL/dx2 = du . W1 with the weights the forward used — four launches: forward + loss + data
 * gradients, weight-gradient slabs, the input-gradient product, update.  io_bf16 != 0: x1, x2 are bfloat16 rows (ldx in
 * elements) and dx1, dx2 are written in bfloat16 (round to nearest even) — the dtype a bf16 extractor works in; the head's own
 * arithmetic stays fp32 (the rows are widened exactly).  io_bf16 needs D0 == 512 and D1, D2 in 145..192.  B <= 16384.
 * Workspace: nplda_train_step_dx_workspace_bytes. */
size_t nplda_train_step_dx_workspace_bytes(int64_t B, int D0, int D1, int D2, int io_bf16);
int nplda_train_step_dx_f32(const void* x1, const void* x2, int64_t B, int64_t ldx, int io_bf16, const float* target,
                            float* const* params, int D0, int D1, int D2, float* const* thetas, const float* betas, int K,
                            float alpha, int kind, float* exp_avg, float* exp_avg_sq, float* step, float lr, float beta1,
                            float beta2, float eps, float weight_decay, void* packed, void* ws, size_t ws_bytes, float* loss,
                            double* loss_sum, float* grad_out, void* dx1, void* dx2, int64_t lddx, nplda_stream_t stream);

/* ---- split-bf16 scoring (opt-in) --------------------------------------------------------------------------- */

/* Same functions as nplda_pack_params_f32 / nplda_score_pairs_f32 / nplda_embed_f32 computed on the bf16 matrix
 * pipe with every fp32 operand split into three bf16 pieces and six MFMA passes per product (csrc/nplda_fwd_bf16x3.h):
 * fp32-class accuracy (the dropped cross terms are <= 2^-23 relative; scores within the same 2e-5 + 1e-5 |s| tolerance
 * of the fp64 oracle), ~1.5x the throughput of the exact-fp32 kernels.  Uses its own packed image. */
size_t nplda_bf16x3_packed_bytes(int D0, int D1, int D2);
int nplda_pack_params_bf16x3(const float* W1, const float* b1, const float* W2, const float* b2,
                             const float* P_sqrt, const float* Q, int D0, int D1, int D2, void* packed,
                             size_t packed_bytes, nplda_stream_t stream);
int nplda_score_pairs_bf16x3(const float* x1, const float* x2, int64_t B, int64_t ldx, const void* packed,
                             int D0, int D1, int D2, float* s, nplda_stream_t stream);
int nplda_embed_bf16x3(const float* x, int64_t N, int64_t ldx, const void* packed, int D0, int D1, int D2,
                       float* z, int64_t ldz, float* q, nplda_stream_t stream);

/* ---- measurement utility ------------------------------------------------------------------------------------- */

/* Shader-clock probe for bench.py (no reference counterpart): one wave that stays resident for window_us microseconds
 * on `stream` and writes out2[0] = shader cycles elapsed (s_memtime), out2[1] = ticks of the constant 100 MHz counter
 * (s_memrealtime) to DEVICE memory; MHz = 100 * out2[0] / out2[1].  Launched on a side stream next to the kernel under
 * test it reports the clock the chip actually holds under that kernel (the roofline peak assumes 2.4 GHz). */
int nplda_clock_probe(uint64_t* out2, unsigned window_us, nplda_stream_t stream);

#if defined(__GNUC__) || defined(__clang__)
#pragma GCC visibility pop
#endif
#ifdef __cplusplus
}
#endif
#endif /* NPLDA_HIP_H */

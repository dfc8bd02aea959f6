"""ctypes binding of libnplda_hip.so (the C ABI of include/nplda_hip.h).

There is deliberately NO fallback here: if the HIP library is missing, or a call returns a
non-zero status, a RuntimeError is raised.  Nothing in the product path computes on the CPU.
"""
import ctypes
import os

_PKG_DIR = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.path.join(_PKG_DIR, "libnplda_hip.so")

_c_f32p = ctypes.c_void_p
_c_i64 = ctypes.c_int64
_c_int = ctypes.c_int
_c_sz = ctypes.c_size_t
_c_vp = ctypes.c_void_p

# name -> (restype, argtypes); must list every symbol include/nplda_hip.h declares
SIGNATURES = {
    "nplda_abi_version": (_c_int, []),
    "nplda_max_dim": (_c_int, []),
    "nplda_strerror": (ctypes.c_char_p, [_c_int]),
    "nplda_build_id": (ctypes.c_char_p, []),
    "nplda_score_pairs_kernel_name": (ctypes.c_char_p, [_c_i64, _c_int, _c_int, _c_int]),
    "nplda_padded_dim": (_c_int, [_c_int, _c_int]),
    "nplda_packed_bytes": (_c_sz, [_c_int, _c_int, _c_int]),
    "nplda_pack_params_f32": (_c_int, [_c_f32p] * 6 + [_c_int] * 3 + [_c_vp, _c_sz, _c_vp]),
    "nplda_score_pairs_f32": (_c_int, [_c_f32p, _c_f32p, _c_i64, _c_i64, _c_vp, _c_int, _c_int, _c_int,
                                       _c_f32p, _c_vp]),
    "nplda_score_pairs_bf16rows_f32": (_c_int, [_c_vp, _c_vp, _c_i64, _c_i64, _c_vp, _c_int, _c_int, _c_int, _c_f32p, _c_vp]),
    "nplda_score_pairs_rows_f32": (_c_int, [_c_f32p, _c_i64, _c_i64, _c_vp, _c_vp, _c_i64, _c_vp, _c_int, _c_int, _c_int,
                                            _c_f32p, _c_vp]),
    "nplda_embed_f32": (_c_int, [_c_f32p, _c_i64, _c_i64, _c_vp, _c_int, _c_int, _c_int, _c_f32p, _c_i64,
                                 _c_f32p, _c_vp]),
    "nplda_embed_rows_f32": (_c_int, [_c_f32p, _c_i64, _c_i64, _c_vp, _c_i64, _c_vp, _c_int, _c_int, _c_int, _c_f32p, _c_i64,
                                      _c_f32p, _c_vp]),
    "nplda_forward_train_f32": (_c_int, [_c_f32p, _c_f32p, _c_i64, _c_i64, _c_vp, _c_int, _c_int, _c_int,
                                         _c_f32p, _c_f32p, _c_f32p, _c_f32p, _c_i64, _c_vp]),
    "nplda_loss_nsums": (_c_int, [_c_int, _c_int]),
    "nplda_loss_sums_f32": (_c_int, [_c_f32p, _c_f32p, _c_i64, ctypes.POINTER(ctypes.c_void_p), _c_int,
                                     ctypes.c_float, _c_int, _c_vp, _c_vp]),
    "nplda_loss_finish_f32": (_c_int, [_c_f32p, _c_f32p, _c_i64, ctypes.POINTER(ctypes.c_void_p),
                                       ctypes.POINTER(ctypes.c_float), _c_int, ctypes.c_float, _c_int, _c_vp,
                                       _c_f32p, _c_f32p, _c_f32p, _c_vp]),
    "nplda_loss_fwd_bwd_f32": (_c_int, [_c_f32p, _c_f32p, _c_i64, ctypes.POINTER(ctypes.c_void_p),
                                        ctypes.POINTER(ctypes.c_float), _c_int, ctypes.c_float, _c_int, _c_vp,
                                        _c_f32p, _c_f32p, _c_f32p, _c_vp]),
    "nplda_score_indexed_f32": (_c_int, [_c_f32p, _c_i64, _c_f32p, _c_i64, _c_vp, _c_vp, _c_i64, _c_vp, _c_int,
                                         _c_int, _c_int, _c_f32p, _c_vp]),
    "nplda_score_embeddings_f32": (_c_int, [_c_f32p, _c_i64, _c_f32p, _c_i64, _c_i64, _c_int, _c_f32p, _c_f32p,
                                            _c_f32p, _c_vp]),
    "nplda_clock_probe": (_c_int, [_c_vp, ctypes.c_uint, _c_vp]),
    "nplda_gather_rows_f32": (_c_int, [_c_f32p, _c_i64, _c_i64, _c_vp, _c_i64, _c_int, _c_f32p, _c_i64, _c_vp]),
    "nplda_cohort_workspace_bytes": (_c_sz, [_c_i64, _c_i64]),
    "nplda_cohort_fused_min_workspace_bytes": (_c_sz, [_c_i64, _c_int, _c_int, _c_int]),
    "nplda_cohort_workspace_bytes_ex": (_c_sz, [_c_i64, _c_i64, _c_int, _c_int, _c_int]),
    "nplda_cohort_stats_f32": (_c_int, [_c_f32p, _c_f32p, _c_i64, _c_f32p, _c_f32p, _c_i64, _c_i64, _c_vp, _c_int,
                                        _c_int, _c_int, _c_int, _c_int, _c_vp, _c_vp, _c_sz, _c_vp]),
    "nplda_cohort_state_bytes": (_c_sz, [_c_i64, _c_int, _c_int, _c_int]),
    "nplda_cohort_prepare_f32": (_c_int, [_c_f32p, _c_f32p, _c_i64, _c_i64, _c_vp, _c_int, _c_int, _c_int, _c_int, _c_vp, _c_sz,
                                          _c_vp]),
    "nplda_cohort_stats_prepared_f32": (_c_int, [_c_f32p, _c_f32p, _c_i64, _c_f32p, _c_f32p, _c_i64, _c_i64, _c_vp, _c_int,
                                                 _c_int, _c_int, _c_int, _c_int, _c_vp, _c_vp, _c_sz, _c_vp, _c_sz, _c_vp]),
    "nplda_row_stats_f32": (_c_int, [_c_f32p, _c_i64, _c_i64, _c_i64, _c_int, _c_int, _c_vp, _c_vp]),
    "nplda_asnorm_apply_f64": (_c_int, [_c_vp, _c_vp, _c_vp, _c_i64, _c_vp, _c_i64, _c_vp, _c_vp]),
    "gb_packed_bytes": (_c_sz, [_c_int, _c_int]),
    "gb_pack_params_f32": (_c_int, [_c_f32p] * 6 + [_c_int, _c_int, _c_vp, _c_sz, _c_vp]),
    "gb_pack_quadform_f32": (_c_int, [_c_f32p, _c_f32p, _c_f32p, _c_f32p, ctypes.c_float, _c_int, _c_int, _c_vp, _c_sz,
                                      _c_vp]),
    "gb_pack_dplda_f32": (_c_int, [_c_f32p, _c_f32p, _c_f32p, _c_f32p, _c_int, _c_int, _c_vp, _c_sz, _c_vp]),
    "gb_score_rows_f32": (_c_int, [_c_f32p, _c_f32p, _c_i64, _c_i64, _c_vp, _c_int, _c_int, _c_f32p, _c_vp]),
    "gb_score_pairs_f32": (_c_int, [_c_f32p, _c_f32p, _c_i64, _c_i64, _c_vp, _c_int, _c_int, _c_f32p, _c_f32p,
                                    _c_vp]),
    "nplda_moments_workspace_bytes": (_c_sz, [_c_i64, _c_int]),
    "nplda_weighted_moments_f32": (_c_int, [_c_f32p, _c_i64, _c_i64, _c_int, _c_f32p, _c_f32p, _c_vp, _c_vp, _c_vp,
                                            _c_int, _c_vp, _c_sz, _c_vp]),
    "nplda_detcost_workspace_bytes": (_c_sz, [_c_i64]),
    "nplda_detcost_sweep_f32": (_c_int, [_c_f32p, _c_f32p, _c_i64, ctypes.POINTER(ctypes.c_float), _c_int, _c_int, _c_f32p,
                                         _c_f32p, _c_f32p, _c_f32p, _c_vp, _c_sz, _c_vp]),
    "nplda_text_scan": (_c_i64, [ctypes.c_char_p, _c_sz, ctypes.POINTER(_c_int)]),
    "nplda_text_lookup": (_c_int, [ctypes.c_char_p, _c_sz, _c_i64, _c_int, _c_int, _c_int, ctypes.c_char_p, _c_vp, _c_vp,
                                   _c_i64, _c_vp, _c_vp, _c_vp, _c_vp, ctypes.POINTER(_c_i64), ctypes.POINTER(_c_i64)]),
    "nplda_scores_write": (_c_int, [ctypes.c_char_p, ctypes.c_char_p, _c_sz, _c_i64, _c_int, ctypes.c_char_p, _c_vp,
                                    _c_int, _c_i64]),
    "nplda_format_f32": (_c_int, [ctypes.c_float, ctypes.c_char_p]),
    "nplda_format_f64": (_c_int, [ctypes.c_double, ctypes.c_char_p]),
    "nplda_text_column_f64": (_c_int, [ctypes.c_char_p, _c_sz, _c_i64, _c_int, _c_vp, _c_i64]),
    "nplda_text_count_unique": (_c_int, [ctypes.c_char_p, _c_sz, _c_i64, _c_int, ctypes.POINTER(_c_i64)]),
    "nplda_text_column_spans": (_c_int, [ctypes.c_char_p, _c_sz, _c_i64, _c_int, _c_i64, _c_vp, _c_vp, _c_i64]),
    "nplda_adam_step_f32": (_c_int, [ctypes.POINTER(ctypes.c_void_p)] * 4 + [ctypes.POINTER(ctypes.c_int64), _c_int, _c_vp,
                                     ctypes.c_float, ctypes.c_float, ctypes.c_float, ctypes.c_float, ctypes.c_float,
                                     _c_vp]),
    "nplda_train_step_workspace_bytes": (_c_sz, [_c_i64, _c_int, _c_int, _c_int]),
    "nplda_train_step_f32": (_c_int, [_c_f32p, _c_f32p, _c_i64, _c_i64, _c_f32p, ctypes.POINTER(ctypes.c_void_p), _c_int,
                                      _c_int, _c_int, ctypes.POINTER(ctypes.c_void_p), ctypes.POINTER(ctypes.c_float),
                                      _c_int, ctypes.c_float, _c_int, _c_vp, _c_vp, _c_vp] + [ctypes.c_float] * 5 +
                             [_c_vp, _c_vp, _c_sz, _c_vp, _c_vp, _c_vp, _c_vp]),
    "nplda_train_step_flat_floats": (_c_sz, [_c_int, _c_int, _c_int]),
    "nplda_train_step_grad_f32": (_c_int, [_c_f32p, _c_f32p, _c_i64, _c_i64, _c_f32p, _c_vp, ctypes.POINTER(ctypes.c_void_p),
                                           _c_int, _c_int, _c_int, ctypes.POINTER(ctypes.c_void_p),
                                           ctypes.POINTER(ctypes.c_float), _c_int, ctypes.c_float, _c_int, _c_vp, _c_vp, _c_vp,
                                           _c_sz, _c_vp, _c_vp]),
    "nplda_train_step_grad_rows_f32": (_c_int, [_c_f32p, _c_i64, _c_i64, _c_vp, _c_vp, _c_i64, _c_f32p, _c_vp,
                                                ctypes.POINTER(ctypes.c_void_p), _c_int, _c_int, _c_int,
                                                ctypes.POINTER(ctypes.c_void_p), ctypes.POINTER(ctypes.c_float), _c_int,
                                                ctypes.c_float, _c_int, _c_vp, _c_vp, _c_vp, _c_sz, _c_vp, _c_vp]),
    "nplda_train_step_grad_dx_f32": (_c_int, [_c_vp, _c_vp, _c_i64, _c_i64, _c_int, _c_f32p, _c_vp,
                                              ctypes.POINTER(ctypes.c_void_p), _c_int, _c_int, _c_int,
                                              ctypes.POINTER(ctypes.c_void_p), ctypes.POINTER(ctypes.c_float), _c_int,
                                              ctypes.c_float, _c_int, _c_vp, _c_vp, _c_vp, _c_sz, _c_vp, _c_vp, _c_vp, _c_i64,
                                              _c_vp]),
    "nplda_train_step_apply_f32": (_c_int, [_c_f32p, ctypes.POINTER(ctypes.c_void_p), _c_int, _c_int, _c_int,
                                            ctypes.POINTER(ctypes.c_void_p), ctypes.POINTER(ctypes.c_float), _c_int,
                                            ctypes.c_float, _c_int, _c_vp, _c_vp, _c_vp] + [ctypes.c_float] * 5 +
                                   [_c_vp, _c_vp, _c_vp, _c_vp]),
    "nplda_train_step_dx_workspace_bytes": (_c_sz, [_c_i64, _c_int, _c_int, _c_int, _c_int]),
    "nplda_train_step_dx_f32": (_c_int, [_c_vp, _c_vp, _c_i64, _c_i64, _c_int, _c_f32p, ctypes.POINTER(ctypes.c_void_p), _c_int,
                                         _c_int, _c_int, ctypes.POINTER(ctypes.c_void_p), ctypes.POINTER(ctypes.c_float),
                                         _c_int, ctypes.c_float, _c_int, _c_vp, _c_vp, _c_vp] + [ctypes.c_float] * 5 +
                                [_c_vp, _c_vp, _c_sz, _c_vp, _c_vp, _c_vp, _c_vp, _c_vp, _c_i64, _c_vp]),
    "nplda_train_step_rows_workspace_bytes": (_c_sz, [_c_i64, _c_int, _c_int, _c_int]),
    "nplda_train_step_rows_f32": (_c_int, [_c_f32p, _c_i64, _c_i64, _c_vp, _c_vp, _c_i64, _c_f32p,
                                           ctypes.POINTER(ctypes.c_void_p), _c_int, _c_int, _c_int,
                                           ctypes.POINTER(ctypes.c_void_p), ctypes.POINTER(ctypes.c_float), _c_int,
                                           ctypes.c_float, _c_int, _c_vp, _c_vp, _c_vp] + [ctypes.c_float] * 5 +
                                  [_c_vp, _c_vp, _c_sz, _c_vp, _c_vp, _c_vp, _c_vp]),
    "nplda_train_step_records_f32": (_c_int, [_c_f32p, _c_i64, _c_i64, _c_vp, _c_vp, _c_i64,
                                              ctypes.POINTER(ctypes.c_void_p), _c_int, _c_int, _c_int,
                                              ctypes.POINTER(ctypes.c_void_p), ctypes.POINTER(ctypes.c_float), _c_int,
                                              ctypes.c_float, _c_int, _c_vp, _c_vp, _c_vp] + [ctypes.c_float] * 5 +
                                     [_c_vp, _c_vp, _c_sz, _c_vp, _c_vp, _c_vp, _c_vp]),
    "nplda_bf16x3_packed_bytes": (_c_sz, [_c_int, _c_int, _c_int]),
    "nplda_pack_params_bf16x3": (_c_int, [_c_f32p] * 6 + [_c_int] * 3 + [_c_vp, _c_sz, _c_vp]),
    "nplda_score_pairs_bf16x3": (_c_int, [_c_f32p, _c_f32p, _c_i64, _c_i64, _c_vp, _c_int, _c_int, _c_int, _c_f32p,
                                          _c_vp]),
    "nplda_embed_bf16x3": (_c_int, [_c_f32p, _c_i64, _c_i64, _c_vp, _c_int, _c_int, _c_int, _c_f32p, _c_i64, _c_f32p,
                                    _c_vp]),
    "nplda_grad_floats": (_c_sz, [_c_int, _c_int, _c_int]),
    "nplda_backward_workspace_bytes": (_c_sz, [_c_i64, _c_int, _c_int, _c_int]),
    "nplda_backward_f32": (_c_int, [_c_f32p, _c_f32p, _c_i64, _c_i64, _c_vp, _c_int, _c_int, _c_int, _c_f32p,
                                    _c_f32p, _c_f32p, _c_f32p, _c_i64, _c_f32p, _c_vp, _c_sz, _c_f32p, _c_vp]),
    "nplda_backward_ex_workspace_bytes": (_c_sz, [_c_i64, _c_int, _c_int, _c_int, _c_int]),
    "nplda_backward_ex_f32": (_c_int, [_c_f32p, _c_f32p, _c_i64, _c_i64, _c_vp, _c_int, _c_int, _c_int, _c_f32p,
                                       _c_f32p, _c_f32p, _c_f32p, _c_i64, _c_f32p, _c_vp, _c_sz, _c_f32p, _c_f32p,
                                       _c_f32p, _c_i64, _c_vp]),
    "nplda_embed_train_f32": (_c_int, [_c_f32p, _c_i64, _c_i64, _c_vp, _c_int, _c_int, _c_int, _c_f32p, _c_f32p,
                                       _c_f32p, _c_i64, _c_vp]),
    "nplda_embed_backward_f32": (_c_int, [_c_f32p, _c_i64, _c_i64, _c_vp, _c_int, _c_int, _c_int, _c_f32p, _c_i64,
                                          _c_f32p, _c_f32p, _c_i64, _c_vp, _c_sz, _c_f32p, _c_f32p, _c_i64, _c_vp]),
    "nplda_score_embeddings_bwd_workspace_bytes": (_c_sz, [_c_i64, _c_int]),
    "nplda_score_embeddings_bwd_f32": (_c_int, [_c_f32p, _c_i64, _c_f32p, _c_i64, _c_i64, _c_int, _c_f32p, _c_f32p,
                                                _c_f32p, _c_f32p, _c_i64, _c_f32p, _c_i64, _c_f32p, _c_f32p, _c_vp,
                                                _c_sz, _c_vp]),
    "nplda_matrix_frag_bytes": (_c_sz, [_c_int, _c_int]),
    "nplda_pack_matrix_f32": (_c_int, [_c_f32p, _c_i64, _c_int, _c_int, _c_int, _c_vp, _c_sz, _c_vp]),
    "nplda_dplda_quadform_f32": (_c_int, [_c_f32p, _c_int, _c_vp, _c_sz, _c_f32p, _c_vp]),
    "nplda_dplda_grad_f32": (_c_int, [_c_f32p, _c_i64, _c_i64, _c_int, _c_f32p, _c_f32p, _c_f32p, _c_vp, _c_sz, _c_vp]),
    "nplda_dplda_update_f32": (_c_int, [_c_f32p, _c_i64, _c_i64, _c_int, _c_f32p, _c_f32p, _c_f32p, _c_f32p, _c_f32p,
                                        ctypes.POINTER(ctypes.c_void_p), _c_f32p, _c_int, _c_vp] + [ctypes.c_float] * 5 +
                               [_c_vp, _c_int, _c_f32p, _c_f32p, _c_vp, _c_vp, _c_sz, _c_vp]),
    "nplda_dplda_update_loss_f32": (_c_int, [_c_f32p, _c_i64, _c_i64, _c_int, _c_f32p, _c_f32p, _c_int,
                                             ctypes.POINTER(ctypes.c_void_p), ctypes.POINTER(ctypes.c_float), _c_int, ctypes.c_float,
                                             _c_vp, _c_f32p, _c_f32p, _c_f32p, _c_f32p, _c_f32p, _c_f32p, _c_f32p,
                                             ctypes.POINTER(ctypes.c_void_p), _c_int, _c_vp] + [ctypes.c_float] * 5 +
                                    [_c_vp, _c_int, _c_f32p, _c_vp, _c_vp, _c_sz, _c_vp]),
    "nplda_rows_matmul_f32": (_c_int, [_c_f32p, _c_i64, _c_i64, _c_int, _c_vp, _c_int, _c_f32p, _c_f32p, _c_f32p,
                                       _c_i64, _c_vp]),
    "nplda_normalize_bwd_paired_f32": (_c_int, [_c_f32p, _c_i64, _c_f32p, _c_i64, _c_f32p, _c_i64, _c_int, _c_f32p,
                                                _c_i64, _c_vp]),
    "nplda_lda_wgrad_workspace_bytes": (_c_sz, [_c_i64, _c_int, _c_int]),
    "nplda_lda_wgrad_f32": (_c_int, [_c_f32p, _c_f32p, _c_i64, _c_i64, _c_f32p, _c_i64, _c_int, _c_int, _c_vp, _c_sz,
                                     _c_f32p, _c_vp]),
    "nplda_lda_dgrad_workspace_bytes": (_c_sz, [_c_int, _c_int]),
    "nplda_lda_dgrad_f32": (_c_int, [_c_f32p, _c_i64, _c_i64, _c_f32p, _c_int, _c_int, _c_vp, _c_sz, _c_f32p, _c_f32p,
                                     _c_i64, _c_vp]),
    "nplda_gather_pairs_mapped_f32": (_c_int, [_c_f32p, _c_i64, _c_i64, _c_vp, _c_i64, _c_vp, _c_vp, _c_i64, _c_int, _c_f32p,
                                               _c_f32p, _c_i64, _c_vp, _c_vp]),
    "nplda_embed_pair_f32": (_c_int, [_c_f32p, _c_i64, _c_f32p, _c_i64, _c_i64, _c_vp, _c_int, _c_int, _c_int, _c_f32p, _c_i64,
                                      _c_f32p, _c_vp]),
    "gb_score_pairs_ex_f32": (_c_int, [_c_f32p, _c_f32p, _c_i64, _c_i64, _c_vp, _c_int, _c_int, _c_f32p, _c_f32p,
                                       _c_f32p, _c_vp]),
}

_lib = None


class NpldaHipError(RuntimeError):
    pass


def load():
    """Load (once) and return the ctypes handle. Import torch first so that the HIP runtime
    already mapped by torch (same SONAME libamdhip64.so.7) is the one the library binds to."""
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(LIB_PATH):
        raise NpldaHipError(
            f"{LIB_PATH} is missing: build it with `python -m neuralplda_amd.build` "
            "(or __graft_entry__.build()). There is no CPU fallback.")
    try:
        import torch  # noqa: F401  (maps torch's HIP runtime before ours is resolved)
    except ImportError:
        pass
    lib = ctypes.CDLL(LIB_PATH)
    for name, (res, args) in SIGNATURES.items():
        fn = getattr(lib, name)  # AttributeError if the symbol is not exported
        fn.restype = res
        fn.argtypes = args
    _lib = lib
    return lib


def build_info():
    """{abi_version, csrc_sha (what the loaded library was built from), tree_sha (the sources next to it, None if absent),
    stale}: a library that travelled to a box without hipcc may be older than the tree it sits in."""
    from . import build as nbuild
    lib = load()
    sha = lib.nplda_build_id()
    sha = sha.decode() if sha else "unknown"
    tree = nbuild.source_sha()
    return {"abi_version": int(lib.nplda_abi_version()), "csrc_sha": sha, "tree_sha": tree,
            "stale": bool(tree is not None and tree != sha)}


NPLDA_EUNSUPPORTED = -95  # include/nplda_hip.h


def check(code, what):
    if code != 0:
        msg = load().nplda_strerror(code)
        raise NpldaHipError(f"{what} failed with status {code}: {msg.decode() if msg else '?'}")


def ptr(t):
    """Device (or host) address of a torch tensor's first element; None -> NULL."""
    return None if t is None else t.data_ptr()


def current_stream(device=None):
    """Raw hipStream_t of torch's current stream on `device` (default: the current device).  Through the C entry points
    torch itself uses: torch.cuda.current_stream() builds a Stream object per call, ~4 us of a ~10 us launch path."""
    import torch
    if device is None:
        idx = torch._C._cuda_getDevice()
    elif isinstance(device, int):
        idx = device
    else:
        idx = device.index
        if idx is None:
            idx = torch._C._cuda_getDevice()
    return torch._C._cuda_getCurrentRawStream(idx)


class on_device:
    """`with on_device(dev):` — torch.cuda.device(dev) without its cost when `dev` is already the current device (the
    common case: one process per GPU); switches and restores otherwise."""
    __slots__ = ("idx", "prev")

    def __init__(self, device):
        import torch
        idx = device if isinstance(device, int) else device.index
        self.idx = torch._C._cuda_getDevice() if idx is None else idx
        self.prev = -1

    def __enter__(self):
        import torch
        cur = torch._C._cuda_getDevice()
        if cur != self.idx:
            self.prev = cur
            torch.cuda.set_device(self.idx)
        return self

    def __exit__(self, *exc):
        if self.prev >= 0:
            import torch
            torch.cuda.set_device(self.prev)
            self.prev = -1
        return False

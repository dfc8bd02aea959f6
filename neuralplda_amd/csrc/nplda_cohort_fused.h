// nplda_cohort_fused.h — host interface of the fused cohort-statistics path (nplda_cohort_fused.hip) for the dispatch
// in nplda_cohort.hip.
#pragma once
#include "nplda_common.h"

namespace nplda {

struct FusedPlan {
    bool eligible;
    int nbands, nx, nsub;             // column bands (a multiple of 8), column tiles, candidate sub-lists per row
    int q, ksub;                      // list bands per column band (a band's columns in q parts, each with its own
                                      // sub-lists), slots per sub-list
    int cap;                          // candidate keys one row may bring to the select kernel (its LDS run)
    float zhi, fhi;                   // proposed candidate fraction fhi and its normal quantile zhi = Phi^-1(fhi)
    size_t fixed_bytes, row_bytes;    // workspace: fixed part and per row (rows are planned in multiples of 128)
    long long max_rows;               // rows one launch may cover (32-bit list offsets)
};

FusedPlan cohort_fused_plan(long long M, int topn, int Mp);
long long cohort_fused_resident_blocks();
// prepared: the fixed part of a workspace in which cohort_fused_prepare has left the cohort's moments (a CohortState): the
// pre-pass is skipped and the covariance image / moment vectors are read from there.
int cohort_fused_run(const FusedPlan& p, const float* z_rows, const float* q_rows, long long R, const float* z_coh,
                     const float* q_coh, long long M, long long ldz, const float* P, int ksteps, int topn, int lowest,
                     double* stats, unsigned char* ws, long long rows_cap, bool prepass, unsigned** fail_rows_out,
                     unsigned** nfail_out, long long resident, hipStream_t st, const unsigned char* prepared = nullptr,
                     int D2 = 0);  // D2: the embedding dimension (0: unknown — the last k-block runs all four k4-steps)
// The cohort-only part of the fused path (Gram matrix + first moments, covariance image folded with P) into `state`
// (plan.fixed_bytes bytes, 256-byte aligned): what every call on the same (model, cohort) would recompute.
int cohort_fused_prepare(const FusedPlan& p, const float* z_coh, const float* q_coh, long long M, long long ldz, const float* P,
                         int ksteps, unsigned char* state, hipStream_t st);

}  // namespace nplda

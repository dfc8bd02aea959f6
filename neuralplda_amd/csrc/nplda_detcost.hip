// Detection-cost sweep of the validation loop on gfx950: NeuralPlda.minc (utils/models.py:406-436 with arr2val,
// :23-27), the exact minimum detection cost, and the equal error rate — SURVEY.md §8 f1.
// The reference walks every target score in Python and counts with torch.where + .item(): O(N_tgt * N) host work that
// dominates validate() (xvector_NeuralPlda_pytorch.py:56-83).  Here: one device radix sort of (score, label) pairs
// (rocPRIM — the plain-library part, like a library GEMM), one prefix scan of the label counts, then ONE sweep kernel
// in which every sorted position is a candidate threshold: a binary search finds the start of its tie run, the two
// prefix counts there give (#targets below, #non-targets at-or-above), the cost for every beta is formed and an
// (value, index) arg-min is reduced wave -> block -> grid in a fixed order (ties -> lowest index, the first
// occurrence a CPU torch.min returns).  HBM-bound: ~N * (sort passes + 24 B) bytes; 2^20 scores in ~100 us.
#include <hip/hip_runtime.h>

#include <cstring>

#include <rocprim/device/device_radix_sort.hpp>
#include <rocprim/device/device_scan.hpp>

#include "nplda_common.h"

namespace {

constexpr int kMaxBeta = 8;
constexpr int kSweepBlocks = 512;

struct Best {
    double v;
    long long i;
};
__device__ __forceinline__ Best better(Best a, Best b) {
    return (b.v < a.v || (b.v == a.v && b.i < a.i)) ? b : a;
}

// label code: 1 target (t > 0.5), 0 non-target (t < 0.5), excluded otherwise (utils/models.py:407-408);
// packed as (target << 32) | non-target so that ONE 64-bit prefix sum carries both counts
__global__ __launch_bounds__(256) void label_kernel(const float* __restrict__ t, long long n,
                                                    unsigned long long* __restrict__ lab) {
    const long long i = (long long)blockIdx.x * 256 + threadIdx.x;
    if (i >= n) return;
    const float v = t[i];
    lab[i] = v > 0.5f ? (1ull << 32) : (v < 0.5f ? 1ull : 0ull);
}

struct SweepArgs {
    const float* key;                  // sorted scores
    const unsigned long long* lab;     // sorted packed labels
    const unsigned long long* pref;    // exclusive prefix sums of lab
    long long n;
    int K, exact;
    float beta[kMaxBeta];
    Best* part;                        // [kSweepBlocks][K + 1]  (slot K: first threshold with P_miss >= P_fa)
};

__device__ __forceinline__ long long lower_bound(const float* __restrict__ key, long long n, float v) {
    long long lo = 0, hi = n;
    while (lo < hi) {
        const long long mid = (lo + hi) >> 1;
        if (key[mid] < v) lo = mid + 1;
        else hi = mid;
    }
    return lo;
}

__global__ __launch_bounds__(256) void sweep_kernel(const SweepArgs a) {
    __shared__ Best red[4][kMaxBeta + 1];
    const long long n = a.n;
    const unsigned long long tot = n ? a.pref[n - 1] + a.lab[n - 1] : 0ull;
    const long long Nt = (long long)(tot >> 32), Nn = (long long)(tot & 0xffffffffull);
    Best best[kMaxBeta + 1];
#pragma unroll
    for (int k = 0; k <= kMaxBeta; ++k) best[k] = Best{INFINITY, 0x7fffffffffffffffll};
    const float fnt = (float)Nt, fnn = (float)Nn;
    for (long long p = (long long)blockIdx.x * 256 + threadIdx.x; p < n; p += (long long)gridDim.x * 256) {
        const float v = a.key[p];
        if (!a.exact) {
            // thresholds = target scores only; counts with the arr2val quirk: "count" = last index of torch.where
            // (count - 1), 1.0 when the set is empty; float32 arithmetic in the reference's order (no fma)
            if (!(a.lab[p] >> 32)) continue;
            const long long lb = lower_bound(a.key, n, v);
            const unsigned long long pr = a.pref[lb];
            const long long c_lt = (long long)(pr >> 32);
            const long long c_ge = Nn - (long long)(pr & 0xffffffffull);
            const float pm = __fdiv_rn(c_lt > 0 ? (float)(c_lt - 1) : 1.0f, fnt);
            const float pf = __fdiv_rn(c_ge > 0 ? (float)(c_ge - 1) : 1.0f, fnn);
            const long long rank = (long long)(a.pref[p] >> 32);  // index in the sorted target list
#pragma unroll
            for (int k = 0; k < kMaxBeta; ++k)
                if (k < a.K) best[k] = better(best[k], Best{(double)__fadd_rn(pm, __fmul_rn(a.beta[k], pf)), rank});
        } else {
            // every distinct score is a threshold ("target" iff s >= th); +inf is added by the final kernel
            if (p > 0 && a.key[p - 1] == v) continue;  // not the start of its tie run
            const unsigned long long pr = a.pref[p];
            const double pm = (double)(long long)(pr >> 32) / (double)(Nt > 0 ? Nt : 1);
            const double pf = (double)(Nn - (long long)(pr & 0xffffffffull)) / (double)(Nn > 0 ? Nn : 1);
#pragma unroll
            for (int k = 0; k < kMaxBeta; ++k)
                if (k < a.K) best[k] = better(best[k], Best{pm + (double)a.beta[k] * pf, p});
            if (pm - pf >= 0.0) best[kMaxBeta] = better(best[kMaxBeta], Best{0.0, p});  // first crossing
        }
    }
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
#pragma unroll
    for (int k = 0; k <= kMaxBeta; ++k) {
        Best b = best[k];
        for (int off = 32; off > 0; off >>= 1) {
            Best o;
            o.v = __shfl_xor(b.v, off);
            o.i = __shfl_xor(b.i, off);
            b = better(b, o);
        }
        if (lane == 0) red[wave][k] = b;
    }
    __syncthreads();
    if (threadIdx.x <= kMaxBeta) {
        const int k = threadIdx.x;
        const Best b = better(better(red[0][k], red[1][k]), better(red[2][k], red[3][k]));
        if (k < a.K) a.part[(size_t)blockIdx.x * (kMaxBeta + 1) + k] = b;
        if (k == kMaxBeta) a.part[(size_t)blockIdx.x * (kMaxBeta + 1) + kMaxBeta] = b;
    }
}

struct FinalArgs {
    const float* key;
    const unsigned long long* lab;
    const unsigned long long* pref;
    long long n;
    int K, exact, nblocks;
    const Best* part;
    float* minc;      // [K]
    float* thr;       // [K]
    float* minc_avg;  // [1]
    float* eer;       // [1] or NULL (exact mode only)
};

// wave k reduces slot k of the block partials (slot kMaxBeta = the EER crossing); then one lane per slot finishes
__global__ __launch_bounds__(64 * (kMaxBeta + 1)) void final_kernel(const FinalArgs a) {
    const int lane = threadIdx.x & 63, slot = threadIdx.x >> 6;
    const long long n = a.n;
    const unsigned long long tot = n ? a.pref[n - 1] + a.lab[n - 1] : 0ull;
    const long long Nt = (long long)(tot >> 32), Nn = (long long)(tot & 0xffffffffull);
    __shared__ double mv[kMaxBeta];
    __shared__ Best cross;
    Best b = Best{INFINITY, 0x7fffffffffffffffll};
    for (int j = lane; j < a.nblocks; j += 64) b = better(b, a.part[(size_t)j * (kMaxBeta + 1) + slot]);
    for (int off = 32; off > 0; off >>= 1) {
        Best o;
        o.v = __shfl_xor(b.v, off);
        o.i = __shfl_xor(b.i, off);
        b = better(b, o);
    }
    if (slot == kMaxBeta && lane == 0) cross = b;
    const int k = slot;
    if (lane == 0 && k < a.K) {
        float th;
        if (!a.exact) {
            // the winner is a rank in the sorted TARGET list: recover its score = the (rank+1)-th target in order
            // (binary search on the prefix counts)
            if (b.i == 0x7fffffffffffffffll) {  // no targets: torch.min over an empty tensor raises in the reference
                th = NAN;
                b.v = NAN;
            } else {
                long long lo = 0, hi = n;  // first p with (#targets in [0, p]) > rank
                while (lo < hi) {
                    const long long mid = (lo + hi) >> 1;
                    const long long inc = (long long)((a.pref[mid] + a.lab[mid]) >> 32);
                    if (inc <= b.i) lo = mid + 1;
                    else hi = mid;
                }
                th = a.key[lo];
            }
        } else {
            // the +inf threshold: P_miss = Nt / max(Nt, 1), P_fa = 0; it sorts after every score
            const double cinf = (double)Nt / (double)(Nt > 0 ? Nt : 1);
            if (b.i == 0x7fffffffffffffffll || cinf < b.v) { b.v = cinf; th = INFINITY; }
            else th = a.key[b.i];
        }
        mv[k] = b.v;
        a.minc[k] = (float)b.v;
        a.thr[k] = th;
    }
    __syncthreads();
    if (threadIdx.x == 0) {
        if (!a.exact) {  // sum(mincs) / len(mincs) on float32 tensors
            float s = (float)mv[0];
            for (int j = 1; j < a.K; ++j) s = __fadd_rn(s, (float)mv[j]);
            a.minc_avg[0] = __fdiv_rn(s, (float)a.K);
        } else {
            double s = 0.0;
            for (int j = 0; j < a.K; ++j) s += mv[j];
            a.minc_avg[0] = (float)(s / a.K);
        }
        if (a.eer) {
            float e = NAN;
            if (a.exact && n > 0 && Nt > 0 && Nn > 0) {
                Best c = cross;
                auto rates = [&](long long p, double& pm, double& pf) {
                    const unsigned long long pr = a.pref[p];
                    pm = (double)(long long)(pr >> 32) / (double)Nt;
                    pf = (double)(Nn - (long long)(pr & 0xffffffffull)) / (double)Nn;
                };
                if (c.i == 0x7fffffffffffffffll) c.i = 0;  // no crossing: argmax of an all-false mask is 0 in the torch form
                double pm1, pf1;
                rates(c.i, pm1, pf1);
                if (c.i == 0) {
                    e = (float)((pm1 + pf1) / 2);
                } else {
                    const long long q = lower_bound(a.key, n, a.key[c.i - 1]);  // previous distinct threshold
                    double pm0, pf0;
                    rates(q, pm0, pf0);
                    const double x0 = pm0 - pf0, x1 = pm1 - pf1;
                    const double w = (x1 - x0) != 0.0 ? -x0 / (x1 - x0) : 0.5;
                    e = (float)(pm0 + w * (pm1 - pm0));
                }
            }
            a.eer[0] = e;
        }
    }
}

struct Plan {
    size_t o_keys, o_lab_in, o_lab, o_pref, o_part, o_tmp, tmp_bytes, total;
};

int make_plan(long long n, Plan* p) {
    const size_t nn = (size_t)(n > 0 ? n : 1);
    size_t sort_bytes = 0, scan_bytes = 0;
    if (hipError_t e = rocprim::radix_sort_pairs(nullptr, sort_bytes, (const float*)nullptr, (float*)nullptr,
                                                 (const unsigned long long*)nullptr, (unsigned long long*)nullptr,
                                                 nn, 0, 32, (hipStream_t)0))
        return (int)e;
    if (hipError_t e = rocprim::exclusive_scan(nullptr, scan_bytes, (const unsigned long long*)nullptr,
                                               (unsigned long long*)nullptr, 0ull, nn,
                                               rocprim::plus<unsigned long long>(), (hipStream_t)0))
        return (int)e;
    auto up = [](size_t b) { return (b + 255) / 256 * 256; };
    size_t o = 0;
    p->o_keys = o; o += up(nn * sizeof(float));
    p->o_lab_in = o; o += up(nn * sizeof(unsigned long long));
    p->o_lab = o; o += up(nn * sizeof(unsigned long long));
    p->o_pref = o; o += up(nn * sizeof(unsigned long long));
    p->o_part = o; o += up((size_t)kSweepBlocks * (kMaxBeta + 1) * sizeof(Best));
    p->tmp_bytes = sort_bytes > scan_bytes ? sort_bytes : scan_bytes;
    p->o_tmp = o; o += up(p->tmp_bytes);
    p->total = o;
    return 0;
}

}  // namespace

extern "C" {

size_t nplda_detcost_workspace_bytes(int64_t N) {
    if (N < 0 || N > 0x7fffffffll) return 0;
    Plan p;
    if (make_plan(N, &p)) return 0;
    return p.total;
}

int nplda_detcost_sweep_f32(const float* scores, const float* target, int64_t N, const float* betas, int K, int exact,
                            float* minc, float* thr, float* minc_avg, float* eer, void* workspace,
                            size_t workspace_bytes, nplda_stream_t stream) {
    if (N < 0 || K < 1 || !betas || !minc || !thr || !minc_avg) return NPLDA_EINVAL;
    if (K > kMaxBeta || N > 0x7fffffffll) return NPLDA_EUNSUPPORTED;
    if (N > 0 && (!scores || !target)) return NPLDA_EINVAL;
    if (!workspace || !nplda_aligned16(workspace)) return NPLDA_EINVAL;
    Plan p;
    if (int rc = make_plan(N, &p)) return rc;
    if (workspace_bytes < p.total) return NPLDA_ENOSPC;
    hipStream_t st = (hipStream_t)stream;
    char* ws = (char*)workspace;
    float* keys = (float*)(ws + p.o_keys);
    unsigned long long* lab_in = (unsigned long long*)(ws + p.o_lab_in);
    unsigned long long* lab = (unsigned long long*)(ws + p.o_lab);
    unsigned long long* pref = (unsigned long long*)(ws + p.o_pref);
    Best* part = (Best*)(ws + p.o_part);
    int nblocks = 0;
    if (N > 0) {
        hipLaunchKernelGGL(label_kernel, dim3((unsigned)((N + 255) / 256)), dim3(256), 0, st, target, (long long)N, lab_in);
        if (int rc = nplda_launch_status()) return rc;
        size_t tb = p.tmp_bytes;
        if (hipError_t e = rocprim::radix_sort_pairs(ws + p.o_tmp, tb, scores, keys, (const unsigned long long*)lab_in,
                                                     lab, (size_t)N, 0, 32, st))
            return (int)e;
        tb = p.tmp_bytes;
        if (hipError_t e = rocprim::exclusive_scan(ws + p.o_tmp, tb, (const unsigned long long*)lab, pref, 0ull,
                                                   (size_t)N, rocprim::plus<unsigned long long>(), st))
            return (int)e;
        nblocks = (int)((N + 255) / 256 < kSweepBlocks ? (N + 255) / 256 : kSweepBlocks);
        SweepArgs a;
        a.key = keys; a.lab = lab; a.pref = pref; a.n = N; a.K = K; a.exact = exact ? 1 : 0; a.part = part;
        for (int k = 0; k < kMaxBeta; ++k) a.beta[k] = k < K ? betas[k] : 0.f;
        hipLaunchKernelGGL(sweep_kernel, dim3(nblocks), dim3(256), 0, st, a);
        if (int rc = nplda_launch_status()) return rc;
    }
    FinalArgs f;
    f.key = keys; f.lab = lab; f.pref = pref; f.n = N; f.K = K; f.exact = exact ? 1 : 0; f.nblocks = nblocks;
    f.part = part; f.minc = minc; f.thr = thr; f.minc_avg = minc_avg; f.eer = eer;
    hipLaunchKernelGGL(final_kernel, dim3(1), dim3(64 * (kMaxBeta + 1)), 0, st, f);
    return nplda_launch_status();
}

}  // extern "C"

// nplda_optim.hip — one-launch Adam over all NPLDA parameter tensors (gfx950).
//
// The reference drives torch.optim.Adam(model.parameters(), lr, weight_decay=1e-5)
// (xvector_NeuralPlda_pytorch.py:139): on the device that is ~10 multi-tensor launches of ~10 us each for
// 117 k parameters, more than the whole forward + backward of a 4096-pair minibatch.  This kernel applies the
// same update (torch's Adam: L2 weight decay folded into the gradient, bias-corrected moments, eps outside the
// square root, no amsgrad) to up to 12 (param, grad, m, v) segments in ONE launch, reading the gradients where
// nplda_backward_f32 / nplda_loss_finish_f32 left them.  The step counter lives on the device so that the
// whole optimisation step can be replayed from a HIP graph.
#include "nplda_adam_math.h"

namespace {

constexpr int kMaxSeg = 12;

struct AdamSeg { float* p; const float* g; float* m; float* v; long long n; };
struct AdamArgs {
    AdamSeg seg[kMaxSeg];
    int nseg;
    long long total;
    float* step;  // [0] steps taken so far (float), [1] arrival ticket of the running launch
    float lr, beta1, beta2, eps, wd;
};

// step[0]: steps taken so far (float, as torch keeps it); step[1]: arrival ticket of the running launch (uint bits, 0
// between launches).  Every block reads step[0] on entry and works with t = step[0] + 1; the LAST block to finish bumps
// step[0] and clears the ticket — by then every block has read the old value, so the increment needs no launch of its
// own (it was a 1-thread kernel: 4.5 us of a 109 us training step) and the pair stays graph-replay safe.
__global__ __launch_bounds__(256) void adam_kernel(const AdamArgs a) {
    const float t = a.step[0] + 1.0f;
    const nplda_adam::Consts c = nplda_adam::consts_for(t, a.lr, a.beta1, a.beta2, a.eps, a.wd);
    const long long stride = (long long)gridDim.x * 256;
    for (long long i = (long long)blockIdx.x * 256 + threadIdx.x; i < a.total; i += stride) {
        long long r = i;
        int s = 0;
#pragma unroll 1
        for (; s < a.nseg - 1 && r >= a.seg[s].n; ++s) r -= a.seg[s].n;
        const AdamSeg& sg = a.seg[s];
        float m = sg.m[r], v = sg.v[r];
        sg.p[r] = nplda_adam::update(sg.p[r], sg.g[r], m, v, c);
        sg.m[r] = m;
        sg.v[r] = v;
    }
    __syncthreads();  // the whole block has read step[0]
    if (threadIdx.x == 0) {
        unsigned* ticket = reinterpret_cast<unsigned*>(a.step + 1);
        // no fence: nothing is handed over inside the launch — the kernel boundary publishes step[0] to the next one
        if (atomicAdd(ticket, 1u) == gridDim.x - 1) {
            a.step[0] = t;
            *ticket = 0u;
        }
    }
}

}  // namespace

extern "C" {

int nplda_adam_step_f32(float* const* params, const float* const* grads, float* const* exp_avg,
                        float* const* exp_avg_sq, const int64_t* numel, int nseg, float* step, float lr, float beta1,
                        float beta2, float eps, float weight_decay, nplda_stream_t stream) {
    if (nseg < 1 || nseg > kMaxSeg || !params || !grads || !exp_avg || !exp_avg_sq || !numel || !step) return NPLDA_EINVAL;
    AdamArgs a = {};
    a.nseg = nseg;
    for (int i = 0; i < nseg; ++i) {
        if (!params[i] || !grads[i] || !exp_avg[i] || !exp_avg_sq[i] || numel[i] < 0) return NPLDA_EINVAL;
        a.seg[i] = AdamSeg{params[i], grads[i], exp_avg[i], exp_avg_sq[i], (long long)numel[i]};
        a.total += numel[i];
    }
    a.step = step; a.lr = lr; a.beta1 = beta1; a.beta2 = beta2; a.eps = eps; a.wd = weight_decay;
    hipStream_t st = (hipStream_t)stream;
    // four elements per thread: the arrival tickets are one atomic per block on one address (~10 ns each, serialised)
    long long blocks = (a.total + 1023) / 1024;  // total == 0: one block that only counts the step
    if (blocks < 1) blocks = 1;
    if (blocks > 2048) blocks = 2048;
    hipLaunchKernelGGL(adam_kernel, dim3((unsigned)blocks), dim3(256), 0, st, a);
    return nplda_launch_status();
}

}  // extern "C"

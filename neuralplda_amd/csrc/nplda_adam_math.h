// nplda_adam_math.h — torch.optim.Adam's element update (L2 weight decay folded into the gradient, bias-corrected
// moments, eps outside the square root, no amsgrad), shared by nplda_optim.hip and the fused training step.
#pragma once
#include "nplda_common.h"

namespace nplda_adam {

struct Consts { float beta1, beta2, eps, wd, step_size, inv_sqrt_bc2; };

// t = steps taken including this one
__device__ __forceinline__ Consts consts_for(float t, float lr, float beta1, float beta2, float eps, float wd) {
    const float bc1 = 1.0f - powf(beta1, t);
    const float bc2 = 1.0f - powf(beta2, t);
    return Consts{beta1, beta2, eps, wd, lr / bc1, 1.0f / sqrtf(bc2)};
}

// returns the updated parameter; m, v are updated in place.  Every contraction is spelled out: left to the compiler
// (fp-contract=fast) the two kernels this is inlined into fuse beta m + (1 - beta) g differently and their
// trajectories part in the last bit at the second step.
__device__ __forceinline__ float update(float p, float grad, float& m, float& v, const Consts& c) {
    const float g = fmaf(c.wd, p, grad);
    m = fmaf(c.beta1, m, (1.0f - c.beta1) * g);
    v = fmaf(c.beta2, v, ((1.0f - c.beta2) * g) * g);
    const float denom = fmaf(sqrtf(v), c.inv_sqrt_bc2, c.eps);
    return fmaf(-c.step_size, m / denom, p);
}

}  // namespace nplda_adam

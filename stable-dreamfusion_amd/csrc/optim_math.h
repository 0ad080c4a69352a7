// optim_math.h — arithmetic of csrc/optim.hip (GradScaler bookkeeping + the Adan update), SDFX_HD so that tests/hostmath
// builds the SAME source with g++ and checks it on the CPU against tests/golden/adan_ref.npz (six steps of the reference's
// own optimizer.Adan) and against torch.amp.GradScaler's documented semantics.
#pragma once

#include <math.h>
#include <stdint.h>

#include "sdfx_math.h"

namespace sdfx {
namespace optim {

// ctl layout (float words; integers are stored as their float value, exact far beyond any step count reached)
//  [0] loss scale S            (in/out)   [1] growth tracker     (in/out)   [2] applied steps k  (in/out)
//  [3] 1/S of this iteration   (out)      [4] clip factor        (out)      [5] skip (1 = overflow) (out)
//  [6] 1-b1^k  [7] 1-b2^k  [8] sqrt(1-b3^k)  (out)      [9] ||g|| unscaled (out)   [10] skipped iterations (in/out)
//
// sumsq = sum of squares of all (scaled) gradients of the iteration, nonfinite = number of workgroups that saw inf / nan.
SDFX_HD void adan_prepare(float* ctl, double sumsq, double nonfinite, float b1, float b2, float b3, float max_grad_norm,
                          float eps, float growth, float backoff, float growth_interval) {
    const float S = ctl[0];
    const float inv = 1.0f / S;
    const bool overflow = nonfinite != 0.0 || !(sumsq <= 1.7976931348623157e308);
    ctl[3] = inv;
    ctl[5] = overflow ? 1.0f : 0.0f;
    if (overflow) {  // GradScaler.update(): back off, restart the growth count (amp_update_scale)
        ctl[0] = S * backoff;
        ctl[1] = 0.0f;
        ctl[4] = 0.0f;
        ctl[9] = INFINITY;
        ctl[10] += 1.0f;
        return;
    }
    float tracker = ctl[1] + 1.0f;
    if (tracker >= growth_interval) {
        const float grown = S * growth;
        if (grown <= 3.402823466e38f) ctl[0] = grown;
        tracker = 0.0f;
    }
    ctl[1] = tracker;
    const float norm = (float)(sqrt(sumsq) * (double)inv);
    ctl[9] = norm;
    // optimizer.py:121-127: clip_global_grad_norm = clamp(max_grad_norm / (||g|| + eps), max = 1)
    ctl[4] = max_grad_norm > 0.f ? fminf(max_grad_norm / (norm + eps), 1.0f) : 1.0f;
    const float k = ctl[2] + 1.0f;
    ctl[2] = k;
    ctl[6] = 1.0f - powf(b1, k);          // optimizer.py:137-141
    ctl[7] = 1.0f - powf(b2, k);
    ctl[8] = sqrtf(1.0f - powf(b3, k));
}

struct AdanHyper {
    float lr, wd, eps, b1, b2, b3;
    int no_prox;
};

// one element of optimizer.py:216-261; gs is the SCALED gradient as backward left it, unscale = (1 / S) * clip
SDFX_HD void adan_one(float& p, float gs, float& m, float& v, float& nn, float& prev, float unscale, bool first, float bc1,
                      float bc2, float bc3, const AdanHyper& h) {
    const float g = gs * unscale;                       // unscale and clip (optimizer.py:154 grad.mul_(clip))
    // optimizer.py:145-148: the first time a parameter has a gradient, pre_grad := grad (difference 0). `prev` is
    // allocated as NaN, so this also holds for a tensor that joins later (e.g. the background MLP, unused while the
    // schedule draws random background colours); overflowed iterations never get here, so NaN is a safe "unset" mark
    const float diff = (first || prev != prev) ? 0.f : g - prev;
    m = m * h.b1 + (1.f - h.b1) * g;                    // exp_avg
    v = v * h.b2 + (1.f - h.b2) * diff;                 // exp_avg_diff
    const float u = g + h.b2 * diff;
    nn = nn * h.b3 + (1.f - h.b3) * u * u;              // exp_avg_sq
    const float denom = sqrtf(nn) / bc3 + h.eps;
    const float step1 = h.lr / bc1, step2 = h.lr * h.b2 / bc2;
    if (h.no_prox) {
        p = p * (1.f - h.lr * h.wd);
        p = p - step1 * (m / denom);
        p = p - step2 * (v / denom);
    } else {                                             // optimizer.py:246-249 (default)
        p = p - step1 * (m / denom);
        p = p - step2 * (v / denom);
        p = p / (1.f + h.lr * h.wd);
    }
    prev = g;
}

}  // namespace optim
}  // namespace sdfx

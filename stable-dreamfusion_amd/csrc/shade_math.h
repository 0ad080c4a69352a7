// shade_math.h — per-sample arithmetic of csrc/shade.hip (normal from the finite-difference stencil, shading, orientation
// term, and the hand-derived backward), SDFX_HD so that tests/hostmath can build the SAME source with g++ and check it on
// the CPU against tests/golden/shade_ref.npz, the output of the reference's own NeRFNetwork.forward + autograd.
#pragma once

#include <math.h>
#include <stddef.h>
#include <stdint.h>

#include "sdfx_math.h"

namespace sdfx {
namespace shade {

constexpr float kNormEps = 1e-20f;  // safe_normalize's clamp (nerf/utils.py:109-110)
constexpr float kFltMax = 3.402823466e38f;

enum { kLambertian = 1, kTextureless = 2, kNormal = 3 };

struct Vec3 {
    float x, y, z;
};

SDFX_HD float nan_to_num_(float v) {  // torch.nan_to_num defaults
    if (v != v) return 0.f;
    if (v > kFltMax) return kFltMax;
    if (v < -kFltMax) return -kFltMax;
    return v;
}

// light direction of ray n: safe_normalize(rays_o[n] + offset)  (nerf/renderer.py:727)
SDFX_HD Vec3 ray_light(const float* rays_o, const float* off, uint32_t n) {
    const float x = rays_o[n * 3 + 0] + off[0], y = rays_o[n * 3 + 1] + off[1], z = rays_o[n * 3 + 2] + off[2];
    const float s = sqrtf(fmaxf(x * x + y * y + z * z, kNormEps));
    return {x / s, y / s, z / s};
}

struct Sample {
    Vec3 raw;      // un-normalised normal
    float q, s;    // |raw|^2 and sqrt(max(q, eps))
    Vec3 y;        // raw / s before nan_to_num
    Vec3 n;        // the normal
    Vec3 d;        // normalised view direction
    float ndl, ndd;
};

// from the seven stencil densities (x, x+e_x, x-e_x, x+e_y, x-e_y, x+e_z, x-e_z) and the un-normalised view direction
SDFX_HD Sample make_sample(const float s[7], const float dir[3], float e, const Vec3& l) {
    Sample p;
    p.raw.x = -(0.5f * (s[1] - s[2]) / e);   // network_grid.py:90-96
    p.raw.y = -(0.5f * (s[3] - s[4]) / e);
    p.raw.z = -(0.5f * (s[5] - s[6]) / e);
    p.q = p.raw.x * p.raw.x + p.raw.y * p.raw.y + p.raw.z * p.raw.z;
    p.s = sqrtf(fmaxf(p.q, kNormEps));
    p.y = {p.raw.x / p.s, p.raw.y / p.s, p.raw.z / p.s};
    p.n = {nan_to_num_(p.y.x), nan_to_num_(p.y.y), nan_to_num_(p.y.z)};
    const float dx = dir[0], dy = dir[1], dz = dir[2];
    const float ds = sqrtf(fmaxf(dx * dx + dy * dy + dz * dz, kNormEps));
    p.d = {dx / ds, dy / ds, dz / ds};
    p.ndl = p.n.x * l.x + p.n.y * l.y + p.n.z * l.z;
    p.ndd = p.n.x * p.d.x + p.n.y * p.d.y + p.n.z * p.d.z;
    return p;
}

// sigma7: [7, cap] densities at x, x+e_x, x-e_x, x+e_y, x-e_y, x+e_z, x-e_z; dirs: [cap, 3] un-normalised
SDFX_HD Sample load_sample(const float* sigma7, const float* dirs, uint32_t cap, uint32_t i, float e, const Vec3& l) {
    float s[7], d[3];
    for (int k = 0; k < 7; k++) s[k] = sigma7[(size_t)k * cap + i];
    d[0] = dirs[(size_t)i * 3 + 0]; d[1] = dirs[(size_t)i * 3 + 1]; d[2] = dirs[(size_t)i * 3 + 2];
    return make_sample(s, d, e, l);
}

// color (network_grid.py:117-130) and the per-sample factor of loss_orient (renderer.py:745); albedo3 only read for mode 1
SDFX_HD void sample_forward(const Sample& p, float ratio, int mode, const float* albedo3, float color[3], float& orient) {
    const float lambert = ratio + (1.f - ratio) * fmaxf(p.ndl, 0.f);
    if (mode == kNormal) {
        color[0] = (p.n.x + 1.f) / 2.f; color[1] = (p.n.y + 1.f) / 2.f; color[2] = (p.n.z + 1.f) / 2.f;
    } else if (mode == kTextureless) {
        color[0] = color[1] = color[2] = lambert;
    } else {
        color[0] = albedo3[0] * lambert; color[1] = albedo3[1] * lambert; color[2] = albedo3[2] * lambert;
    }
    const float o = fmaxf(p.ndd, 0.f);
    orient = o * o;
}

// gradient of (color, orient[, normal]) into the six neighbour densities (dsig6 = d/d s(+x), s(-x), s(+y), s(-y), s(+z),
// s(-z)) and the albedo (mode 1), with torch.autograd's conventions for clamp / nan_to_num / the normaliser's clamp
SDFX_HD void sample_backward(const Sample& p, const Vec3& l, float ratio, int mode, const float* albedo3, const float g[3],
                             const float* dnormal3, float dorient, float e, float dsig6[6], float dalb[3]) {
    Vec3 dn = {0.f, 0.f, 0.f};  // gradient with respect to the (normalised, nan_to_num'ed) normal
    if (dnormal3) dn = {dnormal3[0], dnormal3[1], dnormal3[2]};
    const float lambert = ratio + (1.f - ratio) * fmaxf(p.ndl, 0.f);
    float dlambert = 0.f;
    dalb[0] = dalb[1] = dalb[2] = 0.f;
    if (mode == kNormal) {
        dn.x += g[0] / 2.f; dn.y += g[1] / 2.f; dn.z += g[2] / 2.f;
    } else if (mode == kTextureless) {
        dlambert = g[0] + g[1] + g[2];
    } else {
        dlambert = g[0] * albedo3[0] + g[1] * albedo3[1] + g[2] * albedo3[2];
        dalb[0] = g[0] * lambert; dalb[1] = g[1] * lambert; dalb[2] = g[2] * lambert;
    }
    if (p.ndl >= 0.f) {  // torch's clamp(min=0) backward passes the gradient where input >= bound (equality included)
        const float c = dlambert * (1.f - ratio);
        dn.x += c * l.x; dn.y += c * l.y; dn.z += c * l.z;
    }
    if (p.ndd > 0.f) {  // orient = clamp(n.d, 0)^2
        const float c = dorient * 2.f * p.ndd;
        dn.x += c * p.d.x; dn.y += c * p.d.y; dn.z += c * p.d.z;
    }
    // nan_to_num: no gradient through replaced entries
    const Vec3 dy = {(p.y.x == p.n.x) ? dn.x : 0.f, (p.y.y == p.n.y) ? dn.y : 0.f, (p.y.z == p.n.z) ? dn.z : 0.f};
    // y = raw / s, s = sqrt(clamp(q, eps)): draw = dy / s - raw (dy . raw) / s^3 [q >= eps]
    Vec3 dr = {dy.x / p.s, dy.y / p.s, dy.z / p.s};
    if (p.q >= kNormEps) {
        const float dot = dy.x * p.raw.x + dy.y * p.raw.y + dy.z * p.raw.z;
        const float c = dot / (p.s * p.s * p.s);
        dr.x -= p.raw.x * c; dr.y -= p.raw.y * c; dr.z -= p.raw.z * c;
    }
    const float h = 0.5f / e;  // raw_k = -(0.5 (s_pos - s_neg) / e)
    dsig6[0] = -h * dr.x; dsig6[1] = h * dr.x;
    dsig6[2] = -h * dr.y; dsig6[3] = h * dr.y;
    dsig6[4] = -h * dr.z; dsig6[5] = h * dr.z;
}

}  // namespace shade
}  // namespace sdfx

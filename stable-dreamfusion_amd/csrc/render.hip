// render.hip — everything between the field and the image of a training iteration in ONE kernel each way:
//
//   forward   normal from the 7-point density stencil -> shading -> view-direction normalisation   (csrc/shade_math.h:
//             network_grid.py:81-130, renderer.py:734) -> front-to-back compositing with the T < T_thresh stop
//             (raymarching.cu:500-590) -> per ray: weights_sum, depth, image, and the two regulariser sums
//             sum_i H(clamp(w_i)) (lambda_entropy, nerf/utils.py:571-575) and sum_i w_i clamp(n_i . d_i, 0)^2
//             (loss_orient, renderer.py:744-746)
//   backward  the reference's compositing backward (raymarching.cu:605-706, its grad_weights term included) -> gradient of
//             the shading and of the orientation term -> d sigma at the 7 stencil points and d albedo
//
// It replaces k_shade_forward, k_composite_train_fwd, k_entropy_forward and ~6 elementwise / reduction launches forward
// (k_composite_train_bwd, k_shade_backward, k_entropy_backward and ~8 launches backward). All of them map one wavefront to
// one ray already, so the fusion is a concatenation of their loop bodies: colour, sigma and the per-sample gradients never
// leave registers. Why it matters at 4096 rays: the compositor alone moves 12-18 MB — 2 us at the HBM roofline — and
// measured 12.7 / 35 us, i.e. launch + ramp + one dependent 64-sample chunk chain; the kernels around it each paid the same
// floor again. Per-sample traffic of the fused pair: forward 7*4 (sigma7) + 12 (albedo) + 12 (dirs) + 8 (ts) in, 4 (weights)
// out = 64 B; backward the same 60 B in + 7*4 + 12 out = 100 B.
//
// The shading mode and the ambient ratio are read from DEVICE memory (the per-iteration scalar block of the trainer), so
// one captured HIP graph serves 'lambertian', 'textureless' and 'normal' iterations.
//
// Arithmetic is shared source with the unfused kernels: shade_math.h (checked on the CPU against the reference's own
// NeRFNetwork.forward + autograd, tests/test_hostmath.py) and the scan formulation of raymarching.hip's compositor
// (checked against the oracle and the reference kernels). tests/test_gpu_render.py compares the fused pair with the
// oracle chain shade -> composite -> entropy, forward and backward.
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "sdfx.h"
#include "sdfx_common.h"
#include "shade_math.h"

using namespace sdfx;
using namespace sdfx::shade;

namespace {

constexpr uint32_t kThreads = 256;
constexpr uint32_t kPadBlocks = 64;
constexpr float kAlphaLo = 1e-5f, kAlphaHi = 1.f - 1e-5f;   // clamp of the entropy term

__device__ __forceinline__ float entropy_bits(float w) {
    const float a = fminf(fmaxf(w, kAlphaLo), kAlphaHi);
    return -a * log2f(a) - (1.f - a) * log2f(1.f - a);
}
// d H(clamp(w)) / d w: torch's clamp passes the gradient inside the closed interval
__device__ __forceinline__ float entropy_grad(float w) {
    return (w >= kAlphaLo && w <= kAlphaHi) ? (log2f(1.f - w) - log2f(w)) : 0.f;
}

struct RenderArgs {
    const float* sigma7;   // [7, cap]
    const float* albedo;   // [cap, 3] (rows of the centre points)
    const float* dirs;     // [cap, 3] un-normalised
    const float* ts;       // [cap, 2] (t after the step, dt)
    const int32_t* rays;   // [N, 2] (offset, count)
    const float* rays_o;   // [N, 3]
    const float* light_off;
    const float* ratio_p;  // device scalar
    const float* mode_p;   // device scalar (1, 2, 3 as float) or null -> `mode`
    int mode;
    float e, T_thresh;
    uint32_t cap, n_rays, ray_blocks;
    const int32_t* total_p;
};

__device__ __forceinline__ int shading_mode(const RenderArgs& a) { return a.mode_p ? (int)a.mode_p[0] : a.mode; }

// what one lane reads for one sample: loaded a chunk AHEAD of its use, so that the memory round trip of chunk k + 1 overlaps the
// scan of chunk k (a ray is a chain of dependent 64-sample chunks: tools/ubench/launch_floor.hip prices a round trip at ~1.2 us)
struct Raw {
    float s[7], alb[3], d[3], t, dt;
};
__device__ __forceinline__ Raw load_raw(const RenderArgs& a, uint32_t i, bool valid, bool want_albedo) {
    Raw r;
    if (valid) {
#pragma unroll
        for (int k = 0; k < 7; k++) r.s[k] = a.sigma7[(size_t)k * a.cap + i];
#pragma unroll
        for (int k = 0; k < 3; k++) r.d[k] = a.dirs[(size_t)i * 3 + k];
#pragma unroll
        for (int k = 0; k < 3; k++) r.alb[k] = want_albedo ? a.albedo[(size_t)i * 3 + k] : 0.f;
        const float2 tt = reinterpret_cast<const float2*>(a.ts)[i];
        r.t = tt.x; r.dt = tt.y;
    } else {
#pragma unroll
        for (int k = 0; k < 7; k++) r.s[k] = 0.f;
        r.alb[0] = r.alb[1] = r.alb[2] = 0.f; r.d[0] = r.d[1] = r.d[2] = 0.f; r.t = r.dt = 0.f;
    }
    return r;
}

__global__ __launch_bounds__(kThreads) void k_render_train_fwd(RenderArgs a, float* __restrict__ weights,
                                                                float* __restrict__ weights_sum, float* __restrict__ depth,
                                                                float* __restrict__ image, float* __restrict__ ray_sums) {
    if (blockIdx.x >= a.ray_blocks) {   // padding rows [total, cap) belong to no ray: zero weight
        const uint32_t total = (uint32_t)a.total_p[0];
        for (uint32_t i = total + (blockIdx.x - a.ray_blocks) * kThreads + threadIdx.x; i < a.cap; i += kPadBlocks * kThreads) weights[i] = 0.f;
        return;
    }
    const uint32_t n = (blockIdx.x * kThreads + threadIdx.x) >> 6;
    if (n >= a.n_rays) return;
    const int lane = lane_id();
    const uint32_t offset = (uint32_t)a.rays[n * 2], count = (uint32_t)a.rays[n * 2 + 1];
    if (count == 0 || offset + count > a.cap) {   // raymarching.cu:521-528 (the reference's weights are zero-initialised)
        for (uint32_t k = lane; k < count && offset + k < a.cap; k += kWave) weights[offset + k] = 0.f;
        if (lane == 0) {
            weights_sum[n] = 0; depth[n] = 0; image[n * 3 + 0] = 0; image[n * 3 + 1] = 0; image[n * 3 + 2] = 0;
            ray_sums[n * 2 + 0] = 0; ray_sums[n * 2 + 1] = 0;
        }
        return;
    }
    const int mode = shading_mode(a);
    const float ratio = a.ratio_p[0];
    const Vec3 l = ray_light(a.rays_o, a.light_off, n);

    float T_carry = 1.0f;
    float r = 0, g = 0, b = 0, ws = 0, d = 0, ent = 0, ori = 0;
    bool done = false;
    const bool lamb = mode == kLambertian;
    Raw nxt = load_raw(a, offset + (uint32_t)lane, (uint32_t)lane < count, lamb);
    for (uint32_t base = 0; base < count; base += kWave) {
        const uint32_t k = base + lane;
        const bool valid = k < count;
        const uint32_t i = offset + (valid ? k : 0);
        if (done) {   // past the transmittance cut: zero weights; they still count in the entropy mean
            if (valid) { weights[i] = 0.f; ent += entropy_bits(0.f); }
            continue;
        }
        const Raw cur = nxt;
        nxt = load_raw(a, i + kWave, k + kWave < count, lamb);            // the next chunk's loads are in flight during this scan
        float alpha = 0.f, t = 0.f, c[3] = {0.f, 0.f, 0.f}, o = 0.f;
        if (valid) {
            const Sample p = make_sample(cur.s, cur.d, a.e, l);
            sample_forward(p, ratio, mode, lamb ? cur.alb : nullptr, c, o);
            t = cur.t;
            alpha = 1.0f - __expf(-cur.s[0] * cur.dt);   // raymarching.cu:543 (binarize = false on the training path)
        }
        const float incl = wave_incl_prod(1.0f - alpha, lane);
        float excl = __shfl_up(incl, 1, kWave);
        if (lane == 0) excl = 1.0f;
        const float T_before = T_carry * excl, T_after = T_carry * incl;
        const unsigned long long cut = __ballot(valid && (T_after < a.T_thresh));
        const int first_cut = cut ? (int)__ffsll((long long)cut) - 1 : kWave;
        const float w = (valid && lane <= first_cut) ? alpha * T_before : 0.0f;
        if (valid) {
            weights[i] = w;
            ent += entropy_bits(w);
            ori += w * o;
        }
        r += w * c[0]; g += w * c[1]; b += w * c[2]; ws += w; d += w * t;
        T_carry = T_carry * __shfl(incl, kWave - 1, kWave);
        done = cut != 0ull;
    }
    r = wave_sum(r); g = wave_sum(g); b = wave_sum(b); ws = wave_sum(ws); d = wave_sum(d); ent = wave_sum(ent); ori = wave_sum(ori);
    if (lane == 0) {
        weights_sum[n] = ws; depth[n] = d;
        image[n * 3 + 0] = r; image[n * 3 + 1] = g; image[n * 3 + 2] = b;
        ray_sums[n * 2 + 0] = ent; ray_sums[n * 2 + 1] = ori;
    }
}

__global__ __launch_bounds__(kThreads) void k_render_train_bwd(RenderArgs a, const float* __restrict__ weights_sum,
                                                                const float* __restrict__ depth, const float* __restrict__ image,
                                                                const float* __restrict__ g_weights_sum,
                                                                const float* __restrict__ g_depth, const float* __restrict__ g_image,
                                                                const float* __restrict__ g_ray_sums, float* __restrict__ dsigma7,
                                                                float* __restrict__ dalbedo) {
    const size_t cap = a.cap;
    auto zero_row = [&](uint32_t i) {
#pragma unroll
        for (uint32_t s = 0; s < 7; s++) dsigma7[s * cap + i] = 0.f;
        dalbedo[(size_t)i * 3 + 0] = 0.f; dalbedo[(size_t)i * 3 + 1] = 0.f; dalbedo[(size_t)i * 3 + 2] = 0.f;
    };
    if (blockIdx.x >= a.ray_blocks) {
        const uint32_t total = (uint32_t)a.total_p[0];
        for (uint32_t i = total + (blockIdx.x - a.ray_blocks) * kThreads + threadIdx.x; i < a.cap; i += kPadBlocks * kThreads) zero_row(i);
        return;
    }
    const uint32_t n = (blockIdx.x * kThreads + threadIdx.x) >> 6;
    if (n >= a.n_rays) return;
    const int lane = lane_id();
    const uint32_t offset = (uint32_t)a.rays[n * 2], count = (uint32_t)a.rays[n * 2 + 1];
    if (count == 0 || offset + count > a.cap) {          // raymarching.cu:630: no gradient for such a ray
        for (uint32_t k = lane; k < count && offset + k < a.cap; k += kWave) zero_row(offset + k);
        return;
    }
    const int mode = shading_mode(a);
    const float ratio = a.ratio_p[0];
    const Vec3 l = ray_light(a.rays_o, a.light_off, n);

    const float gi0 = g_image[n * 3 + 0], gi1 = g_image[n * 3 + 1], gi2 = g_image[n * 3 + 2];
    const float gws = g_weights_sum[n], gd = g_depth ? g_depth[n] : 0.f;
    const float g_ent = g_ray_sums ? g_ray_sums[n * 2 + 0] : 0.f, g_ori = g_ray_sums ? g_ray_sums[n * 2 + 1] : 0.f;
    const float r_final = image[n * 3 + 0], g_final = image[n * 3 + 1], b_final = image[n * 3 + 2];
    const float ws_final = weights_sum[n], d_final = depth[n];

    float T_carry = 1.0f, r_c = 0, g_c = 0, b_c = 0, ws_c = 0, d_c = 0;
    bool done = false;
    const bool lamb = mode == kLambertian;
    Raw nxt = load_raw(a, offset + (uint32_t)lane, (uint32_t)lane < count, lamb);
    for (uint32_t base = 0; base < count; base += kWave) {
        const uint32_t k = base + lane;
        const bool valid = k < count;
        const uint32_t i = offset + (valid ? k : 0);
        if (done) {   // samples behind the cut get no gradient (the reference leaves its zero-initialised rows alone)
            if (valid) zero_row(i);
            continue;
        }
        const Raw cur = nxt;
        nxt = load_raw(a, i + kWave, k + kWave < count, lamb);
        float alpha = 0.f, t = 0.f, dt = 0.f, c[3] = {0.f, 0.f, 0.f}, o = 0.f;
        Sample p = {};
        if (valid) {
            p = make_sample(cur.s, cur.d, a.e, l);
            sample_forward(p, ratio, mode, lamb ? cur.alb : nullptr, c, o);
            t = cur.t; dt = cur.dt;
            alpha = 1.0f - __expf(-cur.s[0] * dt);
        }
        const float incl = wave_incl_prod(1.0f - alpha, lane);
        float excl = __shfl_up(incl, 1, kWave);
        if (lane == 0) excl = 1.0f;
        const float T_before = T_carry * excl;
        const float T = T_carry * incl;   // already advanced, as at raymarching.cu:664
        const unsigned long long cut = __ballot(valid && (T < a.T_thresh));
        const int first_cut = cut ? (int)__ffsll((long long)cut) - 1 : kWave;
        const bool contributes = valid && lane <= first_cut;
        const float w = contributes ? alpha * T_before : 0.0f;

        const float r = r_c + wave_incl_sum(w * c[0], lane);
        const float g = g_c + wave_incl_sum(w * c[1], lane);
        const float b = b_c + wave_incl_sum(w * c[2], lane);
        const float ws = ws_c + wave_incl_sum(w, lane);
        const float d = d_c + wave_incl_sum(w * t, lane);

        if (contributes) {
            // compositor (raymarching.cu:664-679): grad_rgb = grad_image * w; grad_weights_i is the entropy term's
            const float gw = g_ent * entropy_grad(w);
            const float grgb[3] = {gi0 * w, gi1 * w, gi2 * w};
            const float gsig = dt * (gi0 * (T * c[0] - (r_final - r)) + gi1 * (T * c[1] - (g_final - g)) +
                                     gi2 * (T * c[2] - (b_final - b)) + (gws + gw) * (T - (ws_final - ws)) +
                                     gd * (T * t - (d_final - d)));
            // shading + orientation term (weights are detached in loss_orient: d/d orient_i = g * w_i)
            float dsig[6], dalb[3];
            sample_backward(p, l, ratio, mode, lamb ? cur.alb : nullptr, grgb, nullptr, g_ori * w, a.e, dsig, dalb);
            dsigma7[i] = gsig;
#pragma unroll
            for (uint32_t s = 0; s < 6; s++) dsigma7[(size_t)(s + 1) * cap + i] = dsig[s];
            dalbedo[(size_t)i * 3 + 0] = dalb[0]; dalbedo[(size_t)i * 3 + 1] = dalb[1]; dalbedo[(size_t)i * 3 + 2] = dalb[2];
        } else if (valid) {
            zero_row(i);
        }
        done = cut != 0ull;
        T_carry = T_carry * __shfl(incl, kWave - 1, kWave);
        r_c = __shfl(r, kWave - 1, kWave); g_c = __shfl(g, kWave - 1, kWave); b_c = __shfl(b, kWave - 1, kWave);
        ws_c = __shfl(ws, kWave - 1, kWave); d_c = __shfl(d, kWave - 1, kWave);
    }
}

int fill_args(RenderArgs& a, const float* sigma7, const float* albedo, const float* dirs, const float* ts, const int32_t* rays,
              const float* rays_o, const float* light_offset, const float* ratio, const float* mode_dev, int mode, float epsilon,
              float T_thresh, uint32_t capacity, uint32_t n_rays, const int32_t* total) {
    a.sigma7 = sigma7; a.albedo = albedo; a.dirs = dirs; a.ts = ts; a.rays = rays; a.rays_o = rays_o; a.light_off = light_offset;
    a.ratio_p = ratio; a.mode_p = mode_dev; a.mode = mode; a.e = epsilon; a.T_thresh = T_thresh; a.cap = capacity; a.n_rays = n_rays;
    a.ray_blocks = (uint32_t)div_up((uint64_t)n_rays * kWave, kThreads); a.total_p = total;
    return 0;
}

}  // namespace

extern "C" {

int sdfx_render_train_forward(const float* sigma7, const float* albedo, const float* dirs, const float* ts, const int32_t* rays,
                              const float* rays_o, const float* light_offset, const float* ratio, const float* mode_dev,
                              int mode, float epsilon, float T_thresh, uint32_t capacity, uint32_t n_rays, const int32_t* total,
                              float* weights, float* weights_sum, float* depth, float* image, float* ray_sums,
                              sdfx_stream_t stream) {
    SDFX_REQUIRE(rays && rays_o && light_offset && ratio && total && weights_sum && depth && image && ray_sums,
                 "render_train_forward: null pointer");
    SDFX_REQUIRE(capacity == 0 || (sigma7 && albedo && dirs && ts && weights), "render_train_forward: null sample buffer");
    SDFX_REQUIRE(mode_dev || (mode >= kLambertian && mode <= kNormal), "render_train_forward: mode must be 1, 2 or 3 (or given on the device)");
    SDFX_REQUIRE(epsilon > 0.f, "render_train_forward: epsilon must be positive");
    if (n_rays == 0) return SDFX_OK;   // (capacity 0 = a view without samples: every ray has count 0 and gets zero outputs)
    RenderArgs a;
    fill_args(a, sigma7, albedo, dirs, ts, rays, rays_o, light_offset, ratio, mode_dev, mode, epsilon, T_thresh, capacity, n_rays, total);
    hipLaunchKernelGGL(k_render_train_fwd, dim3(a.ray_blocks + kPadBlocks), dim3(kThreads), 0, as_stream(stream), a, weights,
                       weights_sum, depth, image, ray_sums);
    return check_launch("render_train_forward");
}

int sdfx_render_train_backward(const float* sigma7, const float* albedo, const float* dirs, const float* ts, const int32_t* rays,
                               const float* rays_o, const float* light_offset, const float* ratio, const float* mode_dev,
                               int mode, float epsilon, float T_thresh, uint32_t capacity, uint32_t n_rays, const int32_t* total,
                               const float* weights_sum, const float* depth, const float* image, const float* grad_weights_sum,
                               const float* grad_depth, const float* grad_image, const float* grad_ray_sums, float* dsigma7,
                               float* dalbedo, sdfx_stream_t stream) {
    SDFX_REQUIRE(rays && rays_o && light_offset && ratio && total && weights_sum && depth && image && grad_weights_sum && grad_image,
                 "render_train_backward: null pointer");
    SDFX_REQUIRE(capacity == 0 || (sigma7 && albedo && dirs && ts && dsigma7 && dalbedo), "render_train_backward: null sample buffer");
    SDFX_REQUIRE(mode_dev || (mode >= kLambertian && mode <= kNormal), "render_train_backward: mode must be 1, 2 or 3 (or given on the device)");
    if (capacity == 0 || n_rays == 0) return SDFX_OK;
    RenderArgs a;
    fill_args(a, sigma7, albedo, dirs, ts, rays, rays_o, light_offset, ratio, mode_dev, mode, epsilon, T_thresh, capacity, n_rays, total);
    hipLaunchKernelGGL(k_render_train_bwd, dim3(a.ray_blocks + kPadBlocks), dim3(kThreads), 0, as_stream(stream), a, weights_sum,
                       depth, image, grad_weights_sum, grad_depth, grad_image, grad_ray_sums, dsigma7, dalbedo);
    return check_launch("render_train_backward");
}

}  // extern "C"

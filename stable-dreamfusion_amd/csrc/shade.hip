// shade.hip — the tail of NeRFNetwork.forward between the field and the compositor, one kernel each way:
//
//   normal   = -[0.5 (s(x+e_k) - s(x-e_k)) / e]_k            finite_difference_normal, network_grid.py:81-96
//   normal   = nan_to_num(normal / sqrt(max(|normal|^2, 1e-20)))   normal(), :98-104 + safe_normalize, utils.py:109
//   lambert  = ratio + (1 - ratio) max(normal . l, 0)        forward(), :117-130; l = per-ray light direction
//   color    = albedo * lambert | lambert | (normal + 1) / 2  ('lambertian' | 'textureless' | 'normal')
//   dirs_n   = dirs / sqrt(max(|dirs|^2, 1e-20))             renderer.py:734
//   orient   = max(normal . dirs_n, 0)^2                     the per-sample factor of loss_orient, renderer.py:744-746
//
// In PyTorch this is ~40 elementwise launches forward and ~60 backward on [M]-sized tensors (M ~ 4e5): once the
// iteration is replayed as a HIP graph they are a third of its GPU time at ~5 us apiece. Here: one wavefront per
// ray walks the ray's samples (the light direction is per ray: safe_normalize(rays_o + offset), renderer.py:727),
// every array is read or written once, coalesced. HBM-bound streaming: 7*4 + 12 + 12 in, 12 + 12 + 4 out per sample.
// Rows of fixed-capacity buffers that belong to no ray (padding) are zero-filled by extra workgroups.
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

inline uint64_t min_u64(uint64_t a, uint64_t b) { return a < b ? a : b; }
// (per-sample arithmetic: shade_math.h, shared with the host test harness)

__global__ __launch_bounds__(kThreads) void k_shade_forward(const float* __restrict__ sigma7, const float* __restrict__ albedo,
                                                            const float* __restrict__ dirs, const int32_t* __restrict__ rays,
                                                            const float* __restrict__ rays_o,
                                                            const float* __restrict__ light_off,
                                                            const float* __restrict__ ratio_p, int mode, float e, uint32_t cap,
                                                            uint32_t n_rays, uint32_t ray_blocks,
                                                            const int32_t* __restrict__ total_p, float* __restrict__ color,
                                                            float* __restrict__ normal, float* __restrict__ orient) {
    if (blockIdx.x >= ray_blocks) {  // padding rows [total, cap): belong to no ray
        const uint32_t total = (uint32_t)total_p[0];
        for (uint32_t i = total + (blockIdx.x - ray_blocks) * kThreads + threadIdx.x; i < cap; i += kPadBlocks * kThreads) {
            color[(size_t)i * 3 + 0] = 0.f; color[(size_t)i * 3 + 1] = 0.f; color[(size_t)i * 3 + 2] = 0.f;
            normal[(size_t)i * 3 + 0] = 0.f; normal[(size_t)i * 3 + 1] = 0.f; normal[(size_t)i * 3 + 2] = 0.f;
            orient[i] = 0.f;
        }
        return;
    }
    const uint32_t n = (blockIdx.x * kThreads + threadIdx.x) >> 6;
    if (n >= n_rays) return;
    const uint32_t offset = (uint32_t)rays[n * 2], count = (uint32_t)rays[n * 2 + 1];
    const Vec3 l = ray_light(rays_o, light_off, n);
    const float ratio = ratio_p[0];
    for (uint32_t k = lane_id(); k < count; k += kWave) {
        const uint32_t i = offset + k;
        if (i >= cap) break;
        const Sample p = load_sample(sigma7, dirs, cap, i, e, l);
        float c[3], o;
        sample_forward(p, ratio, mode, mode == kLambertian ? albedo + (size_t)i * 3 : nullptr, c, o);
        color[(size_t)i * 3 + 0] = c[0]; color[(size_t)i * 3 + 1] = c[1]; color[(size_t)i * 3 + 2] = c[2];
        normal[(size_t)i * 3 + 0] = p.n.x; normal[(size_t)i * 3 + 1] = p.n.y; normal[(size_t)i * 3 + 2] = p.n.z;
        orient[i] = o;
    }
}

__global__ __launch_bounds__(kThreads) void k_shade_backward(const float* __restrict__ sigma7, const float* __restrict__ albedo,
                                                             const float* __restrict__ dirs, const int32_t* __restrict__ rays,
                                                             const float* __restrict__ rays_o,
                                                             const float* __restrict__ light_off,
                                                             const float* __restrict__ ratio_p, int mode, float e, uint32_t cap,
                                                             uint32_t n_rays, uint32_t ray_blocks,
                                                             const int32_t* __restrict__ total_p,
                                                             const float* __restrict__ dcolor, const float* __restrict__ dnormal,
                                                             const float* __restrict__ dorient, float* __restrict__ dsigma7,
                                                             float* __restrict__ dalbedo) {
    if (blockIdx.x >= ray_blocks) {
        const uint32_t total = (uint32_t)total_p[0];
        for (uint32_t i = total + (blockIdx.x - ray_blocks) * kThreads + threadIdx.x; i < cap; i += kPadBlocks * kThreads) {
#pragma unroll
            for (uint32_t r = 0; r < 7; r++) dsigma7[(size_t)r * cap + i] = 0.f;
            if (dalbedo) { dalbedo[(size_t)i * 3 + 0] = 0.f; dalbedo[(size_t)i * 3 + 1] = 0.f; dalbedo[(size_t)i * 3 + 2] = 0.f; }
        }
        return;
    }
    const uint32_t n = (blockIdx.x * kThreads + threadIdx.x) >> 6;
    if (n >= n_rays) return;
    const uint32_t offset = (uint32_t)rays[n * 2], count = (uint32_t)rays[n * 2 + 1];
    const Vec3 l = ray_light(rays_o, light_off, n);
    const float ratio = ratio_p[0];
    for (uint32_t k = lane_id(); k < count; k += kWave) {
        const uint32_t i = offset + k;
        if (i >= cap) break;
        const Sample p = load_sample(sigma7, dirs, cap, i, e, l);
        const float g[3] = {dcolor[(size_t)i * 3 + 0], dcolor[(size_t)i * 3 + 1], dcolor[(size_t)i * 3 + 2]};
        float dsig[6], dalb[3];
        sample_backward(p, l, ratio, mode, mode == kLambertian ? albedo + (size_t)i * 3 : nullptr, g,
                        dnormal ? dnormal + (size_t)i * 3 : nullptr, dorient[i], e, dsig, dalb);
        dsigma7[i] = 0.f;  // the centre density only feeds the compositor
#pragma unroll
        for (uint32_t r = 0; r < 6; r++) dsigma7[(size_t)(r + 1) * cap + i] = dsig[r];
        if (dalbedo) { dalbedo[(size_t)i * 3 + 0] = dalb[0]; dalbedo[(size_t)i * 3 + 1] = dalb[1]; dalbedo[(size_t)i * 3 + 2] = dalb[2]; }
    }
}

// ---- binary entropy of the sample weights (Trainer.train_step's lambda_entropy term, nerf/utils.py:571-575):
//      alphas = weights.clamp(1e-5, 1 - 1e-5);  H = -alphas log2 alphas - (1 - alphas) log2(1 - alphas);  sum over rows < total
constexpr float kAlphaLo = 1e-5f, kAlphaHi = 1.f - 1e-5f;

__global__ __launch_bounds__(kThreads) void k_entropy_forward(const float* __restrict__ w, uint32_t cap,
                                                              const int32_t* __restrict__ total_p, double* __restrict__ out) {
    __shared__ double part[kThreads / 64];
    const uint32_t total = min((uint32_t)total_p[0], cap);
    double acc = 0.0;
    for (uint32_t i = blockIdx.x * kThreads + threadIdx.x; i < total; i += gridDim.x * kThreads) {
        const float a = fminf(fmaxf(w[i], kAlphaLo), kAlphaHi);
        acc += (double)(-a * log2f(a) - (1.f - a) * log2f(1.f - a));
    }
    for (int off = 32; off > 0; off >>= 1) acc += __shfl_down(acc, off);
    if ((threadIdx.x & 63) == 0) part[threadIdx.x >> 6] = acc;
    __syncthreads();
    if (threadIdx.x == 0) {
        double s = 0.0;
        for (uint32_t k = 0; k < kThreads / 64; k++) s += part[k];
        if (s != 0.0) atomicAdd(out, s);
    }
}

__global__ __launch_bounds__(kThreads) void k_entropy_backward(const float* __restrict__ w, uint32_t cap,
                                                               const int32_t* __restrict__ total_p,
                                                               const float* __restrict__ gout, float* __restrict__ dw) {
    const uint32_t total = min((uint32_t)total_p[0], cap);
    const float g = gout[0];
    for (uint32_t i = blockIdx.x * kThreads + threadIdx.x; i < cap; i += gridDim.x * kThreads) {
        const float x = w[i];
        float d = 0.f;
        if (i < total && x >= kAlphaLo && x <= kAlphaHi) d = g * (log2f(1.f - x) - log2f(x));  // clamp passes the gradient inside
        dw[i] = d;
    }
}

}  // namespace

extern "C" {

int sdfx_shade_forward(const float* sigma7, const float* albedo, const float* dirs, const int32_t* rays, const float* rays_o,
                       const float* light_offset, const float* ratio, int mode, float epsilon, uint32_t capacity,
                       uint32_t n_rays, const int32_t* total, float* color, float* normal, float* orient,
                       sdfx_stream_t stream) {
    SDFX_REQUIRE(sigma7 && dirs && rays && rays_o && light_offset && ratio && total && color && normal && orient,
                 "shade_forward: null pointer");
    SDFX_REQUIRE(mode >= kLambertian && mode <= kNormal, "shade_forward: mode must be 1 (lambertian), 2 (textureless) or 3 (normal)");
    SDFX_REQUIRE(mode != kLambertian || albedo, "shade_forward: lambertian shading needs albedo");
    SDFX_REQUIRE(epsilon > 0.f, "shade_forward: epsilon must be positive");
    if (capacity == 0 || n_rays == 0) return SDFX_OK;
    const uint32_t ray_blocks = (uint32_t)div_up((uint64_t)n_rays * kWave, kThreads);
    hipLaunchKernelGGL(k_shade_forward, dim3(ray_blocks + kPadBlocks), dim3(kThreads), 0, as_stream(stream), sigma7, albedo, dirs,
                       rays, rays_o, light_offset, ratio, mode, epsilon, capacity, n_rays, ray_blocks, total, color, normal, orient);
    return check_launch("shade_forward");
}

int sdfx_shade_backward(const float* sigma7, const float* albedo, const float* dirs, const int32_t* rays, const float* rays_o,
                        const float* light_offset, const float* ratio, int mode, float epsilon, uint32_t capacity,
                        uint32_t n_rays, const int32_t* total, const float* dcolor, const float* dnormal, const float* dorient,
                        float* dsigma7, float* dalbedo, sdfx_stream_t stream) {
    SDFX_REQUIRE(sigma7 && dirs && rays && rays_o && light_offset && ratio && total && dcolor && dorient && dsigma7,
                 "shade_backward: null pointer");
    SDFX_REQUIRE(mode >= kLambertian && mode <= kNormal, "shade_backward: mode must be 1, 2 or 3");
    SDFX_REQUIRE(mode != kLambertian || (albedo && dalbedo), "shade_backward: lambertian shading needs albedo and dalbedo");
    if (capacity == 0 || n_rays == 0) return SDFX_OK;
    const uint32_t ray_blocks = (uint32_t)div_up((uint64_t)n_rays * kWave, kThreads);
    hipLaunchKernelGGL(k_shade_backward, dim3(ray_blocks + kPadBlocks), dim3(kThreads), 0, as_stream(stream), sigma7, albedo,
                       dirs, rays, rays_o, light_offset, ratio, mode, epsilon, capacity, n_rays, ray_blocks, total, dcolor,
                       dnormal, dorient, dsigma7, dalbedo);
    return check_launch("shade_backward");
}

int sdfx_entropy_forward(const float* weights, uint32_t capacity, const int32_t* total, double* sum_out, sdfx_stream_t stream) {
    SDFX_REQUIRE(total && sum_out && (weights || capacity == 0), "entropy_forward: null pointer");
    hipStream_t st = as_stream(stream);
    zero_device(sum_out, sizeof(double), st);
    if (capacity == 0) return SDFX_OK;
    const uint32_t blocks = (uint32_t)min_u64(div_up(capacity, kThreads * 4), 256);
    hipLaunchKernelGGL(k_entropy_forward, dim3(blocks), dim3(kThreads), 0, st, weights, capacity, total, sum_out);
    return check_launch("entropy_forward");
}

int sdfx_entropy_backward(const float* weights, uint32_t capacity, const int32_t* total, const float* grad_sum,
                          float* grad_weights, sdfx_stream_t stream) {
    SDFX_REQUIRE(total && grad_sum && (capacity == 0 || (weights && grad_weights)), "entropy_backward: null pointer");
    if (capacity == 0) return SDFX_OK;
    const uint32_t blocks = (uint32_t)min_u64(div_up(capacity, kThreads * 4), 2048);
    hipLaunchKernelGGL(k_entropy_backward, dim3(blocks), dim3(kThreads), 0, as_stream(stream), weights, capacity, total,
                       grad_sum, grad_weights);
    return check_launch("entropy_backward");
}

}  // extern "C"

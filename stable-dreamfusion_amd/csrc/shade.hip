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

using namespace sdfx;

namespace {

constexpr uint32_t kThreads = 256;
constexpr uint32_t kPadBlocks = 64;

inline uint64_t min_u64(uint64_t a, uint64_t b) { return a < b ? a : b; }
constexpr float kNormEps = 1e-20f;  // safe_normalize's clamp (nerf/utils.py:109-110)
constexpr float kFltMax = 3.402823466e38f;

enum { kLambertian = 1, kTextureless = 2, kNormal = 3 };

struct Vec3 {
    float x, y, z;
};

__device__ __forceinline__ float nan_to_num_(float v) {  // torch.nan_to_num defaults
    if (v != v) return 0.f;
    if (v > kFltMax) return kFltMax;
    if (v < -kFltMax) return -kFltMax;
    return v;
}

// light direction of ray n: safe_normalize(rays_o[n] + offset)
__device__ __forceinline__ Vec3 ray_light(const float* __restrict__ rays_o, const float* __restrict__ off, uint32_t n) {
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

__device__ __forceinline__ Sample load_sample(const float* __restrict__ sigma7, const float* __restrict__ dirs, uint32_t cap,
                                              uint32_t i, float e, const Vec3& l) {
    Sample p;
    const float* s = sigma7 + i;
    p.raw.x = -(0.5f * (s[1 * (size_t)cap] - s[2 * (size_t)cap]) / e);
    p.raw.y = -(0.5f * (s[3 * (size_t)cap] - s[4 * (size_t)cap]) / e);
    p.raw.z = -(0.5f * (s[5 * (size_t)cap] - s[6 * (size_t)cap]) / e);
    p.q = p.raw.x * p.raw.x + p.raw.y * p.raw.y + p.raw.z * p.raw.z;
    p.s = sqrtf(fmaxf(p.q, kNormEps));
    p.y = {p.raw.x / p.s, p.raw.y / p.s, p.raw.z / p.s};
    p.n = {nan_to_num_(p.y.x), nan_to_num_(p.y.y), nan_to_num_(p.y.z)};
    const float dx = dirs[(size_t)i * 3 + 0], dy = dirs[(size_t)i * 3 + 1], dz = dirs[(size_t)i * 3 + 2];
    const float ds = sqrtf(fmaxf(dx * dx + dy * dy + dz * dz, kNormEps));
    p.d = {dx / ds, dy / ds, dz / ds};
    p.ndl = p.n.x * l.x + p.n.y * l.y + p.n.z * l.z;
    p.ndd = p.n.x * p.d.x + p.n.y * p.d.y + p.n.z * p.d.z;
    return p;
}

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
        const float lambert = ratio + (1.f - ratio) * fmaxf(p.ndl, 0.f);
        float cx, cy, cz;
        if (mode == kNormal) {
            cx = (p.n.x + 1.f) / 2.f; cy = (p.n.y + 1.f) / 2.f; cz = (p.n.z + 1.f) / 2.f;
        } else if (mode == kTextureless) {
            cx = cy = cz = lambert;
        } else {
            cx = albedo[(size_t)i * 3 + 0] * lambert; cy = albedo[(size_t)i * 3 + 1] * lambert;
            cz = albedo[(size_t)i * 3 + 2] * lambert;
        }
        color[(size_t)i * 3 + 0] = cx; color[(size_t)i * 3 + 1] = cy; color[(size_t)i * 3 + 2] = cz;
        normal[(size_t)i * 3 + 0] = p.n.x; normal[(size_t)i * 3 + 1] = p.n.y; normal[(size_t)i * 3 + 2] = p.n.z;
        const float o = fmaxf(p.ndd, 0.f);
        orient[i] = o * o;
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
        const float gx = dcolor[(size_t)i * 3 + 0], gy = dcolor[(size_t)i * 3 + 1], gz = dcolor[(size_t)i * 3 + 2];
        // gradient with respect to the (normalised, nan_to_num'ed) normal
        Vec3 dn = {0.f, 0.f, 0.f};
        if (dnormal) dn = {dnormal[(size_t)i * 3 + 0], dnormal[(size_t)i * 3 + 1], dnormal[(size_t)i * 3 + 2]};
        const float lambert = ratio + (1.f - ratio) * fmaxf(p.ndl, 0.f);
        float dlambert = 0.f;
        if (mode == kNormal) {
            dn.x += gx / 2.f; dn.y += gy / 2.f; dn.z += gz / 2.f;
        } else if (mode == kTextureless) {
            dlambert = gx + gy + gz;
        } else {
            const float ax = albedo[(size_t)i * 3 + 0], ay = albedo[(size_t)i * 3 + 1], az = albedo[(size_t)i * 3 + 2];
            dlambert = gx * ax + gy * ay + gz * az;
            dalbedo[(size_t)i * 3 + 0] = gx * lambert; dalbedo[(size_t)i * 3 + 1] = gy * lambert;
            dalbedo[(size_t)i * 3 + 2] = gz * lambert;
        }
        if (mode != kLambertian && dalbedo) {
            dalbedo[(size_t)i * 3 + 0] = 0.f; dalbedo[(size_t)i * 3 + 1] = 0.f; dalbedo[(size_t)i * 3 + 2] = 0.f;
        }
        if (p.ndl >= 0.f) {  // torch's clamp(min=0) backward passes the gradient where input >= bound (equality included)
            const float c = dlambert * (1.f - ratio);
            dn.x += c * l.x; dn.y += c * l.y; dn.z += c * l.z;
        }
        if (p.ndd > 0.f) {  // orient = clamp(n.d, 0)^2
            const float c = dorient[i] * 2.f * p.ndd;
            dn.x += c * p.d.x; dn.y += c * p.d.y; dn.z += c * p.d.z;
        }
        // nan_to_num: no gradient through replaced entries
        Vec3 dy = {(p.y.x == p.n.x) ? dn.x : 0.f, (p.y.y == p.n.y) ? dn.y : 0.f, (p.y.z == p.n.z) ? dn.z : 0.f};
        // y = raw / s, s = sqrt(clamp(q, eps)): draw = dy / s - raw (dy . raw) / s^3 [q >= eps]
        Vec3 dr = {dy.x / p.s, dy.y / p.s, dy.z / p.s};
        if (p.q >= kNormEps) {
            const float dot = dy.x * p.raw.x + dy.y * p.raw.y + dy.z * p.raw.z;
            const float c = dot / (p.s * p.s * p.s);
            dr.x -= p.raw.x * c; dr.y -= p.raw.y * c; dr.z -= p.raw.z * c;
        }
        // raw_k = -(0.5 (s_pos - s_neg) / e)
        const float h = 0.5f / e;
        dsigma7[i] = 0.f;
        dsigma7[(size_t)1 * cap + i] = -h * dr.x; dsigma7[(size_t)2 * cap + i] = h * dr.x;
        dsigma7[(size_t)3 * cap + i] = -h * dr.y; dsigma7[(size_t)4 * cap + i] = h * dr.y;
        dsigma7[(size_t)5 * cap + i] = -h * dr.z; dsigma7[(size_t)6 * cap + i] = h * dr.z;
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

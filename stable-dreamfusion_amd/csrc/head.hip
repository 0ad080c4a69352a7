// head.hip — everything between the compositor's per-ray outputs and the guidance network's input, one kernel each way:
//
//   bg      = sigmoid(MLP_bg(FreqEncoder(rays_d)))  or a given colour        network_grid.py:132-153, freqencoder.cu:30-57
//   image   = image + (1 - weights_sum) * bg                                 renderer.py:797-806
//   pred    = [image | weights_sum] laid out [1, C, H, W]                    nerf/utils.py:533-541 (C = 4 latent, 3 RGB)
//   reg     = lambda_opacity mean(weights_sum^2) + lambda_entropy entropy_sum / n + lambda_orient orient_sum / n
//                                                                           nerf/utils.py:563-575, renderer.py:744-746
//
// In PyTorch these are ~25 launches forward and ~35 backward on 4096-element tensors (frequency encode, two GEMMs, ReLU, sigmoid,
// the mix, cat / permute / contiguous, three scalar losses and their sums); inside the replayed HIP graph each of them costs
// 2-5 us of execution plus a 5-15 us dependency gap — together ~0.5 ms of a 3.8 ms iteration (profiles/r02_iteration_trace_152_launches.txt).
//
// 4096 rays are 64 waves: a thread-per-ray kernel leaves three quarters of the chip idle and runs the 39 -> 32 -> 3 background
// MLP (1379 parameters, float32 as sdfx_nerf evaluates it) serially in every lane (65 us forward, 95 us backward when it was
// written that way). The work is laid out per (ray, hidden unit) instead: a workgroup owns 8 (forward) or 32 (backward) rays,
// stages the transposed first-layer weights and the rays' frequency encodings in LDS, and every thread forms hidden units as
// 39-term dot products (weights conflict-free across the lanes, encodings broadcast); the backward forms its 1379 weight-
// gradient entries as dot products over the workgroup's 32 rays and a second kernel adds the per-workgroup partials in a fixed
// order (deterministic).
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "sdfx.h"
#include "sdfx_common.h"

using namespace sdfx;

namespace {

constexpr uint32_t kThreads = 256;
constexpr uint32_t kDeg = 6, kEnc = 3 + 3 * 2 * kDeg /* 39 */, kHid = 32, kOut = 3;
constexpr uint32_t oW1 = 0, oB1 = oW1 + kHid * kEnc, oW2 = oB1 + kHid, oB2 = oW2 + kOut * kHid, kBgParams = oB2 + kOut;   // 1379

struct HeadArgs {
    const float* image_raw;   // [N, 3] compositor output (no background)
    const float* ws;          // [N]
    const float* ray_sums;    // [N, 2] (entropy sum, orientation sum) or null
    const float* rays_d;      // [N, 3] (background network input) or null
    int has_net;              // background MLP given (its tensors are separate __restrict__ kernel parameters: scalar loads)
    const float* bg_color;    // [3] device, used when W1 is null
    const float* lam_entropy; // device scalar
    const float* n_valid;     // device scalar (float): the sample count the two sums are averaged over
    float lam_opacity, lam_orient;
    uint32_t N, C;            // C = 4 (latent: rgb + weights_sum) or 3
};

constexpr uint32_t kRaysF = 8, kRaysB = 32;   // rays per workgroup, forward / backward
constexpr uint32_t kEncP = kEnc + 1, kHidP = kHid + 1;

// element k of the frequency encoding of direction d (k_freq_forward's arithmetic: cos as sin(. + pi/2))
__device__ __forceinline__ float freq_element(const float* __restrict__ d, uint32_t k) {
    if (k < 3) return d[k];
    const uint32_t f = (k - 3) / 6, r = (k - 3) - f * 6, comp = r >= 3 ? r - 3 : r;
    const float a = scalbnf(d[comp], (int)f);
    return sinf(r >= 3 ? a + (kPi / 2) : a);
}

// LDS staging shared by both directions: W1 transposed [kEnc][kHid] (lane j reads word k * 32 + j: conflict-free), the
// encodings [R][kEnc + 1] and the hidden activations [R][kHid + 1]
template <uint32_t R>
struct HeadLds {
    float w1t[kEnc * kHid];
    float enc[R * kEncP];
    float h[R * kHidP];
    float bg[R * 4];
};

template <uint32_t R>
__device__ __forceinline__ void bg_hidden(const HeadArgs& a, const float* __restrict__ W1, const float* __restrict__ b1, uint32_t ray0,
                                          HeadLds<R>& s) {
    for (uint32_t i = threadIdx.x; i < kEnc * kHid; i += kThreads) {
        const uint32_t j = i / kEnc, k = i - j * kEnc;
        s.w1t[k * kHid + j] = W1[i];
    }
    for (uint32_t i = threadIdx.x; i < R * kEnc; i += kThreads) {
        const uint32_t r = i / kEnc, k = i - r * kEnc, n = ray0 + r;
        s.enc[r * kEncP + k] = n < a.N ? freq_element(a.rays_d + (size_t)n * 3, k) : 0.f;
    }
    __syncthreads();
    for (uint32_t i = threadIdx.x; i < R * kHid; i += kThreads) {
        const uint32_t r = i / kHid, j = i - r * kHid;
        float acc = b1[j];
#pragma unroll
        for (uint32_t k = 0; k < kEnc; k++) acc += s.w1t[k * kHid + j] * s.enc[r * kEncP + k];
        s.h[r * kHidP + j] = fmaxf(acc, 0.f);
    }
    __syncthreads();
}

// sigmoid(W2 h + b2) of ray r, channel c
template <uint32_t R>
__device__ __forceinline__ float bg_output(const float* __restrict__ W2, const float* __restrict__ b2, const HeadLds<R>& s, uint32_t r,
                                           uint32_t c) {
    float acc = b2[c];
#pragma unroll
    for (uint32_t j = 0; j < kHid; j++) acc += W2[c * kHid + j] * s.h[r * kHidP + j];
    return 1.0f / (1.0f + expf(-acc));
}

__device__ __forceinline__ double block_sum(double v, double* part) {
    v = wave_sum(v);
    if (lane_id() == 0) part[threadIdx.x >> 6] = v;
    __syncthreads();
    return (part[0] + part[1]) + (part[2] + part[3]);
}

__device__ __forceinline__ float reg_term(const HeadArgs& a, uint32_t n, float w) {
    const float inv_n = 1.0f / a.n_valid[0];
    float r = a.lam_opacity * (w * w) / (float)a.N;
    if (a.ray_sums) r += a.lam_entropy[0] * a.ray_sums[(size_t)n * 2] * inv_n + a.lam_orient * a.ray_sums[(size_t)n * 2 + 1] * inv_n;
    return r;
}

// background network: kRaysF rays per workgroup
__global__ __launch_bounds__(kThreads) void k_head_forward_net(HeadArgs a, const float* __restrict__ W1, const float* __restrict__ b1,
                                                                const float* __restrict__ W2, const float* __restrict__ b2,
                                                                float* __restrict__ pred, double* __restrict__ reg_partials) {
    __shared__ HeadLds<kRaysF> s;
    __shared__ double part[kThreads / 64];
    const uint32_t ray0 = blockIdx.x * kRaysF;
    bg_hidden<kRaysF>(a, W1, b1, ray0, s);
    double reg = 0.0;
    if (threadIdx.x < kRaysF * 4) {
        const uint32_t r = threadIdx.x >> 2, c = threadIdx.x & 3, n = ray0 + r;
        if (n < a.N) {
            const float w = a.ws[n];
            if (c < 3) pred[(size_t)c * a.N + n] = a.image_raw[(size_t)n * 3 + c] + (1 - w) * bg_output<kRaysF>(W2, b2, s, r, c);
            else {
                if (a.C == 4) pred[(size_t)3 * a.N + n] = w;
                reg = (double)reg_term(a, n, w);
            }
        }
    }
    reg = block_sum(reg, part);
    if (threadIdx.x == 0) reg_partials[blockIdx.x] = reg;
}

// background colour: one thread per ray
__global__ __launch_bounds__(kThreads) void k_head_forward_color(HeadArgs a, float* __restrict__ pred, double* __restrict__ reg_partials) {
    __shared__ double part[kThreads / 64];
    const uint32_t n = blockIdx.x * kThreads + threadIdx.x;
    double reg = 0.0;
    if (n < a.N) {
        const float w = a.ws[n];
#pragma unroll
        for (uint32_t c = 0; c < 3; c++) pred[(size_t)c * a.N + n] = a.image_raw[(size_t)n * 3 + c] + (1 - w) * a.bg_color[c];
        if (a.C == 4) pred[(size_t)3 * a.N + n] = w;
        reg = (double)reg_term(a, n, w);
    }
    reg = block_sum(reg, part);
    if (threadIdx.x == 0) reg_partials[blockIdx.x] = reg;
}

// loss_reg = the per-workgroup partials added in a fixed order (deterministic); one wave
__global__ __launch_bounds__(64) void k_head_reg_sum(const double* __restrict__ partials, uint32_t n, float* __restrict__ out) {
    double s = 0.0;
    for (uint32_t i = threadIdx.x; i < n; i += 64) s += partials[i];
    s = wave_sum(s);
    if (threadIdx.x == 0) out[0] = (float)s;
}

// gradients of one ray that do not involve the network: g_image, g_ws (given the background colour of the ray), g_sums
__device__ __forceinline__ void ray_backward(const HeadArgs& a, uint32_t n, const float bg[3], const float gp[3], float g_reg,
                                             const float* __restrict__ g_pred, float* __restrict__ g_ws, float* __restrict__ g_sums) {
    const float w = a.ws[n];
    float gw = -(gp[0] * bg[0] + gp[1] * bg[1] + gp[2] * bg[2]);
    if (a.C == 4) gw += g_pred[(size_t)3 * a.N + n];
    gw += g_reg * a.lam_opacity * 2.f * w / (float)a.N;
    g_ws[n] = gw;
    if (g_sums) {
        const float inv_n = 1.0f / a.n_valid[0];
        g_sums[(size_t)n * 2 + 0] = g_reg * a.lam_entropy[0] * inv_n;
        g_sums[(size_t)n * 2 + 1] = g_reg * a.lam_orient * inv_n;
    }
}

__global__ __launch_bounds__(kThreads) void k_head_backward_color(HeadArgs a, const float* __restrict__ g_pred,
                                                                   const float* __restrict__ g_reg_p, float* __restrict__ g_image,
                                                                   float* __restrict__ g_ws, float* __restrict__ g_sums) {
    const uint32_t n = blockIdx.x * kThreads + threadIdx.x;
    if (n >= a.N) return;
    const float bg[3] = {a.bg_color[0], a.bg_color[1], a.bg_color[2]};
    float gp[3];
#pragma unroll
    for (uint32_t c = 0; c < 3; c++) { gp[c] = g_pred[(size_t)c * a.N + n]; g_image[(size_t)n * 3 + c] = gp[c]; }
    ray_backward(a, n, bg, gp, g_reg_p ? g_reg_p[0] : 0.f, g_pred, g_ws, g_sums);
}

// background network: kRaysB rays per workgroup; wpart [gridDim.x][kBgParams] = this workgroup's share of the weight gradient
__global__ __launch_bounds__(kThreads) void k_head_backward_net(HeadArgs a, const float* __restrict__ W1, const float* __restrict__ b1,
                                                                 const float* __restrict__ W2, const float* __restrict__ b2,
                                                                 const float* __restrict__ g_pred, const float* __restrict__ g_reg_p,
                                                                 float* __restrict__ g_image, float* __restrict__ g_ws,
                                                                 float* __restrict__ g_sums, float* __restrict__ wpart) {
    __shared__ HeadLds<kRaysB> s;
    __shared__ float sdh[kRaysB * kHidP];
    __shared__ float sgp[kRaysB * 4], sdo[kRaysB * 4];
    const uint32_t ray0 = blockIdx.x * kRaysB;
    bg_hidden<kRaysB>(a, W1, b1, ray0, s);
    if (threadIdx.x < kRaysB * 4) {
        const uint32_t r = threadIdx.x >> 2, c = threadIdx.x & 3, n = ray0 + r;
        float bg = 0.f, gp = 0.f, dout = 0.f;
        if (n < a.N && c < 3) {
            bg = bg_output<kRaysB>(W2, b2, s, r, c);
            gp = g_pred[(size_t)c * a.N + n];
            g_image[(size_t)n * 3 + c] = gp;
            dout = gp * (1 - a.ws[n]) * bg * (1 - bg);      // d bg_c = gp_c (1 - w), through the sigmoid
        }
        s.bg[r * 4 + c] = bg; sgp[r * 4 + c] = gp; sdo[r * 4 + c] = dout;
    }
    __syncthreads();
    if (threadIdx.x < kRaysB) {
        const uint32_t r = threadIdx.x, n = ray0 + r;
        if (n < a.N) {
            const float bg[3] = {s.bg[r * 4], s.bg[r * 4 + 1], s.bg[r * 4 + 2]}, gp[3] = {sgp[r * 4], sgp[r * 4 + 1], sgp[r * 4 + 2]};
            ray_backward(a, n, bg, gp, g_reg_p ? g_reg_p[0] : 0.f, g_pred, g_ws, g_sums);
        }
    }
    for (uint32_t i = threadIdx.x; i < kRaysB * kHid; i += kThreads) {
        const uint32_t r = i / kHid, j = i - r * kHid;
        float acc = 0.f;
#pragma unroll
        for (uint32_t c = 0; c < kOut; c++) acc += W2[c * kHid + j] * sdo[r * 4 + c];
        sdh[r * kHidP + j] = s.h[r * kHidP + j] > 0.f ? acc : 0.f;   // ReLU: threshold_backward passes where the output is > 0
    }
    __syncthreads();
    // weight-gradient entries as dot products over the workgroup's rays (rays past N have enc = 0, dout = 0)
    float* out = wpart + (size_t)blockIdx.x * kBgParams;
    for (uint32_t i = threadIdx.x; i < kBgParams; i += kThreads) {
        float acc = 0.f;
        if (i < oB1) {
            const uint32_t j = i / kEnc, k = i - j * kEnc;
#pragma unroll 8
            for (uint32_t r = 0; r < kRaysB; r++) acc += sdh[r * kHidP + j] * s.enc[r * kEncP + k];
        } else if (i < oW2) {
            const uint32_t j = i - oB1;
#pragma unroll 8
            for (uint32_t r = 0; r < kRaysB; r++) acc += sdh[r * kHidP + j];
        } else if (i < oB2) {
            const uint32_t c = (i - oW2) / kHid, j = (i - oW2) - c * kHid;
#pragma unroll 8
            for (uint32_t r = 0; r < kRaysB; r++) acc += sdo[r * 4 + c] * s.h[r * kHidP + j];
        } else {
            const uint32_t c = i - oB2;
#pragma unroll 8
            for (uint32_t r = 0; r < kRaysB; r++) acc += sdo[r * 4 + c];
        }
        out[i] = acc;
    }
}

// 64 entries per workgroup; the partials of an entry are split over 4 threads and joined in a fixed order
__global__ __launch_bounds__(256) void k_head_wgrad_reduce(const float* __restrict__ wpart, uint32_t n_parts, float* __restrict__ dW1,
                                                            float* __restrict__ db1, float* __restrict__ dW2, float* __restrict__ db2) {
    __shared__ float part[4][64];
    const uint32_t e = threadIdx.x & 63, q = threadIdx.x >> 6, i = blockIdx.x * 64 + e;
    float acc = 0.f;
    if (i < kBgParams)
        for (uint32_t k = q; k < n_parts; k += 4) acc += wpart[(size_t)k * kBgParams + i];
    part[q][e] = acc;
    __syncthreads();
    if (q != 0 || i >= kBgParams) return;
    const float v = (part[0][e] + part[1][e]) + (part[2][e] + part[3][e]);
    if (i < oB1) dW1[i - oW1] = v;
    else if (i < oW2) db1[i - oB1] = v;
    else if (i < oB2) dW2[i - oW2] = v;
    else db2[i - oB2] = v;
}

uint64_t reg_partial_bytes(uint32_t N) { return (uint64_t)div_up(N, kRaysF) * sizeof(double) + 8; }

int fill(HeadArgs& a, const float* image_raw, const float* ws, const float* ray_sums, const float* rays_d, const float* W1,
         const float* b1, const float* W2, const float* b2, const float* bg_color, const float* lam_entropy, const float* n_valid,
         float lam_opacity, float lam_orient, uint32_t N, uint32_t C) {
    (void)b1; (void)W2; (void)b2;
    a.image_raw = image_raw; a.ws = ws; a.ray_sums = ray_sums; a.rays_d = rays_d; a.has_net = W1 ? 1 : 0;
    a.bg_color = bg_color; a.lam_entropy = lam_entropy; a.n_valid = n_valid; a.lam_opacity = lam_opacity; a.lam_orient = lam_orient;
    a.N = N; a.C = C;
    return 0;
}

}  // namespace

extern "C" {

uint64_t sdfx_head_scratch_bytes(uint32_t N) {
    return reg_partial_bytes(N) + (uint64_t)div_up(N, kRaysB) * kBgParams * sizeof(float);
}

int sdfx_head_forward(const float* image_raw, const float* weights_sum, const float* ray_sums, const float* rays_d, const float* W1,
                      const float* b1, const float* W2, const float* b2, const float* bg_color, const float* lambda_entropy,
                      const float* n_valid, float lambda_opacity, float lambda_orient, uint32_t N, uint32_t C, float* pred,
                      float* loss_reg, void* scratch, sdfx_stream_t stream) {
    SDFX_REQUIRE(image_raw && weights_sum && n_valid && lambda_entropy && pred && loss_reg && scratch, "head_forward: null pointer");
    SDFX_REQUIRE(C == 3 || C == 4, "head_forward: C must be 3 (RGB) or 4 (latent: RGB + weights_sum)");
    SDFX_REQUIRE((W1 && b1 && W2 && b2 && rays_d) || bg_color, "head_forward: needs the background network (with rays_d) or a background colour");
    if (N == 0) return SDFX_OK;
    HeadArgs a;
    fill(a, image_raw, weights_sum, ray_sums, rays_d, W1, b1, W2, b2, bg_color, lambda_entropy, n_valid, lambda_opacity, lambda_orient, N, C);
    const uint32_t blocks = W1 ? div_up(N, kRaysF) : div_up(N, kThreads);
    hipStream_t st = as_stream(stream);
    if (W1) hipLaunchKernelGGL(k_head_forward_net, dim3(blocks), dim3(kThreads), 0, st, a, W1, b1, W2, b2, pred, static_cast<double*>(scratch));
    else hipLaunchKernelGGL(k_head_forward_color, dim3(blocks), dim3(kThreads), 0, st, a, pred, static_cast<double*>(scratch));
    hipLaunchKernelGGL(k_head_reg_sum, dim3(1), dim3(64), 0, st, static_cast<const double*>(scratch), blocks, loss_reg);
    return check_launch("head_forward");
}

int sdfx_head_backward(const float* image_raw, const float* weights_sum, const float* ray_sums, const float* rays_d, const float* W1,
                       const float* b1, const float* W2, const float* b2, const float* bg_color, const float* lambda_entropy,
                       const float* n_valid, float lambda_opacity, float lambda_orient, uint32_t N, uint32_t C, const float* grad_pred,
                       const float* grad_loss_reg, float* grad_image, float* grad_weights_sum, float* grad_ray_sums, float* dW1,
                       float* db1, float* dW2, float* db2, void* scratch, sdfx_stream_t stream) {
    SDFX_REQUIRE(weights_sum && n_valid && lambda_entropy && grad_pred && grad_image && grad_weights_sum && scratch,
                 "head_backward: null pointer");
    SDFX_REQUIRE(C == 3 || C == 4, "head_backward: C must be 3 or 4");
    SDFX_REQUIRE(!W1 || (b1 && W2 && b2 && rays_d && dW1 && db1 && dW2 && db2), "head_backward: background network needs all of its tensors");
    SDFX_REQUIRE(W1 || bg_color, "head_backward: needs the background network or a background colour");
    if (N == 0) return SDFX_OK;
    HeadArgs a;
    fill(a, image_raw, weights_sum, ray_sums, rays_d, W1, b1, W2, b2, bg_color, lambda_entropy, n_valid, lambda_opacity, lambda_orient, N, C);
    hipStream_t st = as_stream(stream);
    if (!W1) {
        hipLaunchKernelGGL(k_head_backward_color, dim3(div_up(N, kThreads)), dim3(kThreads), 0, st, a, grad_pred, grad_loss_reg, grad_image,
                           grad_weights_sum, grad_ray_sums);
        return check_launch("head_backward");
    }
    const uint32_t blocks = div_up(N, kRaysB);
    float* wpart = reinterpret_cast<float*>(static_cast<char*>(scratch) + reg_partial_bytes(N));
    hipLaunchKernelGGL(k_head_backward_net, dim3(blocks), dim3(kThreads), 0, st, a, W1, b1, W2, b2, grad_pred, grad_loss_reg, grad_image,
                       grad_weights_sum, grad_ray_sums, wpart);
    hipLaunchKernelGGL(k_head_wgrad_reduce, dim3(div_up(kBgParams, 64)), dim3(256), 0, st, wpart, blocks, dW1, db1, dW2, db2);
    return check_launch("head_backward");
}

}  // extern "C"

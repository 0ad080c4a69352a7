// sds.hip — the elementwise arithmetic of the score-distillation step either side of the frozen noise predictor
// (guidance/sd_utils.py:86-159), two kernels instead of ~35 PyTorch launches on 16 384-element tensors, and the view-dependent
// text-embedding mix of Trainer.train_step (nerf/utils.py:448-470), one kernel instead of 9.
//
//   before the UNet   latents = x * 2 - 1 (latent phase)                               sd_utils.py:90
//                     noisy   = sqrt(abar_t) latents + sqrt(1 - abar_t) noise          sd_utils.py:104 (scheduler.add_noise)
//                     input   = [noisy, noisy] in the UNet's dtype, tt = [t, t]        sd_utils.py:106-107
//   after the UNet    eps     = eps_uncond + s (eps_text - eps_uncond)                 sd_utils.py:111-112
//                     grad    = nan_to_num(grad_scale (1 - abar_t) (eps - noise))      sd_utils.py:129-131
//                     loss    = 0.5 sum (latents - (latents - grad))^2 / B             sd_utils.py:157-159
//                     dloss/dx for the caller's backward (the only gradient SDS has)
//
// The random draws (t, noise) stay torch calls made in the reference's order, so a seeded run draws the same numbers with and
// without these kernels. Every intermediate the reference materialises in float16 is rounded to float16 here in the same
// place (PyTorch's half kernels compute in float32 and round once per op): the results are bit-identical to the PyTorch
// expressions except for the order of the loss sum.
#include <hip/hip_runtime.h>
#include <hip/hip_fp16.h>
#include <stdint.h>

#include "sdfx.h"
#include "sdfx_common.h"

using namespace sdfx;

namespace {

constexpr uint32_t kThreads = 256;
constexpr uint32_t kLossThreads = 1024;
constexpr float kFltMax = 3.402823466e38f;

__device__ __forceinline__ float h2f(__half h) { return __half2float(h); }
__device__ __forceinline__ float rh(float v) { return __half2float(__float2half_rn(v)); }   // one float16 rounding

template <bool HALF>
__device__ __forceinline__ float load_lat(const void* p, size_t i) {
    if (HALF) return h2f(static_cast<const __half*>(p)[i]);
    return static_cast<const float*>(p)[i];
}

// one thread per latent element; per = C * h * w elements of one batch item
template <bool HALF>
__global__ __launch_bounds__(kThreads) void k_sds_add_noise(const void* __restrict__ x, int affine, const void* __restrict__ noise,
                                                            const int64_t* __restrict__ t, const float* __restrict__ alphas,
                                                            uint32_t B, uint32_t per, float* __restrict__ latents_out,
                                                            __half* __restrict__ model_input, int64_t* __restrict__ tt) {
    const uint32_t i = blockIdx.x * kThreads + threadIdx.x;
    const uint32_t n = B * per;
    if (i < 2 * B) tt[i] = t[i % B];
    if (i >= n) return;
    const uint32_t b = i / per;
    const float abar = alphas[t[b]];
    float l = load_lat<HALF>(x, i);
    const float e = load_lat<HALF>(noise, i);
    float noisy;
    if (HALF) {   // every op of `a.sqrt() * latents + (1 - a).sqrt() * noise` on float16 tensors rounds to float16
        const float a = rh(abar);
        const float sa = rh(sqrtf(a)), sb = rh(sqrtf(rh(1.0f - a)));
        noisy = rh(rh(sa * l) + rh(sb * e));
    } else {
        if (affine) {
            l = l * 2.0f - 1.0f;
            latents_out[i] = l;
        }
        noisy = sqrtf(abar) * l + sqrtf(1.0f - abar) * e;
    }
    const __half h = __float2half_rn(noisy);
    model_input[i] = h;
    model_input[(size_t)n + i] = h;
}

// one workgroup: 16 384 elements; the loss is one number and must be summed in a fixed order
template <bool HALF>
__global__ __launch_bounds__(kLossThreads) void k_sds_loss(const __half* __restrict__ eps2, const void* __restrict__ noise,
                                                           const void* __restrict__ latents, const int64_t* __restrict__ t,
                                                           const float* __restrict__ alphas, float guidance_scale, float grad_scale,
                                                           float out_scale, uint32_t B, uint32_t per, uint32_t pred_per,
                                                           float* __restrict__ loss, float* __restrict__ grad_latents) {
    __shared__ double part[kLossThreads / kWave];
    const uint32_t n = B * per;
    double acc = 0.0;
    for (uint32_t i = threadIdx.x; i < n; i += kLossThreads) {
        const uint32_t b = i / per;
        // pred_per >= per elements per item in the prediction: a pixel-space UNet with learned variance returns 2 C channels
        // of which the first C are the noise (if_utils.py:92-93 split)
        const size_t j = (size_t)b * pred_per + (i - b * per);
        const float u = h2f(eps2[j]), p = h2f(eps2[(size_t)B * pred_per + j]);
        const float eps = rh(u + rh(guidance_scale * rh(p - u)));            // three float16 tensor ops
        const float gw = grad_scale * (1.0f - alphas[t[b]]);                   // grad_scale * w[:, None, None, None]
        const float l = load_lat<HALF>(latents, i);
        const float diff = HALF ? rh(eps - load_lat<HALF>(noise, i)) : eps - load_lat<HALF>(noise, i);
        float g = gw * diff;
        if (g != g) g = 0.f;                                                   // torch.nan_to_num defaults
        else if (g > kFltMax) g = kFltMax;
        else if (g < -kFltMax) g = -kFltMax;
        const float target = l - g;
        const float d = l - target;                                            // what mse_loss sees (not g: l - g is rounded)
        acc += (double)d * (double)d;
        grad_latents[i] = d / (float)B * out_scale;   // float32 also for float16 latents: the caller rounds once, after its factor
    }
    acc = wave_sum(acc);
    if (lane_id() == 0) part[threadIdx.x / kWave] = acc;
    __syncthreads();
    if (threadIdx.x == 0) {
        double s = 0.0;
        for (uint32_t w = 0; w < kLossThreads / kWave; w++) s += part[w];
        loss[0] = 0.5f * (float)s / (float)B;
    }
}

// out[0] = uncond, out[1] = wf front + ws side + wb back, the three products and two sums each rounded to float16
__global__ __launch_bounds__(kThreads) void k_sds_text_mix(const __half* __restrict__ uncond, const __half* __restrict__ front,
                                                           const __half* __restrict__ side, const __half* __restrict__ back,
                                                           const float* __restrict__ wf, const float* __restrict__ ws,
                                                           const float* __restrict__ wb, uint32_t n, __half* __restrict__ out) {
    const uint32_t i = blockIdx.x * kThreads + threadIdx.x;
    if (i >= n) return;
    const float f = rh(wf[0]), s = rh(ws[0]), b = rh(wb[0]);   // sc[...].to(float16)
    out[i] = uncond[i];
    out[(size_t)n + i] = __float2half_rn(rh(rh(f * h2f(front[i])) + rh(s * h2f(side[i]))) + rh(b * h2f(back[i])));
}

// ---- bilinear resampling to the VAE's input size (sd_utils.py:93, F.interpolate(..., mode='bilinear', align_corners=False)) fused
// with encode_imgs' `2 * imgs - 1` and the cast to the VAE's dtype (sd_utils.py:285) ----
struct Axis {   // PyTorch's area_pixel_compute_source_index + the two taps of upsample_bilinear2d_out_frame
    uint32_t i0, i1;
    float l0, l1;
};
__device__ __forceinline__ Axis axis_taps(uint32_t dst, float scale, uint32_t in_size) {
    float src = scale * ((float)dst + 0.5f) - 0.5f;
    if (src < 0.f) src = 0.f;
    Axis a;
    a.i0 = (uint32_t)src;
    if (a.i0 > in_size - 1) a.i0 = in_size - 1;
    a.i1 = a.i0 + (a.i0 < in_size - 1 ? 1u : 0u);
    a.l1 = src - (float)a.i0;
    a.l0 = 1.0f - a.l1;
    return a;
}

template <bool HALF_OUT>
__global__ __launch_bounds__(kThreads) void k_upsample_fwd(const float* __restrict__ x, uint32_t planes, uint32_t H, uint32_t W, uint32_t OH,
                                                           uint32_t OW, float sh, float sw, int affine, void* __restrict__ out) {
    const uint32_t ox = blockIdx.x * kThreads + threadIdx.x, oy = blockIdx.y, p = blockIdx.z;
    if (ox >= OW) return;
    const Axis ay = axis_taps(oy, sh, H), ax = axis_taps(ox, sw, W);
    const float* xp = x + (size_t)p * H * W;
    float v = ay.l0 * (ax.l0 * xp[(size_t)ay.i0 * W + ax.i0] + ax.l1 * xp[(size_t)ay.i0 * W + ax.i1]) +
              ay.l1 * (ax.l0 * xp[(size_t)ay.i1 * W + ax.i0] + ax.l1 * xp[(size_t)ay.i1 * W + ax.i1]);
    if (affine) v = 2.0f * v - 1.0f;
    const size_t o = ((size_t)p * OH + oy) * OW + ox;
    if (HALF_OUT) static_cast<__half*>(out)[o] = __float2half_rn(v);
    else static_cast<float*>(out)[o] = v;
}

// destination indices whose taps can touch source index s (a superset; the weights decide)
__device__ __forceinline__ void dst_range(uint32_t s, float scale, uint32_t out_size, uint32_t& lo, uint32_t& hi) {
    const float a = ((float)s - 0.5f) / scale - 0.5f, b = ((float)s + 1.5f) / scale - 0.5f;
    const int l = (int)floorf(a) - 1, h = (int)ceilf(b) + 1;
    lo = l < 0 ? 0u : (uint32_t)l;
    hi = h > (int)out_size - 1 ? out_size - 1 : (uint32_t)h;
    if (s == 0) lo = 0;   // clamped source coordinates
}
__device__ __forceinline__ float tap_weight(const Axis& a, uint32_t s) { return (a.i0 == s ? a.l0 : 0.f) + (a.i1 == s ? a.l1 : 0.f); }

// one workgroup per (plane, source row): the rows of grad_out that touch it are folded vertically into LDS (coalesced),
// then every source column folds its horizontal window — the adjoint as a gather, no atomics
template <bool HALF_IN>
__global__ __launch_bounds__(kThreads) void k_upsample_bwd(const void* __restrict__ g, uint32_t H, uint32_t W, uint32_t OH, uint32_t OW,
                                                           float sh, float sw, float out_scale, float* __restrict__ gx) {
    extern __shared__ float col[];   // [OW]
    const uint32_t sy = blockIdx.x, p = blockIdx.y;
    uint32_t ylo, yhi;
    dst_range(sy, sh, OH, ylo, yhi);
    for (uint32_t ox = threadIdx.x; ox < OW; ox += kThreads) {
        float acc = 0.f;
        for (uint32_t oy = ylo; oy <= yhi; oy++) {
            const float wy = tap_weight(axis_taps(oy, sh, H), sy);
            if (wy == 0.f) continue;
            const size_t o = ((size_t)p * OH + oy) * OW + ox;
            acc += wy * (HALF_IN ? h2f(static_cast<const __half*>(g)[o]) : static_cast<const float*>(g)[o]);
        }
        col[ox] = acc;
    }
    __syncthreads();
    for (uint32_t sx = threadIdx.x; sx < W; sx += kThreads) {
        uint32_t xlo, xhi;
        dst_range(sx, sw, OW, xlo, xhi);
        float acc = 0.f;
        for (uint32_t ox = xlo; ox <= xhi; ox++) acc += tap_weight(axis_taps(ox, sw, W), sx) * col[ox];
        gx[((size_t)p * H + sy) * W + sx] = acc * out_scale;
    }
}

}  // namespace

extern "C" {

int sdfx_sds_upsample_forward(const float* x, uint32_t planes, uint32_t H, uint32_t W, uint32_t OH, uint32_t OW, int affine, int out_half,
                              void* out, sdfx_stream_t stream) {
    SDFX_REQUIRE(x && out, "sds_upsample_forward: null pointer");
    SDFX_REQUIRE(planes > 0 && H > 0 && W > 0 && OH > 0 && OW > 0 && OH < 65536 && planes < 65536, "sds_upsample_forward: bad sizes");
    const float sh = (float)H / (float)OH, sw = (float)W / (float)OW;
    const dim3 grid(div_up(OW, kThreads), OH, planes);
    if (out_half) hipLaunchKernelGGL(k_upsample_fwd<true>, grid, dim3(kThreads), 0, as_stream(stream), x, planes, H, W, OH, OW, sh, sw, affine, out);
    else hipLaunchKernelGGL(k_upsample_fwd<false>, grid, dim3(kThreads), 0, as_stream(stream), x, planes, H, W, OH, OW, sh, sw, affine, out);
    return check_launch("sds_upsample_forward");
}

int sdfx_sds_upsample_backward(const void* grad_out, int grad_half, uint32_t planes, uint32_t H, uint32_t W, uint32_t OH, uint32_t OW,
                               int affine, float* grad_x, sdfx_stream_t stream) {
    SDFX_REQUIRE(grad_out && grad_x, "sds_upsample_backward: null pointer");
    SDFX_REQUIRE(planes > 0 && H > 0 && W > 0 && OH > 0 && OW > 0 && OW <= 16384 && planes < 65536, "sds_upsample_backward: bad sizes");
    const float sh = (float)H / (float)OH, sw = (float)W / (float)OW;
    const dim3 grid(H, planes);
    const size_t lds = (size_t)OW * sizeof(float);
    const float out_scale = affine ? 2.0f : 1.0f;
    if (grad_half) hipLaunchKernelGGL(k_upsample_bwd<true>, grid, dim3(kThreads), lds, as_stream(stream), grad_out, H, W, OH, OW, sh, sw, out_scale, grad_x);
    else hipLaunchKernelGGL(k_upsample_bwd<false>, grid, dim3(kThreads), lds, as_stream(stream), grad_out, H, W, OH, OW, sh, sw, out_scale, grad_x);
    return check_launch("sds_upsample_backward");
}

int sdfx_sds_add_noise(const void* x, int is_half, int affine, const void* noise, const int64_t* t, const float* alphas_cumprod,
                       uint32_t B, uint32_t per_item, float* latents_out, void* model_input, int64_t* tt, sdfx_stream_t stream) {
    SDFX_REQUIRE(x && noise && t && alphas_cumprod && model_input && tt, "sds_add_noise: null pointer");
    SDFX_REQUIRE(B > 0 && per_item > 0 && (uint64_t)B * per_item < (1ull << 31), "sds_add_noise: bad sizes B=%u per_item=%u", B, per_item);
    SDFX_REQUIRE(!(is_half && affine), "sds_add_noise: the latent-phase affine map takes float32 input");
    SDFX_REQUIRE(!affine || latents_out, "sds_add_noise: latents_out missing");
    const uint32_t n = B * per_item;
    const dim3 grid(div_up(n > 2 * B ? n : 2 * B, kThreads));
    if (is_half)
        hipLaunchKernelGGL(k_sds_add_noise<true>, grid, dim3(kThreads), 0, as_stream(stream), x, 0, noise, t, alphas_cumprod, B, per_item,
                           latents_out, static_cast<__half*>(model_input), tt);
    else
        hipLaunchKernelGGL(k_sds_add_noise<false>, grid, dim3(kThreads), 0, as_stream(stream), x, affine, noise, t, alphas_cumprod, B,
                           per_item, latents_out, static_cast<__half*>(model_input), tt);
    return check_launch("sds_add_noise");
}

int sdfx_sds_loss(const void* noise_pred, const void* noise, const void* latents, int is_half, const int64_t* t,
                  const float* alphas_cumprod, float guidance_scale, float grad_scale, float out_scale, uint32_t B, uint32_t per_item,
                  uint32_t pred_per_item, float* loss, float* grad_latents, sdfx_stream_t stream) {
    SDFX_REQUIRE(noise_pred && noise && latents && t && alphas_cumprod && loss && grad_latents, "sds_loss: null pointer");
    SDFX_REQUIRE(B > 0 && per_item > 0 && (uint64_t)B * per_item < (1ull << 31), "sds_loss: bad sizes B=%u per_item=%u", B, per_item);
    SDFX_REQUIRE(pred_per_item >= per_item && (uint64_t)2 * B * pred_per_item < (1ull << 31),
                 "sds_loss: pred_per_item=%u must be at least per_item=%u", pred_per_item, per_item);
    if (is_half)
        hipLaunchKernelGGL(k_sds_loss<true>, dim3(1), dim3(kLossThreads), 0, as_stream(stream), static_cast<const __half*>(noise_pred),
                           noise, latents, t, alphas_cumprod, guidance_scale, grad_scale, out_scale, B, per_item, pred_per_item, loss, grad_latents);
    else
        hipLaunchKernelGGL(k_sds_loss<false>, dim3(1), dim3(kLossThreads), 0, as_stream(stream), static_cast<const __half*>(noise_pred),
                           noise, latents, t, alphas_cumprod, guidance_scale, grad_scale, out_scale, B, per_item, pred_per_item, loss, grad_latents);
    return check_launch("sds_loss");
}

int sdfx_sds_text_mix(const void* uncond, const void* front, const void* side, const void* back, const float* w_front,
                      const float* w_side, const float* w_back, uint32_t n, void* out, sdfx_stream_t stream) {
    SDFX_REQUIRE(uncond && front && side && back && w_front && w_side && w_back && out, "sds_text_mix: null pointer");
    if (n == 0) return SDFX_OK;
    hipLaunchKernelGGL(k_sds_text_mix, dim3(div_up(n, kThreads)), dim3(kThreads), 0, as_stream(stream),
                       static_cast<const __half*>(uncond), static_cast<const __half*>(front), static_cast<const __half*>(side),
                       static_cast<const __half*>(back), w_front, w_side, w_back, n, static_cast<__half*>(out));
    return check_launch("sds_text_mix");
}

}  // extern "C"

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
#include <stdlib.h>

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

// what one lane reads for one sample: loaded a round AHEAD of its use (the wave's chunk of the next round is in flight during
// the shading and the scan of this one)
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

// ---- one workgroup per ray, its 64-sample chunks in sibling waves ----------------------------------------------------------------
// Round 2 gave a ray to ONE wave that walked its chunks in turn. At 4096 rays that is ~1 wave per SIMD, and what bounds such a
// wave is not memory (four chunks of loads in flight changed nothing: 15.4 -> 15.2 us) but the LATENCY of its own instruction
// stream: the shading of a chunk is ~500 dependent-ish VALU instructions (IEEE divisions and square roots of safe_normalize, two
// log2f of the entropy term) at ~8 cycles each when nothing else shares the SIMD, ~1.7 us per chunk, five chunks in a row for a ray
// through the density blob. So the chunks of a ray go to kRayWaves sibling waves on different SIMDs: wave w takes chunks w,
// w + kRayWaves, ...; each shades its chunk and scans it locally; the chunk products meet in LDS, every wave rebuilds the
// transmittance at its chunk's start by multiplying the earlier products IN ORDER (the same association as the serial walk, so
// the same bits), and the ray sums are added chunk by chunk in order by wave 0.
// kRays > 1 (measurement variant, SDFX_RENDER_RAYS): kRays rays share one workgroup — kRays groups of kRayWaves sibling waves, each
// group walking its own ray exactly as above; the groups only share the barriers, so every group loops to the longest ray's round
// count (its extra rounds find no valid sample). Fewer, fatter workgroups; per-ray arithmetic and its order are unchanged.
template <uint32_t kRayWaves, uint32_t kRays>
__global__ __launch_bounds__(kRays * kRayWaves * 64) void k_render_train_fwd(RenderArgs a, float* __restrict__ weights,
                                                                   float* __restrict__ weights_sum, float* __restrict__ depth,
                                                                   float* __restrict__ image, float* __restrict__ ray_sums) {
    constexpr uint32_t kRayThreads = kRayWaves * 64, kGroupThreads = kRays * kRayThreads;
    const uint32_t ray_blocks = (a.n_rays + kRays - 1) / kRays;
    if (blockIdx.x >= ray_blocks) {   // padding rows [total, cap) belong to no ray: zero weight
        const uint32_t total = (uint32_t)a.total_p[0];
        for (uint32_t i = total + (blockIdx.x - ray_blocks) * kGroupThreads + threadIdx.x; i < a.cap; i += kPadBlocks * kGroupThreads) weights[i] = 0.f;
        return;
    }
    __shared__ float prod_s[kRays][2][kRayWaves];   // chunk products of the current round (double-buffered across rounds)
    __shared__ float part_s[kRays][kRayWaves][7];   // per-wave partial ray sums
    __shared__ uint32_t rounds_s[kRays];
    const uint32_t rg = kRays > 1 ? threadIdx.x / kRayThreads : 0u;      // this thread's ray within the workgroup
    const uint32_t tr = kRays > 1 ? threadIdx.x % kRayThreads : threadIdx.x;
    float (&prod)[2][kRayWaves] = prod_s[rg];
    float (&part)[kRayWaves][7] = part_s[rg];
    const uint32_t n = blockIdx.x * kRays + rg;
    const int lane = lane_id();
    const uint32_t wv = tr >> 6;
    const bool in_range = n < a.n_rays;
    uint32_t offset = in_range ? (uint32_t)a.rays[n * 2] : 0u, count = in_range ? (uint32_t)a.rays[n * 2 + 1] : 0u;
    bool live_ray = in_range;
    if (in_range && (count == 0 || offset + count > a.cap)) {   // raymarching.cu:521-528 (the reference's weights are zero-initialised)
        for (uint32_t k = tr; k < count && offset + k < a.cap; k += kRayThreads) weights[offset + k] = 0.f;
        if (tr == 0) {
            weights_sum[n] = 0; depth[n] = 0; image[n * 3 + 0] = 0; image[n * 3 + 1] = 0; image[n * 3 + 2] = 0;
            ray_sums[n * 2 + 0] = 0; ray_sums[n * 2 + 1] = 0;
        }
        if (kRays == 1) return;
        live_ray = false; count = 0; offset = 0;      // (its waves keep the other rays' barriers company)
    }
    const int mode = shading_mode(a);
    const float ratio = a.ratio_p[0];
    const Vec3 l = ray_light(a.rays_o, a.light_off, in_range ? n : 0u);
    const bool lamb = mode == kLambertian;
    const uint32_t n_chunks = (count + kWave - 1) / kWave;
    uint32_t rounds = (n_chunks + kRayWaves - 1) / kRayWaves;
    if (kRays > 1) {
        if (tr == 0) rounds_s[rg] = rounds;
        __syncthreads();
#pragma unroll
        for (uint32_t q = 0; q < kRays; q++) rounds = max(rounds, rounds_s[q]);
    }

    float T_round = 1.0f;                          // transmittance at the start of the round's first chunk (same in every wave)
    float r = 0, g = 0, b = 0, ws = 0, d = 0, ent = 0, ori = 0;
    Raw nxt = load_raw(a, offset + wv * kWave + (uint32_t)lane, wv * kWave + (uint32_t)lane < count, lamb);
    for (uint32_t rd = 0; rd < rounds; rd++) {
        const uint32_t k = (rd * kRayWaves + wv) * kWave + lane;
        const bool valid = k < count;
        const uint32_t i = offset + (valid ? k : 0);
        const Raw cur = nxt;
        nxt = load_raw(a, i + kRayThreads, k + kRayThreads < count, lamb);   // this wave's chunk of the next round
        float alpha = 0.f, t = 0.f, c[3] = {0.f, 0.f, 0.f}, o = 0.f;
        if (valid) {
            const Sample p = make_sample(cur.s, cur.d, a.e, l);
            sample_forward(p, ratio, mode, lamb ? cur.alb : nullptr, c, o);
            t = cur.t;
            alpha = 1.0f - __expf(-cur.s[0] * cur.dt);   // raymarching.cu:543 (binarize = false on the training path)
        }
        const float incl = wave_incl_prod(1.0f - alpha, lane);
        const float excl = wave_shift_up1(incl, 1.0f);
        if (lane == kWave - 1) prod[rd & 1][wv] = incl;
        __syncthreads();
        float T_carry = T_round;                   // ((T_round * P_0) * P_1) ... : the serial walk's products, in its order
        float T_next = T_round;
#pragma unroll
        for (uint32_t v = 0; v < kRayWaves; v++) {
            const float pv = prod[rd & 1][v];
            if (v < wv) T_carry = T_carry * pv;
            T_next = T_next * pv;
        }
        T_round = T_next;
        // the cut: the first sample whose transmittance AFTER it falls below T_thresh still counts, nothing behind it does. A
        // chunk that starts below the threshold lies behind the cut (the transmittance never rises)
        const bool behind = T_carry < a.T_thresh;
        const float T_before = T_carry * excl, T_after = T_carry * incl;
        const unsigned long long cut = __ballot(valid && (T_after < a.T_thresh));
        const int first_cut = cut ? (int)__ffsll((long long)cut) - 1 : kWave;
        const float w = (valid && !behind && lane <= first_cut) ? alpha * T_before : 0.0f;
        if (valid) {
            weights[i] = w;
            ent += entropy_bits(w);
            ori += w * o;
        }
        r += w * c[0]; g += w * c[1]; b += w * c[2]; ws += w; d += w * t;
    }
    r = wave_total(r); g = wave_total(g); b = wave_total(b); ws = wave_total(ws); d = wave_total(d); ent = wave_total(ent);
    ori = wave_total(ori);
    if (lane == 0) {
        part[wv][0] = r; part[wv][1] = g; part[wv][2] = b; part[wv][3] = ws; part[wv][4] = d; part[wv][5] = ent; part[wv][6] = ori;
    }
    __syncthreads();
    if (tr == 0 && live_ray) {
        float acc[7];
#pragma unroll
        for (int q = 0; q < 7; q++) {
            acc[q] = part[0][q];
#pragma unroll
            for (uint32_t v = 1; v < kRayWaves; v++) acc[q] += part[v][q];
        }
        weights_sum[n] = acc[3]; depth[n] = acc[4];
        image[n * 3 + 0] = acc[0]; image[n * 3 + 1] = acc[1]; image[n * 3 + 2] = acc[2];
        ray_sums[n * 2 + 0] = acc[5]; ray_sums[n * 2 + 1] = acc[6];
    }
}

template <uint32_t kRayWaves, uint32_t kRays>
__global__ __launch_bounds__(kRays * kRayWaves * 64) void k_render_train_bwd(RenderArgs a, const float* __restrict__ weights_sum,
                                                                   const float* __restrict__ depth, const float* __restrict__ image,
                                                                   const float* __restrict__ g_weights_sum,
                                                                   const float* __restrict__ g_depth, const float* __restrict__ g_image,
                                                                   const float* __restrict__ g_ray_sums, float* __restrict__ dsigma7,
                                                                   float* __restrict__ dalbedo) {
    constexpr uint32_t kRayThreads = kRayWaves * 64, kGroupThreads = kRays * kRayThreads;
    const size_t cap = a.cap;
    auto zero_row = [&](uint32_t i) {
#pragma unroll
        for (uint32_t s = 0; s < 7; s++) dsigma7[s * cap + i] = 0.f;
        dalbedo[(size_t)i * 3 + 0] = 0.f; dalbedo[(size_t)i * 3 + 1] = 0.f; dalbedo[(size_t)i * 3 + 2] = 0.f;
    };
    const uint32_t ray_blocks = (a.n_rays + kRays - 1) / kRays;
    if (blockIdx.x >= ray_blocks) {
        const uint32_t total = (uint32_t)a.total_p[0];
        for (uint32_t i = total + (blockIdx.x - ray_blocks) * kGroupThreads + threadIdx.x; i < a.cap; i += kPadBlocks * kGroupThreads) zero_row(i);
        return;
    }
    __shared__ float prod_s[kRays][2][kRayWaves];
    __shared__ float tot_s[kRays][2][kRayWaves][5];        // chunk totals of (w c_r, w c_g, w c_b, w, w t)
    __shared__ uint32_t rounds_s[kRays];
    const uint32_t rg = kRays > 1 ? threadIdx.x / kRayThreads : 0u;
    const uint32_t tr = kRays > 1 ? threadIdx.x % kRayThreads : threadIdx.x;
    float (&prod)[2][kRayWaves] = prod_s[rg];
    float (&tot)[2][kRayWaves][5] = tot_s[rg];
    const uint32_t n = blockIdx.x * kRays + rg;
    const int lane = lane_id();
    const uint32_t wv = tr >> 6;
    const bool in_range = n < a.n_rays;
    uint32_t offset = in_range ? (uint32_t)a.rays[n * 2] : 0u, count = in_range ? (uint32_t)a.rays[n * 2 + 1] : 0u;
    if (in_range && (count == 0 || offset + count > a.cap)) {          // raymarching.cu:630: no gradient for such a ray
        for (uint32_t k = tr; k < count && offset + k < a.cap; k += kRayThreads) zero_row(offset + k);
        if (kRays == 1) return;
        count = 0; offset = 0;
    }
    const int mode = shading_mode(a);
    const float ratio = a.ratio_p[0];
    const uint32_t ns = in_range ? n : 0u;     // (a ray slot past the last ray reads ray 0's scalars and uses none of them)
    const Vec3 l = ray_light(a.rays_o, a.light_off, ns);

    const float gi0 = g_image[ns * 3 + 0], gi1 = g_image[ns * 3 + 1], gi2 = g_image[ns * 3 + 2];
    const float gws = g_weights_sum[ns], gd = g_depth ? g_depth[ns] : 0.f;
    const float g_ent = g_ray_sums ? g_ray_sums[ns * 2 + 0] : 0.f, g_ori = g_ray_sums ? g_ray_sums[ns * 2 + 1] : 0.f;
    const float r_final = image[ns * 3 + 0], g_final = image[ns * 3 + 1], b_final = image[ns * 3 + 2];
    const float ws_final = weights_sum[ns], d_final = depth[ns];

    const bool lamb = mode == kLambertian;
    const uint32_t n_chunks = (count + kWave - 1) / kWave;
    uint32_t rounds = (n_chunks + kRayWaves - 1) / kRayWaves;
    if (kRays > 1) {
        if (tr == 0) rounds_s[rg] = rounds;
        __syncthreads();
#pragma unroll
        for (uint32_t q = 0; q < kRays; q++) rounds = max(rounds, rounds_s[q]);
    }
    float T_round = 1.0f, acc_round[5] = {0.f, 0.f, 0.f, 0.f, 0.f};   // state at the start of the round's first chunk
    Raw nxt = load_raw(a, offset + wv * kWave + (uint32_t)lane, wv * kWave + (uint32_t)lane < count, lamb);
    for (uint32_t rd = 0; rd < rounds; rd++) {
        const uint32_t k = (rd * kRayWaves + wv) * kWave + lane;
        const bool valid = k < count;
        const uint32_t i = offset + (valid ? k : 0);
        const Raw cur = nxt;
        nxt = load_raw(a, i + kRayThreads, k + kRayThreads < count, lamb);
        float alpha = 0.f, t = 0.f, dt = 0.f, c[3] = {0.f, 0.f, 0.f}, o = 0.f;
        Sample p = {};
        if (valid) {
            p = make_sample(cur.s, cur.d, a.e, l);
            sample_forward(p, ratio, mode, lamb ? cur.alb : nullptr, c, o);
            t = cur.t; dt = cur.dt;
            alpha = 1.0f - __expf(-cur.s[0] * dt);
        }
        const float incl = wave_incl_prod(1.0f - alpha, lane);
        const float excl = wave_shift_up1(incl, 1.0f);
        if (lane == kWave - 1) prod[rd & 1][wv] = incl;
        __syncthreads();
        float T_carry = T_round, T_next = T_round;
#pragma unroll
        for (uint32_t v = 0; v < kRayWaves; v++) {
            const float pv = prod[rd & 1][v];
            if (v < wv) T_carry = T_carry * pv;
            T_next = T_next * pv;
        }
        T_round = T_next;
        const bool behind = T_carry < a.T_thresh;
        const float T_before = T_carry * excl;
        const float T = T_carry * incl;   // already advanced, as at raymarching.cu:664
        const unsigned long long cut = __ballot(valid && (T < a.T_thresh));
        const int first_cut = cut ? (int)__ffsll((long long)cut) - 1 : kWave;
        const bool contributes = valid && !behind && lane <= first_cut;
        const float w = contributes ? alpha * T_before : 0.0f;

        // running accumulators of the forward: chunk-local inclusive sums now, the earlier chunks' totals added in order below
        float sc[5] = {wave_incl_sum(w * c[0], lane), wave_incl_sum(w * c[1], lane), wave_incl_sum(w * c[2], lane),
                       wave_incl_sum(w, lane), wave_incl_sum(w * t, lane)};
        if (lane == kWave - 1) {
#pragma unroll
            for (int q = 0; q < 5; q++) tot[rd & 1][wv][q] = sc[q];
        }
        __syncthreads();
        float carry[5], next[5];
#pragma unroll
        for (int q = 0; q < 5; q++) {
            carry[q] = acc_round[q]; next[q] = acc_round[q];
#pragma unroll
            for (uint32_t v = 0; v < kRayWaves; v++) {
                const float tv = tot[rd & 1][v][q];
                if (v < wv) carry[q] = carry[q] + tv;
                next[q] = next[q] + tv;
            }
            acc_round[q] = next[q];
        }
        const float r = carry[0] + sc[0], g = carry[1] + sc[1], b = carry[2] + sc[2], ws = carry[3] + sc[3], d = carry[4] + sc[4];

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
            zero_row(i);   // behind the cut: no gradient (the reference leaves its zero-initialised rows alone)
        }
    }
}

// Launch shape: sibling waves per ray x rays per workgroup. Measured on the GPU clock at 4096 rays (tools/render_fit.py, replayed graphs;
// profiles/r06_render_rays_per_workgroup.txt), t = fixed + marginal x samples:
//     2 waves, 1 ray  (rounds 3-5)   forward 5.6 us + 11.7 us / Msample   backward 9.5 + 18.4
//     1 wave, 4 rays                 forward 5.1 + 11.7                   backward 6.1 + 21.0
//     2 waves, 2 / 4 rays            forward 6.0 / 6.7 + 12 / 11.4        backward 8.5 / 9.4 + 18.6 / 19.6
//     4 waves, 1 / 2 rays            forward 8.0 / 11.4 + 10.1 / 9.8      backward 14.3 / 17.2 + 16.6 / 17.4
// A 64-thread workgroup per ray is what costs the backward its fixed part (4160 workgroups that each read their ray's scalars, run
// two barriers per round and retire); four rays in a 256-thread workgroup, one wave each, cut it by a third and lose a sixth of the
// streaming rate (a ray's chunks are walked by one wave again). The iteration runs at ~120 samples per ray (0.5 M samples), where the
// second shape is 4 % / 11 % faster; from ~290 samples per ray on the first one wins: chosen by the launch's samples per ray. Results
// are identical up to the order in which a ray's chunk sums are added. SDFX_RENDER_WAVES / SDFX_RENDER_RAYS (devtools library) force a shape.
constexpr uint32_t kWideRaySamples = 256;   // mean samples per ray (capacity / rays) above which the two-wave-per-ray shape is used
void launch_shape(uint32_t capacity, uint32_t n_rays, int& waves, int& rays) {
    const int w = dev_switch("SDFX_RENDER_WAVES", 0), r = dev_switch("SDFX_RENDER_RAYS", 0);
    if (w || r) {
        waves = (w == 1 || w == 4 || w == 8) ? w : 2;
        rays = (r == 2 || r == 4) ? r : 1;
        return;
    }
    const bool wide = (uint64_t)capacity > (uint64_t)kWideRaySamples * n_rays;
    waves = wide ? 2 : 1;
    rays = wide ? 1 : 4;
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
#define SDFX_RFWD(RW_, RR_) hipLaunchKernelGGL((k_render_train_fwd<RW_, RR_>), dim3(div_up(n_rays, RR_) + kPadBlocks), dim3(RR_ * RW_ * 64), 0, \
                                              as_stream(stream), a, weights, weights_sum, depth, image, ray_sums)
    int rw, rr;
    launch_shape(capacity, n_rays, rw, rr);
    if (rr == 2 && rw == 2) SDFX_RFWD(2, 2);
    else if (rr == 4 && rw == 2) SDFX_RFWD(2, 4);
    else if (rr == 2 && rw == 1) SDFX_RFWD(1, 2);
    else if (rr == 4 && rw == 1) SDFX_RFWD(1, 4);
    else if (rr == 2 && rw == 4) SDFX_RFWD(4, 2);
    else if (rw == 1) SDFX_RFWD(1, 1);
    else if (rw == 4) SDFX_RFWD(4, 1);
    else if (rw == 8) SDFX_RFWD(8, 1);
    else SDFX_RFWD(2, 1);
#undef SDFX_RFWD
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
#define SDFX_RBWD(RW_, RR_) hipLaunchKernelGGL((k_render_train_bwd<RW_, RR_>), dim3(div_up(n_rays, RR_) + kPadBlocks), dim3(RR_ * RW_ * 64), 0, \
                                              as_stream(stream), a, weights_sum, depth, image, grad_weights_sum, grad_depth, grad_image,  \
                                              grad_ray_sums, dsigma7, dalbedo)
    int rw, rr;
    launch_shape(capacity, n_rays, rw, rr);
    if (rr == 2 && rw == 2) SDFX_RBWD(2, 2);
    else if (rr == 4 && rw == 2) SDFX_RBWD(2, 4);
    else if (rr == 2 && rw == 1) SDFX_RBWD(1, 2);
    else if (rr == 4 && rw == 1) SDFX_RBWD(1, 4);
    else if (rr == 2 && rw == 4) SDFX_RBWD(4, 2);
    else if (rw == 1) SDFX_RBWD(1, 1);
    else if (rw == 4) SDFX_RBWD(4, 1);
    else if (rw == 8) SDFX_RBWD(8, 1);
    else SDFX_RBWD(2, 1);
#undef SDFX_RBWD
    return check_launch("render_train_backward");
}

}  // extern "C"

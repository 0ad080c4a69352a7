// hostmath.cpp — TEST HARNESS ONLY. Builds the per-sample device arithmetic of
// stable-dreamfusion_amd/csrc/sdfx_math.h for the host (g++, -ffp-contract=off) so the
// not-gpu test suite can compare it with the independent C oracle without a GPU. The loops
// below mirror the thread bodies of k_march_count / k_grid_forward; nothing here ships.
#include <cstdint>
#include <cmath>
#include <cstring>

#include "../../stable-dreamfusion_amd/csrc/sdfx_math.h"
#include "../../stable-dreamfusion_amd/csrc/shade_math.h"
#include "../../stable-dreamfusion_amd/csrc/optim_math.h"

using namespace sdfx;

extern "C" {

// body of k_march_count for every ray, plus tbuf recording
void hm_march_count(const float* rays_o, const float* rays_d, const uint8_t* grid, float bound, int contract,
                    float dt_gamma, uint32_t max_steps, uint32_t N, uint32_t C, uint32_t H, const float* nears,
                    const float* fars, const float* noises, int32_t* counts, float* tbuf) {
    const MarchParams p = make_march_params(bound, contract, dt_gamma, max_steps, C, H);
    for (uint32_t n = 0; n < N; n++) {
        const MarchRay r = make_march_ray(rays_o + (size_t)n * 3, rays_d + (size_t)n * 3);
        const float far = fars[n];
        float t = nears[n];
        t += clampf_(t * p.dt_gamma, p.dt_min, p.dt_max) * noises[n];
        uint32_t step = 0;
        while (t < far && step < max_steps) {
            float dt, cx, cy, cz;
            if (march_probe(r, p, grid, t, dt, cx, cy, cz)) {
                if (tbuf) tbuf[(size_t)n * max_steps + step] = t;
                step++;
                t += dt;
            }
        }
        counts[n] = (int32_t)step;
    }
}

// The wave-per-ray counting pass (k_march_count_wave) emulated lane by lane: 64 consecutive points of the ray's time
// lattice are probed "in parallel" (each lane: occupied? if not, how many lattice steps does the serial march skip?),
// then the serial decision chain is replayed over the 64 results. Must reproduce hm_march_count bit for bit.
void hm_march_count_wave(const float* rays_o, const float* rays_d, const uint8_t* grid, float bound, int contract,
                         float dt_gamma, uint32_t max_steps, uint32_t N, uint32_t C, uint32_t H, const float* nears,
                         const float* fars, const float* noises, int32_t* counts, float* tbuf) {
    const MarchParams p = make_march_params(bound, contract, dt_gamma, max_steps, C, H);
    constexpr uint32_t W = 64;
    for (uint32_t n = 0; n < N; n++) {
        const MarchRay r = make_march_ray(rays_o + (size_t)n * 3, rays_d + (size_t)n * 3);
        const float far = fars[n];
        float base = nears[n];
        base += clampf_(base * p.dt_gamma, p.dt_min, p.dt_max) * noises[n];
        uint32_t step = 0;
        bool done = false;
        while (!done) {
            float tl[W], tafter[W];
            uint32_t hop[W];
            bool occ[W], live[W];
            float t = base;
            for (uint32_t j = 0; j < W; j++) {          // lane j: lattice point j of this chunk
                tl[j] = t;
                t = march_advance(p, t);
            }
            const float t_next_chunk = t;                // lattice point 64
            for (uint32_t j = 0; j < W; j++) {          // "parallel" probes
                live[j] = tl[j] < far;
                occ[j] = false; hop[j] = 1; tafter[j] = tl[j];
                if (!live[j]) continue;
                float tt = tl[j], dt, cx, cy, cz;
                occ[j] = march_probe(r, p, grid, tt, dt, cx, cy, cz, &hop[j]);
                if (occ[j]) { hop[j] = 1; tafter[j] = tl[j] + dt; } else tafter[j] = tt;
            }
            uint32_t q = 0;                              // the serial decision chain over the chunk
            float carry = t_next_chunk;
            for (;;) {
                if (q >= W) { base = carry; break; }
                if (!live[q] || step >= max_steps) { done = true; break; }
                if (occ[q]) {
                    if (tbuf) tbuf[(size_t)n * max_steps + step] = tl[q];
                    step++;
                }
                carry = tafter[q];                       // lattice value at index q + hop[q]
                q += hop[q];
                if (q == W) carry = t_next_chunk;
            }
        }
        counts[n] = (int32_t)step;
    }
}

// bodies of k_shade_forward / k_shade_backward over all rays (same per-sample source as the kernels: shade_math.h)
void hm_shade_forward(const float* sigma7, const float* albedo, const float* dirs, const int32_t* rays, const float* rays_o,
                      const float* light_off, float ratio, int mode, float e, uint32_t cap, uint32_t n_rays, float* color,
                      float* normal, float* orient) {
    using namespace sdfx::shade;
    for (uint32_t n = 0; n < n_rays; n++) {
        const Vec3 l = ray_light(rays_o, light_off, n);
        for (uint32_t k = 0; k < (uint32_t)rays[n * 2 + 1]; k++) {
            const uint32_t i = (uint32_t)rays[n * 2] + k;
            const Sample p = load_sample(sigma7, dirs, cap, i, e, l);
            sample_forward(p, ratio, mode, mode == kLambertian ? albedo + (size_t)i * 3 : nullptr, color + (size_t)i * 3, orient[i]);
            normal[(size_t)i * 3 + 0] = p.n.x; normal[(size_t)i * 3 + 1] = p.n.y; normal[(size_t)i * 3 + 2] = p.n.z;
        }
    }
}

void hm_shade_backward(const float* sigma7, const float* albedo, const float* dirs, const int32_t* rays, const float* rays_o,
                       const float* light_off, float ratio, int mode, float e, uint32_t cap, uint32_t n_rays,
                       const float* dcolor, const float* dorient, float* dsigma7, float* dalbedo) {
    using namespace sdfx::shade;
    for (uint32_t n = 0; n < n_rays; n++) {
        const Vec3 l = ray_light(rays_o, light_off, n);
        for (uint32_t k = 0; k < (uint32_t)rays[n * 2 + 1]; k++) {
            const uint32_t i = (uint32_t)rays[n * 2] + k;
            const Sample p = load_sample(sigma7, dirs, cap, i, e, l);
            float dsig[6];
            sample_backward(p, l, ratio, mode, mode == kLambertian ? albedo + (size_t)i * 3 : nullptr, dcolor + (size_t)i * 3,
                            nullptr, dorient[i], e, dsig, dalbedo + (size_t)i * 3);
            dsigma7[i] = 0.f;
            for (uint32_t r = 0; r < 6; r++) dsigma7[(size_t)(r + 1) * cap + i] = dsig[r];
        }
    }
}

// One iteration of the device-resident optimiser tail over `tensors` parameter tensors, as csrc/optim.hip runs it:
// k_grad_stats (sum of squares in double + non-finite flag), k_adan_prepare, k_adan_update (same source: optim_math.h).
void hm_adan_iteration(uint32_t tensors, float** params, const float** grads, float** m, float** v, float** nn, float** prev,
                       const uint64_t* counts, const float* lrs, const float* wds, float* ctl, float b1, float b2, float b3,
                       float max_grad_norm, float eps, float growth, float backoff, float growth_interval, int no_prox) {
    using namespace sdfx::optim;
    double sumsq = 0.0, bad = 0.0;
    for (uint32_t t = 0; t < tensors; t++)
        for (uint64_t i = 0; i < counts[t]; i++) {
            const float g = grads[t][i];
            if (!(fabsf(g) <= 3.402823466e38f)) bad = 1.0;
            sumsq += (double)g * (double)g;
        }
    adan_prepare(ctl, sumsq, bad, b1, b2, b3, max_grad_norm, eps, growth, backoff, growth_interval);
    if (ctl[5] != 0.f) return;
    const float unscale = ctl[3] * ctl[4];
    const bool first = ctl[2] == 1.0f;
    for (uint32_t t = 0; t < tensors; t++) {
        const AdanHyper h{lrs[t], wds[t], eps, b1, b2, b3, no_prox};
        for (uint64_t i = 0; i < counts[t]; i++)
            adan_one(params[t][i], grads[t][i], m[t][i], v[t][i], nn[t][i], prev[t][i], unscale, first, ctl[6], ctl[7], ctl[8], h);
    }
}

// fixed-point helpers of the binned table-gradient scatter
long long hm_half_to_fixed(uint32_t h) { return half_to_fixed(h); }
float hm_fixed_to_float(long long units) { return fixed_to_float(units); }

// body of k_march_write_tbuf for one ray
void hm_march_write(const float* rays_o, const float* rays_d, float bound, int contract, float dt_gamma,
                    uint32_t max_steps, uint32_t C, uint32_t H, uint32_t n, uint32_t count, const float* tbuf,
                    float* xyzs, float* ts) {
    const MarchParams p = make_march_params(bound, contract, dt_gamma, max_steps, C, H);
    const MarchRay r = make_march_ray(rays_o + (size_t)n * 3, rays_d + (size_t)n * 3);
    for (uint32_t i = 0; i < count; i++) {
        const float t = tbuf[(size_t)n * max_steps + i];
        float cx, cy, cz;
        march_position(r, p, t, cx, cy, cz);
        const float dt = march_dt(p, t);
        xyzs[i * 3 + 0] = cx; xyzs[i * 3 + 1] = cy; xyzs[i * 3 + 2] = cz;
        ts[i * 2 + 0] = t + dt; ts[i * 2 + 1] = dt;
    }
}

// float32 forward of k_grid_forward<3, C=2> for one level (features only)
void hm_grid_forward_d3c2(const float* inputs, const float* table, uint32_t B, uint32_t row0, uint32_t hashmap_size,
                          uint32_t resolution, uint32_t gridtype, int align_corners, uint32_t interp, float* out) {
    for (uint32_t b = 0; b < B; b++) {
        float pos[3], deriv[3];
        uint32_t pg[3];
        bool oob = false;
        for (int d = 0; d < 3; d++) {
            const float v = inputs[(size_t)b * 3 + d];
            if (v < 0 || v > 1) oob = true;
        }
        if (oob) { out[b * 2] = out[b * 2 + 1] = 0; continue; }
        for (int d = 0; d < 3; d++) grid_locate_axis(inputs[(size_t)b * 3 + d], resolution, align_corners != 0, interp, pos[d], deriv[d], pg[d]);
        float res[2] = {0, 0};
        for (uint32_t idx = 0; idx < 8; idx++) {
            float w = 1;
            uint32_t pgl[3];
            for (uint32_t d = 0; d < 3; d++) {
                if ((idx & (1u << d)) == 0) { w *= 1 - pos[d]; pgl[d] = pg[d]; }
                else { w *= pos[d]; pgl[d] = pg[d] + 1 < resolution - 1 ? pg[d] + 1 : resolution - 1; }
            }
            const uint32_t row = grid_row<3>(gridtype, hashmap_size, resolution, pgl);
            res[0] += w * table[(size_t)(row0 + row) * 2 + 0];
            res[1] += w * table[(size_t)(row0 + row) * 2 + 1];
        }
        out[b * 2] = res[0]; out[b * 2 + 1] = res[1];
    }
}

uint32_t hm_morton3D(uint32_t x, uint32_t y, uint32_t z) { return morton3D(x, y, z); }
uint32_t hm_morton3D_invert(uint32_t v) { return morton3D_invert(v); }

}  // extern "C"

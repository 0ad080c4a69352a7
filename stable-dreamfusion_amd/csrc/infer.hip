// infer.hip — the test-time renderer of NeRFRenderer.run_cuda (nerf/renderer.py:759-794) as ONE persistent kernel.
//
// The reference renders a frame with a host loop: march_rays (up to n_step samples per alive ray, raymarching.cu:713-829)
// -> NeRFNetwork.forward (hash-grid encode, MLP, activations) -> composite_rays (raymarching.cu:842-925) -> boolean-mask
// compaction of the alive list (renderer.py:791), n_step = clamp(N / n_alive, 1, 8), until every ray has died or
// max_steps is reached: ~80-130 rounds x (7 launches + one host synchronisation for the alive count), each on buffers sized
// for the 640 000 rays of an 800 x 800 frame. Every ray is independent of every other one, and what a ray does in round r
// depends only on its own state (rays_t, its accumulators) — the round structure exists to batch the field evaluation.
//
// Here one LANE owns one ray from start to finish: march to the next occupied sample (the same march_probe as training
// and as k_march_rays), evaluate the field for that one sample in registers — 16 levels of the hash grid with the
// reference-exact half accumulation (grid_point.h), the 32-64-64-4 MLP on v_dot2 with scalar-loaded weights (field_mlp.h,
// the arithmetic of k_field_forward), trunc_exp / sigmoid / density blob — and composite it (T = 1 - weights_sum,
// raymarching.cu:884-905), until T < T_thresh, the ray leaves the box, or it has taken max_steps samples. No alive list, no
// compaction, no intermediate sample buffers, no host round trips: the samples of a ray never leave its lane.
// Lanes whose ray is finished pull the next unstarted ray from a device-side counter (one wave-aggregated atomic per
// refill), so a wave stays full until the frame runs out of rays: the compaction the reference does between rounds with a
// boolean mask, done by the persistent workgroups themselves.
//
// Shading 'albedo' (what Trainer.eval_step / test_step use, nerf/utils.py:730-775) is fused; the other shadings need the 6
// finite-difference neighbours and keep the host loop. Values: identical operation order per ray as the reference loop,
// so the result matches the host-paced loop run with the v_dot2 field kernels bit for bit up to the last-bit differences
// of expf / __expf ordering that both share (tests/test_gpu_infer.py).
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdlib.h>

#include "sdfx.h"
#include "sdfx_common.h"
#include "field_mlp.h"
#include "grid_point.h"

using namespace sdfx;
using namespace sdfx::grid;
using namespace sdfx::fieldmlp;

namespace {

constexpr uint32_t kThreads = 256;
constexpr uint32_t kLevels = 16;

struct InferArgs {
    const float* rays_o; const float* rays_d; const float* nears; const float* fars; const float* noises;   // noises nullable
    const uint8_t* bitfield;
    MarchParams mp;
    LevelConst lv[kLevels];
    float bound, blob_density, inv_2r2, T_thresh;
    uint32_t n_rays, max_steps, vec16;
    uint32_t* next_ray;          // device counter, starts at 0
    float* weights_sum; float* depth; float* image; int32_t* n_samples;   // n_samples nullable: samples taken per ray
};

// WAVES: minimum waves per SIMD the register allocation must allow (2: 198 VGPRs, 4 levels per gather round; 3 and 4: 124 VGPRs,
// 2 levels per round). The kernel is latency-bound — a sample step is a chain of dependent gather rounds — so occupancy is traded
// against gathers in flight per lane. Measured on an 800 x 800 frame (52 M samples): 37.1 / 31.3 / 31.3 ms for 2 / 3 / 4
// (tools/infer_bench.py, SDFX_INFER_WAVES); default 3.
template <uint32_t INTERP, bool ALIGN, bool HASHGRID, int WAVES>
__global__ __launch_bounds__(kThreads, WAVES) void k_render_infer(InferArgs a, const uint32_t* __restrict__ P,
                                                                       const __half* __restrict__ table) {
    // (P and table are separate __restrict__ parameters, not members of `a`: only then may the compiler assume that the
    // kernel's stores do not alias them and fetch the wave-uniform MLP weights with scalar loads — as members they were
    // 849 vector loads + 1100 v_readlane per sample step)
    __shared__ uint32_t enc_lds[kLevels][kThreads];   // feature pairs of the thread's current sample, [level][thread]
    const int lane = lane_id();
    uint32_t ray = 0xffffffffu;
    bool active = false, exhausted = false;
    MarchRay r = {};
    float t = 0.f, far = 0.f, ws = 0.f, dep = 0.f, cr = 0.f, cg = 0.f, cb = 0.f;
    uint32_t steps = 0;

    for (;;) {
        // ---- refill: lanes without a ray take the next unstarted ones (one atomic per wave) ----
        if (!exhausted) {
            const unsigned long long want = __ballot(!active);
            if (want) {
                const uint32_t n_want = (uint32_t)__popcll(want);
                uint32_t base = 0;
                if (lane == __ffsll((long long)want) - 1) base = atomicAdd(a.next_ray, n_want);
                base = __shfl(base, __ffsll((long long)want) - 1, kWave);
                if (!active) {
                    const uint32_t mine = base + (uint32_t)__popcll(want & ((1ull << lane) - 1ull));
                    if (mine < a.n_rays) {
                        ray = mine; active = true;
                        r = make_march_ray(a.rays_o + (size_t)ray * 3, a.rays_d + (size_t)ray * 3);
                        far = a.fars[ray];
                        t = a.nears[ray];                                   // rays_t = nears.clone(), renderer.py:768
                        if (a.noises) t += clampf_(t * a.mp.dt_gamma, a.mp.dt_min, a.mp.dt_max) * a.noises[ray];   // raymarching.cu:756
                        ws = dep = cr = cg = cb = 0.f;
                        steps = 0;
                    }
                }
                if (base + n_want >= a.n_rays) exhausted = true;             // wave-uniform: the counter has passed the end
            }
        }
        if (!__any(active)) break;

        // ---- march to the next occupied sample (raymarching.cu:760-827) ----
        bool have = false;
        float dt = 0.f, px = 0.f, py = 0.f, pz = 0.f;
        if (active) {
            // Step cap: exactly max_steps samples per ray. The reference's host loop (renderer.py:768-794) counts ROUNDS of
            // n_step = clamp(N / n_alive, 1, 8) samples and stops once the sum of the rounds' n_step reaches max_steps, so a ray
            // that is still alive then has taken between max_steps and max_steps + 7 samples, depending on how many OTHER rays
            // were alive in each round — a frame-global quantity a per-ray kernel cannot (and should not) reproduce. Rays that
            // terminate by T < T_thresh or by leaving the box (all rays of a converged scene) are unaffected.
            while (t < far && steps < a.max_steps) {
                if (march_probe(r, a.mp, a.bitfield, t, dt, px, py, pz)) { have = true; break; }
            }
        }
        bool finished = active && !have;                                     // left the box (or hit the step cap): ray is done

        // ---- field at the sample (network_grid.py:68-78 under autocast) ----
        if (have) {
            t += dt;                                                         // ts = (t after the step, dt)
            steps++;
            Acts acts;
            const float pw[3] = {px, py, pz};
            float x01[3];
#pragma unroll
            for (int d = 0; d < 3; d++) x01[d] = (pw[d] + a.bound) / (2 * a.bound);            // GridEncoder.forward, grid.py:157
            const bool oob = x01[0] < 0 || x01[0] > 1 || x01[1] < 0 || x01[1] > 1 || x01[2] < 0 || x01[2] > 1;
            // kBatch levels at a time: their 4 kBatch gathers are all issued before the first is consumed — a sample step is a
            // chain of dependent memory round trips, so more levels per round = fewer rounds (at ~30 VGPRs a level). The loop
            // over the batches is NOT unrolled (16 unrolled levels x their hashed / dense variants were 160 KB of code, more
            // than the instruction cache); the features go through a per-thread LDS column so that the MLP still gets them
            // in statically indexed registers (a dynamically indexed register array would live in scratch).
            constexpr uint32_t kBatch = WAVES <= 2 ? 4u : 2u;
#pragma unroll 1
            for (uint32_t l0 = 0; l0 < kLevels; l0 += kBatch) {
                LevelPoint lp[kBatch];
                LevelData ld[kBatch];
#pragma unroll
                for (uint32_t j = 0; j < kBatch; j++) level_prepare<INTERP, ALIGN, HASHGRID>(a.lv[l0 + j], x01, lp[j]);
#pragma unroll
                for (uint32_t j = 0; j < kBatch; j++) level_gather(table, a.lv[l0 + j], lp[j], true, ld[j]);
#pragma unroll
                for (uint32_t j = 0; j < kBatch; j++) enc_lds[l0 + j][threadIdx.x] = oob ? 0u : level_reduce(lp[j], ld[j], true);
            }
#pragma unroll
            for (uint32_t l = 0; l < kLevels; l++) acts.enc[l] = as_h2(enc_lds[l][threadIdx.x]);
            mlp_forward(P, acts);
            const float z = acts.h3[0] + a.blob_density * expf(-(px * px + py * py + pz * pz) * a.inv_2r2);
            const float sigma = expf(z);                                     // trunc_exp forward
            const float alb[3] = {1.0f / (1.0f + expf(-acts.h3[1])), 1.0f / (1.0f + expf(-acts.h3[2])), 1.0f / (1.0f + expf(-acts.h3[3]))};
            // ---- composite (raymarching.cu:884-905) ----
            const float alpha = 1.0f - __expf(-sigma * dt);
            const float T = 1 - ws;
            const float w = alpha * T;
            ws += w;
            dep += w * t;
            cr += w * alb[0]; cg += w * alb[1]; cb += w * alb[2];
            if (T < a.T_thresh) finished = true;                             // the reference stops AFTER accumulating this sample
        }
        if (finished) {
            a.weights_sum[ray] = ws; a.depth[ray] = dep;
            a.image[(size_t)ray * 3 + 0] = cr; a.image[(size_t)ray * 3 + 1] = cg; a.image[(size_t)ray * 3 + 2] = cb;
            if (a.n_samples) a.n_samples[ray] = (int32_t)steps;
            active = false;
        }
    }
}

}  // namespace

extern "C" {

int sdfx_render_infer(const float* rays_o, const float* rays_d, const float* nears, const float* fars, const float* noises,
                      const uint8_t* density_bitfield, float bound, int contract, float dt_gamma, uint32_t max_steps, uint32_t N,
                      uint32_t C, uint32_t H, const void* embeddings_half, const int32_t* offsets_host, uint32_t num_levels, float S,
                      uint32_t base_resolution, uint32_t gridtype, int align_corners, uint32_t interp, const uint32_t* field_packed,
                      float blob_density, float blob_radius, float T_thresh, uint32_t* next_ray_counter, float* weights_sum,
                      float* depth, float* image, int32_t* n_samples, sdfx_stream_t stream) {
    SDFX_REQUIRE(rays_o && rays_d && nears && fars && density_bitfield && embeddings_half && offsets_host && field_packed &&
                     next_ray_counter && weights_sum && depth && image, "render_infer: null pointer");
    SDFX_REQUIRE(num_levels == kLevels, "render_infer: the fused field needs the 16-level, 2-feature hash grid of the -O configuration");
    SDFX_REQUIRE(gridtype <= 1 && interp <= 1, "render_infer: bad enum");
    SDFX_REQUIRE(blob_radius > 0.f && bound > 0.f, "render_infer: blob_radius and bound must be positive");
    SDFX_REQUIRE((reinterpret_cast<uintptr_t>(embeddings_half) % 16) == 0, "render_infer: the table must be 16-byte aligned");
    if (N == 0) return SDFX_OK;
    hipStream_t st = as_stream(stream);
    InferArgs a;
    memset(&a, 0, sizeof(a));
    a.rays_o = rays_o; a.rays_d = rays_d; a.nears = nears; a.fars = fars; a.noises = noises; a.bitfield = density_bitfield;
    a.mp = make_march_params(bound, contract, dt_gamma, max_steps, C, H);
    for (uint32_t l = 0; l < kLevels; l++) a.lv[l] = make_level_const(offsets_host, l, S, base_resolution);
    a.bound = bound; a.blob_density = blob_density; a.inv_2r2 = 1.0f / (2.0f * blob_radius * blob_radius);
    a.T_thresh = T_thresh; a.n_rays = N; a.max_steps = max_steps;
    a.vec16 = (reinterpret_cast<uintptr_t>(embeddings_half) % 16) == 0 ? 1u : 0u;
    a.next_ray = next_ray_counter; a.weights_sum = weights_sum; a.depth = depth; a.image = image; a.n_samples = n_samples;
    zero_device(next_ray_counter, sizeof(uint32_t), st);
    const uint32_t* P_ = field_packed;
    const __half* T_ = static_cast<const __half*>(embeddings_half);
    const int waves = [] { const int w = dev_switch("SDFX_INFER_WAVES", 3); return w < 2 ? 2 : (w > 4 ? 4 : w); }();
    // persistent workgroups: enough waves to fill the chip at the chosen occupancy, never more than the rays need
    const uint32_t max_blocks = 256u * (uint32_t)waves;
    const uint32_t blocks = div_up(N, kThreads) < max_blocks ? div_up(N, kThreads) : max_blocks;
#define SDFX_INFER(INTERP_, ALIGN_, HASH_)                                                                                    \
    do {                                                                                                                      \
        if (waves == 2) hipLaunchKernelGGL((k_render_infer<INTERP_, ALIGN_, HASH_, 2>), dim3(blocks), dim3(kThreads), 0, st, a, P_, T_);      \
        else if (waves == 3) hipLaunchKernelGGL((k_render_infer<INTERP_, ALIGN_, HASH_, 3>), dim3(blocks), dim3(kThreads), 0, st, a, P_, T_); \
        else hipLaunchKernelGGL((k_render_infer<INTERP_, ALIGN_, HASH_, 4>), dim3(blocks), dim3(kThreads), 0, st, a, P_, T_);                 \
    } while (0)
    const int sel = (interp ? 4 : 0) | (align_corners ? 2 : 0) | (gridtype == 0 ? 1 : 0);
    switch (sel) {
        case 0: SDFX_INFER(0u, false, false); break;
        case 1: SDFX_INFER(0u, false, true); break;
        case 2: SDFX_INFER(0u, true, false); break;
        case 3: SDFX_INFER(0u, true, true); break;
        case 4: SDFX_INFER(1u, false, false); break;
        case 5: SDFX_INFER(1u, false, true); break;
        case 6: SDFX_INFER(1u, true, false); break;
        default: SDFX_INFER(1u, true, true); break;
    }
#undef SDFX_INFER
    return check_launch("render_infer");
}

}  // extern "C"

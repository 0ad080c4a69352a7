/*
 * sdfx.h — C ABI of libsdfx_hip.so, the MI355X (gfx950) implementation of the Instant-NGP
 * volumetric hot path of ashawkey/stable-dreamfusion.
 *
 * One entry point per function of the reference's pybind `_backend` modules (19 of them):
 *   raymarching  — raymarching/src/raymarching.h:7-18   (bindings.cpp:5-18)
 *   gridencoder  — gridencoder/src/gridencoder.h:11-15  (bindings.cpp:5-8)
 *   shencoder    — shencoder/src/shencoder.h:8-9        (bindings.cpp:5-6)
 *   freqencoder  — freqencoder/src/freqencoder.h:7-10   (bindings.cpp:5-6)
 * plus a small number of extensions the reference realises in Python (stable ray
 * compaction, nerf/renderer.py:791) or that remove a copy at the boundary.
 *
 * Conventions
 *   - Every pointer is a DEVICE pointer (HBM) unless the name ends in `_host`.
 *   - The caller allocates everything; the library never allocates, frees or retains
 *     memory.  Kernels that need working memory take an explicit `scratch` pointer and
 *     report the size they need through the matching `*_scratch_bytes` function.
 *   - `stream` is a hipStream_t (passed as void* so the header needs no HIP include).
 *     All work is enqueued on it; no entry point synchronises the device.
 *   - Return value: 0 on success, a negative SDFX_E_* code otherwise; the message is
 *     available from sdfx_last_error() (thread-local).
 *   - `is_half` selects the table element type of the grid encoder: 0 = float32,
 *     1 = IEEE float16 (what torch autocast feeds the reference, gridencoder/grid.py:46-47).
 *   - Zero-initialisation contracts are the reference's: outputs documented as
 *     "pre-zeroed" are only partially written (raymarching/raymarching.py:249-251,284,
 *     309-310; gridencoder/grid.py:84).
 */
#ifndef SDFX_H_
#define SDFX_H_

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

typedef void* sdfx_stream_t; /* hipStream_t */

enum {
    SDFX_OK = 0,
    SDFX_E_INVALID = -1,     /* bad argument (unsupported D / C / degree, null pointer, ...) */
    SDFX_E_LAUNCH = -2,      /* HIP reported a launch error                                  */
    SDFX_E_UNSUPPORTED = -3  /* combination the reference also rejects                       */
};

const char* sdfx_last_error(void);

/*
 * Extension — padding rows of fixed-capacity sample buffers. An iteration replayed from a HIP graph marches into buffers of a
 * fixed capacity >= the sample total (which only the device knows at replay time); rows past the total are padding. With a limit
 * set (calling thread only; total = NULL clears it), the D = 3, C = 2 hinted encoder forward, the field forward / backward and
 * the binned table-gradient scatter treat row r of a [k, period, ...] batch as padding when (r % period) >= total[0] (period 0:
 * r >= total[0]): they neither read nor write such rows and skip tiles that hold nothing else. total is a DEVICE pointer read
 * when the kernel runs. Set it around calls whose buffers share one sample axis, clear it afterwards.
 */
void sdfx_set_row_limit(const int32_t* total, uint32_t period);
/*
 * Extension — finite-difference stencil batches formed inside the kernels. The -O iteration evaluates the field at each sample
 * and at x +- epsilon along each axis (nerf/network_grid.py:81-96): a [7, M, 3] batch that is a pure function of the M samples.
 * With a source set (calling thread only; xyzs = NULL clears it), the grid-encoder forward / backward (all four kernels), the
 * binned scatter and the field forward / backward take row r of a B = 7 M row batch from xyzs[r % M] and slab r / M — slab 0 the
 * sample, slabs 1..6 = +x, -x, +y, -y, +z, -z with the whole offset point clamped to [-bound, bound] — instead of reading their
 * `inputs` / `x` arguments, which may then be NULL: world coordinates for the density blob, (p + bound) * float(1 / two_bound)
 * for the encoder (gridencoder/grid.py:157 as PyTorch evaluates it). Bit-identical to passing the tensors
 * sdfx_field_stencil_points writes. xyzs is a DEVICE pointer [M, 3] float32 read when the kernel runs; B must equal 7 M.
 */
void sdfx_set_stencil_source(const float* xyzs, uint32_t M, float epsilon, float bound, double two_bound);
/* version / build info string (arch, git-less) */
const char* sdfx_build_info(void);

/* Whether the dispatcher deals the workgroups of a launch to the eight XCDs round-robin (workgroup b -> XCD (b + c) mod 8), which
 * is what the level-per-XCD work plans of the encoder kernels assume for L2 residency of the tables. One probe launch on the
 * current device at the first call FOR THAT DEVICE (reads HW_REG_XCC_ID per workgroup; allocates and launches on the null stream:
 * call it once outside any stream capture), cached per device: 1 = holds, 0 = does not, -1 = probe failed.
 * Results never depend on it; bench.py reports it and tests/test_gpu_02_parity.py asserts it on the hardware under test. */
int sdfx_xcd_round_robin(void);

/* ------------------------------------------------------------------ raymarching: utils */

/* raymarching.cu:148-156 near_far_from_aabb(rays_o, rays_d, aabb, N, min_near, nears, fars) */
int sdfx_near_far_from_aabb(const float* rays_o, const float* rays_d, const float* aabb, uint32_t N, float min_near,
                            float* nears, float* fars, sdfx_stream_t stream);

/* raymarching.cu:201-209 sph_from_ray(rays_o, rays_d, radius, N, coords[N,2]) */
int sdfx_sph_from_ray(const float* rays_o, const float* rays_d, float radius, uint32_t N, float* coords,
                      sdfx_stream_t stream);

/* raymarching.cu:229-232 morton3D(coords[N,3] i32, N, indices[N] i32) */
int sdfx_morton3D(const int32_t* coords, uint32_t N, int32_t* indices, sdfx_stream_t stream);

/* raymarching.cu:257-260 morton3D_invert(indices[N], N, coords[N,3]) */
int sdfx_morton3D_invert(const int32_t* indices, uint32_t N, int32_t* coords, sdfx_stream_t stream);

/* raymarching.cu:292-300 packbits(grid[C*H^3] f32, N = C*H^3/8, density_thresh, bitfield[N] u8) */
int sdfx_packbits(const float* grid, uint32_t N, float density_thresh, uint8_t* bitfield, sdfx_stream_t stream);

/* raymarching.cu:321-326 flatten_rays(rays[N,2], N, M, res[M] pre-zeroed) */
int sdfx_flatten_rays(const int32_t* rays, uint32_t N, uint32_t M, int32_t* res, sdfx_stream_t stream);

/* ---------------------------------------------------------------- raymarching: training */

/*
 * raymarching.cu:477-489 march_rays_train.  Same two-call protocol as the reference
 * (raymarching/raymarching.py:240-254):
 *   pass 1: xyzs == dirs == ts == NULL — counts the occupied samples of every ray (at most
 *           max_steps), writes rays[n] = (offset, count) and adds the total to counter[0].
 *           Offsets are the exclusive prefix sum of the counts in ray order (the reference
 *           hands them out with atomicAdd in completion order; any consumer addresses
 *           samples through `rays`, so only the assignment differs, deterministically).
 *   pass 2: xyzs/dirs/ts [M,3],[M,3],[M,2] pre-zeroed — writes the samples.
 * `scratch` (optional, may be NULL): N*max_steps floats. When the same scratch is given to
 * both passes, pass 1 records the sample positions and pass 2 becomes a coalesced,
 * sample-parallel write instead of a second serial march.
 */
int sdfx_march_rays_train(const float* rays_o, const float* rays_d, const uint8_t* grid, float bound, int contract,
                          float dt_gamma, uint32_t max_steps, uint32_t N, uint32_t C, uint32_t H, const float* nears,
                          const float* fars, float* xyzs, float* dirs, float* ts, int32_t* rays, int32_t* counter,
                          const float* noises, float* scratch, sdfx_stream_t stream);

/*
 * Extension — pass 2 of march_rays_train for an iteration with a fixed sample capacity (HIP-graph replay): writes the samples
 * recorded in `scratch` by pass 1 into xyzs/dirs/ts [capacity, .] (NOT pre-zeroed: rows from counter[0] to capacity are zeroed
 * here, as the pre-zeroed buffers of raymarching/raymarching.py:243-245 would leave them), and copies rays_o, rays_d [N,3],
 * rays [N,2], counter[0] (as int32 and as float32) into the out_* buffers, so that pass 1 of the next iteration may overwrite
 * its inputs while this iteration trains. One launch.
 */
int sdfx_march_rays_train_stage_write(const float* rays_o, const float* rays_d, float bound, int contract, float dt_gamma,
                                      uint32_t max_steps, uint32_t N, uint32_t C, uint32_t H, const int32_t* rays,
                                      const int32_t* counter, const float* scratch, uint32_t capacity, float* xyzs, float* dirs,
                                      float* ts, float* out_rays_o, float* out_rays_d, int32_t* out_rays, int32_t* out_total,
                                      float* out_n_valid, sdfx_stream_t stream);
uint64_t sdfx_march_rays_train_scratch_bytes(uint32_t N, uint32_t max_steps);

/* raymarching.cu:582-590 composite_rays_train_forward (weights [M] pre-zeroed) */
int sdfx_composite_rays_train_forward(const float* sigmas, const float* rgbs, const float* ts, const int32_t* rays,
                                      uint32_t M, uint32_t N, float T_thresh, int binarize, float* weights,
                                      float* weights_sum, float* depth, float* image, sdfx_stream_t stream);

/* raymarching.cu:698-706 composite_rays_train_backward (grad_sigmas [M], grad_rgbs [M,3] pre-zeroed) */
int sdfx_composite_rays_train_backward(const float* grad_weights, const float* grad_weights_sum, const float* grad_depth,
                                       const float* grad_image, const float* sigmas, const float* rgbs, const float* ts,
                                       const int32_t* rays, const float* weights_sum, const float* depth,
                                       const float* image, uint32_t M, uint32_t N, float T_thresh, int binarize,
                                       float* grad_sigmas, float* grad_rgbs, sdfx_stream_t stream);

/* --------------------------------------------------------------- raymarching: inference */

/* raymarching.cu:832-839 march_rays (xyzs/dirs/ts [n_alive*n_step, ...] pre-zeroed) */
int sdfx_march_rays(uint32_t n_alive, uint32_t n_step, const int32_t* rays_alive, const float* rays_t,
                    const float* rays_o, const float* rays_d, float bound, int contract, float dt_gamma,
                    uint32_t max_steps, uint32_t C, uint32_t H, const uint8_t* grid, const float* nears,
                    const float* fars, float* xyzs, float* dirs, float* ts, const float* noises, sdfx_stream_t stream);

/* raymarching.cu:928-934 composite_rays (mutates rays_alive, rays_t, weights_sum, depth, image) */
int sdfx_composite_rays(uint32_t n_alive, uint32_t n_step, float T_thresh, int binarize, int32_t* rays_alive,
                        float* rays_t, const float* sigmas, const float* rgbs, const float* ts, float* weights_sum,
                        float* depth, float* image, sdfx_stream_t stream);

/*
 * Extension — stable compaction of the alive-ray list.  The reference does this with a
 * boolean mask in Python, `rays_alive = rays_alive[rays_alive >= 0]` (nerf/renderer.py:791).
 * out[0..count) = the entries of in[0..n) that are >= 0, order preserved; count_out[0] = count.
 * scratch: sdfx_compact_rays_scratch_bytes(n) bytes.
 */
int sdfx_compact_rays(const int32_t* rays_alive_in, uint32_t n, int32_t* rays_alive_out, int32_t* count_out,
                      void* scratch, sdfx_stream_t stream);
uint64_t sdfx_compact_rays_scratch_bytes(uint32_t n);

/*
 * Extension — the occupancy-grid refresh of NeRFRenderer.update_extra_state (nerf/renderer.py:1102-1149) without index tensors
 * and without reading the mean density back to the host (csrc/occupancy.hip):
 *   sdfx_occupancy_points   xyzs[m] = jittered sample position of cell m of the MORTON-ordered grid of one cascade
 *                           (renderer.py:1123-1133; bound_cascade = min(2^cascade, bound)); jitter from `noise` [H^3, 3]
 *                           (in the reference's meshgrid order n = (x H + y) H + z) or, when NULL, from Philox4x32-10 keyed
 *                           by (seed, cascade) with counter n. The density of point m is then the new value of cell m.
 *   sdfx_occupancy_update   grid = max(grid * decay, sigmas) where grid >= 0 (renderer.py:1137-1139); stats[0] += sum,
 *                           stats[1] += count of those cells (doubles; zeroed first when reset_stats != 0). `stats` holds
 *                           sdfx_occupancy_stats_doubles() doubles, zero before the first use: behind the two totals sit the
 *                           per-workgroup partials, added up in a fixed order (the mean is bit-reproducible).
 *   sdfx_occupancy_pack     bitfield = packbits(grid, min(stats[0] / stats[1], density_thresh)) (renderer.py:1140-1147,
 *                           raymarching.cu:267-300); mean_out[0] (optional) = the mean, for reporting.
 */
uint32_t sdfx_occupancy_stats_doubles(void);
int sdfx_occupancy_points(uint32_t H, double bound_cascade, const float* noise, uint64_t seed, uint32_t cascade, float* xyzs,
                          sdfx_stream_t stream);
int sdfx_occupancy_update(float* density_grid_cascade, const float* sigmas, uint32_t n_cells, float decay, double* stats,
                          int reset_stats, sdfx_stream_t stream);
int sdfx_occupancy_pack(const float* density_grid, uint32_t n_cells, const double* stats, float density_thresh, uint8_t* bitfield,
                        float* mean_out, sdfx_stream_t stream);

/*
 * Extension — the test-time renderer of NeRFRenderer.run_cuda (nerf/renderer.py:759-794: march_rays -> field -> composite_rays
 * -> mask compaction, in a host loop) as ONE persistent kernel for 'albedo' shading and the -O field (16 x 2 half hash grid,
 * 32-64-64-4 MLP): one lane per ray from start to finish, finished lanes pull the next ray from `next_ray_counter` (device,
 * 4 bytes; zeroed by the call). Per ray the operations and their order are the reference loop's.
 *   rays_o / rays_d [N, 3], nears / fars [N]; noises [N] or NULL (the perturb=True start jitter of raymarching.cu:756-757);
 *   embeddings_half: the float16 table; offsets_host: L + 1 level offsets on the host; field_packed: sdfx_field_pack output;
 *   out: weights_sum / depth [N], image [N, 3] (without background), n_samples [N] or NULL (samples taken per ray).
 */
int sdfx_render_infer(const float* rays_o, const float* rays_d, const float* nears, const float* fars, const float* noises,
                      const uint8_t* density_bitfield, float bound, int contract, float dt_gamma, uint32_t max_steps, uint32_t N,
                      uint32_t C, uint32_t H, const void* embeddings_half, const int32_t* offsets_host, uint32_t num_levels, float S,
                      uint32_t base_resolution, uint32_t gridtype, int align_corners, uint32_t interp, const uint32_t* field_packed,
                      float blob_density, float blob_radius, float T_thresh, uint32_t* next_ray_counter, float* weights_sum,
                      float* depth, float* image, int32_t* n_samples, sdfx_stream_t stream);

/* ------------------------------------------------------------------------- gridencoder */

/* Host-only (no GPU work): the per-XCD work list sdfx_grid_encode_forward_hint would use for these arguments, 4 integers per
 * segment (xcd, level, first tile, tiles); an XCD walks its segments in order. Returns the number of segments, < 0 on error. */
int sdfx_grid_forward_plan(const int32_t* offsets_host, uint32_t max_level, float S, uint32_t H, int is_half, uint32_t B,
                           uint32_t slabs, float step, int32_t* segments, uint32_t max_segments, uint32_t* tiles_per_level);
/* Host-only: costs[l] (l < max_level) = what the plan above prices a tile of level l at, in table lines looked up by a wave: the
 * model max(distinct lines per wave at the step hint, VALU floor), or — for the one configuration it was measured on, the -O grid at
 * the iteration's step — that model corrected by a per-XCD timeline; 1.0 for every level without a step hint. Returns max_level. */
int sdfx_grid_forward_level_costs(const int32_t* offsets_host, uint32_t max_level, float S, uint32_t H, uint32_t slabs, float step,
                                  double* costs);

/* Host-only (no GPU work): the per-XCD item ranges the binned backward's first kernel would use for a batch of B points:
 * ranges[2k], ranges[2k + 1] = [start, end) of XCD k over the (virtual level, tile) items in virtual-level-major order,
 * *tiles = tiles per level. balance 0: equal counts (default), 1: cut by the per-level cost table (env SDFX_GRIDBWD_BALANCE=1
 * selects it for the kernel; SDFX_GRIDBWD_LEVEL_COST overrides the table). Returns the number of levels, < 0 on error. */
int sdfx_grid_backward_plan(const int32_t* offsets_host, uint32_t max_level, float S, uint32_t H, uint32_t B, int balance,
                            int32_t* ranges, uint32_t* tiles);

/*
 * gridencoder.cu:467-490 grid_encode_forward.
 *   inputs [B,D] f32 in [0,1]; embeddings [sum, C] f32|f16; offsets [L+1] i32 (device);
 *   offsets_host: the same L+1 values in host memory (the launch plan needs the level sizes;
 *   passing them avoids a device->host copy and keeps the call asynchronous);
 *   outputs [L,B,C] (out_layout 0, the reference's) or [B,L*C] (out_layout 1, the permute of
 *   gridencoder/grid.py:64 folded into the store); dy_dx [B, L*D*C] or NULL.
 *   S = log2(per_level_scale); H = base resolution; gridtype 0 hash / 1 tiled; interp 0 linear / 1 smoothstep.
 */
int sdfx_grid_encode_forward(const float* inputs, const void* embeddings, const int32_t* offsets,
                             const int32_t* offsets_host, void* outputs, uint32_t B, uint32_t D, uint32_t C, uint32_t L,
                             uint32_t max_level, float S, uint32_t H, void* dy_dx, uint32_t gridtype, int align_corners,
                             uint32_t interp, int is_half, int out_layout, sdfx_stream_t stream);

/*
 * Extension — sdfx_grid_encode_forward with locality hints; the outputs are identical, only speed depends on them.
 *   slabs  1, or 7: the B points are 7 slabs of B/7 points and points i, i + B/7, ..., i + 6B/7 are the finite-difference
 *          stencil of one sample (x, x +- e along each axis; nerf/network_grid.py:81-96 evaluated in one call). The seven
 *          points are then processed by neighbouring lanes, which share table lines at every level.
 *   step   > 0: expected distance, in input units ([0,1]), between consecutive points of a slab when they are consecutive
 *          samples of a ray (dt_min / (2 bound)); < 0: consecutive points walk a space-filling curve through a regular grid of
 *          points |step| apart, so that 64 consecutive points are a 4 x 4 x 4 block (the Morton-ordered cell centres of the
 *          occupancy refresh: step = -1 / grid size); 0 = unknown. step > 0 drives the per-XCD split of the levels (cost model);
 *          step < 0 selects the hinted kernel with an even split.
 */
int sdfx_grid_encode_forward_hint(const float* inputs, const void* embeddings, const int32_t* offsets,
                             const int32_t* offsets_host, void* outputs, uint32_t B, uint32_t D, uint32_t C, uint32_t L,
                             uint32_t max_level, float S, uint32_t H, void* dy_dx, uint32_t gridtype, int align_corners,
                             uint32_t interp, int is_half, int out_layout, uint32_t slabs, float step,
                                  sdfx_stream_t stream);

/*
 * gridencoder.cu:492-522 grid_encode_backward.  grad [L,B,C] (grad_layout 0) or [B,L*C]
 * (grad_layout 1); grad_embeddings pre-zeroed, same type as the table; dy_dx / grad_inputs
 * NULL unless the forward saved dy_dx.
 */
int sdfx_grid_encode_backward(const void* grad, const float* inputs, const void* embeddings, const int32_t* offsets,
                              const int32_t* offsets_host, void* grad_embeddings, uint32_t B, uint32_t D, uint32_t C,
                              uint32_t L, uint32_t max_level, float S, uint32_t H, const void* dy_dx, void* grad_inputs,
                              uint32_t gridtype, int align_corners, uint32_t interp, int is_half, int grad_layout,
                              sdfx_stream_t stream);

/*
 * Extension — the same table gradient as sdfx_grid_encode_backward for D = 3, C = 2 (no dy_dx), computed
 * by binning contributions per 2048-row bucket and reducing each bucket in LDS instead of issuing one
 * device-scope atomic per (sample, level, corner). `scratch`: device memory of `scratch_bytes` bytes,
 * 16-byte aligned; samples are processed in chunks that fit it (see *_scratch_bytes for sizing). Its
 * contents need no initialisation and are not preserved between calls.
 * Half tables: every row receives the EXACT sum of its half-rounded contributions (64-bit fixed point),
 * rounded once when it is added to grad_embeddings — for any distribution of the samples (list overflow
 * goes to exact per-bucket spill accumulators) and independent of the order of arrival: bit-reproducible.
 * Float tables: float32 sums per bucket; heavy or overflowing buckets fall back to float atomics.
 * Returns SDFX_E_UNSUPPORTED for other D / C so the caller can use sdfx_grid_encode_backward.
 * *_stats (synchronises `stream`), out[4]: [0] = buckets of the last launch on `scratch` that overflowed into their spill
 * accumulator, [1] = 0 (reserved), [2] / [3] = buckets / launches that overflowed on this device since the library was loaded.
 */
int sdfx_grid_encode_backward_binned(const void* grad, const float* inputs, const int32_t* offsets_host,
                                     void* grad_embeddings, uint32_t B, uint32_t D, uint32_t C, uint32_t L,
                                     uint32_t max_level, float S, uint32_t H, uint32_t gridtype, int align_corners,
                                     uint32_t interp, int is_half, int grad_layout, void* scratch, uint64_t scratch_bytes,
                                     sdfx_stream_t stream);
uint64_t sdfx_grid_encode_backward_binned_scratch_bytes(const int32_t* offsets_host, uint32_t L, uint32_t max_level, float S,
                                                         uint32_t H, uint32_t chunk_points, int is_half);
int sdfx_grid_encode_backward_binned_stats(const void* scratch, uint32_t* out, sdfx_stream_t stream);

/* gridencoder.cu:662-668 grad_total_variation (adds into grad; f32 or f16 by is_half) */
int sdfx_grad_total_variation(const void* inputs, const void* embeddings, void* grad, const int32_t* offsets,
                              const int32_t* offsets_host, float weight, uint32_t B, uint32_t D, uint32_t C, uint32_t L,
                              float S, uint32_t H, uint32_t gridtype, int align_corners, int is_half,
                              sdfx_stream_t stream);

/* gridencoder.cu:705-713 grad_weight_decay (B = table rows; adds into grad) */
int sdfx_grad_weight_decay(const void* embeddings, void* grad, const int32_t* offsets, float weight, uint32_t B,
                           uint32_t C, uint32_t L, int is_half, sdfx_stream_t stream);

/* -------------------------------------------------------------------------- freqencoder */

/* freqencoder.cu:97-110 freq_encode_forward(inputs[B,D], B, D, deg, C = D + 2*D*deg, outputs[B,C]) */
int sdfx_freq_encode_forward(const float* inputs, uint32_t B, uint32_t D, uint32_t deg, uint32_t C, float* outputs,
                             sdfx_stream_t stream);

/* freqencoder.cu:113-129 freq_encode_backward(grad[B,C], outputs[B,C], ..., grad_inputs[B,D]) */
int sdfx_freq_encode_backward(const float* grad, const float* outputs, uint32_t B, uint32_t D, uint32_t deg, uint32_t C,
                              float* grad_inputs, sdfx_stream_t stream);

/* ---------------------------------------------------------------------------- shencoder */

/* shencoder.cu:399-417 sh_encode_forward(inputs[B,3], outputs[B,C*C], B, D=3, C=degree<=8, dy_dx[B,3*C*C]|NULL) */
int sdfx_sh_encode_forward(const float* inputs, float* outputs, uint32_t B, uint32_t D, uint32_t C, float* dy_dx,
                           sdfx_stream_t stream);

/* shencoder.cu:419-439 sh_encode_backward(grad[B,C*C], inputs, B, D, C, dy_dx, grad_inputs[B,3] pre-zeroed, accumulated into) */
int sdfx_sh_encode_backward(const float* grad, const float* inputs, uint32_t B, uint32_t D, uint32_t C,
                            const float* dy_dx, float* grad_inputs, sdfx_stream_t stream);

/* ------------------------------------------------------------------------------- field */

/*
 * Extension — the field's "tiny MLP" fused into one kernel pair. Replaces, for the fp16-autocast -O
 * configuration, nerf/network_grid.py:13-32 (sigma_net: Linear(32,64) ReLU Linear(64,64) ReLU Linear(64,4))
 * plus the output activations of common_forward (:68-78): sigma = exp(h0 + density_blob(x)),
 * albedo = sigmoid(h1..3); density_blob = nerf/renderer.py:338-349 ('exp' branch); the backward of the
 * exp clamps its argument at 15 (activation.py:13-16). fp16 inputs and weights, float32 accumulation,
 * layer outputs rounded to fp16 — the arithmetic of the autocast GEMMs it stands in for.
 *
 *   enc        fp16 features, enc_layout 0: [16, B, 2] (the encoder's level-major layout), 1: [B, 32]
 *   x          [B, 3] float32 sample positions (world coordinates, for the density blob)
 *   packed     sdfx_field_packed_words() 32-bit words written by sdfx_field_pack from the six float32
 *              torch parameters (w1 [64,32], b1 [64], w2 [64,64], b2 [64], w3 [4,64], b3 [4])
 *   sigma [B], albedo [B,3] float32 outputs; dsigma / dalbedo their gradients
 *   denc       gradient of the features, fp16, same layout as enc
 *   scratch    sdfx_field_backward_scratch_bytes(B) bytes (per-workgroup weight-gradient partial sums)
 *   dw1..db3   float32 parameter gradients (overwritten)
 */
uint32_t sdfx_field_packed_words(void);
uint64_t sdfx_field_backward_scratch_bytes(uint32_t B);
/* Extension — the 7-point finite-difference stencil batch of network_grid.py:81-96 from the M sample positions: points [7, M, 3]
 * = (x, x + eps e_x, x - eps e_x, ... e_z), the six offset points clamped to [-bound, bound] (network_grid.py:84-89), and
 * unit [7, M, 3] = (points + bound) * float32(1 / two_bound), the encoder's input (gridencoder/grid.py:157) as PyTorch evaluates
 * tensor / scalar; two_bound = 2 * bound in double precision. */
int sdfx_field_stencil_points(const float* xyzs, uint32_t M, float epsilon, float bound, double two_bound, float* points, float* unit,
                              sdfx_stream_t stream);
int sdfx_field_pack(const float* w1, const float* b1, const float* w2, const float* b2, const float* w3, const float* b3,
                    uint32_t* packed, sdfx_stream_t stream);
/* sdfx_set_albedo_rows(rows): until reset with 0, the `albedo` output of sdfx_field_forward and the `dalbedo` input of
 * sdfx_field_backward hold only the FIRST `rows` rows of the batch (this thread's calls): albedo is not stored for the rows behind them
 * and their d-albedo is taken as zero. A 7-point stencil batch [7, M, ...] needs the albedo of its M base samples only.
 * sdfx_field_albedo_rows_ok(B, enc_layout): 1 when both calls honour it for such a batch (the [L, B, 2]-layout kernels). */
void sdfx_set_albedo_rows(uint32_t rows);
int sdfx_field_albedo_rows_ok(uint32_t B, int enc_layout);
int sdfx_field_forward(const void* enc, int enc_layout, const float* x, const uint32_t* packed, uint32_t B,
                       float blob_density, float blob_radius, float* sigma, float* albedo, sdfx_stream_t stream);
int sdfx_field_backward(const void* enc, int enc_layout, const float* x, const uint32_t* packed, uint32_t B,
                        float blob_density, float blob_radius, const float* dsigma, const float* dalbedo, void* denc,
                        float* scratch, float* dw1, float* db1, float* dw2, float* db2, float* dw3, float* db3,
                        sdfx_stream_t stream);

/* ------------------------------------------------------ normal / shading glue (extension) */

/*
 * Extension — the part of NeRFNetwork.forward between the field and the compositor (nerf/network_grid.py:81-130:
 * finite-difference normal from the six neighbour densities, safe_normalize + nan_to_num, Lambertian / textureless /
 * normal shading), the view-direction normalisation (nerf/renderer.py:734) and the per-sample factor
 * clamp(normal . dir, 0)^2 of loss_orient (renderer.py:744-746). ~100 elementwise PyTorch launches per iteration
 * in the reference; one kernel each way here (one wavefront per ray: the light direction
 * safe_normalize(rays_o[n] + light_offset) is per ray, renderer.py:727).
 *
 *   sigma7 [7, capacity] f32 (x, x+e_x, x-e_x, x+e_y, x-e_y, x+e_z, x-e_z); albedo [capacity, 3] (mode 1 only);
 *   dirs [capacity, 3] un-normalised; rays [n_rays, 2] (offset, count); rays_o [n_rays, 3]; light_offset [3];
 *   ratio: device scalar (ambient ratio); total: device int32, rows >= total are padding (outputs zeroed);
 *   mode 1 = lambertian (albedo * lambert), 2 = textureless (lambert), 3 = normal ((n + 1) / 2).
 *   backward: dnormal may be NULL; dsigma7 [7, capacity] (row 0 is zero: the centre density only feeds the
 *   compositor); dalbedo [capacity, 3] for mode 1, else NULL.
 */
int sdfx_shade_forward(const float* sigma7, const float* albedo, const float* dirs, const int32_t* rays, const float* rays_o,
                       const float* light_offset, const float* ratio, int mode, float epsilon, uint32_t capacity,
                       uint32_t n_rays, const int32_t* total, float* color, float* normal, float* orient,
                       sdfx_stream_t stream);
int sdfx_shade_backward(const float* sigma7, const float* albedo, const float* dirs, const int32_t* rays, const float* rays_o,
                        const float* light_offset, const float* ratio, int mode, float epsilon, uint32_t capacity,
                        uint32_t n_rays, const int32_t* total, const float* dcolor, const float* dnormal, const float* dorient,
                        float* dsigma7, float* dalbedo, sdfx_stream_t stream);

/*
 * Extension — shading + compositing + regulariser sums of a training iteration in one kernel each way (csrc/render.hip):
 * sdfx_shade_forward -> sdfx_composite_rays_train_forward -> sdfx_entropy_forward and the orientation term of
 * nerf/renderer.py:744-746, with the same arithmetic (csrc/shade_math.h; raymarching.cu:500-706), one wavefront per ray.
 *   sigma7 [7, capacity] densities at x, x+e_x, x-e_x, x+e_y, x-e_y, x+e_z, x-e_z; albedo [capacity, 3] at x; dirs [capacity, 3]
 *   un-normalised; ts [capacity, 2]; rays [n_rays, 2] (offset, count); rays_o [n_rays, 3]; light_offset [3]; ratio: device
 *   scalar; mode_dev: device scalar holding 1 / 2 / 3 as a float (lambertian / textureless / normal) or NULL -> `mode`;
 *   total [1]: number of valid rows (rows in [total, capacity) are padding: zero weight, zero gradient).
 *   forward out: weights [capacity], weights_sum / depth [n_rays], image [n_rays, 3],
 *                ray_sums [n_rays, 2] = (sum_i H(clamp(w_i, 1e-5, 1 - 1e-5)) in bits, sum_i w_i clamp(n_i . d_i, 0)^2) over the ray's samples.
 *   backward in: the forward's weights_sum / depth / image and the gradients of weights_sum, depth (NULL = 0), image and
 *                ray_sums (NULL = 0); out: dsigma7 [7, capacity], dalbedo [capacity, 3] (every row written).
 */
int sdfx_render_train_forward(const float* sigma7, const float* albedo, const float* dirs, const float* ts, const int32_t* rays,
                              const float* rays_o, const float* light_offset, const float* ratio, const float* mode_dev,
                              int mode, float epsilon, float T_thresh, uint32_t capacity, uint32_t n_rays, const int32_t* total,
                              float* weights, float* weights_sum, float* depth, float* image, float* ray_sums,
                              sdfx_stream_t stream);
int sdfx_render_train_backward(const float* sigma7, const float* albedo, const float* dirs, const float* ts, const int32_t* rays,
                               const float* rays_o, const float* light_offset, const float* ratio, const float* mode_dev,
                               int mode, float epsilon, float T_thresh, uint32_t capacity, uint32_t n_rays, const int32_t* total,
                               const float* weights_sum, const float* depth, const float* image, const float* grad_weights_sum,
                               const float* grad_depth, const float* grad_image, const float* grad_ray_sums, float* dsigma7,
                               float* dalbedo, sdfx_stream_t stream);

/*
 * Extension — everything between the compositor's per-ray outputs and the guidance network's input in one kernel each way
 * (csrc/head.hip): background = sigmoid(MLP(FreqEncoder_6(rays_d))) (nerf/network_grid.py:132-153; W1 [32, 39], b1 [32], W2 [3, 32],
 * b2 [3] float32) or the colour bg_color [3] when W1 is NULL; image + (1 - weights_sum) * background (nerf/renderer.py:797-806);
 * pred [C, N] = that image (+ weights_sum as 4th channel when C = 4), i.e. the [1, C, H, W] tensor of nerf/utils.py:533-541;
 * loss_reg [1] = lambda_opacity mean(weights_sum^2) + (lambda_entropy ray_sums[:, 0] + lambda_orient ray_sums[:, 1]).sum() / n_valid
 * (nerf/utils.py:563-575, nerf/renderer.py:744-746). lambda_entropy and n_valid are device scalars; ray_sums may be NULL.
 * backward: gradients with respect to image_raw, weights_sum, ray_sums and the four network tensors from grad_pred [C, N] and
 * grad_loss_reg [1] (NULL = 0). scratch: sdfx_head_scratch_bytes(N) bytes.
 */
uint64_t sdfx_head_scratch_bytes(uint32_t N);
int sdfx_head_forward(const float* image_raw, const float* weights_sum, const float* ray_sums, const float* rays_d, const float* W1,
                      const float* b1, const float* W2, const float* b2, const float* bg_color, const float* lambda_entropy,
                      const float* n_valid, float lambda_opacity, float lambda_orient, uint32_t N, uint32_t C, float* pred,
                      float* loss_reg, void* scratch, sdfx_stream_t stream);
int sdfx_head_backward(const float* image_raw, const float* weights_sum, const float* ray_sums, const float* rays_d, const float* W1,
                       const float* b1, const float* W2, const float* b2, const float* bg_color, const float* lambda_entropy,
                       const float* n_valid, float lambda_opacity, float lambda_orient, uint32_t N, uint32_t C, const float* grad_pred,
                       const float* grad_loss_reg, float* grad_image, float* grad_weights_sum, float* grad_ray_sums, float* dW1,
                       float* db1, float* dW2, float* db2, void* scratch, sdfx_stream_t stream);

/*
 * Extension — the elementwise arithmetic of StableDiffusion.train_step either side of the frozen UNet
 * (guidance/sd_utils.py:86-159), B items of per_item = C * h * w latent elements each.
 * sdfx_sds_add_noise: latents = x * 2 - 1 if affine (latent phase, float32 x, written to latents_out) else x (float32, or
 * float16 when is_half: the VAE's output); model_input [2B, per_item] float16 = [noisy, noisy] with
 * noisy = sqrt(abar[t]) latents + sqrt(1 - abar[t]) noise (sd_utils.py:104-106), tt [2B] = [t, t] (sd_utils.py:107).
 * noise has the dtype of the latents (torch.randn_like), t is int64 [B], alphas_cumprod float32 [1000].
 * sdfx_sds_loss: noise_pred float16 [2B, pred_per_item] = (unconditional, text) halves, the noise in the first per_item
 * elements of every item (pred_per_item = per_item for Stable Diffusion, 2 per_item for DeepFloyd IF's learned-variance UNet,
 * guidance/if_utils.py:90-93); classifier-free guidance, w(t) = 1 - abar[t],
 * grad = nan_to_num(grad_scale w (eps - noise)), loss[0] = 0.5 sum (latents - (latents - grad))^2 / B (sd_utils.py:111-159) and
 * grad_latents (float32) = out_scale * dloss/dlatents (out_scale = 2 folds the latent phase's x * 2 - 1).
 * sdfx_sds_text_mix: out [2, n] float16 = (uncond, w_front front + w_side side + w_back back), the interpolated text
 * embedding of nerf/utils.py:448-470 stacked under the unconditional one; the weights are device float32 scalars.
 * Float16 intermediates are rounded where PyTorch's tensor expressions round them.
 */
int sdfx_sds_add_noise(const void* x, int is_half, int affine, const void* noise, const int64_t* t, const float* alphas_cumprod,
                       uint32_t B, uint32_t per_item, float* latents_out, void* model_input, int64_t* tt, sdfx_stream_t stream);
int sdfx_sds_loss(const void* noise_pred, const void* noise, const void* latents, int is_half, const int64_t* t,
                  const float* alphas_cumprod, float guidance_scale, float grad_scale, float out_scale, uint32_t B, uint32_t per_item,
                  uint32_t pred_per_item, float* loss, float* grad_latents, sdfx_stream_t stream);
/* Bilinear resampling [planes, H, W] -> [planes, OH, OW] with PyTorch's align_corners=False arithmetic (sd_utils.py:93), optionally
 * followed by encode_imgs' 2 x - 1 (sd_utils.py:285; affine) and the cast to float16 (out_half); the backward is the exact
 * adjoint as a gather (no atomics: deterministic), grad_out float16 (grad_half) or float32, times 2 when affine. */
int sdfx_sds_upsample_forward(const float* x, uint32_t planes, uint32_t H, uint32_t W, uint32_t OH, uint32_t OW, int affine, int out_half,
                              void* out, sdfx_stream_t stream);
int sdfx_sds_upsample_backward(const void* grad_out, int grad_half, uint32_t planes, uint32_t H, uint32_t W, uint32_t OH, uint32_t OW,
                               int affine, float* grad_x, sdfx_stream_t stream);
int sdfx_sds_text_mix(const void* uncond, const void* front, const void* side, const void* back, const float* w_front,
                      const float* w_side, const float* w_back, uint32_t n, void* out, sdfx_stream_t stream);

/*
 * Extension — the entropy regulariser of Trainer.train_step (nerf/utils.py:571-575) on the compositing weights:
 * sum_out[0] = sum over rows < total of H(clamp(w, 1e-5, 1 - 1e-5)), H the binary entropy in bits (double; the
 * caller divides by the sample total for the reference's .mean()); backward: grad_weights = grad_sum[0] * dH/dw
 * inside the clamp range, 0 outside and on padding rows.
 */
int sdfx_entropy_forward(const float* weights, uint32_t capacity, const int32_t* total, double* sum_out, sdfx_stream_t stream);
int sdfx_entropy_backward(const float* weights, uint32_t capacity, const int32_t* total, const float* grad_sum,
                          float* grad_weights, sdfx_stream_t stream);

/* ------------------------------------------------------------ optimiser tail (extension) */

/*
 * Extension — loss scaling + Adan without a host round trip. The reference does this part in Python:
 * torch.cuda.amp.GradScaler (nerf/utils.py:1047-1052: scale(loss).backward(); step(optimizer); update()) and
 * Adan.step (optimizer.py:109-209, update rule :216-261), both of which read device values back to the host
 * every iteration (found_inf; the clip factor at optimizer.py:125-127). These entry points keep that state in
 * device memory so that an iteration has no host dependency after the sample count is known.
 *
 *   ctl    float32[sdfx_adan_ctl_words()]:  [0] loss scale (set before first use)  [1] growth tracker
 *          [2] optimiser steps applied  [3] 1/scale  [4] clip factor  [5] 1 = this iteration overflowed
 *          [6..8] bias corrections  [9] unscaled gradient norm  [10] iterations skipped
 *   stats  float64[sdfx_amp_grad_stats_doubles()], zero before the first use: [0] sum of squares of all (scaled) gradients,
 *          [1] non-finite count, the rest is the kernels' own (per-workgroup partials added up in a fixed order, so that the two
 *          totals do not depend on the order the workgroups ran in)
 *
 * Per iteration: sdfx_amp_grad_stats over all gradient tensors, sdfx_adan_prepare (also applies
 * GradScaler.update()'s growth/back-off to ctl[0] and clears stats), sdfx_adan_update over all parameter
 * tensors (a no-op when ctl[5] != 0, as GradScaler.step() skips optimizer.step()). The tensor lists are HOST
 * arrays of device pointers / element counts / per-tensor lr and weight decay; they travel in the kernel
 * arguments, so one launch covers up to 16 tensors. `grad_is_half` (NULL: none): entry t != 0 says gradient t is stored as float16
 * (the hash table's gradient as the scatter leaves it; float16 -> float32 is exact, so the update equals the one its float32 copy
 * would give). `half_copies` (NULL, or one entry per tensor, each NULL or a float16 buffer of
 * that tensor's size): the update also writes half(p) there — the fp16 table the next forward gathers from (gridencoder/grid.py:46-47
 * casts the whole table every call; an overflowed iteration leaves parameter and copy untouched, so they stay in step).
 */
uint32_t sdfx_adan_ctl_words(void);
uint32_t sdfx_amp_grad_stats_doubles(void);
int sdfx_amp_grad_stats(const void* const* grads, const uint8_t* grad_is_half, const uint64_t* counts, uint32_t tensors, double* stats,
                        sdfx_stream_t stream);
int sdfx_adan_prepare(float* ctl, double* stats, float beta1, float beta2, float beta3, float max_grad_norm, float eps,
                      float growth_factor, float backoff_factor, uint32_t growth_interval, sdfx_stream_t stream);
int sdfx_adan_update(float* const* params, const void* const* grads, const uint8_t* grad_is_half, float* const* exp_avg, float* const* exp_avg_diff,
                     float* const* exp_avg_sq, float* const* pre_grad, void* const* half_copies, const uint64_t* counts, const float* lrs,
                     const float* weight_decays, uint32_t tensors, const float* ctl, float eps, float beta1, float beta2,
                     float beta3, int no_prox, sdfx_stream_t stream);

/* ------------------------------------------------------------ DMTet fine-tune stage (extension; BASELINE configs[4]) */

/*
 * Marching tetrahedra — `class DMTet.__call__` of nerf/renderer.py:94-178 (occupancy masks, torch.unique over the sorted edges of the
 * valid tetrahedra, index remapping, two gathers through the triangle table: ~25 tensor operations and a sort per iteration) as
 * three kernels with IDENTICAL outputs: vertex order, face order, indices, float32 vertex positions.
 *   edges      int32 [E, 2]: the unique edges (a < b) of the WHOLE grid in lexicographic order — static, built once on the host
 *   tets       int32 [F, 4]; tet_edges int32 [F, 6]: position in `edges` of each tetrahedron's edges (0,1) (0,2) (0,3) (1,2) (1,3) (2,3)
 *   count      counts[0..2] = (vertices V, one-triangle tetrahedra F1, two-triangle tetrahedra F2); scratch keeps per-workgroup
 *              offsets for `emit`, which must be called with the same sdf
 *   emit       edge_vid [E] (-1: no vertex), verts [>= V, 3], vert_edges [>= V, 2] (the grid edge of each vertex), faces [>= F1 + 2 F2, 3]:
 *              rows [0, F1) from the one-triangle tetrahedra, then pairs of rows from the two-triangle ones, each in grid order
 *   backward   grad_pos [N, 3] / grad_sdf [N] += gradients of the interpolation v = p_a (-s_b / (s_a - s_b)) + p_b (s_a / (s_a - s_b))
 */
uint64_t sdfx_marching_tets_scratch_bytes(uint32_t E, uint32_t F);
int sdfx_marching_tets_count(const float* sdf, const int32_t* edges, uint32_t E, const int32_t* tets, uint32_t F, void* scratch,
                             int32_t* counts, sdfx_stream_t stream);
int sdfx_marching_tets_emit(const float* pos, const float* sdf, const int32_t* edges, uint32_t E, const int32_t* tets,
                            const int32_t* tet_edges, uint32_t F, const void* scratch, const int32_t* counts, int32_t* edge_vid,
                            float* verts, int32_t* vert_edges, uint32_t cap_verts, int32_t* faces, uint32_t cap_faces,
                            sdfx_stream_t stream);
int sdfx_marching_tets_backward(const float* grad_verts, uint32_t V, const int32_t* vert_edges, const float* pos, const float* sdf,
                                float* grad_pos, float* grad_sdf, sdfx_stream_t stream);

/*
 * Mesh rasterisation — the three nvdiffrast calls of run_dmtet (nerf/renderer.py:900 dr.rasterize, :903-904 dr.interpolate,
 * :932-933 dr.antialias), B = 1. nvdiffrast is a third-party dependency absent from the reference checkout: these follow its
 * published contract (csrc/raster.hip). pos_clip [N, 4], tri int32 [F, 3], rast [H, W, 4] = (u, v, z/w, triangle id + 1; 0 =
 * background), row 0 = NDC y -1. Backward entry points ADD into grad_pos / grad_attr (zero them first) and WRITE grad_rast /
 * grad_color. adj_opp int32 [F, 3]: vertex opposite to edge k (vertices k, k + 1) in the triangle across that edge, -1: none.
 */
uint64_t sdfx_rasterize_scratch_bytes(uint32_t H, uint32_t W);
int sdfx_rasterize_forward(const float* pos_clip, const int32_t* tri, uint32_t N, uint32_t F, uint32_t H, uint32_t W, void* scratch,
                           float* rast, sdfx_stream_t stream);
int sdfx_rasterize_backward(const float* pos_clip, const int32_t* tri, uint32_t N, uint32_t H, uint32_t W, const float* rast,
                            const float* grad_rast, float* grad_pos, sdfx_stream_t stream);
int sdfx_interpolate_forward(const float* attr, const int32_t* tri, uint32_t C, uint32_t H, uint32_t W, const float* rast, float* out,
                             sdfx_stream_t stream);
int sdfx_interpolate_backward(const float* attr, const int32_t* tri, uint32_t C, uint32_t H, uint32_t W, const float* rast,
                              const float* grad_out, float* grad_attr, float* grad_rast, sdfx_stream_t stream);
int sdfx_antialias_forward(const float* color, const float* rast, const float* pos_clip, const int32_t* tri, const int32_t* adj_opp,
                           uint32_t N, uint32_t C, uint32_t H, uint32_t W, float* out, sdfx_stream_t stream);
int sdfx_antialias_backward(const float* color, const float* rast, const float* pos_clip, const int32_t* tri, const int32_t* adj_opp,
                            uint32_t N, uint32_t C, uint32_t H, uint32_t W, const float* grad_out, float* grad_color, float* grad_pos,
                            sdfx_stream_t stream);

/* ---------------------------------------------------------------- frozen prior: GroupNorm + SiLU (extension) */

/*
 * Extension (no reference kernel: the reference gets these layers from diffusers, guidance/sd_utils.py:37-65) —
 * y = act(GroupNorm_G(x) * gamma + beta) on fp16 CHANNELS-LAST activations x[N, HW, C] (NHWC memory), act = SiLU when
 * `silu` != 0 else identity, and its input gradient; gamma / beta fp16, frozen (no parameter gradients). Statistics in
 * float32 / double, combined in a fixed order (bit-reproducible). Needs C % 8 == 0, C % G == 0, C <= 2560, G <= 64.
 * mean_rstd[N, G, 2] float32 receives what the backward needs (NULL when no backward will follow).
 * pre[N, C] fp16 or NULL: the norm is taken of x + pre[n, c] (a convolution's bias + the time-embedding projection of a ResNet
 * block, which then need no elementwise launch of their own); the backward takes the same vector.
 * scratch: sdfx_group_norm_scratch_bytes(N, HW, C, G) bytes of float32, uninitialised.
 * sdfx_add_bias_residual: out = a + b + bias[c] on the same layout (the tail of a ResNet / transformer block); out may alias a or b.
 */
uint64_t sdfx_group_norm_scratch_bytes(uint32_t N, uint32_t HW, uint32_t C, uint32_t G);
int sdfx_group_norm_forward(const void* x, const void* pre, const void* gamma, const void* beta, uint32_t N, uint32_t HW, uint32_t C, uint32_t G,
                            float eps, int silu, void* y, float* mean_rstd, float* scratch, sdfx_stream_t stream);
int sdfx_group_norm_backward(const void* x, const void* pre, const void* dy, const void* gamma, const void* beta, const float* mean_rstd,
                             uint32_t N, uint32_t HW, uint32_t C, uint32_t G, int silu, void* dx, float* scratch, sdfx_stream_t stream);
int sdfx_add_bias_residual(const void* a, const void* b, const void* bias, uint32_t N, uint32_t HW, uint32_t C, void* out, sdfx_stream_t stream);
/* out[rows, n] = x[rows, :n] * gelu(x[rows, n:]) (exact erf GELU; the GEGLU of the UNet's feed-forward blocks), fp16, n % 8 == 0 */
int sdfx_geglu(const void* x, uint64_t rows, uint32_t n, void* out, sdfx_stream_t stream);

/* ---------------------------------------------------------------- frozen prior: 3 x 3 convolutions on the matrix cores (extension) */

/*
 * Extension (no reference kernel: the reference gets these layers from diffusers, guidance/sd_utils.py:37-65) —
 * y[N, Ho, Wo, Cout] = conv3x3(x[N, H, W, Cin], w[Cout, 3, 3, Cin], padding 1, stride 1 | 2) + bias[Cout] + residual[N, Ho, Wo, Cout]
 * on fp16 CHANNELS-LAST maps (NHWC memory; the weight is a channels-last [Cout, Cin, 3, 3] tensor), float32 accumulation, forward
 * only (frozen weights, no gradient): an implicit GEMM on v_mfma_f32_32x32x16_f16 (csrc/conv.hip). bias / residual may be NULL;
 * y may alias residual. `upsample` != 0: the taps walk the nearest-neighbour 2 x upsampling of x ([N, 2H, 2W, Cin], never
 * materialised). Needs Cin % 64 == 0, Cout % 64 == 0, maps below 2 GiB. Small maps split K over workgroups that write float32
 * partials to `scratch` (sdfx_conv3x3_scratch_bytes(...) bytes, 0 = none needed), summed in a fixed order.
 * splitk / tile_rows: 0 = chosen by shape (what callers pass); > 0 force the K split / 64- or 128-row tiles (measurements).
 */
uint64_t sdfx_conv3x3_scratch_bytes(uint32_t N, uint32_t H, uint32_t W, uint32_t Cin, uint32_t Cout, uint32_t stride, uint32_t upsample,
                                    int splitk, int tile_rows);
int sdfx_conv3x3_forward(const void* x, const void* w, const void* bias, const void* residual, uint32_t N, uint32_t H, uint32_t W,
                         uint32_t Cin, uint32_t Cout, uint32_t stride, uint32_t upsample, int splitk, int tile_rows, void* y,
                         float* scratch, sdfx_stream_t stream);
/* The halo form for stride 1 and (upsampled) rows of 8 / 16 / 32 / 64 pixels in whole 128-pixel tiles (sdfx_conv3x3_packed_ok): a tile's halo of
 * one 64-channel chunk is staged once for all 9 taps, and the weights — packed once per frozen tensor by sdfx_conv3x3_pack_weights into
 * MFMA fragment order, Cout * 9 * Cin halves — go from memory straight into operand registers. Same arguments and result (up to the
 * summation order) as sdfx_conv3x3_forward with stride 1. */
int sdfx_conv3x3_packed_ok(uint32_t N, uint32_t H, uint32_t W, uint32_t Cin, uint32_t Cout, uint32_t upsample);
uint64_t sdfx_conv3x3_packed_scratch_bytes(uint32_t N, uint32_t H, uint32_t W, uint32_t Cin, uint32_t Cout, uint32_t upsample, int splitk);
int sdfx_conv3x3_pack_weights(const void* w, uint32_t Cin, uint32_t Cout, void* packed, sdfx_stream_t stream);
int sdfx_conv3x3_packed_forward(const void* x, const void* packed, const void* bias, const void* residual, uint32_t N, uint32_t H, uint32_t W,
                                uint32_t Cin, uint32_t Cout, uint32_t upsample, int splitk, void* y, float* scratch, sdfx_stream_t stream);
/* The same kernel with one tap: y[M, N] = x[M, K] . w[N, K]^T + bias[N] + residual[M, N] (bias / residual may be NULL; y may alias
 * residual) — the small projections of the transformer blocks with their residual sums in the epilogue. K % 64 == 0, N % 64 == 0. */
uint64_t sdfx_linear_scratch_bytes(uint32_t M, uint32_t K, uint32_t N, int splitk, int tile_rows);
int sdfx_linear_forward(const void* x, const void* w, const void* bias, const void* residual, uint32_t M, uint32_t K, uint32_t N, int splitk,
                        int tile_rows, void* y, float* scratch, sdfx_stream_t stream);

/* ---------------------------------------------------------------- frozen prior: attention on the matrix cores (extension) */

/*
 * Extension (no reference kernel: the reference gets these layers from diffusers, guidance/sd_utils.py:37-65) —
 * o[B, Nq, H, D] (contiguous) = softmax(q k^T * scale) v per (batch, head), fp16 with float32 accumulation and statistics, forward
 * only (the UNet of the SDS step takes no gradient): online softmax on v_mfma_f32_32x32x16_f16 (csrc/attention.hip).
 * q[b, n, h, :] is read at q + b * q_strides[0] + n * q_strides[1] + h * q_strides[2] (elements; unit channel stride, every row
 * 16-byte aligned), k / v likewise over Nk keys. D in {40, 80, 160} (the head widths of the SD-1.5 UNet at 8 heads).
 * waves: 0 = workgroup size chosen by shape (what callers pass); 1 | 2 | 4 force it (measurements).
 */
int sdfx_attention_forward(const void* q, const void* k, const void* v, uint32_t B, uint32_t H, uint32_t Nq, uint32_t Nk, uint32_t D,
                           const uint32_t* q_strides, const uint32_t* k_strides, const uint32_t* v_strides, float scale, int waves,
                           void* o, sdfx_stream_t stream);

#ifdef __cplusplus
}
#endif
#endif /* SDFX_H_ */

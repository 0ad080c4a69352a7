// raymarching.hip — gfx950 kernels for occupancy-grid ray marching, volume compositing and
// ray compaction, behind the C ABI of include/sdfx.h.
//
// Written for CDNA4 from the behaviour of the reference's raymarching extension
// (raymarching/src/raymarching.cu; line citations at each kernel), not from its code:
//   * compositing is one 64-lane wavefront per ray with wave-wide product/sum scans instead
//     of one serial thread per ray (4096 rays = 4096 waves, not 64);
//   * the training march counts once, assigns offsets with a deterministic prefix sum
//     (no atomics), and writes samples from a recorded t-buffer with coalesced stores
//     instead of marching every ray a second time;
//   * alive-ray compaction is a ballot/prefix-sum kernel instead of a host-side mask.
#include "sdfx_common.h"

using namespace sdfx;

// =========================================================================================
// utils
// =========================================================================================

// ray / AABB slab test — semantics of raymarching.cu:91-145
__global__ void k_near_far_from_aabb(const float* __restrict__ rays_o, const float* __restrict__ rays_d,
                                     const float* __restrict__ aabb, uint32_t N, float min_near,
                                     float* __restrict__ nears, float* __restrict__ fars) {
    const uint32_t n = blockIdx.x * blockDim.x + threadIdx.x;
    if (n >= N) return;
    const float ox = rays_o[n * 3 + 0], oy = rays_o[n * 3 + 1], oz = rays_o[n * 3 + 2];
    const float dx = rays_d[n * 3 + 0], dy = rays_d[n * 3 + 1], dz = rays_d[n * 3 + 2];
    const float rdx = 1 / dx, rdy = 1 / dy, rdz = 1 / dz;
    const float kMiss = 3.402823466e+38f;  // numeric_limits<float>::max()

    float near = (aabb[0] - ox) * rdx;
    float far = (aabb[3] - ox) * rdx;
    if (near > far) { const float c = near; near = far; far = c; }

    float near_y = (aabb[1] - oy) * rdy;
    float far_y = (aabb[4] - oy) * rdy;
    if (near_y > far_y) { const float c = near_y; near_y = far_y; far_y = c; }

    if (near > far_y || near_y > far) { nears[n] = kMiss; fars[n] = kMiss; return; }
    if (near_y > near) near = near_y;
    if (far_y < far) far = far_y;

    float near_z = (aabb[2] - oz) * rdz;
    float far_z = (aabb[5] - oz) * rdz;
    if (near_z > far_z) { const float c = near_z; near_z = far_z; far_z = c; }

    if (near > far_z || near_z > far) { nears[n] = kMiss; fars[n] = kMiss; return; }
    if (near_z > near) near = near_z;
    if (far_z < far) far = far_z;

    if (near < min_near) near = min_near;
    nears[n] = near;
    fars[n] = far;
}

// far intersection with the background sphere -> (theta, phi) in [-1,1] — raymarching.cu:162-198
__global__ void k_sph_from_ray(const float* __restrict__ rays_o, const float* __restrict__ rays_d, float radius,
                               uint32_t N, float* __restrict__ coords) {
    const uint32_t n = blockIdx.x * blockDim.x + threadIdx.x;
    if (n >= N) return;
    const float ox = rays_o[n * 3 + 0], oy = rays_o[n * 3 + 1], oz = rays_o[n * 3 + 2];
    const float dx = rays_d[n * 3 + 0], dy = rays_d[n * 3 + 1], dz = rays_d[n * 3 + 2];
    const float A = dx * dx + dy * dy + dz * dz;
    const float B = ox * dx + oy * dy + oz * dz;
    const float C = ox * ox + oy * oy + oz * oz - radius * radius;
    const float t = (-B + sqrtf(B * B - A * C)) / A;
    const float x = ox + t * dx, y = oy + t * dy, z = oz + t * dz;
    const float theta = atan2f(sqrtf(x * x + z * z), y);
    const float phi = atan2f(z, x);
    coords[n * 2 + 0] = 2 * theta * kRPi - 1;
    coords[n * 2 + 1] = phi * kRPi;
}

// raymarching.cu:214-226
__global__ void k_morton3D(const int32_t* __restrict__ coords, uint32_t N, int32_t* __restrict__ indices) {
    const uint32_t n = blockIdx.x * blockDim.x + threadIdx.x;
    if (n >= N) return;
    indices[n] = (int32_t)morton3D((uint32_t)coords[n * 3], (uint32_t)coords[n * 3 + 1], (uint32_t)coords[n * 3 + 2]);
}

// raymarching.cu:237-254
__global__ void k_morton3D_invert(const int32_t* __restrict__ indices, uint32_t N, int32_t* __restrict__ coords) {
    const uint32_t n = blockIdx.x * blockDim.x + threadIdx.x;
    if (n >= N) return;
    const int32_t ind = indices[n];
    coords[n * 3 + 0] = (int32_t)morton3D_invert((uint32_t)(ind >> 0));
    coords[n * 3 + 1] = (int32_t)morton3D_invert((uint32_t)(ind >> 1));
    coords[n * 3 + 2] = (int32_t)morton3D_invert((uint32_t)(ind >> 2));
}

// 8 densities -> 1 byte (raymarching.cu:267-289). Each thread reads two float4 (32 B, fully
// coalesced across the wave) and writes one byte; `vec` is false only for a base pointer that
// is not 16-byte aligned (a tensor view with a storage offset).
__global__ void k_packbits(const float* __restrict__ grid, uint32_t N, float thresh, uint8_t* __restrict__ bitfield,
                           int vec) {
    const uint32_t n = blockIdx.x * blockDim.x + threadIdx.x;
    if (n >= N) return;
    float v[8];
    if (vec) {
        const float4 a = reinterpret_cast<const float4*>(grid)[(size_t)n * 2];
        const float4 b = reinterpret_cast<const float4*>(grid)[(size_t)n * 2 + 1];
        v[0] = a.x; v[1] = a.y; v[2] = a.z; v[3] = a.w;
        v[4] = b.x; v[5] = b.y; v[6] = b.z; v[7] = b.w;
    } else {
#pragma unroll
        for (int i = 0; i < 8; i++) v[i] = grid[(size_t)n * 8 + i];
    }
    uint32_t bits = 0;
#pragma unroll
    for (int i = 0; i < 8; i++) bits |= (v[i] > thresh) ? (1u << i) : 0u;
    bitfield[n] = (uint8_t)bits;
}

// ray id per sample (raymarching.cu:303-319). One wave per ray, lanes stride the ray's slice
// so the stores are contiguous.
__global__ void k_flatten_rays(const int32_t* __restrict__ rays, uint32_t N, uint32_t M, int32_t* __restrict__ res) {
    const uint32_t n = (blockIdx.x * blockDim.x + threadIdx.x) >> 6;
    if (n >= N) return;
    const uint32_t offset = (uint32_t)rays[n * 2];
    const uint32_t count = (uint32_t)rays[n * 2 + 1];
    for (uint32_t i = lane_id(); i < count; i += kWave) {
        if (offset + i < M) res[offset + i] = (int32_t)n;
    }
}

// =========================================================================================
// training march
// =========================================================================================

#ifdef SDFX_DEVTOOLS
// Pass 1, thread per ray — the literal shape of the reference's kernel (raymarching.cu:337-475 with xyzs == nullptr), superseded
// by k_march_count_wave below and kept in the devtools library only (SDFX_MARCH_WAVE=0; tools/march_bench.py and the A/B test).
// One wave per workgroup so the 64-ray groups land on as many CUs as possible; every ray
// records the ray time of each emitted sample in tbuf[n, step] when a scratch buffer is given.
__global__ __launch_bounds__(64) void k_march_count(const float* __restrict__ rays_o, const float* __restrict__ rays_d,
                                                     const uint8_t* __restrict__ grid, MarchParams p,
                                                     uint32_t max_steps, uint32_t N, const float* __restrict__ nears,
                                                     const float* __restrict__ fars, const float* __restrict__ noises,
                                                     int32_t* __restrict__ rays, float* __restrict__ tbuf) {
    const uint32_t n = blockIdx.x * blockDim.x + threadIdx.x;
    if (n >= N) return;
    const MarchRay r = make_march_ray(rays_o + (size_t)n * 3, rays_d + (size_t)n * 3);
    const float far = fars[n];
    float t = nears[n];
    t += clampf_(t * p.dt_gamma, p.dt_min, p.dt_max) * noises[n];  // raymarching.cu:389-391
    float* trow = tbuf ? tbuf + (size_t)n * max_steps : nullptr;
    uint32_t step = 0;
    while (t < far && step < max_steps) {
        float dt, cx, cy, cz;
        if (march_probe(r, p, grid, t, dt, cx, cy, cz)) {
            if (trow) trow[step] = t;
            step++;
            t += dt;
        }
    }
    rays[n * 2 + 1] = (int32_t)step;
}
#endif

// Pass 1, one WAVE per ray (the default since round 2: 107-184 us against 250-330 us for the thread-per-ray kernel on
// the init / blobs / full grids, identical counts, offsets and sample times on the MI355X — tests/test_gpu_02_parity.py,
// tools/march_bench.py; SDFX_MARCH_WAVE=0 selects the thread-per-ray kernel in the devtools library). Every ray time the march visits lies on one occupancy-independent lattice (march_advance), so 64 consecutive
// lattice points are probed at once — each lane: is my cell occupied, and if not, how many lattice points does the
// serial march skip from here (the literal do-while of raymarching.cu:459-462)? — and the serial decision chain is
// then replayed over the 64 results with scalar ballots / readlanes. The dependent global loads of the bitfield, which
// bound the thread-per-ray kernel (one ~1 us load per probe, ~100-300 probes per ray), become 64 loads in flight.
__global__ __launch_bounds__(256) void k_march_count_wave(const float* __restrict__ rays_o, const float* __restrict__ rays_d,
                                                          const uint8_t* __restrict__ grid, MarchParams p,
                                                          uint32_t max_steps, uint32_t N, const float* __restrict__ nears,
                                                          const float* __restrict__ fars, const float* __restrict__ noises,
                                                          int32_t* __restrict__ rays, float* __restrict__ tbuf) {
    const uint32_t n = (blockIdx.x * blockDim.x + threadIdx.x) >> 6;
    if (n >= N) return;  // wave-uniform
    const uint32_t lane = (uint32_t)lane_id();
    const MarchRay r = make_march_ray(rays_o + (size_t)n * 3, rays_d + (size_t)n * 3);
    const float far = fars[n];
    float base = nears[n];
    base += clampf_(base * p.dt_gamma, p.dt_min, p.dt_max) * noises[n];  // raymarching.cu:389-391
    float* trow = tbuf ? tbuf + (size_t)n * max_steps : nullptr;
    uint32_t step = 0;
    bool done = false;
    while (!done) {
        // lane j takes lattice point j of this chunk (sequential float adds: the lattice is not a closed form)
        float t = base, tl = base;
        for (uint32_t j = 0; j < kWave; j++) {
            if (j == lane) tl = t;
            t = march_advance(p, t);
        }
        const float t_next_chunk = t;
        const bool live = tl < far;
        bool occ = false;
        uint32_t hop = 1;
        float tafter = tl;
        if (live) {
            float tt = tl, dt, cx, cy, cz;
            uint32_t h = 1;
            occ = march_probe(r, p, grid, tt, dt, cx, cy, cz, &h);
            if (occ) {
                tafter = tl + dt;
            } else {
                hop = h;
                tafter = tt;
            }
        }
        const uint64_t live_mask = __ballot(live), occ_mask = __ballot(occ);
        // the serial decision chain over the chunk (everything below is wave-uniform)
        uint32_t q = 0, emitted = 0;
        uint64_t emit_mask = 0;
        float carry = t_next_chunk;
        for (;;) {
            if (q >= kWave) { base = carry; break; }
            if (!((live_mask >> q) & 1ull) || step + emitted >= max_steps) { done = true; break; }
            if ((occ_mask >> q) & 1ull) { emit_mask |= 1ull << q; emitted++; }
            const uint32_t h = (uint32_t)__shfl((int)hop, (int)q);
            carry = __shfl(tafter, (int)q);
            q += h;
            if (q == kWave) carry = t_next_chunk;
        }
        if (trow && ((emit_mask >> lane) & 1ull)) trow[step + (uint32_t)__popcll(emit_mask & ((1ull << lane) - 1ull))] = tl;
        step += emitted;
    }
    if (lane == 0) rays[n * 2 + 1] = (int32_t)step;
}

// Offsets = exclusive prefix sum of the counts in ray order, starting from counter[0]; the
// total is added to counter[0] (the reference's atomicAdd bookkeeping, raymarching.cu:470-474,
// made deterministic). Single workgroup; N is a few thousand rays on the training path.
__global__ __launch_bounds__(256) void k_scan_counts(int32_t* __restrict__ rays, uint32_t N,
                                                      int32_t* __restrict__ counter) {
    // 256 threads: the prefetched counting pass runs on a side stream BESIDE the training kernels, and a 1024-thread workgroup needs
    // 16 free wave slots on one CU at once — beside k_grid_fwd it sat in the queue for up to 120 us (profiles/r06_iteration_trace_*);
    // four waves find room at once, and 16 rounds of 256 rays cost the same few microseconds as 4 rounds of 1024
    constexpr uint32_t kScanThreads = 256, kScanWaves = kScanThreads / 64;
    __shared__ uint32_t wave_tot[kScanWaves];
    const int lane = lane_id();
    const int wid = (int)(threadIdx.x >> 6);
    uint32_t base = (uint32_t)counter[0];
    for (uint32_t start = 0; start < N; start += kScanThreads) {
        const uint32_t i = start + threadIdx.x;
        const uint32_t c = i < N ? (uint32_t)rays[i * 2 + 1] : 0u;
        const uint32_t incl = wave_incl_sum_u32(c, lane);
        if (lane == kWave - 1) wave_tot[wid] = incl;
        __syncthreads();
        uint32_t woff = 0, tot = 0;
#pragma unroll
        for (int w = 0; w < (int)kScanWaves; w++) {
            const uint32_t v = wave_tot[w];
            if (w < wid) woff += v;
            tot += v;
        }
        if (i < N) rays[i * 2] = (int32_t)(base + woff + incl - c);
        base += tot;
        __syncthreads();
    }
    if (threadIdx.x == 0) counter[0] = (int32_t)base;
}

// Pass 2 (fast path): one wave per ray, lanes stride the recorded sample times; position, dt
// and t+dt are recomputed with the identical operations the counting pass used.
__global__ void k_march_write_tbuf(const float* __restrict__ rays_o, const float* __restrict__ rays_d, MarchParams p,
                                   uint32_t max_steps, uint32_t N, const int32_t* __restrict__ rays,
                                   const float* __restrict__ tbuf, float* __restrict__ xyzs, float* __restrict__ dirs,
                                   float* __restrict__ ts) {
    const uint32_t n = (blockIdx.x * blockDim.x + threadIdx.x) >> 6;
    if (n >= N) return;
    const uint32_t offset = (uint32_t)rays[n * 2];
    const uint32_t count = (uint32_t)rays[n * 2 + 1];
    if (count == 0) return;
    const MarchRay r = make_march_ray(rays_o + (size_t)n * 3, rays_d + (size_t)n * 3);
    const float* trow = tbuf + (size_t)n * max_steps;
    for (uint32_t i = lane_id(); i < count; i += kWave) {
        const float t = trow[i];
        float cx, cy, cz;
        march_position(r, p, t, cx, cy, cz);
        const float dt = march_dt(p, t);
        const size_t s = (size_t)offset + i;
        xyzs[s * 3 + 0] = cx; xyzs[s * 3 + 1] = cy; xyzs[s * 3 + 2] = cz;
        dirs[s * 3 + 0] = r.dx; dirs[s * 3 + 1] = r.dy; dirs[s * 3 + 2] = r.dz;
        ts[s * 2 + 0] = t + dt;
        ts[s * 2 + 1] = dt;
    }
}

// Pass 2 for a fixed-capacity iteration (HIP-graph replay): the write of k_march_write_tbuf, plus what PyTorch did around it
// in eight more launches — the rows between the sample total and the capacity are zeroed here instead of pre-zeroing all
// three buffers, and the staging copies of the rays (origins, directions, (offset, count), the total as int and as float) that
// free the staging buffers for the next iteration's counting pass are made by the wave that owns the ray.
__global__ __launch_bounds__(256) void k_march_stage_write(const float* __restrict__ rays_o, const float* __restrict__ rays_d,
                                                            MarchParams p, uint32_t max_steps, uint32_t N,
                                                            const int32_t* __restrict__ rays, const int32_t* __restrict__ counter,
                                                            const float* __restrict__ tbuf, uint32_t capacity, uint32_t ray_blocks,
                                                            float* __restrict__ xyzs, float* __restrict__ dirs, float* __restrict__ ts,
                                                            float* __restrict__ out_rays_o, float* __restrict__ out_rays_d,
                                                            int32_t* __restrict__ out_rays, int32_t* __restrict__ out_total,
                                                            float* __restrict__ out_n_valid) {
    if (blockIdx.x >= ray_blocks) {   // padding rows [total, capacity): 8 floats each
        const uint32_t total = (uint32_t)counter[0];
        const uint32_t first = total < capacity ? total : capacity;
        const uint64_t words = (uint64_t)(capacity - first);
        const uint64_t stride = (uint64_t)(gridDim.x - ray_blocks) * 256;
        for (uint64_t i = (uint64_t)(blockIdx.x - ray_blocks) * 256 + threadIdx.x; i < words * 3; i += stride) {
            xyzs[(size_t)first * 3 + i] = 0.f;
            dirs[(size_t)first * 3 + i] = 0.f;
            if (i < words * 2) ts[(size_t)first * 2 + i] = 0.f;
        }
        return;
    }
    const uint32_t n = (blockIdx.x * 256 + threadIdx.x) >> 6;
    if (blockIdx.x == 0 && threadIdx.x == 0) {
        out_total[0] = counter[0];
        out_n_valid[0] = (float)counter[0];
    }
    if (n >= N) return;
    const int lane = lane_id();
    if (lane < 3) {
        out_rays_o[(size_t)n * 3 + lane] = rays_o[(size_t)n * 3 + lane];
        out_rays_d[(size_t)n * 3 + lane] = rays_d[(size_t)n * 3 + lane];
    } else if (lane < 5) {
        out_rays[(size_t)n * 2 + (lane - 3)] = rays[(size_t)n * 2 + (lane - 3)];
    }
    const uint32_t offset = (uint32_t)rays[n * 2];
    const uint32_t count = (uint32_t)rays[n * 2 + 1];
    if (count == 0) return;
    const MarchRay r = make_march_ray(rays_o + (size_t)n * 3, rays_d + (size_t)n * 3);
    const float* trow = tbuf + (size_t)n * max_steps;
    for (uint32_t i = (uint32_t)lane; i < count; i += kWave) {
        const float t = trow[i];
        float cx, cy, cz;
        march_position(r, p, t, cx, cy, cz);
        const float dt = march_dt(p, t);
        const size_t s = (size_t)offset + i;
        if (s >= capacity) break;      // (a capacity below the total is a caller error; never write past the buffers)
        xyzs[s * 3 + 0] = cx; xyzs[s * 3 + 1] = cy; xyzs[s * 3 + 2] = cz;
        dirs[s * 3 + 0] = r.dx; dirs[s * 3 + 1] = r.dy; dirs[s * 3 + 2] = r.dz;
        ts[s * 2 + 0] = t + dt;
        ts[s * 2 + 1] = dt;
    }
}

// Pass 2 (no scratch): replay the march per ray and write as it goes (raymarching.cu:432-447).
__global__ __launch_bounds__(64) void k_march_write_replay(const float* __restrict__ rays_o,
                                                            const float* __restrict__ rays_d,
                                                            const uint8_t* __restrict__ grid, MarchParams p, uint32_t N,
                                                            const float* __restrict__ nears,
                                                            const float* __restrict__ fars,
                                                            const float* __restrict__ noises,
                                                            const int32_t* __restrict__ rays, float* __restrict__ xyzs,
                                                            float* __restrict__ dirs, float* __restrict__ ts) {
    const uint32_t n = blockIdx.x * blockDim.x + threadIdx.x;
    if (n >= N) return;
    const MarchRay r = make_march_ray(rays_o + (size_t)n * 3, rays_d + (size_t)n * 3);
    size_t s = (size_t)(uint32_t)rays[n * 2];
    const uint32_t num_steps = (uint32_t)rays[n * 2 + 1];
    const float far = fars[n];
    float t = nears[n];
    t += clampf_(t * p.dt_gamma, p.dt_min, p.dt_max) * noises[n];
    uint32_t step = 0;
    while (t < far && step < num_steps) {
        float dt, cx, cy, cz;
        if (march_probe(r, p, grid, t, dt, cx, cy, cz)) {
            step++;
            t += dt;
            xyzs[s * 3 + 0] = cx; xyzs[s * 3 + 1] = cy; xyzs[s * 3 + 2] = cz;
            dirs[s * 3 + 0] = r.dx; dirs[s * 3 + 1] = r.dy; dirs[s * 3 + 2] = r.dz;
            ts[s * 2 + 0] = t;
            ts[s * 2 + 1] = dt;
            s++;
        }
    }
}

// =========================================================================================
// training composite — one wavefront per ray
// =========================================================================================

// alpha of one sample (raymarching.cu:543-544)
__device__ __forceinline__ float sample_alpha(float sigma, float dt, int binarize) {
    const float real_alpha = 1.0f - __expf(-sigma * dt);
    return binarize ? (real_alpha > 0.5f ? 1.0f : 0.0f) : real_alpha;
}

// Front-to-back compositing (raymarching.cu:500-579). Lanes hold 64 consecutive samples; the
// transmittance is an exclusive product scan carried across 64-sample chunks. The serial
// reference stops after the first sample whose post-update T drops below T_thresh; here that
// is the first lane whose inclusive product (times the carry) is below the threshold —
// lanes after it contribute nothing. Writes every weight of the ray's slice (zeros after the
// cut), so the caller's zero-init is honoured either way.
__global__ __launch_bounds__(256) void k_composite_train_fwd(const float* __restrict__ sigmas,
                                                              const float* __restrict__ rgbs,
                                                              const float* __restrict__ ts,
                                                              const int32_t* __restrict__ rays, uint32_t M, uint32_t N,
                                                              float T_thresh, int binarize, float* __restrict__ weights,
                                                              float* __restrict__ weights_sum, float* __restrict__ depth,
                                                              float* __restrict__ image) {
    const uint32_t n = (blockIdx.x * blockDim.x + threadIdx.x) >> 6;
    if (n >= N) return;
    const int lane = lane_id();
    const uint32_t offset = (uint32_t)rays[n * 2];
    const uint32_t num_steps = (uint32_t)rays[n * 2 + 1];

    if (num_steps == 0 || offset + num_steps > M) {  // raymarching.cu:521-528
        if (lane == 0) {
            weights_sum[n] = 0; depth[n] = 0;
            image[n * 3 + 0] = 0; image[n * 3 + 1] = 0; image[n * 3 + 2] = 0;
        }
        return;
    }

    float T_carry = 1.0f;
    float r = 0, g = 0, b = 0, ws = 0, d = 0;
    bool done = false;
    for (uint32_t base = 0; base < num_steps; base += kWave) {
        const uint32_t i = base + lane;
        const bool valid = i < num_steps;
        const size_t s = (size_t)offset + i;
        if (done) {  // past the transmittance cut: only the zero weights remain to be written
            if (valid) weights[s] = 0.0f;
            continue;
        }
        float sigma = 0, dt = 0, t = 0, cr = 0, cg = 0, cb = 0;
        if (valid) {
            sigma = sigmas[s];
            const float2 tt = reinterpret_cast<const float2*>(ts)[s];
            t = tt.x; dt = tt.y;
            cr = rgbs[s * 3 + 0]; cg = rgbs[s * 3 + 1]; cb = rgbs[s * 3 + 2];
        }
        const float alpha = valid ? sample_alpha(sigma, dt, binarize) : 0.0f;
        const float incl = wave_incl_prod(1.0f - alpha, lane);
        const float excl = wave_shift_up1(incl, 1.0f);
        const float T_before = T_carry * excl;
        const float T_after = T_carry * incl;
        const unsigned long long cut = __ballot(valid && (T_after < T_thresh));
        const int first_cut = cut ? (int)__ffsll((long long)cut) - 1 : kWave;
        const bool contributes = valid && lane <= first_cut;
        const float w = contributes ? alpha * T_before : 0.0f;
        if (valid) weights[s] = w;
        r += w * cr; g += w * cg; b += w * cb; ws += w; d += w * t;
        T_carry = T_carry * wave_last(incl);
        done = cut != 0ull;
    }
    r = wave_sum(r); g = wave_sum(g); b = wave_sum(b); ws = wave_sum(ws); d = wave_sum(d);
    if (lane == 0) {
        weights_sum[n] = ws;
        depth[n] = d;
        image[n * 3 + 0] = r; image[n * 3 + 1] = g; image[n * 3 + 2] = b;
    }
}

// Backward (raymarching.cu:605-695), restated literally as a scan:
//   grad_rgb_i   = grad_image * w_i
//   grad_sigma_i = dt_i * ( sum_c gI_c (T_i rgb_ic - (C_c - C_ic)) + (gWS + gW_i)(T_i - (WS - WS_i))
//                           + gD (T_i t_i - (Dp - Dp_i)) )
// with T_i the transmittance AFTER sample i and X_i the inclusive prefix sums of the forward
// accumulators; C, WS, Dp are the saved forward outputs. Samples after the cut keep the
// caller's zeros.
__global__ __launch_bounds__(256) void k_composite_train_bwd(
    const float* __restrict__ grad_weights, const float* __restrict__ grad_weights_sum,
    const float* __restrict__ grad_depth, const float* __restrict__ grad_image, const float* __restrict__ sigmas,
    const float* __restrict__ rgbs, const float* __restrict__ ts, const int32_t* __restrict__ rays,
    const float* __restrict__ weights_sum, const float* __restrict__ depth, const float* __restrict__ image, uint32_t M,
    uint32_t N, float T_thresh, int binarize, float* __restrict__ grad_sigmas, float* __restrict__ grad_rgbs) {
    const uint32_t n = (blockIdx.x * blockDim.x + threadIdx.x) >> 6;
    if (n >= N) return;
    const int lane = lane_id();
    const uint32_t offset = (uint32_t)rays[n * 2];
    const uint32_t num_steps = (uint32_t)rays[n * 2 + 1];
    if (num_steps == 0 || offset + num_steps > M) return;  // raymarching.cu:630

    const float gi0 = grad_image[n * 3 + 0], gi1 = grad_image[n * 3 + 1], gi2 = grad_image[n * 3 + 2];
    const float gws = grad_weights_sum[n], gd = grad_depth[n];
    const float r_final = image[n * 3 + 0], g_final = image[n * 3 + 1], b_final = image[n * 3 + 2];
    const float ws_final = weights_sum[n], d_final = depth[n];

    float T_carry = 1.0f;
    float r_c = 0, g_c = 0, b_c = 0, ws_c = 0, d_c = 0;  // carries of the prefix sums
    for (uint32_t base = 0; base < num_steps; base += kWave) {
        const uint32_t i = base + lane;
        const bool valid = i < num_steps;
        const size_t s = (size_t)offset + i;
        float sigma = 0, dt = 0, t = 0, cr = 0, cg = 0, cb = 0, gw = 0;
        if (valid) {
            sigma = sigmas[s];
            const float2 tt = reinterpret_cast<const float2*>(ts)[s];
            t = tt.x; dt = tt.y;
            cr = rgbs[s * 3 + 0]; cg = rgbs[s * 3 + 1]; cb = rgbs[s * 3 + 2];
            gw = grad_weights[s];
        }
        const float alpha = valid ? sample_alpha(sigma, dt, binarize) : 0.0f;
        const float incl = wave_incl_prod(1.0f - alpha, lane);
        const float excl = wave_shift_up1(incl, 1.0f);
        const float T_before = T_carry * excl;
        const float T = T_carry * incl;  // already advanced, as at raymarching.cu:664
        const unsigned long long cut = __ballot(valid && (T < T_thresh));
        const int first_cut = cut ? (int)__ffsll((long long)cut) - 1 : kWave;
        const bool contributes = valid && lane <= first_cut;
        const float w = contributes ? alpha * T_before : 0.0f;

        const float r = r_c + wave_incl_sum(w * cr, lane);
        const float g = g_c + wave_incl_sum(w * cg, lane);
        const float b = b_c + wave_incl_sum(w * cb, lane);
        const float ws = ws_c + wave_incl_sum(w, lane);
        const float d = d_c + wave_incl_sum(w * t, lane);

        if (contributes) {
            grad_rgbs[s * 3 + 0] = gi0 * w;
            grad_rgbs[s * 3 + 1] = gi1 * w;
            grad_rgbs[s * 3 + 2] = gi2 * w;
            grad_sigmas[s] = dt * (gi0 * (T * cr - (r_final - r)) + gi1 * (T * cg - (g_final - g)) +
                                   gi2 * (T * cb - (b_final - b)) + (gws + gw) * (T - (ws_final - ws)) +
                                   gd * (T * t - (d_final - d)));
        }
        if (cut != 0ull) break;  // wave-uniform
        T_carry = T_carry * wave_last(incl);
        r_c = wave_last(r);
        g_c = wave_last(g);
        b_c = wave_last(b);
        ws_c = wave_last(ws);
        d_c = wave_last(d);
    }
}

// =========================================================================================
// inference march / composite (n_step <= 8 samples per alive ray per call)
// =========================================================================================

// raymarching.cu:713-829. Thread per alive ray: the work per call is a handful of probes and
// there are up to 640k alive rays, so ray-level parallelism already fills the chip.
__global__ __launch_bounds__(256) void k_march_rays(uint32_t n_alive, uint32_t n_step,
                                                     const int32_t* __restrict__ rays_alive,
                                                     const float* __restrict__ rays_t, const float* __restrict__ rays_o,
                                                     const float* __restrict__ rays_d, MarchParams p,
                                                     const uint8_t* __restrict__ grid, const float* __restrict__ fars,
                                                     float* __restrict__ xyzs, float* __restrict__ dirs,
                                                     float* __restrict__ ts, const float* __restrict__ noises) {
    const uint32_t n = blockIdx.x * blockDim.x + threadIdx.x;
    if (n >= n_alive) return;
    const int32_t index = rays_alive[n];
    const MarchRay r = make_march_ray(rays_o + (size_t)index * 3, rays_d + (size_t)index * 3);
    const float far = fars[index];
    float t = rays_t[index];
    t += clampf_(t * p.dt_gamma, p.dt_min, p.dt_max) * noises[n];  // raymarching.cu:756-757 (noise by slot n)
    size_t s = (size_t)n * n_step;
    uint32_t step = 0;
    while (t < far && step < n_step) {
        float dt, cx, cy, cz;
        if (march_probe(r, p, grid, t, dt, cx, cy, cz)) {
            xyzs[s * 3 + 0] = cx; xyzs[s * 3 + 1] = cy; xyzs[s * 3 + 2] = cz;
            dirs[s * 3 + 0] = r.dx; dirs[s * 3 + 1] = r.dy; dirs[s * 3 + 2] = r.dz;
            t += dt;
            ts[s * 2 + 0] = t;
            ts[s * 2 + 1] = dt;
            s++;
            step++;
        }
    }
}

// raymarching.cu:842-925 — accumulates in place with T = 1 - weights_sum.
__global__ __launch_bounds__(256) void k_composite_rays(uint32_t n_alive, uint32_t n_step, float T_thresh, int binarize,
                                                         int32_t* __restrict__ rays_alive, float* __restrict__ rays_t,
                                                         const float* __restrict__ sigmas,
                                                         const float* __restrict__ rgbs, const float* __restrict__ ts,
                                                         float* __restrict__ weights_sum, float* __restrict__ depth,
                                                         float* __restrict__ image) {
    const uint32_t n = blockIdx.x * blockDim.x + threadIdx.x;
    if (n >= n_alive) return;
    const int32_t index = rays_alive[n];
    size_t s = (size_t)n * n_step;
    float t = 0.0f;
    float d = depth[index], r = image[index * 3 + 0], g = image[index * 3 + 1], b = image[index * 3 + 2];
    float weight_sum = weights_sum[index];
    uint32_t step = 0;
    while (step < n_step) {
        const float t_s = ts[s * 2 + 0];
        if (t_s == 0) break;  // ray ended inside this call (slot never written)
        const float alpha = sample_alpha(sigmas[s], ts[s * 2 + 1], binarize);
        const float T = 1 - weight_sum;
        const float weight = alpha * T;
        weight_sum += weight;
        t = t_s;
        d += weight * t;
        r += weight * rgbs[s * 3 + 0];
        g += weight * rgbs[s * 3 + 1];
        b += weight * rgbs[s * 3 + 2];
        if (T < T_thresh) break;
        s++;
        step++;
    }
    if (step < n_step) rays_alive[n] = -1;
    else rays_t[index] = t;
    weights_sum[index] = weight_sum;
    depth[index] = d;
    image[index * 3 + 0] = r; image[index * 3 + 1] = g; image[index * 3 + 2] = b;
}

// =========================================================================================
// stable compaction of the alive list: ballot + popcount + prefix sums (nerf/renderer.py:791)
// =========================================================================================
constexpr int kCompactBlock = 256;

__global__ __launch_bounds__(kCompactBlock) void k_compact_count(const int32_t* __restrict__ in, uint32_t n,
                                                                   uint32_t* __restrict__ block_counts) {
    __shared__ uint32_t wave_cnt[kCompactBlock / kWave];
    const uint32_t i = blockIdx.x * kCompactBlock + threadIdx.x;
    const bool keep = i < n && in[i] >= 0;
    const unsigned long long m = __ballot(keep);
    if (lane_id() == 0) wave_cnt[threadIdx.x >> 6] = (uint32_t)__popcll(m);
    __syncthreads();
    if (threadIdx.x == 0) {
        uint32_t c = 0;
        for (int w = 0; w < kCompactBlock / kWave; w++) c += wave_cnt[w];
        block_counts[blockIdx.x] = c;
    }
}

// in-place exclusive scan of the per-block counts; total -> count_out[0]
__global__ __launch_bounds__(1024) void k_compact_scan(uint32_t* __restrict__ block_counts, uint32_t nblocks,
                                                        int32_t* __restrict__ count_out) {
    __shared__ uint32_t wave_tot[16];
    const int lane = lane_id();
    const int wid = (int)(threadIdx.x >> 6);
    uint32_t base = 0;
    for (uint32_t start = 0; start < nblocks; start += 1024) {
        const uint32_t i = start + threadIdx.x;
        const uint32_t c = i < nblocks ? block_counts[i] : 0u;
        const uint32_t incl = wave_incl_sum_u32(c, lane);
        if (lane == kWave - 1) wave_tot[wid] = incl;
        __syncthreads();
        uint32_t woff = 0, tot = 0;
#pragma unroll
        for (int w = 0; w < 16; w++) {
            const uint32_t v = wave_tot[w];
            if (w < wid) woff += v;
            tot += v;
        }
        if (i < nblocks) block_counts[i] = base + woff + incl - c;
        base += tot;
        __syncthreads();
    }
    if (threadIdx.x == 0) count_out[0] = (int32_t)base;
}

__global__ __launch_bounds__(kCompactBlock) void k_compact_scatter(const int32_t* __restrict__ in, uint32_t n,
                                                                     const uint32_t* __restrict__ block_offsets,
                                                                     int32_t* __restrict__ out) {
    __shared__ uint32_t wave_cnt[kCompactBlock / kWave];
    const uint32_t i = blockIdx.x * kCompactBlock + threadIdx.x;
    const int32_t v = i < n ? in[i] : -1;
    const bool keep = v >= 0;
    const unsigned long long m = __ballot(keep);
    const int lane = lane_id();
    const int wid = (int)(threadIdx.x >> 6);
    if (lane == 0) wave_cnt[wid] = (uint32_t)__popcll(m);
    __syncthreads();
    uint32_t woff = 0;
    for (int w = 0; w < wid; w++) woff += wave_cnt[w];
    if (keep) {
        const uint32_t rank = (uint32_t)__popcll(m & ((1ull << lane) - 1ull));
        out[block_offsets[blockIdx.x] + woff + rank] = v;
    }
}

// =========================================================================================
// C ABI
// =========================================================================================
extern "C" {

int sdfx_near_far_from_aabb(const float* rays_o, const float* rays_d, const float* aabb, uint32_t N, float min_near,
                            float* nears, float* fars, sdfx_stream_t stream) {
    SDFX_REQUIRE(rays_o && rays_d && aabb && nears && fars, "near_far_from_aabb: null pointer");
    if (N == 0) return SDFX_OK;
    hipLaunchKernelGGL(k_near_far_from_aabb, dim3(div_up(N, 256)), dim3(256), 0, as_stream(stream), rays_o, rays_d, aabb,
                       N, min_near, nears, fars);
    return check_launch("near_far_from_aabb");
}

int sdfx_sph_from_ray(const float* rays_o, const float* rays_d, float radius, uint32_t N, float* coords,
                      sdfx_stream_t stream) {
    SDFX_REQUIRE(rays_o && rays_d && coords, "sph_from_ray: null pointer");
    if (N == 0) return SDFX_OK;
    hipLaunchKernelGGL(k_sph_from_ray, dim3(div_up(N, 256)), dim3(256), 0, as_stream(stream), rays_o, rays_d, radius, N,
                       coords);
    return check_launch("sph_from_ray");
}

int sdfx_morton3D(const int32_t* coords, uint32_t N, int32_t* indices, sdfx_stream_t stream) {
    SDFX_REQUIRE(coords && indices, "morton3D: null pointer");
    if (N == 0) return SDFX_OK;
    hipLaunchKernelGGL(k_morton3D, dim3(div_up(N, 256)), dim3(256), 0, as_stream(stream), coords, N, indices);
    return check_launch("morton3D");
}

int sdfx_morton3D_invert(const int32_t* indices, uint32_t N, int32_t* coords, sdfx_stream_t stream) {
    SDFX_REQUIRE(coords && indices, "morton3D_invert: null pointer");
    if (N == 0) return SDFX_OK;
    hipLaunchKernelGGL(k_morton3D_invert, dim3(div_up(N, 256)), dim3(256), 0, as_stream(stream), indices, N, coords);
    return check_launch("morton3D_invert");
}

int sdfx_packbits(const float* grid, uint32_t N, float density_thresh, uint8_t* bitfield, sdfx_stream_t stream) {
    SDFX_REQUIRE(grid && bitfield, "packbits: null pointer");
    if (N == 0) return SDFX_OK;
    const int vec = (reinterpret_cast<uintptr_t>(grid) & 15u) == 0;
    hipLaunchKernelGGL(k_packbits, dim3(div_up(N, 256)), dim3(256), 0, as_stream(stream), grid, N, density_thresh,
                       bitfield, vec);
    return check_launch("packbits");
}

int sdfx_flatten_rays(const int32_t* rays, uint32_t N, uint32_t M, int32_t* res, sdfx_stream_t stream) {
    SDFX_REQUIRE(rays && (res || M == 0), "flatten_rays: null pointer");
    if (N == 0 || M == 0) return SDFX_OK;
    hipLaunchKernelGGL(k_flatten_rays, dim3(div_up((uint64_t)N * kWave, 256)), dim3(256), 0, as_stream(stream), rays, N,
                       M, res);
    return check_launch("flatten_rays");
}

uint64_t sdfx_march_rays_train_scratch_bytes(uint32_t N, uint32_t max_steps) {
    return (uint64_t)N * (uint64_t)max_steps * sizeof(float);
}

int sdfx_march_rays_train(const float* rays_o, const float* rays_d, const uint8_t* grid, float bound, int contract,
                          float dt_gamma, uint32_t max_steps, uint32_t N, uint32_t C, uint32_t H, const float* nears,
                          const float* fars, float* xyzs, float* dirs, float* ts, int32_t* rays, int32_t* counter,
                          const float* noises, float* scratch, sdfx_stream_t stream) {
    SDFX_REQUIRE(rays_o && rays_d && grid && nears && fars && rays && counter && noises,
                 "march_rays_train: null pointer");
    SDFX_REQUIRE(max_steps > 0 && H > 0 && C > 0, "march_rays_train: max_steps, C and H must be positive");
    if (N == 0) return SDFX_OK;
    const MarchParams p = make_march_params(bound, contract, dt_gamma, max_steps, C, H);
    hipStream_t st = as_stream(stream);
    if (xyzs == nullptr) {  // pass 1
#ifdef SDFX_DEVTOOLS
        if (dev_switch("SDFX_MARCH_WAVE", 1) == 0)
            hipLaunchKernelGGL(k_march_count, dim3(div_up(N, 64)), dim3(64), 0, st, rays_o, rays_d, grid, p, max_steps, N,
                               nears, fars, noises, rays, scratch);
        else
#endif
            hipLaunchKernelGGL(k_march_count_wave, dim3(div_up((uint64_t)N * kWave, 256)), dim3(256), 0, st, rays_o, rays_d, grid,
                               p, max_steps, N, nears, fars, noises, rays, scratch);
        hipLaunchKernelGGL(k_scan_counts, dim3(1), dim3(256), 0, st, rays, N, counter);
        return check_launch("march_rays_train(count)");
    }
    SDFX_REQUIRE(dirs && ts, "march_rays_train: dirs/ts must be given together with xyzs");
    if (scratch) {
        hipLaunchKernelGGL(k_march_write_tbuf, dim3(div_up((uint64_t)N * kWave, 256)), dim3(256), 0, st, rays_o, rays_d,
                           p, max_steps, N, rays, scratch, xyzs, dirs, ts);
    } else {
        hipLaunchKernelGGL(k_march_write_replay, dim3(div_up(N, 64)), dim3(64), 0, st, rays_o, rays_d, grid, p, N, nears,
                           fars, noises, rays, xyzs, dirs, ts);
    }
    return check_launch("march_rays_train(write)");
}

int sdfx_march_rays_train_stage_write(const float* rays_o, const float* rays_d, float bound, int contract, float dt_gamma,
                                      uint32_t max_steps, uint32_t N, uint32_t C, uint32_t H, const int32_t* rays,
                                      const int32_t* counter, const float* scratch, uint32_t capacity, float* xyzs, float* dirs,
                                      float* ts, float* out_rays_o, float* out_rays_d, int32_t* out_rays, int32_t* out_total,
                                      float* out_n_valid, sdfx_stream_t stream) {
    SDFX_REQUIRE(rays_o && rays_d && rays && counter && scratch && out_rays_o && out_rays_d && out_rays && out_total && out_n_valid,
                 "march_rays_train_stage_write: null pointer");
    SDFX_REQUIRE(capacity == 0 || (xyzs && dirs && ts), "march_rays_train_stage_write: null sample buffer");
    SDFX_REQUIRE(max_steps > 0 && H > 0 && C > 0, "march_rays_train_stage_write: max_steps, C and H must be positive");
    if (N == 0) return SDFX_OK;
    const MarchParams p = make_march_params(bound, contract, dt_gamma, max_steps, C, H);
    const uint32_t ray_blocks = div_up((uint64_t)N * kWave, 256);
    const uint32_t pad_blocks = capacity ? 64 : 0;
    hipLaunchKernelGGL(k_march_stage_write, dim3(ray_blocks + pad_blocks), dim3(256), 0, as_stream(stream), rays_o, rays_d, p,
                       max_steps, N, rays, counter, scratch, capacity, ray_blocks, xyzs, dirs, ts, out_rays_o, out_rays_d, out_rays,
                       out_total, out_n_valid);
    return check_launch("march_rays_train_stage_write");
}

int sdfx_composite_rays_train_forward(const float* sigmas, const float* rgbs, const float* ts, const int32_t* rays,
                                      uint32_t M, uint32_t N, float T_thresh, int binarize, float* weights,
                                      float* weights_sum, float* depth, float* image, sdfx_stream_t stream) {
    SDFX_REQUIRE(rays && weights_sum && depth && image, "composite_rays_train_forward: null pointer");
    SDFX_REQUIRE(M == 0 || (sigmas && rgbs && ts && weights), "composite_rays_train_forward: null sample pointer");
    if (N == 0) return SDFX_OK;
    hipLaunchKernelGGL(k_composite_train_fwd, dim3(div_up((uint64_t)N * kWave, 256)), dim3(256), 0, as_stream(stream),
                       sigmas, rgbs, ts, rays, M, N, T_thresh, binarize, weights, weights_sum, depth, image);
    return check_launch("composite_rays_train_forward");
}

int sdfx_composite_rays_train_backward(const float* grad_weights, const float* grad_weights_sum, const float* grad_depth,
                                       const float* grad_image, const float* sigmas, const float* rgbs, const float* ts,
                                       const int32_t* rays, const float* weights_sum, const float* depth,
                                       const float* image, uint32_t M, uint32_t N, float T_thresh, int binarize,
                                       float* grad_sigmas, float* grad_rgbs, sdfx_stream_t stream) {
    SDFX_REQUIRE(grad_weights_sum && grad_depth && grad_image && rays && weights_sum && depth && image,
                 "composite_rays_train_backward: null pointer");
    SDFX_REQUIRE(M == 0 || (grad_weights && sigmas && rgbs && ts && grad_sigmas && grad_rgbs),
                 "composite_rays_train_backward: null sample pointer");
    if (N == 0 || M == 0) return SDFX_OK;
    hipLaunchKernelGGL(k_composite_train_bwd, dim3(div_up((uint64_t)N * kWave, 256)), dim3(256), 0, as_stream(stream),
                       grad_weights, grad_weights_sum, grad_depth, grad_image, sigmas, rgbs, ts, rays, weights_sum, depth,
                       image, M, N, T_thresh, binarize, grad_sigmas, grad_rgbs);
    return check_launch("composite_rays_train_backward");
}

int sdfx_march_rays(uint32_t n_alive, uint32_t n_step, const int32_t* rays_alive, const float* rays_t,
                    const float* rays_o, const float* rays_d, float bound, int contract, float dt_gamma,
                    uint32_t max_steps, uint32_t C, uint32_t H, const uint8_t* grid, const float* nears,
                    const float* fars, float* xyzs, float* dirs, float* ts, const float* noises, sdfx_stream_t stream) {
    (void)nears;
    SDFX_REQUIRE(rays_alive && rays_t && rays_o && rays_d && grid && fars && xyzs && dirs && ts && noises,
                 "march_rays: null pointer");
    SDFX_REQUIRE(max_steps > 0 && H > 0 && C > 0, "march_rays: max_steps, C and H must be positive");
    if (n_alive == 0 || n_step == 0) return SDFX_OK;
    const MarchParams p = make_march_params(bound, contract, dt_gamma, max_steps, C, H);
    hipLaunchKernelGGL(k_march_rays, dim3(div_up(n_alive, 256)), dim3(256), 0, as_stream(stream), n_alive, n_step,
                       rays_alive, rays_t, rays_o, rays_d, p, grid, fars, xyzs, dirs, ts, noises);
    return check_launch("march_rays");
}

int sdfx_composite_rays(uint32_t n_alive, uint32_t n_step, float T_thresh, int binarize, int32_t* rays_alive,
                        float* rays_t, const float* sigmas, const float* rgbs, const float* ts, float* weights_sum,
                        float* depth, float* image, sdfx_stream_t stream) {
    SDFX_REQUIRE(rays_alive && rays_t && sigmas && rgbs && ts && weights_sum && depth && image,
                 "composite_rays: null pointer");
    if (n_alive == 0) return SDFX_OK;
    hipLaunchKernelGGL(k_composite_rays, dim3(div_up(n_alive, 256)), dim3(256), 0, as_stream(stream), n_alive, n_step,
                       T_thresh, binarize, rays_alive, rays_t, sigmas, rgbs, ts, weights_sum, depth, image);
    return check_launch("composite_rays");
}

uint64_t sdfx_compact_rays_scratch_bytes(uint32_t n) { return (uint64_t)div_up(n ? n : 1, kCompactBlock) * sizeof(uint32_t); }

int sdfx_compact_rays(const int32_t* rays_alive_in, uint32_t n, int32_t* rays_alive_out, int32_t* count_out,
                      void* scratch, sdfx_stream_t stream) {
    SDFX_REQUIRE(count_out && scratch, "compact_rays: null pointer");
    hipStream_t st = as_stream(stream);
    if (n == 0) {
        zero_device(count_out, sizeof(int32_t), st);
        return check_launch("compact_rays(empty)");
    }
    SDFX_REQUIRE(rays_alive_in && rays_alive_out, "compact_rays: null pointer");
    SDFX_REQUIRE(rays_alive_in != rays_alive_out, "compact_rays: in-place compaction is not supported");
    const uint32_t nblocks = div_up(n, kCompactBlock);
    uint32_t* bc = static_cast<uint32_t*>(scratch);
    hipLaunchKernelGGL(k_compact_count, dim3(nblocks), dim3(kCompactBlock), 0, st, rays_alive_in, n, bc);
    hipLaunchKernelGGL(k_compact_scan, dim3(1), dim3(1024), 0, st, bc, nblocks, count_out);
    hipLaunchKernelGGL(k_compact_scatter, dim3(nblocks), dim3(kCompactBlock), 0, st, rays_alive_in, n, bc,
                       rays_alive_out);
    return check_launch("compact_rays");
}

}  // extern "C"

// occupancy.hip — the occupancy-grid refresh of NeRFRenderer.update_extra_state (nerf/renderer.py:1102-1149) as three
// small kernels around the field evaluation, with no host synchronisation and no index tensors:
//
//   k_occ_points   cell m of the Morton-ordered grid -> its jittered sample position: coords = morton3D_invert(m),
//                  x = (2 c / (H-1) - 1) (bound_c - h) + (2 u - 1) h, h = bound_c / H            (renderer.py:1123-1133)
//                  The reference builds coords [H^3, 3] (25 MB), their Morton codes [H^3] and scatters the densities back
//                  through them (`tmp_grid[cas, indices] = sigmas`); generating the points IN Morton order makes the
//                  density of point m the new value of cell m — no coords, no indices, no scatter — and neighbouring
//                  lanes are neighbouring cells, which the hash-grid encoder likes (shared table lines).
//                  Jitter u: either the caller's array (laid out as the reference's torch.rand_like(cas_xyzs), i.e. by
//                  meshgrid index n = (x H + y) H + z — used by the parity tests), or Philox4x32-10 keyed by (seed, cascade)
//                  with counter n: reproducible for a given seed whatever the launch shape.
//   k_occ_update   density_grid = max(density_grid * decay, sigma) where density_grid >= 0 (renderer.py:1137-1139), and
//                  the sum and count of the updated valid cells (for the mean, :1140): per-workgroup partials in fixed slots,
//                  added up in slot order by the last workgroup to arrive (bit-reproducible: no floating-point atomics).
//   k_occ_pack     threshold = min(mean, density_thresh) computed ON THE DEVICE from those two numbers (the reference
//                  reads the mean back with .item(), :1140-1144), then the bit packing of raymarching.cu:267-300.
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "sdfx.h"
#include "sdfx_common.h"

using namespace sdfx;

namespace {

// Philox4x32-10 (Salmon et al., SC'11): counter (c0..c3), key (k0, k1) -> 4 x 32 random bits
__device__ __forceinline__ void philox4x32_10(uint32_t c[4], uint32_t k0, uint32_t k1) {
#pragma unroll
    for (int r = 0; r < 10; r++) {
        const uint64_t p0 = (uint64_t)0xD2511F53u * c[0], p1 = (uint64_t)0xCD9E8D57u * c[2];
        const uint32_t n0 = (uint32_t)(p1 >> 32) ^ c[1] ^ k0, n2 = (uint32_t)(p0 >> 32) ^ c[3] ^ k1;
        c[1] = (uint32_t)p1; c[3] = (uint32_t)p0; c[0] = n0; c[2] = n2;
        k0 += 0x9E3779B9u; k1 += 0xBB67AE85u;
    }
}
__device__ __forceinline__ float u01(uint32_t bits) { return (float)(bits >> 8) * 0x1p-24f; }   // [0, 1), 24 bits as torch.rand

__global__ __launch_bounds__(256) void k_occ_points(uint32_t H, uint32_t n_cells, float s, float h, const float* __restrict__ noise,
                                                     uint64_t seed, uint32_t cascade, float* __restrict__ xyzs) {
    const uint32_t m = blockIdx.x * blockDim.x + threadIdx.x;
    if (m >= n_cells) return;
    const uint32_t x = morton3D_invert(m), y = morton3D_invert(m >> 1), z = morton3D_invert(m >> 2);
    const uint32_t n = (x * H + y) * H + z;              // index of this cell in the reference's meshgrid order
    float u[3];
    if (noise) {
        u[0] = noise[(size_t)n * 3 + 0]; u[1] = noise[(size_t)n * 3 + 1]; u[2] = noise[(size_t)n * 3 + 2];
    } else {
        uint32_t c[4] = {n, cascade, 0u, 0u};
        philox4x32_10(c, (uint32_t)seed, (uint32_t)(seed >> 32));
        u[0] = u01(c[0]); u[1] = u01(c[1]); u[2] = u01(c[2]);
    }
    // s = bound_c - half_grid_size, h = half_grid_size = bound_c / H: Python doubles in the reference (renderer.py:1127-1131),
    // rounded to float32 when they meet the float32 tensors — computed that way on the host
    const float cc[3] = {(float)x, (float)y, (float)z};
    // xyzs = 2 * coords.float() / (grid_size - 1) - 1 (renderer.py:1124). On the GPU — where the reference runs this —
    // PyTorch divides a tensor by a Python scalar as a multiplication by the float32 reciprocal (BinaryDivTrueKernel.cu:
    // "compute a * reciprocal(b)"), which differs from a true division in the last bit for some cells; matched here.
    const float inv = 1.0f / (float)(H - 1);
#pragma unroll
    for (int d = 0; d < 3; d++) {
        const float g = (2 * cc[d]) * inv - 1;
        xyzs[(size_t)m * 3 + d] = g * s + (u[d] * 2 - 1) * h;
    }
}

// `stats` (float64 words; sdfx_occupancy_stats_doubles() of them): [0] sum, [1] count, [2] arrival ticket of the launch in flight,
// [3 ...] per-workgroup (sum, count) partials. Each workgroup fills its own slot, the one that arrives last adds the slots in slot
// order: the mean — and with it the occupancy threshold — is the same bits whatever order the workgroups ran in.
constexpr uint32_t kOccHeader = 3, kOccMaxBlocks = 2048;

__global__ __launch_bounds__(256) void k_occ_update(float* __restrict__ grid, const float* __restrict__ sigmas, uint32_t n_cells,
                                                     float decay, double* __restrict__ stats) {
    __shared__ double part[2][4];
    __shared__ double red[2][256];
    __shared__ int is_last;
    double sum = 0.0, cnt = 0.0;
    for (uint32_t i = blockIdx.x * 256 + threadIdx.x; i < n_cells; i += gridDim.x * 256) {
        const float old = grid[i];
        if (old >= 0.f) {                                 // valid_mask (renderer.py:1137); never-visited cells hold -1
            const float v = fmaxf(old * decay, sigmas[i]);
            grid[i] = v;
            sum += (double)v; cnt += 1.0;
        }
    }
    for (int o = 32; o > 0; o >>= 1) { sum += __shfl_down(sum, o); cnt += __shfl_down(cnt, o); }
    if ((threadIdx.x & 63) == 0) { part[0][threadIdx.x >> 6] = sum; part[1][threadIdx.x >> 6] = cnt; }
    __syncthreads();
    double* __restrict__ slots = stats + kOccHeader;
    unsigned long long* ticket = reinterpret_cast<unsigned long long*>(stats + 2);
    if (threadIdx.x == 0) {
        __hip_atomic_store(&slots[2 * blockIdx.x], part[0][0] + part[0][1] + part[0][2] + part[0][3], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        __hip_atomic_store(&slots[2 * blockIdx.x + 1], part[1][0] + part[1][1] + part[1][2] + part[1][3], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        __threadfence();
        is_last = atomicAdd(ticket, 1ull) == (unsigned long long)gridDim.x - 1;
    }
    __syncthreads();
    if (!is_last) return;
    __threadfence();
    double s = 0.0, c = 0.0;
    for (uint32_t i = threadIdx.x; i < gridDim.x; i += 256) {
        s += __hip_atomic_load(&slots[2 * i], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        c += __hip_atomic_load(&slots[2 * i + 1], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    }
    red[0][threadIdx.x] = s; red[1][threadIdx.x] = c;
    __syncthreads();
    for (uint32_t o = 128; o > 0; o >>= 1) {
        if (threadIdx.x < o) { red[0][threadIdx.x] += red[0][threadIdx.x + o]; red[1][threadIdx.x] += red[1][threadIdx.x + o]; }
        __syncthreads();
    }
    if (threadIdx.x == 0) {
        stats[0] += red[0][0];      // the cascades' launches follow one another on the stream
        stats[1] += red[1][0];
        *ticket = 0ull;
    }
}

__global__ __launch_bounds__(256) void k_occ_pack(const float* __restrict__ grid, uint32_t n_bytes, const double* __restrict__ stats,
                                                   float max_thresh, uint8_t* __restrict__ bitfield, float* __restrict__ mean_out) {
    const float mean = (float)(stats[0] / stats[1]);      // torch.mean of the valid cells (NaN if there are none, as torch's)
    const float thresh = fminf(mean, max_thresh);         // min(self.mean_density, self.density_thresh)
    const uint32_t n = blockIdx.x * blockDim.x + threadIdx.x;
    if (n == 0 && mean_out) mean_out[0] = mean;
    if (n >= n_bytes) return;
    const float4 a = reinterpret_cast<const float4*>(grid)[(size_t)n * 2];
    const float4 b = reinterpret_cast<const float4*>(grid)[(size_t)n * 2 + 1];
    const float v[8] = {a.x, a.y, a.z, a.w, b.x, b.y, b.z, b.w};
    uint32_t bits = 0;
#pragma unroll
    for (int i = 0; i < 8; i++) bits |= (v[i] > thresh) ? (1u << i) : 0u;   // raymarching.cu:285
    bitfield[n] = (uint8_t)bits;
}

}  // namespace

extern "C" {

uint32_t sdfx_occupancy_stats_doubles(void) { return kOccHeader + 2 * kOccMaxBlocks; }

int sdfx_occupancy_points(uint32_t H, double bound_cascade, const float* noise, uint64_t seed, uint32_t cascade, float* xyzs,
                          sdfx_stream_t stream) {
    SDFX_REQUIRE(xyzs, "occupancy_points: null pointer");
    SDFX_REQUIRE(H >= 2 && H <= 1024 && (H & (H - 1)) == 0, "occupancy_points: H must be a power of two in [2, 1024]");
    const uint32_t n = H * H * H;
    const double half = bound_cascade / (double)H;
    hipLaunchKernelGGL(k_occ_points, dim3(div_up(n, 256)), dim3(256), 0, as_stream(stream), H, n, (float)(bound_cascade - half),
                       (float)half, noise, seed, cascade, xyzs);
    return check_launch("occupancy_points");
}

int sdfx_occupancy_update(float* density_grid_cascade, const float* sigmas, uint32_t n_cells, float decay, double* stats, int reset_stats,
                          sdfx_stream_t stream) {
    SDFX_REQUIRE(density_grid_cascade && sigmas && stats, "occupancy_update: null pointer");
    hipStream_t st = as_stream(stream);
    if (reset_stats) zero_device(stats, kOccHeader * sizeof(double), st);
    if (n_cells == 0) return SDFX_OK;
    const uint32_t blocks = div_up(n_cells, 256 * 8) < kOccMaxBlocks ? div_up(n_cells, 256 * 8) : kOccMaxBlocks;
    hipLaunchKernelGGL(k_occ_update, dim3(blocks), dim3(256), 0, st, density_grid_cascade, sigmas, n_cells, decay, stats);
    return check_launch("occupancy_update");
}

int sdfx_occupancy_pack(const float* density_grid, uint32_t n_cells, const double* stats, float density_thresh, uint8_t* bitfield,
                        float* mean_out, sdfx_stream_t stream) {
    SDFX_REQUIRE(density_grid && stats && bitfield, "occupancy_pack: null pointer");
    SDFX_REQUIRE(n_cells % 8 == 0 && (reinterpret_cast<uintptr_t>(density_grid) % 16) == 0,
                 "occupancy_pack: n_cells must be a multiple of 8 and the grid 16-byte aligned");
    if (n_cells == 0) return SDFX_OK;
    hipLaunchKernelGGL(k_occ_pack, dim3(div_up(n_cells / 8, 256)), dim3(256), 0, as_stream(stream), density_grid, n_cells / 8, stats,
                       density_thresh, bitfield, mean_out);
    return check_launch("occupancy_pack");
}

}  // extern "C"

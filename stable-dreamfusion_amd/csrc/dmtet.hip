// dmtet.hip — marching tetrahedra of the DMTet fine-tune stage (BASELINE configs[4]): the mesh extraction the reference does with
// ~25 tensor operations per iteration in `class DMTet.__call__` (nerf/renderer.py:94-178: boolean masks, torch.unique over the
// sorted edges of the valid tetrahedra, index remapping, two gathers through the triangle table), as three small kernels with
// IDENTICAL outputs — vertex order, face order and vertex indices included — plus the backward of the vertex interpolation.
//
// Why the outputs can be identical without sorting anything on the device: the tetrahedral grid is static. torch.unique(dim=0)
// returns the edges of the VALID tetrahedra in lexicographic order, and that is a sub-sequence of the lexicographically sorted
// list of ALL edges of the grid, which the host builds once (sdfx_nerf/dmtet.py). An edge gets a vertex when exactly one of its
// ends is inside (sdf > 0), and such an edge always belongs to a valid tetrahedron; so
//     vertex id of an edge = number of crossing edges before it in the global list          (one prefix sum over E flags),
//     faces                = first the one-triangle tetrahedra in grid order, then the two-triangle ones (renderer.py:170-173)
//                                                                                           (two prefix sums over F counts),
// with the corner of a face looked up as triangle_table[tetindex][c] -> local edge -> global edge (static [F, 6] map) -> vertex id.
//
// Vertex position (renderer.py:154-162), float32, same operations in the same order (no contraction):
//     den = s_a + (-s_b);  v = p_a * ((-s_b) / den) + p_b * (s_a / den)        for the edge (a, b), a < b.
#include "sdfx_common.h"

using namespace sdfx;

namespace {

constexpr uint32_t kBlock = 256, kPer = 4, kChunk = kBlock * kPer;   // elements per workgroup

// renderer.py:97-114 (triangle_table), :115 (num_triangles_table)
__constant__ int8_t c_tri[16][6] = {{-1, -1, -1, -1, -1, -1}, {1, 0, 2, -1, -1, -1}, {4, 0, 3, -1, -1, -1}, {1, 4, 2, 1, 3, 4},
                                    {3, 1, 5, -1, -1, -1},    {2, 3, 0, 2, 5, 3},    {1, 4, 0, 1, 5, 4},    {4, 2, 5, -1, -1, -1},
                                    {4, 5, 2, -1, -1, -1},    {4, 1, 0, 4, 5, 1},    {3, 2, 0, 3, 5, 2},    {1, 3, 5, -1, -1, -1},
                                    {4, 1, 2, 4, 3, 1},       {3, 0, 4, -1, -1, -1}, {2, 0, 1, -1, -1, -1}, {-1, -1, -1, -1, -1, -1}};
__constant__ uint8_t c_ntri[16] = {0, 1, 1, 2, 1, 2, 2, 1, 1, 2, 2, 1, 2, 1, 1, 0};

__device__ __forceinline__ bool crossing(const float* __restrict__ sdf, const int32_t* __restrict__ edges, uint32_t e) {
    return (sdf[edges[2 * e]] > 0.f) != (sdf[edges[2 * e + 1]] > 0.f);
}
__device__ __forceinline__ uint32_t tet_index(const float* __restrict__ sdf, const int32_t* __restrict__ tets, uint32_t t) {
    uint32_t idx = 0;
#pragma unroll
    for (uint32_t k = 0; k < 4; k++) idx |= (sdf[tets[4 * t + k]] > 0.f ? 1u : 0u) << k;   // renderer.py:164-165
    return idx;
}

static_assert(kWave == 64, "block_sum / block_excl index their per-wave slots with threadIdx.x >> 6 (gfx950: wave64)");
// block sum of a per-thread count (all threads get the total)
__device__ __forceinline__ uint32_t block_sum(uint32_t v, uint32_t* sh) {
    v = wave_sum(v);
    if ((threadIdx.x & 63u) == 0) sh[threadIdx.x >> 6] = v;
    __syncthreads();
    uint32_t s = 0;
#pragma unroll
    for (uint32_t w = 0; w < kBlock / 64; w++) s += sh[w];
    __syncthreads();
    return s;
}
// exclusive prefix of a per-thread count within the block
__device__ __forceinline__ uint32_t block_excl(uint32_t v, uint32_t* sh) {
    const int lane = lane_id();
    const uint32_t incl = wave_incl_sum_u32(v, lane);
    if (lane == kWave - 1) sh[threadIdx.x >> 6] = incl;
    __syncthreads();
    uint32_t off = 0;
    for (uint32_t w = 0; w < (threadIdx.x >> 6); w++) off += sh[w];
    __syncthreads();
    return off + incl - v;
}

// pass 1: per workgroup, the number of crossing edges / one-triangle / two-triangle tetrahedra of its chunk.
// blocks [0, nbE) count edges, blocks [nbE, nbE + nbF) count tetrahedra
__global__ __launch_bounds__(kBlock) void k_mt_count(const float* __restrict__ sdf, const int32_t* __restrict__ edges, uint32_t E,
                                                      const int32_t* __restrict__ tets, uint32_t F, uint32_t nbE,
                                                      uint32_t* __restrict__ counts) {   // counts: [nbE] | [nbF] ones | [nbF] twos
    __shared__ uint32_t sh[kBlock / 64];
    if (blockIdx.x < nbE) {
        uint32_t c = 0;
#pragma unroll
        for (uint32_t j = 0; j < kPer; j++) {
            const uint32_t e = blockIdx.x * kChunk + threadIdx.x * kPer + j;
            if (e < E && crossing(sdf, edges, e)) c++;
        }
        c = block_sum(c, sh);
        if (threadIdx.x == 0) counts[blockIdx.x] = c;
        return;
    }
    const uint32_t b = blockIdx.x - nbE, nbF = gridDim.x - nbE;
    uint32_t c1 = 0, c2 = 0;
#pragma unroll
    for (uint32_t j = 0; j < kPer; j++) {
        const uint32_t t = b * kChunk + threadIdx.x * kPer + j;
        if (t < F) {
            const uint32_t n = c_ntri[tet_index(sdf, tets, t)];
            c1 += n == 1u; c2 += n == 2u;
        }
    }
    c1 = block_sum(c1, sh);
    c2 = block_sum(c2, sh);
    if (threadIdx.x == 0) { counts[nbE + b] = c1; counts[nbE + nbF + b] = c2; }
}

// pass 2 (one workgroup): exclusive scans of the three count arrays in place; totals -> out[0..2] = (V, F1, F2)
__global__ __launch_bounds__(1024) void k_mt_scan(uint32_t* __restrict__ counts, uint32_t nbE, uint32_t nbF, int32_t* __restrict__ out) {
    __shared__ uint32_t sh[16];
    __shared__ uint32_t carry;
    for (uint32_t which = 0; which < 3; which++) {
        uint32_t* a = counts + (which == 0 ? 0 : nbE + (which - 1) * nbF);
        const uint32_t n = which == 0 ? nbE : nbF;
        if (threadIdx.x == 0) carry = 0;
        __syncthreads();
        for (uint32_t base = 0; base < n; base += 1024) {
            const uint32_t i = base + threadIdx.x;
            const uint32_t v = i < n ? a[i] : 0u;
            const int lane = lane_id();
            const uint32_t incl = wave_incl_sum_u32(v, lane);
            if (lane == kWave - 1) sh[threadIdx.x >> 6] = incl;
            __syncthreads();
            uint32_t off = carry;
            for (uint32_t w = 0; w < (threadIdx.x >> 6); w++) off += sh[w];
            if (i < n) a[i] = off + incl - v;
            __syncthreads();
            if (threadIdx.x == 1023) carry = off + incl;
            __syncthreads();
        }
        if (threadIdx.x == 0) out[which] = (int32_t)carry;
        __syncthreads();
    }
}

// pass 3a: vertex ids of the crossing edges (-1 elsewhere), vertex positions and the (a, b) pair each vertex came from
__global__ __launch_bounds__(kBlock) void k_mt_emit_verts(const float* __restrict__ pos, const float* __restrict__ sdf,
                                                           const int32_t* __restrict__ edges, uint32_t E,
                                                           const uint32_t* __restrict__ block_off, int32_t* __restrict__ edge_vid,
                                                           float* __restrict__ verts, int32_t* __restrict__ vert_edges, uint32_t cap) {
    __shared__ uint32_t sh[kBlock / 64];
    bool cr[kPer];
    uint32_t c = 0;
#pragma unroll
    for (uint32_t j = 0; j < kPer; j++) {
        const uint32_t e = blockIdx.x * kChunk + threadIdx.x * kPer + j;
        cr[j] = e < E && crossing(sdf, edges, e);
        c += cr[j];
    }
    uint32_t id = block_off[blockIdx.x] + block_excl(c, sh);
#pragma unroll
    for (uint32_t j = 0; j < kPer; j++) {
        const uint32_t e = blockIdx.x * kChunk + threadIdx.x * kPer + j;
        if (e >= E) break;
        if (!cr[j]) { edge_vid[e] = -1; continue; }
        edge_vid[e] = (int32_t)id;
        if (id < cap) {
            const int32_t a = edges[2 * e], b = edges[2 * e + 1];
            const float sa = sdf[a], nsb = -sdf[b];                 // renderer.py:156
            const float den = sa + nsb;                               // :158
            const float w0 = nsb / den, w1 = sa / den;                // :160 (flip, then divide)
#pragma unroll
            for (uint32_t d = 0; d < 3; d++) verts[(size_t)id * 3 + d] = pos[(size_t)a * 3 + d] * w0 + pos[(size_t)b * 3 + d] * w1;   // :161
            vert_edges[(size_t)id * 2] = a; vert_edges[(size_t)id * 2 + 1] = b;
        }
        id++;
    }
}

// pass 3b: faces. One-triangle tetrahedra fill rows [0, F1), two-triangle ones rows [F1, F1 + 2 F2), each in grid order
__global__ __launch_bounds__(kBlock) void k_mt_emit_faces(const float* __restrict__ sdf, const int32_t* __restrict__ tets,
                                                           const int32_t* __restrict__ tet_edges, uint32_t F,
                                                           const uint32_t* __restrict__ off1, const uint32_t* __restrict__ off2,
                                                           const int32_t* __restrict__ totals, const int32_t* __restrict__ edge_vid,
                                                           int32_t* __restrict__ faces, uint32_t cap_faces) {
    __shared__ uint32_t sh[kBlock / 64];
    uint32_t idx[kPer], n[kPer], c1 = 0, c2 = 0;
#pragma unroll
    for (uint32_t j = 0; j < kPer; j++) {
        const uint32_t t = blockIdx.x * kChunk + threadIdx.x * kPer + j;
        idx[j] = 0; n[j] = 0;
        if (t < F) { idx[j] = tet_index(sdf, tets, t); n[j] = c_ntri[idx[j]]; }
        c1 += n[j] == 1u; c2 += n[j] == 2u;
    }
    uint32_t r1 = off1[blockIdx.x] + block_excl(c1, sh);
    uint32_t r2 = off2[blockIdx.x] + block_excl(c2, sh);
    const uint32_t F1 = (uint32_t)totals[1];
#pragma unroll
    for (uint32_t j = 0; j < kPer; j++) {
        const uint32_t t = blockIdx.x * kChunk + threadIdx.x * kPer + j;
        if (t >= F || n[j] == 0u) continue;
        const uint32_t row = n[j] == 1u ? r1 : F1 + 2u * r2;
        for (uint32_t q = 0; q < 3u * n[j]; q++) {
            const int32_t v = edge_vid[tet_edges[(size_t)t * 6 + (uint32_t)c_tri[idx[j]][q]]];   // renderer.py:170-172
            const uint32_t o = row * 3u + q;
            if (o < cap_faces * 3u) faces[o] = v;
        }
        if (n[j] == 1u) r1++; else r2++;
    }
}

// backward of the interpolation: v = p_a w0 + p_b w1, w0 = -s_b / den, w1 = s_a / den, den = s_a - s_b
//   d/dp_a = g w0, d/dp_b = g w1, d/ds_a = (s_b / den^2) g.(p_a - p_b), d/ds_b = (s_a / den^2) g.(p_b - p_a)
// accumulated with float atomics (a grid vertex has ~14 edges, a few of them crossing)
__global__ __launch_bounds__(256) void k_mt_backward(const float* __restrict__ gverts, uint32_t V, const int32_t* __restrict__ vert_edges,
                                                      const float* __restrict__ pos, const float* __restrict__ sdf,
                                                      float* __restrict__ gpos, float* __restrict__ gsdf) {
    const uint32_t i = blockIdx.x * 256 + threadIdx.x;
    if (i >= V) return;
    const int32_t a = vert_edges[2 * i], b = vert_edges[2 * i + 1];
    const float sa = sdf[a], sb = sdf[b];
    const float den = sa - sb, w0 = -sb / den, w1 = sa / den, inv2 = 1.0f / (den * den);
    float dot = 0.f;
#pragma unroll
    for (uint32_t d = 0; d < 3; d++) {
        const float g = gverts[(size_t)i * 3 + d];
        dot += g * (pos[(size_t)a * 3 + d] - pos[(size_t)b * 3 + d]);
        if (gpos) {
            atomicAdd(&gpos[(size_t)a * 3 + d], g * w0);
            atomicAdd(&gpos[(size_t)b * 3 + d], g * w1);
        }
    }
    if (gsdf) {
        atomicAdd(&gsdf[a], sb * inv2 * dot);
        atomicAdd(&gsdf[b], -sa * inv2 * dot);
    }
}

}  // namespace

extern "C" {

uint64_t sdfx_marching_tets_scratch_bytes(uint32_t E, uint32_t F) {
    return ((uint64_t)div_up(E ? E : 1, kChunk) + 2ull * div_up(F ? F : 1, kChunk)) * sizeof(uint32_t);
}

/* pass 1 + 2: counts[0..2] = (vertices, one-triangle tetrahedra, two-triangle tetrahedra); the per-workgroup offsets stay in
 * `scratch` for sdfx_marching_tets_emit (same sdf, same grid) */
int sdfx_marching_tets_count(const float* sdf, const int32_t* edges, uint32_t E, const int32_t* tets, uint32_t F, void* scratch,
                             int32_t* counts, sdfx_stream_t stream) {
    SDFX_REQUIRE(sdf && edges && tets && scratch && counts, "marching_tets_count: null pointer");
    SDFX_REQUIRE(E > 0 && F > 0, "marching_tets_count: empty grid");
    const uint32_t nbE = div_up(E, kChunk), nbF = div_up(F, kChunk);
    hipStream_t st = as_stream(stream);
    uint32_t* c = static_cast<uint32_t*>(scratch);
    hipLaunchKernelGGL(k_mt_count, dim3(nbE + nbF), dim3(kBlock), 0, st, sdf, edges, E, tets, F, nbE, c);
    hipLaunchKernelGGL(k_mt_scan, dim3(1), dim3(1024), 0, st, c, nbE, nbF, counts);
    return check_launch("marching_tets_count");
}

/* pass 3: edge_vid [E], verts [cap_verts, 3], vert_edges [cap_verts, 2], faces [cap_faces, 3] (int32); capacities >= the counts */
int sdfx_marching_tets_emit(const float* pos, const float* sdf, const int32_t* edges, uint32_t E, const int32_t* tets,
                            const int32_t* tet_edges, uint32_t F, const void* scratch, const int32_t* counts, int32_t* edge_vid,
                            float* verts, int32_t* vert_edges, uint32_t cap_verts, int32_t* faces, uint32_t cap_faces,
                            sdfx_stream_t stream) {
    SDFX_REQUIRE(pos && sdf && edges && tets && tet_edges && scratch && counts && edge_vid, "marching_tets_emit: null pointer");
    SDFX_REQUIRE((cap_verts == 0 || (verts && vert_edges)) && (cap_faces == 0 || faces), "marching_tets_emit: null output");
    const uint32_t nbE = div_up(E, kChunk), nbF = div_up(F, kChunk);
    hipStream_t st = as_stream(stream);
    const uint32_t* c = static_cast<const uint32_t*>(scratch);
    hipLaunchKernelGGL(k_mt_emit_verts, dim3(nbE), dim3(kBlock), 0, st, pos, sdf, edges, E, c, edge_vid, verts, vert_edges, cap_verts);
    hipLaunchKernelGGL(k_mt_emit_faces, dim3(nbF), dim3(kBlock), 0, st, sdf, tets, tet_edges, F, c + nbE, c + nbE + nbF, counts, edge_vid,
                       faces, cap_faces);
    return check_launch("marching_tets_emit");
}

/* gradients of a loss through the vertex positions: grad_pos [N, 3] and grad_sdf [N] are ADDED to (zero them first); either may be NULL */
int sdfx_marching_tets_backward(const float* grad_verts, uint32_t V, const int32_t* vert_edges, const float* pos, const float* sdf,
                                float* grad_pos, float* grad_sdf, sdfx_stream_t stream) {
    if (V == 0) return SDFX_OK;
    SDFX_REQUIRE(grad_verts && vert_edges && pos && sdf, "marching_tets_backward: null pointer");
    hipLaunchKernelGGL(k_mt_backward, dim3(div_up(V, 256)), dim3(256), 0, as_stream(stream), grad_verts, V, vert_edges, pos, sdf, grad_pos,
                       grad_sdf);
    return check_launch("marching_tets_backward");
}

}  // extern "C"

// grid_common.h — launch plan, table element types and per-corner arithmetic shared by the
// grid-encoder translation units (gridencoder.hip, gridencoder_bwd_binned.hip).
#pragma once

#include "sdfx_common.h"

#include <math.h>

namespace sdfx {
namespace grid {

constexpr uint32_t kMaxLevels = 32;
constexpr uint32_t kXcds = 8;
constexpr uint32_t kTile = 256;  // work items per workgroup

struct GridPlan {
    uint32_t res[kMaxLevels];      // per-level resolution
    uint32_t off[kMaxLevels + 1];  // per-level first row (host copy of `offsets`)
    uint32_t start[kXcds];         // [start, end) item range of each XCD in the virtual-level-major list
    uint32_t end[kXcds];
    uint32_t order[kMaxLevels];    // virtual level -> level (fine and coarse levels interleaved)
    uint32_t tiles;                // tiles per level
    uint32_t vec16;                // table base is 16-byte aligned: paired 16-byte gathers allowed
};

// (uint32_t)ceil(exp2f(level * S) * H) in float32 — gridencoder.cu:133
inline uint32_t level_resolution(uint32_t level, float S, uint32_t H) {
    return (uint32_t)ceilf(exp2f((float)level * S) * (float)H);
}

// Build the plan: resolutions, offsets, and the per-XCD ranges of the item list.
//
// Items are (virtual level, tile) pairs in virtual-level-major order, cut into 8 equal contiguous ranges, one
// per XCD. The virtual order interleaves the levels from both ends — [L-1, 0, L-2, 1, ...] — because the
// cost of a level is the number of distinct 128-byte table lines a wave touches (measured: ~2.4 CU-cycles per
// line, independent of the access width; tools/ubench/gather_width.hip): fine levels cost 4 lines per sample,
// coarse levels a fraction of one when neighbouring lanes are neighbouring samples. Pairing a fine with a
// coarse level gives every XCD the same load (and, with 16 levels, exactly two levels = at most 4 MiB of fp16
// table in its 4 MiB L2).
inline GridPlan make_plan(const int32_t* offsets_host, uint32_t levels, float S, uint32_t H, uint32_t C, uint32_t elem_bytes,
                   uint64_t items_per_level) {
    (void)C; (void)elem_bytes;
    GridPlan p;
    memset(&p, 0, sizeof(p));
    for (uint32_t l = 0; l < levels; l++) {
        p.res[l] = level_resolution(l, S, H);
        p.off[l] = (uint32_t)offsets_host[l];
    }
    p.off[levels] = (uint32_t)offsets_host[levels];
    p.tiles = div_up(items_per_level, kTile);
    for (uint32_t v = 0, lo = 0, hi = levels; v < levels; v++) p.order[v] = (v & 1u) ? lo++ : --hi;
    const uint64_t total = (uint64_t)levels * p.tiles;
    for (uint32_t k = 0; k < kXcds; k++) {
        p.start[k] = (uint32_t)(total * k / kXcds);
        p.end[k] = (uint32_t)(total * (k + 1) / kXcds);
    }
    return p;
}

inline uint32_t plan_grid_size(const GridPlan& p) {
    uint32_t longest = 0;
    for (uint32_t k = 0; k < kXcds; k++) {
        const uint32_t len = p.end[k] - p.start[k];
        if (len > longest) longest = len;
    }
    return longest * kXcds;
}

// workgroup `block` of the launch -> (level, tile); false when it has no item. Workgroups are dealt to the XCDs round-robin
// (block b -> XCD b mod 8; checked at start-up, see sdfx_xcd_round_robin), so XCD k walks its own range [start[k], end[k]).
__device__ __forceinline__ bool plan_item(const GridPlan& p, uint32_t block, uint32_t& level, uint32_t& tile) {
    const uint32_t xcd = block % kXcds;
    const uint32_t local = block / kXcds;
    if (local >= p.end[xcd] - p.start[xcd]) return false;
    const uint32_t item = p.start[xcd] + local;
    const uint32_t virt = item / p.tiles;
    level = p.order[virt];
    tile = item - virt * p.tiles;
    return true;
}
__device__ __forceinline__ bool plan_item(const GridPlan& p, uint32_t& level, uint32_t& tile) { return plan_item(p, blockIdx.x, level, tile); }

// ---- forward of the hot-path configuration (gridencoder_fwd.hip) ----
bool fast_forward_enabled();
bool launch_forward_d3c2(const float* inputs, const void* table, const int32_t* offsets_host, void* outputs, uint32_t B,
                         uint32_t L, uint32_t max_level, float S, uint32_t H, uint32_t gridtype, int align_corners,
                         uint32_t interp, int is_half, int out_layout, uint32_t slabs, float step, hipStream_t st);

// ---- table element types ----------------------------------------------------------------
template <bool HALF> struct Elem;
template <> struct Elem<false> {
    using type = float;
    static __device__ __forceinline__ float load(const float* p) { return *p; }
    static __device__ __forceinline__ void store(float* p, float v) { *p = v; }
    // value as the reference's scalar_t would hold it
    static __device__ __forceinline__ float round(float v) { return v; }
};
template <> struct Elem<true> {
    using type = __half;
    static __device__ __forceinline__ float load(const __half* p) { return __half2float(*p); }
    static __device__ __forceinline__ void store(__half* p, float v) { *p = __float2half_rn(v); }
    // float -> half -> float with the conversion kept opaque to the optimiser: written as plain casts,
    // hipcc folds `half(w * float(g))` into v_fma_mixlo_f16, which rounds the exact product ONCE, whereas the
    // reference's at::Half arithmetic rounds the product to float32 first and then to half (a 1-ulp
    // difference in ~0.1 % of features; found by running the reference kernel itself, oracle/_ref).
    static __device__ __forceinline__ float round(float v) {
        _Float16 h;
        asm volatile("v_cvt_f16_f32 %0, %1" : "=v"(h) : "v"(v));
        return (float)h;
    }
};

// One vertex row = C elements. Rows are loaded / stored as a few wide words.
template <typename T, uint32_t C>
struct Row {
    static constexpr uint32_t kBytes = sizeof(T) * C;
    static constexpr uint32_t kWord = kBytes >= 16 ? 16 : kBytes;  // bytes per access (2..16)
    static constexpr uint32_t kWords = kBytes / kWord;
    T v[C];

    __device__ __forceinline__ void load(const T* p) {
        if constexpr (kWord == 16) {
#pragma unroll
            for (uint32_t i = 0; i < kWords; i++) reinterpret_cast<uint4*>(v)[i] = reinterpret_cast<const uint4*>(p)[i];
        } else if constexpr (kWord == 8) {
            *reinterpret_cast<uint2*>(v) = *reinterpret_cast<const uint2*>(p);
        } else if constexpr (kWord == 4) {
            *reinterpret_cast<uint32_t*>(v) = *reinterpret_cast<const uint32_t*>(p);
        } else {
            v[0] = p[0];
        }
    }
    __device__ __forceinline__ void store(T* p) const {
        if constexpr (kWord == 16) {
#pragma unroll
            for (uint32_t i = 0; i < kWords; i++) reinterpret_cast<uint4*>(p)[i] = reinterpret_cast<const uint4*>(v)[i];
        } else if constexpr (kWord == 8) {
            *reinterpret_cast<uint2*>(p) = *reinterpret_cast<const uint2*>(v);
        } else if constexpr (kWord == 4) {
            *reinterpret_cast<uint32_t*>(p) = *reinterpret_cast<const uint32_t*>(v);
        } else {
            p[0] = v[0];
        }
    }
};

// corner `idx` of the cell: weight and vertex coordinates (gridencoder.cu:171-184)
template <uint32_t D>
__device__ __forceinline__ float corner(uint32_t idx, const float pos[D], const uint32_t pos_grid[D], uint32_t resolution,
                                        uint32_t pgl[D]) {
    float w = 1;
#pragma unroll
    for (uint32_t d = 0; d < D; d++) {
        if ((idx & (1u << d)) == 0) {
            w *= 1 - pos[d];
            pgl[d] = pos_grid[d];
        } else {
            w *= pos[d];
            pgl[d] = min(pos_grid[d] + 1, resolution - 1);
        }
    }
    return w;
}


}  // namespace grid
}  // namespace sdfx

// grid_point.h — pieces of the D = 3, C = 2 hash-grid forward shared by csrc/gridencoder_fwd.hip (one (point, level) per
// thread, level-major over the XCDs) and csrc/infer.hip (all levels of a sample in one thread): the per-level constants, the
// reference-exact accumulation (gridencoder.cu:168-195) and the evaluation of one level at one point.
#pragma once

#include "grid_common.h"

namespace sdfx {
namespace grid {

struct LevelConst {   // 32 bytes, one s_load_dwordx8
    uint32_t res, row0, size, m1, m2, flags, pad0, pad1;   // flags: 1 = hashed, 2 = size is a power of two, 4 = 24-bit multiplies give the rows
};

// per-level constants from the host copy of `offsets` (gridencoder.cu:61-79, 133)
inline LevelConst make_level_const(const int32_t* offsets_host, uint32_t level, float S, uint32_t H) {
    LevelConst c;
    memset(&c, 0, sizeof(c));
    c.res = level_resolution(level, S, H);
    c.row0 = (uint32_t)offsets_host[level];
    c.size = (uint32_t)offsets_host[level + 1] - c.row0;
    uint64_t stride = 1;   // dense index while the strides fit (d = 0 always does)
    stride *= c.res;
    if (stride <= c.size) { c.m1 = (uint32_t)stride; stride *= c.res; }
    if (stride <= c.size) { c.m2 = (uint32_t)stride; stride *= c.res; }
    c.flags = (stride > c.size ? 1u : 0u) | ((c.size & (c.size - 1u)) == 0u ? 2u : 0u);
    // level_prepare_uniform: full-rate v_mul_u32_u24 instead of the quarter-rate 32-bit multiply. Dense rows need the exact products
    // (coordinate < res and strides < 2^24, product < size < 2^32); hashed rows only the product's bits below log2(size), which a
    // 24 x 24-bit product has right when the size is a power of two of at most 2^24 (the primes are taken modulo 2^24)
    const bool small = c.res < (1u << 24) && c.m1 < (1u << 24) && c.m2 < (1u << 24);
    if ((c.flags & 1u) ? ((c.flags & 2u) && c.size <= (1u << 24) && c.res < (1u << 24)) : small) c.flags |= 4u;
    return c;
}

typedef _Float16 half2_t __attribute__((ext_vector_type(2)));
typedef float float2_t __attribute__((ext_vector_type(2)));

// results[ch] += w * grid[index + ch] with results and grid of the table's type (gridencoder.cu:168-195).
// Half tables: the reference converts the float32 product to at::Half, then adds two at::Half values — in float32,
// rounded to half again. v_cvt_pk_f16_f32 is the first rounding (kept opaque so hipcc cannot fuse product and rounding
// into v_fma_mixlo_f16, which rounds the exact product once). v_pk_add_f16 is the second: the float32 sum of two halves
// rounded to half equals the correctly rounded half sum, because float32's 24 significand bits >= 2 * 11 + 2 (double
// rounding is innocuous; tests/test_hostmath.py checks the identity over all exponent pairs).
template <bool HALF> struct Acc2;
template <> struct Acc2<true> {
    half2_t acc = {(_Float16)0.0f, (_Float16)0.0f};
    __device__ __forceinline__ void add(float w, uint32_t row) {
        const half2_t g = __builtin_bit_cast(half2_t, row);
        const float2_t gf = {(float)g.x, (float)g.y};
        const float2_t p = gf * w;
        half2_t ph;
        asm volatile("v_cvt_pk_f16_f32 %0, %1, %2" : "=v"(ph) : "v"(p.x), "v"(p.y));
        acc = acc + ph;
    }
    __device__ __forceinline__ void store(__half* out, bool zero) const {
        *reinterpret_cast<uint32_t*>(out) = zero ? 0u : __builtin_bit_cast(uint32_t, acc);
    }
};
template <> struct Acc2<false> {
    float a0 = 0.0f, a1 = 0.0f;
    __device__ __forceinline__ void add(float w, uint2 row) {
        a0 = a0 + w * __uint_as_float(row.x);
        a1 = a1 + w * __uint_as_float(row.y);
    }
    __device__ __forceinline__ void store(float* out, bool zero) const {
        *reinterpret_cast<float2*>(out) = zero ? make_float2(0.f, 0.f) : make_float2(a0, a1);
    }
};

__device__ __forceinline__ uint32_t pick4(const uint4& b, uint32_t j) {   // dword j (0..3) of a 16-byte block
    const uint32_t lo = (j & 2u) ? b.z : b.x;
    const uint32_t hi = (j & 2u) ? b.w : b.y;
    return (j & 1u) ? hi : lo;
}


// One level at one point, in three phases so that a caller can keep the gathers of SEVERAL levels in flight together:
//   level_prepare  cell, interpolation weights and the row indices of the four x-pairs (no memory access)
//   level_gather   the gathers (one 16-byte block per x-pair + the odd second row)
//   level_reduce   the reference-exact accumulation (gridencoder.cu:168-195), corners in the reference's order
// `x` in [0, 1]^3 (the caller handles out-of-range points: gridencoder.cu:105-130 writes zeros for them).
struct LevelPoint {
    uint32_t r0[4], r1[4];     // rows of the corners (x, y, z) and (x + 1, y, z) for the four (y, z) corners
    float ax1, ay1, az1;       // interpolation weights of the "+1" vertices (the others are 1 - these)
    uint32_t cx, cy, cz;       // the cell (lower vertex) — only the binned scatter looks at it
};
struct LevelData {
    uint4 blk[4];
    uint32_t extra[4];
};

template <uint32_t INTERP, bool ALIGN, bool HASHGRID>
__device__ __forceinline__ void level_prepare(const LevelConst& lc, const float x[3], LevelPoint& p) {
    const bool hashed = HASHGRID && (lc.flags & 1u);
    const bool pow2 = (lc.flags & 2u) != 0u;
    float pos[3], deriv;
    uint32_t pg[3], pn[3];
#pragma unroll
    for (int d = 0; d < 3; d++) {
        grid_locate_axis(x[d], lc.res, ALIGN, INTERP, pos[d], deriv, pg[d]);
        pg[d] = min(pg[d], lc.res - 1u);
        pn[d] = min(pg[d] + 1u, lc.res - 1u);
    }
    (void)deriv;
    p.ax1 = pos[0]; p.ay1 = pos[1]; p.az1 = pos[2];
    p.cx = pg[0]; p.cy = pg[1]; p.cz = pg[2];
    // Branch-free in the level's kind (both the hashed and the dense index are formed and one is masked in): this function is
    // inlined once per level of a batch, and control flow on per-level flags made the compiler unswitch the batch loop into
    // every combination of kinds (12 000 instructions, more than the instruction cache holds).
    const bool need_mod = (lc.flags & 3u) == 1u;   // not fully dense and not a power of two (flags: make_level_const)
    const uint32_t hm = hashed ? 0xffffffffu : 0u;
    const uint32_t wm = pow2 ? lc.size - 1u : 0xffffffffu;
    const uint32_t hy[2] = {pg[1] * 2654435761u, pn[1] * 2654435761u};
    const uint32_t hz[2] = {pg[2] * 805459861u, pn[2] * 805459861u};
    const uint32_t sy[2] = {pg[1] * lc.m1, pn[1] * lc.m1};
    const uint32_t sz[2] = {pg[2] * lc.m2, pn[2] * lc.m2};
#pragma unroll
    for (int k = 0; k < 4; k++) {
        const uint32_t yh = hy[k & 1] ^ hz[k >> 1], yd = sy[k & 1] + sz[k >> 1];
        uint32_t i0 = (((pg[0] ^ yh) & hm) | ((pg[0] + yd) & ~hm)) & wm;
        uint32_t i1 = (((pn[0] ^ yh) & hm) | ((pn[0] + yd) & ~hm)) & wm;
        // index % hashmap_size (gridencoder.cu:78): the mask above for a power-of-two size; a fully dense level (strides of all
        // three axes fit: x + y res + z res^2 < res^3 <= size) never exceeds its size; what is left — a level that is hashed, or
        // tiled with a truncated stride, AND whose size is no power of two — is a property of the level, decided on the scalar unit
        if (need_mod) { i0 %= lc.size; i1 %= lc.size; }
        p.r0[k] = i0; p.r1[k] = i1;
    }
}

// The same rows and weights with the level's KIND decided by (wave-uniform) control flow — for kernels that handle one level per
// workgroup (the binned scatter): no second set of products, and 24-bit multiplies where the level allows (flags & 4).
template <uint32_t INTERP, bool ALIGN, bool HASHGRID>
__device__ __forceinline__ void level_prepare_uniform(const LevelConst& lc, const float x[3], LevelPoint& p) {
    const bool hashed = HASHGRID && (lc.flags & 1u);
    const bool pow2 = (lc.flags & 2u) != 0u, mul24 = (lc.flags & 4u) != 0u;
    float pos[3], deriv;
    uint32_t pg[3], pn[3];
#pragma unroll
    for (int d = 0; d < 3; d++) {
        grid_locate_axis(x[d], lc.res, ALIGN, INTERP, pos[d], deriv, pg[d]);
        pg[d] = min(pg[d], lc.res - 1u);
        pn[d] = min(pg[d] + 1u, lc.res - 1u);
    }
    (void)deriv;
    p.ax1 = pos[0]; p.ay1 = pos[1]; p.az1 = pos[2];
    p.cx = pg[0]; p.cy = pg[1]; p.cz = pg[2];
    const bool need_mod = (lc.flags & 3u) == 1u;
    const uint32_t wm = pow2 ? lc.size - 1u : 0xffffffffu;
    uint32_t ty[2], tz[2];   // the y and z terms of the row
    const uint32_t fy = hashed ? 2654435761u : lc.m1, fz = hashed ? 805459861u : lc.m2;
    if (mul24) {
        ty[0] = __umul24(pg[1], fy & 0xFFFFFFu); ty[1] = __umul24(pn[1], fy & 0xFFFFFFu);
        tz[0] = __umul24(pg[2], fz & 0xFFFFFFu); tz[1] = __umul24(pn[2], fz & 0xFFFFFFu);
    } else {
        ty[0] = pg[1] * fy; ty[1] = pn[1] * fy;
        tz[0] = pg[2] * fz; tz[1] = pn[2] * fz;
    }
#pragma unroll
    for (int k = 0; k < 4; k++) {
        uint32_t i0, i1;
        if (hashed) {
            const uint32_t yz = ty[k & 1] ^ tz[k >> 1];
            i0 = (pg[0] ^ yz) & wm; i1 = (pn[0] ^ yz) & wm;
        } else {
            const uint32_t yz = ty[k & 1] + tz[k >> 1];
            i0 = (pg[0] + yz) & wm; i1 = (pn[0] + yz) & wm;
        }
        if (need_mod) { i0 %= lc.size; i1 %= lc.size; }   // (see level_prepare)
        p.r0[k] = i0; p.r1[k] = i1;
    }
}

__device__ __forceinline__ void level_gather(const __half* __restrict__ table, const LevelConst& lc, const LevelPoint& p, bool vec16,
                                             LevelData& d) {
    const __half* tab = table + (size_t)lc.row0 * 2;
    if (vec16) {
#pragma unroll
        for (int k = 0; k < 4; k++) d.blk[k] = *reinterpret_cast<const uint4*>(tab + (size_t)(p.r0[k] & ~3u) * 2);
#pragma unroll
        for (int k = 0; k < 4; k++) {
            d.extra[k] = 0u;
            if ((p.r0[k] ^ p.r1[k]) >= 4u) d.extra[k] = *reinterpret_cast<const uint32_t*>(tab + (size_t)p.r1[k] * 2);
        }
    } else {   // table not 16-byte aligned: one gather per corner; blk[k].x / .y carry the two rows
#pragma unroll
        for (int k = 0; k < 4; k++) {
            d.blk[k].x = *reinterpret_cast<const uint32_t*>(tab + (size_t)p.r0[k] * 2);
            d.blk[k].y = *reinterpret_cast<const uint32_t*>(tab + (size_t)p.r1[k] * 2);
            d.extra[k] = 0u;
        }
    }
}

__device__ __forceinline__ uint32_t level_reduce(const LevelPoint& p, const LevelData& d, bool vec16) {
    const float ax[2] = {1 - p.ax1, p.ax1}, ay[2] = {1 - p.ay1, p.ay1}, az[2] = {1 - p.az1, p.az1};
    Acc2<true> acc;
#pragma unroll
    for (int k = 0; k < 4; k++) {   // corner order of gridencoder.cu:168-195 (x fastest); weights ((1 * a_x) * a_y) * a_z
        uint32_t v0, v1;
        if (vec16) {
            v0 = pick4(d.blk[k], p.r0[k] & 3u);
            v1 = (p.r0[k] ^ p.r1[k]) >= 4u ? d.extra[k] : pick4(d.blk[k], p.r1[k] & 3u);
        } else {
            v0 = d.blk[k].x; v1 = d.blk[k].y;
        }
        const float wy = ay[k & 1], wz = az[k >> 1];
        acc.add(((1 * ax[0]) * wy) * wz, v0);
        acc.add(((1 * ax[1]) * wy) * wz, v1);
    }
    return __builtin_bit_cast(uint32_t, acc.acc);
}

}  // namespace grid
}  // namespace sdfx

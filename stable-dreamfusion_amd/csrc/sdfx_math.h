// sdfx_math.h — per-sample arithmetic shared by the gfx950 kernels.
//
// Everything here is written so that, compiled with -ffp-contract=off, it performs the same
// IEEE-754 operations in the same order as the expressions of the reference kernels it
// stands in for (citations inline; paths relative to the reference checkout).  The functions
// are SDFX_HD so that tests/hostmath can also build them with g++ and compare them against
// the independent C oracle without a GPU; the shipped library only ever calls them from
// device code.
#pragma once

#include <stdint.h>
#include <math.h>

#if defined(__HIPCC__)
#define SDFX_HD __host__ __device__ __forceinline__
#else
#define SDFX_HD inline
#endif

namespace sdfx {

constexpr float kSqrt3 = 1.7320508075688772f;  // raymarching.cu:19
constexpr float kPi = 3.141592653589793f;      // raymarching.cu:21
constexpr float kRPi = 0.3183098861837907f;    // raymarching.cu:22

SDFX_HD float signf_(float x) { return copysignf(1.0f, x); }                          // raymarching.cu:30-32
SDFX_HD float clampf_(float x, float lo, float hi) { return fminf(hi, fmaxf(lo, x)); } // raymarching.cu:34-36

// frexpf exponent of a non-negative float: [0.5,1) -> 0, [1,2) -> 1, 0 -> 0 (raymarching.cu:45)
SDFX_HD int frexp_exponent(float v) {
    int e;
    (void)frexpf(v, &e);
    return e;
}

SDFX_HD int mip_from_pos(float x, float y, float z, float max_cascade) {  // raymarching.cu:42-47
    const float mx = fmaxf(fabsf(x), fmaxf(fabsf(y), fabsf(z)));
    return (int)fminf(max_cascade - 1, fmaxf(0.0f, (float)frexp_exponent(mx)));
}

SDFX_HD int mip_from_dt(float dt, float H, float max_cascade) {  // raymarching.cu:49-54
    const float mx = (float)((double)(dt * H) * 0.5);
    return (int)fminf(max_cascade - 1, fmaxf(0.0f, (float)frexp_exponent(mx)));
}

SDFX_HD uint32_t expand_bits(uint32_t v) {  // raymarching.cu:56-63
    v = (v * 0x00010001u) & 0xFF0000FFu;
    v = (v * 0x00000101u) & 0x0F00F00Fu;
    v = (v * 0x00000011u) & 0xC30C30C3u;
    v = (v * 0x00000005u) & 0x49249249u;
    return v;
}

SDFX_HD uint32_t morton3D(uint32_t x, uint32_t y, uint32_t z) {  // raymarching.cu:65-71
    return expand_bits(x) | (expand_bits(y) << 1) | (expand_bits(z) << 2);
}

SDFX_HD uint32_t morton3D_invert(uint32_t x) {  // raymarching.cu:73-81
    x = x & 0x49249249u;
    x = (x | (x >> 2)) & 0xc30c30c3u;
    x = (x | (x >> 4)) & 0x0f00f00fu;
    x = (x | (x >> 8)) & 0xff0000ffu;
    x = (x | (x >> 16)) & 0x0000ffffu;
    return x;
}

// ---------------------------------------------------------------------------------------
// Occupancy-grid DDA (raymarching.cu:396-464 and :760-827 share this body).
// ---------------------------------------------------------------------------------------
struct MarchRay {
    float ox, oy, oz, dx, dy, dz, rdx, rdy, rdz;
};

struct MarchParams {
    float bound, dt_gamma, dt_min, dt_max, rH, H3, Hf, Cf;
    uint32_t H;
    int contract;
};

SDFX_HD MarchParams make_march_params(float bound, int contract, float dt_gamma, uint32_t max_steps, uint32_t C,
                                      uint32_t H) {
    MarchParams p;
    p.bound = bound;
    p.dt_gamma = dt_gamma;
    p.dt_min = 2 * kSqrt3 / (float)max_steps;  // raymarching.cu:385
    p.dt_max = 2 * kSqrt3 * bound / (float)H;  // raymarching.cu:386
    p.rH = 1 / (float)H;                       // raymarching.cu:378
    p.H3 = (float)(H * H * H);                 // raymarching.cu:379 (uint32 product, then float)
    p.Hf = (float)H;
    p.Cf = (float)C;
    p.H = H;
    p.contract = contract;
    return p;
}

SDFX_HD MarchRay make_march_ray(const float* o, const float* d) {
    MarchRay r;
    r.ox = o[0]; r.oy = o[1]; r.oz = o[2];
    r.dx = d[0]; r.dy = d[1]; r.dz = d[2];
    r.rdx = 1 / r.dx; r.rdy = 1 / r.dy; r.rdz = 1 / r.dz;  // raymarching.cu:377
    return r;
}

// Sample position for ray time t: clamp to the box, optional L-inf contraction
// (raymarching.cu:398-400, 411-419).  Returns mag = max |coordinate| before contraction.
SDFX_HD float march_position(const MarchRay& r, const MarchParams& p, float t, float& cx, float& cy, float& cz) {
    const float x = clampf_(r.ox + t * r.dx, -p.bound, p.bound);
    const float y = clampf_(r.oy + t * r.dy, -p.bound, p.bound);
    const float z = clampf_(r.oz + t * r.dz, -p.bound, p.bound);
    cx = x; cy = y; cz = z;
    const float mag = fmaxf(fabsf(x), fmaxf(fabsf(y), fabsf(z)));
    if (p.contract && mag > 1) {
        const float Linf_scale = (2 - 1 / mag) / mag;
        cx *= Linf_scale;
        cy *= Linf_scale;
        cz *= Linf_scale;
    }
    return mag;
}

SDFX_HD float march_dt(const MarchParams& p, float t) { return clampf_(t * p.dt_gamma, p.dt_min, p.dt_max); }

// One probe of the DDA at ray time t. Returns true if the cell is occupied (the caller
// emits the sample and advances by dt); otherwise advances t past the empty voxel.
// `grid` is the packed occupancy bitfield (bit i of byte b = cell 8b+i, raymarching.cu:427).
template <typename GridPtr>
SDFX_HD bool march_probe(const MarchRay& r, const MarchParams& p, GridPtr grid, float& t, float& dt, float& cx,
                         float& cy, float& cz, uint32_t* advanced = nullptr) {
    // position is clamped BEFORE the level is chosen (raymarching.cu:398-405)
    const float x = clampf_(r.ox + t * r.dx, -p.bound, p.bound);
    const float y = clampf_(r.oy + t * r.dy, -p.bound, p.bound);
    const float z = clampf_(r.oz + t * r.dz, -p.bound, p.bound);

    dt = clampf_(t * p.dt_gamma, p.dt_min, p.dt_max);

    const int la = mip_from_pos(x, y, z, p.Cf);
    const int lb = mip_from_dt(dt, p.Hf, p.Cf);
    const int level = la > lb ? la : lb;

    const float mip_bound = fminf(scalbnf(1.0f, level), p.bound);
    const float mip_rbound = 1 / mip_bound;

    cx = x; cy = y; cz = z;
    const float mag = fmaxf(fabsf(x), fmaxf(fabsf(y), fabsf(z)));
    if (p.contract && mag > 1) {
        const float Linf_scale = (2 - 1 / mag) / mag;
        cx *= Linf_scale;
        cy *= Linf_scale;
        cz *= Linf_scale;
    }

    // `0.5 * (cx * mip_rbound + 1) * H`: the 0.5 literal makes this a double expression in
    // the reference (raymarching.cu:422-424); the clamp then narrows it back to float.
    const float Hm1 = (float)(p.H - 1);
    const int nx = (int)clampf_((float)(0.5 * (double)(cx * mip_rbound + 1) * (double)p.H), 0.0f, Hm1);
    const int ny = (int)clampf_((float)(0.5 * (double)(cy * mip_rbound + 1) * (double)p.H), 0.0f, Hm1);
    const int nz = (int)clampf_((float)(0.5 * (double)(cz * mip_rbound + 1) * (double)p.H), 0.0f, Hm1);

    // float index arithmetic, as in the reference (H3 is a float there, raymarching.cu:379,426)
    const uint32_t index = (uint32_t)((float)level * p.H3 + (float)morton3D((uint32_t)nx, (uint32_t)ny, (uint32_t)nz));
    const bool occ = (grid[index / 8] & (1 << (index % 8))) != 0;
    if (occ) return true;

    uint32_t hops = 0;  // how many times t was advanced (the wave-per-ray march needs the lattice index it lands on)
    if (p.contract && mag > 1) {
        t += dt;  // contraction: no voxel skipping (raymarching.cu:449-450)
        hops = 1;
    } else {
        // distance to the exit face of the current voxel (raymarching.cu:454-463)
        const float tx = (((nx + 0.5f + 0.5f * signf_(r.dx)) * p.rH * 2 - 1) * mip_bound - cx) * r.rdx;
        const float ty = (((ny + 0.5f + 0.5f * signf_(r.dy)) * p.rH * 2 - 1) * mip_bound - cy) * r.rdy;
        const float tz = (((nz + 0.5f + 0.5f * signf_(r.dz)) * p.rH * 2 - 1) * mip_bound - cz) * r.rdz;
        const float tt = t + fmaxf(0.0f, fminf(tx, fminf(ty, tz)));
        do {
            dt = clampf_(t * p.dt_gamma, p.dt_min, p.dt_max);
            t += dt;
            hops++;
        } while (t < tt);
    }
    if (advanced) *advanced = hops;
    return false;
}

// Every ray time the march ever visits lies on ONE sequence that does not depend on the occupancy grid:
// L[0] = start, L[k+1] = L[k] + clamp(L[k] * dt_gamma, dt_min, dt_max) — the emit path (raymarching.cu:446), the
// skip loop (:459-462) and the contraction path (:450) all advance t by exactly this expression of the current t.
SDFX_HD float march_advance(const MarchParams& p, float t) { return t + clampf_(t * p.dt_gamma, p.dt_min, p.dt_max); }

// ---------------------------------------------------------------------------------------
// Multi-resolution grid (gridencoder.cu:45-79, 133-160).
// ---------------------------------------------------------------------------------------
template <uint32_t D>
SDFX_HD uint32_t fast_hash(const uint32_t pos_grid[D]) {  // gridencoder.cu:45-58
    constexpr uint32_t primes[7] = {1u, 2654435761u, 805459861u, 3674653429u, 2097192037u, 1434869437u, 2165219737u};
    uint32_t result = 0;
#pragma unroll
    for (uint32_t i = 0; i < D; ++i) result ^= pos_grid[i] * primes[i];
    return result;
}

// row index (NOT multiplied by C) of a grid vertex: dense stride index while the level fits,
// spatial hash otherwise, modulo the level size (gridencoder.cu:61-79)
template <uint32_t D>
SDFX_HD uint32_t grid_row(uint32_t gridtype, uint32_t hashmap_size, uint32_t resolution, const uint32_t pos_grid[D]) {
    uint32_t stride = 1;
    uint32_t index = 0;
#pragma unroll
    for (uint32_t d = 0; d < D; d++) {
        if (stride <= hashmap_size) {
            index += pos_grid[d] * stride;
            stride *= resolution;
        }
    }
    if (gridtype == 0 && stride > hashmap_size) index = fast_hash<D>(pos_grid);
    // `index % hashmap_size` (gridencoder.cu:78) without the ~35-instruction software division in the two cases
    // that cover every level of the default configuration: a power-of-two level size (hashed levels are 2^19
    // rows) and an index already in range (dense levels: x + y*res + z*res^2 < res^3 <= size). Same value.
    if ((hashmap_size & (hashmap_size - 1u)) == 0u) return index & (hashmap_size - 1u);
    return index < hashmap_size ? index : index % hashmap_size;
}

SDFX_HD float smoothstep_(float v) { return v * v * (3.0f - 2.0f * v); }          // gridencoder.cu:34-37
SDFX_HD float smoothstep_derivative_(float v) { return 6 * v * (1.0f - v); }      // gridencoder.cu:39-42

// continuous -> (cell, fractional weight) for one axis (gridencoder.cu:143-159)
SDFX_HD void grid_locate_axis(float in, uint32_t resolution, bool align_corners, uint32_t interp, float& pos,
                              float& pos_deriv, uint32_t& pos_grid) {
    if (align_corners) {
        pos = in * (float)(resolution - 1);
        const uint32_t f = (uint32_t)floorf(pos);
        pos_grid = f < resolution - 2 ? f : resolution - 2;
    } else {
        pos = fminf(fmaxf(in * (float)resolution - 0.5f, 0.0f), (float)(resolution - 1));
        pos_grid = (uint32_t)floorf(pos);
    }
    pos -= (float)pos_grid;
    if (interp == 1) {
        pos_deriv = smoothstep_derivative_(pos);
        pos = smoothstep_(pos);
    } else {
        pos_deriv = 1.0f;
    }
}

// ---------------------------------------------------------------------------------------
// Exact accumulation of half-precision values (gridencoder_bwd_binned.hip, K2 / K3).
// Every finite half is an integer multiple of 2^-24 (the smallest subnormal) below 2^16, so value * 2^24 is an
// integer below 2^40: a 64-bit integer accumulator sums millions of them exactly, in any order.
// ---------------------------------------------------------------------------------------
SDFX_HD long long half_to_fixed(uint32_t h) {  // h: IEEE binary16 bit pattern of a FINITE value
    const uint32_t e = (h >> 10) & 31u, m = h & 1023u;
    const unsigned long long mag = (unsigned long long)(e ? (m | 1024u) : m) << (e ? e - 1u : 0u);
    return (h & 0x8000u) ? -(long long)mag : (long long)mag;
}
// |sum| < 2^63 * 2^-24; the double is exact up to 2^53 units, the float conversion rounds once
SDFX_HD float fixed_to_float(long long units) { return (float)((double)units * 0x1p-24); }

}  // namespace sdfx

// gridencoder_fwd.hip — the forward of the hash / tiled grid encoder for the configuration of the hot path
// (D = 3, C = 2, no dy_dx): same arithmetic as k_grid_forward (gridencoder.hip; reference gridencoder.cu:82-249),
// bit for bit, organised around what bounds it on MI355X.
//
// What bounds it (tools/ubench/gather_width.hip, gather_policy.hip; profiles/README.md): a divergent gather costs the
// CU's vector memory pipe ~2.4 cycles per DISTINCT 128-byte line per wave instruction — whatever the access width
// (4 / 8 / 16 bytes per lane), whatever the cache policy (sc0 / sc1 / nt), and whether the line hits in L1 or not.
// The eight corners of a sample at a hashed level are 4 such lines (the x-pair shares one), so the kernel's time is
// (distinct lines per wave, summed over its gathers) x 2.4 cycles, on the XCD that got the most of them. Therefore:
//
//   * fewer distinct lines per wave: the seven points of a finite-difference stencil (x, x +- e along each axis;
//     network_grid.py:81-96) are evaluated by NEIGHBOURING LANES — 9 samples x 7 points per wave — when the caller
//     says the batch is `slabs` = 7 slabs of B/7 points. x +- e land in the same 128-byte line as x at every level (the
//     x prime of the spatial hash is 1) and the other four share the coarse cells: -30 % lines on ray-ordered samples
//     (1657 instead of 2353 per 64 points over the 16 levels). Only the lane -> point map changes; layouts do not.
//   * one level per XCD at a time, the same load on every XCD: a hashed level is a 2 MiB fp16 table, an XCD's L2 is 4 MiB
//     and is shared with the streamed coordinates and features. Measured on ray-ordered stencil batches (B = 1.8 M):
//     an XCD walking two levels with alternating tiles is 15-20 % slower than walking them one after the other, even when
//     the second table is 20 KB (the gather-bound fine waves hold the wave slots the latency-bound coarse waves need);
//     fp32 tables (4 MiB a level) +50 %. So the levels form ONE SEQUENCE that is cut into 8 per-XCD ranges of equal
//     modelled COST (not equal tile count), each XCD walking its range in order: cost per wave =
//     max(distinct lines, VALU time in line units), lines from a model of ray-ordered samples (lines_per_wave) at the
//     caller's step hint. A level then costs 267 lines per wave at the fine end and 97 at the coarse end, and the even
//     two-levels-per-XCD split (275 vs 133 lines) becomes 8 equal shares: 335 -> 287 us. Without a hint every level
//     costs the same and the split is the even one.
//   * VALU off the critical path: level constants in scalar registers from one 32-byte record, hash / stride terms shared
//     by the x-pair, packed half arithmetic (v_pk_mul_f32, v_cvt_pk_f16_f32, v_pk_add_f16: bit-identical to at::Half's
//     round-after-every-operation, see Acc2), and all gathers of a lane issued before the first result is touched
//     (more points per thread were measured slower: 2 -> -2 %, 4 -> -10 %).
#include "grid_point.h"
#include "dev_stamps.h"

#include <math.h>
#include <stdlib.h>

#include <type_traits>

using namespace sdfx;
using namespace sdfx::grid;

namespace {

SDFX_DEV_CTL_DEFINE   // devtools build: per-workgroup timestamps (dev_stamps.h); nothing in the product build

constexpr uint32_t kMaxSegs = 6;       // per XCD
constexpr uint32_t kGroup = 7;         // points per stencil
constexpr uint32_t kGroupsPerWave = 9; // 9 x 7 = 63 lanes

constexpr uint32_t kNoLevel = 0xffffffffu;
// tiles [first, first + count) of `level`, `tpw` (>= 1) consecutive tiles per workgroup. A PAIR plan (FwdPlan::pair) evaluates a second
// level, `level2` (kNoLevel: none), on the same tiles in the same waves (k_grid_fwd_pair)
struct Seg { uint32_t level, first, count, tpw, level2; };
struct FwdPlan {
    LevelConst lv[kMaxLevels];
    Seg seg[kXcds][kMaxSegs];
    uint32_t ntiles[kXcds];   // workgroups of each XCD (sum of its segments)
    uint32_t vec16;           // bit l: level l gathers an x-pair with ONE 16-byte load (table base 16-byte aligned) + a select; clear: two
                              // 4-byte gathers — more texture-path work, fewer instructions: right where the level is VALU-bound
    uint32_t slabs;           // 1, or 7 = stencil batch [7, B/7, 3] evaluated with the 7 points of a sample in neighbouring lanes
    uint32_t slab_points;     // B / slabs
    // measurement aid (SDFX_GRID_PLAN=sample_major): every XCD takes 1/8 of the tiles and evaluates ALL levels of a tile before the
    // next tile — the table access pattern of a kernel that fuses the encode into the MLP (all 16 levels of a sample in one
    // place): no XCD's L2 can hold the 23 MB of tables. The A/B against the level-major default prices that fusion.
    uint32_t sample_major, tiles, levels;
    // measurement aid (SDFX_GRID_LDS=1): bit l set = level l's whole table (<= 48 KiB: levels 0 and 1 of the -O grid) is copied
    // into LDS by a workgroup that then walks kLdsTiles tiles of that level, gathering with ds_read instead of through the
    // texture-address path — north_star's "LDS staging", measured in profiles/r03_encode_lds_levels.txt
    uint32_t lds_mask, lds_bytes;
    // Round 6: every wave evaluates TWO levels of its tile — a fine one and a coarse one, (L-1, 0), (L-2, 1), ... — with the fine
    // level's gathers in flight while it forms the coarse level's rows (k_grid_fwd_pair; half tables). The units of the level sequence
    // are then these pairs, an XCD's L2 holds the two tables of the pair it is walking, and the coordinates of a tile are formed once
    // for both levels.
    uint32_t pair;
};
constexpr uint32_t kLdsTiles = 16;

// workgroup -> (level, tile) through the XCD's segment list (walked in order); false when there is nothing to do
__device__ __forceinline__ bool fwd_item(const FwdPlan& p, uint32_t& level, uint32_t& tile, uint32_t& seg_end, uint32_t& tpw, uint32_t& level2) {
    tpw = 1u; level2 = kNoLevel;
    const uint32_t xcd = blockIdx.x % kXcds;
    uint32_t local = blockIdx.x / kXcds;
    if (p.sample_major) {
        const uint32_t per = (p.tiles + kXcds - 1) / kXcds;
        tile = xcd * per + local / p.levels;
        level = p.levels - 1u - local % p.levels;
        seg_end = tile + 1;
        return local < per * p.levels && tile < p.tiles;
    }
    if (local >= p.ntiles[xcd]) return false;
#pragma unroll
    for (uint32_t s = 0; s < kMaxSegs; s++) {
        const Seg sg = p.seg[xcd][s];
        const uint32_t wgs = (sg.count + sg.tpw - 1u) / (sg.tpw ? sg.tpw : 1u);   // workgroups of the segment
        if (local < wgs) {
            level = sg.level; tile = sg.first + local * sg.tpw; seg_end = sg.first + sg.count; tpw = sg.tpw; level2 = sg.level2;
            // an LDS-resident level: one workgroup in kLdsTiles takes that many consecutive tiles, the others have nothing to do
            if ((p.lds_mask >> sg.level) & 1u) return local % kLdsTiles == 0;
            return true;
        }
        local -= wgs;
    }
    return false;
}

// two consecutive rows of a C = 2 table at the alignment of ONE row
template <bool HALF> struct PairT;
template <> struct __attribute__((packed, aligned(4))) PairT<true> { uint32_t a, b; };
template <> struct __attribute__((packed, aligned(8))) PairT<false> { uint2 a, b; };

template <bool HALF, uint32_t INTERP, bool ALIGN, bool HASHGRID, bool LDS>
__global__ __launch_bounds__(kTile) void k_grid_fwd(const float* __restrict__ inputs,
                                                     const typename Elem<HALF>::type* __restrict__ table,
                                                     typename Elem<HALF>::type* __restrict__ outputs, uint32_t B, uint32_t L,
                                                     FwdPlan plan, int out_layout, const int32_t* __restrict__ row_total,
                                                     StencilSrc src) {
    using T = typename Elem<HALF>::type;
    constexpr uint32_t C = 2;
    constexpr uint32_t RB = HALF ? 4u : 2u;   // rows per 16-byte block
    using RowT = typename std::conditional<HALF, uint32_t, uint2>::type;
    constexpr uint32_t P = 1;   // points per thread (the loops below are written for any P; 2 and 4 were measured slower)
    constexpr uint32_t P_TILE = P * kTile;
    uint32_t level, tile0, seg_end, tpw, level2_unused;
    if (!fwd_item(plan, level, tile0, seg_end, tpw, level2_unused)) return;
    SDFX_STAMP_BEGIN
    // padding rows of a fixed-capacity batch (sdfx_set_row_limit): samples >= row_total[0] are neither read nor written, and a
    // tile of nothing else ends here. (Stencil batches: the sample is the row within the slab; otherwise the row itself.)
    const uint32_t n_rows = plan.slabs == kGroup ? plan.slab_points : B;
    const uint32_t n_live = row_total ? min(n_rows, (uint32_t)row_total[0]) : n_rows;

    const LevelConst lc = plan.lv[level];
    const bool hashed = HASHGRID && (lc.flags & 1u);
    const bool dense = (lc.flags & 1u) == 0u && lc.res >= 2u && !(LDS && HALF && ((plan.lds_mask >> level) & 1u));   // every stride fits: rows never wrap (a one-vertex level and the LDS measurement aid keep the block gathers)
    const bool pow2 = (lc.flags & 2u) != 0u;
    const T* tab = table + (size_t)lc.row0 * C;
    // a row's address = the level's (uniform) base + a 32-bit byte offset: the load takes its base from scalar registers and ONE vector
    // register of offset (a level of 2^32 bytes or more goes to k_grid_forward: launch_forward_d3c2)
    auto at = [&](uint32_t row) -> const char* { return reinterpret_cast<const char*>(tab) + (uint32_t)(row * (uint32_t)(C * sizeof(T))); };
    // SDFX_GRID_LDS (measurement aid): the level's table in LDS, kLdsTiles tiles per workgroup
    extern __shared__ uint4 lds_tab[];   // dynamic: kLdsBytes when the plan has an LDS-resident level, nothing otherwise
    const bool in_lds = LDS && HALF && ((plan.lds_mask >> level) & 1u);   // (LDS = false: the default kernel, none of this is compiled in)
    uint32_t tile_end = min(tile0 + tpw, seg_end);   // (tpw > 1: a workgroup walks consecutive tiles of a VALU-bound coarse level)
    if (in_lds) {
        if constexpr (HALF) {
            const uint4* src = reinterpret_cast<const uint4*>(tab);
            const uint32_t n16 = (lc.size * 4u + 15u) / 16u;      // rows are padded to multiples of 8 (grid.py:131): whole blocks
            for (uint32_t i = threadIdx.x; i < n16; i += kTile) lds_tab[i] = src[i];
            __syncthreads();
        }
        tile_end = min(tile0 + kLdsTiles, seg_end);
    }
    for (uint32_t tile = tile0; tile < tile_end; tile++) {
    {
        const uint32_t first_slot = tile * P_TILE;
        const uint32_t first = plan.slabs == kGroup ? (first_slot >> 6) * kGroupsPerWave : first_slot;
        if (first >= n_live) continue;
    }

    // ---- slots of this thread -> points ----
    uint32_t pt[P], sk[P], sm[P];   // row of the batch; stencil point and base sample of that row (stencil batches)
    bool live[P];
#pragma unroll
    for (uint32_t j = 0; j < P; j++) {
        const uint32_t slot = (tile * P + j) * kTile + threadIdx.x;
        if (plan.slabs == kGroup) {   // lane = 7 * (sample within the wave) + stencil point; lane 63 idles
            const uint32_t lane = slot & 63u, g = lane / kGroup;
            const uint32_t sample = (slot >> 6) * kGroupsPerWave + g;
            live[j] = lane < kGroup * kGroupsPerWave && sample < n_live;
            sk[j] = lane - g * kGroup;
            sm[j] = sample;
            pt[j] = sk[j] * plan.slab_points + sample;
        } else {
            live[j] = slot < n_live;
            pt[j] = slot;
            sk[j] = 0; sm[j] = 0;
            if (src.xyzs) { sk[j] = stencil_slab(slot < B ? slot : 0u, src.M); sm[j] = (slot < B ? slot : 0u) - sk[j] * src.M; }
        }
        if (!live[j]) { pt[j] = 0; sk[j] = 0; sm[j] = 0; }   // a valid address to load from; nothing is stored
    }

    float xin[P][3];
#pragma unroll
    for (uint32_t j = 0; j < P; j++) {
        if (src.xyzs) {
            // stencil batch formed here (sdfx_set_stencil_source): the seven lanes of a sample read the same 12 bytes — one or two
            // 128-byte lines per wave and coordinate instead of one or two per SLAB
            const float x[3] = {src.xyzs[(size_t)sm[j] * 3], src.xyzs[(size_t)sm[j] * 3 + 1], src.xyzs[(size_t)sm[j] * 3 + 2]};
            float p[3];
            stencil_world(src, sk[j], x, p);
#pragma unroll
            for (int d = 0; d < 3; d++) xin[j][d] = (p[d] + src.bound) * src.inv;
        } else {
#pragma unroll
            for (int d = 0; d < 3; d++) xin[j][d] = inputs[(size_t)pt[j] * 3 + d];
        }
    }

    // index % hashmap_size (gridencoder.cu:78): a mask for a power-of-two size; nothing for a fully dense level (strides of all three
    // axes fit: x + y res + z res^2 < res^3 <= size); a real modulo only for a level that is hashed, or tiled with a truncated
    // stride, AND whose size is no power of two — a property of the level, decided on the scalar unit (make_level_const: flags)
    const bool need_mod = (lc.flags & 3u) == 1u;
    const uint32_t wmask = pow2 ? lc.size - 1u : 0xffffffffu;
    auto wrap = [&](uint32_t idx) -> uint32_t {
        idx &= wmask;
        if (need_mod) idx %= lc.size;
        return idx;
    };
    const bool mul24 = (lc.flags & 4u) != 0u;   // 24-bit multiplies give the rows (full rate; the 32-bit multiply is quarter rate)

    // Phase 1: cell, weights and the row indices of the 4 x-pairs of every point (no memory access)
    float ax[P][2], ay[P][2], az[P][2];
    uint32_t r0[P][4], r1[P][4];
    bool oob[P], xstep[P], xfar[P];
#pragma unroll
    for (uint32_t j = 0; j < P; j++) {
        oob[j] = xin[j][0] < 0 || xin[j][0] > 1 || xin[j][1] < 0 || xin[j][1] > 1 || xin[j][2] < 0 || xin[j][2] > 1;  // gridencoder.cu:105
        float pos[3], deriv;
        uint32_t pg[3], pn[3];
#pragma unroll
        for (int d = 0; d < 3; d++) {
            grid_locate_axis(xin[j][d], lc.res, ALIGN, INTERP, pos[d], deriv, pg[d]);
            // an out-of-range or NaN coordinate must still give an in-range vertex (its loads are issued, its result is not used)
            pg[d] = min(pg[d], lc.res - 1u);
            pn[d] = min(pg[d] + 1u, lc.res - 1u);   // the "+1" vertex (gridencoder.cu:181)
        }
        (void)deriv;
        ax[j][0] = 1 - pos[0]; ax[j][1] = pos[0];
        ay[j][0] = 1 - pos[1]; ay[j][1] = pos[1];
        az[j][0] = 1 - pos[2]; az[j][1] = pos[2];
        // rows of the four (y, z) corners' x-pairs (gridencoder.cu:45-79). The level's kind is a property of the workgroup: two
        // straight-line versions behind ONE uniform branch (as selects on a uniform condition every corner paid both the xor and
        // the add, plus the select: 16 of the ~270 vector instructions of a tile)
        {
            const uint32_t fy = hashed ? 2654435761u : lc.m1, fz = hashed ? 805459861u : lc.m2;
            uint32_t ty[2], tz[2];
            if (mul24) {
                ty[0] = __umul24(pg[1], fy & 0xFFFFFFu); ty[1] = __umul24(pn[1], fy & 0xFFFFFFu);
                tz[0] = __umul24(pg[2], fz & 0xFFFFFFu); tz[1] = __umul24(pn[2], fz & 0xFFFFFFu);
            } else {
                ty[0] = pg[1] * fy; ty[1] = pn[1] * fy;
                tz[0] = pg[2] * fz; tz[1] = pn[2] * fz;
            }
            if (hashed) {
#pragma unroll
                for (int k = 0; k < 4; k++) {
                    const uint32_t yz = ty[k & 1] ^ tz[k >> 1];
                    r0[j][k] = wrap(pg[0] ^ yz);
                    r1[j][k] = wrap(pn[0] ^ yz);
                }
            } else if (dense) {   // x + y res + z res^2 < res^3 <= size (make_level_const): nothing to wrap, and r1 = r0 + 1 unless
                                  // x is the grid's last vertex (x + 1 clamped: r1 = r0). r0 here = the FIRST row of the two-row load
                                  // of Phase 2: the pair's own r0, or at the last vertex the row before it — never past the table
                const uint32_t xb = pg[0] - (pn[0] == pg[0] ? 1u : 0u);
#pragma unroll
                for (int k = 0; k < 4; k++) {
                    r0[j][k] = xb + (ty[k & 1] + tz[k >> 1]);
                    r1[j][k] = r0[j][k] + 1u;
                }
            } else {   // a tiled grid whose strides do not fit: the dense index wrapped (gridencoder.cu:61-79 without the hash)
#pragma unroll
                for (int k = 0; k < 4; k++) {
                    const uint32_t yz = ty[k & 1] + tz[k >> 1];
                    r0[j][k] = wrap(pg[0] + yz);
                    r1[j][k] = wrap(pn[0] + yz);
                }
            }
        }
        xstep[j] = pn[0] != pg[0];                                   // (false only at the grid's last vertex)
        xfar[j] = ((pg[0] ^ pn[0]) & wmask) >= RB;                   // hashed power-of-two levels: r0 ^ r1 = (x ^ (x + 1)) & mask for all four pairs
    }

    // Phase 2: every gather of the thread is issued before the first result is touched (4 P in flight per lane, plus the
    // second gathers of x-pairs that straddle two 16-byte blocks)
    RowT v0[P][4], v1[P][4];
    if (dense) {
        // a dense level is x-major: the pair's rows are consecutive — ONE 2-row load at the row's own alignment (gfx950 takes a
        // dwordx2 at 4-byte alignment), nothing to pick from a block and never a second gather. At the grid's last vertex both
        // corners are the load's SECOND row (see Phase 1)
        PairT<HALF> pr[P][4];
#pragma unroll
        for (uint32_t j = 0; j < P; j++) {
#pragma unroll
            for (int k = 0; k < 4; k++) pr[j][k] = *reinterpret_cast<const PairT<HALF>*>(at(r0[j][k]));
        }
#pragma unroll
        for (uint32_t j = 0; j < P; j++) {
#pragma unroll
            for (int k = 0; k < 4; k++) {
                v0[j][k] = xstep[j] ? pr[j][k].a : pr[j][k].b;
                v1[j][k] = pr[j][k].b;
            }
        }
    } else if ((plan.vec16 >> level) & 1u) {
        // rows r0 and r1 = row(x + 1) nearly always share an aligned 16-byte block (the hash's x prime is 1, dense
        // levels are x-major): one gather serves both corners of the pair
        // Does a pair straddle two blocks? At a hashed level of power-of-two size r0 ^ r1 = (x ^ (x + 1)) & mask for all four pairs of
        // a point: ONE test and one masked branch around its four second gathers (BY_POINT); otherwise pair by pair. Two copies of
        // the block behind a uniform branch, so that neither pays for the other's test.
        auto gather16 = [&](auto by_point) {
            constexpr bool BY_POINT = decltype(by_point)::value;
            auto pair_far = [&](uint32_t j, int k) -> bool { return BY_POINT ? xfar[j] : ((r0[j][k] ^ r1[j][k]) >= RB); };
            uint4 blk[P][4];
#pragma unroll
            for (uint32_t j = 0; j < P; j++) {
#pragma unroll
                for (int k = 0; k < 4; k++) {
                    if (in_lds) blk[j][k] = lds_tab[r0[j][k] >> 2];
                    else blk[j][k] = *reinterpret_cast<const uint4*>(at(r0[j][k] & ~(RB - 1)));
                }
            }
            RowT extra[P][4];
            auto second = [&](uint32_t j, int k) {
                if constexpr (HALF) {
                    if (in_lds) extra[j][k] = reinterpret_cast<const uint32_t*>(lds_tab)[r1[j][k]];
                    else extra[j][k] = *reinterpret_cast<const RowT*>(at(r1[j][k]));
                } else {
                    extra[j][k] = *reinterpret_cast<const RowT*>(at(r1[j][k]));
                }
            };
#pragma unroll
            for (uint32_t j = 0; j < P; j++) {
                if constexpr (BY_POINT) {
                    if (xfar[j]) {
#pragma unroll
                        for (int k = 0; k < 4; k++) second(j, k);
                    }
                } else {
#pragma unroll
                    for (int k = 0; k < 4; k++)
                        if (pair_far(j, k)) second(j, k);
                }
            }
#pragma unroll
            for (uint32_t j = 0; j < P; j++) {
#pragma unroll
                for (int k = 0; k < 4; k++) {
                    if constexpr (HALF) {
                        v0[j][k] = pick4(blk[j][k], r0[j][k] & 3u);
                        v1[j][k] = pick4(blk[j][k], r1[j][k] & 3u);
                    } else {
                        v0[j][k] = (r0[j][k] & 1u) ? make_uint2(blk[j][k].z, blk[j][k].w) : make_uint2(blk[j][k].x, blk[j][k].y);
                        v1[j][k] = (r1[j][k] & 1u) ? make_uint2(blk[j][k].z, blk[j][k].w) : make_uint2(blk[j][k].x, blk[j][k].y);
                    }
                    if (pair_far(j, k)) v1[j][k] = extra[j][k];
                }
            }
        };
        if (hashed && !need_mod) gather16(std::true_type{});
        else gather16(std::false_type{});
    } else {
#pragma unroll
        for (uint32_t j = 0; j < P; j++) {
#pragma unroll
            for (int k = 0; k < 4; k++) {
                v0[j][k] = *reinterpret_cast<const RowT*>(at(r0[j][k]));
                v1[j][k] = *reinterpret_cast<const RowT*>(at(r1[j][k]));
            }
        }
    }

    // Phase 3: accumulate in the reference's corner order and store
#pragma unroll
    for (uint32_t j = 0; j < P; j++) {
        Acc2<HALF> acc;
#pragma unroll
        for (int k = 0; k < 4; k++) {   // corner order of gridencoder.cu:168-195 (x fastest); weights ((1 * a_x) * a_y) * a_z
            const float wy = ay[j][k & 1], wz = az[j][k >> 1];
            acc.add(((1 * ax[j][0]) * wy) * wz, v0[j][k]);
            acc.add(((1 * ax[j][1]) * wy) * wz, v1[j][k]);
        }
        if (live[j]) {
            T* out = out_layout == 0 ? outputs + ((size_t)level * B + pt[j]) * C : outputs + ((size_t)pt[j] * L + level) * C;
            acc.store(out, oob[j]);
        }
    }
    }   // tiles of this workgroup
    SDFX_STAMP_END(1u, level, tile0)
}

// ---- two levels per wave (round 6) ---------------------------------------------------------------------------------------
// What the per-level counters say (profiles/r06_encode_levels_pmc.txt): a wave of a fine hashed level spends 0.6 of its cycles parked
// on its gathers while its SIMD issues vector instructions a third of the time, a wave of a coarse level the other way round — and
// with one level per XCD at a time the two never share a CU. Mixing them workgroup by workgroup does not help (measured, see the
// head of the file: the fine workgroups end up holding the wave slots). Here the mix is inside every wave: the fine level's rows are
// formed and its gathers issued, THEN the coarse level's rows are formed and its loads issued (vector memory returns in order:
// the fine level's data is complete when `vmcnt` has dropped to the coarse level's count), then the two are reduced and stored in that
// order. Same arithmetic per level as k_grid_fwd, bit for bit; the coordinates of the tile are formed once.
struct LevelKind {   // wave-uniform properties of a level (scalar registers)
    LevelConst lc;
    const char* tab;
    uint32_t wmask;
    bool hashed, dense, need_mod, mul24, vec16;
};
template <bool HASHGRID>
__device__ __forceinline__ LevelKind level_kind(const FwdPlan& plan, const __half* table, uint32_t level) {
    LevelKind k;
    k.lc = plan.lv[level];
    k.hashed = HASHGRID && (k.lc.flags & 1u);
    k.dense = (k.lc.flags & 1u) == 0u && k.lc.res >= 2u;
    k.need_mod = (k.lc.flags & 3u) == 1u;
    k.mul24 = (k.lc.flags & 4u) != 0u;
    k.vec16 = ((plan.vec16 >> level) & 1u) != 0u;
    k.wmask = (k.lc.flags & 2u) ? k.lc.size - 1u : 0xffffffffu;
    k.tab = reinterpret_cast<const char*>(table + (size_t)k.lc.row0 * 2);
    return k;
}
// A level's KIND decides the shape of its loads, and a value loaded in one arm of a branch and used after the join is copied at the
// join — which waits for it. So the kinds of the two levels are template parameters of the tile loop (one uniform dispatch per
// workgroup): between the issue of a level's loads and their use there is no join that merges them.
enum : uint32_t {
    kKindDense = 0,    // every stride fits: x-major rows that never wrap — one two-row load per x-pair
    kKindHash16 = 1,   // hashed, power-of-two size, 16-byte aligned table: one 16-byte block per pair, ONE straddle test per point
    kKindOther = 2,    // anything else (unaligned tables, sizes that need a modulo, tiled grids with truncated strides, one-vertex levels):
                       // k_grid_fwd's general code, values picked as soon as loaded
    kKindNone = 3,     // no second level
};
__device__ __forceinline__ uint32_t kind_of(const LevelKind& k) {
    if (k.dense) return kKindDense;
    if (k.hashed && !k.need_mod && k.vec16) return kKindHash16;
    return kKindOther;
}

// cell, weights and the y / z terms of the rows (k_grid_fwd's Phase 1)
struct Located { uint32_t pg[3], pn[3], ty[2], tz[2]; };
template <uint32_t INTERP, bool ALIGN>
__device__ __forceinline__ void locate(const LevelKind& k, const float xin[3], float pos[3], Located& o) {
    const LevelConst& lc = k.lc;
    float deriv;
#pragma unroll
    for (int d = 0; d < 3; d++) {
        grid_locate_axis(xin[d], lc.res, ALIGN, INTERP, pos[d], deriv, o.pg[d]);
        o.pg[d] = min(o.pg[d], lc.res - 1u);
        o.pn[d] = min(o.pg[d] + 1u, lc.res - 1u);
    }
    (void)deriv;
    const uint32_t fy = k.hashed ? 2654435761u : lc.m1, fz = k.hashed ? 805459861u : lc.m2;
    if (k.mul24) {
        o.ty[0] = __umul24(o.pg[1], fy & 0xFFFFFFu); o.ty[1] = __umul24(o.pn[1], fy & 0xFFFFFFu);
        o.tz[0] = __umul24(o.pg[2], fz & 0xFFFFFFu); o.tz[1] = __umul24(o.pn[2], fz & 0xFFFFFFu);
    } else {
        o.ty[0] = o.pg[1] * fy; o.ty[1] = o.pn[1] * fy;
        o.tz[0] = o.pg[2] * fz; o.tz[1] = o.pn[2] * fz;
    }
}
__device__ __forceinline__ uint32_t reduce8(const float pos[3], const uint32_t v0[4], const uint32_t v1[4]) {
    const float ax[2] = {1 - pos[0], pos[0]}, ay[2] = {1 - pos[1], pos[1]}, az[2] = {1 - pos[2], pos[2]};
    Acc2<true> acc;
#pragma unroll
    for (int c = 0; c < 4; c++) {   // corner order of gridencoder.cu:168-195 (x fastest); weights ((1 * a_x) * a_y) * a_z
        const float wy = ay[c & 1], wz = az[c >> 1];
        acc.add(((1 * ax[0]) * wy) * wz, v0[c]);
        acc.add(((1 * ax[1]) * wy) * wz, v1[c]);
    }
    return __builtin_bit_cast(uint32_t, acc.acc);
}

template <uint32_t KIND> struct PointLevel;

template <> struct PointLevel<kKindNone> {
    template <uint32_t INTERP, bool ALIGN> __device__ __forceinline__ void start(const LevelKind&, const float*) {}
    __device__ __forceinline__ uint32_t reduce(const LevelKind&) const { return 0u; }
};

template <> struct PointLevel<kKindDense> {   // 12 registers while the loads are in flight
    PairT<true> pr[4];
    float pos[3];
    bool xstep;
    template <uint32_t INTERP, bool ALIGN> __device__ __forceinline__ void start(const LevelKind& k, const float xin[3]) {
        Located o;
        locate<INTERP, ALIGN>(k, xin, pos, o);
        // r1 = r0 + 1 unless x is the grid's last vertex (x + 1 clamped: r1 = r0): the load then starts one row earlier and both corners
        // are its SECOND row — never past the table
        const uint32_t xb = o.pg[0] - (o.pn[0] == o.pg[0] ? 1u : 0u);
#pragma unroll
        for (int c = 0; c < 4; c++)
            pr[c] = *reinterpret_cast<const PairT<true>*>(k.tab + (uint32_t)((xb + (o.ty[c & 1] + o.tz[c >> 1])) * 4u));
        xstep = o.pn[0] != o.pg[0];
    }
    __device__ __forceinline__ uint32_t reduce(const LevelKind&) const {
        uint32_t v0[4], v1[4];
#pragma unroll
        for (int c = 0; c < 4; c++) { v0[c] = xstep ? pr[c].a : pr[c].b; v1[c] = pr[c].b; }
        return reduce8(pos, v0, v1);
    }
};

template <> struct PointLevel<kKindHash16> {   // 25 registers while the loads are in flight
    uint4 blk[4];
    uint32_t extra[4];
    float pos[3];
    uint32_t sel;   // bits 2c, 2c+1 = r0[c] & 3; 8+2c, 9+2c = r1[c] & 3; bit 16 = the pairs straddle two blocks
    template <uint32_t INTERP, bool ALIGN> __device__ __forceinline__ void start(const LevelKind& k, const float xin[3]) {
        Located o;
        locate<INTERP, ALIGN>(k, xin, pos, o);
        uint32_t r0[4], r1[4];
#pragma unroll
        for (int c = 0; c < 4; c++) {
            const uint32_t yz = o.ty[c & 1] ^ o.tz[c >> 1];
            r0[c] = (o.pg[0] ^ yz) & k.wmask;
            r1[c] = (o.pn[0] ^ yz) & k.wmask;
        }
#pragma unroll
        for (int c = 0; c < 4; c++) blk[c] = *reinterpret_cast<const uint4*>(k.tab + (uint32_t)((r0[c] & ~3u) * 4u));
        // r0 ^ r1 = (x ^ (x + 1)) & mask for all four pairs of the point: one test, one masked branch around the four second gathers
        const bool far = ((o.pg[0] ^ o.pn[0]) & k.wmask) >= 4u;
        if (far) {
#pragma unroll
            for (int c = 0; c < 4; c++) extra[c] = *reinterpret_cast<const uint32_t*>(k.tab + (uint32_t)(r1[c] * 4u));
        }
        sel = far ? (1u << 16) : 0u;
#pragma unroll
        for (int c = 0; c < 4; c++) sel |= ((r0[c] & 3u) << (2 * c)) | ((r1[c] & 3u) << (8 + 2 * c));
    }
    __device__ __forceinline__ uint32_t reduce(const LevelKind&) const {
        uint32_t v0[4], v1[4];
        const bool far = (sel >> 16) & 1u;
#pragma unroll
        for (int c = 0; c < 4; c++) {
            // (the block as four opaque VALUES: left as members of *this, the compiler folds pick4's selects into ONE load from a
            // computed address — which pins the whole struct in scratch memory)
            uint4 b = blk[c];
            asm("" : "+v"(b.x), "+v"(b.y), "+v"(b.z), "+v"(b.w));
            v0[c] = pick4(b, (sel >> (2 * c)) & 3u);
            v1[c] = pick4(b, (sel >> (8 + 2 * c)) & 3u);
            if (far) v1[c] = extra[c];
        }
        return reduce8(pos, v0, v1);
    }
};

template <> struct PointLevel<kKindOther> {   // the general code: the eight rows as soon as they are loaded (11 registers afterwards)
    uint32_t v0[4], v1[4];
    float pos[3];
    template <uint32_t INTERP, bool ALIGN> __device__ __forceinline__ void start(const LevelKind& k, const float xin[3]) {
        const LevelConst& lc = k.lc;
        Located o;
        locate<INTERP, ALIGN>(k, xin, pos, o);
        auto wrap = [&](uint32_t idx) -> uint32_t {
            idx &= k.wmask;
            if (k.need_mod) idx %= lc.size;
            return idx;
        };
        auto at = [&](uint32_t row) -> const char* { return k.tab + (uint32_t)(row * 4u); };
        if (k.dense) {
            const uint32_t xb = o.pg[0] - (o.pn[0] == o.pg[0] ? 1u : 0u);
            PairT<true> pr[4];
#pragma unroll
            for (int c = 0; c < 4; c++) pr[c] = *reinterpret_cast<const PairT<true>*>(at(xb + (o.ty[c & 1] + o.tz[c >> 1])));
#pragma unroll
            for (int c = 0; c < 4; c++) { v0[c] = o.pn[0] != o.pg[0] ? pr[c].a : pr[c].b; v1[c] = pr[c].b; }
            return;
        }
        uint32_t r0[4], r1[4];
#pragma unroll
        for (int c = 0; c < 4; c++) {
            const uint32_t yz = k.hashed ? (o.ty[c & 1] ^ o.tz[c >> 1]) : (o.ty[c & 1] + o.tz[c >> 1]);
            r0[c] = wrap(k.hashed ? (o.pg[0] ^ yz) : (o.pg[0] + yz));
            r1[c] = wrap(k.hashed ? (o.pn[0] ^ yz) : (o.pn[0] + yz));
        }
        if (k.vec16) {
            uint4 blk[4];
            uint32_t extra[4];
#pragma unroll
            for (int c = 0; c < 4; c++) blk[c] = *reinterpret_cast<const uint4*>(at(r0[c] & ~3u));
#pragma unroll
            for (int c = 0; c < 4; c++) {
                extra[c] = 0u;
                if ((r0[c] ^ r1[c]) >= 4u) extra[c] = *reinterpret_cast<const uint32_t*>(at(r1[c]));
            }
#pragma unroll
            for (int c = 0; c < 4; c++) {
                v0[c] = pick4(blk[c], r0[c] & 3u);
                v1[c] = (r0[c] ^ r1[c]) >= 4u ? extra[c] : pick4(blk[c], r1[c] & 3u);
            }
        } else {
#pragma unroll
            for (int c = 0; c < 4; c++) {
                v0[c] = *reinterpret_cast<const uint32_t*>(at(r0[c]));
                v1[c] = *reinterpret_cast<const uint32_t*>(at(r1[c]));
            }
        }
    }
    __device__ __forceinline__ uint32_t reduce(const LevelKind&) const { return reduce8(pos, v0, v1); }
};

// the tiles of one workgroup at the levels of kinds KA and KB
template <uint32_t INTERP, bool ALIGN, uint32_t KA, uint32_t KB>
__device__ __forceinline__ void pair_tiles(const float* __restrict__ inputs, __half* __restrict__ outputs, uint32_t B, uint32_t L,
                                           const FwdPlan& plan, int out_layout, uint32_t n_live, const StencilSrc& src, const LevelKind& ka,
                                           const LevelKind& kb, uint32_t level, uint32_t level2, uint32_t tile0, uint32_t tile_end) {
    for (uint32_t tile = tile0; tile < tile_end; tile++) {
        // ---- slot of this thread -> point (k_grid_fwd's map) ----
        const uint32_t slot = tile * kTile + threadIdx.x;
        {
            const uint32_t first_slot = tile * kTile;
            const uint32_t first = plan.slabs == kGroup ? (first_slot >> 6) * kGroupsPerWave : first_slot;
            if (first >= n_live) continue;
        }
        uint32_t pt, sk = 0, sm = 0;
        bool live;
        if (plan.slabs == kGroup) {
            const uint32_t lane = slot & 63u, g = lane / kGroup;
            const uint32_t sample = (slot >> 6) * kGroupsPerWave + g;
            live = lane < kGroup * kGroupsPerWave && sample < n_live;
            sk = lane - g * kGroup;
            sm = sample;
            pt = sk * plan.slab_points + sample;
        } else {
            live = slot < n_live;
            pt = slot;
            if (src.xyzs) { sk = stencil_slab(slot < B ? slot : 0u, src.M); sm = (slot < B ? slot : 0u) - sk * src.M; }
        }
        if (!live) { pt = 0; sk = 0; sm = 0; }
        float xin[3];
        if (src.xyzs) {
            const float x[3] = {src.xyzs[(size_t)sm * 3], src.xyzs[(size_t)sm * 3 + 1], src.xyzs[(size_t)sm * 3 + 2]};
            float p[3];
            stencil_world(src, sk, x, p);
#pragma unroll
            for (int d = 0; d < 3; d++) xin[d] = (p[d] + src.bound) * src.inv;
        } else {
#pragma unroll
            for (int d = 0; d < 3; d++) xin[d] = inputs[(size_t)pt * 3 + d];
        }
        const bool oob = xin[0] < 0 || xin[0] > 1 || xin[1] < 0 || xin[1] > 1 || xin[2] < 0 || xin[2] > 1;  // gridencoder.cu:105
        auto out_of = [&](uint32_t lv) -> __half* {
            return out_layout == 0 ? outputs + ((size_t)lv * B + pt) * 2 : outputs + ((size_t)pt * L + lv) * 2;
        };

        // Both levels' loads in flight together when the second level is a dense one (12 registers of loads: 63 in all, 8 waves per
        // SIMD); two hashed levels would need 87 registers that way, so those are evaluated one after the other (they are the middle
        // of the level sequence, where neither level waits much for its gathers) and share the tile's coordinates only.
#ifndef SDFX_PAIR_OVERLAP_HASH
#define SDFX_PAIR_OVERLAP_HASH 0   // measurement aid: 1 = two hashed levels in flight together as well (build with -DSDFX_PAIR_WAVES=5:
                                   // 388 us against 386 us for this kernel on the same box, profiles/r06_encode_pair_plan.txt)
#endif
        constexpr bool kOverlap = KB == kKindDense || (SDFX_PAIR_OVERLAP_HASH && KB == kKindHash16);
        PointLevel<KA> qa;
        PointLevel<KB> qb;
        qa.template start<INTERP, ALIGN>(ka, xin);
        __builtin_amdgcn_sched_barrier(0);
        if constexpr (kOverlap) {
            qb.template start<INTERP, ALIGN>(kb, xin);
            __builtin_amdgcn_sched_barrier(0);
        }
        {
            const uint32_t r = qa.reduce(ka);
            if (live) *reinterpret_cast<uint32_t*>(out_of(level)) = oob ? 0u : r;
        }
        if constexpr (KB != kKindNone) {
            if constexpr (!kOverlap) {
                __builtin_amdgcn_sched_barrier(0);
                qb.template start<INTERP, ALIGN>(kb, xin);
            }
            const uint32_t r = qb.reduce(kb);
            if (live) *reinterpret_cast<uint32_t*>(out_of(level2)) = oob ? 0u : r;
        }
    }
}

#ifndef SDFX_PAIR_WAVES
#define SDFX_PAIR_WAVES 8   // waves per SIMD the register allocation aims at (measurement aid: -DSDFX_PAIR_WAVES=8 / 6 / 5)
#endif
template <uint32_t INTERP, bool ALIGN, bool HASHGRID>
__global__ __launch_bounds__(kTile) __attribute__((amdgpu_waves_per_eu(SDFX_PAIR_WAVES, SDFX_PAIR_WAVES)))
void k_grid_fwd_pair(const float* __restrict__ inputs, const __half* __restrict__ table, __half* __restrict__ outputs, uint32_t B,
                     uint32_t L, FwdPlan plan, int out_layout, const int32_t* __restrict__ row_total, StencilSrc src) {
    uint32_t level, tile0, seg_end, tpw, level2;
    if (!fwd_item(plan, level, tile0, seg_end, tpw, level2)) return;
    SDFX_STAMP_BEGIN
    const uint32_t n_rows = plan.slabs == kGroup ? plan.slab_points : B;
    const uint32_t n_live = row_total ? min(n_rows, (uint32_t)row_total[0]) : n_rows;
    const bool two = level2 != kNoLevel;   // (uniform; an odd level count leaves one level alone)
    const LevelKind ka = level_kind<HASHGRID>(plan, table, level);
    const LevelKind kb = level_kind<HASHGRID>(plan, table, two ? level2 : level);
    const uint32_t tile_end = min(tile0 + tpw, seg_end);
    const uint32_t kind_a = kind_of(ka) == kKindHash16 ? kKindHash16 : kKindOther, kind_b = two ? kind_of(kb) : kKindNone;
#define SDFX_PAIR_CASE(KA_, KB_)                                                                                                     \
    if (kind_a == KA_ && kind_b == KB_)                                                                                              \
        pair_tiles<INTERP, ALIGN, KA_, KB_>(inputs, outputs, B, L, plan, out_layout, n_live, src, ka, kb, level, level2, tile0, tile_end);
#ifndef SDFX_PAIR_COMBOS
#define SDFX_PAIR_COMBOS 0xFF
#endif
    if (false) {}
#if SDFX_PAIR_COMBOS & 1
    else SDFX_PAIR_CASE(kKindHash16, kKindDense)
#endif
#if SDFX_PAIR_COMBOS & 2
    else SDFX_PAIR_CASE(kKindHash16, kKindHash16)
#endif
#if SDFX_PAIR_COMBOS & 4
    else SDFX_PAIR_CASE(kKindHash16, kKindOther)
#endif
#if SDFX_PAIR_COMBOS & 8
    else SDFX_PAIR_CASE(kKindHash16, kKindNone)
#endif
#if SDFX_PAIR_COMBOS & 16
    else SDFX_PAIR_CASE(kKindOther, kKindDense)
#endif
#if SDFX_PAIR_COMBOS & 32
    else SDFX_PAIR_CASE(kKindOther, kKindHash16)
#endif
#if SDFX_PAIR_COMBOS & 64
    else SDFX_PAIR_CASE(kKindOther, kKindOther)
#endif
#if SDFX_PAIR_COMBOS & 128
    else SDFX_PAIR_CASE(kKindOther, kKindNone)
#endif
#undef SDFX_PAIR_CASE
    SDFX_STAMP_END(1u, level, tile0)
}

// ---- host: the plan -------------------------------------------------------------------------------------------------

// Distinct 128-byte table lines per wave (64 lanes: 9 samples x 7 stencil points, or 64 consecutive samples) as a
// function of u = grid cells per sample step, for ray-ordered samples: measured by simulating the kernel's addresses on
// the samples of a 4096-ray view (the finite-difference stencil of network_grid.py:81, e = 0.01, step 1/591 of the unit
// cube; profiles/README.md). Piecewise linear in log u. Includes the second gather of an x-pair that straddles two
// 16-byte blocks (a quarter of the lanes).
double lines_per_wave(double u, bool stencil) {
    static const double us[16] = {0.0271, 0.0389, 0.0525, 0.0728, 0.0998, 0.137, 0.190, 0.261, 0.360, 0.499, 0.689, 0.951, 1.315, 1.816, 2.509, 3.465};
    static const double st7[16] = {7.6, 9.7, 11.8, 14.6, 18.9, 25.9, 36.3, 52.8, 79.9, 106.6, 136.7, 181.3, 212.8, 242.1, 252.8, 267.4};
    static const double ray[16] = {14.3, 19.0, 23.3, 30.1, 38.4, 50.0, 65.9, 87.3, 116.5, 155.3, 205.3, 269.2, 320.1, 319.6, 319.6, 319.3};
    const double* c = stencil ? st7 : ray;
    if (u <= us[0]) return c[0];
    if (u >= us[15]) return c[15];
    const double lu = log(u);
    for (int i = 0; i < 15; i++) {
        if (u <= us[i + 1]) {
            const double t = (lu - log(us[i])) / (log(us[i + 1]) - log(us[i]));
            return c[i] + t * (c[i + 1] - c[i]);
        }
    }
    return c[15];
}

// The measured correction below was taken on ONE configuration: the -O grid (16 levels from 16 to 2048: S = log2(1.3819), so the
// levels' u = res x step are exactly the table's knots) at the iteration's step (1 / 591 of the unit cube). It is applied to that
// configuration only (within 20 % of the step); every other grid or step is priced by the model max(lines per wave, VALU floor),
// which needs no measurement. The plan's price list is readable through sdfx_grid_forward_level_costs (tests/test_grid_plan.py
// checks the balance property against it, not against a copy of these numbers).
bool measured_stencil_config(uint32_t levels, float S, uint32_t H, float step) {
    const double s_ref = log2(2048.0 / 16.0) / 15.0, step_ref = 1.0 / 591.0;
    return levels == 16 && H == 16 && fabs((double)S - s_ref) < 1e-4 && step > 0.8 * step_ref && step < 1.25 * step_ref;
}

// What a tile of a level COSTS in the launch, in the same units, for stencil batches: the per-XCD timeline of one launch at
// B = 3.26 M (tools/xcd_timeline.py, round 5) corrected the model "max(lines, VALU floor = 97)" level by level — an XCD's finish time
// over the mean, applied to the levels it walked: the levels around one cell per step (u = 0.95-1.8) are 6-13 % cheaper than
// their line count says, the levels of u = 0.2-0.7 up to 11 % dearer than the VALU floor. With this table the eight XCDs finish
// within 5 % of each other (24 % before) and the launch is 4-6 % shorter (profiles/r05_encode_split_by_timeline_costs.txt).
// Second pass after the kernel's dense levels got their two-row loads and the hashed ones lost a few instructions
// (profiles/r05_encode_dense_levels.txt): a dense level's tile costs 0.71 of the table's entry, the hashed entries below u = 1 moved
// by 1-7 %.
double stencil_tile_cost(double u, bool dense) {
    static const double us[16] = {0.0271, 0.0389, 0.0525, 0.0728, 0.0998, 0.137, 0.190, 0.261, 0.360, 0.499, 0.689, 0.951, 1.315, 1.816, 2.509, 3.465};
    static const double cost[16] = {98, 98, 98, 98, 98, 91, 101, 109, 108, 114, 141, 151, 180, 233, 265, 281};
    const double k = dense ? 0.71 : 1.0;
    if (u <= us[0]) return k * cost[0];
    if (u >= us[15]) return k * cost[15];
    const double lu = log(u);
    for (int i = 0; i < 15; i++)
        if (u <= us[i + 1]) return k * (cost[i] + (lu - log(us[i])) / (log(us[i + 1]) - log(us[i])) * (cost[i + 1] - cost[i]));
    return k * cost[15];
}

// What a tile of a PAIR costs on the configuration of measured_stencil_config, by the fine level's u = res x step (knots = levels 8-15 of
// the -O grid, whose partners are levels 7-0): mean workgroup time of the pair in one launch at B = 3.26 M (tools/pair_ab.py,
// profiles/r06_encode_pair_plan.txt), in the units of the per-level table. Against the sum of the two levels' prices the pairs of two
// hashed levels (8+7 ... 11+4) come out 3-12 % dearer, the fine + dense pairs within 1.5 %: with the sum the XCDs finished 13-15 % of the
// span apart, with this table 3 %.
double stencil_pair_cost(double u_fine) {
    static const double us[8] = {0.360, 0.499, 0.689, 0.951, 1.315, 1.816, 2.509, 3.465};
    static const double cost[8] = {243, 229, 239, 236, 261, 300, 330, 350};
    if (u_fine <= us[0]) return cost[0];
    if (u_fine >= us[7]) return cost[7];
    const double lu = log(u_fine);
    for (int i = 0; i < 7; i++)
        if (u_fine <= us[i + 1]) return cost[i] + (lu - log(us[i])) / (log(us[i + 1]) - log(us[i])) * (cost[i + 1] - cost[i]);
    return cost[7];
}

struct Unit { uint32_t level; double cost; uint32_t level2; };   // cost per tile (pair plans: of both levels)

// what the plan prices a tile of level l at (the measured table on its configuration, the model elsewhere, 1 without a usable hint)
double level_tile_cost(const LevelConst& c, uint32_t levels, float S, uint32_t H, uint32_t slabs, float step, bool balance, double valu_lines) {
    if (!(balance && step > 0.f)) return 1.0;
    const double lines = lines_per_wave((double)c.res * fabs((double)step), slabs == kGroup);
    const bool table = slabs == kGroup && measured_stencil_config(levels, S, H, step) && dev_switch("SDFX_GRID_COST_TABLE", 1);
    return table ? stencil_tile_cost((double)c.res * step, (c.flags & 1u) == 0u && c.res >= 2u) : (lines > valu_lines ? lines : valu_lines);
}

// ... and a tile of the pair (hi, lo) of a pair plan. The measured table on its configuration; elsewhere a model of what the table shows:
// the coarse partner's work hides behind the fine level's gathers, so a pair costs its fine level's lines (x 1.3: the L1 misses of a fine
// hashed level, the correction the per-level table also carries) or the vector work of two levels (2.42 VALU floors = 235 line units),
// whichever is longer. Against the measured pairs of the -O grid this is within 6 % (235, 235, 235, 236, 277, 315, 329, 348 for
// 243, 229, 239, 236, 261, 300, 330, 350); the plain sum of the two levels' prices left the XCDs 31 % of the span apart (454 us where
// the table gives 371 and one level per workgroup 424: profiles/r06_encode_pair_plan.txt).
double pair_tile_cost(const LevelConst& hi, const LevelConst& lo, uint32_t levels, float S, uint32_t H, uint32_t slabs, float step, bool balance,
                      double valu_lines) {
    if (!(balance && step > 0.f)) return 2.0;
    if (slabs == kGroup && measured_stencil_config(levels, S, H, step) && dev_switch("SDFX_GRID_COST_TABLE", 1))
        return stencil_pair_cost((double)hi.res * step);
    const double lh = lines_per_wave((double)hi.res * fabs((double)step), slabs == kGroup);
    const double ll = lines_per_wave((double)lo.res * fabs((double)step), slabs == kGroup);
    const double gathers = 1.3 * (lh > ll ? lh : ll), valu = 2.42 * valu_lines;
    return gathers > valu ? gathers : valu;
}

bool pair_plan_enabled(uint32_t elem_bytes) { return elem_bytes == 2 && dev_switch("SDFX_GRID_PAIR", 1) == 1; }

FwdPlan make_fwd_plan(const int32_t* offsets_host, uint32_t levels, float S, uint32_t H, uint32_t elem_bytes, uint32_t B,
                      uint32_t slabs, float step, bool balance, double valu_lines) {
    constexpr uint32_t P = 1;
    FwdPlan p;
    memset(&p, 0, sizeof(p));
    double lines[kMaxLevels];
    for (uint32_t l = 0; l < levels; l++) {
        p.lv[l] = make_level_const(offsets_host, l, S, H);
        const LevelConst& c = p.lv[l];
        lines[l] = lines_per_wave((double)c.res * fabs((double)step), slabs == kGroup);
    }
    p.slabs = slabs == kGroup && B % kGroup == 0 ? kGroup : 1u;
    p.slab_points = B / p.slabs;
    const uint64_t slots = p.slabs == kGroup ? (uint64_t)div_up(p.slab_points, kGroupsPerWave) * 64u : B;
    const uint32_t T = div_up(slots, (uint64_t)kTile * P);   // tiles per level
    p.tiles = T; p.levels = levels;
    {
        // SDFX_GRID_LDS = largest level table (bytes) to keep in LDS: 16384 = level 0 of the -O grid, 49152 = levels 0 and 1
        const uint32_t lds = [] { const int v = dev_switch("SDFX_GRID_LDS", 0); return (uint32_t)(v < 0 ? 0 : (v > 65536 ? 65536 : v)); }();
        if (lds && elem_bytes == 2)
            for (uint32_t l = 0; l < levels; l++)
                if ((uint64_t)p.lv[l].size * 4u <= lds) { p.lds_mask |= 1u << l; if (p.lv[l].size * 4u > p.lds_bytes) p.lds_bytes = (p.lv[l].size * 4u + 15u) & ~15u; }
    }
    {
        const char* e = dev_string("SDFX_GRID_PLAN");   // devtools build only
        p.sample_major = (e && !strcmp(e, "sample_major")) ? 1u : 0u;
    }

    // ---- the sequence of levels and what a tile of each costs ----
    Unit units[kMaxLevels];
    uint32_t nu = 0;
    // step < 0 — points in space-filling-curve order (64 consecutive points = a 4 x 4 x 4 block of a regular grid; the occupancy refresh)
    // — is NOT priced: a line-count model of such blocks (rows shared in y and z: all but the three finest levels at the VALU floor)
    // was built and its balanced split measured 628 us against 404 us for the even fine / coarse pairing below (and 474 us for
    // k_grid_forward; profiles/r06_refresh_encode_morton.txt): the fine hashed levels cost what their L1 misses cost, not what their
    // distinct lines suggest. The pairing [L-1, 0, L-2, 1, ...] gives every XCD one fine and one coarse level, which is about even.
    p.pair = (pair_plan_enabled(elem_bytes) && !p.sample_major && !p.lds_mask) ? 1u : 0u;
    if (p.pair) {
        // units = (fine, coarse) pairs from both ends of the level sequence, (L-1, 0), (L-2, 1), ..., an odd count's middle level alone;
        // a pair's tile is priced at the sum of its levels' prices (what the overlap inside the wave takes off is about the same
        // share for every pair of the -O grid: profiles/r06_encode_pair_timeline.txt)
        // SDFX_GRID_PAIR_MIDDLE = 0 (measurement aid): only a DENSE level is paired; the hashed levels left over in the middle of the
        // sequence go one level per workgroup (their two 2 MiB tables would fill an XCD's 4 MiB L2)
        const bool pair_hashed = dev_switch("SDFX_GRID_PAIR_MIDDLE", 1) == 1;
        uint32_t lo = 0, hi = levels;
        while (lo < hi) {
            --hi;
            const bool lo_dense = lo < hi && (p.lv[lo].flags & 1u) == 0u && p.lv[lo].res >= 2u;
            if (lo < hi && (pair_hashed || lo_dense)) {
                units[nu++] = {hi, pair_tile_cost(p.lv[hi], p.lv[lo], levels, S, H, slabs, step, balance, valu_lines), lo};
                lo++;
            } else {
                units[nu++] = {hi, level_tile_cost(p.lv[hi], levels, S, H, slabs, step, balance, valu_lines), kNoLevel};
            }
        }
    } else if (!(balance && step > 0.f)) {   // no (usable) information: every level costs the same; the order [L-1, 0, L-2, 1, ...] of GridPlan
        for (uint32_t v = 0, lo = 0, hi = levels; v < levels; v++) units[nu++] = {(v & 1u) ? lo++ : --hi, 1.0, kNoLevel};
    } else {                          // fine to coarse; a tile costs its gathers or its VALU work, whichever is longer
        for (uint32_t l = levels; l-- > 0;) units[nu++] = {l, level_tile_cost(p.lv[l], levels, S, H, slabs, step, balance, valu_lines), kNoLevel};
    }

    // measurement aids (devtools build): SDFX_GRID_ONLY_LEVEL = l: the launch evaluates level l alone, spread over all eight XCDs (what
    // a tile of that level costs with the whole GPU on it: tools/xcd_timeline.py); SDFX_GRID_LEVEL_COST = "c0,c1,...": cost per tile by level
    {
        const int only = dev_switch("SDFX_GRID_ONLY_LEVEL", -1);
        if (only >= 0 && (uint32_t)only < levels) { units[0] = {(uint32_t)only, 1.0, kNoLevel}; nu = 1; }
        if (const char* e = dev_string("SDFX_GRID_LEVEL_COST")) {
            double c[kMaxLevels];
            uint32_t n = 0;
            while (*e && n < kMaxLevels) {
                char* end = nullptr;
                const double v = strtod(e, &end);
                if (end == e) break;
                c[n++] = v;
                e = (*end == ',') ? end + 1 : end;
            }
            for (uint32_t u = 0; u < nu; u++) {
                if (units[u].level < n && c[units[u].level] > 0) units[u].cost = c[units[u].level];
                if (units[u].level2 != kNoLevel && units[u].level2 < n && c[units[u].level2] > 0) units[u].cost += c[units[u].level2];
            }
        }
    }

    // ---- cut the sequence into 8 ranges of equal cost ----
    double total = 0;
    for (uint32_t u = 0; u < nu; u++) total += units[u].cost * T;
    // position of boundary k as (unit, column): the same function closes XCD k-1 and opens XCD k
    uint32_t bu[kXcds + 1], bc[kXcds + 1];
    for (uint32_t k = 0; k <= kXcds; k++) {
        const double target = total * k / kXcds;
        double cum = 0;
        uint32_t u = 0;
        while (u + 1 < nu && cum + units[u].cost * T <= target) { cum += units[u].cost * T; u++; }
        double col = (target - cum) / units[u].cost;
        if (k == kXcds) { u = nu - 1; col = T; }
        bu[k] = u;
        bc[k] = col <= 0 ? 0u : (col >= T ? T : (uint32_t)(col + 0.5));
    }
    for (uint32_t k = 0; k < kXcds; k++) {
        uint32_t ns = 0;
        for (uint32_t u = bu[k]; u <= bu[k + 1] && u < nu; u++) {
            const uint32_t first = u == bu[k] ? bc[k] : 0u;
            const uint32_t last = u == bu[k + 1] ? bc[k + 1] : T;
            if (last <= first) continue;
            if (ns == kMaxSegs) { p.ntiles[0] = 0xffffffffu; return p; }   // more levels in one range than a Seg list holds: caller falls back
            // Consecutive tiles per workgroup (round 5): a workgroup's fixed part — kernel arguments, its place in the plan, the
            // level's constants, start and retire — was a third of a 2.4 us coarse tile. 8 tiles at the levels whose cost is the VALU
            // floor, 2 at the others: 601-656 -> 480 us per launch at B = 3.26 M on one box (1 / 1: 601-656; 4 / 1: 519-570; 8 / 1:
            // 501-533; 16 / 1: 507-516; 8 / 2: 480; 16 / 4: 479; profiles/r05_encode_tiles_per_workgroup.txt). SDFX_GRID_TPW /
            // SDFX_GRID_TPW_FINE: the devtools library's knobs for that A/B.
            const uint32_t tpw_coarse = [] { const int v = dev_switch("SDFX_GRID_TPW", 8); return (uint32_t)(v < 1 ? 1 : (v > 64 ? 64 : v)); }();
            const uint32_t tpw_fine = [] { const int v = dev_switch("SDFX_GRID_TPW_FINE", 2); return (uint32_t)(v < 1 ? 1 : (v > 64 ? 64 : v)); }();
            // (an LDS-resident level — the SDFX_GRID_LDS measurement aid — has its own walk of kLdsTiles tiles per workgroup: one tile per plan slot)
            // (curve-ordered batches: 8 tiles per workgroup at every level — 439 / 428 / 420 / 409 us for 1 / 2 / 4 / 8, same process)
            // (pair plans: a workgroup's tile is two levels' worth of work — SDFX_GRID_TPW_PAIR consecutive tiles)
            const uint32_t tpw_pair = [] { const int v = dev_switch("SDFX_GRID_TPW_PAIR", 2); return (uint32_t)(v < 1 ? 1 : (v > 64 ? 64 : v)); }();
            const uint32_t tpw = (p.pair && units[u].level2 != kNoLevel) ? tpw_pair
                                 : ((p.lds_mask >> units[u].level) & 1u) ? 1u
                                 : step < 0.f ? tpw_coarse
                                 : (balance && step > 0.f && lines[units[u].level] <= valu_lines) ? tpw_coarse : tpw_fine;
            p.seg[k][ns++] = {units[u].level, first, last - first, tpw, units[u].level2};
            p.ntiles[k] += (last - first + tpw - 1u) / tpw;
        }
    }
    return p;
}

uint32_t fwd_grid_size(const FwdPlan& p) {
    if (p.sample_major) return ((p.tiles + kXcds - 1) / kXcds) * p.levels * kXcds;
    uint32_t longest = 0;
    for (uint32_t k = 0; k < kXcds; k++) longest = p.ntiles[k] > longest ? p.ntiles[k] : longest;
    return longest * kXcds;
}

// ---- implementation switches: dev_switch (sdfx_common.h) — constants in the product library ----

template <bool HALF>
void launch(const float* inputs, const void* table, void* outputs, uint32_t B, uint32_t L, const FwdPlan& plan, uint32_t gridtype,
            int align_corners, uint32_t interp, int out_layout, hipStream_t st) {
    using T = typename Elem<HALF>::type;
    const uint32_t grid = fwd_grid_size(plan);
    // sdfx_set_row_limit: honoured when its period is this batch's sample axis (computing padding rows anyway is harmless)
    const RowLimit rl = row_limit();
    const uint32_t axis = plan.slabs == kGroup ? plan.slab_points : B;
    const int32_t* row_total = (rl.total && (rl.period == axis || (rl.period == 0 && plan.slabs != kGroup))) ? rl.total : nullptr;
    const StencilSrc src = stencil_src();   // validated by the caller: src.M * 7 == B when set
    dev_ctl_sync();
#define SDFX_FWD(INTERP_, ALIGN_, HASH_)                                                                               \
    do {                                                                                                               \
        if constexpr (HALF) {                                                                                          \
            if (plan.pair) {                                                                                           \
                hipLaunchKernelGGL((k_grid_fwd_pair<INTERP_, ALIGN_, HASH_>), dim3(grid), dim3(kTile), 0, st, inputs,  \
                                   static_cast<const __half*>(table), static_cast<__half*>(outputs), B, L, plan, out_layout, row_total, src); \
                break;                                                                                                 \
            }                                                                                                          \
        }                                                                                                              \
        if (plan.lds_mask)                                                                                             \
            hipLaunchKernelGGL((k_grid_fwd<HALF, INTERP_, ALIGN_, HASH_, true>), dim3(grid), dim3(kTile), plan.lds_bytes, st, inputs,   \
                               static_cast<const T*>(table), static_cast<T*>(outputs), B, L, plan, out_layout, row_total, src);         \
        else                                                                                                           \
            hipLaunchKernelGGL((k_grid_fwd<HALF, INTERP_, ALIGN_, HASH_, false>), dim3(grid), dim3(kTile), 0, st, inputs,               \
                               static_cast<const T*>(table), static_cast<T*>(outputs), B, L, plan, out_layout, row_total, src);         \
    } while (0)
    const int sel = (interp ? 4 : 0) | (align_corners ? 2 : 0) | (gridtype == 0 ? 1 : 0);
    switch (sel) {
        case 0: SDFX_FWD(0u, false, false); break;
        case 1: SDFX_FWD(0u, false, true); break;
        case 2: SDFX_FWD(0u, true, false); break;
        case 3: SDFX_FWD(0u, true, true); break;
        case 4: SDFX_FWD(1u, false, false); break;
        case 5: SDFX_FWD(1u, false, true); break;
        case 6: SDFX_FWD(1u, true, false); break;
        default: SDFX_FWD(1u, true, true); break;
    }
#undef SDFX_FWD
}

}  // namespace

namespace sdfx {
namespace grid {

bool fast_forward_enabled() { return dev_switch("SDFX_GRID_FWD", 1) == 1; }

// D = 3, C = 2, no dy_dx. `slabs`, `step`: locality hints (sdfx_grid_encode_forward_hint); results do not depend on them.
// Returns false when the plan does not fit (more than kMaxSegs levels in one XCD's range): the caller uses k_grid_forward.
bool launch_forward_d3c2(const float* inputs, const void* table, const int32_t* offsets_host, void* outputs, uint32_t B,
                         uint32_t L, uint32_t max_level, float S, uint32_t H, uint32_t gridtype, int align_corners,
                         uint32_t interp, int is_half, int out_layout, uint32_t slabs, float step, hipStream_t st) {
    // SDFX_GRID_VALU_LINES: VALU time of one wave of one level in units of table lines (232 instructions / 2.4 cycles a
    // line = 97; measured optimum of the split on MI355X: 75 -> 315 us, 97 -> 287 us, 120 -> 308 us at B = 1.8 M)
    const double valu_lines = (double)dev_switch("SDFX_GRID_VALU_LINES", 97);
    FwdPlan plan = make_fwd_plan(offsets_host, max_level, S, H, is_half ? 2u : 4u, B, slabs, step,
                                 dev_switch("SDFX_GRID_BALANCE", 1) == 1, valu_lines);
    if (plan.ntiles[0] == 0xffffffffu) return false;
    for (uint32_t l = 0; l < max_level; l++)   // the kernel addresses a level's rows by 32-bit byte offsets
        if ((uint64_t)plan.lv[l].size * (is_half ? 4u : 8u) >= (1ull << 32)) return false;
    plan.vec16 = (reinterpret_cast<uintptr_t>(table) % 16) == 0 ? 0xffffffffu : 0u;
    {   // SDFX_GRID_SCALAR_BELOW = r (measurement aid): levels of resolution < r gather every corner with its own 4-byte load
        const uint32_t r = (uint32_t)dev_switch("SDFX_GRID_SCALAR_BELOW", 0);
        const uint32_t from = (uint32_t)dev_switch("SDFX_GRID_SCALAR_FROM", 0);   // ... and levels of resolution >= from (0: none)
        for (uint32_t l = 0; l < max_level; l++)
            if (plan.lv[l].res < r || (from && plan.lv[l].res >= from)) plan.vec16 &= ~(1u << l);
    }
    if (dev_switch("SDFX_GRID_PLAN_DEBUG", 0)) {   // one line per XCD: its segments
        for (uint32_t k = 0; k < kXcds; k++) {
            fprintf(stderr, "[grid plan] B=%u slabs=%u step=%g xcd %u: %u workgroups:", B, plan.slabs, (double)step, k, plan.ntiles[k]);
            for (uint32_t sgi = 0; sgi < kMaxSegs; sgi++) {
                const Seg& sg = plan.seg[k][sgi];
                if (sg.count && sg.level2 != kNoLevel) fprintf(stderr, "  L%u+L%u[%u,+%u)", sg.level, sg.level2, sg.first, sg.count);
                else if (sg.count) fprintf(stderr, "  L%u[%u,+%u)", sg.level, sg.first, sg.count);
            }
            fprintf(stderr, "\n");
        }
    }
    if (is_half) launch<true>(inputs, table, outputs, B, L, plan, gridtype, align_corners, interp, out_layout, st);
    else launch<false>(inputs, table, outputs, B, L, plan, gridtype, align_corners, interp, out_layout, st);
    return true;
}

}  // namespace grid
}  // namespace sdfx

// Host-only: the cost per tile the plan prices level l at (costs[l], l < max_level; the unit is one table line looked up by a wave) —
// the measured table for the configuration it was measured on, the model max(lines, VALU floor) otherwise, 1.0 without a step hint.
extern "C" int sdfx_grid_forward_level_costs(const int32_t* offsets_host, uint32_t max_level, float S, uint32_t H, uint32_t slabs,
                                             float step, double* costs) {
    if (!offsets_host || !costs || max_level < 1 || max_level > kMaxLevels) return -1;
    const double valu_lines = (double)dev_switch("SDFX_GRID_VALU_LINES", 97);
    const bool balance = dev_switch("SDFX_GRID_BALANCE", 1) == 1;
    for (uint32_t l = 0; l < max_level; l++) {
        const LevelConst c = make_level_const(offsets_host, l, S, H);
        costs[l] = level_tile_cost(c, max_level, S, H, slabs, step, balance, valu_lines);
    }
    // a pair plan (half tables) prices PAIRS: the fine level of a pair is listed at the pair's price less its partner's, so that
    // sum(costs[l] x tiles) over an XCD's segments is what the plan balanced
    if (pair_plan_enabled(2u)) {
        for (uint32_t lo = 0, hi = max_level; lo + 1 < hi; lo++) {
            --hi;
            costs[hi] = pair_tile_cost(make_level_const(offsets_host, hi, S, H), make_level_const(offsets_host, lo, S, H), max_level, S, H, slabs, step,
                                       balance, valu_lines) - costs[lo];
        }
    }
    return (int)max_level;
}

// Host-only: the per-XCD work list the forward would use, 4 integers per segment (xcd, level, first tile, tiles);
// returns the number of segments (<= max_segments), tiles per level in *tiles_per_level.
extern "C" int sdfx_grid_forward_plan(const int32_t* offsets_host, uint32_t max_level, float S, uint32_t H, int is_half, uint32_t B,
                                      uint32_t slabs, float step, int32_t* segments, uint32_t max_segments,
                                      uint32_t* tiles_per_level) {
    if (!offsets_host || !segments || max_level < 1 || max_level > kMaxLevels || B == 0) return -1;
    const FwdPlan plan = make_fwd_plan(offsets_host, max_level, S, H, is_half ? 2u : 4u, B, slabs, step,
                                       dev_switch("SDFX_GRID_BALANCE", 1) == 1, (double)dev_switch("SDFX_GRID_VALU_LINES", 97));
    if (plan.ntiles[0] == 0xffffffffu) return -3;
    const uint64_t slots = plan.slabs == kGroup ? (uint64_t)div_up(plan.slab_points, kGroupsPerWave) * 64u : B;
    if (tiles_per_level) *tiles_per_level = div_up(slots, (uint64_t)kTile);
    uint32_t n = 0;
    for (uint32_t k = 0; k < kXcds; k++) {
        for (uint32_t sgi = 0; sgi < kMaxSegs; sgi++) {
            const Seg& sg = plan.seg[k][sgi];
            if (!sg.count) continue;
            if (n == max_segments) return -2;
            int32_t* o = segments + 4 * n++;
            o[0] = (int32_t)k; o[1] = (int32_t)sg.level; o[2] = (int32_t)sg.first; o[3] = (int32_t)sg.count;
            if (sg.level2 != kNoLevel) {   // a pair plan's segment: the same tiles at the partner level
                if (n == max_segments) return -2;
                int32_t* o2 = segments + 4 * n++;
                o2[0] = (int32_t)k; o2[1] = (int32_t)sg.level2; o2[2] = (int32_t)sg.first; o2[3] = (int32_t)sg.count;
            }
        }
    }
    return (int)n;
}


